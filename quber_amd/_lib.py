"""ctypes binding of libquber_hip.so (include/quber_hip.h).  No fallback: if the library is missing or
cannot be loaded the import of anything that needs it raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QUBER_LIB: another build of the same library (diagnostic builds with in-kernel time stamps live in a scratch directory,
# tools/diag_build.sh - never in quber_amd/csrc); the default is the in-tree product library
LIB_PATH = os.environ.get("QUBER_LIB") or os.path.join(_HERE, "libquber_hip.so")


class QuberConfig(C.Structure):
    _fields_ = [
        ("height", C.c_int32), ("width", C.c_int32), ("max_batch", C.c_int32), ("max_instances", C.c_int32),
        ("resnet_depth", C.c_int32), ("res5_dilation", C.c_int32), ("backbone_fusion_layers", C.c_int32),
        ("head_fusion_layers", C.c_int32), ("error_classes", C.c_int32), ("gaussian_sigma", C.c_int32),
        ("nms_kernel", C.c_int32), ("top_k", C.c_int32), ("stuff_area", C.c_int32),
        ("min_instance_area", C.c_int32), ("label_divisor", C.c_int32), ("with_network", C.c_int32),
        ("center_threshold", C.c_float), ("boundary_ratio", C.c_float),
        ("pixel_mean", C.c_float * 6), ("pixel_std", C.c_float * 6),
        ("eee_mask_on", C.c_int32), ("eee_boundary_on", C.c_int32), ("hierarchical", C.c_int32),
        ("fusion_feat", C.c_int32), ("fusion_pred", C.c_int32), ("n_levels", C.c_int32),
        ("level_heads", (C.c_int32 * 5) * 5), ("fusion_add", C.c_int32), ("streams", C.c_int32),
        ("compute_dtype", C.c_int32), ("encode_legacy_f32", C.c_int32),
        ("convs_dim", C.c_int32), ("head_channels", C.c_int32),
    ]


class QuberError(RuntimeError):
    pass


_P = C.c_void_p
_I = C.c_int32
# name -> (restype, argtypes); every symbol include/quber_hip.h declares
SIGNATURES = {
    "quber_default_config": (None, [C.POINTER(QuberConfig)]),
    "quber_last_error": (C.c_char_p, []),
    "quber_version": (C.c_char_p, []),
    "quber_create": (C.c_int, [C.POINTER(QuberConfig), C.POINTER(_P)]),
    "quber_destroy": (None, [_P]),
    "quber_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "quber_finalize_weights": (C.c_int, [_P]),
    "quber_num_weights": (C.c_int, [_P]),
    "quber_weight_spec": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]),
    "quber_encode_initial_masks": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "quber_encode_label_map": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "quber_workspace_bytes": (C.c_int64, [_P]),
    "quber_explicit_error_maps": (C.c_int, [_P, _P, _I, _P, _I, _I, _P, _P]),
    "quber_forward": (C.c_int, [_P, _P, _P, _P, _I, _P, _P]),
    "quber_forward_profiled": (C.c_int, [_P, _P, _P, _P, _I, _P, _P, C.POINTER(C.c_double * 3), C.POINTER(_I * 3)]),
    "quber_profile_begin": (C.c_int, [_P]),
    "quber_profile_end": (C.c_int, [_P, _P]),
    "quber_profile_num_stages": (C.c_int, [_P]),
    "quber_profile_stage": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(_I)]),
    "quber_postprocess": (C.c_int, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "quber_extract_masks": (C.c_int, [_P, _P, _P, _I, _I, _P, _P]),
    "quber_contingency_workspace_bytes": (C.c_int64, [_I]),
    "quber_label_contingency": (C.c_int, [_P, _P, C.c_int64, _I, _P, _P]),
    "quber_boundary_workspace_bytes": (C.c_int64, [_I, _I, _I]),
    "quber_boundary_overlap": (C.c_int, [_P, _P, _I, _I, _P, _I, _I, _I, _P, _P, _P]),
    "quber_foreground_filter": (C.c_int, [_P, _I, _I, _P, _I, _I, C.c_int64, _P, _P, _P]),
    "quber_normalize_depth": (C.c_int, [_P, _I, C.c_int64, C.c_double, C.c_double, _P, _P, _P]),
    "quber_inpaint_telea_u8": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "quber_inpaint_depth_u8": (C.c_int, [_P, _I, _I, _I, _P]),
    "quber_inpaint_depth_workspace_bytes": (C.c_int64, [_I, _I, _I]),
    "quber_inpaint_depth_device": (C.c_int, [_P, _I, _I, _I, _I, _P, C.c_int64, _P, _P]),
    "quber_resize_u8": (C.c_int, [_P, _I, _I, _I, _P, _I, _I, _I, _P]),
    "quber_debug_tensor": (C.c_int, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(_I * 4), C.POINTER(_I)]),
    "quber_debug_tensor_elem_size": (C.c_int32, [_P, C.c_char_p]),
    "quber_forward_flops": (C.c_double, [_P]),
    "quber_forward_flops_executed": (C.c_double, [_P]),
    "quber_forward_flops_padding": (C.c_double, [_P]),
    "quber_set_tuning": (None, [_I, _I]),
    "quber_set_option": (C.c_int, [_P, _I, _I]),
    "quber_get_option": (C.c_int, [_P, _I, C.POINTER(_I)]),
    "quber_num_ops": (C.c_int, [_P]),
    "quber_op_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(_I), C.POINTER(C.c_double), C.POINTER(_I)]),
    "quber_debug_persistent_segments": (C.c_int32, [_I, _I, _I, _I, _I, _P, _I]),
    "quber_debug_persistent_fixup": (C.c_int32, [_I, _I, _I, _I, _I, _I, _P, _P, _I]),
    "quber_op_conv1x1_f16": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P, _P]),
    "quber_op_conv2d_f16": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _P]),
    "quber_op_conv1x1_dual": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P]),
    "quber_op_conv2d": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
    "quber_op_conv3x3_winograd": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _I, _P, _P, C.c_int64, _P, _P]),
    "quber_op_groupnorm": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, C.c_float, _I, _P, _P, _P]),
    "quber_op_bilinear": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "quber_op_maxpool3x3s2": (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "quber_op_group_pixels": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QuberError(
            f"{LIB_PATH} not found: build it with `make -C quber_amd/csrc` (or __graft_entry__.build()); "
            "quber_amd has no CPU fallback")
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so); this library is linked against the system one by soname.
    # Whichever is loaded first serves both - but loaded in the other order the process ends up with two runtimes and the
    # second sees no device ("no HIP device available" from quber_create after __graft_entry__.build() had loaded this library
    # before anything imported torch).  The host side is torch-based anyway: make the order deterministic.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # QUBER_WINOGRAD = auto (default) | f6 | f4 | f2 | off : the Winograd tiles the wide 3x3 layers may use.  Each layer's
    # algorithm is fixed when the plan is built, from its geometry only (never from the batch).  auto = F(4x4,3x3), or
    # F(2x2) where its tiles fit the map better: both measure the direct kernel's error against float64 at tap level
    # (profiles/r02a_parity_report.txt).  "f6" opts into F(6x6,3x3) where it multiplies >= 10 % less still: 2.5x the
    # tap-level error (still 5x inside the 1e-4 bar), +4.5 % throughput at batch 16.  "f2" = 0.8x the throughput,
    # "off" runs every layer as a plain implicit GEMM.
    mode = os.environ.get("QUBER_WINOGRAD", "auto").lower()
    if mode not in ("auto", "f6", "f4", "f2", "off"):
        raise QuberError(f"QUBER_WINOGRAD={mode!r}: expected auto, f6, f4, f2 or off")
    lib.quber_set_tuning(6, 1 if mode == "off" else 0)
    lib.quber_set_tuning(9, {"f2": 2, "f4": 4, "f6": 6}.get(mode, 0))
    # QUBER_PERSIST = 0 | 1: persistent convolution launches (csrc/conv_persist.hip, tuning key 13)
    if "QUBER_PERSIST" in os.environ:
        lib.quber_set_tuning(13, int(os.environ["QUBER_PERSIST"]))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise QuberError(load().quber_last_error().decode())
