"""Thin host-side driver of libquber_hip.so: owns one ``quber_ctx`` and hands torch (ROCm) tensors'
device pointers and the current HIP stream to the C ABI.  torch is plumbing only (device memory,
streams); every computation on the hot path happens in the HIP library.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from . import arch
from .config import arch_kwargs


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def set_arch(qc, depth=50, backbone_fusion_layers=2, head_fusion_layers=3, error_classes=4, eee_mask_on=False,
             eee_boundary_on=True, hierarchical=True, hierarchy=arch.DEFAULT_HIERARCHY, fusion_target=("feat", "pred"),
             streams=2, fusion_add=False, convs_dim=128, head_channels=32):
    """Write the architecture switches (keyword arguments of arch.param_specs) into a quber_config."""
    qc.resnet_depth, qc.backbone_fusion_layers, qc.head_fusion_layers = depth, backbone_fusion_layers, head_fusion_layers
    qc.error_classes, qc.eee_mask_on, qc.eee_boundary_on = error_classes, int(eee_mask_on), int(eee_boundary_on)
    qc.hierarchical = int(hierarchical)
    qc.streams = streams
    qc.fusion_add = int(fusion_add)
    qc.convs_dim, qc.head_channels = int(convs_dim), int(head_channels)
    qc.fusion_feat, qc.fusion_pred = int("feat" in fusion_target), int("pred" in fusion_target)
    levels = list(hierarchy) if hierarchical else []
    qc.n_levels = len(levels)
    for i in range(5):
        for j in range(5):
            qc.level_heads[i][j] = -1
    for i, lvl in enumerate(levels):
        for j, k in enumerate(lvl):
            qc.level_heads[i][j] = arch.HEADS.index(k)
    return qc


# predictor.py:345-346 subtracts a float32 index array from a float64 scalar.  numpy < 2 (the reference pins 1.23.1,
# INSTALL.md:14) keeps that in float32 (value-based casting); numpy >= 2 promotes to float64.  The drop-in reproduces what
# the reference computes UNDER THE NUMPY OF THE CALLING PROCESS - so that swapping the predictor in changes no bit of a1 in
# either environment; QUBER_ENCODE_LEGACY_F32=0/1 overrides.  (C default: 0, the mode the golden fixtures pin.)
ENCODE_LEGACY_F32 = int(os.environ.get("QUBER_ENCODE_LEGACY_F32", int(np.lib.NumpyVersion(np.__version__) < "2.0.0")))


def make_config(height=480, width=640, max_batch=1, max_instances=64, cfg=None, with_network=True):
    """quber_config from an (optional) validated yaml CfgNode."""
    lib = _lib.load()
    qc = _lib.QuberConfig()
    lib.quber_default_config(C.byref(qc))
    qc.encode_legacy_f32 = ENCODE_LEGACY_F32
    qc.height, qc.width, qc.max_batch, qc.max_instances = height, width, max_batch, max_instances
    qc.with_network = 1 if with_network else 0
    if cfg is not None:
        m = cfg.MODEL
        set_arch(qc, **arch_kwargs(cfg))
        qc.res5_dilation = m.RESNETS.RES5_DILATION
        qc.gaussian_sigma = cfg.INPUT.get("GAUSSIAN_SIGMA", 10)
        qc.nms_kernel = m.PANOPTIC_DEEPLAB.NMS_KERNEL
        qc.top_k = m.PANOPTIC_DEEPLAB.TOP_K_INSTANCE
        qc.stuff_area = m.PANOPTIC_DEEPLAB.STUFF_AREA
        qc.center_threshold = m.PANOPTIC_DEEPLAB.CENTER_THRESHOLD
        for i in range(min(6, len(m.PIXEL_MEAN))):
            qc.pixel_mean[i] = float(m.PIXEL_MEAN[i])
            qc.pixel_std[i] = float(m.PIXEL_STD[i])
    return qc


def normalize_depth(depth, min_val=250.0, max_val=1500.0):
    """depth: device tensor uint16 / int16-viewed-as-uint16 / float32 [..., H, W] -> (u8 [..., H, W, 3], zero mask u8 [..., H, W]).
    Device-side eval/preprocess_utils.py:12-28."""
    lib = _lib.load()
    assert depth.is_cuda and depth.is_contiguous()
    if depth.dtype == torch.float32:
        is_float = 1
    elif depth.dtype in (torch.uint16, torch.int16):
        is_float = 0
    else:
        raise TypeError("normalize_depth expects uint16 or float32 depth")
    out = torch.empty(tuple(depth.shape) + (3,), dtype=torch.uint8, device=depth.device)
    zero = torch.empty(tuple(depth.shape), dtype=torch.uint8, device=depth.device)
    _lib.check(lib.quber_normalize_depth(_ptr(depth), is_float, depth.numel(), float(min_val), float(max_val), _ptr(out),
                                         _ptr(zero), _stream()))
    return out, zero


def resize_u8(img, dh, dw, linear=True):
    """cv2.resize(img, (dw, dh), interpolation=INTER_LINEAR | INTER_NEAREST) on the device: img u8 [H,W] or [H,W,C] tensor."""
    lib = _lib.load()
    assert img.is_cuda and img.dtype == torch.uint8 and img.is_contiguous() and img.dim() in (2, 3)
    ch = 1 if img.dim() == 2 else img.shape[2]
    out = torch.empty((dh, dw) + (() if img.dim() == 2 else (ch,)), dtype=torch.uint8, device=img.device)
    _lib.check(lib.quber_resize_u8(_ptr(img), img.shape[0], img.shape[1], ch, _ptr(out), dh, dw, int(bool(linear)), _stream()))
    return out


_INPAINT_WS = {}
_INPAINT_WS_MAX = 32       # ~22 MB each at 640x480


def inpaint_depth(depth3, kernel_size=3):
    """inpaint_depth of eval/preprocess_utils.py:44-64 on the device (csrc/inpaint_dev.hip; bit-equal to the host restatement
    quber_inpaint_depth_u8): depth3 u8 [H,W,3] or [B,H,W,3] device tensor -> the same shape.  Asynchronous on the current stream."""
    lib = _lib.load()
    assert depth3.is_cuda and depth3.dtype == torch.uint8 and depth3.is_contiguous() and depth3.shape[-1] == 3 and depth3.dim() in (3, 4)
    B = 1 if depth3.dim() == 3 else depth3.shape[0]
    H, W = depth3.shape[-3], depth3.shape[-2]
    need = lib.quber_inpaint_depth_workspace_bytes(B, H, W)
    import threading
    # one workspace per (stream, calling thread): the launches of one call are ordered on their stream, but two threads that share
    # a stream interleave theirs
    key = (depth3.device, torch.cuda.current_stream().cuda_stream, threading.get_ident())
    ws = _INPAINT_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=depth3.device)
        _INPAINT_WS.pop(key, None)
        while len(_INPAINT_WS) >= _INPAINT_WS_MAX:          # worker threads and side streams come and go (predict_stream): oldest first
            _INPAINT_WS.pop(next(iter(_INPAINT_WS)))
        _INPAINT_WS[key] = ws
    out = torch.empty_like(depth3)
    _lib.check(lib.quber_inpaint_depth_device(_ptr(depth3), B, H, W, int(kernel_size), _ptr(ws), ws.numel(), _ptr(out), _stream()))
    return out


class Engine:
    """One context on the current device.  All methods are asynchronous on torch's current stream."""

    def __init__(self, qcfg, device=None):
        if not torch.cuda.is_available():
            raise _lib.QuberError("quber_amd needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.qcfg = qcfg
        self.H, self.W = qcfg.height, qcfg.width
        self.planes = 4 + qcfg.error_classes * (int(bool(qcfg.eee_mask_on)) + int(bool(qcfg.eee_boundary_on)))
        self.cap = qcfg.top_k
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.quber_create(C.byref(qcfg), C.byref(h)))
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize(self.device)
            self.lib.quber_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- options (include/quber_hip.h: the keys of quber_set_option) ----
    def set_option(self, key, value):
        """This engine only.  Plan-time keys must be set before load_state_dict(); launch-time keys any time."""
        _lib.check(self.lib.quber_set_option(self.h, int(key), int(value)))

    def get_option(self, key):
        v = C.c_int32()
        _lib.check(self.lib.quber_get_option(self.h, int(key), C.byref(v)))
        return v.value

    # ---- weights ----
    def weight_specs(self):
        out = []
        name, numel = C.c_char_p(), C.c_int64()
        for i in range(self.lib.quber_num_weights(self.h)):
            _lib.check(self.lib.quber_weight_spec(self.h, i, C.byref(name), C.byref(numel)))
            out.append((name.value.decode(), numel.value))
        return out

    def load_state_dict(self, sd):
        """sd: name -> numpy / torch array in torch layout (detectron2-compatible keys)."""
        for name, numel in self.weight_specs():
            if name not in sd:
                raise KeyError(f"state_dict lacks '{name}'")
            v = sd[name]
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            v = np.ascontiguousarray(v, dtype=np.float32)
            if v.size != numel:
                raise ValueError(f"'{name}' has {v.size} elements, expected {numel}")
            _lib.check(self.lib.quber_set_weight(self.h, name.encode(), C.c_void_p(v.ctypes.data), v.size))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.quber_finalize_weights(self.h))

    def plan(self):
        """[(name, kind, flops at batch 1, launches)] of quber_forward in execution order."""
        out = []
        name, kind, fl, ln = C.c_char_p(), C.c_int32(), C.c_double(), C.c_int32()
        for i in range(self.lib.quber_num_ops(self.h)):
            _lib.check(self.lib.quber_op_info(self.h, i, C.byref(name), C.byref(kind), C.byref(fl), C.byref(ln)))
            out.append((name.value.decode(), ("conv", "norm", "other")[kind.value], fl.value, ln.value))
        return out

    def forward_flops(self):
        return self.lib.quber_forward_flops(self.h)

    def forward_flops_executed(self):
        return self.lib.quber_forward_flops_executed(self.h)

    def forward_flops_padding(self):
        """The part of forward_flops_executed() spent on the padding of Winograd tiles."""
        return self.lib.quber_forward_flops_padding(self.h)

    # ---- hot path ----
    def encode(self, masks, out=None):
        """masks u8 [B,N,H,W] (device) -> f32 [B,3,H,W]."""
        B, N = masks.shape[:2]
        assert masks.dtype == torch.uint8 and masks.is_contiguous() and masks.shape[2:] == (self.H, self.W)
        if out is None:
            out = torch.empty((B, 3, self.H, self.W), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.quber_encode_initial_masks(self.h, _ptr(masks), B, N, _ptr(out), _stream()))
        return out

    def encode_label_map(self, labels, n, out=None):
        """labels i32 [B,H,W] (device; 0 = none, 1..n = instance) -> f32 [B,3,H,W], as encode() on the masks (labels == i + 1)."""
        B = labels.shape[0]
        assert labels.dtype == torch.int32 and labels.is_contiguous() and labels.shape[1:] == (self.H, self.W)
        if out is None:
            out = torch.empty((B, 3, self.H, self.W), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.quber_encode_label_map(self.h, _ptr(labels), B, n, _ptr(out), _stream()))
        return out

    def workspace_bytes(self):
        return self.lib.quber_workspace_bytes(self.h)

    def error_maps(self, init_masks, gt_masks, out=None):
        B, N = init_masks.shape[:2]
        Ng = gt_masks.shape[1]
        assert init_masks.dtype == torch.uint8 and gt_masks.dtype == torch.uint8
        assert init_masks.is_contiguous() and gt_masks.is_contiguous()
        if out is None:
            out = torch.empty((B, 2, 4, self.H, self.W), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.quber_explicit_error_maps(self.h, _ptr(init_masks), N, _ptr(gt_masks), Ng, B, _ptr(out),
                                                      _stream()))
        return out

    def forward(self, bgr, depth, offsets, out=None):
        """bgr, depth u8 [B,H,W,3] (depth None for a single-stream model: `bgr` is then THE image, rgb or depth);
        offsets f32 [B,3,H,W] -> logits f32 [B,planes,H,W]."""
        B = bgr.shape[0]
        assert bgr.dtype == torch.uint8 and offsets.dtype == torch.float32
        assert bgr.shape == (B, self.H, self.W, 3) and offsets.shape == (B, 3, self.H, self.W)
        assert bgr.is_contiguous() and offsets.is_contiguous()
        if self.qcfg.streams == 2:
            assert depth is not None and depth.dtype == torch.uint8 and depth.shape == bgr.shape and depth.is_contiguous()
        else:
            depth = None
        if out is None:
            out = torch.empty((B, self.planes, self.H, self.W), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.quber_forward(self.h, _ptr(bgr), _ptr(depth), _ptr(offsets), B, _ptr(out), _stream()))
        return out

    def forward_profiled(self, bgr, depth, offsets, out=None):
        """forward with HIP events around every launch group -> (logits, {kind: (ms, launches)})."""
        B = bgr.shape[0]
        if out is None:
            out = torch.empty((B, self.planes, self.H, self.W), dtype=torch.float32, device=self.device)
        ms, cnt = (C.c_double * 3)(), (C.c_int32 * 3)()
        _lib.check(self.lib.quber_forward_profiled(self.h, _ptr(bgr), _ptr(depth), _ptr(offsets), B, _ptr(out),
                                                   _stream(), C.byref(ms), C.byref(cnt)))
        return out, {k: (ms[i], cnt[i]) for i, k in enumerate(("conv", "norm", "other"))}

    def profile_begin(self):
        """Start a stage profile: every call on this engine until profile_end() brackets its kernels with HIP events."""
        _lib.check(self.lib.quber_profile_begin(self.h))

    def profile_end(self):
        """-> {stage: dict(ms, bytes, flops, launches)}; synchronises the current stream."""
        _lib.check(self.lib.quber_profile_end(self.h, _stream()))
        out = {}
        name, ms, by, fl, ln = C.c_char_p(), C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        for i in range(self.lib.quber_profile_num_stages(self.h)):
            _lib.check(self.lib.quber_profile_stage(self.h, i, C.byref(name), C.byref(ms), C.byref(by), C.byref(fl), C.byref(ln)))
            out[name.value.decode()] = {"ms": ms.value, "bytes": by.value, "flops": fl.value, "launches": ln.value}
        return out

    def alloc_post(self, B):
        d, cap = self.device, self.cap
        return {
            "panoptic": torch.empty((B, self.H, self.W), dtype=torch.float32, device=d),
            "count": torch.empty((B,), dtype=torch.int32, device=d),
            "labels": torch.empty((B, cap), dtype=torch.float32, device=d),
            "scores": torch.empty((B, cap), dtype=torch.float32, device=d),
            "boxes": torch.empty((B, cap, 4), dtype=torch.float32, device=d),
            "centers": torch.zeros((B, cap, 2), dtype=torch.int32, device=d),
            "ncenters": torch.empty((B,), dtype=torch.int32, device=d),
        }

    def postprocess(self, logits, out=None):
        B, planes = logits.shape[:2]
        assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.shape[2:] == (self.H, self.W)
        o = out if out is not None else self.alloc_post(B)
        _lib.check(self.lib.quber_postprocess(
            self.h, _ptr(logits), planes, B, _ptr(o["panoptic"]), _ptr(o["count"]), _ptr(o["labels"]),
            _ptr(o["scores"]), _ptr(o["boxes"]), _ptr(o["centers"]), _ptr(o["ncenters"]), _stream()))
        return o

    def extract_masks(self, post, max_inst, out=None):
        B = post["panoptic"].shape[0]
        if out is None:
            out = torch.empty((B, max_inst, self.H, self.W), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.quber_extract_masks(self.h, _ptr(post["panoptic"]), _ptr(post["labels"]), B, max_inst,
                                                _ptr(out), _stream()))
        return out

    def debug_tensor(self, name, batch):
        """NHWC copy [batch,H,W,C] of a named intermediate of the last forward."""
        p, dims, cs = C.c_void_p(), (C.c_int32 * 4)(), C.c_int32()
        _lib.check(self.lib.quber_debug_tensor(self.h, name.encode(), C.byref(p), C.byref(dims), C.byref(cs)))
        Bm, H, W, Cc = list(dims)
        es = self.lib.quber_debug_tensor_elem_size(self.h, name.encode())       # 2: the fp16 data path stores activations as fp16
        n = (batch * H * W - 1) * cs.value + Cc        # the view may be a channel slice of a wider buffer
        buf = torch.empty(n, dtype=torch.float16 if es == 2 else torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(C.c_void_p(buf.data_ptr()), p, n * es, 3)
        if rc != 0:
            raise _lib.QuberError(f"hipMemcpy failed ({rc})")
        return torch.as_strided(buf, (batch, H, W, Cc), (H * W * cs.value, W * cs.value, cs.value, 1)).float()
