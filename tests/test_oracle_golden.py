"""CPU: the oracle restatements against the fixtures produced from the imported reference modules."""
import os

import numpy as np
import pytest
import torch

from conftest import golden, load_encode_case
from oracle import adapter_np, encode_np, errmaps_np, postproc_ref


@pytest.mark.parametrize("path", golden("encode"), ids=os.path.basename)
def test_encode_matches_reference(path):
    masks, out, fg = load_encode_case(path)
    got = encode_np.encode_initial_masks(masks)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got.view(np.uint32), out.view(np.uint32))
    np.testing.assert_array_equal(errmaps_np.masks_to_fg_mask(masks), fg)


def test_fg_union_wraps_like_reference():
    z = np.load(golden("fgunion")[0])
    np.testing.assert_array_equal(errmaps_np.masks_to_fg_mask(z["masks"]), z["fg"])
    assert z["fg"][3, 4] == 0 and z["fg"][6, 10] == 1     # 256 x 255 wraps to 0; 100 x 255 does not


@pytest.mark.parametrize("path", golden("centers"), ids=os.path.basename)
def test_centers(path):
    z = np.load(path)
    got = postproc_ref.find_centers(torch.from_numpy(z["center"]))
    np.testing.assert_array_equal(got.numpy(), z["out"])


@pytest.mark.parametrize("path", golden("group"), ids=os.path.basename)
def test_group(path):
    z = np.load(path)
    got = postproc_ref.group_pixels(torch.from_numpy(z["centers"]), torch.from_numpy(z["offsets"]), chunk=4099)
    np.testing.assert_array_equal(got.numpy().astype(np.int32), z["out"])


@pytest.mark.parametrize("path", golden("panoptic"), ids=os.path.basename)
def test_panoptic(path):
    z = np.load(path)
    lg = torch.from_numpy(z["fg_logit"])
    fg = lg.sigmoid().round()
    np.testing.assert_array_equal(fg.numpy(), z["fg"])
    pan, ctr = postproc_ref.panoptic(fg, torch.from_numpy(z["center"]), torch.from_numpy(z["offsets"]))
    assert pan.dtype == torch.float32
    np.testing.assert_array_equal(pan.numpy(), z["pan"])
    np.testing.assert_array_equal(ctr.numpy(), z["centers"][0])


def test_boundary_hand_derived():
    # rectangle well inside: band of width d on the inside; touching the border counts as boundary
    h, w = 60, 80
    m = np.zeros((h, w), np.uint8)
    m[10:40, 20:70] = 255
    b = errmaps_np.mask_to_boundary(m, dilation_ratio=0.05)      # d = round(0.05*100) = 5
    exp = m.copy()
    exp[15:35, 25:65] = 0
    np.testing.assert_array_equal(b, exp)
    m2 = np.zeros((h, w), np.uint8)
    m2[0:20, 0:30] = 1
    b2 = errmaps_np.mask_to_boundary(m2, dilation_ratio=0.03)    # d = 3
    exp2 = m2.copy()
    exp2[3:17, 3:27] = 0
    np.testing.assert_array_equal(b2, exp2)
    # thin mask vanishes entirely under erosion -> whole mask is boundary
    m3 = np.zeros((h, w), np.uint8)
    m3[30:34, 5:60] = 1
    np.testing.assert_array_equal(errmaps_np.mask_to_boundary(m3, 0.03), m3)
    assert errmaps_np.boundary_width(480, 640, 0.01) == 8
    assert errmaps_np.boundary_width(720, 1280, 0.01) == 15


def test_boundary_band_against_scipy_morphology():
    """The boundary half of a2 (util.py:72-90) has no runnable reference here (cv2 absent).  An INDEPENDENT implementation of the same
    published operation - binary erosion by a 3x3 square, d iterations, everything outside the image background: what cv2.erode
    computes on the mask inside its one-pixel ring of zeros, the ring staying zero through every iteration - is scipy.ndimage's
    binary_erosion(border_value=0).  Irregular synthetic instance masks (truncated by the frame border, holes, thin parts), several
    sizes and ratios; a cross-check of the restatement, not a pin."""
    from scipy import ndimage
    from quber_amd import synth
    for seed, (h, w), ratio in ((1, (96, 128), 0.01), (2, (96, 128), 0.02), (3, (120, 160), 0.01), (4, (480, 640), 0.01)):
        sc = synth.make_scene(seed, h, w, 6)
        d = errmaps_np.boundary_width(h, w, ratio)
        for m in sc["masks"]:
            m = (m > 0).astype(np.uint8)
            m[h // 3:h // 3 + 3, w // 4:w // 4 + 3] = 0                      # a hole
            er = ndimage.binary_erosion(m, structure=np.ones((3, 3), bool), iterations=d, border_value=0).astype(np.uint8)
            np.testing.assert_array_equal(errmaps_np.mask_to_boundary(m, ratio), m - er)
    full = np.ones((40, 50), np.uint8)                                       # a mask that fills the frame: the band hugs the border
    np.testing.assert_array_equal(errmaps_np.mask_to_boundary(full, 0.05),
                                  full - ndimage.binary_erosion(full, structure=np.ones((3, 3), bool), iterations=3, border_value=0))


def test_quadruple_is_one_hot():
    rng = np.random.default_rng(0)
    from quber_amd import synth
    gt, init = synth.make_masks(rng, 6, 96, 128)
    e = errmaps_np.explicit_error_maps(init.astype(np.uint8), gt.astype(np.uint8))
    assert e.shape == (2, 4, 96, 128)
    np.testing.assert_array_equal(e.sum(1), np.ones((2, 96, 128), np.uint8))


@pytest.mark.parametrize("path", golden("depthnorm"), ids=os.path.basename)
def test_normalize_depth(path):
    z = np.load(path)
    got = adapter_np.normalize_depth(z["depth"], float(z["lo"]), float(z["hi"]))
    np.testing.assert_array_equal(got, z["out"])


@pytest.mark.parametrize("path", golden("lmffnet"), ids=os.path.basename)
def test_lmffnet_oracle_matches_reference(path):
    from oracle import lmffnet_torch as L
    from quber_amd import lmff_arch
    z = np.load(path)
    w = {k: torch.from_numpy(v) for k, v in lmff_arch.init_state_dict(seed=int(z["seed"])).items()}
    with torch.no_grad():
        got = L.forward(L.preprocess(z["rgb"], z["depth"]), w)[0].numpy()
    np.testing.assert_allclose(got, z["logits"], rtol=0, atol=1e-5)
    assert len(w) == 402 and sum(v.numel() for v in w.values()) == 1356859 - 65     # the reference state_dict minus its 65 num_batches_tracked counters


def _same_metrics(got, exp):
    assert set(got) == set(exp)
    for k, v in exp.items():
        if v is None:
            assert got[k] is None, k
        else:
            assert float(got[k]) == v, (k, got[k], v)


@pytest.mark.parametrize("path", golden("metrics"), ids=os.path.basename)
def test_metrics_oracle_matches_reference(path):
    import json
    from oracle import metrics_np
    z = np.load(path)
    _same_metrics(metrics_np.multilabel_metrics(z["pred"], z["gt"]), json.loads(str(z["result"])))


def test_munkres_matches_vendored_solver():
    import json
    from conftest import GOLDEN
    from oracle.assign_py import assign as oracle_assign            # the oracle's own scalar-loop solver
    from quber_amd.eval.assignment import munkres_assign           # the product's numpy-mask solver
    z = np.load(os.path.join(GOLDEN, "munkres_cases.npz"))
    exp = json.load(open(os.path.join(GOLDEN, "munkres_expected.json")))
    for k in z.files:
        m = z[k]
        assert [list(a) for a in munkres_assign(m.max() - m)] == exp[k], k
        assert [list(a) for a in oracle_assign(m.max() - m)] == exp[k], k


def test_cv2_resize_restatement_hand_derived():
    """OpenCV's 8-bit resize, restated (parity unpinned: no cv2 in the image).  Hand-derived cases:
    * 2 -> 4 columns of [0, 100]: fx = -0.25 (clamped: 0), 0.25, 0.75, 1.25 (clamped: 100)  -> weights 2048/0, 1536/512,
      512/1536, 2048/0 -> horizontal 0, 51200, 153600, 204800 -> (((2048 * (h >> 4)) >> 16) + 2) >> 2 = 0, 25, 75, 100
    * exactly half scale in both directions = the 2x2 area filter (a + b + c + d + 2) >> 2
    * nearest: sx = floor(dx * sw / dw), the last source column / row clamped"""
    np.testing.assert_array_equal(adapter_np.cv2_resize_linear_u8(np.array([[0, 100]], np.uint8), 4, 1), [[0, 25, 75, 100]])
    img = np.array([[10, 20, 30, 40], [50, 60, 70, 81]], np.uint8)
    np.testing.assert_array_equal(adapter_np.cv2_resize_linear_u8(img, 2, 1), [[(10 + 20 + 50 + 60 + 2) >> 2, (30 + 40 + 70 + 81 + 2) >> 2]])
    np.testing.assert_array_equal(adapter_np.cv2_resize_nearest(np.arange(12).reshape(3, 4), 6, 5)[:, [0, 1, 2, 5]],
                                  [[0, 0, 1, 3], [0, 0, 1, 3], [4, 4, 5, 7], [4, 4, 5, 7], [8, 8, 9, 11]])
    same = np.random.default_rng(0).integers(0, 256, (7, 9, 3)).astype(np.uint8)
    np.testing.assert_array_equal(adapter_np.cv2_resize_linear_u8(same, 9, 7), same)      # identity size: weights 2048 / 0
    # a constant image stays constant through the fixed-point passes
    assert (adapter_np.cv2_resize_linear_u8(np.full((5, 6), 203, np.uint8), 11, 13) == 203).all()


def test_boundary_restatement_hand_derived():
    """seg2bmap / disk / boundary_overlap (evaluation.py:21-54, utilities.py:672-697) restated without OpenCV / skimage
    (parity unpinned).  Hand-derived cases for the contour semantics the restatement assumes."""
    from oracle import metrics_np as M
    seg = np.zeros((9, 9), np.uint8)
    seg[2:7, 2:7] = 1
    exp = seg.copy()
    exp[3:6, 3:6] = 0
    np.testing.assert_array_equal(M.seg2bmap(seg), exp)                       # a solid block: its one-pixel rim
    ring = np.zeros((11, 11), np.uint8)
    ring[1:10, 1:10] = 1
    ring[3:8, 3:8] = 0                                                        # a hole ...
    ring[5, 5] = 1                                                            # ... with an object nested in it
    b = M.seg2bmap(ring)
    outer = np.zeros_like(ring)
    outer[1:10, 1:10] = 1
    outer[2:9, 2:9] = 0
    np.testing.assert_array_equal(b, outer)            # RETR_EXTERNAL: neither the hole's border nor the nested object
    edge = np.zeros((6, 8), np.uint8)
    edge[0:3, 0:4] = 1                                                        # touches the frame: frame-side pixels are border
    e = edge.copy()
    e[1, 1:3] = 0
    np.testing.assert_array_equal(M.seg2bmap(edge), e)
    diag = np.zeros((5, 5), np.uint8)
    diag[1, 1] = diag[2, 2] = 1
    np.testing.assert_array_equal(M.seg2bmap(diag), diag)
    corner = np.ones((4, 4), np.uint8)                                         # an inner pixel that only touches the outside
    corner[0, 0] = 0                                                           # diagonally is not on the 8-connected border
    c = np.pad(corner, 1)
    bc = M.seg2bmap(c)
    assert bc[2, 2] == 0 and bc[1, 2] == 1 and bc[2, 1] == 1
    assert M.disk(1).sum() == 5 and M.disk(2).sum() == 13 and M.disk(3).sum() == 29 and M.disk(3)[0].tolist() == [0, 0, 0, 1, 0, 0, 0]
    # two unit squares two pixels apart: every border pixel of one lies inside the other's dilation by disk(3)
    a, g = np.zeros((12, 16), bool), np.zeros((12, 16), bool)
    a[4:8, 3:7] = True
    g[4:8, 5:9] = True
    assert M.boundary_overlap(a, g, bound_th=3) == (12, 12)
    assert np.ceil(0.003 * np.linalg.norm((480, 640))) == 3


def test_inpaint_telea_restatement():
    """cv2.inpaint(..., 3, INPAINT_TELEA) restated (parity unpinned: no OpenCV in the image).  The library's host function and
    the independent numpy restatement agree exactly; hand-derivable properties: pixels outside the mask are untouched, a
    constant image is reproduced exactly in holes away from the frame, a smooth depth-like image to a few grey levels."""
    import ctypes as C
    from oracle import inpaint_np
    from quber_amd import _lib
    from quber_amd.eval.refiner_model import inpaint_depth
    lib = _lib.load()

    def run(img, mask, r=3):
        out = np.empty_like(img)
        assert lib.quber_inpaint_telea_u8(C.c_void_p(img.ctypes.data), C.c_void_p(mask.ctypes.data), img.shape[0], img.shape[1],
                                          r, C.c_void_p(out.ctypes.data)) == 0
        return out

    h, w = 48, 64
    yy, xx = np.mgrid[0:h, 0:w]
    mask = np.zeros((h, w), np.uint8)
    mask[10:18, 20:33] = 1
    mask[30:33, 5:40] = 1
    mask[40:48, 60:64] = 1                                         # touches the frame
    rng = np.random.default_rng(0)
    for name, img in (("const", np.full((h, w), 117, np.uint8)), ("ramp", (2 * xx + yy).astype(np.uint8)),
                      ("noise", rng.integers(0, 256, (h, w)).astype(np.uint8))):
        src = img.copy()
        src[mask != 0] = 0
        got = run(src, mask)
        np.testing.assert_array_equal(got, inpaint_np.inpaint_telea_u8(src, mask), err_msg=name)
        np.testing.assert_array_equal(got[mask == 0], src[mask == 0])
        if name == "const":
            np.testing.assert_array_equal(got, img)
        if name == "ramp":                                             # slope 2 / px; holes away from the frame
            e = np.abs(got.astype(int) - img.astype(int))
            assert e[10:18, 20:33].max() <= 6 and e[30:33, 5:40].max() <= 4
    # inpaint_depth (eval/preprocess_utils.py:44-64): only zero pixels change, all three channels alike
    d = np.clip(60 + yy * 2 + 10 * np.sin(xx / 9.0), 1, 255).astype(np.uint8)
    d3 = np.repeat(d[:, :, None], 3, 2)
    d3[12:20, 30:41] = 0
    d3[0:3, 0:4] = 0
    out = inpaint_depth(d3)
    np.testing.assert_array_equal(out, inpaint_np.inpaint_depth(d3))
    assert (out[d3 != 0] == d3[d3 != 0]).all() and (out[12:20, 30:41] > 0).all()
    assert np.abs(out[12:20, 30:41, 0].astype(int) - d[12:20, 30:41]).max() <= 8
    # three DIFFERENT channels (not what normalize_depth produces, but what the function accepts): a pixel is masked only where all
    # three are 0, yet every zero ELEMENT takes its channel's filled value (np.where(depth == 0, ...), element-wise)
    d3b = d3.copy()
    d3b[..., 1] = np.clip(d3b[..., 1].astype(int) + 7, 0, 255)
    d3b[12:20, 30:41] = 0
    d3b[25:27, 50:60, 2] = 0                                           # zero in one channel only: not part of the mask
    np.testing.assert_array_equal(inpaint_depth(d3b), inpaint_np.inpaint_depth(d3b))
    no_holes = np.repeat(d[:, :, None], 3, 2)
    np.testing.assert_array_equal(inpaint_depth(no_holes), no_holes)


@pytest.mark.parametrize("path", golden("wiring"), ids=os.path.basename)
def test_reference_wiring_fixture(path):
    """EXTRA EVIDENCE (it pins nothing): tests/golden/wiring_*.npz are outputs of the REFERENCE's own module code -
    mask_refiner/model.py (MaskRefiner.forward, MaskRefinerInsEmbedHead incl. the hierarchy loop :738-762, FusionLayers,
    SinglePredictionHead / SinglePredictor, the instance extraction :313-356) and backbone/resnet.py (DeepLabStem, ResNet stages,
    RGBDFusionBackbone) - imported unmodified with stand-ins for the detectron2 / fvcore / monai symbols they use
    (oracle/gen_wiring.py), on quber_amd.arch's seeded weights (loaded by key: the reference's module tree names every tensor as
    arch.param_specs does).  The oracle restatement must reproduce every head output, the label map and the instances.
    The detectron2 layers themselves (Conv2d wrapper, FrozenBN, BottleneckBlock, ASPP, DeepLabV3PlusHead.layers) are the generator's
    stand-ins, i.e. the same recollection of detectron2 the oracle rests on: that part stays unpinned."""
    import ast
    from oracle.network_torch import ArchCfg, MaskRefinerNet
    from quber_amd import arch
    z = np.load(path)
    kw = ast.literal_eval(str(z["arch_kwargs"]))
    sd = arch.init_state_dict(seed=int(z["seed"]), loud_heads=True, center_bias=float(z["center_bias"]), **kw)
    okw = dict(kw, hierarchy=[list(l) for l in kw["hierarchy"]], fusion_target=list(kw["fusion_target"]))
    net = MaskRefinerNet(ArchCfg(**okw)).eval()
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    np.testing.assert_array_equal(encode_np.encode_initial_masks(z["masks"]), z["offsets"])
    image = torch.from_numpy(np.concatenate([z["rgb"], z["depth"]], -1)).permute(2, 0, 1)[None]
    with torch.no_grad():
        out = net(image, torch.from_numpy(z["offsets"][None]))
    heads = [k[5:] for k in z.files if k.startswith("head_")]
    assert set(heads) == set(out) and len(heads) >= 4
    for k in heads:
        ref = torch.from_numpy(z["head_" + k])
        assert out[k].shape == ref.shape, k
        assert float(ref.abs().max()) > 0.3, k                                       # loud heads: the check means something
        # two CPU torch evaluations of the same graph: summation order of fused vs unfused ops only
        assert float((out[k] - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (k, float((out[k] - ref).abs().max()))
    # model.py:266-289 (sem_seg_postprocess at the frame's own size = identity) and :290-356 on the REFERENCE's logits: bit-exact
    np.testing.assert_array_equal(z["sem_seg"], z["head_foreground"][0])
    o = postproc_ref.postprocess(torch.from_numpy(z["head_foreground"][0]), torch.from_numpy(z["head_center"][0]), torch.from_numpy(z["head_offset"][0]))
    np.testing.assert_array_equal(o["panoptic"].numpy(), z["panoptic"])
    assert len(z["inst_scores"]) >= 1
    np.testing.assert_array_equal(o["masks"].numpy(), z["inst_masks"])
    np.testing.assert_array_equal(o["boxes"].numpy(), z["inst_boxes"])
    np.testing.assert_array_equal(o["classes"].numpy(), z["inst_classes"])
    np.testing.assert_allclose(o["scores"].numpy(), z["inst_scores"], rtol=1e-6, atol=1e-7)
