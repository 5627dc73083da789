"""GPU: parity that bites, on the plan that is benchmarked.

The reference's predictor init N(0, 0.001) (model.py:418-419) gives logits of ~3e-3 and K = 0 instances, so a 1e-4 check
on those logits proves little and post-processing runs on an empty scene.  These tests use the "loud" weight set
(arch.init_state_dict(loud_heads=True): predictors N(0, 0.25), centre bias calibrated for ~N peaks per frame): logits
are O(1), every head output depends on the features, and K ~ N instances reach grouping / merge / extraction.

Bars (BASELINE.json north_star: float within 1e-4, label maps bit-exact):
  * seven intermediate taps: max |d| <= 1e-4 * max(1, max |ref|)   (measured 2e-6 .. 8e-6, profiles/r02a_parity_report.txt)
  * head outputs in head units: |d| <= 1e-4 on fg / centre / error logits; the offset planes carry the common-stride
    factor 4 (model.py:695-700), so their bar is 4e-4 px.  (The fp32 oracle itself is 7e-5 from a float64 evaluation.)
  * post-processing of the HIP logits: label map, labels, boxes, masks bit-exact against the oracle's post-processing
"""
import numpy as np
import pytest
import torch

from oracle import encode_np, postproc_ref
from oracle.network_torch import ArchCfg, MaskRefinerNet
from quber_amd import arch, engine, synth

pytestmark = pytest.mark.gpu

TOL = 1e-4
STRIDE = 4


def _oracle(sd, **kw):
    kw = dict(kw)
    if "hierarchy" in kw:
        kw["hierarchy"] = [list(l) for l in kw["hierarchy"]]
    if "fusion_target" in kw:
        kw["fusion_target"] = list(kw["fusion_target"])
    net = MaskRefinerNet(ArchCfg(**kw)).eval()
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return net


def _scene(seed, b, h, w, n, single=False):
    batch = synth.make_batch(seed, b, h, w, n)
    offs = np.stack([encode_np.encode_initial_masks(m) for m in batch["masks"]])
    parts = [torch.from_numpy(batch["rgb"])] + ([] if single else [torch.from_numpy(batch["depth"])])
    return batch, offs, torch.cat(parts, -1).permute(0, 3, 1, 2)


def loud_state_dict(seed, image, offs, n, **kw):
    """Loud predictors with the centre bias calibrated (on the oracle, first two frames) so that ~n peaks pass 0.3."""
    sd0 = arch.init_state_dict(seed=seed, loud_heads=True, **kw)
    with torch.no_grad():
        out = _oracle(sd0, **kw)(image[:2], torch.from_numpy(offs[:2]))
    return arch.init_state_dict(seed=seed, loud_heads=True, center_bias=arch.calibrate_center_bias(out["center"], n), **kw)


def _check_heads(logits, ref, planes):
    """logits [B,P,H,W] from the HIP path against the oracle's head dict, in head units."""
    o = 0
    for key in planes:
        c = ref[key].shape[1]
        d = float((logits[:, o:o + c] - ref[key]).abs().max())
        bar = TOL * STRIDE if key == "offset" else TOL
        assert d < bar, f"{key}: max |d| = {d:.2e} (bar {bar:.0e})"
        o += c
    assert o == logits.shape[1]


def _rel(got, ref):
    return float((got - ref).abs().max() / max(1.0, float(ref.abs().max())))


_PLAN_REF = {}


def _plan_reference(h, w, b, n):
    """scene, calibrated loud weights and the oracle's taps / heads for a configuration - computed once per size (host time)."""
    key = (h, w, b, n)
    if key not in _PLAN_REF:
        batch, offs, image = _scene(7, b, h, w, n)
        sd = loud_state_dict(0, image, offs, n)
        taps = {}
        with torch.no_grad():
            ref = _oracle(sd)(image, torch.from_numpy(offs), taps)
        _PLAN_REF.clear()                                    # one configuration at a time: the taps are large
        _PLAN_REF[key] = (batch, offs, sd, taps, ref, {})
    return _PLAN_REF[key]


@pytest.mark.parametrize("h,w,b,n,dtype", [(480, 640, 16, 20, 0), (480, 640, 16, 20, 3), (720, 1280, 1, 30, 0), (720, 1280, 1, 30, 3)],
                         ids=["b16-640x480-f32", "b16-640x480-bf16x3", "b1-1280x720-f32", "b1-1280x720-bf16x3"])
def test_benchmarked_plan_taps_heads_and_instances(h, w, b, n, dtype):
    """BASELINE.json configs[1] (batch 16, 640x480, N = 20) and configs[2] (1280x720, N = 30) on the default plan, in the
    exact fp32 MFMA mode (compute_dtype 0) and in the fp32-equivalent bf16x3 mode (compute_dtype 3) - the SAME bars."""
    batch, offs, sd, taps, ref, e2e = _plan_reference(h, w, b, n)
    qc = engine.make_config(h, w, max_batch=b, max_instances=n)
    qc.compute_dtype = dtype
    eng = engine.Engine(qc, "cuda:0")
    eng.load_state_dict(sd)
    masks = torch.from_numpy(batch["masks"]).cuda()
    bgr, dep = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda()
    enc = eng.encode(masks)
    np.testing.assert_array_equal(enc.cpu().numpy().view(np.uint32), offs.view(np.uint32))        # a1 bit-exact
    lg = eng.forward(bgr, dep, enc)
    post = eng.postprocess(lg)
    mx = int(post["count"].max().item())
    pm = eng.extract_masks(post, max(mx, 1)).cpu().numpy()
    lgc = lg.cpu()
    for name in ("res2", "res3", "res5", "y", "feat_eee_boundary", "z1", "feat_center"):
        got = eng.debug_tensor(name, b).cpu().permute(0, 3, 1, 2)
        assert _rel(got, taps[name]) < TOL, name
    _check_heads(lgc, ref, ("foreground", "center", "offset", "eee_boundary"))
    # a8-a11 on the HIP logits: bit-exact against the oracle's post-processing of the same logits, with K ~ N instances
    ks, same = [], []
    for i in range(b):
        o = postproc_ref.postprocess(lgc[i, 0:1], lgc[i, 1:2], lgc[i, 2:4])
        k = len(o["labels"])
        ks.append(k)
        np.testing.assert_array_equal(post["panoptic"][i].cpu().numpy(), o["panoptic"].numpy())
        assert int(post["count"][i]) == k
        np.testing.assert_array_equal(post["labels"][i, :k].cpu().numpy(), o["labels"].numpy())
        if k:
            np.testing.assert_array_equal(post["boxes"][i, :k].cpu().numpy(), o["boxes"].numpy())
            np.testing.assert_array_equal(pm[i, :k].astype(bool), o["masks"].numpy())
            np.testing.assert_allclose(post["scores"][i, :k].cpu().numpy(), o["scores"].numpy(), rtol=2e-5, atol=1e-6)
        # end to end (HIP logits -> HIP labels) against (oracle logits -> oracle labels): threshold-straddling pixels may flip
        if i % 4 == 0:                      # (the oracle's grouping takes ~1 s per frame on the host: every fourth frame)
            if i not in e2e:
                e2e[i] = postproc_ref.postprocess(ref["foreground"][i], ref["center"][i], ref["offset"][i])["panoptic"]
            same.append(float((post["panoptic"][i].cpu() == e2e[i]).float().mean()))
    assert np.mean(ks) >= 15, ks
    assert min(same) > 0.9999, same
    if b > 1:
        # the same frame alone: every layer's algorithm is fixed at plan time, so only the split-K partitioning (a
        # re-association of the same fp32 sums, chosen per launch) differs - the batch-1 result meets the same bars
        lg1 = eng.forward(bgr[:1], dep[:1], enc[:1]).cpu()
        _check_heads(lg1, {k: v[:1] for k, v in ref.items()}, ("foreground", "center", "offset", "eee_boundary"))
        d1 = (lg1[0] - lgc[0]).abs()
        assert float(d1[[0, 1, 4, 5, 6, 7]].max()) < TOL and float(d1[2:4].max()) < TOL * STRIDE
        p1 = eng.postprocess(lg1.cuda())["panoptic"][0].cpu()
        assert float((p1 == post["panoptic"][0].cpu()).float().mean()) > 0.9999
    eng.close()


VARIANTS = {
    "default": dict(),
    # run_eval.py's default config: ...-hf-m-b-f-c-o-l3-e2-b8.yaml (5 levels, 2 error classes, mask + boundary)
    "m-b-f-c-o-e2": dict(eee_mask_on=True, error_classes=2, fusion_target=("pred", "feat"),
                         hierarchy=(("eee_mask",), ("eee_boundary",), ("foreground",), ("center",), ("offset",))),
    "m-b-fco-feat": dict(eee_mask_on=True, fusion_target=("feat",),
                         hierarchy=(("eee_mask",), ("eee_boundary",), ("foreground", "center", "offset"))),
    "bfco-single-level": dict(hierarchy=(("eee_boundary", "foreground", "center", "offset"),)),
    "noeee-flat": dict(hierarchical=False, eee_boundary_on=False, error_classes=2),
    "mb-fco-e33-pred": dict(eee_mask_on=True, error_classes=3, fusion_target=("pred",),
                            hierarchy=(("eee_mask", "eee_boundary"), ("foreground", "center", "offset"))),
    "l0-backbone-fusion": dict(backbone_fusion_layers=0),
    "single-stream": dict(streams=1),
    "add-fusion-l3": dict(fusion_add=True, backbone_fusion_layers=3),    # Base-Mask-Refiner.yaml's own defaults
    "r101": dict(depth=101),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variant_taps_and_heads_loud(name):
    """Every architecture variant: all intermediate taps the plan exposes and every head output, loud predictors."""
    kw = VARIANTS[name]
    h, w, b, n = 128, 160, 2, 4
    single = kw.get("streams", 2) == 1
    batch, offs, image = _scene(5, b, h, w, n, single)
    sd = loud_state_dict(4, image, offs, n, **kw)
    net = _oracle(sd, **kw)
    taps = {}
    with torch.no_grad():
        ref = net(image, torch.from_numpy(offs), taps)
    qc = engine.set_arch(engine.make_config(h, w, max_batch=b), **kw)
    eng = engine.Engine(qc, "cuda:0")
    eng.load_state_dict(sd)
    lg = eng.forward(torch.from_numpy(batch["rgb"]).cuda(), None if single else torch.from_numpy(batch["depth"]).cuda(),
                     torch.from_numpy(offs).cuda()).cpu()
    checked = 0
    for tname, tref in taps.items():
        if tref.dim() != 4:
            continue
        try:
            got = eng.debug_tensor(tname, b).cpu().permute(0, 3, 1, 2)
        except Exception:
            continue                                   # a tap the oracle records but the plan keeps inside a fused buffer
        assert got.shape == tref.shape, tname
        assert _rel(got, tref) < TOL, tname
        checked += 1
    assert checked >= 5, checked
    planes = ["foreground", "center", "offset"] + (["eee_boundary"] if "eee_boundary" in ref else []) + \
             (["eee_mask"] if "eee_mask" in ref else [])
    _check_heads(lg, ref, planes)
    eng.close()


# the reduced-precision modes' OWN tolerances (max over the frame, relative to the tap's largest magnitude / head units)
# (measured, profiles/r02e_half_precision.txt: fp16 taps 1.1e-3 .. 5.9e-3, heads 3.4e-2, fg IoU 0.9964; bf16 taps 0.8e-2 ..
# 4.7e-2, heads 0.24, fg IoU 0.9715.)  Label-map EQUALITY is not asserted for these modes: with untrained "loud" heads the
# centre scores sit densely around the 0.3 threshold and the argmin grouping has near-ties everywhere, so a 1e-2
# perturbation re-partitions whole instances (60-70 % of the pixels keep their label) - a property of random heads, not of
# the arithmetic; the foreground IoU and the head-output bars are what the modes are held to.
HALF_TOL = {"fp16": dict(dtype=2, taps=1e-2, heads=5e-2, fg_iou=0.99),
            "bf16": dict(dtype=1, taps=8e-2, heads=5e-1, fg_iou=0.96)}


_HALF_REF = {}


def _half_reference(h, w, b, n):
    """scene, weights, oracle taps / heads / label maps at 1024x1024 - computed once for both modes (host time)."""
    if not _HALF_REF:
        batch, offs, image = _scene(11, b, h, w, n)
        sd = loud_state_dict(0, image, offs, n)
        taps = {}
        with torch.no_grad():
            ref = _oracle(sd)(image, torch.from_numpy(offs), taps)
        pans = [postproc_ref.postprocess(ref["foreground"][i], ref["center"][i], ref["offset"][i])["panoptic"] for i in range(b)]
        _HALF_REF.update(batch=batch, offs=offs, sd=sd, taps=taps, ref=ref, pans=pans)
    return _HALF_REF


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_half_precision_mode_config5_1024x1024(mode):
    """BASELINE.json configs[4] stand-in (the R50 refiner with 16-bit operands, fp32 accumulation, at 1024x1024):
    quber_config.compute_dtype = 2 (fp16, what configs[4] names) or 1 (bf16).  Tolerances are the mode's own (HALF_TOL),
    against the fp32 oracle; the fp32 default keeps the 1e-4 bar above.  Post-processing stays bit-exact on the mode's
    own logits."""
    tol = HALF_TOL[mode]
    h, w, b, n = 1024, 1024, 2, 20
    R = _half_reference(h, w, b, n)
    batch, offs, sd, taps, ref = R["batch"], R["offs"], R["sd"], R["taps"], R["ref"]
    qc = engine.make_config(h, w, max_batch=b, max_instances=n)
    qc.compute_dtype = tol["dtype"]
    eng = engine.Engine(qc, "cuda:0")
    eng.load_state_dict(sd)
    lg = eng.forward(torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda(), torch.from_numpy(offs).cuda())
    post = eng.postprocess(lg)
    lgc = lg.cpu()
    errs = {name: _rel(eng.debug_tensor(name, b).cpu().permute(0, 3, 1, 2), taps[name])
            for name in ("res2", "res3", "res5", "y", "feat_eee_boundary", "z1", "feat_center")}
    exp = torch.cat([ref["foreground"], ref["center"], ref["offset"] / STRIDE, ref["eee_boundary"]], 1)
    got = lgc.clone()
    got[:, 2:4] /= STRIDE
    errs["heads"] = float((got - exp).abs().max())
    ious, same = [], []
    for i in range(b):
        pan = post["panoptic"][i].cpu()
        a, b_ = pan >= 0, R["pans"][i] >= 0
        ious.append(float((a & b_).sum()) / float((a | b_).sum()))
        same.append(float((pan == R["pans"][i]).float().mean()))
        o = postproc_ref.postprocess(lgc[i, 0:1], lgc[i, 1:2], lgc[i, 2:4])
        np.testing.assert_array_equal(pan.numpy(), o["panoptic"].numpy())
    print(f"\n[{mode}] errors {errs}, fg IoU {ious}, label maps equal {same}")
    assert all(v < tol["taps"] for k, v in errs.items() if k != "heads"), errs
    assert errs["heads"] < tol["heads"], errs
    assert min(ious) >= tol["fg_iou"], ious
    eng.close()
