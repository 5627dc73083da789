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
from oracle import fp64_anchor as fa
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


_PLAN = {}
MODES = {"f32": 0, "bf16x3": 3}


def _plan_results(h, w, b, n):
    """Everything the benchmarked-plan tests assert on, computed ONCE per configuration:
      * the scene and the calibrated loud weights;
      * per arithmetic mode (exact fp32 MFMA, bf16x3): the HIP path's encode, taps, logits, post-processing, masks, and the
        same for frame 0 alone (batch 1 on the batch-b engine);
      * the oracle streamed frame by frame in float32 AND float64 on a worker thread (fa.OracleStream; batch 1, as the
        reference runs), while this thread evaluates the oracle's post-processing of every candidate's logits:
        max |x - fp64| per tap / head for HIP and for the fp32 oracle (fa.AnchorErrors), the explanation of every
        flipped label pixel (fa.explain_label_flips), and the classic HIP-vs-fp32-oracle distances."""
    key = (h, w, b, n)
    if key in _PLAN:
        return _PLAN[key]
    batch, offs, image = _scene(7, b, h, w, n)
    sd = loud_state_dict(0, image, offs, n)
    stream = fa.OracleStream(sd, image, offs)                # starts computing now
    masks = torch.from_numpy(batch["masks"]).cuda()
    bgr, dep = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda()
    modes = {}
    for mode, dtype in MODES.items():
        qc = engine.make_config(h, w, max_batch=b, max_instances=n)
        qc.compute_dtype = dtype
        eng = engine.Engine(qc, "cuda:0")
        eng.load_state_dict(sd)
        enc = eng.encode(masks)
        lg = eng.forward(bgr, dep, enc)
        post = eng.postprocess(lg)
        mx = int(post["count"].max().item())
        pm = eng.extract_masks(post, max(mx, 1)).cpu().numpy()
        m = {"enc": enc.cpu().numpy(), "lg": lg.cpu(), "post": {k: v.cpu() for k, v in post.items()}, "pm": pm,
             "taps": {name: eng.debug_tensor(name, b).cpu().permute(0, 3, 1, 2) for name in fa.TAPS},
             "anchor": fa.AnchorErrors(), "flips": [], "flip_error": None, "dec": [], "inst": [],
             "tap_rel32": {name: 0.0 for name in fa.TAPS}}
        if b > 1:
            lg1 = eng.forward(bgr[:1], dep[:1], enc[:1])
            m["lg1"], m["pan1"] = lg1.cpu(), eng.postprocess(lg1)["panoptic"][0].cpu()
        eng.close()
        modes[mode] = m
    del masks, bgr, dep
    torch.cuda.empty_cache()
    lg32 = []
    for fr in stream:
        i = fr["i"]
        l32, l64 = fa.cat_heads(fr["out32"]), fa.cat_heads(fr["out64"])
        lg32.append(l32)
        dec32 = fa.decide(l32[0])
        for m in modes.values():
            for name in fa.TAPS:
                t32, t64, th = fr["taps32"][name], fr["taps64"][name], m["taps"][name][i:i + 1]
                m["anchor"].add(name, th, t32, t64, scale=max(1.0, float(t64.abs().max())))
                m["tap_rel32"][name] = max(m["tap_rel32"][name], _rel(th, t32))
            m["anchor"].add_heads(m["lg"][i:i + 1], l32, l64)
            # a8-a11 by the oracle on the HIP logits (shared by the bit-exactness checks and the flip explanation)
            dec = fa.decide(m["lg"][i])
            m["dec"].append(dec)
            m["inst"].append(postproc_ref.extract_instances(dec["pan"], m["lg"][i, 0:1], m["lg"][i, 1:2]))
            try:
                m["flips"].append(fa.explain_label_flips(None, None, l64[0], pan_hip=m["post"]["panoptic"][i], dec_hip=dec, dec_o32=dec32))
            except AssertionError as e:                      # reported by the test of that mode, not by whichever ran first
                m["flip_error"] = m["flip_error"] or f"frame {i}: {e}"
    for m in modes.values():
        del m["taps"]
    lg32 = torch.cat(lg32)
    ref = {"foreground": lg32[:, 0:1], "center": lg32[:, 1:2], "offset": lg32[:, 2:4], "eee_boundary": lg32[:, 4:8]}
    _PLAN[key] = {"offs": offs, "ref": ref, "lg32": lg32, "modes": modes}
    return _PLAN[key]


PLAN_CASES = [(480, 640, 16, 20, "f32"), (480, 640, 16, 20, "bf16x3"), (720, 1280, 1, 30, "f32"), (720, 1280, 1, 30, "bf16x3")]
PLAN_IDS = ["b16-640x480-f32", "b16-640x480-bf16x3", "b1-1280x720-f32", "b1-1280x720-bf16x3"]


@pytest.mark.parametrize("h,w,b,n,mode", PLAN_CASES, ids=PLAN_IDS)
def test_benchmarked_plan_taps_heads_and_instances(h, w, b, n, mode):
    """BASELINE.json configs[1] (batch 16, 640x480, N = 20) and configs[2] (1280x720, N = 30) on the default plan, in the
    exact fp32 MFMA mode (compute_dtype 0) and in the fp32-equivalent bf16x3 mode (compute_dtype 3) - the SAME bars."""
    R = _plan_results(h, w, b, n)
    m, ref = R["modes"][mode], R["ref"]
    np.testing.assert_array_equal(m["enc"].view(np.uint32), R["offs"].view(np.uint32))            # a1 bit-exact
    for name, v in m["tap_rel32"].items():
        assert v < TOL, (name, v)
    _check_heads(m["lg"], ref, ("foreground", "center", "offset", "eee_boundary"))
    # a8-a11 on the HIP logits: bit-exact against the oracle's post-processing of the same logits, with K ~ N instances
    post, pm, ks = m["post"], m["pm"], []
    for i in range(b):
        o = m["inst"][i]
        k = len(o["labels"])
        ks.append(k)
        np.testing.assert_array_equal(post["panoptic"][i].numpy(), m["dec"][i]["pan"].numpy())
        assert int(post["count"][i]) == k
        np.testing.assert_array_equal(post["labels"][i, :k].numpy(), o["labels"].numpy())
        if k:
            np.testing.assert_array_equal(post["boxes"][i, :k].numpy(), o["boxes"].numpy())
            np.testing.assert_array_equal(pm[i, :k].astype(bool), o["masks"].numpy())
            np.testing.assert_allclose(post["scores"][i, :k].numpy(), o["scores"].numpy(), rtol=2e-5, atol=1e-6)
    assert np.mean(ks) >= 15, ks
    if b > 1:
        # the same frame alone: every layer's algorithm is fixed at plan time, so only the split-K partitioning (a
        # re-association of the same fp32 sums, chosen per launch) differs - the batch-1 result meets the same bars
        lg1 = m["lg1"]
        _check_heads(lg1, {k: v[:1] for k, v in ref.items()}, ("foreground", "center", "offset", "eee_boundary"))
        d1 = (lg1[0] - m["lg"][0]).abs()
        assert float(d1[[0, 1, 4, 5, 6, 7]].max()) < TOL and float(d1[2:4].max()) < TOL * STRIDE
        assert float((m["pan1"] == post["panoptic"][0]).float().mean()) > 0.9999


@pytest.mark.parametrize("h,w,b,n,mode", PLAN_CASES, ids=PLAN_IDS)
def test_benchmarked_plan_float64_anchor(h, w, b, n, mode):
    """The stated tolerance ("within 1e-4 (float) / bit-exact (label maps)", BASELINE.json north_star) adjudicated against a
    float64 evaluation of the oracle network on ALL frames of the benchmarked plan (oracle/fp64_anchor.py):
      (b) per tap and per head, max |HIP - fp64| <= 1.5 x max |oracle_fp32 - fp64|: the HIP path is no further from the
          exact result than the reference's own fp32 arithmetic (taps relative to the tap's magnitude; heads in head units,
          the offset planes also in raw pixels, where the x4 of model.py:700 makes the literal 1e-4 unreachable for fp32
          on either side);
      (d) the literal 1e-4 against the fp32 oracle on fg / centre / error logits and on the offsets in head units;
      (c) every pixel where the label map of the HIP path differs from the oracle's (oracle logits -> oracle labels) is a
          float64 near-tie of the decision that produced it (margins = the stated tolerance), none unexplained."""
    R = _plan_results(h, w, b, n)
    m = R["modes"][mode]
    tot = fa.summarize(m["flips"])
    print(f"\n[{PLAN_IDS[PLAN_CASES.index((h, w, b, n, mode))]}] float64 anchor\n{m['anchor'].table()}\nlabel flips: {tot}")
    ok, bad = m["anchor"].verdict()
    assert ok, "\n".join(bad)
    assert m["flip_error"] is None, m["flip_error"]
    assert len(m["flips"]) == b
    assert tot["label_map_equal_fraction"] > 0.9999, tot


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
    # ...-m-b-f-c-o-l2-b2-cdim256-hcha64.yaml / ...-cdim256.yaml: INS_EMBED_HEAD.CONVS_DIM 256, HEAD_CHANNELS 64 (model.py:610-651)
    "cdim256-hcha64": dict(convs_dim=256, head_channels=64, eee_mask_on=True, head_fusion_layers=2,
                           hierarchy=(("eee_mask",), ("eee_boundary",), ("foreground",), ("center",), ("offset",))),
    "cdim256": dict(convs_dim=256, eee_mask_on=True, head_fusion_layers=2,
                    hierarchy=(("eee_mask",), ("eee_boundary",), ("foreground",), ("center",), ("offset",))),
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


# The reduced-precision modes' OWN tolerances, from an operand-rounding error model stated BEFORE any measurement
# (VERDICT r02 item 4): u = unit round-off of the 16-bit format (fp16: 2^-11, bf16: 2^-8).  Every convolution rounds its two
# operands to the format (and, in the fp16 data path, its output once more when it is stored); with independent roundings the
# relative error of a tap after L layers in sequence grows like sqrt(L) * u, and the maximum over 10^7 elements sits ~2.5
# standard deviations out: tap bar = 2.5 * sqrt(L) * u with L = 63 convolutions on the deepest path (stem 3, 16 bottlenecks
# x 3, backbone fusion 3, ASPP 2, decoder 4, head fusion 4 + head 3) - fp16 9.7e-3 -> 1e-2, bf16 7.8e-2 -> 8e-2.  A head
# output is a 32-term sum of such features with the loud predictor weights (sigma 0.25, feature magnitudes up to ~10): 5 x the
# tap bar - fp16 5e-2, bf16 5e-1 - in head units.  Foreground IoU against the oracle's map: >= 0.99 / 0.96.
# Label-map EQUALITY is not asserted for these modes: with untrained loud heads the centre scores sit densely around the 0.3
# threshold and the argmin grouping has near-ties everywhere (a property of random heads, not of the arithmetic);
# test_reduced_precision_on_structured_outputs below is where the modes are held to instance masks.
HALF_TOL = {"fp16": dict(dtype=2, taps=1e-2, heads=5e-2, fg_iou=0.99),
            "bf16": dict(dtype=1, taps=8e-2, heads=5e-1, fg_iou=0.96)}

_HALF = {}


def _half_results(h, w, b, n):
    """BASELINE configs[4] stand-in at its own batch (8): both 16-bit modes through the HIP path, then the fp32 oracle streamed
    frame by frame (batch 1, worker thread) against both - computed once."""
    if _HALF:
        return _HALF
    batch, offs, image = _scene(11, b, h, w, n)
    sd = loud_state_dict(0, image, offs, n)
    stream = fa.OracleStream(sd, image, offs, fp64=False)
    bgr, dep, off = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda(), torch.from_numpy(offs).cuda()
    modes = {}
    for mode, tol in HALF_TOL.items():
        qc = engine.make_config(h, w, max_batch=b, max_instances=n)
        qc.compute_dtype = tol["dtype"]
        eng = engine.Engine(qc, "cuda:0")
        eng.load_state_dict(sd)
        lg = eng.forward(bgr, dep, off)
        post = eng.postprocess(lg)
        modes[mode] = {"lg": lg.cpu(), "pan": post["panoptic"].cpu(),
                       "taps": {name: eng.debug_tensor(name, b).cpu().permute(0, 3, 1, 2) for name in fa.TAPS},
                       "errs": {name: 0.0 for name in fa.TAPS + ("heads",)}, "ious": [], "same": [], "post_exact": True}
        eng.close()
    del bgr, dep, off
    torch.cuda.empty_cache()
    for fr in stream:
        i = fr["i"]
        exp = fa.cat_heads(fr["out32"])
        exp[:, 2:4] /= STRIDE
        pan32 = fa.decide(fa.cat_heads(fr["out32"])[0])["pan"]
        for m in modes.values():
            for name in fa.TAPS:
                m["errs"][name] = max(m["errs"][name], _rel(m["taps"][name][i:i + 1], fr["taps32"][name]))
            got = m["lg"][i:i + 1].clone()
            got[:, 2:4] /= STRIDE
            m["errs"]["heads"] = max(m["errs"]["heads"], float((got - exp).abs().max()))
            pan = m["pan"][i]
            a, b_ = pan >= 0, pan32 >= 0
            m["ious"].append(float((a & b_).sum()) / float((a | b_).sum()))
            m["same"].append(float((pan == pan32).float().mean()))
            if i % 4 == 0:            # post-processing stays bit-exact on the mode's own logits (two of the eight frames)
                m["post_exact"] &= bool(torch.equal(pan, fa.decide(m["lg"][i])["pan"]))
    for m in modes.values():
        del m["taps"]
    _HALF.update(modes)
    return _HALF


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_half_precision_mode_config5_1024x1024(mode):
    """BASELINE.json configs[4] stand-in (the R50 refiner with 16-bit operands, fp32 accumulation, 1024x1024, batch 8):
    quber_config.compute_dtype = 2 - the fp16 DATA PATH: activations and weights fp16 in HBM, what configs[4] names - or 1
    (bf16 operands, fp32 activations).  Tolerances are the mode's own (HALF_TOL, from the error model above), against the
    fp32 oracle; the fp32 default keeps the 1e-4 bar.  Post-processing stays bit-exact on the mode's own logits."""
    tol = HALF_TOL[mode]
    m = _half_results(1024, 1024, 8, 20)[mode]
    print(f"\n[{mode}] errors {m['errs']}, fg IoU min {min(m['ious']):.4f}, label maps equal {[round(v, 3) for v in m['same']]}")
    assert all(v < tol["taps"] for k, v in m["errs"].items() if k != "heads"), m["errs"]
    assert m["errs"]["heads"] < tol["heads"], m["errs"]
    assert min(m["ious"]) >= tol["fg_iou"], m["ious"]
    assert m["post_exact"]


def _structured_logits(h, w, n, seed):
    """Well-separated instances as a TRAINED refiner would emit them (SURVEY 8d "engineered heads" - here as logit maps, since
    no weight set can carry the input channels through the GroupNorm layers unchanged): foreground logit = 4 tanh(signed
    distance / 2 px) of the union of the ground-truth masks, centre = Gaussian (sigma 8 px, peak 0.95) at each centroid,
    offsets = centroid - pixel inside each mask (+ a smooth 0.3 px ripple so that nothing is exactly tied).
    -> float64 [8, h, w] in the plane order of quber_forward, and the number of instances."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    gt, _ = synth.make_masks(rng, n, h, w, perturb=False)
    union = gt.any(0)
    sd = ndimage.distance_transform_edt(union) - ndimage.distance_transform_edt(~union)
    sd = np.where(union, sd - 0.5, sd + 0.5)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    lg = np.zeros((8, h, w))
    lg[0] = 4.0 * np.tanh(sd / 2.0)
    for m in gt:
        ys, xs = np.nonzero(m)
        cy, cx = float(np.floor(ys.mean())), float(np.floor(xs.mean()))     # a pixel centre: one peak, no plateau of tied maxima
        lg[1] = np.maximum(lg[1], 0.95 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 8.0 ** 2)))
        lg[2][m] = cy - yy[m]
        lg[3][m] = cx - xx[m]
    lg[2] += 0.3 * np.sin(yy / 7.0) * np.cos(xx / 11.0)
    lg[3] += 0.3 * np.cos(yy / 9.0) * np.sin(xx / 5.0)
    lg[4:] = rng.standard_normal((4, h, w))
    return lg, n


# head-unit tolerances of the arithmetic modes, STATED (exact fp32 / bf16x3: BASELINE's 1e-4; 16-bit operand modes: HALF_TOL)
STRUCT_TOL = {"f32": (0, 1e-4), "bf16x3": (3, 1e-4), "fp16": (2, HALF_TOL["fp16"]["heads"]), "bf16": (1, HALF_TOL["bf16"]["heads"])}


def test_reduced_precision_on_structured_outputs():
    """What each arithmetic mode's error does to WELL-SEPARATED instances (the random loud heads above give blobs with near-ties
    everywhere, where 'IoU delta' means little for the 16-bit modes).  Structured logit maps S (trained-refiner-like) are
    perturbed by the mode's MEASURED error field on a real frame of the same size - E = logits(mode) - logits(oracle fp32) of
    the random-weight network, whose activations are O(1) like a trained one's - and both S and S + E go through the HIP
    post-processing.  Bars, stated before the run: |E| within the mode's head tolerance, and every pixel whose label changes
    is within that tolerance of a decision boundary of S (fa.explain_label_flips with the mode's margins) - so the mask
    IoU of each instance is bounded by the pixels in its tolerance band; reported per mode."""
    h, w, b, n = 480, 640, 2, 12
    batch, offs, image = _scene(21, b, h, w, 20)
    sd = loud_state_dict(0, image, offs, 20)
    with torch.no_grad():
        ref = fa.cat_heads(_oracle(sd)(image, torch.from_numpy(offs)))
    S64 = [torch.from_numpy(_structured_logits(h, w, n, 40 + i)[0]) for i in range(b)]
    S = torch.stack([s.float() for s in S64])
    bgr, dep, off = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda(), torch.from_numpy(offs).cuda()
    rows = []
    for mode, (dtype, tol) in STRUCT_TOL.items():
        qc = engine.make_config(h, w, max_batch=b, max_instances=20)
        qc.compute_dtype = dtype
        eng = engine.Engine(qc, "cuda:0")
        eng.load_state_dict(sd)
        E = eng.forward(bgr, dep, off).cpu() - ref
        E[:, 2:4] /= STRIDE                                     # head units
        e_max = float(E.abs().max())
        assert e_max <= tol, (mode, e_max)
        cand = S.clone()
        cand[:, :4] += E[:, :4] * torch.tensor([1.0, 1.0, STRIDE, STRIDE]).view(1, 4, 1, 1)
        p0, p1 = eng.postprocess(S.cuda()), eng.postprocess(cand.cuda())
        ious, flips, extra = [], 0, 0
        for i in range(b):
            a, c = p0["panoptic"][i].cpu(), p1["panoptic"][i].cpu()
            assert int(p0["count"][i]) == n                     # the structured scene: every instance found
            rep = fa.explain_label_flips(cand[i], S[i], S64[i], pan_hip=c, eps_logit=tol, eps_dist=2 * np.sqrt(2) * tol * STRIDE)
            flips += rep["flipped"]
            extra += int(p1["count"][i]) - n
            for lab in torch.unique(a[a > 0]).tolist():          # each structured instance against its best match
                ma = a == lab
                cands = torch.unique(c[ma & (c > 0)]).tolist()
                ious.append(max([float((ma & (c == l2)).sum()) / float((ma | (c == l2)).sum()) for l2 in cands] + [0.0]))
        rows.append((mode, e_max, flips, extra, min(ious), float(np.mean(ious))))
        eng.close()
    print("\n| mode | max head error (head units) | label pixels changed (2 frames) | instances gained / lost | min matched IoU | mean matched IoU |\n|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %.2e | %d | %+d | %.6f | %.6f |" % r)
    by = {r[0]: r for r in rows}
    assert by["f32"][4] == 1.0 and by["bf16x3"][4] == 1.0        # fp32-class modes: not one pixel of a well-separated instance moves
    assert by["fp16"][4] >= 0.999 and by["fp16"][3] == 0
    assert by["bf16"][4] >= 0.95                                 # (bf16 operands may add a spurious centre: its 0.5 bar is wider than 0.95 - 0.3)
