"""CPU: the float64-anchored adjudication itself (oracle/fp64_anchor.py), exercised without a GPU.

The "candidate" here is the fp32 oracle's logits plus a perturbation: at the level of fp32 round-off every flipped label
pixel must be explained as a float64 near-tie; a perturbation far above the stated tolerance must be refused."""
import numpy as np
import pytest
import torch

from oracle import encode_np
from quber_amd import arch, synth
from oracle import fp64_anchor as fa


@pytest.fixture(scope="module")
def scene():
    h, w, b, n = 192, 256, 2, 6
    batch = synth.make_batch(5, b, h, w, n)
    offs = np.stack([encode_np.encode_initial_masks(m) for m in batch["masks"]])
    image = torch.cat([torch.from_numpy(batch["rgb"]), torch.from_numpy(batch["depth"])], -1).permute(0, 3, 1, 2)
    sd0 = arch.init_state_dict(seed=0, loud_heads=True)
    net = fa.build_net(sd0)
    with torch.no_grad():
        out = net(image, torch.from_numpy(offs))
    sd = arch.init_state_dict(seed=0, loud_heads=True, center_bias=arch.calibrate_center_bias(out["center"], n))
    taps = {}
    with torch.no_grad():
        ref = fa.build_net(sd)(image, torch.from_numpy(offs), taps)
    o64 = {i: (fa.cat_heads(out)[0], t) for i, out, t in fa.oracle64(sd, image, offs)}
    return fa.cat_heads(ref), taps, o64


def test_fp32_oracle_is_close_to_float64(scene):
    lg32, taps32, o64 = scene
    for i, (lg64, t64) in o64.items():
        assert float((lg32[i].double() - lg64).abs().max()) < 2e-4
        for k in fa.TAPS:
            assert float((taps32[k][i].double() - t64[k][0]).abs().max() / t64[k].abs().max()) < 1e-5, k


def test_roundoff_level_flips_are_explained(scene):
    lg32, _, o64 = scene
    g = torch.Generator().manual_seed(1)
    reps = []
    for i, (lg64, _) in o64.items():
        noise = torch.randn(lg32[i].shape, generator=g) * 2e-5
        noise[2:4] *= fa.STRIDE
        reps.append(fa.explain_label_flips(lg32[i] + noise, lg32[i], lg64))
    tot = fa.summarize(reps)
    assert tot["flipped"] > 0, "the perturbation should straddle some thresholds (otherwise this test proves nothing)"
    assert tot["flipped"] == tot["A_fg_threshold"] + tot["B_argmin_tie"] + tot["C_centre_list"] + tot["D_area_or_relabel"]
    assert tot["max_abs_fg64_at_A"] <= fa.EPS_LOGIT and tot["max_dist_gap_at_B"] <= fa.EPS_DIST


def test_gross_perturbation_is_refused(scene):
    lg32, _, o64 = scene
    lg64, _ = o64[0]
    bad = lg32[0].clone()
    bad[0] += 5e-3                      # 50 x the stated tolerance on the foreground logits
    with pytest.raises(AssertionError):
        fa.explain_label_flips(bad, lg32[0], lg64)
    bad = lg32[0].clone()
    bad[2:4] += 0.05                    # 0.05 px on the offsets: argmin flips that are no near-ties
    with pytest.raises(AssertionError):
        fa.explain_label_flips(bad, lg32[0], lg64)


def test_anchor_verdict_bars():
    e = fa.AnchorErrors()
    x64 = torch.zeros(1, 8, 4, 4, dtype=torch.float64)
    o32 = x64.float() + 4e-5
    e.add_heads(x64.float() + 5e-5, o32, x64)
    ok, bad = e.verdict()
    assert ok, bad
    e = fa.AnchorErrors()
    e.add_heads(x64.float() + 7e-5, o32, x64)           # 1.75 x the oracle's own distance
    ok, bad = e.verdict()
    assert not ok and all("1.5 x" in b for b in bad)
    e = fa.AnchorErrors()
    e.add_heads(x64.float() - 8e-5, o32, x64)           # 1.2e-4 from the fp32 oracle: the literal bar
    ok, bad = e.verdict()
    assert not ok and any("oracle_fp32| = 1.20e-04" in b for b in bad)
