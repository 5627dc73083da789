"""GPU, randomised: the byte / index kernels of the path (a1 encode, a2 error maps, a8-a11 post-processing + mask extraction)
against the oracle on random geometries - odd frame sizes, 0 ... 60 instances, empty / overlapping / border-touching masks,
arbitrary non-zero mask values - bit-exact, as the fixed-size tests of test_gpu_parity.py are.  Seeds are fixed: a failure
reproduces.  (The convolution kernels have their own randomised cross-check, tools/conv_fuzz.py.)"""
import numpy as np
import pytest
import torch

from oracle import encode_np, errmaps_np, postproc_ref
from quber_amd import engine, synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def random_masks(rng, n, h, w):
    """n uint8 masks: rectangles / ellipses / unions, some empty, some overlapping, some touching the frame, random values"""
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((n, h, w), np.uint8)
    for i in range(n):
        kind = rng.integers(0, 5)
        if kind == 0:
            continue                                            # empty
        cy, cx = rng.integers(0, h), rng.integers(0, w)
        ry, rx = rng.integers(1, max(2, h // 3)), rng.integers(1, max(2, w // 3))
        if kind in (1, 2):
            m = (np.abs(yy - cy) <= ry) & (np.abs(xx - cx) <= rx)
        else:
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        if kind == 4:                                           # two blobs
            cy2, cx2 = rng.integers(0, h), rng.integers(0, w)
            m |= (np.abs(yy - cy2) <= max(1, ry // 2)) & (np.abs(xx - cx2) <= max(1, rx // 2))
        out[i][m] = rng.choice([1, 255, int(rng.integers(1, 256))])
    return out


GEOMS = [(int(h), int(w), int(n), s) for s, (h, w, n) in enumerate(
    [(33, 47, 3), (64, 64, 0), (75, 101, 9), (96, 128, 17), (120, 67, 1), (131, 257, 33), (200, 150, 60), (17, 300, 5),
     (240, 320, 21), (301, 203, 12), (480, 640, 40), (97, 513, 7)])]


@pytest.mark.parametrize("h,w,n,seed", GEOMS)
def test_encode_random(h, w, n, seed):
    rng = np.random.default_rng(100 + seed)
    masks = np.stack([random_masks(rng, n, h, w) for _ in range(2)]) if n else np.zeros((2, 0, h, w), np.uint8)
    e = engine.Engine(engine.make_config(h, w, max_batch=2, max_instances=max(n, 1), with_network=False), "cuda:0")
    got = e.encode(dev(masks)).cpu().numpy()
    for b in range(2):
        exp = encode_np.encode_initial_masks(masks[b]) if n else np.zeros((3, h, w), np.float32)
        np.testing.assert_array_equal(got[b].view(np.uint32), exp.view(np.uint32))
    e.close()


@pytest.mark.parametrize("h,w,n,seed", [g for g in GEOMS if g[2] > 0])
def test_error_maps_random(h, w, n, seed):
    rng = np.random.default_rng(200 + seed)
    init, gt = random_masks(rng, n, h, w), random_masks(rng, max(1, n - seed % 3), h, w)
    init[init > 0] = 255                                        # one value per mask set (grey-level sets: test_gpu_parity)
    gt[gt > 0] = 255
    e = engine.Engine(engine.make_config(h, w, max_batch=1, max_instances=max(n, 1), with_network=False), "cuda:0")
    got = e.error_maps(dev(init[None]), dev(gt[None])).cpu().numpy()[0]
    np.testing.assert_array_equal(got, errmaps_np.explicit_error_maps(init, gt))
    e.close()


@pytest.mark.parametrize("h,w,n,seed", [g for g in GEOMS if g[2] > 0 and g[0] >= 33] + [(96, 128, 40, 50), (240, 320, 3, 51)])
def test_postprocess_random(h, w, n, seed):
    """logits that contain about n instances (built from a random scene by synth.fake_head_outputs) at several noise levels"""
    rng = np.random.default_rng(300 + seed)
    sc = synth.make_scene(300 + seed, h, w, n)
    enc = encode_np.encode_initial_masks(sc["masks"])
    lg, ce, of = synth.fake_head_outputs(enc, sc["masks"], rng, noise=float(rng.choice([0.1, 0.4, 0.8])))
    e = engine.Engine(engine.make_config(h, w, max_batch=1, max_instances=max(n, 1), with_network=False), "cuda:0")
    post = e.postprocess(dev(np.concatenate([lg, ce, of])[None]))
    ref = postproc_ref.postprocess(torch.from_numpy(lg), torch.from_numpy(ce), torch.from_numpy(of))
    np.testing.assert_array_equal(post["panoptic"].cpu().numpy()[0], ref["panoptic"].numpy())
    k = int(post["count"].cpu().numpy()[0])
    assert k == len(ref["labels"])
    np.testing.assert_array_equal(post["labels"].cpu().numpy()[0][:k], ref["labels"].numpy())
    np.testing.assert_array_equal(post["boxes"].cpu().numpy()[0][:k], ref["boxes"].numpy())
    if k:
        masks = e.extract_masks(post, k).cpu().numpy()[0]
        np.testing.assert_array_equal(masks.astype(bool), ref["masks"].numpy())
    e.close()


@pytest.mark.parametrize("h,w,b,seed,dtype", [(200, 264, 5, 0, 0), (168, 300, 7, 1, 0), (250, 333, 3, 2, 0), (168, 300, 7, 1, 2), (250, 333, 3, 2, 2)],
                         ids=["200x264-f32", "168x300-f32", "250x333-f32", "168x300-fp16path", "250x333-fp16path"])
def test_network_random_frames(h, w, b, seed, dtype):
    """the whole network at frame sizes nobody tuned for (ragged Winograd tiles, tile counts on both sides of the
    persistent-launch thresholds, odd stride-2 maps), loud predictors, against the oracle: seven taps and the head outputs.
    dtype 2: the fp16 data path (fp16 tensors in HBM) at the same odd sizes, held to its own tolerance (taps 1e-2, heads
    5e-2 in head units: tests/test_gpu_loud_parity.py HALF_TOL)."""
    from oracle.network_torch import MaskRefinerNet
    from quber_amd import arch
    sd = arch.init_state_dict(seed=20 + seed, loud_heads=True, center_bias=-1.5)
    net = MaskRefinerNet().eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    batch = synth.make_batch(40 + seed, b, h, w, 6)
    offs = np.stack([encode_np.encode_initial_masks(m) for m in batch["masks"]])
    qc = engine.make_config(h, w, max_batch=b, max_instances=6)
    qc.compute_dtype = dtype
    tap_tol, head_tol = (1e-2, 5e-2) if dtype == 2 else (1e-4, 1e-4)
    eng = engine.Engine(qc, "cuda:0")
    eng.load_state_dict(sd)
    logits = eng.forward(dev(batch["rgb"]), dev(batch["depth"]), dev(offs)).cpu()
    image = torch.cat([torch.from_numpy(batch["rgb"]), torch.from_numpy(batch["depth"])], -1).permute(0, 3, 1, 2)
    taps = {}
    with torch.no_grad():
        ref = net(image, torch.from_numpy(offs), taps)
    for name in ("res2", "res3", "res5", "y", "feat_eee_boundary", "z1", "feat_center"):
        got = eng.debug_tensor(name, b).cpu().permute(0, 3, 1, 2)
        assert float((got - taps[name]).abs().max() / max(1.0, float(taps[name].abs().max()))) < tap_tol, name
    exp = torch.cat([ref["foreground"], ref["center"], ref["offset"], ref["eee_boundary"]], 1)
    d = (logits - exp).abs()
    # head units: the offset planes carry the common stride 4 of model.py:700
    assert float(d[:, :2].max()) < head_tol and float(d[:, 2:4].max()) < 4 * head_tol and float(d[:, 4:].max()) < head_tol
    eng.close()


def test_network_launch_structures_random_sizes():
    """The network on random (frame size, batch, arithmetic mode): repeated forwards bit-equal, side lanes == one stream bit for bit, the frames
    of a batch == the frames one by one up to re-association (tools/network_fuzz.py; ~1 100 more cases on record, most of them over nine architecture variants: profiles/r20_network_fuzz*.txt).
    The structural net under the fixed-size parity tests: it is what would have caught profiles/r20_h8_affine_race.md two rounds earlier."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("network_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "network_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    assert mod.run(12, 2, lines.append) == 0, "\n".join(lines)
