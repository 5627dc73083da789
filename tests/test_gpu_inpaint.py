"""GPU: inpaint_depth (eval/preprocess_utils.py:44-64; cv2.inpaint TELEA) on the device, csrc/inpaint_dev.hip, against the host
restatement it must equal BIT FOR BIT (quber_inpaint_depth_u8, itself checked against oracle/inpaint_np.py on the CPU).  OpenCV is
absent from the image: parity with cv2 itself is unpinned for both."""
import time

import numpy as np
import pytest
import torch

from quber_amd import engine
from quber_amd.eval.refiner_model import inpaint_depth as host_inpaint

pytestmark = pytest.mark.gpu


def depth_image(seed, h, w):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    d = 90 + 40 * np.sin(xx / 37.0 + seed) + 30 * np.cos(yy / 23.0) + rng.integers(0, 3, (h, w))
    for _ in range(4):                                   # object steps
        y, x, hh, ww = int(rng.integers(0, h - 20)), int(rng.integers(0, w - 20)), int(rng.integers(10, h // 3)), int(rng.integers(10, w // 3))
        d[y:y + hh, x:x + ww] += rng.integers(-40, 40)
    return np.clip(d, 1, 255).astype(np.uint8)


def punch(d, rng, n_holes, max_side):
    for _ in range(n_holes):
        hh, ww = int(rng.integers(1, max_side)), int(rng.integers(1, max_side))
        y, x = int(rng.integers(-2, d.shape[0] - 1)), int(rng.integers(-2, d.shape[1] - 1))
        d[max(y, 0):y + hh, max(x, 0):x + ww] = 0
    return d


CASES = [
    ("scattered_holes", 480, 640, 12, 34, 1),            # the adapter's frame size: a dozen independent regions
    ("many_small", 240, 320, 120, 7, 2),                 # single pixels and specks, many of them within reach of each other
    ("crowded", 96, 128, 40, 14, 3),                     # holes closer than the interaction distance: merged components
    ("borders", 64, 80, 10, 20, 4),                      # holes through the image frame
    ("one_big", 200, 260, 1, 150, 5),
]


@pytest.mark.parametrize("name,h,w,n_holes,side,seed", CASES, ids=[c[0] for c in CASES])
def test_device_inpaint_equals_host(name, h, w, n_holes, side, seed):
    rng = np.random.default_rng(seed)
    d = punch(depth_image(seed, h, w), rng, n_holes, side)
    d3 = np.ascontiguousarray(np.repeat(d[:, :, None], 3, 2))
    t0 = time.perf_counter()
    want = host_inpaint(d3)
    t_host = time.perf_counter() - t0
    dev = torch.from_numpy(d3).cuda()
    got = engine.inpaint_depth(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = engine.inpaint_depth(dev)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert (want[d3 == 0] > 0).any()
    if h * w <= 96 * 128:
        # the small cases against the ORACLE itself (oracle/inpaint_np.py, the restatement of eval/preprocess_utils.py:44-64 with
        # cv2.inpaint's TELEA algorithm in pure Python - seconds at this size), not only against the product's host function
        from oracle import inpaint_np
        np.testing.assert_array_equal(got.cpu().numpy(), inpaint_np.inpaint_depth(d3))
    print(f"\n[inpaint {name} {h}x{w}] {int((d == 0).sum())} hole pixels: host {t_host * 1e3:.2f} ms, device {t_dev * 1e3:.2f} ms")


def test_device_inpaint_batch_and_channels():
    """A batch of frames in one call (components of all frames side by side), a frame without holes among them, and three DIFFERENT
    channels (every zero element takes its own channel's fill; a pixel is masked only where all three are zero)."""
    rng = np.random.default_rng(9)
    frames = []
    for b in range(4):
        d = depth_image(20 + b, 120, 160)
        if b != 2:
            d = punch(d, rng, 8, 18)
        frames.append(np.repeat(d[:, :, None], 3, 2))
    frames[3] = frames[3].copy()
    frames[3][..., 1] = np.where(frames[3][..., 1] > 0, np.clip(frames[3][..., 1].astype(int) + 9, 1, 255), 0)
    frames[3][30:33, 40:60, 2] = 0                        # zero in one channel only: not part of the mask, yet replaced by its fill
    batch = np.ascontiguousarray(np.stack(frames))
    got = engine.inpaint_depth(torch.from_numpy(batch).cuda()).cpu().numpy()
    for b in range(4):
        np.testing.assert_array_equal(got[b], host_inpaint(batch[b]), err_msg=f"frame {b}")
    np.testing.assert_array_equal(got[2], batch[2])       # nothing to fill: unchanged


def test_adapter_paths_agree(tmp_path):
    """MaskRefiner's pre-processing (eval/refiner_model.py:246-263) with inpaint="device" and inpaint="host" (default): the frames
    handed to the predictor are identical."""
    from PIL import Image
    from quber_amd import synth
    from quber_amd.eval.refiner_model import MaskRefiner
    sc = synth.make_scene(5, 480, 640, 4)
    Image.fromarray(sc["rgb"][:, :, ::-1].copy()).save(tmp_path / "rgb.png")
    mm = sc["depth"][:, :, 0].astype(np.uint16) * 5 + 300
    rng = np.random.default_rng(1)
    for _ in range(10):
        y, x = int(rng.integers(0, 440)), int(rng.integers(0, 600))
        mm[y:y + int(rng.integers(2, 30)), x:x + int(rng.integers(2, 40))] = 0
    mm[:6, 200:260] = 0                                    # a hole through the frame
    Image.fromarray(mm).save(tmp_path / "depth.png")
    dev_ref = MaskRefiner(None, None, dataset="OSD", inpaint="device")
    host_ref = MaskRefiner(None, None, dataset="OSD")
    assert dev_ref.inpaint == "device" and host_ref.inpaint == "host"
    a = dev_ref._load(str(tmp_path / "rgb.png"), str(tmp_path / "depth.png"), sc["masks"] != 0)
    b = host_ref._load(str(tmp_path / "rgb.png"), str(tmp_path / "depth.png"), sc["masks"] != 0)
    np.testing.assert_array_equal(a["depth"], b["depth"])
    np.testing.assert_array_equal(a["rgb"], b["rgb"])
    assert (a["depth"] > 0).all()                          # every hole was filled
    np.testing.assert_array_equal(a["depth_dev"].cpu().numpy(), a["depth"])


@pytest.mark.parametrize("seed", range(16))
def test_device_inpaint_fuzz(seed):
    """Random frame sizes (odd ones included), hole counts and hole sizes - single pixels, stripes one pixel wide, holes that merge,
    holes that leave the frame, whole rows missing - device against host, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(12, 150)), int(rng.integers(12, 200))
    d = depth_image(seed, 160, 200)[:h, :w].copy()
    kind = seed % 4
    if kind == 0:
        d = punch(d, rng, int(rng.integers(1, 30)), int(rng.integers(2, 25)))
    elif kind == 1:                                       # specks and one-pixel-wide stripes
        d[rng.random((h, w)) < 0.03] = 0
        d[int(rng.integers(0, h)), :] = 0
        d[:, int(rng.integers(0, w))] = 0
    elif kind == 2:                                       # few pixels known
        keep = rng.random((h, w)) < 0.15
        d = np.where(keep, d, 0).astype(np.uint8)
    else:
        d = punch(d, rng, 3, max(3, min(h, w) // 2))
    if not (d == 0).any():
        d[h // 2, w // 2] = 0
    if (d == 0).all():
        d[0, 0] = 77
    d3 = np.ascontiguousarray(np.repeat(d[:, :, None], 3, 2))
    got = engine.inpaint_depth(torch.from_numpy(d3).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, host_inpaint(d3))
