"""GPU: LMFFNet foreground network + 30 % overlap post-filter against the fixtures generated
from the imported reference module and against the oracle restatement."""
import os

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import lmffnet_torch as L
from quber_amd import lmff_arch, synth
from quber_amd.foreground.predictor import LmffEngine, filter_masks, lmffNet

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("path", golden("lmffnet"), ids=os.path.basename)
def test_lmffnet_golden(path):
    z = np.load(path)
    h, w = z["rgb"].shape[:2]
    net = LmffEngine(lmff_arch.init_state_dict(seed=int(z["seed"])), h, w, 2)
    bgr = torch.from_numpy(np.stack([z["rgb"], z["rgb"][::-1].copy()])).cuda()
    dep = torch.from_numpy(np.stack([z["depth"], z["depth"][::-1].copy()])).cuda()
    lg = net.logits(bgr, dep).cpu().numpy()
    assert np.abs(lg[0] - z["logits"]).max() < TOL
    fg, _ = net.foreground(bgr, dep)
    agree = (fg[0].cpu().numpy().astype(bool) == L.foreground_mask(z["logits"])).mean()
    assert agree > 0.999
    net.eng.close()


def test_lmffnet_full_frame_and_filter():
    sd = lmff_arch.init_state_dict(seed=3)
    sc = synth.make_scene(4, 480, 640, 10)
    w = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        ref = L.forward(L.preprocess(sc["rgb"], sc["depth"]), w)[0].numpy()
    net = LmffEngine(sd, 480, 640, 1)
    bgr, dep = torch.from_numpy(sc["rgb"][None]).cuda(), torch.from_numpy(sc["depth"][None]).cuda()
    masks = (sc["masks"] != 0)
    lg = net.logits(bgr, dep).cpu().numpy()[0]
    assert np.abs(lg - ref).max() < TOL
    fg, counts = net.foreground(bgr, dep, torch.from_numpy(masks.astype(np.uint8)[None]).cuda())
    fg_np = fg[0].cpu().numpy().astype(bool)
    # integer counts are exact for the HIP foreground mask; the kept set equals the reference rule on that mask
    c = counts[0].cpu().numpy()
    for k, m in enumerate(masks):
        assert c[k, 0] == np.sum(m & fg_np) and c[k, 1] == np.sum(m)
    kept = filter_masks(list(masks), counts[0])
    exp = L.overlap_filter(list(masks), fg_np)
    assert len(kept) == len(exp) and all(np.array_equal(a, b) for a, b in zip(kept, exp))
    net.eng.close()


def test_lmffnet_predictor_dropin(tmp_path):
    from PIL import Image
    sc = synth.make_scene(6, 480, 640, 4)
    Image.fromarray(sc["rgb"][:, :, ::-1].copy()).save(tmp_path / "rgb.png")
    Image.fromarray((sc["depth"][:, :, 0].astype(np.uint16) * 5 + 300)).save(tmp_path / "depth.png")
    with pytest.warns(UserWarning):
        p = lmffNet(str(tmp_path / "missing.pth"))
    fg = p.predict(str(tmp_path / "rgb.png"), str(tmp_path / "depth.png"))
    assert fg.shape == (480, 640) and fg.dtype == np.bool_
    from quber_amd.eval.refiner_model import MaskRefiner
    ref = MaskRefiner(None, None, dataset="OSD", foreground_filter=True, lmffnet_weights=str(tmp_path / "missing.pth"))
    masks, out, secs, fgm = ref.predict(str(tmp_path / "rgb.png"), str(tmp_path / "depth.png"), sc["masks"] != 0, None)
    assert fgm is not None and fgm.shape == (480, 640)
