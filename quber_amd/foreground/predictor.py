"""Drop-in for the reference's LMFFNet foreground predictor (foreground_segmentation/predictor.py:57-99 ``lmffNet``)
and the 30 % overlap post-filter built on it (eval/refiner_model.py:273-277), on the MI355X HIP path.

``lmffNet(weight_path).predict(rgb_path, depth_path) -> bool [480, 640]`` (class 2 = foreground) keeps the reference's
signature; ``predict_arrays`` / ``filter_masks`` are the array-level forms the adapter uses.  Without a checkpoint
(the reference's ``rgbd_lmffnet.pth`` is not in its repository) seeded synthetic weights are used.
``predict`` in-paints the normalised, resized depth (``inpaint_depth(depth_img, factor=1)``, predictor.py:77) with the same
TELEA restatement the refiner adapter uses (quber_amd/eval/refiner_model.py:inpaint_depth; ``lmffNet(inpaint=False)``
skips it), so the stand-alone foreground mask and the adapter's are computed from the same depth image."""
import ctypes as C
import os
import warnings

import numpy as np
import torch
from PIL import Image

from .. import _lib, engine as qengine, lmff_arch

W, H = 640, 480


class LmffEngine:
    """One LMFFNet context (quber_config.with_network = 2) for a fixed frame size / batch capacity."""

    def __init__(self, state_dict, height=H, width=W, max_batch=1, device="cuda:0"):
        qc = qengine.make_config(height, width, max_batch=max_batch)
        qc.with_network = 2
        self.eng = qengine.Engine(qc, device)
        self.eng.planes = 3
        self.eng.load_state_dict(state_dict)
        self.device = self.eng.device
        self.H, self.W = height, width

    def logits(self, bgr, depth):
        """bgr, depth u8 [B,H,W,3] device tensors -> f32 [B,3,H,W]."""
        B = bgr.shape[0]
        assert bgr.dtype == torch.uint8 and depth.dtype == torch.uint8 and bgr.is_contiguous() and depth.is_contiguous()
        assert bgr.shape == (B, self.H, self.W, 3) and depth.shape == bgr.shape
        out = torch.empty((B, 3, self.H, self.W), dtype=torch.float32, device=self.device)
        lib = self.eng.lib
        _lib.check(lib.quber_forward(self.eng.h, C.c_void_p(bgr.data_ptr()), C.c_void_p(depth.data_ptr()), C.c_void_p(0), B,
                                     C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def foreground(self, bgr, depth, masks=None):
        """-> (fg u8 [B,H,W], counts i64 [B,K,2] = (|mask & fg|, |mask|) or None).  masks: u8 [B,K,H,W] or None."""
        lg = self.logits(bgr, depth)
        B = lg.shape[0]
        fg = torch.empty((B, self.H, self.W), dtype=torch.uint8, device=self.device)
        K = 0 if masks is None else masks.shape[1]
        counts = torch.zeros((B, max(K, 1), 2), dtype=torch.int64, device=self.device)
        if K:
            assert masks.dtype == torch.uint8 and masks.is_contiguous() and masks.shape[2:] == (self.H, self.W)
        _lib.check(self.eng.lib.quber_foreground_filter(
            C.c_void_p(lg.data_ptr()), 3, 2, C.c_void_p(masks.data_ptr() if K else 0), B, K, self.H * self.W,
            C.c_void_p(fg.data_ptr()), C.c_void_p(counts.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return fg, (counts if K else None)


def filter_masks(masks_bool, counts, ratio=0.3):
    """eval/refiner_model.py:275-277: keep masks with |mask & fg| / |mask| > 0.3 (float64 division, like numpy)."""
    c = counts.cpu().numpy().astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        keep = c[:, 0] / c[:, 1] > ratio
    return [m for m, k in zip(masks_bool, keep) if k]


class lmffNet:
    def __init__(self, weight_path="./foreground_segmentation/rgbd_lmffnet.pth", device="cuda:0", seed=0, inpaint=True):
        self.inpaint = inpaint
        if weight_path is not None and os.path.exists(weight_path):
            ck = torch.load(weight_path, map_location="cpu", weights_only=False)
            sd = {k: v.numpy() for k, v in ck["model"].items() if not k.endswith("num_batches_tracked")}
        else:
            warnings.warn(f"LMFFNet weights '{weight_path}' not found; using seeded synthetic weights")
            sd = lmff_arch.init_state_dict(seed=seed)
        self.state_dict = sd
        self.device = device
        self.net = LmffEngine(sd, H, W, 1, device)

    def predict_arrays(self, bgr, depth3):
        """bgr u8 [480,640,3] (cv2.imread order), depth3 u8 [480,640,3] (normalised) -> bool [480,640]."""
        b = torch.from_numpy(np.ascontiguousarray(bgr)[None]).to(self.device)
        d = torch.from_numpy(np.ascontiguousarray(depth3)[None]).to(self.device)
        fg, _ = self.net.foreground(b, d)
        return fg[0].cpu().numpy().astype(bool)

    def predict(self, rgb_path, depth_path):
        from .. import engine as qengine
        from ..eval.refiner_model import inpaint_depth, normalize_depth
        bgr = np.asarray(Image.open(rgb_path).convert("RGB"))[:, :, ::-1]
        if "npy" in depth_path:
            depth = normalize_depth(np.load(depth_path), 0.25, 1.5)
        else:
            depth = normalize_depth(np.asarray(Image.open(depth_path)))

        def resize(img, linear):                    # cv2.resize(img, (W, H)[, INTER_NEAREST]) on the device
            if img.shape[:2] == (H, W):
                return img
            t = torch.from_numpy(np.ascontiguousarray(img)).to(self.device)
            return qengine.resize_u8(t, H, W, linear).cpu().numpy()

        depth = resize(depth, False)
        if self.inpaint:
            depth = inpaint_depth(depth)                # foreground_segmentation/predictor.py:77
        return self.predict_arrays(resize(bgr, True), depth)
