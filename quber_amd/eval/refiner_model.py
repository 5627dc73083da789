"""Drop-in for the reference's ``eval/refiner_model.py:MaskRefiner`` adapter (reference lines 214-297):
``MaskRefiner(config_file, weights_file, dataset).predict(rgb_path, depth_path, initial_masks, fg_mask)
-> (refined_masks bool [K,H,W] | [], output dict, seconds, fg_mask)``.

Built: file loading, resize to 640x480, ``normalize_depth`` (eval/preprocess_utils.py:12-28; on the device for
uint16 / float32 depth, bit-exact against the imported reference function), nearest depth resize, the HIP predictor,
the OCID zero-depth masking (refiner_model.py:279-288).
The ``dataset == 'armbench'`` branch (refiner_model.py:226-244: RGB only, shortest edge 800 / longest 1333, nearest
resize of the masks) is built too; any frame size is accepted by the HIP path.
The LMFFNet foreground post-filter (refiner_model.py:273-277) runs on the HIP path too when ``foreground_filter=True``
(the reference always applies it; it is opt-in here because the reference's ``rgbd_lmffnet.pth`` is not available and
seeded weights give a meaningless foreground).
Not built: ``cv2.inpaint`` TELEA depth in-painting (zero-depth pixels are left at 0 here; frames without
zero depth are unaffected because the reference only rewrites zero pixels, preprocess_utils.py:63).  cv2 / imageio are absent, so images are read with PIL and resized with PIL's
bilinear filter, which is not bit-identical to cv2.resize.
"""
import time

import numpy as np
from PIL import Image

from ..maskrefiner.predictor import MaskRefinerPredictor

W = 640
H = 480


def normalize_depth(depth, min_val=250.0, max_val=1500.0):
    depth = np.array(depth, dtype=np.float64 if depth.dtype == np.float64 else np.float32)
    depth = np.clip(depth, min_val, max_val)
    depth = (depth - min_val) / (max_val - min_val) * 255
    if depth.ndim == 2:
        depth = depth[..., None]
    return np.uint8(np.repeat(depth, 3, -1))


def _resize_nearest(img, w, h):
    ys = (np.arange(h) * (img.shape[0] / h)).astype(np.int64).clip(0, img.shape[0] - 1)
    xs = (np.arange(w) * (img.shape[1] / w)).astype(np.int64).clip(0, img.shape[1] - 1)
    return img[ys][:, xs]


def resize_shortest_edge_shape(oldh, oldw, short_edge_length=800, max_size=1333):
    """[d2] ResizeShortestEdge.get_output_shape, used by the armbench branch (refiner_model.py:228)."""
    scale = short_edge_length * 1.0 / min(oldh, oldw)
    newh, neww = (short_edge_length, scale * oldw) if oldh < oldw else (scale * oldh, short_edge_length)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


class MaskRefiner:
    def __init__(self, config_file=None, weights_file=None, dataset="OSD", device="cuda:0", foreground_filter=False,
                 lmffnet_weights="./foreground_segmentation/rgbd_lmffnet.pth"):
        self.refiner_predictor = MaskRefinerPredictor(config_file, weights_file=weights_file, device=device)
        self.dataset = dataset
        self.lmffnet = None
        if foreground_filter:
            from ..foreground.predictor import lmffNet
            self.lmffnet = lmffNet(lmffnet_weights, device=device)

    def predict(self, rgb_path, depth_path, initial_masks, fg_mask=None):
        rgb = np.asarray(Image.open(rgb_path).convert("RGB"))[:, :, ::-1]        # BGR like cv2.imread
        if self.dataset == "armbench":
            h, w = resize_shortest_edge_shape(rgb.shape[0], rgb.shape[1], 800, 1333)
            rgb = np.asarray(Image.fromarray(np.ascontiguousarray(rgb)).resize((w, h), Image.BILINEAR))
            initial_masks = np.asarray(initial_masks)
            if initial_masks.dtype == np.bool_:
                initial_masks = np.uint8(initial_masks) * 255
            initial_masks = np.array([_resize_nearest(m, w, h) for m in initial_masks])
            start = time.time()
            output = self.refiner_predictor.predict(np.ascontiguousarray(rgb), None, initial_masks)[0]
            refined = output["instances"].to("cpu").pred_masks.numpy() if "instances" in output else []
            return refined, output, time.time() - start, None
        depth = np.load(depth_path) if "npy" in depth_path else np.asarray(Image.open(depth_path))
        if rgb.shape[:2] != (H, W):
            rgb = np.asarray(Image.fromarray(np.ascontiguousarray(rgb)).resize((W, H), Image.BILINEAR))
        zero_depth = np.where(depth == 0)
        lo, hi = (0.25, 1.5) if "npy" in depth_path else (250.0, 1500.0)
        if depth.ndim == 2 and depth.dtype in (np.uint16, np.float32):
            import torch
            from .. import engine as qengine
            d = torch.from_numpy(np.array(depth)).to(self.refiner_predictor.device)
            depth = qengine.normalize_depth(d, lo, hi)[0].cpu().numpy()
        else:
            depth = normalize_depth(depth, lo, hi)
        if depth.shape[:2] != (H, W):
            depth = _resize_nearest(depth, W, H)
        initial_masks = np.asarray(initial_masks)
        if initial_masks.dtype == np.bool_:
            initial_masks = np.uint8(initial_masks) * 255

        start = time.time()
        output = self.refiner_predictor.predict(np.ascontiguousarray(rgb), depth, initial_masks)[0]
        if "instances" not in output:
            refined = []
        else:
            refined = output["instances"].to("cpu").pred_masks.numpy()
        fg = None
        if self.lmffnet is not None:
            import torch
            from ..foreground.predictor import filter_masks
            dev = self.refiner_predictor.device
            b = torch.from_numpy(np.ascontiguousarray(rgb)[None]).to(dev)
            d = torch.from_numpy(np.ascontiguousarray(depth)[None]).to(dev)
            m = torch.from_numpy(np.ascontiguousarray(refined, dtype=np.uint8)[None]).to(dev) if len(refined) else None
            fg_t, counts = self.lmffnet.net.foreground(b, d, m)
            fg = fg_t[0].cpu().numpy().astype(bool)
            if len(refined):
                refined = np.asarray(filter_masks(refined, counts[0]))
        elapsed = time.time() - start
        if self.dataset == "OCID" and len(refined):
            refined = refined.copy()
            for m in refined:
                m[zero_depth] = False
        return refined, output, elapsed, fg
