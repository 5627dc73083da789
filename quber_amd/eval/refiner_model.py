"""Drop-in for the reference's ``eval/refiner_model.py:MaskRefiner`` adapter (reference lines 214-297):
``MaskRefiner(config_file, weights_file, dataset).predict(rgb_path, depth_path, initial_masks, fg_mask)
-> (refined_masks bool [K,H,W] | [], output dict, seconds, fg_mask)``.

Built, all on the device: ``cv2.resize`` of the BGR image to 640x480 (INTER_LINEAR, OpenCV's 8-bit fixed-point path,
``quber_resize_u8``), ``normalize_depth`` (eval/preprocess_utils.py:12-28; bit-exact against the imported reference
function), the nearest depth / mask resizes, the HIP predictor, the LMFFNet foreground post-filter (refiner_model.py:
273-277) and the OCID zero-depth masking (refiner_model.py:279-288).  The ``dataset == 'armbench'`` branch
(refiner_model.py:226-244: RGB only, shortest edge 800 / longest 1333, nearest resize of the masks) is built too.

Return semantics follow the reference exactly: the foreground filter decides which masks survive only for
``dataset == 'OCID'`` (refiner_model.py:279-288 is the only place ``filt_masks`` is returned); every other dataset gets
the unfiltered refined masks plus the foreground mask.  The reference always constructs ``lmffNet()``; here the filter
is opt-in (``foreground_filter=True``) because the reference's ``rgbd_lmffnet.pth`` does not ship and seeded weights give
a meaningless foreground - with it off, ``fg_mask`` is None and the OCID branch masks zero depth on the unfiltered masks.

``inpaint_depth`` (eval/preprocess_utils.py:44-64: the pixels whose NORMALISED depth is 0 - holes and everything nearer than
``min_val`` - are filled with ``cv2.inpaint(..., 3, cv2.INPAINT_TELEA)``): ``inpaint="host"`` (default) runs
``quber_inpaint_depth_u8`` (csrc/inpaint.hip: Telea's fast-marching method restated in the form OpenCV implements it; parity with cv2
unpinned, own tolerance - OpenCV is not in the image) on the calling thread, as the reference does; ``inpaint="device"`` runs
csrc/inpaint_dev.hip, which marches the independent hole regions one wave each and equals the host function bit for bit
(tests/test_gpu_inpaint.py).  ``MaskRefiner(inpaint=False)`` skips it.
cv2 / imageio are absent from the image, so files are read with PIL.
"""
import time

import numpy as np
import torch
from PIL import Image

from .. import engine as qengine
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


def inpaint_depth(depth, kernel_size=3):
    """eval/preprocess_utils.py:44-64 with factor = 1: mask = pixels whose three channels are 0, dilated by a 3x3 square;
    TELEA in-painting with radius 3; only the zero pixels of the input are replaced.  One host call (csrc/inpaint.hip:
    inpaint_depth_u8_host) that releases the GIL from the mask preparation to the merge: predict_stream's worker threads run it
    side by side (the numpy form of the mask preparation held the interpreter lock for ~2 ms per frame)."""
    import ctypes as C
    from .. import _lib
    lib = _lib.load()
    depth = np.ascontiguousarray(depth, dtype=np.uint8)
    if depth.ndim != 3 or depth.shape[2] != 3:
        raise ValueError("inpaint_depth expects a [H, W, 3] uint8 depth image")
    h, w = depth.shape[:2]
    out = np.empty_like(depth)
    _lib.check(lib.quber_inpaint_depth_u8(C.c_void_p(depth.ctypes.data), h, w, kernel_size, C.c_void_p(out.ctypes.data)))
    return out


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
                 lmffnet_weights="./foreground_segmentation/rgbd_lmffnet.pth", inpaint="host"):
        self.refiner_predictor = MaskRefinerPredictor(config_file, weights_file=weights_file, device=device)
        self.dataset = dataset
        self.lmffnet = None
        # "host" / True (default): csrc/inpaint.hip on the calling (worker) thread; "device": csrc/inpaint_dev.hip, bit-equal, for hosts
        # short of CPU cores - a march occupies one wave per hole region for milliseconds and, streamed beside the refiner's batches,
        # costs more GPU time than the host cores it frees are worth on this box: 193 against 278 frames/s (profiles/r09f_adapter_stream.txt);
        # False: none
        self.inpaint = "host" if inpaint is True else inpaint
        if foreground_filter:
            from ..foreground.predictor import lmffNet
            self.lmffnet = lmffNet(lmffnet_weights, device=device)

    def _dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.refiner_predictor.device)

    def _resize(self, img, h, w, linear):
        """cv2.resize(img, (w, h)) on the device; img: numpy u8 [H,W] / [H,W,C]."""
        if img.shape[:2] == (h, w):
            return np.ascontiguousarray(img)
        return qengine.resize_u8(self._dev(img), h, w, linear).cpu().numpy()

    def _load(self, rgb_path, depth_path, initial_masks):
        """File reading and the adapter's pre-processing of one frame (eval/refiner_model.py:224-263): everything before the
        reference's timer starts.  Safe to run on a worker thread (predict_stream): its device work goes to a side stream."""
        rgb = np.asarray(Image.open(rgb_path).convert("RGB"))[:, :, ::-1]        # BGR like cv2.imread
        initial_masks = np.asarray(initial_masks)
        if initial_masks.dtype == np.bool_:
            initial_masks = np.uint8(initial_masks) * 255
        if self.dataset == "armbench":
            h, w = resize_shortest_edge_shape(rgb.shape[0], rgb.shape[1], 800, 1333)
            rgb = self._resize(rgb, h, w, linear=True)                           # cv2.resize(rgb_img, (w, h))
            initial_masks = np.array([self._resize(m, h, w, linear=False) for m in initial_masks])   # INTER_NEAREST
            return {"rgb": np.ascontiguousarray(rgb), "depth": None, "masks": initial_masks, "zero_depth": None}
        depth = np.load(depth_path) if "npy" in depth_path else np.asarray(Image.open(depth_path))
        rgb = self._resize(rgb, H, W, linear=True)                               # cv2.resize(rgb_img, (W, H))
        zero_depth = np.where(depth == 0)
        lo, hi = (0.25, 1.5) if "npy" in depth_path else (250.0, 1500.0)
        # normalise -> nearest resize -> in-paint: on the device from end to end (refiner_model.py:250-255), one copy back for the
        # numpy-in API of the predictor (the batched stream keeps the device tensor)
        if depth.ndim == 2 and depth.dtype in (np.uint16, np.float32):
            d_dev = qengine.normalize_depth(self._dev(np.array(depth)), lo, hi)[0]
        else:
            d_dev = self._dev(normalize_depth(depth, lo, hi))
        if tuple(d_dev.shape[:2]) != (H, W):
            d_dev = qengine.resize_u8(d_dev, H, W, linear=False)                 # cv2.resize(..., INTER_NEAREST)
        if self.inpaint == "device":
            d_dev = qengine.inpaint_depth(d_dev)                                 # refiner_model.py:255, csrc/inpaint_dev.hip
            depth = d_dev.cpu().numpy()
        elif self.inpaint:
            depth = inpaint_depth(d_dev.cpu().numpy())                           # the host function (bit-equal; csrc/inpaint.hip)
            d_dev = None
        else:
            depth = d_dev.cpu().numpy()
        return {"rgb": np.ascontiguousarray(rgb), "depth": depth, "masks": initial_masks, "zero_depth": zero_depth, "depth_dev": d_dev}

    def _refine(self, fr):
        """The reference's timed region and what follows it (eval/refiner_model.py:265-297) on a loaded frame."""
        start = time.time()
        if self.dataset == "armbench":
            output = self.refiner_predictor.predict(fr["rgb"], None, fr["masks"])[0]
        else:
            output = self.refiner_predictor.predict(fr["rgb"], fr["depth"], fr["masks"])[0]
        refined = output["instances"].to("cpu").pred_masks.numpy() if "instances" in output else []
        return self._finish(fr, output, refined, start)

    def _finish(self, fr, output, refined, start):
        """What follows the refiner call inside and after the reference's timed region (eval/refiner_model.py:273-297): the
        LMFFNet foreground filter, the elapsed time, the OCID zero-depth masking."""
        if self.dataset == "armbench":
            return refined, output, time.time() - start, None
        rgb, depth, zero_depth = fr["rgb"], fr["depth"], fr["zero_depth"]
        fg, filt = None, refined
        if self.lmffnet is not None:
            from ..foreground.predictor import filter_masks
            m = self._dev(np.ascontiguousarray(refined, dtype=np.uint8)[None]) if len(refined) else None
            fg_t, counts = self.lmffnet.net.foreground(self._dev(rgb[None]), self._dev(depth[None]), m)
            fg = fg_t[0].cpu().numpy().astype(bool)
            if len(refined):
                filt = np.asarray(filter_masks(refined, counts[0]))              # inter / area > 0.3 (refiner_model.py:274-276)
        elapsed = time.time() - start
        if self.dataset == "OCID":
            # refiner_model.py:279-288: only here do the FILTERED masks (minus zero-depth pixels) replace refined_masks
            out = []
            for m in filt:
                m = m.copy()
                m[zero_depth] = False
                out.append(m)
            refined = np.asarray(out)
        return refined, output, elapsed, fg

    def predict(self, rgb_path, depth_path, initial_masks, fg_mask=None):
        return self._refine(self._load(rgb_path, depth_path, initial_masks))

    def predict_stream(self, items, workers=2, batch=1):
        """items: iterable of (rgb_path, depth_path, initial_masks[, fg_mask]) -> yields predict()'s tuple per item, in order.
        The reference's evaluation loop (eval/eval_utils.py:235-286) calls predict() frame after frame; the host side of a frame -
        file decoding and, above all, the TELEA depth in-painting (9-14 ms, two to three times the refiner's GPU time) - is
        independent of the previous frame's refinement, so it runs ahead on `workers` threads (the in-painting is a ctypes call
        and releases the GIL; the workers' device work - resize, depth normalisation, and with batch > 1 the upload of the frame -
        goes to their own HIP stream).

        batch = k > 1: k consecutive frames of one size are refined by ONE engine call (the batched form of a1 ... a11 the
        reference lacks: it always runs batch 1, predictor.py:358), with the next batch already enqueued while the results of the
        current one are copied out - the GPU's batched throughput through the reference's own adapter API.  The per-frame tuples
        are the ones predict() returns (`seconds` = the batch's device time / k); a frame's logits may differ from its batch-1
        logits in the last bits (split-K partitions follow the launch size), which tests/test_gpu_network.py bounds and explains."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        import threading
        dev = self.refiner_predictor.device
        sides = {}
        k = max(1, int(batch))

        def load(item):
            tid = threading.get_ident()
            if tid not in sides:
                sides[tid] = torch.cuda.Stream(device=dev)
            side = sides[tid]
            with torch.cuda.stream(side):
                fr = self._load(item[0], item[1], item[2])
                if k > 1:                        # the frame goes to the device here, off the main thread
                    m = fr["masks"]
                    fr["d_rgb"] = torch.from_numpy(fr["rgb"]).to(dev, non_blocking=True)
                    if fr.get("depth_dev") is not None:
                        fr["d_depth"] = fr["depth_dev"]          # already there (normalised, resized, in-painted on the device)
                    else:
                        fr["d_depth"] = None if fr["depth"] is None else torch.from_numpy(np.ascontiguousarray(fr["depth"])).to(dev, non_blocking=True)
                    # (the encoder tests the mask bytes for non-zero, csrc/encode.hip: bool / 0-255 masks upload as they are)
                    fr["d_masks"] = torch.from_numpy(np.ascontiguousarray(m.view(np.uint8) if m.dtype == np.bool_ else m.astype(np.uint8, copy=False))).to(dev, non_blocking=True)
                    fr["ready"] = torch.cuda.Event()
                    fr["ready"].record(side)
                    side.synchronize()           # (pageable sources: the copies are complete when this returns)
                return fr

        def enqueue(frs):
            model = self.refiner_predictor.model
            n = max(f["d_masks"].shape[0] for f in frs)
            H_, W_ = frs[0]["d_rgb"].shape[:2]
            for f in frs:
                torch.cuda.current_stream().wait_event(f["ready"])
            d_masks = torch.zeros((len(frs), n, H_, W_), dtype=torch.uint8, device=dev)
            for b, f in enumerate(frs):
                if f["d_masks"].shape[0]:
                    d_masks[b, :f["d_masks"].shape[0]] = f["d_masks"]
            d_rgb = torch.stack([f["d_rgb"] for f in frs])
            two = self.refiner_predictor.depth_on and self.refiner_predictor.rgb_on
            if not self.refiner_predictor.rgb_on:                 # depth-only: the image IS the depth map (predictor.py:296-298)
                d_rgb = torch.stack([f["d_depth"] for f in frs])
            d_depth = torch.stack([f["d_depth"] for f in frs]) if two else None
            return model.enqueue_batch(d_rgb, d_depth, d_masks, slots=max(32, n + 12), capacity=k)

        def collect(frs, hd):
            outs, ms, host = self.refiner_predictor.model.collect_batch(hd, host_masks=True)
            for fr, output, refined in zip(frs, outs, host):
                # `seconds` of a streamed frame = its share of the batch's device time (+ the post-filter, if any, inside _finish)
                yield self._finish(fr, output, refined, time.time() - ms * 1e-3 / len(frs))

        with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
            it = iter(items)
            ahead = deque()
            depth = max(workers, 2 * k if k > 1 else 0)
            for item in it:
                ahead.append(pool.submit(load, item))
                if len(ahead) > depth:
                    break

            def next_frame():
                fr = ahead.popleft().result()
                nxt = next(it, None)
                if nxt is not None:
                    ahead.append(pool.submit(load, nxt))
                return fr

            if k == 1:
                while ahead:
                    yield self._refine(next_frame())
                return
            pending = None                        # (frames, handle) of the batch in flight
            carry = None                          # a loaded frame that did not fit the current group (other size)
            while ahead or carry is not None:
                group = [carry] if carry is not None else []
                carry = None
                while len(group) < k and ahead:
                    fr = next_frame()
                    if group and fr["d_rgb"].shape != group[0]["d_rgb"].shape:
                        carry = fr
                        break
                    group.append(fr)
                hd = enqueue(group)
                if pending is not None:
                    yield from collect(*pending)
                pending = (group, hd)
            if pending is not None:
                yield from collect(*pending)
