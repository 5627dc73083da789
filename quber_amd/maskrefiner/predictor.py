"""Drop-in for ``maskrefiner.predictor.MaskRefinerPredictor`` (reference maskrefiner/predictor.py:207-359)
running on the MI355X HIP path.

Same constructor and ``predict`` signatures, same return structure (a list with one dict holding
``sem_seg`` [1,H,W] logits, ``eee_boundary`` [4,H,W] logits, ``panoptic_seg`` (f32 [H,W] labels, None)
and, when any instance survives, ``instances`` with ``pred_masks`` bool [K,H,W], ``scores``,
``pred_boxes``, ``pred_classes`` - reference model.py:304-356).  Deliberate differences:
  * none of the reference constructor's side effects (dataset loader, output dir, hard-coded
    weights path: predictor.py:226-243);
  * the initial-mask encoding, network and grouping all run on the GPU; ``predict_batch`` exposes the
    batched form the reference lacks (it always runs batch 1, predictor.py:358).
There is no CPU fallback: construction fails if the HIP library or a GPU is missing.
"""
import os
import warnings

import numpy as np
import torch

from .. import arch, config as qconfig, engine as qengine
from ..structures import Boxes, Instances

LABEL_DIVISOR = 1000  # maskrefiner/data/datasets/register_uoais_sim_panoptic.py:177-186


def load_checkpoint(path):
    """detectron2 .pth ({'model': state_dict}) or a plain state_dict / .npz -> name -> numpy f32."""
    if path.endswith(".npz"):
        z = np.load(path)
        return {k: z[k] for k in z.files}
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck.get("model", ck) if isinstance(ck, dict) else ck
    return {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in sd.items()}


class _FrameStaging:
    """Buffers of the batch-1 call path for one frame size, allocated once: ONE pinned host block and ONE device block for the
    frame's inputs (bgr | depth | initial masks, uploaded in a few pipelined pieces), the device-side intermediates, and pinned
    word for the instance count."""

    def __init__(self, eng, n_cap):
        H, W, dev = eng.H, eng.W, eng.device
        hw = H * W
        self.n_cap = n_cap
        self.pin_in = torch.empty(6 * hw + n_cap * hw, dtype=torch.uint8).pin_memory()
        self.np_in = self.pin_in.numpy()
        self.dev_in = torch.empty(self.pin_in.shape, dtype=torch.uint8, device=dev)
        self.offsets = torch.empty((1, 3, H, W), dtype=torch.float32, device=dev)
        self.post = eng.alloc_post(1)
        self.pin_count = torch.empty((1,), dtype=torch.int32).pin_memory()
        self.done = torch.cuda.Event()


class RefinerModel:
    """The ``predictor.model`` object: ``model(list[dict]) -> list[dict]`` in the detectron2 convention
    (reference MaskRefiner.forward, model.py:115-358).  Engines are cached per (H, W, batch capacity)."""

    def __init__(self, cfg, state_dict, device):
        self.cfg = cfg
        self.state_dict = state_dict
        self.device = torch.device(device)
        self._engines = {}
        self._retired = {}        # (H, W) -> the engine a larger one replaced last; see engine_for
        self._staging = {}
        self.training = False

    def eval(self):
        return self

    def engine_for(self, h, w, batch, n_masks=64):
        key = (h, w)
        eng = self._engines.get(key)
        if eng is None or eng.qcfg.max_batch < batch or eng.qcfg.max_instances < n_masks:
            # grow, never shrink: alternating workloads must not trigger repeated multi-GB rebuilds
            if eng is not None:
                batch, n_masks = max(batch, eng.qcfg.max_batch), max(n_masks, eng.qcfg.max_instances)
                # not closed here: a batch enqueued on it may still be waiting for collect_batch (predict_stream keeps one batch in
                # flight).  The replaced engine is closed as soon as that batch has been collected (collect_batch), or - with nothing
                # in flight - right away; close() below takes whatever is left.
                if getattr(eng, "_in_flight", 0) > 0:
                    self._retired.setdefault(key, []).append(eng)
                else:
                    torch.cuda.current_stream().synchronize()      # its last launches may still be running
                    eng.close()
            qc = qengine.make_config(h, w, max_batch=max(batch, 1), max_instances=max(64, n_masks), cfg=self.cfg)
            eng = qengine.Engine(qc, self.device)
            eng.load_state_dict(self.state_dict)
            self._engines[key] = eng
            self._staging.pop(key, None)
        return eng

    def staging_for(self, eng, n_masks):
        key = (eng.H, eng.W)
        stg = self._staging.get(key)
        if stg is None or stg.n_cap < n_masks:
            stg = _FrameStaging(eng, max(64, n_masks))
            self._staging[key] = stg
        return stg

    # -- device-side pipeline on already-resident tensors --
    def run(self, bgr, depth, offsets):
        B, H, W = bgr.shape[:3]
        eng = self.engine_for(H, W, B)
        logits = eng.forward(bgr, depth, offsets)
        post = eng.postprocess(logits)
        return eng, logits, post

    def frame_dict(self, eng, logits_b, post, b, k, masks_b):
        """The reference's output dict of one frame (model.py:304-356) from the device-side results."""
        qc = eng.qcfg
        ncls, o = qc.error_classes, 4
        r = {"sem_seg": logits_b[0:1], "panoptic_seg": (post["panoptic"][b], None)}
        if qc.eee_boundary_on:                       # model.py:310-313
            r["eee_boundary"] = logits_b[o:o + ncls]
            o += ncls
        if qc.eee_mask_on:
            r["eee_mask"] = logits_b[o:o + ncls]
        if k > 0:
            labels = post["labels"][b, :k]
            inst = Instances((eng.H, eng.W))
            inst.pred_masks = masks_b
            inst.scores = post["scores"][b, :k]
            inst.pred_boxes = Boxes(post["boxes"][b, :k])
            inst.pred_classes = (torch.div(labels, LABEL_DIVISOR, rounding_mode="floor") - 1).to(torch.int64)
            r["instances"] = inst
        return r

    def results(self, eng, logits, post):
        """One D2H of the small per-frame tables, then mask extraction for exactly max(count) slots."""
        B = logits.shape[0]
        count = post["count"].cpu().numpy()
        kmax = int(count.max()) if B else 0
        masks = eng.extract_masks(post, kmax) if kmax > 0 else None
        return [self.frame_dict(eng, logits[b], post, b, int(count[b]), masks[b, :int(count[b])].bool() if count[b] > 0 else None)
                for b in range(B)]

    def predict_one(self, bgr, depth, masks):
        """The reference's call path (one frame per call, predictor.py:287-359) with nothing allocated per call but the
        outputs: the inputs go through one pinned block in a few pipelined H2D copies; the instance count comes back through
        a pinned word and exactly `count` masks are extracted."""
        H, W = bgr.shape[:2]
        n = int(masks.shape[0])
        eng = self.engine_for(H, W, 1, n)
        stg = self.staging_for(eng, n)
        hw = H * W
        two = depth is not None
        stg.done.synchronize()                       # the previous call's H2D copies have read the pinned block
        # host copy into the pinned block and H2D, pipelined: the images first, then the masks in a few pieces - the DMA of a
        # piece runs while the host copies the next one (8 MB at N = 20: 0.3 ms of memcpy + 0.3 ms of PCIe, overlapped)
        o = (6 if two else 3) * hw
        np.copyto(stg.np_in[:3 * hw].reshape(H, W, 3), bgr, casting="unsafe")
        if two:
            np.copyto(stg.np_in[3 * hw:6 * hw].reshape(H, W, 3), depth, casting="unsafe")
        stg.dev_in[:o].copy_(stg.pin_in[:o], non_blocking=True)
        if n:
            # the encoder tests the mask bytes for non-zero (csrc/encode.hip), so uint8 / bool masks upload as they are
            src = masks.view(np.uint8) if masks.dtype == np.bool_ else masks
            step = max(1, (n + 2) // 3)
            for a in range(0, n, step):
                b = min(n, a + step)
                np.copyto(stg.np_in[o + a * hw:o + b * hw].reshape(b - a, H, W), src[a:b], casting="unsafe")
                stg.dev_in[o + a * hw:o + b * hw].copy_(stg.pin_in[o + a * hw:o + b * hw], non_blocking=True)
        stg.done.record()
        d_bgr = stg.dev_in[:3 * hw].view(1, H, W, 3)
        d_dep = stg.dev_in[3 * hw:6 * hw].view(1, H, W, 3) if two else None
        if n:
            eng.encode(stg.dev_in[o:o + n * hw].view(1, n, H, W), stg.offsets)
        else:
            stg.offsets.zero_()
        logits = eng.forward(d_bgr, d_dep, stg.offsets)            # fresh tensor: owned by the caller through the dict
        post = eng.postprocess(logits, stg.post)
        stg.pin_count.copy_(post["count"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        k = int(stg.pin_count[0])
        post_out = {"panoptic": post["panoptic"].clone(), "labels": post["labels"].clone(), "scores": post["scores"].clone(),
                    "boxes": post["boxes"].clone()}
        masks_b = None
        if k > 0:
            masks_b = eng.extract_masks(post, k)[0].view(torch.bool)      # the kernel writes 0 / 1 bytes
        # (the caller's ``.to('cpu')`` of the masks is a plain D2H copy into fresh pageable memory: 0.17 ms for 5 MB, which a
        # prefetch into a pinned buffer plus the copy out of it does not beat - tools/predict_profile.py)
        return self.frame_dict(eng, logits[0], post_out, 0, k, masks_b)

    # -- batched form on device-resident frames, split into "enqueue" and "collect" so that a caller can keep one batch in flight --
    def enqueue_batch(self, d_bgr, d_depth, d_masks, slots=32, capacity=0):
        """d_bgr / d_depth: u8 [B,H,W,3] (depth None for single-stream configs), d_masks: u8 [B,N,H,W] on the device.  Enqueues a1 ...
        a11 and the extraction of the first `slots` instance masks of every frame on the current stream WITHOUT synchronising;
        returns a handle for collect_batch()."""
        B, H, W = d_bgr.shape[:3]
        eng = self.engine_for(H, W, max(B, capacity), d_masks.shape[1])      # (`capacity`: the caller's batch size - a short last batch must not rebuild)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if d_masks.shape[1]:
            offsets = eng.encode(d_masks)
        else:
            offsets = torch.zeros((B, 3, H, W), dtype=torch.float32, device=self.device)
        logits = eng.forward(d_bgr, d_depth, offsets)
        post = eng.postprocess(logits)
        slots = min(max(1, slots), eng.cap)
        masks = eng.extract_masks(post, slots)
        count = torch.empty((B,), dtype=torch.int32).pin_memory()
        count.copy_(post["count"], non_blocking=True)
        e1.record()
        eng._in_flight = getattr(eng, "_in_flight", 0) + 1
        return {"eng": eng, "logits": logits, "post": post, "masks": masks, "slots": slots, "count": count, "e0": e0, "e1": e1}

    def collect_batch(self, hd, host_masks=False):
        """-> (list of the reference's per-frame output dicts, device milliseconds of the whole batch[, per-frame numpy masks]).
        host_masks: also the refined masks of every frame as numpy bool [K_b, H, W] - what the reference's caller takes with
        output['instances'].to('cpu').pred_masks.numpy() - through ONE device-to-host copy for the whole batch (the arrays of a
        batch are views of one host block)."""
        hd["e1"].synchronize()
        eng, post, count = hd["eng"], hd["post"], hd["count"].numpy()
        kmax = int(count.max()) if len(count) else 0
        ready = hd["e1"]
        masks = hd["masks"]
        if kmax > hd["slots"]:                       # rare: more instances than pre-extracted slots - extracted now, on the caller's stream
            masks = eng.extract_masks(post, kmax)
            ready = torch.cuda.Event()
            ready.record()
        outs = [self.frame_dict(eng, hd["logits"][b], post, b, int(count[b]), masks[b, :int(count[b])].view(torch.bool) if count[b] > 0 else None)
                for b in range(len(count))]
        ms = hd["e0"].elapsed_time(hd["e1"])
        eng._in_flight = max(0, getattr(eng, "_in_flight", 1) - 1)
        if not host_masks:
            self._release_retired(eng)
            return outs, ms
        if kmax == 0:
            self._release_retired(eng)
            return outs, ms, [[] for _ in count]
        # The copy runs on a stream of its own, behind this batch's end event only: on the caller's stream it would queue behind
        # the NEXT batch, which predict_stream has already enqueued - and the host would wait 33 ms for masks that are ready.
        side = self._copy_stream()
        side.wait_event(ready)
        with torch.cuda.stream(side):
            host = masks[:, :kmax].contiguous().cpu().numpy().view(np.bool_)
        self._release_retired(eng)
        return outs, ms, [host[b, :int(count[b])] if count[b] > 0 else [] for b in range(len(count))]

    def _release_retired(self, eng):
        """An engine that a larger one has replaced and whose last batch in flight has just been collected: closed now.  (The tensors in
        the dicts collect_batch returned are torch allocations the engine wrote INTO - logits, post tables, masks - not buffers of the
        engine's context, so closing it invalidates nothing the caller holds.)"""
        if getattr(eng, "_in_flight", 0) > 0:
            return
        for key, lst in list(self._retired.items()):
            if eng in lst:
                lst.remove(eng)
                torch.cuda.synchronize(self.device)
                eng.close()
            if not lst:
                self._retired.pop(key, None)

    def close(self):
        """Release every engine (plan buffers, workspaces, weights in kernel layout) this model holds."""
        torch.cuda.synchronize(self.device)
        for lst in self._retired.values():
            for e in lst:
                e.close()
        for e in self._engines.values():
            e.close()
        self._retired.clear()
        self._engines.clear()
        self._staging.clear()

    def _copy_stream(self):
        st = getattr(self, "_d2h_stream", None)
        if st is None:
            st = self._d2h_stream = torch.cuda.Stream(device=self.device)
        return st

    def __call__(self, batched_inputs):
        dev = self.device
        imgs = torch.stack([x["image"] for x in batched_inputs]).to(dev)
        offs = torch.stack([x["initial_pred_offset"] for x in batched_inputs]).to(dev, torch.float32).contiguous()
        nch = 3 * qconfig.arch_kwargs(self.cfg)["streams"]
        if imgs.shape[1] != nch:
            raise ValueError(f"expected a {nch}-channel image, got {imgs.shape[1]}")
        hwc = imgs.to(torch.uint8).permute(0, 2, 3, 1)
        bgr = hwc[..., :3].contiguous()
        depth = hwc[..., 3:].contiguous() if nch == 6 else None
        eng, logits, post = self.run(bgr, depth, offs)
        return self.results(eng, logits, post)


class MaskRefinerPredictor:
    def __init__(self, config_file=None, dataset_name="uoais_sim_val_panoptic", weights_file=None, device="cuda:0",
                 seed=0, state_dict=None):
        if config_file is None:
            self.cfg = qconfig.canonical_cfg()
        else:
            self.cfg = qconfig.merge_from_file(qconfig.get_cfg(), config_file)
        qconfig.validate(self.cfg)
        self.depth_on = self.cfg.INPUT.DEPTH_ON
        self.rgb_on = self.cfg.INPUT.RGB_ON
        self.input_format = self.cfg.INPUT.FORMAT
        assert self.input_format in ["RGB", "BGR"], self.input_format
        self.sigma = 10
        kw = qconfig.arch_kwargs(self.cfg)
        path = weights_file
        if path is not None and not os.path.exists(path) and config_file is not None:
            # the reference derives the path from the config location (predictor.py:222-225)
            path = config_file.replace(".yaml", "/{}".format(weights_file)).replace("configs", "output")
        if state_dict is not None:                       # weights handed over in memory (bench / tests)
            sd, path = state_dict, "<state_dict>"
        elif path is not None and os.path.exists(path):
            sd = load_checkpoint(path)
        else:
            if weights_file is not None:
                warnings.warn(f"weights '{weights_file}' not found; using seeded synthetic weights")
            sd = arch.init_state_dict(seed=seed, **kw)
        self.cfg.MODEL.WEIGHTS = path or "<synthetic seed %d>" % seed
        self.model = RefinerModel(self.cfg, sd, device)
        self.device = torch.device(device)
        self.fast_path = os.environ.get("QUBER_PREDICT_FAST", "1") != "0"     # 0: the general batched path for single frames too

    # -- reference signature (predictor.py:287) --
    def predict(self, rgb_img, depth_img=None, perturbed_masks=None):
        masks = np.zeros((0,) + rgb_img.shape[:2], np.uint8) if perturbed_masks is None else np.asarray(perturbed_masks)
        if not self.fast_path or masks.dtype not in (np.uint8, np.bool_) or masks.ndim != 3:
            return self.predict_batch(rgb_img[None], None if depth_img is None else depth_img[None], [masks])
        if self.depth_on and depth_img is None:
            raise ValueError("this config has INPUT.DEPTH_ON: a depth image is required")
        if not self.rgb_on:                               # depth-only: the image IS the depth map (predictor.py:296-298)
            rgb_img, depth_img = depth_img, None
        elif not self.depth_on:
            depth_img = None
        return [self.model.predict_one(rgb_img, depth_img, masks)]

    def predict_batch(self, rgb_imgs, depth_imgs, masks_list):
        """rgb_imgs/depth_imgs: u8 [B,H,W,3] arrays; masks_list: B arrays u8/bool [N_b,H,W].  -> list of B dicts."""
        B, H, W = rgb_imgs.shape[:3]
        if self.depth_on and depth_imgs is None:
            raise ValueError("this config has INPUT.DEPTH_ON: a depth image is required")
        if not self.rgb_on:                               # depth-only: the image IS the depth map (predictor.py:296-298)
            rgb_imgs, depth_imgs = depth_imgs, None
        elif not self.depth_on:
            depth_imgs = None
        n = max([len(m) for m in masks_list] + [1])
        mk = np.zeros((B, n, H, W), np.uint8)
        for b, m in enumerate(masks_list):
            if len(m):
                mk[b, :len(m)] = np.asarray(m) != 0
        dev = self.device
        eng = self.model.engine_for(H, W, B, n)
        d_masks = torch.from_numpy(mk).to(dev)
        bgr = torch.from_numpy(np.ascontiguousarray(rgb_imgs, dtype=np.uint8)).to(dev)
        depth = None if depth_imgs is None else torch.from_numpy(np.ascontiguousarray(depth_imgs, dtype=np.uint8)).to(dev)
        offsets = eng.encode(d_masks)
        logits = eng.forward(bgr, depth, offsets)
        post = eng.postprocess(logits)
        return self.model.results(eng, logits, post)
