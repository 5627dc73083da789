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


class RefinerModel:
    """The ``predictor.model`` object: ``model(list[dict]) -> list[dict]`` in the detectron2 convention
    (reference MaskRefiner.forward, model.py:115-358).  Engines are cached per (H, W, batch capacity)."""

    def __init__(self, cfg, state_dict, device):
        self.cfg = cfg
        self.state_dict = state_dict
        self.device = torch.device(device)
        self._engines = {}
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
                eng.close()
            qc = qengine.make_config(h, w, max_batch=max(batch, 1), max_instances=max(64, n_masks), cfg=self.cfg)
            eng = qengine.Engine(qc, self.device)
            eng.load_state_dict(self.state_dict)
            self._engines[key] = eng
        return eng

    # -- device-side pipeline on already-resident tensors --
    def run(self, bgr, depth, offsets):
        B, H, W = bgr.shape[:3]
        eng = self.engine_for(H, W, B)
        logits = eng.forward(bgr, depth, offsets)
        post = eng.postprocess(logits)
        return eng, logits, post

    def results(self, eng, logits, post):
        """One D2H of the small per-frame tables, then mask extraction for exactly max(count) slots."""
        B = logits.shape[0]
        count = post["count"].cpu().numpy()
        kmax = int(count.max()) if B else 0
        masks = eng.extract_masks(post, kmax) if kmax > 0 else None
        out = []
        qc = eng.qcfg
        ncls, o = qc.error_classes, 4
        for b in range(B):
            r = {"sem_seg": logits[b, 0:1], "panoptic_seg": (post["panoptic"][b], None)}
            o = 4
            if qc.eee_boundary_on:                       # model.py:310-313
                r["eee_boundary"] = logits[b, o:o + ncls]
                o += ncls
            if qc.eee_mask_on:
                r["eee_mask"] = logits[b, o:o + ncls]
            k = int(count[b])
            if k > 0:
                labels = post["labels"][b, :k]
                inst = Instances((eng.H, eng.W))
                inst.pred_masks = masks[b, :k].bool()
                inst.scores = post["scores"][b, :k]
                inst.pred_boxes = Boxes(post["boxes"][b, :k])
                inst.pred_classes = (torch.div(labels, LABEL_DIVISOR, rounding_mode="floor") - 1).to(torch.int64)
                r["instances"] = inst
            out.append(r)
        return out

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
                 seed=0):
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
        if path is not None and os.path.exists(path):
            sd = load_checkpoint(path)
        else:
            if weights_file is not None:
                warnings.warn(f"weights '{weights_file}' not found; using seeded synthetic weights")
            sd = arch.init_state_dict(seed=seed, **kw)
        self.cfg.MODEL.WEIGHTS = path or "<synthetic seed %d>" % seed
        self.model = RefinerModel(self.cfg, sd, device)
        self.device = torch.device(device)

    # -- reference signature (predictor.py:287) --
    def predict(self, rgb_img, depth_img=None, perturbed_masks=None):
        masks = np.zeros((0,) + rgb_img.shape[:2], np.uint8) if perturbed_masks is None else np.asarray(perturbed_masks)
        return self.predict_batch(rgb_img[None], None if depth_img is None else depth_img[None], [masks])

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
