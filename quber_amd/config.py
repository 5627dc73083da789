"""Minimal yaml + ``_BASE_`` config reader for the keys the refiner inference path consumes.

detectron2 / yacs are not available, so this mirrors just enough of
``get_cfg(); add_panoptic_deeplab_config(cfg); add_mask_refiner_config(cfg); cfg.merge_from_file(f)``
(maskrefiner/predictor.py:211-214) for the ~25 keys read on the hot path.  Defaults are those of
maskrefiner/config.py:6-102 and of detectron2 v0.6; yaml keys outside this
set (SOLVER, DATALOADER, ...) are kept verbatim and ignored.
"""
import copy
import os

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        if isinstance(d, dict):
            return CfgNode({k: CfgNode.wrap(v) for k, v in d.items()})
        return d

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = CfgNode.wrap(copy.deepcopy(v))

    def clone(self):
        return CfgNode.wrap(copy.deepcopy(dict(self)))


_DEFAULTS = {
    "VERSION": 2,
    "SEED": -1,
    "MODEL": {
        "META_ARCHITECTURE": "MaskRefiner",
        "DEVICE": "cuda",
        "WEIGHTS": "",
        "PIXEL_MEAN": [103.53, 116.28, 123.675],
        "PIXEL_STD": [1.0, 1.0, 1.0],
        "BACKBONE": {"NAME": "build_resnet_deeplab_rgbd_fusion_backbone", "FREEZE_AT": 2,
                     "FUSION_STRATEGY": "concat", "NUM_FUSION_LAYERS": 3, "WEIGHTS": "", "FREEZE_LAYERS": False},
        "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res4"], "NUM_GROUPS": 1, "NORM": "FrozenBN",
                    "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True, "RES5_DILATION": 1, "RES2_OUT_CHANNELS": 256,
                    "STEM_OUT_CHANNELS": 64, "STEM_TYPE": "deeplab", "RES4_DILATION": 1,
                    "RES5_MULTI_GRID": [1, 2, 4], "DEFORM_ON_PER_STAGE": [False] * 4},
        "SEM_SEG_HEAD": {"USE_DEPTHWISE_SEPARABLE_CONV": False},
        "INS_EMBED_HEAD": {
            "NAME": "PanopticDeepLabInsEmbedHead", "IN_FEATURES": ["res2", "res3", "res5"],
            "PROJECT_FEATURES": ["res2", "res3"], "PROJECT_CHANNELS": [32, 64], "ASPP_CHANNELS": 256,
            "ASPP_DILATIONS": [6, 12, 18], "ASPP_DROPOUT": 0.1, "HEAD_CHANNELS": 32, "CONVS_DIM": 128,
            "COMMON_STRIDE": 4, "NORM": "SyncBN", "EEE_MASK_ON": False, "EEE_POST_PROCESS_ON": False,
            "EEE_BOUNDARY_ON": True, "HIERARCHICAL_FUSION_ON": False,
            "HIERARCHY": [["eee_mask", "eee_boundary"], ["foreground", "center", "offset"]],
            "NUM_FUSION_LAYERS": 3, "FUSION_STRATEGY": "concat", "FUSION_TARGET": ["feat", "pred"],
            "ERROR_TYPE": "e3"},
        "PANOPTIC_DEEPLAB": {"STUFF_AREA": 2048, "CENTER_THRESHOLD": 0.1, "NMS_KERNEL": 7, "TOP_K_INSTANCE": 200,
                             "PREDICT_INSTANCES": True, "USE_DEPTHWISE_SEPARABLE_CONV": False,
                             "SIZE_DIVISIBILITY": -1, "BENCHMARK_NETWORK_SPEED": False},
    },
    "INPUT": {"FORMAT": "BGR", "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "OFFSET_INPUT_ON": False,
              "DEPTH_ON": False, "RGB_ON": True, "GAUSSIAN_SIGMA": 10, "CROP": {"ENABLED": False}},
    "DATASETS": {"TRAIN": (), "TEST": ()},
}


def get_cfg():
    return CfgNode.wrap(copy.deepcopy(_DEFAULTS))


def _load_with_base(path):
    with open(path) as f:
        node = yaml.safe_load(f) or {}
    base = node.pop("_BASE_", None)
    if base is None:
        return node
    if not os.path.isabs(base):
        base = os.path.join(os.path.dirname(path), base)
    merged = CfgNode.wrap(_load_with_base(base))
    merged.merge(node)
    return merged


def merge_from_file(cfg, path):
    cfg.merge(_load_with_base(path))
    return cfg


def arch_kwargs(cfg):
    """The architecture switches of a validated cfg, as keyword arguments of quber_amd.arch.param_specs."""
    m, h = cfg.MODEL, cfg.MODEL.INS_EMBED_HEAD
    return dict(depth=m.RESNETS.DEPTH, backbone_fusion_layers=m.BACKBONE.NUM_FUSION_LAYERS,
                head_fusion_layers=h.NUM_FUSION_LAYERS, error_classes=ERROR_CLASSES[h.ERROR_TYPE],
                eee_mask_on=bool(h.EEE_MASK_ON), eee_boundary_on=bool(h.EEE_BOUNDARY_ON),
                hierarchical=bool(h.HIERARCHICAL_FUSION_ON), hierarchy=tuple(tuple(l) for l in h.HIERARCHY),
                fusion_target=tuple(h.FUSION_TARGET),
                streams=2 if m.BACKBONE.NAME == "build_resnet_deeplab_rgbd_fusion_backbone" else 1,
                fusion_add=m.BACKBONE.FUSION_STRATEGY == "add",
                convs_dim=int(h.CONVS_DIM), head_channels=int(h.HEAD_CHANNELS))


def canonical_cfg():
    """The 'QuBER' architecture of SURVEY.md section 8 (seed77/...-hf-b-fco-l3-b8.yaml over Base-Mask-Refiner.yaml)."""
    cfg = get_cfg()
    cfg.merge({
        "MODEL": {
            "BACKBONE": {"FUSION_STRATEGY": "concat", "NUM_FUSION_LAYERS": 2, "FREEZE_AT": 0},
            "RESNETS": {"OUT_FEATURES": ["res2", "res3", "res5"], "RES5_DILATION": 2},
            "PIXEL_MEAN": [103.53, 116.28, 123.675, 127.5, 127.5, 127.5],
            "PIXEL_STD": [1, 1, 1, 1, 1, 1],
            "INS_EMBED_HEAD": {"NAME": "MaskRefinerInsEmbedHead", "NORM": "GN", "HIERARCHICAL_FUSION_ON": True,
                               "EEE_MASK_ON": False, "EEE_BOUNDARY_ON": True,
                               "HIERARCHY": [["eee_boundary"], ["foreground", "center", "offset"]],
                               "NUM_FUSION_LAYERS": 3, "FUSION_TARGET": ["feat", "pred"], "ERROR_TYPE": "e3"},
            "PANOPTIC_DEEPLAB": {"CENTER_THRESHOLD": 0.3, "NMS_KERNEL": 7, "TOP_K_INSTANCE": 200, "STUFF_AREA": 2048,
                                 "USE_DEPTHWISE_SEPARABLE_CONV": True},
        },
        "INPUT": {"MIN_SIZE_TEST": 480, "MAX_SIZE_TEST": 640, "OFFSET_INPUT_ON": True, "DEPTH_ON": True,
                  "RGB_ON": True},
        "DATASETS": {"TRAIN": ("uoais_sim_train_panoptic",), "TEST": ("uoais_sim_val_panoptic",)},
    })
    return cfg


class UnsupportedConfig(ValueError):
    pass


ERROR_CLASSES = {"e3": 4, "e33": 3, "e2": 2, "e32": 2}


def validate(cfg):
    """Raise UnsupportedConfig for variants outside the built hot path (SURVEY.md 8f rank 4 lists them as 'next')."""
    m, h = cfg.MODEL, cfg.MODEL.INS_EMBED_HEAD

    def need(cond, what):
        if not cond:
            raise UnsupportedConfig(f"quber_amd: {what} is not built yet (see DESIGN.md, out of scope / next)")

    need(m.META_ARCHITECTURE == "MaskRefiner", f"META_ARCHITECTURE {m.META_ARCHITECTURE}")
    need(m.BACKBONE.NAME in ("build_resnet_deeplab_rgbd_fusion_backbone", "build_resnet_deeplab_fusion_backbone"),
         f"backbone {m.BACKBONE.NAME}")
    need(cfg.INPUT.OFFSET_INPUT_ON, "a refiner without the initial-mask offset input")
    if m.BACKBONE.NAME == "build_resnet_deeplab_rgbd_fusion_backbone":
        need(m.BACKBONE.FUSION_STRATEGY in ("concat", "add"), f"backbone FUSION_STRATEGY {m.BACKBONE.FUSION_STRATEGY}")
        need(cfg.INPUT.DEPTH_ON and cfg.INPUT.RGB_ON, "the RGB-D fusion backbone without both RGB_ON and DEPTH_ON")
        need(len(m.PIXEL_MEAN) == 6 and len(m.PIXEL_STD) == 6, "PIXEL_MEAN/STD without 6 entries")
    else:
        need(bool(cfg.INPUT.DEPTH_ON) != bool(cfg.INPUT.RGB_ON), "a single-stream backbone with both RGB_ON and DEPTH_ON")
        need(len(m.PIXEL_MEAN) >= 3 and len(m.PIXEL_STD) >= 3, "PIXEL_MEAN/STD without 3 entries")
    need(list(m.RESNETS.OUT_FEATURES) == ["res2", "res3", "res5"], "RESNETS.OUT_FEATURES other than res2/res3/res5")
    need(m.RESNETS.STEM_TYPE == "deeplab" and m.RESNETS.NORM == "FrozenBN", "non-deeplab stem / non-frozen BN")
    need(m.RESNETS.DEPTH in (50, 101, 152), f"ResNet depth {m.RESNETS.DEPTH}")
    need(h.NAME == "MaskRefinerInsEmbedHead" and h.NORM == "GN", "head other than MaskRefinerInsEmbedHead/GN")
    heads = {"foreground", "center", "offset"} | ({"eee_mask"} if h.EEE_MASK_ON else set()) | \
        ({"eee_boundary"} if h.EEE_BOUNDARY_ON else set())
    if h.HIERARCHICAL_FUSION_ON:
        flat = [k for lvl in h.HIERARCHY for k in lvl]
        # an enabled head that the hierarchy omits is never evaluated by the reference's layers() (model.py:738-762) while its
        # forward() still indexes output_dict[key] (model.py:701-708: KeyError at inference): the reference cannot run such a yaml
        missing = sorted(heads - set(flat))
        if missing:
            raise UnsupportedConfig(f"quber_amd: HIERARCHY omits the enabled head(s) {missing}: the reference itself raises KeyError "
                                    f"for them at inference (maskrefiner/modeling/mask_refiner/model.py:701-708)")
        need(1 <= len(h.HIERARCHY) <= 5 and sorted(flat) == sorted(heads),
             "a HIERARCHY that does not list every enabled head exactly once")
        need(set(h.FUSION_TARGET) <= {"feat", "pred"} and len(h.FUSION_TARGET) > 0 or len(h.HIERARCHY) == 1,
             "FUSION_TARGET outside {feat, pred}")
    need(not m.SEM_SEG_HEAD.USE_DEPTHWISE_SEPARABLE_CONV, "depthwise-separable head convs")
    need(h.ERROR_TYPE in ERROR_CLASSES, f"ERROR_TYPE {h.ERROR_TYPE}")
    need(list(h.PROJECT_CHANNELS) == [32, 64] and h.ASPP_CHANNELS == 256 and h.COMMON_STRIDE == 4
         and list(h.ASPP_DILATIONS) == [6, 12, 18], "non-default PROJECT_CHANNELS / ASPP_CHANNELS / COMMON_STRIDE / ASPP_DILATIONS")
    # CONVS_DIM / HEAD_CHANNELS are plan parameters (GroupNorm(32) needs multiples of 32; the predictor kernel reads 32 or 64 channels)
    need(h.CONVS_DIM in (128, 256) and h.HEAD_CHANNELS in (32, 64), f"CONVS_DIM {h.CONVS_DIM} / HEAD_CHANNELS {h.HEAD_CHANNELS}")
    return cfg
