#!/usr/bin/env python3
"""EXTRA EVIDENCE for the network oracle (it pins nothing: VERDICT r05 item 6c).  TEST INFRASTRUCTURE, build container only.

The reference's own module code - maskrefiner/modeling/mask_refiner/model.py (MaskRefiner.forward :115-358, SinglePredictionHead /
SinglePredictor / FusionLayers :369-458, MaskRefinerInsEmbedHead :461-764 with the hierarchy loop :738-762) and
maskrefiner/modeling/backbone/resnet.py (DeepLabStem :24-76, ResNet :128-330, build_resnet_deeplab_fusion_backbone :358-449,
RGBDFusionBackbone :453-507) - cannot be imported here because detectron2, fvcore and monai are absent.  This script supplies
STAND-INS for the detectron2 / fvcore / monai symbols those two files import (restated from detectron2 v0.6's documented behaviour,
SURVEY.md Appendix B; every stand-in is a few lines of torch), imports the two reference files unmodified by file path, builds
`MaskRefiner(cfg)`, loads quber_amd.arch's seeded state_dict into it (keys must match exactly), runs one small frame through the
reference-authored forward and stores inputs + outputs as tests/golden/wiring_<variant>.npz.

What that executes that no other fixture does: the reference's module wiring - channel order of the RGB-D input split, the
"depth_" key prefix, stage strides / dilations / multi-grid, the fusion Sequentials, the cat order [y, feat..., act(pred)...] of the
hierarchy loop, one FusionLayers per level, softmax vs sigmoid per key, the x4 up-sampling and offset scaling, post-processing and
the per-instance extraction of model.py:313-356.  What it does NOT do: pin the detectron2 parts (Conv2d wrapper, FrozenBatchNorm2d,
BottleneckBlock, ASPP, DeepLabV3PlusHead.layers, ImageList, sem_seg_postprocess, BitMasks) - those are this file's stand-ins, i.e.
the same recollection of detectron2 the oracle itself rests on.  tests/test_oracle_golden.py::test_reference_wiring_fixture compares
oracle/network_torch.py + oracle/postproc_ref.py with the stored outputs.

usage (build container): python3 oracle/gen_wiring.py [variant ...]   -> tests/golden/wiring_*.npz
"""
import copy
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------------------------------------------------------------------
# stand-ins for the detectron2 / fvcore / monai symbols the two reference files import
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


class Registry(dict):                                   # fvcore.common.registry.Registry: register() decorator + get()
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            return lambda o: self.register(o) or o
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


def configurable(init):                                 # detectron2.config.configurable on __init__: cls(cfg, ...) -> cls(**from_config(cfg, ...))
    def wrapped(self, *args, **kwargs):
        first = args[0] if args else kwargs.get("cfg")
        if hasattr(first, "MODEL") and hasattr(type(self), "from_config"):
            init(self, **type(self).from_config(*args, **kwargs))
        else:
            init(self, *args, **kwargs)
    return wrapped


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class FrozenBatchNorm2d(nn.Module):                     # [d2] eps 1e-5; y = x * (w * rsqrt(var + eps)) + (b - mean * scale)
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def get_norm(norm, out_channels):
    if norm is None or (isinstance(norm, str) and len(norm) == 0):
        return None
    if isinstance(norm, str):
        norm = {"BN": nn.BatchNorm2d, "SyncBN": nn.BatchNorm2d, "FrozenBN": FrozenBatchNorm2d,
                "GN": lambda c: nn.GroupNorm(32, c)}[norm]
    return norm(out_channels)


class Conv2d(nn.Conv2d):                                # [d2] wrapper: conv -> norm (if any) -> activation (if any)
    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


class CNNBlockBase(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self


class Backbone(nn.Module):
    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


class BottleneckBlock(CNNBlockBase):                    # [d2] detectron2/modeling/backbone/resnet.py
    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN", stride_in_1x1=False,
                 dilation=1):
        super().__init__(in_channels, out_channels, stride)
        self.shortcut = (Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False, norm=get_norm(norm, out_channels))
                         if in_channels != out_channels else None)
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=s1, bias=False, norm=get_norm(norm, bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=s3, padding=1 * dilation, bias=False,
                            groups=num_groups, dilation=dilation, norm=get_norm(norm, bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False, norm=get_norm(norm, out_channels))

    def forward(self, x):
        out = F.relu_(self.conv1(x))
        out = F.relu_(self.conv2(out))
        out = self.conv3(out)
        out = out + (self.shortcut(x) if self.shortcut is not None else x)
        return F.relu_(out)


class _Unavailable(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("not on the refiner's inference path")


class ASPP(nn.Module):                                  # [d2] detectron2/layers/aspp.py
    def __init__(self, in_channels, out_channels, dilations, *, norm, activation, pool_kernel_size=None, dropout=0.0,
                 use_depthwise_separable_conv=False):
        super().__init__()
        assert len(dilations) == 3 and not use_depthwise_separable_conv
        self.pool_kernel_size, self.dropout = pool_kernel_size, dropout
        use_bias = norm == ""
        self.convs = nn.ModuleList()
        self.convs.append(Conv2d(in_channels, out_channels, kernel_size=1, bias=use_bias, norm=get_norm(norm, out_channels),
                                 activation=copy.deepcopy(activation)))
        for d in dilations:
            self.convs.append(Conv2d(in_channels, out_channels, kernel_size=3, padding=d, dilation=d, bias=use_bias,
                                     norm=get_norm(norm, out_channels), activation=copy.deepcopy(activation)))
        pool = nn.AdaptiveAvgPool2d(1) if pool_kernel_size is None else nn.AvgPool2d(kernel_size=pool_kernel_size, stride=1)
        self.convs.append(nn.Sequential(pool, Conv2d(in_channels, out_channels, 1, bias=True, activation=copy.deepcopy(activation))))
        self.project = Conv2d(5 * out_channels, out_channels, kernel_size=1, bias=use_bias, norm=get_norm(norm, out_channels),
                              activation=copy.deepcopy(activation))

    def forward(self, x):
        size = x.shape[-2:]
        res = [conv(x) for conv in self.convs]
        res[-1] = F.interpolate(res[-1], size=size, mode="bilinear", align_corners=False)
        res = self.project(torch.cat(res, dim=1))
        return F.dropout(res, self.dropout, training=self.training) if self.dropout > 0 else res


class DeepLabV3PlusHead(nn.Module):                     # [d2] detectron2/projects/deeplab/semantic_seg.py, decoder_only form
    def __init__(self, input_shape, *, project_channels, aspp_dilations, aspp_dropout, decoder_channels, common_stride, norm,
                 train_size, loss_weight=1.0, loss_type="cross_entropy", ignore_value=-1, num_classes=None,
                 use_depthwise_separable_conv=False):
        super().__init__()
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        in_channels = [x[1].channels for x in input_shape]
        aspp_channels = decoder_channels[-1]
        self.common_stride = common_stride
        self.decoder_only = num_classes is None
        self.use_depthwise_separable_conv = use_depthwise_separable_conv
        assert self.decoder_only and not use_depthwise_separable_conv and train_size is None
        assert len(project_channels) == len(self.in_features) - 1 and len(decoder_channels) == len(self.in_features)
        self.decoder = nn.ModuleDict()
        use_bias = norm == ""
        for idx, in_ch in enumerate(in_channels):
            stage = nn.ModuleDict()
            if idx == len(self.in_features) - 1:
                project_conv = ASPP(in_ch, aspp_channels, aspp_dilations, norm=norm, activation=F.relu, pool_kernel_size=None,
                                    dropout=aspp_dropout, use_depthwise_separable_conv=False)
                fuse_conv = None
            else:
                project_conv = Conv2d(in_ch, project_channels[idx], kernel_size=1, bias=use_bias,
                                      norm=get_norm(norm, project_channels[idx]), activation=F.relu)
                fuse_conv = nn.Sequential(
                    Conv2d(project_channels[idx] + decoder_channels[idx + 1], decoder_channels[idx], kernel_size=3, padding=1, bias=use_bias,
                           norm=get_norm(norm, decoder_channels[idx]), activation=F.relu),
                    Conv2d(decoder_channels[idx], decoder_channels[idx], kernel_size=3, padding=1, bias=use_bias,
                           norm=get_norm(norm, decoder_channels[idx]), activation=F.relu))
            stage["project_conv"] = project_conv
            stage["fuse_conv"] = fuse_conv
            self.decoder[self.in_features[idx]] = stage

    def layers(self, features):
        y = None
        for f in self.in_features[::-1]:
            proj_x = self.decoder[f]["project_conv"](features[f])
            if self.decoder[f]["fuse_conv"] is None:
                y = proj_x
            else:
                y = F.interpolate(y, size=proj_x.size()[2:], mode="bilinear", align_corners=False)
                y = self.decoder[f]["fuse_conv"](torch.cat([proj_x, y], dim=1))
        return y


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        h, w = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            h, w = -(-h // size_divisibility) * size_divisibility, -(-w // size_divisibility) * size_divisibility
        out = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (h, w), pad_value)
        for i, t in enumerate(tensors):
            out[i, ..., :t.shape[-2], :t.shape[-1]] = t
        return ImageList(out, sizes)


def sem_seg_postprocess(result, img_size, output_height, output_width):
    result = result[:, :img_size[0], :img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


class Boxes:
    def __init__(self, tensor):
        self.tensor = tensor

    @staticmethod
    def cat(lst):
        return Boxes(torch.cat([b.tensor for b in lst], 0))


class BitMasks:
    def __init__(self, tensor):
        self.tensor = tensor.to(torch.bool)

    def get_bounding_boxes(self):
        boxes = torch.zeros(self.tensor.shape[0], 4, dtype=torch.float32)
        x_any, y_any = torch.any(self.tensor, dim=1), torch.any(self.tensor, dim=2)
        for i in range(self.tensor.shape[0]):
            x, y = torch.where(x_any[i])[0], torch.where(y_any[i])[0]
            if len(x) > 0 and len(y) > 0:
                boxes[i] = torch.as_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1], dtype=torch.float32)
        return Boxes(boxes)


class Instances:
    def __init__(self, image_size, **kw):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", dict(kw))

    def __setattr__(self, k, v):
        self._fields[k] = v

    def __getattr__(self, k):
        if k.startswith("_") or k not in self._fields:
            raise AttributeError(k)
        return self._fields[k]

    @staticmethod
    def cat(lst):
        out = Instances(lst[0]._image_size)
        for k in lst[0]._fields:
            vals = [i._fields[k] for i in lst]
            out._fields[k] = Boxes.cat(vals) if isinstance(vals[0], Boxes) else torch.cat(vals, 0)
        return out


class _Meta:
    thing_dataset_id_to_contiguous_id = {1: 0}          # tools/register_uoais_sim_panoptic.py:177-186: one "object" class
    label_divisor = 1000


def install_stand_ins():
    fill = lambda m: None                               # the weights are overwritten by load_state_dict
    _module("fvcore"); _module("fvcore.nn"); _module("fvcore.nn.weight_init", c2_msra_fill=fill, c2_xavier_fill=fill)
    backbone_reg, meta_reg = Registry("BACKBONE"), Registry("META_ARCH")
    _module("detectron2")
    _module("detectron2.config", configurable=configurable)
    _module("detectron2.data", MetadataCatalog=types.SimpleNamespace(get=lambda name: _Meta))
    _module("detectron2.layers", Conv2d=Conv2d, DepthwiseSeparableConv2d=_Unavailable, ShapeSpec=ShapeSpec, get_norm=get_norm,
            CNNBlockBase=CNNBlockBase, DeformConv=_Unavailable, ModulatedDeformConv=_Unavailable)
    _module("detectron2.modeling", META_ARCH_REGISTRY=meta_reg, SEM_SEG_HEADS_REGISTRY=Registry("SEM_SEG_HEADS"), BACKBONE_REGISTRY=backbone_reg,
            build_backbone=lambda cfg: backbone_reg.get(cfg.MODEL.BACKBONE.NAME)(cfg, ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))),
            build_sem_seg_head=lambda *a, **k: None)
    _module("detectron2.modeling.backbone", Backbone=Backbone)
    _module("detectron2.modeling.backbone.resnet", BasicStem=_Unavailable, BottleneckBlock=BottleneckBlock, DeformBottleneckBlock=_Unavailable,
            BasicBlock=_Unavailable, Backbone=Backbone)
    _module("detectron2.modeling.backbone.build", BACKBONE_REGISTRY=backbone_reg)
    _module("detectron2.modeling.postprocessing", sem_seg_postprocess=sem_seg_postprocess)
    _module("detectron2.projects"); _module("detectron2.projects.deeplab", DeepLabV3PlusHead=DeepLabV3PlusHead)
    _module("detectron2.projects.deeplab.loss", DeepLabCE=_Unavailable)
    _module("detectron2.structures", BitMasks=BitMasks, ImageList=ImageList, Instances=Instances)
    _module("detectron2.utils"); _module("detectron2.utils.registry", Registry=Registry)
    _module("monai"); _module("monai.losses", DiceLoss=lambda **k: None)


def import_reference():
    """The two reference files, unmodified, by path; model.py's `from .post_processing import ...` resolves inside a package stub."""
    install_stand_ins()
    spec = importlib.util.spec_from_file_location("ref_resnet", os.path.join(REF, "maskrefiner/modeling/backbone/resnet.py"))
    resnet = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(resnet)
    pkg = types.ModuleType("ref_mask_refiner")
    pkg.__path__ = [os.path.join(REF, "maskrefiner/modeling/mask_refiner")]
    sys.modules["ref_mask_refiner"] = pkg
    spec = importlib.util.spec_from_file_location("ref_mask_refiner.model", os.path.join(pkg.__path__[0], "model.py"))
    model = importlib.util.module_from_spec(spec)
    sys.modules["ref_mask_refiner.model"] = model
    spec.loader.exec_module(model)
    return resnet, model


def reference_cfg(**kw):
    """quber_amd.config's canonical cfg + the keys only the reference's constructors read (losses, deformable convs, crop)."""
    from quber_amd import config as qconfig
    cfg = qconfig.canonical_cfg()
    h = cfg.MODEL.INS_EMBED_HEAD
    h.merge({"FOREGROUND_LOSS_WEIGHT": 1.0, "FOREGROUND_LOSS_TYPE": "hard_pixel_mining", "FOREGROUND_LOSS_TOP_K": 0.2,
             "CENTER_LOSS_WEIGHT": 200.0, "OFFSET_LOSS_WEIGHT": 0.01, "EEE_MASK_LOSS_TYPE": "cross_entropy", "EEE_MASK_LOSS_WEIGHT": 1.0,
             "EEE_BOUNDARY_LOSS_TYPE": "cross_entropy", "EEE_BOUNDARY_LOSS_WEIGHT": 1.0})
    cfg.MODEL.RESNETS.merge({"DEFORM_MODULATED": False, "DEFORM_NUM_GROUPS": 1})
    cfg.MODEL.SEM_SEG_HEAD.USE_DEPTHWISE_SEPARABLE_CONV = False
    for k, v in kw.items():
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return qconfig.validate(cfg)


VARIANTS = {
    # the canonical QuBER config (seed77/...-hf-b-fco-l3-b8.yaml)
    "b_fco": {},
    # eval/run_eval.py:15's default: 5 levels, e2, mask + boundary, FUSION_TARGET [pred, feat]
    "m_b_f_c_o_e2": {"MODEL.INS_EMBED_HEAD.EEE_MASK_ON": True, "MODEL.INS_EMBED_HEAD.ERROR_TYPE": "e2", "MODEL.INS_EMBED_HEAD.FUSION_TARGET": ["pred", "feat"],
                     "MODEL.INS_EMBED_HEAD.HIERARCHY": [["eee_mask"], ["eee_boundary"], ["foreground"], ["center"], ["offset"]]},
    # ...-l2-b2-cdim256-hcha64.yaml
    "cdim256_hcha64": {"MODEL.INS_EMBED_HEAD.EEE_MASK_ON": True, "MODEL.INS_EMBED_HEAD.NUM_FUSION_LAYERS": 2, "MODEL.INS_EMBED_HEAD.CONVS_DIM": 256,
                       "MODEL.INS_EMBED_HEAD.HEAD_CHANNELS": 64,
                       "MODEL.INS_EMBED_HEAD.HIERARCHY": [["eee_mask"], ["eee_boundary"], ["foreground"], ["center"], ["offset"]]},
    # Base-Mask-Refiner.yaml's backbone defaults: add-fusion, three fusion layers; flat (non-hierarchical) heads
    "add_l3_flat": {"MODEL.BACKBONE.FUSION_STRATEGY": "add", "MODEL.BACKBONE.NUM_FUSION_LAYERS": 3, "MODEL.INS_EMBED_HEAD.HIERARCHICAL_FUSION_ON": False},
    # a deeper backbone (resnet.py:330-356: 3 / 4 / 23 / 3 blocks) with the three-class error map e33
    "r101_e33": {"MODEL.RESNETS.DEPTH": 101, "MODEL.INS_EMBED_HEAD.ERROR_TYPE": "e33"},
    # mask error map only (no boundary head), two classes (e32), one head-fusion layer, three levels
    "m_f_co_e32_l1": {"MODEL.INS_EMBED_HEAD.EEE_MASK_ON": True, "MODEL.INS_EMBED_HEAD.EEE_BOUNDARY_ON": False, "MODEL.INS_EMBED_HEAD.ERROR_TYPE": "e32",
                      "MODEL.INS_EMBED_HEAD.NUM_FUSION_LAYERS": 1,
                      "MODEL.INS_EMBED_HEAD.HIERARCHY": [["eee_mask"], ["foreground"], ["center", "offset"]]},
}


def stable_under_perturbation(fg, center, offset, pan, trials=6, eps=1e-4):
    """Would another correct fp32 evaluation (logits within 1e-4 in head units; offsets x4) give the same label map?  A fixture whose
    instance sits at the 512-pixel area filter, or whose centre peak sits at the 0.3 threshold, flips wholesale on 1e-5 of noise
    and would test nothing but that coincidence."""
    from oracle import postproc_ref
    g = torch.Generator().manual_seed(1)
    for _ in range(trials):
        nz = lambda t, s: t + (torch.rand(t.shape, generator=g) * 2 - 1) * s
        o = postproc_ref.postprocess(nz(fg, eps), nz(center, eps), nz(offset, 4 * eps))
        if float((o["panoptic"] != pan).float().mean()) > 2e-3:
            return False
    return True


def run_scene(net, arch, kw, sc, weight_seed, n):
    """One frame through the reference-authored forward on the seeded loud weights (centre bias calibrated on the reference's own centre
    logits); None if no instance survives.  out["stable"]: the label map survives 1e-4 perturbations of the logits."""
    from oracle import encode_np
    h, w = sc["rgb"].shape[:2]
    offs = encode_np.encode_initial_masks(sc["masks"])
    image = torch.from_numpy(np.concatenate([sc["rgb"], sc["depth"]], -1)).permute(2, 0, 1).float()
    inp = [{"image": image, "initial_pred_offset": torch.from_numpy(offs), "height": h, "width": w}]

    def load(sd):
        ref_keys = {k for k in net.state_dict() if not k.endswith("num_batches_tracked")}
        assert ref_keys == set(sd), sorted(ref_keys ^ set(sd))[:8]          # the reference's module tree names every tensor as arch.param_specs does
        missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)

    grabbed = {}
    hook = net.ins_embed_head.register_forward_hook(lambda m, i, o: grabbed.update({k: v.detach().clone() for k, v in o[0].items()}))
    try:
        load(arch.init_state_dict(seed=weight_seed, loud_heads=True, **kw))
        with torch.no_grad():
            net(inp)
        bias = arch.calibrate_center_bias(grabbed["center"], n)
        load(arch.init_state_dict(seed=weight_seed, loud_heads=True, center_bias=bias, **kw))
        with torch.no_grad():
            res = net(inp)[0]
    finally:
        hook.remove()
    if "instances" not in res:               # no instance survived: the fixture would not exercise model.py:313-356
        return None
    stable = stable_under_perturbation(grabbed["foreground"][0], grabbed["center"][0], grabbed["offset"][0], res["panoptic_seg"][0])
    out = {"stable": np.bool_(stable),"rgb": sc["rgb"], "depth": sc["depth"], "masks": sc["masks"], "offsets": offs, "seed": np.int64(weight_seed),
           "center_bias": np.float64(bias), "arch_kwargs": np.array(repr(kw)),
           "panoptic": res["panoptic_seg"][0].numpy(), "sem_seg": res["sem_seg"].numpy()}
    for k, v in grabbed.items():
        out["head_" + k] = v.numpy()
    for k in ("eee_boundary", "eee_mask"):
        if k in res:
            out["out_" + k] = res[k].numpy()
    ins = res["instances"]
    out.update(inst_masks=ins.pred_masks.numpy(), inst_scores=ins.scores.numpy(), inst_boxes=ins.pred_boxes.tensor.numpy(),
               inst_classes=ins.pred_classes.numpy())
    return out


def main():
    from quber_amd import arch, config as qconfig, synth
    _, model = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    h, w, n = 96, 128, 3           # (at 64 x 96 the instances sit near the 512-pixel area filter: no stable scene)
    only = set(sys.argv[1:])               # optional: the variants to (re)generate; the seeds of a variant depend on its position only
    for vi, (name, over) in enumerate(VARIANTS.items()):
        if only and name not in only:
            continue
        cfg = reference_cfg(**over)
        kw = qconfig.arch_kwargs(cfg)
        net = model.MaskRefiner(cfg).eval()
        # the first scene whose label map survives 1e-4 perturbations of the logits; failing that, the first with an instance, flagged
        # `stable = False` (random loud heads put the centre maxima on the frame's border, where the x4 bilinear up-sampling repeats
        # rows: exact two-pixel plateaus, two centres per maximum (post_processing.py:9-41 keeps both), and which of the twins owns a
        # pixel - hence which instance passes the 512-pixel filter - turns on the last bit of a logit)
        out, fallback = None, None
        for scene_seed in range(40 + 12 * vi, 52 + 12 * vi):
            cand = run_scene(net, arch, kw, synth.make_scene(scene_seed, h, w, n), 30 + vi, n)
            if cand is None:
                continue
            cand["scene_seed"] = np.int64(scene_seed)
            if bool(cand["stable"]):
                out = cand
                break
            fallback = fallback or cand
        out = out or fallback
        assert out is not None, name
        scene_seed = int(out["scene_seed"])
        np.savez_compressed(os.path.join(OUT, f"wiring_{name}.npz"), **out)
        print(f"wiring_{name}.npz: scene {scene_seed} (stable {bool(out['stable'])}), {len(net.state_dict())} tensors in the reference's module tree, heads "
              f"{sorted(k[5:] for k in out if k.startswith('head_'))}, {len(out['inst_scores'])} instances, labels {sorted(set(out['panoptic'].ravel().tolist()))}")


if __name__ == "__main__":
    main()
