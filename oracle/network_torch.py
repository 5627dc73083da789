"""ORACLE (test infrastructure, never shipped): pure-torch CPU restatement of the QuBER
mask-refiner network (encoder + decoder + hierarchical error->refine heads).

Parity status: **unpinned** for this file. The reference network cannot be imported here
(detectron2 / fvcore / monai are absent), so this restatement was written by reading
  - maskrefiner/modeling/backbone/resnet.py:24-76   (DeepLabStem)
  - maskrefiner/modeling/backbone/resnet.py:358-449 (stage layout: strides, dilations, multi-grid)
  - maskrefiner/modeling/backbone/resnet.py:453-507 (RGBDFusionBackbone: split 3/3/3, cat, 1x1+GN+ReLU, 3x3+GN+ReLU)
  - maskrefiner/modeling/mask_refiner/model.py:369-458 (SinglePredictionHead, SinglePredictor, FusionLayers)
  - maskrefiner/modeling/mask_refiner/model.py:610-651 (decoder_channels = [128,128,256])
  - maskrefiner/modeling/mask_refiner/model.py:689-764 (x4 bilinear, offset*4, hierarchy loop)
  - maskrefiner/modeling/mask_refiner/model.py:137-153 (normalise + concat)
and the detectron2 v0.6 layouts listed in SURVEY.md Appendix B (BottleneckBlock, ASPP,
DeepLabV3PlusHead, FrozenBatchNorm2d).  Module/attribute names are chosen so that
``state_dict()`` keys equal the reference's, which is the structural pin
tests/test_oracle_network.py checks (key list + parameter count).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import nn


@dataclass
class ArchCfg:
    depth: int = 50
    stem_out: int = 64
    res2_out: int = 256
    res5_dilation: int = 2
    res5_multi_grid: List[int] = field(default_factory=lambda: [1, 2, 4])
    backbone_fusion_layers: int = 2          # MODEL.BACKBONE.NUM_FUSION_LAYERS
    project_channels: List[int] = field(default_factory=lambda: [32, 64])
    aspp_channels: int = 256
    aspp_dilations: List[int] = field(default_factory=lambda: [6, 12, 18])
    head_channels: int = 32
    convs_dim: int = 128
    common_stride: int = 4
    head_fusion_layers: int = 3              # MODEL.INS_EMBED_HEAD.NUM_FUSION_LAYERS
    error_classes: int = 4                   # ERROR_TYPE e3 -> 4, e33 -> 3, e2 / e32 -> 2
    eee_mask_on: bool = False
    eee_boundary_on: bool = True
    hierarchical: bool = True
    hierarchy: List[List[str]] = field(default_factory=lambda: [["eee_boundary"], ["foreground", "center", "offset"]])
    fusion_target: List[str] = field(default_factory=lambda: ["feat", "pred"])
    fusion_add: bool = False                 # MODEL.BACKBONE.FUSION_STRATEGY "add"
    streams: int = 2                         # 2: RGBDFusionBackbone; 1: plain ResNet (rgb-only / depth-only configs)
    repeat_fusion: bool = False              # re-evaluate the head-fusion stack once per key, as model.py:760-762 does
                                             # (same values; only the CPU-baseline timing uses it)
    pixel_mean: List[float] = field(default_factory=lambda: [103.53, 116.28, 123.675, 127.5, 127.5, 127.5])
    pixel_std: List[float] = field(default_factory=lambda: [1.0] * 6)


BLOCKS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


class FrozenBN(nn.Module):
    """[d2] FrozenBatchNorm2d: y = x*scale + shift with scale = w*rsqrt(var+eps)."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + shift.reshape(1, -1, 1, 1).to(x.dtype)


class NConv(nn.Conv2d):
    """[d2] Conv2d wrapper: conv -> norm (attribute ``norm``) -> activation."""

    def __init__(self, *a, norm=None, relu=False, **kw):
        super().__init__(*a, **kw)
        self.norm = norm
        self.relu = relu

    def forward(self, x):
        x = super().forward(x)
        if self.norm is not None:
            x = self.norm(x)
        return F.relu(x) if self.relu else x


class Stem(nn.Module):
    # resnet.py:24-76
    def __init__(self, cin, cout):
        super().__init__()
        h = cout // 2
        self.conv1 = NConv(cin, h, 3, stride=2, padding=1, bias=False, norm=FrozenBN(h))
        self.conv2 = NConv(h, h, 3, padding=1, bias=False, norm=FrozenBN(h))
        self.conv3 = NConv(h, cout, 3, padding=1, bias=False, norm=FrozenBN(cout))

    def forward(self, x):
        x = F.relu(self.conv1(x))
        x = F.relu(self.conv2(x))
        x = F.relu(self.conv3(x))
        return F.max_pool2d(x, 3, 2, 1)


class Bottleneck(nn.Module):
    # [d2] BottleneckBlock, stride_in_1x1=True, num_groups=1
    def __init__(self, cin, cout, mid, stride, dilation):
        super().__init__()
        self.shortcut = None
        if cin != cout:
            self.shortcut = NConv(cin, cout, 1, stride=stride, bias=False, norm=FrozenBN(cout))
        self.conv1 = NConv(cin, mid, 1, stride=stride, bias=False, norm=FrozenBN(mid))
        self.conv2 = NConv(mid, mid, 3, padding=dilation, dilation=dilation, bias=False, norm=FrozenBN(mid))
        self.conv3 = NConv(mid, cout, 1, bias=False, norm=FrozenBN(cout))

    def forward(self, x):
        out = F.relu(self.conv1(x))
        out = F.relu(self.conv2(out))
        out = self.conv3(out)
        sc = self.shortcut(x) if self.shortcut is not None else x
        return F.relu(out + sc)


def stage_specs(cfg: ArchCfg):
    """(cin, cout, mid, [stride per block], [dilation per block]) for res2..res5 (resnet.py:405-447)."""
    nb = BLOCKS[cfg.depth]
    cin, cout, mid = cfg.stem_out, cfg.res2_out, 64
    out = []
    for idx, stage in enumerate(range(2, 6)):
        dil = cfg.res5_dilation if stage == 5 else 1
        first = 1 if (idx == 0 or dil > 1) else 2
        strides = [first] + [1] * (nb[idx] - 1)
        dils = [dil] * nb[idx] if stage != 5 else [dil * g for g in cfg.res5_multi_grid]
        out.append((cin, cout, mid, strides, dils))
        cin, cout, mid = cout, cout * 2, mid * 2
    return out


class Stream(nn.Module):
    """One ResNet-DeepLab stream; ``prefix`` reproduces the 'depth_' stage names (resnet.py:169)."""

    def __init__(self, cfg: ArchCfg, prefix=""):
        super().__init__()
        self.stem = Stem(6, cfg.stem_out)
        self.names = []
        for i, (cin, cout, mid, strides, dils) in enumerate(stage_specs(cfg)):
            blocks = []
            for s, d in zip(strides, dils):
                blocks.append(Bottleneck(cin, cout, mid, s, d))
                cin = cout
            name = f"{prefix}res{i + 2}"
            self.add_module(name, nn.Sequential(*blocks))
            self.names.append(name)

    def forward(self, x):
        feats = {}
        x = self.stem(x)
        for i, n in enumerate(self.names):
            x = getattr(self, n)(x)
            feats[f"res{i + 2}"] = x
        return feats


class Backbone(nn.Module):
    # resnet.py:453-507
    def __init__(self, cfg: ArchCfg):
        super().__init__()
        self.rgb_backbone = Stream(cfg, "")
        self.depth_backbone = Stream(cfg, "depth_")
        ch = {"res2": 256, "res3": 512, "res5": 2048}
        self.add = cfg.fusion_add
        for k, c in ch.items():
            seq = nn.Sequential()
            if not cfg.fusion_add:                      # resnet.py:472-475: only the concat strategy has the 1x1 reduction
                seq.add_module("conv", nn.Conv2d(2 * c, c, 1))
                seq.add_module("gn", nn.GroupNorm(32, c))
                seq.add_module("relu", nn.ReLU())
            if k != "res5":
                for i in range(cfg.backbone_fusion_layers):
                    seq.add_module(f"conv{i}", nn.Conv2d(c, c, 3, padding=1))
                    seq.add_module(f"gn{i}", nn.GroupNorm(32, c))
                    seq.add_module(f"relu{i}", nn.ReLU())
            self.add_module(f"fusion_{k}", seq)

    def forward(self, x):
        rgb = torch.cat([x[:, :3], x[:, 6:]], 1)
        dep = torch.cat([x[:, 3:6], x[:, 6:]], 1)
        fr, fd = self.rgb_backbone(rgb), self.depth_backbone(dep)
        return {k: getattr(self, f"fusion_{k}")(fr[k] + fd[k] if self.add else torch.cat([fr[k], fd[k]], 1))
                for k in ("res2", "res3", "res5")}


def gn_conv(cin, cout, k, dilation=1):
    pad = dilation if k == 3 else 0
    return NConv(cin, cout, k, padding=pad, dilation=dilation, bias=False, norm=nn.GroupNorm(32, cout), relu=True)


class ASPP(nn.Module):
    # [d2] ASPP with pool_kernel_size=None (train_size None: model.py:612-616)
    def __init__(self, cin, cout, dilations):
        super().__init__()
        self.convs = nn.ModuleList([gn_conv(cin, cout, 1)] + [gn_conv(cin, cout, 3, d) for d in dilations])
        self.convs.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), NConv(cin, cout, 1, bias=True, relu=True)))
        self.project = gn_conv(5 * cout, cout, 1)

    def forward(self, x):
        size = x.shape[-2:]
        res = [c(x) for c in self.convs]
        res[-1] = F.interpolate(res[-1], size=size, mode="bilinear", align_corners=False)
        return self.project(torch.cat(res, 1))  # dropout is a no-op in eval


class PredHead(nn.Module):
    # model.py:369-411
    def __init__(self, cin, ch):
        super().__init__()
        self.head = nn.Sequential(gn_conv(cin, cin, 3), gn_conv(cin, ch, 3))

    def forward(self, x):
        return self.head(x)


class Predictor(nn.Module):
    # model.py:413-422
    def __init__(self, ch, cout):
        super().__init__()
        self.predictor = NConv(ch, cout, 1)

    def forward(self, x):
        return self.predictor(x)


class HeadFusion(nn.Module):
    # model.py:424-458
    def __init__(self, cin, cout, n):
        super().__init__()
        self.fusion_layers = nn.ModuleList(
            [NConv(cin, cout, 1, bias=True, norm=nn.BatchNorm2d(cout), relu=True)]
            + [NConv(cout, cout, 3, padding=1, bias=True, norm=nn.BatchNorm2d(cout), relu=True) for _ in range(n)]
        )

    def forward(self, x):
        for layer in self.fusion_layers:
            x = layer(x)
        return x


class InsEmbedHead(nn.Module):
    def __init__(self, cfg: ArchCfg):
        super().__init__()
        self.cfg = cfg
        dec = nn.ModuleDict()
        dec["res2"] = nn.ModuleDict(
            {"project_conv": gn_conv(256, cfg.project_channels[0], 1),
             "fuse_conv": nn.Sequential(gn_conv(cfg.project_channels[0] + cfg.convs_dim, cfg.convs_dim, 3),
                                        gn_conv(cfg.convs_dim, cfg.convs_dim, 3))})
        dec["res3"] = nn.ModuleDict(
            {"project_conv": gn_conv(512, cfg.project_channels[1], 1),
             "fuse_conv": nn.Sequential(gn_conv(cfg.project_channels[1] + cfg.aspp_channels, cfg.convs_dim, 3),
                                        gn_conv(cfg.convs_dim, cfg.convs_dim, 3))})
        dec["res5"] = nn.ModuleDict({"project_conv": ASPP(2048, cfg.aspp_channels, cfg.aspp_dilations)})
        self.decoder = dec
        d, h = cfg.convs_dim, cfg.head_channels
        self.out_ch = {"foreground": 1, "center": 1, "offset": 2}
        if cfg.eee_mask_on:
            self.out_ch["eee_mask"] = cfg.error_classes
        if cfg.eee_boundary_on:
            self.out_ch["eee_boundary"] = cfg.error_classes
        for name, c in self.out_ch.items():
            self.add_module(f"{name}_pred_head", PredHead(d, h))
            self.add_module(f"{name}_predictor", Predictor(h, c))
        # model.py:576-608: one FusionLayers per hierarchy level >= 1
        if cfg.hierarchical:
            for i in range(1, len(cfg.hierarchy)):
                cin = d
                if "feat" in cfg.fusion_target:
                    cin += h * len(cfg.hierarchy[i - 1])
                if "pred" in cfg.fusion_target:
                    cin += sum(self.out_ch[k] for k in cfg.hierarchy[i - 1])
                self.add_module(f"fusion_layers_{i}", HeadFusion(cin, d, cfg.head_fusion_layers))

    def decode(self, feats):
        y = self.decoder["res5"]["project_conv"](feats["res5"])
        for k in ("res3", "res2"):
            p = self.decoder[k]["project_conv"](feats[k])
            y = F.interpolate(y, size=p.shape[-2:], mode="bilinear", align_corners=False)
            y = self.decoder[k]["fuse_conv"](torch.cat([p, y], 1))
        return y

    def forward(self, feats, taps=None, size=None):
        cfg = self.cfg
        y = self.decode(feats)
        feat, out = {}, {}
        levels = cfg.hierarchy if cfg.hierarchical else [list(self.out_ch)]
        for i, keys in enumerate(levels):
            x = y
            if i > 0:                                             # model.py:749-762
                yp = y
                if "feat" in cfg.fusion_target:
                    for pk in levels[i - 1]:
                        yp = torch.cat([yp, feat[pk]], 1)
                if "pred" in cfg.fusion_target:
                    for pk in levels[i - 1]:
                        o = out[pk]
                        yp = torch.cat([yp, o.softmax(1) if "eee" in pk else o.sigmoid()], 1)
                # the reference evaluates this stack once per key (model.py:760-762); results are identical
                x = getattr(self, f"fusion_layers_{i}")(yp)
                if taps is not None:
                    taps[f"z{i}"] = x
            for j, k in enumerate(keys):
                if i > 0 and j > 0 and cfg.repeat_fusion:
                    x = getattr(self, f"fusion_layers_{i}")(yp)
                feat[k] = getattr(self, f"{k}_pred_head")(x)
                out[k] = getattr(self, f"{k}_predictor")(feat[k])
        if taps is not None:
            taps.update({"y": y})
            taps.update({f"feat_{k}": v for k, v in feat.items()})
        s = cfg.common_stride
        up = {k: F.interpolate(v, scale_factor=s, mode="bilinear", align_corners=False) for k, v in out.items()}
        up["offset"] = up["offset"] * s
        if size is not None:   # [d2] sem_seg_postprocess: crop to the image, then resize to (height, width) = identity
            up = {k: v[:, :, :size[0], :size[1]] for k, v in up.items()}
        return up


class MaskRefinerNet(nn.Module):
    """image u8/float [B,6,H,W] (BGR + 3x depth) and initial_pred_offset f32 [B,3,H,W] -> logits dict."""

    def __init__(self, cfg: ArchCfg = None):
        super().__init__()
        self.cfg = cfg or ArchCfg()
        # build_resnet_deeplab_rgbd_fusion_backbone (resnet.py:510-519) or build_resnet_deeplab_fusion_backbone (:358-449)
        self.backbone = Backbone(self.cfg) if self.cfg.streams == 2 else Stream(self.cfg, "")
        self.ins_embed_head = InsEmbedHead(self.cfg)
        nch = 3 * self.cfg.streams
        self.register_buffer("pixel_mean", torch.tensor(self.cfg.pixel_mean[:nch]).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(self.cfg.pixel_std[:nch]).view(-1, 1, 1), False)

    def forward(self, image, offsets, taps=None):
        x = (image.to(self.pixel_mean.dtype) - self.pixel_mean) / self.pixel_std   # model.py:138
        x = torch.cat([x, offsets.to(x.dtype)], 1)                                # model.py:153
        feats = self.backbone(x)
        feats = {k: feats[k] for k in ("res2", "res3", "res5")}
        if taps is not None:
            taps.update(feats)
        return self.ins_embed_head(feats, taps, size=tuple(image.shape[-2:]))
