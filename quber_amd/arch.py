"""Parameter inventory of the refiner (detectron2-compatible ``state_dict`` keys) and a
seeded initialiser.  No trained weights ship with the reference (predictor.py:226 hard-codes a path
on the authors' machine), so benchmarks and parity tests run on these synthetic weights; a real
checkpoint's ``state_dict`` (``model`` entry of a detectron2 .pth) loads through the same keys.
"""
from collections import OrderedDict

import numpy as np

BLOCKS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


HEADS = ("foreground", "center", "offset", "eee_mask", "eee_boundary")      # head ids of the C ABI
DEFAULT_HIERARCHY = (("eee_boundary",), ("foreground", "center", "offset"))


def head_out_channels(error_classes=4, eee_mask_on=False, eee_boundary_on=True):
    ch = {"foreground": 1, "center": 1, "offset": 2}
    if eee_mask_on:
        ch["eee_mask"] = error_classes
    if eee_boundary_on:
        ch["eee_boundary"] = error_classes
    return ch


def param_specs(depth=50, backbone_fusion_layers=2, head_fusion_layers=3, error_classes=4, eee_mask_on=False,
                eee_boundary_on=True, hierarchical=True, hierarchy=DEFAULT_HIERARCHY, fusion_target=("feat", "pred"),
                streams=2, fusion_add=False, convs_dim=128, head_channels=32):
    """OrderedDict name -> (shape, kind); kind in conv|bn_w|bn_b|bn_m|bn_v|gn_w|gn_b|bias|pred_w|pred_b.
    convs_dim / head_channels: INS_EMBED_HEAD.CONVS_DIM / HEAD_CHANNELS (model.py:610-651: decoder_channels =
    [CONVS_DIM, CONVS_DIM, ASPP_CHANNELS]; model.py:514-531 head widths)."""
    cd, hc = convs_dim, head_channels
    s = OrderedDict()

    def conv(n, co, ci, k, bias=False):
        s[n + ".weight"] = ((co, ci, k, k), "conv")
        if bias:
            s[n + ".bias"] = ((co,), "bias")

    def bn(n, c):
        for t, kind in (("weight", "bn_w"), ("bias", "bn_b"), ("running_mean", "bn_m"), ("running_var", "bn_v")):
            s[f"{n}.{t}"] = ((c,), kind)

    def gn(n, c):
        s[n + ".weight"] = ((c,), "gn_w")
        s[n + ".bias"] = ((c,), "gn_b")

    stream_list = (("backbone.rgb_backbone.", ""), ("backbone.depth_backbone.", "depth_")) if streams == 2 \
        else (("backbone.", ""),)
    for stream, pre in stream_list:
        for i, (co, ci) in enumerate(((32, 6), (32, 32), (64, 32))):
            conv(f"{stream}stem.conv{i + 1}", co, ci, 3)
            bn(f"{stream}stem.conv{i + 1}.norm", co)
        cin, cout, mid = 64, 256, 64
        for st, nb in enumerate(BLOCKS[depth]):
            for b in range(nb):
                p = f"{stream}{pre}res{st + 2}.{b}."
                if cin != cout:
                    conv(p + "shortcut", cout, cin, 1)
                    bn(p + "shortcut.norm", cout)
                conv(p + "conv1", mid, cin, 1)
                bn(p + "conv1.norm", mid)
                conv(p + "conv2", mid, mid, 3)
                bn(p + "conv2.norm", mid)
                conv(p + "conv3", cout, mid, 1)
                bn(p + "conv3.norm", cout)
                cin = cout
            cout, mid = cout * 2, mid * 2
    for k, c in ((("res2", 256), ("res3", 512), ("res5", 2048)) if streams == 2 else ()):
        p = f"backbone.fusion_{k}."
        if not fusion_add:
            conv(p + "conv", c, 2 * c, 1, bias=True)
            gn(p + "gn", c)
        if k != "res5":
            for i in range(backbone_fusion_layers):
                conv(p + f"conv{i}", c, c, 3, bias=True)
                gn(p + f"gn{i}", c)
    h = "ins_embed_head."
    a = h + "decoder.res5.project_conv."
    conv(a + "convs.0", 256, 2048, 1)
    gn(a + "convs.0.norm", 256)
    for i in (1, 2, 3):
        conv(a + f"convs.{i}", 256, 2048, 3)
        gn(a + f"convs.{i}.norm", 256)
    conv(a + "convs.4.1", 256, 2048, 1, bias=True)
    conv(a + "project", 256, 1280, 1)
    gn(a + "project.norm", 256)
    for k, cin_f, pc, up in (("res3", 512, 64, 256), ("res2", 256, 32, cd)):
        p = h + f"decoder.{k}."
        conv(p + "project_conv", pc, cin_f, 1)
        gn(p + "project_conv.norm", pc)
        conv(p + "fuse_conv.0", cd, pc + up, 3)
        gn(p + "fuse_conv.0.norm", cd)
        conv(p + "fuse_conv.1", cd, cd, 3)
        gn(p + "fuse_conv.1.norm", cd)
    out_ch = head_out_channels(error_classes, eee_mask_on, eee_boundary_on)
    for name, co in out_ch.items():
        p = h + f"{name}_pred_head.head."
        conv(p + "0", cd, cd, 3)
        gn(p + "0.norm", cd)
        conv(p + "1", hc, cd, 3)
        gn(p + "1.norm", hc)
        s[h + f"{name}_predictor.predictor.weight"] = ((co, hc, 1, 1), "pred_w")
        s[h + f"{name}_predictor.predictor.bias"] = ((co,), "pred_b")
    if hierarchical:                                     # model.py:576-608
        for i in range(1, len(hierarchy)):
            cin = cd
            if "feat" in fusion_target:
                cin += hc * len(hierarchy[i - 1])
            if "pred" in fusion_target:
                cin += sum(out_ch[k] for k in hierarchy[i - 1])
            f = h + f"fusion_layers_{i}.fusion_layers."
            conv(f + "0", cd, cin, 1, bias=True)
            bn(f + "0.norm", cd)
            for j in range(head_fusion_layers):
                conv(f + f"{j + 1}", cd, cd, 3, bias=True)
                bn(f + f"{j + 1}.norm", cd)
    return s


LOUD_PRED_SIGMA = 0.25


def init_state_dict(seed=0, loud_heads=False, center_bias=0.0, **kw):
    """Seeded float32 numpy state_dict.  Convs: He-normal (fan_out) for the encoder as c2_msra_fill
    (resnet.py:64-66), Xavier-uniform for the heads as c2_xavier_fill (model.py:405-406), predictors
    N(0, 0.001) (model.py:418-419).  Norm statistics are random but benign so that folding is exercised;
    the last BN of every bottleneck is damped to keep the residual trunk O(1) without training.

    loud_heads: the faithful N(0, 0.001) predictors give logits of ~3e-3 (a 1e-4 check on them proves little, and
    no centre ever reaches the 0.3 threshold: K = 0).  The "loud" set draws the final 1x1 predictors from
    N(0, LOUD_PRED_SIGMA) instead, so every head output is O(1) and sensitive to the features, and shifts the centre
    head by `center_bias` (see calibrate_center_bias) so that post-processing sees K ~ N instances per frame."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, (shape, kind) in param_specs(**kw).items():
        if kind == "conv":
            co, ci, k, _ = shape
            if name.startswith("backbone.") and not name.startswith("backbone.fusion_"):
                v = rng.normal(0, np.sqrt(2.0 / (co * k * k)), shape)
            else:
                lim = np.sqrt(3.0) * np.sqrt(2.0 / ((ci + co) * k * k))
                v = rng.uniform(-lim, lim, shape)
        elif kind == "bn_w":
            v = rng.uniform(0.5, 1.5, shape)
            if ".conv3.norm" in name:
                v = v * 0.3
        elif kind in ("bn_b", "bn_m"):
            v = rng.normal(0, 0.1, shape)
        elif kind == "bn_v":
            v = rng.uniform(0.5, 1.5, shape)
        elif kind == "gn_w":
            v = rng.uniform(0.8, 1.2, shape)
        elif kind in ("gn_b", "bias"):
            v = rng.normal(0, 0.05, shape)
        elif kind == "pred_w":
            v = rng.normal(0, LOUD_PRED_SIGMA if loud_heads else 0.001, shape)
        elif kind == "pred_b":
            v = np.zeros(shape)
            if loud_heads and "center_predictor" in name:
                v = v + center_bias
        else:
            raise AssertionError(kind)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def calibrate_center_bias(center, target=20, threshold=0.3, nms_kernel=7):
    """center: float tensor [B,1,H,W] or [B,H,W] of centre-head outputs computed with center_bias = 0.
    Returns the bias that makes `target` local maxima per frame (on average) exceed `threshold` after the
    7x7 max-pool NMS of post_processing.py:9-41 (a constant shift does not move the maxima)."""
    import torch
    import torch.nn.functional as F
    c = torch.as_tensor(center).float()
    if c.dim() == 3:
        c = c[:, None]
    pooled = F.max_pool2d(c, nms_kernel, 1, nms_kernel // 2)
    cuts = []
    for b in range(c.shape[0]):
        # distinct peak VALUES: the x4 bilinear up-sampling of the quarter-resolution head output makes every maximum a plateau of
        # 2-4 equal pixels - cutting between two pixels of one plateau would put the threshold exactly ON a peak
        peaks = torch.unique(c[b][c[b] == pooled[b]], sorted=True).flip(0)
        k = min(target, peaks.numel() - 1)
        cuts.append(float(0.5 * (peaks[k - 1] + peaks[k])) if k >= 1 else float(peaks[0]) - 1.0)
    return float(threshold - np.median(cuts))


def num_parameters(specs=None):
    specs = specs or param_specs()
    return int(sum(np.prod(sh) for sh, _ in specs.values()))
