"""ORACLE (test infrastructure, never shipped): functional restatement of the LMFFNet foreground network used by the
reference's post-filter (eval/refiner_model.py:273-277 -> foreground_segmentation/predictor.py:57-99 ->
foreground_segmentation/lmffnet.py:283-341).  Weights are a dict keyed like the reference module's state_dict.

Pinned: the reference module imports here (torch only); tests/golden/lmffnet_*.npz hold its outputs for seeded
weights (oracle/gen_golden.py) and tests/test_oracle_golden.py checks this file against them (<= 1e-5)."""
import numpy as np
import torch
import torch.nn.functional as F

# dilation rates of the two SEM-B stacks (foreground_segmentation/lmffnet.py:299, 305); restated here, not imported
# from the product
SEM1_DIL = (2, 2, 2)
SEM2_DIL = (4, 4, 8, 8, 16, 16, 32, 32)

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def preprocess(bgr_u8, depth_u8):
    """predictor.py:79-83 + eval/preprocess_utils.py:82-96: u8 HWC x2 -> f32 [1,6,H,W] (the BGR image is standardised
    with the RGB statistics, as the reference does)."""
    img = np.zeros(bgr_u8.shape, np.float32)
    for i in range(3):
        img[..., i] = (bgr_u8[..., i] / 255. - MEAN[i]) / STD[i]
    a = torch.from_numpy(img).permute(2, 0, 1).float()[None]
    d = torch.from_numpy(np.ascontiguousarray(depth_u8)).permute(2, 0, 1).float()[None] / 255
    return torch.cat([a, d], 1)


def forward(x, w):
    def bnp(t, n):
        t = F.batch_norm(t, w[n + ".bn.running_mean"], w[n + ".bn.running_var"], w[n + ".bn.weight"], w[n + ".bn.bias"],
                         False, 0.0, 1e-3)
        return F.prelu(t, w[n + ".acti.weight"])

    def conv(t, n, stride=1, pad=0, dil=1, groups=1):
        return F.conv2d(t, w[n + ".conv.weight"], None, stride, pad, dil, groups)

    def cbp(t, n, **kw):
        return bnp(conv(t, n, **kw), n + ".bn_prelu")

    def sem(t, n, d):
        c = t.shape[1]
        o = cbp(t, n + ".conv3x3", pad=1)
        left = cbp(o[:, :c // 4], n + ".dconv_left", pad=1, groups=c // 4)
        right = cbp(o[:, c // 4:], n + ".dconv_right", pad=d, dil=d, groups=c // 4)
        o = cbp(torch.cat([left, right], 1), n + ".conv3x3_resume.conv3x3", pad=1)
        o = conv(o, n + ".conv3x3_resume.conv1x1_resume")
        return bnp(o + t, n + ".bn_relu_1")

    def down(t, n, cout):
        o = conv(t, n + ".conv3x3", stride=2, pad=1)
        if t.shape[1] < cout:
            o = torch.cat([o, F.max_pool2d(t, 2, 2)], 1)
        return bnp(o, n + ".bn_prelu")

    def inject(t, r):
        for _ in range(r):
            t = F.avg_pool2d(t, 3, 2, 1)
        return t

    def pmca(t, n):
        c = t.shape[1]
        o1 = F.conv2d(F.adaptive_avg_pool2d(t, (2, 2)), w[n + ".conv2x2.conv.weight"], None, 1, 0, 1, c)
        s = (o1 + F.adaptive_avg_pool2d(t, 1)).flatten(1)
        s = F.prelu(F.linear(s, w[n + ".SE_Block.fc.0.weight"]), w[n + ".SE_Block.fc.1.weight"])
        s = torch.sigmoid(F.linear(s, w[n + ".SE_Block.fc.2.weight"]))
        return s[:, :, None, None] * t

    init = x
    for i in range(3):
        init = cbp(init, f"Init_Block.init_conv.{i}", stride=2 if i == 0 else 1, pad=1)
    ffa = conv(bnp(torch.cat([init, inject(x, 1)], 1), "FFM_A.bn_prelu"), "FFM_A.conv1x1")
    d1 = down(ffa, "downsample_1", 64)
    s1 = d1
    for i, d in enumerate(SEM1_DIL):
        s1 = sem(s1, f"SEM_B_Block1.SEM_B_Block.SEM_Block_1{i}", d)
    fb1 = conv(bnp(torch.cat([s1, pmca(d1, "FFM_B1.PMCA"), inject(x, 2)], 1), "FFM_B1.bn_prelu"), "FFM_B1.conv1x1")
    d2 = down(fb1, "downsample_2", 128)
    s2 = d2
    for i, d in enumerate(SEM2_DIL):
        s2 = sem(s2, f"SEM_B_Block2.SEM_B_Block.SEM_Block_2{i}", d)
    fb2 = conv(bnp(torch.cat([s2, pmca(d2, "FFM_B2.PMCA"), inject(x, 3)], 1), "FFM_B2.bn_prelu"), "FFM_B2.conv1x1")
    # MAD (lmffnet.py:232-280)
    h, ww = fb2.shape[-2:]
    up = lambda t, f: F.interpolate(t, [h * f, ww * f], mode="bilinear", align_corners=False)
    cat = torch.cat([conv(fb1, "MAD.mid_layer_1x1"), up(conv(fb2, "MAD.deep_layer_1x1"), 2)], 1)
    att = torch.sigmoid(conv(cbp(cat, "MAD.DwConv1", pad=1, groups=48), "MAD.PwConv1"))
    o = up(conv(cbp(fb2, "MAD.DwConv2", pad=1, groups=fb2.shape[1]), "MAD.PwConv2"), 2)
    return up(o * att, 8)


def foreground_mask(logits):
    """predictor.py:85,98: argmax over the 3 classes, foreground = class 2."""
    return (np.argmax(logits, axis=0) == 2)


def overlap_filter(masks, fg, ratio=0.3):
    """eval/refiner_model.py:274-277: keep masks whose foreground overlap exceeds 30 %."""
    return [m for m in masks if np.sum(np.bitwise_and(m, fg)) / np.sum(m) > ratio]
