"""Parameter inventory + seeded initialiser of the LMFFNet foreground net (reference
foreground_segmentation/lmffnet.py:283-341; keys = that module's ``state_dict()`` keys, which is what the reference's
``rgbd_lmffnet.pth`` checkpoint holds under 'model', foreground_segmentation/predictor.py:60-62)."""
from collections import OrderedDict

import numpy as np

SEM1_DIL = (2, 2, 2)
SEM2_DIL = (4, 4, 8, 8, 16, 16, 32, 32)


def param_specs(classes=3):
    s = OrderedDict()

    def conv(n, co, ci, k):
        s[n + ".conv.weight"] = ((co, ci, k, k), "conv")

    def bnp(n, c):
        for t, kind in (("bn.weight", "bn_w"), ("bn.bias", "bn_b"), ("bn.running_mean", "bn_m"), ("bn.running_var", "bn_v"),
                        ("acti.weight", "prelu")):
            s[f"{n}.{t}"] = ((c,), kind)

    def cbp(n, co, ci, k):
        conv(n, co, ci, k)
        bnp(n + ".bn_prelu", co)

    for i, ci in enumerate((6, 32, 32)):
        cbp(f"Init_Block.init_conv.{i}", 32, ci, 3)
    bnp("FFM_A.bn_prelu", 38)
    conv("FFM_A.conv1x1", 38, 38, 1)
    conv("downsample_1.conv3x3", 26, 38, 3)
    bnp("downsample_1.bn_prelu", 64)

    def sem(prefix, c):
        cbp(prefix + ".conv3x3", c // 2, c, 3)
        for side in ("dconv_left", "dconv_right"):
            cbp(f"{prefix}.{side}", c // 4, 1, 3)
        bnp(prefix + ".bn_relu_1", c)
        cbp(prefix + ".conv3x3_resume.conv3x3", c // 2, c // 2, 3)
        conv(prefix + ".conv3x3_resume.conv1x1_resume", c, c // 2, 1)

    for i in range(len(SEM1_DIL)):
        sem(f"SEM_B_Block1.SEM_B_Block.SEM_Block_1{i}", 64)

    def ffm_b(n, cin, cp):
        conv(n + ".PMCA.conv2x2", cp, 1, 2)
        s[n + ".PMCA.SE_Block.fc.0.weight"] = ((cp // 8, cp), "fc")
        s[n + ".PMCA.SE_Block.fc.1.weight"] = ((1,), "prelu")
        s[n + ".PMCA.SE_Block.fc.2.weight"] = ((cp, cp // 8), "fc")
        bnp(n + ".bn_prelu", cin)
        conv(n + ".conv1x1", cin, cin, 1)

    ffm_b("FFM_B1", 134, 64)
    conv("downsample_2.conv3x3", 128, 134, 3)
    bnp("downsample_2.bn_prelu", 128)
    for i in range(len(SEM2_DIL)):
        sem(f"SEM_B_Block2.SEM_B_Block.SEM_Block_2{i}", 128)
    ffm_b("FFM_B2", 262, 128)
    conv("MAD.mid_layer_1x1", 16, 134, 1)
    conv("MAD.deep_layer_1x1", 32, 262, 1)
    cbp("MAD.DwConv1", 48, 1, 3)
    conv("MAD.PwConv1", classes, 48, 1)
    cbp("MAD.DwConv2", 262, 1, 3)
    conv("MAD.PwConv2", classes, 262, 1)
    return s


def init_state_dict(seed=0, classes=3):
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, (shape, kind) in param_specs(classes).items():
        if kind == "conv":
            fan_in = shape[1] * shape[2] * shape[3]
            v = rng.normal(0, np.sqrt(2.0 / fan_in) * 0.8, shape)
        elif kind == "fc":
            v = rng.normal(0, np.sqrt(1.0 / shape[1]), shape)
        elif kind == "bn_w":
            v = rng.uniform(0.6, 1.4, shape)
        elif kind in ("bn_b", "bn_m"):
            v = rng.normal(0, 0.1, shape)
        elif kind == "bn_v":
            v = rng.uniform(0.5, 1.5, shape)
        elif kind == "prelu":
            v = rng.uniform(0.1, 0.4, shape)
        else:
            raise AssertionError(kind)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out
