#!/usr/bin/env python3
"""Times the LMFFNet foreground network + overlap filter (the reference's post-filter, eval/refiner_model.py:273-277)
at 640x480 on the HIP path."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import lmff_arch, synth  # noqa: E402
from quber_amd.foreground.predictor import LmffEngine  # noqa: E402

for B in (1, 16):
    net = LmffEngine(lmff_arch.init_state_dict(0), 480, 640, B)
    b = synth.make_batch(3, B, 480, 640, 20)
    bgr, dep = torch.from_numpy(b["rgb"]).cuda(), torch.from_numpy(b["depth"]).cuda()
    masks = torch.from_numpy((b["masks"] != 0).astype(np.uint8)).cuda()
    for _ in range(3):
        net.foreground(bgr, dep, masks)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        net.foreground(bgr, dep, masks)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("batch %2d: %.3f ms per step = %.3f ms/frame; %.2f GFLOP/frame (convs), plan of %d launch groups" % (
        B, ms, ms / B, net.eng.forward_flops() / 1e9, len(net.eng.plan())))
    net.eng.close()
