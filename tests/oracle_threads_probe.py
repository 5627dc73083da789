#!/usr/bin/env python3
"""How many host threads serve the oracle best on this box?  One 640x480 frame, batch 1 (as the reference runs), float32 and
float64, per torch.set_num_threads value.  Test infrastructure (imports oracle/); writes nothing.
usage: tests/oracle_threads_probe.py [threads,comma,separated]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import encode_np, postproc_ref  # noqa: E402
from quber_amd import arch, synth  # noqa: E402
from oracle import fp64_anchor as fa  # noqa: E402

H, W, N = 480, 640, 20
threads = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 16, 32, 64, 128]
b = synth.make_batch(7, 1, H, W, N)
offs = torch.from_numpy(np.stack([encode_np.encode_initial_masks(m) for m in b["masks"]]))
image = torch.cat([torch.from_numpy(b["rgb"]), torch.from_numpy(b["depth"])], -1).permute(0, 3, 1, 2)
sd = arch.init_state_dict(seed=0, loud_heads=True, center_bias=-1.68)
n32, n64 = fa.build_net(sd), fa.build_net(sd, torch.float64)
print(f"usable cpus {len(os.sched_getaffinity(0))}, torch default threads {torch.get_num_threads()}", flush=True)
for t in threads:
    torch.set_num_threads(t)
    row = []
    for net, x, o in ((n32, image, offs), (n64, image.double(), offs.double())):
        with torch.no_grad():
            net(x, o)
            t0 = time.perf_counter()
            out = net(x, o)
            row.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    postproc_ref.postprocess(out["foreground"][0].float(), out["center"][0].float(), out["offset"][0].float())
    row.append(time.perf_counter() - t0)
    print(f"threads {t:4d}: fp32 {row[0]:.2f} s, fp64 {row[1]:.2f} s, post-processing {row[2]:.2f} s", flush=True)
