#!/usr/bin/env python3
"""Times encode + post-processing + mask extraction at the benchmark shape on logits that contain K ~ N real
instances (derived from the initial-mask encoding), i.e. the post-processing load a trained refiner produces.
With the seeded synthetic weights the network's own heads are near-constant (K = 0..1), see DESIGN.md."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import engine, synth  # noqa: E402

B, H, W, N = 16, 480, 640, 20
eng = engine.Engine(engine.make_config(H, W, max_batch=B, max_instances=N, with_network=False), "cuda:0")
batch = synth.make_batch(7, B, H, W, N)
frames = []
enc_all = eng.encode(torch.from_numpy(batch["masks"]).cuda()).cpu().numpy()      # the library's own encoder builds the inputs
for i in range(B):
    enc = enc_all[i]
    lg, ce, of = synth.fake_head_outputs(enc, batch["masks"][i], np.random.default_rng(i), noise=0.4)
    frames.append(np.concatenate([lg, ce, of, np.zeros((4, H, W), np.float32)]))
logits = torch.from_numpy(np.stack(frames)).cuda()
masks = torch.from_numpy(batch["masks"]).cuda()
post = eng.alloc_post(B)
out = torch.empty((B, 32, H, W), dtype=torch.uint8, device="cuda")
offs = torch.empty((B, 3, H, W), device="cuda")


def run():
    eng.encode(masks, offs)
    eng.postprocess(logits, post)
    eng.extract_masks(post, 32, out)


for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print("instances per frame:", post["count"].tolist())
print("encode + postprocess + extract_masks, batch %d, K~%d: %.3f ms per step" % (B, N, e0.elapsed_time(e1) / 20))
