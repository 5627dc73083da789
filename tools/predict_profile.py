#!/usr/bin/env python3
"""Where does a drop-in MaskRefinerPredictor.predict() call spend its host time?  Per-call wall times of the fast path, the
general path and the engine alone, with the per-phase split of the fast path (monkey-patched timers).  GPU box only.
usage: tools/predict_profile.py [calls=100]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import arch, synth  # noqa: E402
from quber_amd.maskrefiner.predictor import MaskRefinerPredictor  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
H, W, N, B = 480, 640, 20, 8
host = synth.make_batch(7, B, H, W, N)
sd = arch.init_state_dict(seed=0, loud_heads=True, center_bias=-1.68)
pred = MaskRefinerPredictor(None, device="cuda:0", state_dict=sd)


def call(i):
    t0 = time.perf_counter()
    out = pred.predict(host["rgb"][i % B], host["depth"][i % B], host["masks"][i % B])[0]
    t1 = time.perf_counter()
    m = out["instances"].to("cpu").pred_masks.numpy() if "instances" in out else []
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(m)


for fast in (True, False, True):
    pred.fast_path = fast
    for i in range(8):
        call(i)
    r = np.array([call(i)[:2] for i in range(calls)])
    tot = r.sum(1)
    print(f"fast_path={fast}: total median {np.median(tot):.3f} ms (p10 {np.percentile(tot, 10):.3f}, p90 {np.percentile(tot, 90):.3f}, max {tot.max():.3f}) | "
          f"predict() median {np.median(r[:, 0]):.3f} | to(cpu).numpy() median {np.median(r[:, 1]):.3f}; slow calls (> 2x median): "
          f"{[(i, round(float(t), 1)) for i, t in enumerate(tot) if t > 2 * np.median(tot)][:10]}", flush=True)

# phase split of the fast path
pred.fast_path = True
m = pred.model
eng = m.engine_for(H, W, 1, N)
stg = m.staging_for(eng, N)
ph = {}


def T(name, t0):
    ph.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)


for i in range(calls):
    bgr, depth, masks = host["rgb"][i % B], host["depth"][i % B], host["masks"][i % B]
    hw, n = H * W, masks.shape[0]
    t = time.perf_counter(); stg.done.synchronize(); T("wait prev H2D", t)
    t = time.perf_counter()
    np.copyto(stg.np_in[:3 * hw].reshape(H, W, 3), bgr)
    np.copyto(stg.np_in[3 * hw:6 * hw].reshape(H, W, 3), depth)
    np.copyto(stg.np_in[6 * hw:6 * hw + n * hw].reshape(n, H, W), masks)
    T("numpy -> pinned (%.1f MB)" % ((6 + n) * hw / 1e6), t)
    t = time.perf_counter()
    used = (6 + n) * hw
    stg.dev_in[:used].copy_(stg.pin_in[:used], non_blocking=True); stg.done.record()
    T("issue H2D", t)
    t = time.perf_counter()
    eng.encode(stg.dev_in[6 * hw:used].view(1, n, H, W), stg.offsets)
    lg = eng.forward(stg.dev_in[:3 * hw].view(1, H, W, 3), stg.dev_in[3 * hw:6 * hw].view(1, H, W, 3), stg.offsets)
    post = eng.postprocess(lg, stg.post)
    stg.pin_count.copy_(post["count"], non_blocking=True)
    T("enqueue encode + forward + postprocess", t)
    t = time.perf_counter(); torch.cuda.current_stream().synchronize(); T("sync (GPU finishes the step)", t)
    k = int(stg.pin_count[0])
    t = time.perf_counter()
    mb = eng.extract_masks(post, k)[0].view(torch.bool)
    out = mb.cpu().numpy()
    T("extract + D2H into fresh pageable memory (%.1f MB)" % (k * hw / 1e6), t)
print("fast path phases, median ms (p90):")
for k, v in ph.items():
    print(f"  {k:45s} {np.median(v):7.3f} ({np.percentile(v, 90):.3f})")
