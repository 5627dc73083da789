#!/usr/bin/env python3
"""End-to-end rate of the evaluation adapter (quber_amd/eval/refiner_model.py:MaskRefiner), file -> refined masks, as
eval/eval_utils.py:277 drives it: predict() frame after frame against predict_stream() (host pre-processing one frame
ahead on a worker thread).  Synthetic 640x480 frames with depth holes (so the TELEA in-painting has work), written to a
temporary directory as PNG.  GPU box only.  usage: tools/adapter_bench.py [frames=40] [instances=20]"""
import os
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import arch, synth  # noqa: E402
from quber_amd.eval.refiner_model import MaskRefiner, inpaint_depth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
with tempfile.TemporaryDirectory() as d:
    items = []
    rng = np.random.default_rng(0)
    for i in range(8):
        sc = synth.make_scene(30 + i, 480, 640, N)
        Image.fromarray(sc["rgb"][:, :, ::-1].copy()).save(os.path.join(d, f"rgb{i}.png"))
        mm = sc["depth"][:, :, 0].astype(np.uint16) * 5 + 300
        for _ in range(12):                                            # ~8 000 zero-depth pixels in a dozen holes
            y, x = int(rng.integers(0, 440)), int(rng.integers(0, 600))
            mm[y:y + 22, x:x + 30] = 0
        Image.fromarray(mm).save(os.path.join(d, f"depth{i}.png"))
        items.append((os.path.join(d, f"rgb{i}.png"), os.path.join(d, f"depth{i}.png"), sc["masks"] != 0, None))
    ref = MaskRefiner(None, None, dataset="OSD")
    ref.refiner_predictor.model.state_dict = arch.init_state_dict(seed=0, loud_heads=True, center_bias=-1.68)
    ref.refiner_predictor.model._engines.clear()
    work = [items[i % 8] for i in range(F)]
    for it in items[:3]:
        ref.predict(*it)
    t0 = time.perf_counter()
    refined = [ref.predict(*it)[2] for it in work]
    seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = list(ref.predict_stream(work))
    stream = time.perf_counter() - t0
    batched = {}
    ncpu = os.cpu_count() or 4
    for wk, bt in ((4, 4), (8, 8), (min(12, ncpu), 16), (min(16, ncpu), 16)):
        work_b = [items[i % 8] for i in range(max(F, 6 * bt))]
        list(ref.predict_stream(work_b[:2 * bt], workers=wk, batch=bt))            # engine of this capacity, worker streams
        t0 = time.perf_counter()
        res = list(ref.predict_stream(work_b, workers=wk, batch=bt))
        dtb = time.perf_counter() - t0
        batched[(wk, bt)] = (len(work_b) / dtb, float(np.mean([len(r[0]) for r in res])))
    t0 = time.perf_counter()
    for it in work[:8]:
        ref._load(*it[:3])
    load = (time.perf_counter() - t0) / 8
    # what the load consists of (one frame)
    import torch
    from quber_amd import engine as qengine
    ph = {}
    t = time.perf_counter(); rgb = np.asarray(Image.open(items[0][0]).convert("RGB"))[:, :, ::-1]; ph["decode rgb png"] = time.perf_counter() - t
    t = time.perf_counter(); dep = np.asarray(Image.open(items[0][1])); ph["decode depth png"] = time.perf_counter() - t
    t = time.perf_counter(); z = np.where(dep == 0); ph["np.where(depth == 0)"] = time.perf_counter() - t
    t = time.perf_counter(); d3 = qengine.normalize_depth(ref._dev(np.array(dep)), 250.0, 1500.0)[0].cpu().numpy(); ph["normalize_depth (device, incl. copies)"] = time.perf_counter() - t
    t = time.perf_counter(); inpaint_depth(d3); ph["inpaint_depth (host TELEA)"] = time.perf_counter() - t
    print({k: round(v * 1e3, 2) for k, v in ph.items()})
    fr = ref._load(*items[0][:3])
    t0 = time.perf_counter()
    inpaint_depth(np.ascontiguousarray(np.where(fr["depth"] > 0, fr["depth"], 0)))
    print(f"{F} frames 640x480, N = {N}: predict() frame after frame {F / seq:.1f} frames/s ({seq / F * 1e3:.2f} ms/frame; reference-timed region "
          f"median {np.median(refined) * 1e3:.2f} ms); predict_stream() {F / stream:.1f} frames/s ({stream / F * 1e3:.2f} ms/frame); "
          f"host pre-processing alone (file decode + resize + normalise + TELEA) {load * 1e3:.2f} ms/frame")
    for (wk, bt), (fps, k) in batched.items():
        print(f"  predict_stream(workers={wk}, batch={bt}): {fps:.1f} frames/s = {fps * N:.0f} refined masks/s file-to-masks ({k:.1f} instances out per frame; {ncpu} host CPUs visible)")
