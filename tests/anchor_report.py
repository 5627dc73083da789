#!/usr/bin/env python3
"""Float64-anchor table (oracle/fp64_anchor.py) of the HIP network per tuning configuration: where does the distance to
float64 come from (accumulation order, Winograd transforms), and what does a knob buy?  GPU box only; writes nothing.
usage: tests/anchor_report.py [--frames 4] [--dtype 0] --configs "default;13=0;13=0,3=8;6=1,13=0,3=8"
Lives under tests/ because it runs the oracle (test infrastructure)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import encode_np  # noqa: E402
from quber_amd import _lib, arch, engine, synth  # noqa: E402
from oracle import fp64_anchor as fa  # noqa: E402
from tests.conftest import oracle_threads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--instances", type=int, default=20)
    ap.add_argument("--dtype", type=int, default=0)
    ap.add_argument("--configs", default="default")
    a = ap.parse_args()
    torch.set_num_threads(oracle_threads())
    B, H, W, N = a.frames, a.height, a.width, a.instances
    lib = _lib.load()
    batch = synth.make_batch(7, B, H, W, N)
    offs = np.stack([encode_np.encode_initial_masks(m) for m in batch["masks"]])
    image = torch.cat([torch.from_numpy(batch["rgb"]), torch.from_numpy(batch["depth"])], -1).permute(0, 3, 1, 2)
    sd = arch.init_state_dict(seed=0, loud_heads=True, center_bias=-1.68)
    bgr, dep, off = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda(), torch.from_numpy(offs).cuda()
    cands = {}
    for cfg in a.configs.split(";"):
        kv = [tuple(int(x) for x in s.split("=")) for s in cfg.split(",") if "=" in s]
        for k, v in kv:
            lib.quber_set_tuning(k, v)
        qc = engine.make_config(H, W, max_batch=B, max_instances=N)
        qc.compute_dtype = a.dtype
        eng = engine.Engine(qc, "cuda:0")
        eng.load_state_dict(sd)
        lg = eng.forward(bgr, dep, off)
        torch.cuda.synchronize()
        cands[cfg] = {"lg": lg.cpu(), "taps": {n: eng.debug_tensor(n, B).cpu().permute(0, 3, 1, 2) for n in fa.TAPS},
                      "anchor": fa.AnchorErrors()}
        eng.close()
        for k, v in kv:
            lib.quber_set_tuning(k, {6: 0, 13: 1, 3: 0, 9: 0, 20: 0, 21: 2, 25: 1, 27: 160}.get(k, 0))
    for fr in fa.OracleStream(sd, image, offs):
        i = fr["i"]
        l32, l64 = fa.cat_heads(fr["out32"]), fa.cat_heads(fr["out64"])
        for c in cands.values():
            for n in fa.TAPS:
                t64 = fr["taps64"][n]
                c["anchor"].add(n, c["taps"][n][i:i + 1], fr["taps32"][n], t64, scale=max(1.0, float(t64.abs().max())))
            c["anchor"].add_heads(c["lg"][i:i + 1], l32, l64)
    for cfg, c in cands.items():
        ok, bad = c["anchor"].verdict()
        print(f"\n### tuning {cfg} (dtype {a.dtype}, {B} frames): {'PASS' if ok else 'FAIL'} at ratio {fa.RATIO}\n{c['anchor'].table()}", flush=True)


if __name__ == "__main__":
    main()
