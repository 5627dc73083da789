#!/usr/bin/env python3
"""Tap-level parity of the HIP network against the oracle at the BENCHMARKED configuration (batch 16, 640x480, N = 20,
"loud" predictor set so that logits are O(1) and K ~ N instances come out), per convolution-algorithm mode.
Prints one markdown table per mode: relative error of the seven intermediate taps, absolute error of the logits,
label-map agreement.  GPU box only; writes nothing.  usage: tests/parity_report.py [--batch 16] [--modes auto,off,f2,f4,f6].  Lives under tests/ because it runs the oracle
(test infrastructure: nothing outside tests/, smoke() and the bench's CPU baseline may)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import encode_np, postproc_ref  # noqa: E402
from oracle.network_torch import MaskRefinerNet  # noqa: E402
from quber_amd import _lib, arch, engine, synth  # noqa: E402

TAPS = ("res2", "res3", "res5", "y", "feat_eee_boundary", "z1", "feat_center")
MODES = {"auto": (0, 0), "off": (1, 0), "f2": (0, 2), "f4": (0, 4), "f6": (0, 6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--instances", type=int, default=20)
    ap.add_argument("--modes", default="auto,off,f2,f4,f6")
    ap.add_argument("--fp64", action="store_true", help="also run the oracle in float64 on frame 0 (its own fp32 error)")
    a = ap.parse_args()
    B, H, W, N = a.batch, a.height, a.width, a.instances
    lib = _lib.load()
    batch = synth.make_batch(7, B, H, W, N)
    offs = np.stack([encode_np.encode_initial_masks(m) for m in batch["masks"]])
    image = torch.cat([torch.from_numpy(batch["rgb"]), torch.from_numpy(batch["depth"])], -1).permute(0, 3, 1, 2)

    def oracle(sd, frames, dtype=torch.float32):
        net = MaskRefinerNet().eval()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        net = net.to(dtype)
        taps = {}
        with torch.no_grad():
            out = net(image[frames].to(dtype) if dtype != torch.float32 else image[frames],
                      torch.from_numpy(offs[frames]).to(dtype), taps)
        return out, taps

    sd0 = arch.init_state_dict(seed=0, loud_heads=True)
    out0, _ = oracle(sd0, slice(0, 2))
    bias = arch.calibrate_center_bias(out0["center"], N)
    sd = arch.init_state_dict(seed=0, loud_heads=True, center_bias=bias)
    t0 = time.time()
    ref, rtaps = oracle(sd, slice(0, B))
    print(f"oracle: {B} frames in {time.time() - t0:.1f} s, centre bias {bias:.4f}", flush=True)
    exp = torch.cat([ref["foreground"], ref["center"], ref["offset"], ref["eee_boundary"]], 1)
    if a.fp64:
        r64, t64 = oracle(sd, slice(0, 1), torch.float64)
        e64 = torch.cat([r64["foreground"], r64["center"], r64["offset"], r64["eee_boundary"]], 1)
        print("oracle fp32 vs fp64 (frame 0): logits max abs %.2e; taps rel %s" % (
            float((exp[:1].double() - e64).abs().max()),
            {k: "%.1e" % float((rtaps[k][:1].double() - t64[k]).abs().max() / t64[k].abs().max()) for k in TAPS}), flush=True)
    bgr, dep, off = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda(), torch.from_numpy(offs).cuda()
    for mode in a.modes.split(","):
        k6, k9 = MODES[mode]
        lib.quber_set_tuning(6, k6)
        lib.quber_set_tuning(9, k9)
        eng = engine.Engine(engine.make_config(H, W, max_batch=B, max_instances=N), "cuda:0")
        eng.load_state_dict(sd)
        lg = eng.forward(bgr, dep, off)
        post = eng.postprocess(lg)
        torch.cuda.synchronize()
        lgc = lg.cpu()
        row = {}
        for name in TAPS:
            got = eng.debug_tensor(name, B).cpu().permute(0, 3, 1, 2)
            row[name] = float((got - rtaps[name]).abs().max() / max(1.0, float(rtaps[name].abs().max())))
        d = (lgc - exp).abs()
        planes = {"fg": d[:, 0].max(), "centre": d[:, 1].max(), "offset": d[:, 2:4].max(), "eee": d[:, 4:].max()}
        same, ks = [], []
        for i in range(B):
            o = postproc_ref.postprocess(ref["foreground"][i], ref["center"][i], ref["offset"][i])
            same.append(float((post["panoptic"][i].cpu() == o["panoptic"]).float().mean()))
            ks.append(len(o["labels"]))
        print(f"| {mode} | executed/algorithmic {eng.forward_flops_executed() / eng.forward_flops():.3f} | taps rel: "
              + ", ".join(f"{k} {v:.1e}" for k, v in row.items())
              + " | logits abs: " + ", ".join(f"{k} {float(v):.1e}" for k, v in planes.items())
              + f" | label maps equal {np.mean(same):.6f} (min {np.min(same):.6f}), K mean {np.mean(ks):.1f} |", flush=True)
        eng.close()
        del eng
    lib.quber_set_tuning(6, 0)
    lib.quber_set_tuning(9, 0)


if __name__ == "__main__":
    main()
