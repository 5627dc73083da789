#!/usr/bin/env python3
"""Per-layer view of the convolution kernels.

  run    : build the engine, run a few forwards at the benchmark shape and write the launch plan to JSON
           (meant to be executed under `rocprofv3 --kernel-trace --output-format csv`)
  report : join the plan with rocprofv3's kernel_trace.csv -> per-layer time / TFLOP/s table (markdown)
"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(a):
    import torch
    from quber_amd import _lib, arch, engine, synth
    for kv in (a.tuning.split(",") if a.tuning else []):
        _lib.load().quber_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    qc = engine.make_config(a.height, a.width, max_batch=a.batch)
    qc.compute_dtype = a.compute_dtype
    eng = engine.Engine(qc, "cuda:0")
    eng.load_state_dict(arch.init_state_dict(seed=0, loud_heads=True))
    b = synth.make_batch(7, a.batch, a.height, a.width, 20)
    masks, bgr, depth = (torch.from_numpy(b[k]).cuda() for k in ("masks", "rgb", "depth"))
    offs = eng.encode(masks)
    for _ in range(a.iters):
        eng.forward(bgr, depth, offs)
    torch.cuda.synchronize()
    json.dump({"batch": a.batch, "iters": a.iters, "compute_dtype": a.compute_dtype, "plan": eng.plan()}, open(a.plan, "w"))


def report(a):
    """One row per convolution op: its conv_igemm launch plus the launches that belong to it (Winograd input / output
    transforms, split-K reduce), taken from the last forward in the trace."""
    meta = json.load(open(a.plan))
    plan, B = meta["plan"], meta["batch"]
    convs = [p for p in plan if p[1] == "conv"]
    allk = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    name = lambda r: r["Kernel_Name"]
    # one group of launches per convolution op: [wino_input] conv_igemm+ (pk_fixup | splitk_reduce)* [wino_output] | wino_fused; a Winograd op
    # whose groups share one input transform (the heads of a level) has several GEMM launches between its two transforms
    groups, i = [], 0
    while i < len(allk):
        n_ = name(allk[i])
        if "wino_input" in n_:
            j = i + 1
            while j < len(allk) and "wino_output" not in name(allk[j]):
                j += 1
            groups.append(("winograd", allk[i:j + 1]))
            i = j + 1
        elif "wino_fused" in n_:                  # a Winograd layer as ONE kernel (wino_fused.hip)
            groups.append(("winograd single-kernel", [allk[i]]))
            i += 1
        elif "stem_conv1" in n_:                  # a3 + stem.conv1 as one kernel (stem.hip)
            groups.append(("fused with a3", [allk[i]]))
            i += 1
        elif any(t in n_ for t in ("conv_h8", "conv_x8")):     # the LDS-DMA pipeline kernels (conv_h8.hip fp16, conv_x8.hip bf16x3)
            groups.append(("direct " + ("h8" if "conv_h8" in n_ else "x8"), [allk[i]]))
            i += 1
        elif "conv_igemm" in n_:
            j = i + 1
            while j < len(allk) and any(t in name(allk[j]) for t in ("splitk_reduce", "pk_fixup")):
                j += 1
            groups.append(("direct", allk[i:j]))
            i = j
        else:
            i += 1
    n = len(convs)
    # the last forward only; a "conv3 + shortcut" op whose dual-input launch does not cover a small batch runs as two direct launches of the same
    # grid (shortcut, then conv3 with the residual): both belong to the op's row
    first = max(i for i, (_, ks) in enumerate(groups) if "stem_conv1" in name(ks[0]) or (i == 0))
    last_fw = groups[first:] if "stem_conv1" in name(groups[first][1][0]) else groups[-n:]
    extra = len(last_fw) - n
    if extra > 0:
        merged, i = [], 0
        for c in convs:
            path, ks = last_fw[i]
            i += 1
            if extra > 0 and "+ shortcut" in c[0] and path == "direct" and i < len(last_fw) and last_fw[i][0] == "direct" and \
                    all(ks[0][f] == last_fw[i][1][0][f] for f in ("Grid_Size_X", "Grid_Size_Z")):
                ks = ks + last_fw[i][1]
                path = "direct (shortcut and conv3 as two launches)"
                i += 1
                extra -= 1
            merged.append((path, ks))
        assert i == len(last_fw) and extra == 0, (len(last_fw), n, extra)
        groups = merged
    assert len(groups) % n == 0 and len(groups) >= n, (len(groups), n)
    tot_t = tot_f = 0.0
    out = ["| # | layer (first weight key) | path, GEMM tile | GFLOP (batch %d, algorithmic) | ms | TFLOP/s |" % B,
           "|---|---|---|---|---|---|"]
    for i, (c, (path, ks)) in enumerate(zip(convs, groups[-n:])):
        ms = sum(dur(r) for r in ks)
        r = next(k for k in ks if any(t in name(k) for t in ("conv_igemm", "conv_h8", "conv_x8", "wino_fused", "stem_conv1")))
        fl = c[2] * B
        if "stem_conv1" in name(r):
            tile = ("matrix pipe" if "h16" in name(r) else "vector FMA") + ", 8 x 32 pixels x 32 ch per block"
        elif any(t in name(r) for t in ("conv_h8", "conv_x8")):
            nm, targs = name(r), name(r).split("<")[1].split(">")[0].replace(" ", "")
            kind = ("DMA gather 256x256" if "conv_h8_kernel" in nm else "LDS patch 8x32 px x 256 ch" if "conv_h8w_kernel" in nm else
                    "LDS patch 8x32 px x %d ch" % (128 if targs.startswith("4") else 32 if targs.count(",") == 4 and targs.endswith("true") else 64) if "conv_h8p_kernel" in nm else
                    "LDS-resident filters, patch 8x32 px x %s ch" % targs if "conv_h8s_kernel" in nm else "DMA gather 256x128")
            tile = kind + " persistent, " + targs
        elif "wino_fused" in name(r):
            tile = "16 tiles x 64 ch" if "fused64" in name(r) else "32 tiles x 32 ch"
        else:
            tile = ("persistent " if "conv_igemm_pk" in name(r) else "") + name(r).split("<")[1].split(">")[0].replace(" ", "")
        out.append("| %d | %s | %s %s | %.1f | %.3f | %.1f |" % (i, c[0].replace("backbone.", "b.").replace("ins_embed_head.", "h."),
                                                               path, tile, fl / 1e9, ms, fl / ms / 1e9))
        tot_t += ms
        tot_f += fl
    out.append("| | **all convolutions** | | %.1f | %.3f | %.1f |" % (tot_f / 1e9, tot_t, tot_f / tot_t / 1e9))
    print("\n".join(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["run", "report"])
    ap.add_argument("--plan", default="gpurun_out/plan.json")
    ap.add_argument("--trace")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--tuning", default="", help="quber_set_tuning knobs, e.g. 13=1")
    ap.add_argument("--compute-dtype", type=int, default=0, help="quber_config.compute_dtype (0 fp32 MFMA, 3 bf16x3, 2 fp16)")
    a = ap.parse_args()
    run(a) if a.mode == "run" else report(a)
