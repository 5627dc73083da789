#!/usr/bin/env python3
"""Diagnostic (timing-only) build of the 128x128 convolution kernel: per-wave cycle shares of one K-slice iteration
(s_memtime stamps, cdna_hip_programming.md section 7).  Read the SHARES, not the absolute time."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib
from tools.conv_bench import LAYERS
lib = _lib.load()
lib.quber_set_debug_buffer.argtypes = [C.c_void_p]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
names = ["issue next slice's loads", "ds_read + 64 MFMA", "barrier 1 (wait for the block)", "vmcnt wait + ds_write", "barrier 2"]
for (name, B, H, W, Cin, Cout, k, s, d, res) in LAYERS:
    if Cout < 128 or (B * H * W // 128) * (Cout // 128) < 512:
        continue
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
    y = torch.empty(B, H, W, Cout, device="cuda"); packed = torch.empty(Cout * ((k * k * Cin + 31) // 32 * 32), device="cuda")
    tiles = ((B * H * W + 127) // 128) * ((Cout + 127) // 128)
    dbg = torch.zeros(tiles * 4 * 12, dtype=torch.int64, device="cuda")
    tms = {}
    for mode in ("prod", "diag"):
        ts = []
        for it in range(5):
            lib.quber_set_debug_buffer(p(dbg) if mode == "diag" else None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, s, d * (k // 2), d, None, None, None, 0, p(packed), p(y), st))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        tms[mode] = min(ts)
    lib.quber_set_debug_buffer(None)
    fl = 2.0 * B * H * W * Cin * k * k * Cout
    print("   kernel time: production %.3f ms (%.1f TF/s), stamped build %.3f ms (%.1f TF/s)" % (tms["prod"], fl / tms["prod"] / 1e9, tms["diag"], fl / tms["diag"] / 1e9))
    raw = dbg.cpu().numpy().reshape(-1, 12).astype(np.float64)
    v = raw[:, :5]
    ghz = np.median(raw[:, 5] / np.maximum(raw[:, 6], 1)) * 0.1
    print("   in-kernel shader clock (d s_memtime / d s_memrealtime x 100 MHz), median over waves: %.2f GHz" % ghz)
    nk = (k * k * Cin + 31) // 32
    tot = v.sum(1) / nk
    print("   per-wave cycles per K-slice: p10 %.0f  p50 %.0f  p90 %.0f  max %.0f  mean %.0f" % (np.percentile(tot, 10), np.percentile(tot, 50), np.percentile(tot, 90), tot.max(), tot.mean()))
    loop_ms = raw[:, 6] / 100e3
    print("   K-loop wall time per block (ms): p10 %.3f p50 %.3f p90 %.3f max %.3f; blocks %d -> %.2f per (CU x 3 slots)" % (np.percentile(loop_ms, 10), np.percentile(loop_ms, 50), np.percentile(loop_ms, 90), loop_ms.max(), v.shape[0] // 4, v.shape[0] / 4 / 768))
    # ---- residency timeline per CU (HW_ID: cu [11:8], sh [12], se [15:13]; XCC_ID low bits) ----
    ids = dbg.cpu().numpy().reshape(-1, 12)
    w0 = ids[::4]                                   # wave 0 of every block
    hw, xcc = (w0[:, 10] >> 32) & 0xffffffff, w0[:, 10] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    bst, ks, en = w0[:, 7].astype(np.float64), w0[:, 8].astype(np.float64), w0[:, 9].astype(np.float64)
    t0k, t1k = bst.min(), en.max()
    print("   kernel span %.3f ms; distinct CUs seen %d; per block: prologue %.1f us, K loop %.1f us, epilogue %.1f us (medians)" % (
        (t1k - t0k) / 100e3, len(np.unique(cu)), np.median(ks - bst) / 100, np.median(raw[::4, 6]) / 100, np.median(en - ks - raw[::4, 6]) / 100))
    occ, gaps = [], []
    for c in np.unique(cu):
        m = cu == c
        ev = sorted([(a, 1) for a in bst[m]] + [(b, -1) for b in en[m]])
        cur, last, area = 0, t0k, 0.0
        for tm, dlt in ev:
            area += cur * (tm - last); last = tm; cur += dlt
        occ.append(area / (t1k - t0k))
        gaps.append(m.sum())
    print("   mean resident blocks per CU over the kernel span: %.2f (min %.2f max %.2f); blocks per CU: min %d max %d" % (np.mean(occ), np.min(occ), np.max(occ), min(gaps), max(gaps)))
    med = np.median(v, 0) / nk
    print(f"{name}: K-slices {nk}; cycles per K-slice per wave (median over {v.shape[0]} waves): total {med.sum():.0f}")
    for n, c in zip(names, med):
        print(f"    {n:34s} {c:8.0f}  {100 * c / med.sum():5.1f} %")
