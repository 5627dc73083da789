"""Diagnostic: is a forward of the same inputs on ONE stream repeatable, bit for bit?  Per tap, under launch-time option settings.
usage: python3 tools/repeat_probe.py <dtype> <max_batch> <b> [HxW] [k=v,k=v ...]   (each further argument = one setting to try)"""
import sys
import torch
sys.path.insert(0, ".")
from quber_amd import arch, engine, synth

dtype, maxb, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rest = sys.argv[4:]
h, w = 480, 640
if rest and "x" in rest[0]:
    h, w = (int(v) for v in rest[0].split("x"))
    rest = rest[1:]
n = 12
sd = arch.init_state_dict(seed=3, loud_heads=True, center_bias=-1.6)
batch = synth.make_batch(50 + b, b, h, w, n)
base = None
for setting in [""] + rest:
    qc = engine.make_config(h, w, max_batch=maxb, max_instances=n)
    qc.compute_dtype = dtype
    eng = engine.Engine(qc, "cuda:0")
    eng.set_option(24, 0)
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        eng.set_option(int(k), int(v))
    eng.load_state_dict(sd)
    bgr, dep = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda()
    off = eng.encode(torch.from_numpy(batch["masks"]).cuda())
    names = ("res2", "res3", "res5", "y", "z1")
    ref, reft = None, None
    worst = {k: 0.0 for k in names + ("logits",)}
    frames = set()
    for rep in range(6):
        out = eng.forward(bgr, dep, off).clone()
        taps = {k: eng.debug_tensor(k, b).clone().float() for k in names}
        if ref is None:
            ref, reft = out, taps
            continue
        d = (out - ref).abs()
        worst["logits"] = max(worst["logits"], float(d.max()))
        frames |= {i for i in range(b) if float(d[i].max()) > 0}
        for k in names:
            dk = (taps[k] - reft[k]).abs()
            worst[k] = max(worst[k], float(dk.max()))
    vs = ""
    if base is None:
        base = (ref, reft)
    else:        # against the first setting (the defaults): how far apart two kernel choices are
        vs = ", vs defaults " + str({k: round(float((reft[k] - base[1][k]).abs().max()), 5) for k in names}) + f" logits {float((ref - base[0]).abs().max()):.4f}" \
             + " (tap scale " + str({k: round(float(base[1][k].abs().max()), 2) for k in names}) + ")"
    print(f"dtype {dtype} {h}x{w} max_batch {maxb} b {b} [{setting or 'defaults'}]: max diff between repeats {worst}, frames {sorted(frames)}{vs}", flush=True)
    eng.close()
