"""Diagnostic: side lanes against one stream, per batch and tap (bit-equality expected).  usage: python3 tools/lanes_diff_probe.py <dtype> <max_batch> b1 b2 ..."""
import sys
import torch
sys.path.insert(0, ".")
from quber_amd import arch, engine, synth

dtype, maxb = int(sys.argv[1]), int(sys.argv[2])
bs = [int(x) for x in sys.argv[3:]]
h, w, n = 480, 640, 12
sd = arch.init_state_dict(seed=3, loud_heads=True, center_bias=-1.6)
qc = engine.make_config(h, w, max_batch=maxb, max_instances=n)
qc.compute_dtype = dtype
eng = engine.Engine(qc, "cuda:0")
eng.load_state_dict(sd)
for b in bs:
    batch = synth.make_batch(50 + b, b, h, w, n)
    bgr, dep = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda()
    off = eng.encode(torch.from_numpy(batch["masks"]).cuda())
    eng.set_option(24, 0)
    one = eng.forward(bgr, dep, off).clone()
    one2 = eng.forward(bgr, dep, off).clone()
    taps = {k: eng.debug_tensor(k, b).clone().float() for k in ("res2", "res3", "res5", "y")}
    eng.set_option(24, 1)
    for rep in range(4):
        out = eng.forward(bgr, dep, off)
        d = (out - one).abs()
        td = {k: float((eng.debug_tensor(k, b).float() - v).abs().max()) for k, v in taps.items()}
        bad_frames = [i for i in range(b) if float(d[i].max()) > 0]
        print(f"dtype {dtype} max_batch {maxb} b {b} rep {rep}: one-stream repeat equal {bool(torch.equal(one, one2))}, logits max diff {float(d.max()):.3e}, frames differing {bad_frames}, taps {td}", flush=True)
eng.close()
