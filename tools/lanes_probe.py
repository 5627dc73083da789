import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from quber_amd import _lib, arch, engine, synth
lib = _lib.load()
H, W, N = 480, 640, 20
sd = arch.init_state_dict(seed=0, loud_heads=True, center_bias=-1.68)
DT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LV = 1
for B in (1, 2):
    b = synth.make_batch(7, B, H, W, N)
    qc = engine.make_config(H, W, max_batch=max(B, MB), max_instances=N)
    qc.compute_dtype = DT
    eng = engine.Engine(qc, "cuda:0")
    eng.load_state_dict(sd)
    m, bgr, dep = (torch.from_numpy(b[k]).cuda() for k in ("masks", "rgb", "depth"))
    off = eng.encode(m)
    outs = {}
    for lanes in (0, LV):
        eng.set_option(24, lanes)
        lg = eng.forward(bgr, dep, off)
        torch.cuda.synchronize()
        taps = {n: eng.debug_tensor(n, B).clone() for n in ("res2", "res3", "res5", "y")}
        for _ in range(10): eng.forward(bgr, dep, off)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): eng.forward(bgr, dep, off)
        torch.cuda.synchronize()
        outs[lanes] = (lg.clone(), taps, (time.perf_counter() - t0) / 50 * 1e3)
    outs[1] = outs[LV]
    same = torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(outs[0][1][k], outs[1][1][k]) for k in outs[0][1])
    print("   max |lanes - one stream| on the logits:", float((outs[0][0] - outs[1][0]).abs().max()), {k: float((outs[0][1][k] - outs[1][1][k]).abs().max()) for k in outs[0][1]})
    print(f"batch {B}: forward {outs[0][2]:.3f} ms on one stream, {outs[1][2]:.3f} ms with side lanes; results identical: {same}")
    # repeated runs with lanes stay identical (no races on shared workspaces)
    eng.set_option(24, LV)
    ok = all(torch.equal(eng.forward(bgr, dep, off), outs[1][0]) for _ in range(20))
    print("   20 repeated lane runs identical:", ok)
    eng.close()
