"""Diagnostic: the frames of a batch of b against the same frames refined one by one (another engine, max_batch 1): per-frame semantics
(GroupNorm per image, no cross-frame op), so the logits may differ by the re-association of fp32 sums only (split-K and tile choices depend
on the batch).  A tile scaled, skipped or computed twice by a launch structure that only some batch sizes produce shows up here.
usage: python3 tools/batch_invariance_probe.py <dtype> [HxW] b1 b2 ..."""
import sys
import torch
sys.path.insert(0, ".")
from quber_amd import arch, engine, synth

dtype = int(sys.argv[1])
rest = sys.argv[2:]
h, w = 480, 640
if rest and "x" in rest[0]:
    h, w = (int(v) for v in rest[0].split("x"))
    rest = rest[1:]
bs = [int(v) for v in rest]
n = 12
sd = arch.init_state_dict(seed=3, loud_heads=True, center_bias=-1.6)


def make(maxb):
    qc = engine.make_config(h, w, max_batch=maxb, max_instances=n)
    qc.compute_dtype = dtype
    e = engine.Engine(qc, "cuda:0")
    e.load_state_dict(sd)
    return e


one = make(1)
batch = synth.make_batch(90, max(bs), h, w, n)
bgr, dep, masks = (torch.from_numpy(batch[k]).cuda() for k in ("rgb", "depth", "masks"))
single = torch.cat([one.forward(bgr[i:i + 1], dep[i:i + 1], one.encode(masks[i:i + 1])).clone() for i in range(max(bs))])
scale = float(single.abs().max())
for b in bs:
    e = make(b)
    out = e.forward(bgr[:b], dep[:b], e.encode(masks[:b]))
    d = (out - single[:b]).abs().amax((1, 2, 3))
    print(f"dtype {dtype} {h}x{w} batch {b}: max |batch - one by one| per frame {[round(float(v), 6) for v in d]} (logit scale {scale:.2f})", flush=True)
    e.close()
one.close()
