#!/usr/bin/env python3
"""Does a hipGraph replay of encode + forward reproduce the eager results on a new frame?  (found the hipMemsetAsync issue)"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from quber_amd import arch, engine, synth
h,w,n=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
sd=arch.init_state_dict(seed=4)
eng=engine.Engine(engine.make_config(h,w,max_batch=1,max_instances=n),"cuda:0"); eng.load_state_dict(sd)
dev="cuda:0"
masks=torch.empty((1,n,h,w),dtype=torch.uint8,device=dev); bgr=torch.empty((1,h,w,3),dtype=torch.uint8,device=dev); depth=torch.empty((1,h,w,3),dtype=torch.uint8,device=dev)
offsets=torch.empty((1,3,h,w),dtype=torch.float32,device=dev); logits=torch.empty((1,eng.planes,h,w),dtype=torch.float32,device=dev)
def load(seed):
    sc=synth.make_scene(seed,h,w,n); masks.copy_(torch.from_numpy(sc["masks"][None])); bgr.copy_(torch.from_numpy(sc["rgb"][None])); depth.copy_(torch.from_numpy(sc["depth"][None]))
def step():
    eng.encode(masks,offsets); eng.forward(bgr,depth,offsets,logits)
names=["res2","res3","res5","y","feat_eee_boundary","z1","feat_center"]
load(11)
side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): step()
torch.cuda.current_stream().wait_stream(side)
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g): step()
load(12); step(); torch.cuda.synchronize()
e_off=offsets.clone(); e_log=logits.clone(); e_t={k:eng.debug_tensor(k,1).clone() for k in names}
step(); torch.cuda.synchronize(); print("eager twice equal:", torch.equal(e_log,logits))
offsets.zero_(); logits.zero_()
g.replay(); torch.cuda.synchronize()
print("offsets equal", torch.equal(e_off,offsets), "logits equal", torch.equal(e_log,logits), float((e_log-logits).abs().max()))
for k in names:
    t=eng.debug_tensor(k,1); print(k, torch.equal(t,e_t[k]), float((t-e_t[k]).abs().max()))
load(11); step(); torch.cuda.synchronize(); l11=logits.clone()
load(12); g.replay(); torch.cuda.synchronize(); print("replay(12) == eager(11)?", torch.equal(logits,l11), " == eager(12)?", torch.equal(logits,e_log))
