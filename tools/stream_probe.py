#!/usr/bin/env python3
"""Where MaskRefiner.predict_stream(batch=k) spends a batch: frames/s per (workers, batch), the main thread's enqueue / collect phases,
and the worker side (_load) alone.  INPAINT=host|device selects where inpaint_depth runs.  GPU box only.
usage: [INPAINT=device] tools/stream_probe.py        (results: profiles/r09f_adapter_stream.txt)"""
import os, sys, time, tempfile
import numpy as np
from PIL import Image
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quber_amd import arch, synth
from quber_amd.eval import refiner_model as rm
from quber_amd.eval.refiner_model import MaskRefiner
import torch
N=20
with tempfile.TemporaryDirectory() as d:
    items=[]; rng=np.random.default_rng(0)
    for i in range(8):
        sc=synth.make_scene(30+i,480,640,N)
        Image.fromarray(sc["rgb"][:,:,::-1].copy()).save(os.path.join(d,f"rgb{i}.png"))
        mm=sc["depth"][:,:,0].astype(np.uint16)*5+300
        for _ in range(12):
            y,x=int(rng.integers(0,440)),int(rng.integers(0,600)); mm[y:y+22,x:x+30]=0
        Image.fromarray(mm).save(os.path.join(d,f"depth{i}.png"))
        items.append((os.path.join(d,f"rgb{i}.png"),os.path.join(d,f"depth{i}.png"),sc["masks"]!=0,None))
    ref=MaskRefiner(None,None,dataset="OSD",inpaint=os.environ.get("INPAINT","host"))
    print("inpaint =", ref.inpaint)
    ref.refiner_predictor.model.state_dict=arch.init_state_dict(seed=0,loud_heads=True,center_bias=-1.68)
    ref.refiner_predictor.model._engines.clear()
    work=[items[i%8] for i in range(128)]
    for wk,bt in ((16,16),(32,16)):
        list(ref.predict_stream(work[:2*bt],workers=wk,batch=bt))
        t0=time.perf_counter(); res=list(ref.predict_stream(work,workers=wk,batch=bt)); dt=time.perf_counter()-t0
        print(f"workers={wk} batch={bt}: {len(work)/dt:.1f} frames/s", flush=True)
    # main-thread phases at (32,16): time enqueue / collect
    model=ref.refiner_predictor.model
    oe, oc = model.enqueue_batch, model.collect_batch
    tt={"enq":0.0,"col":0.0,"n":0}
    def e2(*a,**k):
        t=time.perf_counter(); r=oe(*a,**k); tt["enq"]+=time.perf_counter()-t; tt["n"]+=1; return r
    def c2(*a,**k):
        t=time.perf_counter(); r=oc(*a,**k); tt["col"]+=time.perf_counter()-t; return r
    model.enqueue_batch, model.collect_batch = e2, c2
    t0=time.perf_counter(); res=list(ref.predict_stream(work,workers=32,batch=16)); dt=time.perf_counter()-t0
    print(f"instrumented: {len(work)/dt:.1f} frames/s; per batch: total {dt/tt['n']*1e3:.1f} ms, enqueue_batch {tt['enq']/tt['n']*1e3:.1f} ms, collect_batch (sync + dicts) {tt['col']/tt['n']*1e3:.1f} ms")
    # load alone with 32 threads
    from concurrent.futures import ThreadPoolExecutor
    for wk in (16,):
        with ThreadPoolExecutor(wk) as pool:
            t0=time.perf_counter(); list(pool.map(lambda it: ref._load(*it[:3]), work)); dt=time.perf_counter()-t0
        print(f"_load alone on {wk} threads: {len(work)/dt:.1f} frames/s", flush=True)
