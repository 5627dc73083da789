#!/usr/bin/env python3
"""Tile shapes of the direct kernel on the batched-GEMM shapes the Winograd path produces (P positions x tiles rows, K = Cin)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from quber_amd import _lib
lib=_lib.load(); lib.quber_set_tuning(2,1)
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
p=lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
for (name,M,K,N) in [("F6 128>128 s4",64*8640,128,128),("F6 256>256 s4",64*8640,256,256),("F6 512>512 s8",64*2240,512,512),("F6 256>256 s16 x2",64*2*560,256,256),("F4 128>32 s4",36*19200,128,32),("F4 512>512 d2",36*2*16*80,512,512),("F6 64>64 s4 x2",64*2*8640,64,64)]:
    x=torch.randn(1,M,1,K,device="cuda"); w=torch.randn(N,K,1,1,device="cuda")/np.sqrt(K)
    y=torch.empty(1,M,1,N,device="cuda"); packed=torch.empty(N*K,device="cuda")
    fl=2.0*M*K*N; res=[]
    for tile in (0,1,2,4):
        if tile==4 and N>64: continue
        lib.quber_set_tuning(4,tile); ts=[]
        for rd in range(5):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(3): _lib.check(lib.quber_op_conv2d(p(x),1,M,1,K,p(w),N,1,1,0,1,p(None),p(None),p(None),0,p(packed),p(y),st))
            e1.record(); torch.cuda.synchronize()
            if rd: ts.append(e0.elapsed_time(e1)/3)
        res.append("%s %.3f ms (%.0f TF)"%({0:"auto",1:"64x64",2:"128x128",4:"256x32"}[tile],np.median(ts),fl/np.median(ts)/1e9))
    print(name.ljust(22)," | ".join(res),flush=True)
lib.quber_set_tuning(4,0)
