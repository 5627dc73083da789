#!/usr/bin/env python3
"""ASPP d = 18 layer: dense K loop vs skipping the filter rows that are zero padding (tuning key 11), by split factor."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from quber_amd import _lib
lib=_lib.load(); lib.quber_set_tuning(2,1)
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
p=lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
B,H,W,Cin,Cout,d=16,30,40,2048,256,18
x=torch.randn(B,H,W,Cin,device="cuda"); w=torch.randn(Cout,Cin,3,3,device="cuda")/np.sqrt(Cin*9)
y=torch.empty(B,H,W,Cout,device="cuda"); packed=torch.empty(Cout*9*Cin,device="cuda")
for skip in (0,1,0,1):
    lib.quber_set_tuning(11,skip)
    for S in (0,2,3,4,6,8):
        lib.quber_set_tuning(3,S)
        ts=[]
        for rd in range(6):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): _lib.check(lib.quber_op_conv2d(p(x),B,H,W,Cin,p(w),Cout,3,1,d,d,p(None),p(None),p(None),0,p(packed),p(y),st))
            e1.record(); torch.cuda.synchronize()
            if rd: ts.append(e0.elapsed_time(e1)/3)
        print(f"skip={skip} S={S}: {np.median(ts):.3f} ms")
