#!/usr/bin/env python3
"""Error of the direct kernel and of the three Winograd variants against float64 on one 3x3 layer (profiles/r01l_winograd_error.txt)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from quber_amd import _lib
lib=_lib.load(); lib.quber_set_tuning(2,1)
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
p=lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
torch.manual_seed(0)
for name,gen in [("randn",lambda s: torch.randn(s,device="cuda")),("relu(randn)+0.5",lambda s: torch.randn(s,device="cuda").relu()+0.5),("uniform[0,4]",lambda s: torch.rand(s,device="cuda")*4)]:
    for (B,H,W,Cin,Cout,d) in [(2,60,80,512,512,1),(2,120,160,128,128,1),(2,30,40,512,512,2)]:
        x=gen((B,H,W,Cin)); w=torch.randn(Cout,Cin,3,3,device="cuda")/np.sqrt(Cin*9)
        yd=torch.empty(B,H,W,Cout,device="cuda"); packed=torch.empty(Cout*9*Cin,device="cuda")
        _lib.check(lib.quber_op_conv2d(p(x),B,H,W,Cin,p(w),Cout,3,1,d,d,p(None),p(None),p(None),0,p(packed),p(yd),st))
        ref=torch.nn.functional.conv2d(x.permute(0,3,1,2).double(), w.double(), None,1,d,d).permute(0,2,3,1)
        out=[]
        for m in (2,4,6):
            P=(m+2)**2; tiles=B*d*d*((-(-H//d)+m-1)//m)*((-(-W//d)+m-1)//m)
            u=torch.empty(P*Cout*Cin,device="cuda"); ws=torch.empty(P*tiles*(Cin+Cout),device="cuda"); y=torch.empty_like(yd)
            _lib.check(lib.quber_op_conv3x3_winograd(p(x),B,H,W,Cin,p(w),Cout,d,m,p(None),p(None),0,p(u),p(ws),ws.numel(),p(y),st))
            out.append(float((y.double()-ref).abs().max()/ref.abs().max()))
        ed=float((yd.double()-ref).abs().max()/ref.abs().max())
        print(f"{name:18s} C={Cin:4d} d={d}: max err / max|y|  direct {ed:.1e}  F2 {out[0]:.1e}  F4 {out[1]:.1e}  F6 {out[2]:.1e}   (max|y| {float(ref.abs().max()):.2f}, rms {float(ref.pow(2).mean().sqrt()):.2f})")
