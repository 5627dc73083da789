import ctypes as C, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from quber_amd import _lib
from test_gpu_h8 import pack, run
lib = _lib.load()
g = torch.Generator().manual_seed(1)
B, H, W, cin, cout, k = 2, 40, 52, 64, 256, 3
x = torch.randn((B, H, W, cin), generator=g).half().cuda()
w = pack((torch.randn((cout, cin, k, k), generator=g) / 24).half(), 1).cuda()
lib.quber_set_tuning(32, 1)
out = {}
for m in (0, 1):
    lib.quber_set_tuning(31, m)
    out[m], _ = run(lib, x, w, cout, k, 1, 1, 1, 1, None, None, None, False, 0)
print("equal:", torch.equal(out[0], out[1]), "maxdiff", float((out[0].float() - out[1].float()).abs().max()))
