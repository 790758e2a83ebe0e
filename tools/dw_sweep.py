#!/usr/bin/env python3
"""split-K sweep for the token-reduction (dW) GEMM at the bench shapes."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd import _lib as L, kernels as K
N, E = 18432, 384
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, E, device="cuda", generator=g).to(dt); x4 = torch.randn(N, 4 * E, device="cuda", generator=g).to(dt)
gw = torch.zeros(4 * E, E, device="cuda"); gw1 = torch.zeros(E, E, device="cuda")
XC = L.BF_LAY_XC
h = L.lib()
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); h.bf_prof_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14); h.bf_prof_report(buf, len(buf)); h.bf_prof_enable(0)
    (k, v), = json.loads(buf.value.decode()).items()
    return v["ms"] / v["calls"] * 1e3, v["flops"] / v["calls"]
for sk in (2, 4, 7, 11, 15, 22, 30):
    us, fl = timeit(lambda: K.gemm(dt, 4 * E, E, N, K.operand(x4, 4 * E, layout=XC), K.operand(x, E, layout=XC), K.epilogue(gw, E, out_mode=L.BF_OUT_ATOMIC_F32), splitk=sk))
    us2, fl2 = timeit(lambda: K.gemm(dt, E, E, N, K.operand(x, E, layout=XC), K.operand(x, E, layout=XC), K.epilogue(gw1, E, out_mode=L.BF_OUT_ATOMIC_F32), splitk=sk * 4))
    print(f"splitk {sk:3d}: fc1 dW 1536x384 {us:8.1f} us {fl/us/1e6:7.1f} TF | 384x384 (splitk {sk*4:3d}) {us2:8.1f} us {fl2/us2/1e6:7.1f} TF")
