#!/usr/bin/env python3
"""One GEMM shape of the trunk, a few launches (for rocprofv3 --pmc).  Usage: one_gemm.py qkv|fc1|out|dwqkv [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd import _lib as L, kernels as K  # noqa: E402
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, E = 18432, 384
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x = r(N, E).to(dt)
if which in ("qkv", "fc1", "out"):
    Nn = {"qkv": 3 * E, "fc1": 4 * E, "out": E}[which]
    w = (r(Nn, E) * 0.05).to(dt); bias = r(Nn)
    o = torch.empty(N, Nn, device="cuda", dtype=dt)
    o2 = torch.empty(N, Nn, device="cuda", dtype=dt) if which == "fc1" else None
    fn = lambda: K.gemm(dt, N, Nn, E, K.operand(x, E), K.operand(w, E), K.epilogue(o, Nn, bias=bias, gelu_out=o2))
else:
    dy = r(N, 3 * E).to(dt)
    out = torch.zeros(3 * E, E, device="cuda"); cs = torch.zeros(3 * E, device="cuda")
    fn = lambda: K.gemm_tokred(dy, x, out, accumulate=True, colsum=cs)
for _ in range(reps):
    fn()
torch.cuda.synchronize()
