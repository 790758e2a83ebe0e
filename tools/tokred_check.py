#!/usr/bin/env python3
"""Error map of bf_gemm_tokred against fp64 per 16 x 16 output tile (debugging aid).  Usage: python tools/tokred_check.py Nout Kin M"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd import kernels as K  # noqa: E402
Nout, Kin, M = (int(a) for a in sys.argv[1:4])
g = torch.Generator(device="cuda").manual_seed(3)
dy = torch.randn(M, Nout, device="cuda", generator=g).bfloat16()
x = torch.randn(M, Kin, device="cuda", generator=g).bfloat16()
out = torch.zeros(Nout, Kin, device="cuda"); cs = torch.zeros(Nout, device="cuda")
assert K.gemm_tokred(dy, x, out, accumulate=False, colsum=cs)
ref = dy.double().t() @ x.double()
err = (out.double() - ref).abs().reshape(Nout // 16, 16, Kin // 16, 16).amax(dim=(1, 3)) / ref.abs().max()
print("max rel err %.3g, colsum err %.3g" % (float(err.max()), float((cs.double() - dy.double().sum(0)).abs().max() / dy.double().sum(0).abs().max())))
bad = (err > 1e-5).nonzero()
print("bad 16x16 tiles: %d of %d" % (len(bad), err.numel()))
for r in range(err.shape[0]):
    print("".join("x" if e > 1e-5 else "." for e in err[r].tolist()))
