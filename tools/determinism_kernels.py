#!/usr/bin/env python3
"""Kernel-level run-to-run determinism at the debed / trunk shapes (bf16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd import kernels as K, _lib as L  # noqa: E402
g = torch.Generator(device="cuda").manual_seed(0)
dt = torch.bfloat16
def cmp(name, f):
    a = f(); b = f()
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(a, b)):
        if x is None: continue
        nd = int((x != y).sum())
        print(f"{name}[{i}]: differing {nd} / {x.numel()}  rel {float((x.float()-y.float()).norm()/y.float().norm().clamp_min(1e-30)):.3e}")
for (F, S, C, gelu) in ((16, 36864, 96, True), (16, 9216, 96, True), (16, 144, 384, False), (16, 2304, 96, True)):
    x = torch.randn(F * S, C, device="cuda", generator=g).to(dt)
    dy = torch.randn(F * S, C, device="cuda", generator=g).to(dt)
    w = torch.randn(C, device="cuda", generator=g); b = torch.randn(C, device="cuda", generator=g)
    mean, rstd, sc, sh = K.in_stats(x, F, S, C, w, b)
    cmp(f"in_stats F{F} S{S} C{C}", lambda: K.in_stats(x, F, S, C, w, b))
    cmp(f"in_bwd   F{F} S{S} C{C} gelu{int(gelu)}", lambda: K.in_bwd(dy, x, F, S, C, mean, rstd, w, b, gelu=gelu))
