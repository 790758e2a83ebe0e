#!/usr/bin/env python3
"""Times the three attention geometries of one FiLMAViT-small block at the bench configuration (B=8, T=16, 12x12 tokens, E=384,
6 heads) through bf_attn_fwd / bf_attn_bwd with the library's per-launch HIP-event timing.  Usage: python tools/attn_bench.py [reps]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd import _lib as L  # noqa: E402
from bubbleformer_amd.ops import _dt, _p, _stream  # noqa: E402

B, T, hh, ww, E, heads = int(os.environ.get("ATTN_B", 8)), 16, 12, 12, 384, 6
d = E // heads
S = hh * ww
F = B * T
N = F * S
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(N, 3 * E, device="cuda", generator=g).bfloat16()
dout = torch.randn(N, E, device="cuda", generator=g).bfloat16()
out = torch.zeros(N, E, device="cuda", dtype=torch.bfloat16)
dqkv = torch.zeros_like(qkv)
prm = [torch.ones(d, device="cuda"), torch.zeros(d, device="cuda"), torch.ones(d, device="cuda"), torch.zeros(d, device="cuda"),
       0.1 * torch.randn(32, heads, device="cuda", generator=g), torch.ones(heads, device="cuda")]
grads = [torch.zeros_like(t) for t in prm]
ws_floats = 1024 * (4 * 128 + 32 * 16 + 16)
ws = torch.empty(ws_floats, device="cuda")
GEO = {"temporal (L=16, stride S)": (B * S, T, S, T * S, 1, S), "axial w (L=12, contiguous)": (F * hh, ww, 1, ww, 0, 1),
       "axial h (L=12, stride w)": (F * ww, hh, ww, S, 1, ww)}
h = L.lib()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    h.bf_prof_enable(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14)
    h.bf_prof_report(buf, len(buf))
    h.bf_prof_enable(0)
    return {k: v["ms"] / v["calls"] * 1e3 for k, v in json.loads(buf.value.decode()).items()}


for name, geo in GEO.items():
    f = lambda: L.check(h.bf_attn_fwd(_dt(qkv.dtype), _p(qkv), _p(out), *geo, heads, d, *[_p(t) for t in prm], 1.0, 0, _stream()), "fwd")
    b = lambda: L.check(h.bf_attn_bwd(_dt(qkv.dtype), _p(qkv), _p(dout), _p(dqkv), *geo, heads, d, *[_p(t) for t in prm], *[_p(t) for t in grads],
                                      1.0, 0, _p(ws), ws_floats, _stream()), "bwd")
    print(f"{name:30s} fwd {timed(f)}  bwd {timed(b)}")
