#!/usr/bin/env python3
"""Per-tile time and fixed cost of the weight-stationary streaming GEMM: the QKV / fc1 shapes at 1x, 2x, 4x the bench token count."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd import _lib as L, kernels as K  # noqa: E402
E = 384
dt = torch.bfloat16
h = L.lib()
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); h.bf_prof_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14); h.bf_prof_report(buf, len(buf)); h.bf_prof_enable(0)
    (k, v), = json.loads(buf.value.decode()).items()
    return k, v["ms"] / v["calls"] * 1e3
for Nn, gelu2 in ((3 * E, False), (4 * E, True), (E, False)):
    w = (torch.randn(Nn, E, device="cuda", generator=g) * 0.05).to(dt)
    bias = torch.randn(Nn, device="cuda", generator=g)
    prev = None
    for mult in (1, 2, 4):
        M = 18432 * mult
        x = torch.randn(M, E, device="cuda", generator=g).to(dt)
        o = torch.empty(M, Nn, device="cuda", dtype=dt)
        o2 = torch.empty(M, Nn, device="cuda", dtype=dt) if gelu2 else None
        name, us = timeit(lambda: K.gemm(dt, M, Nn, E, K.operand(x, E), K.operand(w, E), K.epilogue(o, Nn, bias=bias, gelu_out=o2)))
        fl = 2.0 * M * Nn * E
        print(f"N={Nn:5d} M={M:6d} {name:24s} {us:8.1f} us {fl / us / 1e6:7.1f} TF  out {M * Nn * 2 * (2 if gelu2 else 1) / us / 1e6:6.2f} TB/s" + (f"   marginal {(us - prev) :7.1f} us per {18432 * (mult - mult // 2)} rows" if prev else ""), flush=True)
        prev = us
        del x, o, o2
