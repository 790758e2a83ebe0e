#!/usr/bin/env python3
"""Whole-frame inference projections (bf_frame_linear) at the batch-1 rollout shapes: time per launch, and the same launches with
every token row aliased to row 0 (lda = 0: the operand comes from L1 / L2 hits) to separate the memory system from the kernel's
own structure.  Usage: python tools/frame_fwd_bench.py [frames]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd import _lib as L, kernels as K  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S, E = 144, 384
M = frames * S
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, E, device="cuda", generator=g).to(dt)
hid = torch.randn(M, 4 * E, device="cuda", generator=g).to(dt)
nw, nb = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda")


def timeit(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def alias(t):
    return t.as_strided(t.shape, (0, 1))


cases = []
for name, N, Kd, kw in (("qkv+norm", 3 * E, E, dict(norm=(nw, nb))), ("qkv", 3 * E, E, {}), ("outproj+norm", E, E, dict(norm=(nw, nb))), ("outproj", E, E, {}),
                        ("outproj+next", E, E, "next"), ("fc1", 4 * E, E, dict(gelu=True)), ("fc2+in", E, 4 * E, "in"), ("fc2+in+next", E, 4 * E, "in+next")):
    w = (torch.randn(N, Kd, device="cuda", generator=g) * 0.05).to(dt)
    a = hid if Kd == 4 * E else x
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g).to(dt)
    one, zero = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    if kw == "in":
        kw = dict(resid=resid, out_norm=(one, zero, one))
    elif kw == "in+next":
        kw = dict(resid=resid, out_norm=(one, zero, one), next_norm=(one, zero))
    elif kw == "next":
        kw = dict(resid=resid, colscale=one, colshift=zero, next_norm=(one, zero))
    t_real = timeit(lambda: K.frame_linear(a, w, frames, S, bias=bias, **kw))
    t_alias = timeit(lambda: K.frame_linear(alias(a), w, frames, S, bias=bias, **kw))
    t_alias2 = timeit(lambda: K.frame_linear(alias(a), alias(w), frames, S, bias=bias, **kw))
    byt = (S * Kd + (N // max(1, (N * frames) // max(1, 1))) * 0) * 2
    print(f"{name:14s} N={N:5d} K={Kd:5d}: {t_real:6.1f} us   rows aliased {t_alias:6.1f} us   rows + weights aliased {t_alias2:6.1f} us", flush=True)
