#!/usr/bin/env python3
"""Per-launch time of the InstanceNorm statistics / backward kernels at the trunk shape (library HIP-event profiler)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bubbleformer_amd import _lib as L, kernels as K

h = L.lib()
for Fr in (128, 64, 32):
    S, C = 144, 384
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(Fr, S, C, device="cuda", generator=g).bfloat16()
    dy = torch.randn(Fr, S, C, device="cuda", generator=g).bfloat16()
    w = torch.randn(C, device="cuda", generator=g)
    b = torch.randn(C, device="cuda", generator=g)
    for _ in range(3):
        mean, rstd, _, _ = K.in_stats(x, Fr, S, C, w, b)
        K.in_bwd(dy, x, Fr, S, C, mean, rstd, w, b)
    torch.cuda.synchronize()
    h.bf_prof_enable(1)
    for _ in range(30):
        mean, rstd, _, _ = K.in_stats(x, Fr, S, C, w, b)
        K.in_bwd(dy, x, Fr, S, C, mean, rstd, w, b)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    h.bf_prof_report(buf, len(buf))
    h.bf_prof_enable(0)
    rep = json.loads(buf.value.decode())
    print(Fr, {k: round(v["ms"] / v["calls"] * 1e3, 1) for k, v in rep.items()})
