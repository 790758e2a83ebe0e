#!/usr/bin/env python3
"""Times the GEMM shapes of one FiLMAViT-small block at the bench configuration (N = 18432 tokens, E = 384) through bf_gemm,
with the library's own per-launch HIP-event timing.  Usage: python tools/gemm_bench.py [reps]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd import _lib as L, kernels as K  # noqa: E402

N, E, S = 18432, 384, 144
F = N // S
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x = r(N, E).to(dt); x4 = r(N, 4 * E).to(dt); x3 = r(N, 3 * E).to(dt)
w_qkv = (r(3 * E, E) * 0.05).to(dt); w_o = (r(E, E) * 0.05).to(dt); w1 = (r(4 * E, E) * 0.05).to(dt); w2 = (r(E, 4 * E) * 0.05).to(dt)
sc = r(F, E); sh = r(F, E); bias3 = r(3 * E); bias4 = r(4 * E); bias1 = r(E)
o3 = torch.empty(N, 3 * E, device="cuda", dtype=dt); o4 = torch.empty(N, 4 * E, device="cuda", dtype=dt); o4b = torch.empty(N, 4 * E, device="cuda", dtype=dt); o1 = torch.empty(N, E, device="cuda", dtype=dt)
gw = torch.zeros(4 * E, E, device="cuda"); gw2 = torch.zeros(E, 4 * E, device="cuda"); gw3 = torch.zeros(3 * E, E, device="cuda")
aff = dict(pro=L.BF_PRO_AFFINE, sc=sc, sh=sh, rows_per_frame=S, nch=E)
XC = L.BF_LAY_XC

CASES = {
    "fwd qkv   A-affine  N=1152 K=384 ": lambda: K.gemm(dt, N, 3 * E, E, K.operand(x, E, **aff), K.operand(w_qkv, E), K.epilogue(o3, 3 * E, bias=bias3)),
    "fwd qkv   plain     N=1152 K=384 ": lambda: K.gemm(dt, N, 3 * E, E, K.operand(x, E), K.operand(w_qkv, E), K.epilogue(o3, 3 * E, bias=bias3)),
    "fwd fc1   gelu2-epi N=1536 K=384 ": lambda: K.gemm(dt, N, 4 * E, E, K.operand(x, E), K.operand(w1, E), K.epilogue(o4, 4 * E, bias=bias4, gelu_out=o4b)),
    "fwd fc2   plain     N=384  K=1536": lambda: K.gemm(dt, N, E, 4 * E, K.operand(x4, 4 * E), K.operand(w2, 4 * E), K.epilogue(o1, E, bias=bias1)),
    "fwd out   A-affine+resid N=384 K=384": lambda: K.gemm(dt, N, E, E, K.operand(x, E, **aff), K.operand(w_o, E), K.epilogue(o1, E, colscale=bias1, colshift=bias1, aux_mode=L.BF_AUX_ADD, aux=x, ld_aux=E)),
    "dA  fc2   dgelu-epi N=1536 K=384 ": lambda: K.gemm(dt, N, 4 * E, E, K.operand(x, E), K.operand(w2, 4 * E, layout=XC), K.epilogue(o4, 4 * E, aux_mode=L.BF_AUX_DGELU, aux=x4, ld_aux=4 * E)),
    "dA  fc1   +resid    N=384  K=1536": lambda: K.gemm(dt, N, E, 4 * E, K.operand(x4, 4 * E), K.operand(w1, E, layout=XC), K.epilogue(o1, E, aux_mode=L.BF_AUX_ADD, aux=x, ld_aux=E)),
    "dA  qkv   plain     N=384  K=1152": lambda: K.gemm(dt, N, E, 3 * E, K.operand(x3, 3 * E), K.operand(w_qkv, E, layout=XC), K.epilogue(o1, E)),
    "dW  fc1   plain     1536x384 K=N ": lambda: K.gemm(dt, 4 * E, E, N, K.operand(x4, 4 * E, layout=XC), K.operand(x, E, layout=XC), K.epilogue(gw, E, out_mode=L.BF_OUT_ATOMIC_F32), splitk=11),
    "dW  fc2   plain     384x1536 K=N ": lambda: K.gemm(dt, E, 4 * E, N, K.operand(x, E, layout=XC), K.operand(x4, 4 * E, layout=XC), K.epilogue(gw2, 4 * E, out_mode=L.BF_OUT_ATOMIC_F32), splitk=11),
    "dW  qkv   plain     1152x384 K=N ": lambda: K.gemm(dt, 3 * E, E, N, K.operand(x3, 3 * E, layout=XC), K.operand(x, E, layout=XC), K.epilogue(gw3, E, out_mode=L.BF_OUT_ATOMIC_F32), splitk=19),
    "dW  qkv   B-affine  1152x384 K=N ": lambda: K.gemm(dt, 3 * E, E, N, K.operand(x3, 3 * E, layout=XC), K.operand(x, E, layout=XC, **aff), K.epilogue(gw3, E, out_mode=L.BF_OUT_ATOMIC_F32), splitk=19),
}
h = L.lib()
print(f"{'case':42s} {'us':>9s} {'TFLOP/s':>9s} {'alg GB/s':>9s}")
for name, fn in CASES.items():
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
    (k, v), = json.loads(buf.value.decode()).items()
    us = v["ms"] / v["calls"] * 1e3
    print(f"{name:42s} {us:9.1f} {v['flops'] / v['calls'] / us / 1e6:9.1f} {v['bytes'] / v['calls'] / us / 1e3:9.1f}")
