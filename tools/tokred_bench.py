#!/usr/bin/env python3
"""Times the four weight-gradient GEMMs of a FiLMAViT-small block at the bench shape through bf_gemm_tokred and through the split-K
atomic form of bf_gemm, alone on the chip.  Usage: python tools/tokred_bench.py [reps]"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd import _lib as L, kernels as K  # noqa: E402
N, E = 18432, 384
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g).to(dt)
x1, x3, x4 = r(N, E), r(N, 3 * E), r(N, 4 * E)
XC = L.BF_LAY_XC
h = L.lib()
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); h.bf_prof_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14); h.bf_prof_report(buf, len(buf)); h.bf_prof_enable(0)
    return {k: (v["ms"] / v["calls"] * 1e3, v["flops"] / v["calls"], v["bytes"] / v["calls"]) for k, v in json.loads(buf.value.decode()).items()}
for name, dy, x in (("qkv 1152x384", x3, x1), ("out 384x384", x1, x1), ("fc1 1536x384", x4, x1), ("fc2 384x1536", x1, x4)):
    Nout, Kin = dy.shape[1], x.shape[1]
    out = torch.zeros(Nout, Kin, device="cuda"); cs = torch.zeros(Nout, device="cuda")
    res = timeit(lambda: K.gemm_tokred(dy, x, out, accumulate=True, colsum=cs))
    tiles = (Nout // 128) * (Kin // 128)
    sk = max(1, min(round(256 / tiles), N // 64 // 4))
    old = timeit(lambda: K.gemm(dt, Nout, Kin, N, K.operand(dy, Nout, layout=XC), K.operand(x, Kin, layout=XC), K.epilogue(out, Kin, out_mode=L.BF_OUT_ATOMIC_F32, colsum=cs), splitk=sk))
    line = f"{name:14s}"
    for k, (us, fl, by) in res.items():
        line += f" | {k} {us:7.1f} us" + (f" {fl / us / 1e6:7.1f} TF {by / us / 1e3:7.1f} GB/s" if fl else "")
    for k, (us, fl, by) in old.items():
        line += f" || atomic split-K {sk}: {us:7.1f} us {fl / us / 1e6:7.1f} TF"
    print(line, flush=True)
