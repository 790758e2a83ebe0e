#!/usr/bin/env python3
"""Isolated timing of the whole-frame fused kernel (gemm_frame.hip) against the two-kernel path it replaces, at the bench shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bubbleformer_amd import kernels as K


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    Fr, S, N = 128, 144, 384
    M = Fr * S
    g = torch.Generator(device="cuda").manual_seed(0)
    for Kd, with_add in ((1152, True), (384, False)):
        A = torch.randn(M, Kd, device="cuda", generator=g).bfloat16()
        W = (torch.randn(Kd, N, device="cuda", generator=g) / Kd ** 0.5).bfloat16()
        x = torch.randn(Fr, S, N, device="cuda", generator=g).bfloat16()
        add = torch.randn(M, N, device="cuda", generator=g).bfloat16() if with_add else None
        w = torch.randn(N, device="cuda", generator=g)
        b = torch.randn(N, device="cuda", generator=g)
        mean, rstd, _, _ = K.in_stats(x, Fr, S, N, w, b)
        dyb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        opA, opW, epi = K.operand(A, Kd, K.L.BF_LAY_KC), K.operand(W, N, K.L.BF_LAY_XC), K.epilogue(dyb, N)
        t_f = timeit(lambda: K.gemm_inbwd_frames(A, W, x.view(M, N), S, mean, rstd, w, add=add))
        t_g = timeit(lambda: K.gemm(torch.bfloat16, M, N, Kd, opA, opW, epi))
        t_i = timeit(lambda: K.in_bwd(dyb.view(Fr, S, N), x, Fr, S, N, mean, rstd, w, b, add=add.view(Fr, S, N) if with_add else None))
        print(f"K={Kd} add={with_add}: fused {t_f:.1f} us | gemm {t_g:.1f} + in_bwd {t_i:.1f} = {t_g + t_i:.1f} us  ({2.0 * M * N * Kd / t_f / 1e6:.0f} TFLOP/s fused)")


if __name__ == "__main__":
    main()
