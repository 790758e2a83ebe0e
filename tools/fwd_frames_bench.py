"""The forward frame-pair kernel (bf_gemm_fwd_frames) against the launches it replaces, timed alone with HIP events at the bench shape
(128 frames of 144 tokens, E = 384): out-projection (K = 384) + the next stage's norm1; fc2 (K = 1536) + MLP-branch norm + residual (+ next
norm1).  PYTHONPATH=. python tools/fwd_frames_bench.py"""
import ctypes as C
import torch
from bubbleformer_amd import _lib as L, kernels as K
from bubbleformer_amd.ops import _p, _stream
lib = L.lib()
Fr, S, N = 128, 144, 384
M = Fr * S
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
resid = rn(M, N).bfloat16()
bias, alpha, beta, w1, b1, w2, b2 = (rn(N) for _ in range(7))
gt = 0.5 + torch.rand(Fr, N, device="cuda", generator=g)
drop = torch.ones(Fr // 16, device="cuda")
st = lambda: [torch.empty(Fr, N, device="cuda") for _ in range(4)]
buf = lambda: torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ws = torch.empty(lib.bf_in_ws_floats(1, Fr, S, N), device="cuda")


def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3


def rec(w, b, gp, s, res, out):
    return L.FrameNorm(_p(w), _p(b), _p(gp), 1, _p(s[0]), _p(s[1]), _p(s[2]), _p(s[3]), _p(res), _p(out))


for Kd, kind in ((384, "outproj"), (1536, "fc2")):
    A = (rn(M, Kd) * 0.7).bfloat16()
    W = (rn(N, Kd) / Kd ** 0.5).bfloat16()
    Wt = W.t().contiguous()
    o1, o2, o3, s1, s2 = buf(), buf(), buf(), st(), st()
    if kind == "outproj":
        epi = K.epilogue(o1, N, colscale=alpha, colshift=beta, aux_mode=L.BF_AUX_ADD, aux=resid, ld_aux=N, rowscale=drop, rows_per_group=16 * S)
        gemm = lambda: K.gemm(torch.bfloat16, M, N, Kd, K.operand(A, Kd), K.operand(W, Kd), epi)
        n2 = rec(w2, b2, None, s2, None, o3)
        plain = lambda: lib.bf_gemm_fwd_frames(1, M, N, Kd, _p(A), Kd, _p(Wt), N, None, _p(alpha), _p(beta), _p(drop), 16, _p(resid), _p(o1), S, None, None, _stream())
        fused = lambda: lib.bf_gemm_fwd_frames(1, M, N, Kd, _p(A), Kd, _p(Wt), N, None, _p(alpha), _p(beta), _p(drop), 16, _p(resid), _p(o1), S, None, C.byref(n2), _stream())
        print("%s: stream gemm %.1f us | pair fwd plain %.1f | + next norm %.1f" % (kind, t(gemm), t(plain), t(fused)))
    else:
        epi = K.epilogue(o1, N, bias=bias)
        gemm = lambda: K.gemm(torch.bfloat16, M, N, Kd, K.operand(A, Kd), K.operand(W, Kd), epi)
        n1, n2 = rec(w1, b1, gt, s1, resid, o2), rec(w2, b2, None, s2, None, o3)
        plain = lambda: lib.bf_gemm_fwd_frames(1, M, N, Kd, _p(A), Kd, _p(Wt), N, _p(bias), None, None, None, 1, None, _p(o1), S, None, None, _stream())
        f1 = lambda: lib.bf_gemm_fwd_frames(1, M, N, Kd, _p(A), Kd, _p(Wt), N, _p(bias), None, None, None, 1, None, _p(o1), S, C.byref(n1), None, _stream())
        f2 = lambda: lib.bf_gemm_fwd_frames(1, M, N, Kd, _p(A), Kd, _p(Wt), N, _p(bias), None, None, None, 1, None, _p(o1), S, C.byref(n1), C.byref(n2), _stream())
        print("%s: stream gemm %.1f us | pair fwd plain %.1f | + MLP norm %.1f | + next norm %.1f" % (kind, t(gemm), t(plain), t(f1), t(f2)))
x = rn(Fr, S, N).bfloat16()
stats = lambda: lib.bf_in_stats(1, _p(x), Fr, S, N, _p(w1), _p(b1), None, 1, None, _p(s1[0]), _p(s1[1]), _p(s1[2]), _p(s1[3]), _p(ws), _stream())
apply_ = lambda: lib.bf_affine_apply(1, _p(x), _p(resid), _p(s1[2]), _p(s1[3]), _p(o1), M, S, N, _stream())
print("separate: bf_in_stats %.1f us, bf_affine_apply %.1f us" % (t(stats), t(apply_)))
