"""Frame-pair data-gradient kernel with and without the chained second InstanceNorm backward, and the separate launch it replaces, timed alone
(HIP events, QKV data-gradient shape of the bench).  PYTHONPATH=. python tools/chain_bench.py   -> MI355X: plain 28.2 us, chained 38.3 us, in_bwd 25.8 us
(the last through its Python wrapper: ~18.5 us of kernel)."""
import torch, time
from bubbleformer_amd import _lib as L, kernels as K
from bubbleformer_amd.ops import _p, _stream
lib = L.lib()
Fr, S, N, Kd = 128, 144, 384, 1152
M = Fr * S
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.randn(M, Kd, device="cuda", generator=g) * 0.5).bfloat16()
W = (torch.randn(Kd, N, device="cuda", generator=g) / Kd ** 0.5).bfloat16()
x = torch.randn(Fr, S, N, device="cuda", generator=g).bfloat16()
add = torch.randn(M, N, device="cuda", generator=g).bfloat16()
w, b = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
mean, rstd, _, _ = K.in_stats(x, Fr, S, N, w, b)
z = torch.randn(Fr, S, N, device="cuda", generator=g).bfloat16()
mean3, rstd3, _, _ = K.in_stats(z, Fr, S, N, w, b)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); dz = torch.empty_like(out)
ws = torch.zeros(Fr * N * 2, device="cuda"); cws = torch.zeros_like(ws)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
plain = lambda: lib.bf_gemm_inbwd_frames(1, M, N, Kd, _p(A), Kd, _p(W), N, _p(x), _p(add), _p(out), S, _p(mean), _p(rstd), _p(w), _p(ws), None, 1, _stream())
chain = lambda: lib.bf_gemm_inbwd_frames_chain(1, M, N, Kd, _p(A), Kd, _p(W), N, _p(x), _p(add), _p(out), S, _p(mean), _p(rstd), _p(w), _p(ws), None, 1, _p(z), _p(dz), _p(mean3), _p(rstd3), _p(w), None, 1, _p(cws), _stream())
inb = lambda: K.in_bwd(out.view(Fr, S, N), z, Fr, S, N, mean3, rstd3, w, b)
print("plain %.1f us  chained %.1f us  in_bwd alone %.1f us" % (t(plain), t(chain), t(inb)))
