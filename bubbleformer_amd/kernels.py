"""Thin tensor-level wrappers over the kernel-level C entry points (used by the parity tests and by tools).
No autograd here: these call exactly one native entry point each."""
import ctypes as C

import torch

from . import _lib as L
from .ops import _dt, _p, _stream


def operand(t=None, ld=0, layout=L.BF_LAY_KC, seglen=0, segstride=0, gw=0, gh=0, gc=0, pro=L.BF_PRO_NONE, sc=None, sh=None,
            rows_per_frame=0, nch=0) -> L.Operand:
    return L.Operand(_p(t), ld, layout, seglen, segstride, gw, gh, gc, pro, _p(sc), _p(sh), rows_per_frame, nch)


def epilogue(c, ldc, bias=None, colscale=None, colshift=None, aux_mode=L.BF_AUX_NONE, aux=None, ld_aux=0, out_mode=L.BF_OUT_STORE,
             seglen=0, segstride=0, gw=0, gh=0, gc=0, gelu_out=None, colsum=None, rowscale=None, rows_per_group=0) -> L.Epilogue:
    return L.Epilogue(_p(bias), _p(colscale), _p(colshift), aux_mode, _p(aux), ld_aux, out_mode, _p(c), ldc, seglen, segstride, gw, gh, gc, _p(gelu_out), _p(colsum), _p(rowscale), rows_per_group)


def gemm(dtype, M, N, K, A: L.Operand, B: L.Operand, E: L.Epilogue, splitk=1):
    L.check(L.lib().bf_gemm(_dt(dtype), M, N, K, C.byref(A), C.byref(B), C.byref(E), splitk, _stream()), "bf_gemm")


def gemm_tokred(dy, x, out, accumulate=False, colsum=None):
    """Weight-gradient GEMM out[Nout][Kin] (+)= dy^T x through bf_gemm_tokred; returns False when the shape is not covered."""
    M, Nout = dy.shape
    Kin = x.shape[1]
    n = L.lib().bf_gemm_tokred_ws_floats(Nout, Kin, M)
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    rc = L.lib().bf_gemm_tokred(_dt(x.dtype), Nout, Kin, M, _p(dy), dy.stride(0), _p(x), x.stride(0), _p(out), int(accumulate), _p(colsum),
                                _p(ws), n, _stream())
    if rc == 1:
        return False
    L.check(rc, "bf_gemm_tokred")
    return True


def in_stats(x, frames, S, Cc, w, b, g=None, gdiv=1, gb=None):
    dev = x.device
    mean, rstd, sc, sh = (torch.empty(frames, Cc, dtype=torch.float32, device=dev) for _ in range(4))
    ws = torch.empty(L.lib().bf_in_ws_floats(_dt(x.dtype), frames, S, Cc), dtype=torch.float32, device=dev)
    L.check(L.lib().bf_in_stats(_dt(x.dtype), _p(x), frames, S, Cc, _p(w), _p(b), _p(g), gdiv, _p(gb), _p(mean), _p(rstd), _p(sc),
                                _p(sh), _p(ws), _stream()), "bf_in_stats")
    return mean, rstd, sc, sh


def in_bwd(dy, x, frames, S, Cc, mean, rstd, w, b, add=None, g=None, gdiv=1, gelu=False, use_ws=True):
    """InstanceNorm backward through the C ABI: returns (dx, dw, db)."""
    dev = x.device
    dx = torch.empty_like(x)
    dw = torch.zeros(Cc, dtype=torch.float32, device=dev)
    db = torch.zeros(Cc, dtype=torch.float32, device=dev)
    ws = torch.empty(L.lib().bf_in_ws_floats(_dt(x.dtype), frames, S, Cc), dtype=torch.float32, device=dev) if use_ws else None
    L.check(L.lib().bf_in_bwd(_dt(x.dtype), _p(dy), _p(x), _p(add), _p(dx), frames, S, Cc, _p(mean), _p(rstd), _p(w), _p(b), _p(g), gdiv,
                              int(gelu), _p(dw), _p(db), None, None, _p(ws), _stream()), "bf_in_bwd")
    return dx, dw, db


def gemm_inbwd_frames(A, B, x, S, mean, rstd, w, add=None, fscale=None, fdiv=1):
    """Fused data-gradient GEMM + InstanceNorm backward (whole-frame tiles): returns (dx, ws) or None when the shape is not covered."""
    M, K = A.shape
    N = B.shape[1]
    out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    ws = torch.zeros((M // S) * N * 2, dtype=torch.float32, device=x.device)
    rc = L.lib().bf_gemm_inbwd_frames(_dt(x.dtype), M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(x), _p(add), _p(out), S, _p(mean),
                                      _p(rstd), _p(w), _p(ws), _p(fscale), fdiv, _stream())
    if rc == 1:
        return None
    L.check(rc, "bf_gemm_inbwd_frames")
    return out, ws


def attn_fwd(qkv, out, nseq, Lq, inner, outer_stride, inner_stride, tok_stride, heads, d, qw, qb, kw, kb, emb, hscale, out_scale=1.0,
             accumulate=False):
    L.check(L.lib().bf_attn_fwd(_dt(qkv.dtype), _p(qkv), _p(out), nseq, Lq, inner, outer_stride, inner_stride, tok_stride, heads, d,
                                _p(qw), _p(qb), _p(kw), _p(kb), _p(emb), _p(hscale), out_scale, int(accumulate), _stream()), "bf_attn_fwd")


def check_frame_linear(A, W, frames, S, nw=None, nb=None, bias=None, colscale=None, colshift=None, resid=None, ew=None, eb=None, eg=None,
                       xw=None, xb=None) -> None:
    """Everything bf_frame_linear assumes about its operands, checked on the host: the kernel addresses frames * S rows of A / out / resid
    and N or K entries of every table through raw pointers, so a mismatched call would read or write out of bounds on the GPU."""
    E = L.BubbleformerHipError
    if A.dim() != 2 or W.dim() != 2:
        raise E("frame_linear: A must be [tokens][K], W [N][K]")
    M, K = A.shape
    N = W.shape[0]
    if frames < 1 or S < 1 or frames * S != M:
        raise E(f"frame_linear: frames * tokens_per_frame = {frames} * {S} != {M} rows of A")
    if W.shape[1] != K:
        raise E(f"frame_linear: W has K = {W.shape[1]}, A has K = {K}")
    if A.dtype != torch.bfloat16 or W.dtype != torch.bfloat16:
        raise E(f"frame_linear: A and W must be bfloat16 (got {A.dtype}, {W.dtype})")
    if not A.is_cuda or W.device != A.device:
        raise E("frame_linear: A and W must be on the same ROCm device")
    if A.stride(1) != 1 or W.stride(1) != 1:
        raise E("frame_linear: A and W need unit inner strides")
    if resid is not None and (resid.shape != (M, N) or resid.dtype != torch.bfloat16 or resid.stride(1) != 1 or resid.device != A.device):
        raise E(f"frame_linear: resid must be a bfloat16 [{M}][{N}] tensor with a unit inner stride on {A.device}")
    for name, t, n in (("norm weight", nw, K), ("norm bias", nb, K), ("bias", bias, N), ("colscale", colscale, N), ("colshift", colshift, N),
                       ("out_norm weight", ew, N), ("out_norm bias", eb, N), ("out_norm scale", eg, N), ("next_norm weight", xw, N),
                       ("next_norm bias", xb, N)):
        if t is None:
            continue
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < n or t.device != A.device:
            raise E(f"frame_linear: {name} must be a contiguous fp32 tensor of at least {n} elements on {A.device}")
    if (nw is None) != (nb is None) or (colscale is None) != (colshift is None) or (xw is None) != (xb is None):
        raise E("frame_linear: norm / colscale+colshift / next_norm tables come in pairs")
    if ew is not None and (eb is None or eg is None):
        raise E("frame_linear: out_norm needs (weight, bias, scale)")


def frame_linear(A, W, frames, S, norm=None, bias=None, colscale=None, colshift=None, resid=None, gelu=False, out_norm=None, next_norm=None):
    """Whole-frame inference projection (bf_frame_linear).  norm = (w, b): InstanceNorm in front; out_norm = (w, b, g): out = resid + g * IN(A W^T + bias);
    next_norm = (w, b): also return IN(out) * w + b.  Returns the output (or (out, out_n)), or None when the shape is not covered."""
    nw, nb = norm if norm is not None else (None, None)
    ew, eb, eg = out_norm if out_norm is not None else (None, None, None)
    xw, xb = next_norm if next_norm is not None else (None, None)
    check_frame_linear(A, W, frames, S, nw, nb, bias, colscale, colshift, resid, ew, eb, eg, xw, xb)
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, dtype=A.dtype, device=A.device)
    out_n = torch.empty_like(out) if next_norm is not None else None
    rc = L.lib().bf_frame_linear(_dt(A.dtype), frames, S, K, N, _p(A), A.stride(0), _p(W), W.stride(0), _p(nw), _p(nb), _p(bias), _p(colscale),
                                 _p(colshift), _p(resid), resid.stride(0) if resid is not None else 0, int(gelu), _p(ew), _p(eb), _p(eg),
                                 _p(out), out.stride(0), _p(xw), _p(xb), _p(out_n), N if out_n is not None else 0, _stream())
    if rc == 1:
        return None
    L.check(rc, "bf_frame_linear")
    return out if out_n is None else (out, out_n)
