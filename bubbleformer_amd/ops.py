"""autograd bindings of the stage-level HIP entry points.

Activations between stages are token-major tensors of shape (B, T, h, w, E) (contiguous) in the compute dtype
(torch.float32 = exact-fp32 parity mode, torch.bfloat16 = throughput mode).  The reference's logical layout
(B, T, E, h, w) is a zero-copy ``permute`` of that memory (see ``as_reference_layout`` / ``as_tokens``).

Parameters stay ordinary fp32 ``nn.Parameter``s owned by the modules; every Function returns one gradient per
parameter so autograd accumulation hooks (and therefore DDP bucket hooks) fire per stage during backward.
"""
import collections
import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _lib as L

_SCRATCH = {}


def _require_gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise L.BubbleformerHipError(
            "bubbleformer_amd runs only on a ROCm GPU (gfx950): got a tensor on %s; there is no CPU fallback" % t.device)


def _dt(t: torch.dtype) -> int:
    if t == torch.float32:
        return L.BF_DTYPE_F32
    if t == torch.bfloat16:
        return L.BF_DTYPE_BF16
    raise L.BubbleformerHipError(f"unsupported compute dtype {t}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def make_dims(dtype, B, T, h, w, E, heads, attn_scale=True, feat_scale=True, patch=0, cin=0, cout=0, nfluid=0) -> L.Dims:
    return L.Dims(_dt(dtype), B, T, h, w, E, heads, int(bool(attn_scale)), int(bool(feat_scale)), patch, cin, cout, nfluid)


def _dims_key(d: L.Dims):
    return tuple(getattr(d, n) for n, _ in L.Dims._fields_)


_SCRATCH_BYTES = {}


_SCRATCH_PINNED: list = []      # arenas a HIP-graph capture has seen: a captured graph holds their raw addresses, so they are never freed


def scratch_for(d: L.Dims, device) -> torch.Tensor:
    """The transient arena of the current stream on `device`, grown to the largest problem seen on that stream (a ragged last batch or
    a validation batch do not pin further arenas): stage calls are stream-ordered and a stage never keeps scratch contents across
    calls, so every shape can share it.  One arena per (device, stream): two streams of one device never share scratch, whatever
    their host threads do (BF_SCRATCH_SHARED=1 opts into ONE arena per device for callers that order their streams themselves).
    An arena that was handed out while its stream was being captured into a HIP graph is pinned: the graph's kernels hold its raw
    address, so a later, larger problem gets a NEW arena and the captured one stays alive until clear_scratch()."""
    key = _dims_key(d)
    n = _SCRATCH_BYTES.get(key)
    if n is None:
        n = L.lib().bf_scratch_bytes(C.byref(d))
        if n < 0:
            L.check(-1, "bf_scratch_bytes")
        _SCRATCH_BYTES[key] = n
    slot = str(device) if os.environ.get("BF_SCRATCH_SHARED") == "1" else (str(device), _stream())
    capturing = torch.cuda.is_current_stream_capturing()
    buf = _SCRATCH.get(slot)
    if buf is None or buf.numel() < n:
        if buf is not None and not capturing:       # the library's side stream may still read the old arena: order it before the arena can be recycled
            L.check(L.lib().bf_side_join(_stream()), "bf_side_join")
        buf = torch.empty(n, dtype=torch.uint8, device=device)
        _SCRATCH[slot] = buf
    if capturing and not any(b is buf for b in _SCRATCH_PINNED):
        _SCRATCH_PINNED.append(buf)
    return buf


def clear_scratch() -> None:
    """Release the scratch arenas, the pinned ones included (after the work -- and every captured graph -- that used them is gone)."""
    if _SCRATCH:
        L.check(L.lib().bf_side_join(_stream()), "bf_side_join")
    _SCRATCH.clear()
    _SCRATCH_PINNED.clear()


def _saved(nbytes: int, device, what: str) -> torch.Tensor:
    if nbytes < 0:
        L.check(-1, what)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _f32c(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if p is None:
        return None
    _require_gpu(p)
    if p.dtype != torch.float32:
        raise L.BubbleformerHipError("parameters must be fp32 (master weights); got %s" % p.dtype)
    return p if p.is_contiguous() else p.contiguous()


def _grad_views(params: Sequence[Optional[torch.Tensor]]):
    """One zeroed flat fp32 buffer for a stage's parameter gradients, and per-parameter views into it."""
    total = sum(((p.numel() + 3) // 4) * 4 for p in params if p is not None)
    dev = next(p.device for p in params if p is not None)
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    views, off = [], 0
    for p in params:
        if p is None:
            views.append(None)
            continue
        views.append(flat[off:off + p.numel()].view(p.shape))
        off += ((p.numel() + 3) // 4) * 4
    return flat, views


# "direct gradient" mode (used by trainer.TrainStep): when every parameter of a stage has a registered fp32 gradient
# slot (a view into the flat gradient buffer), the kernels accumulate straight into it and autograd gets None -- no
# per-parameter temporaries, no 500 tiny accumulate kernels per step.  `on_ready` tells the bucket reducer that a
# stage's gradients are enqueued on the current stream.  Outside this mode gradients are returned to autograd as usual
# (so DDP / optimizer hooks of an unmodified training script still fire).
_DIRECT = {"slots": None, "on_ready": None, "on_join": None}


def set_direct_grad_slots(slots, on_ready=None, on_join=None) -> None:
    """slots: {param.data_ptr(): fp32 gradient view} or None to switch the mode off.  on_join(): called when everything the stages so
    far put on the library's side stream has been ordered on the current stream (see _joined)."""
    _DIRECT["slots"] = slots
    _DIRECT["on_ready"] = on_ready
    _DIRECT["on_join"] = on_join


# Deferred weight-gradient work (bf_side_defer): a trunk stage's side-stream GEMMs may still read its saved activations and its
# incoming gradient while the NEXT stage runs, so those tensors are kept alive for two more stage calls (the library orders the
# side work before the end of the next stage; the caching allocator knows nothing about the library's side stream).
_DEFER = {"on": False, "keep": collections.deque(maxlen=2)}


def set_side_defer(on: bool) -> None:
    """on: opt in (TrainStep does, around forward + backward).  off: join the side stream on the current stream and drop the kept tensors."""
    h = L.lib()
    h.bf_side_defer(1 if on else 0)
    _DEFER["on"] = bool(on)
    if not on:
        L.check(h.bf_side_join(_stream()), "bf_side_join")
        _DEFER["keep"].clear()


def _joined() -> None:
    """Deferred mode, before the patch-embedding backward (the last, long stage of a backward pass): join the side stream here, on
    the Python side, so that the gradient buckets of the processor blocks can be handed to the all-reduce BEFORE that stage's kernels
    are enqueued and travel under them, instead of after."""
    if _DEFER["on"]:
        L.check(L.lib().bf_side_join(_stream()), "bf_side_join")
        if _DIRECT["on_join"] is not None:
            _DIRECT["on_join"]()


def _stage_grads(params: Sequence[Optional[torch.Tensor]]):
    """-> (gradient tensors the kernels accumulate into, what to hand back to autograd)"""
    slots = _DIRECT["slots"]
    if slots is not None:
        gs = [None if p is None else slots.get(p.data_ptr()) for p in params]
        if all((p is None) == (g is None) for p, g in zip(params, gs)):
            return gs, [None] * len(params), True
    _, views = _grad_views(params)
    return views, views, False


def _stage_done(params, direct: bool) -> None:
    if not direct:      # gradients go back through autograd on this stream: nothing of the stage may still be in flight on the side stream
        L.check(L.lib().bf_side_join(_stream()), "bf_side_join")
    if direct and _DIRECT["on_ready"] is not None:
        _DIRECT["on_ready"]([p.data_ptr() for p in params if p is not None])


def as_tokens(x: torch.Tensor) -> torch.Tensor:
    """(..., E, h, w) reference layout -> (..., h, w, E) token-major contiguous (zero-copy when x is already a
    permuted view of token-major memory)."""
    nd = x.dim()
    return x.permute(*range(nd - 3), nd - 2, nd - 1, nd - 3).contiguous()


def as_reference_layout(tok: torch.Tensor) -> torch.Tensor:
    """(..., h, w, E) token-major -> logical (..., E, h, w) view (no copy)."""
    nd = tok.dim()
    return tok.permute(*range(nd - 3), nd - 1, nd - 3, nd - 2)


# ------------------------------------------------------------------------------------------------ processor blocks
_PREPARED: dict = {}      # (kind, pointer of the stage's first parameter) -> (dims key, saved record with prepared weights)


_CHAIN = {"next": None}
_LAST_SPATIAL = {"v": None}
_LAST_TEMPORAL = {"v": None}      # (output pointer, stochastic-depth factors, T) of the temporal stage just run: the spatial stage behind it remembers them


def chain_next(params, kind: str = "temporal") -> None:
    """Announce the stage (`kind`, its parameters in ``_lib.TEMPORAL_FIELDS`` / ``SPATIAL_FIELDS`` order) that will consume the output of the
    stage called next: when both were prepared by `prepare_stages` for the same shape, that stage's opening InstanceNorm is computed by the
    last launch of the stage in front of it (bf_stage_chain_next: the temporal stage's out-projection, the spatial stage's fc2 + MLP-branch
    norm; bit-identical results, one launch and one read of the activation less)."""
    on = _PREPARED and os.environ.get("BF_STAGE_CHAIN", "1") != "0"
    _CHAIN["next"] = _stage_key(kind, [_f32c(p) for p in params]) if on else None


def _stage_key(kind: str, params) -> tuple:
    return kind, next(p.data_ptr() for p in params if p is not None)


def prepare_stages(tok: torch.Tensor, heads: int, attn_scale: bool, feat_scale: bool, stages) -> None:
    """Parameter preparation of all trunk stages of one forward in one launch per 12 stages (bf_prep_stages) instead of one launch
    inside every stage.  stages: [(kind, params, drop_mlp or None)] in call order; each stage's `saved` record is allocated here and
    handed to the stage's forward, which must follow with the same tok shape / dtype.  bf16 on the GPU only; otherwise a no-op."""
    _PREPARED.clear()
    if not tok.is_cuda or tok.dtype != torch.bfloat16 or not stages or os.environ.get("BF_PREP_AHEAD", "1") == "0":
        return
    B, T, h, w, E = tok.shape
    d = make_dims(tok.dtype, B, T, h, w, E, heads, attn_scale, feat_scale)
    keys = {"temporal": _dims_key(make_dims(tok.dtype, B, T, h, w, E, heads, attn_scale, True)), "spatial": _dims_key(d)}
    lib = L.lib()
    n = len(stages)
    nb = {"temporal": lib.bf_temporal_saved_bytes(C.byref(d)), "spatial": lib.bf_spatial_saved_bytes(C.byref(d))}
    kinds = (C.c_int32 * n)()
    pp, sp, dp = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    keep = []
    for i, (kind, params, drop_mlp) in enumerate(stages):
        params = [_f32c(p) for p in params]
        st = (L.TemporalParams if kind == "temporal" else L.SpatialParams)(*[_p(p) for p in params])
        saved = _saved(nb[kind], tok.device, f"bf_{kind}_saved_bytes")
        if drop_mlp is not None:
            drop_mlp = drop_mlp.contiguous().float()
        keep.append((st, params, drop_mlp))
        kinds[i] = 0 if kind == "temporal" else 1
        pp[i], sp[i], dp[i] = C.addressof(st), _p(saved), _p(drop_mlp)
        _PREPARED[_stage_key(kind, params)] = (keys[kind], saved, drop_mlp, st)
    rc = lib.bf_prep_stages(C.byref(d), n, kinds, pp, sp, dp, _stream())
    if rc != 0:
        _PREPARED.clear()
        L.check(min(rc, 0), "bf_prep_stages")


def discard_prepared() -> None:
    _PREPARED.clear()


# ------------------------------------------------------------------------------------------------ inference forward of the trunk
_EVAL_ARENAS: dict = {}     # (device, parameter pointers, dims) -> [weights stamp, arena]: bf16 weight copies + out-projection folds
_WEIGHTS_EPOCH = [0]        # bumped by this package's own in-place optimizer kernels (they write parameters behind torch's version counters)


def _weights_changed() -> None:
    _WEIGHTS_EPOCH[0] += 1


def _evict_eval_arenas(limit: int = 8) -> None:
    """Least recently used first; entries a captured graph reads are kept."""
    for k in list(_EVAL_ARENAS):
        if len(_EVAL_ARENAS) < limit:
            break
        if not _EVAL_ARENAS[k][2]:
            del _EVAL_ARENAS[k]


def refresh_eval_weights(owner: int) -> None:
    """Re-prepare, in place and on the current stream, the inference weights of every cache entry of `owner` whose parameters changed
    since they were prepared (torch version counters / this package's optimizer epoch).  utils.rollout.GraphedForward calls it before
    each replay: a captured graph contains only bf_trunk_eval_fwd reading the arena, so without it a graph kept across optimizer
    steps would replay with the weights of capture time."""
    for key, ent in _EVAL_ARENAS.items():
        if key[0] != int(owner) or ent[3] is None:
            continue
        d, n, kinds, pp, _structs, plist = ent[3]
        flat = [p for ps in plist for p in ps if p is not None]
        stamp = (_WEIGHTS_EPOCH[0], tuple(p._version for p in flat))
        if stamp != ent[0]:
            L.check(L.lib().bf_trunk_eval_prepare(C.byref(d), n, kinds, pp, _p(ent[1]), _stream()), "bf_trunk_eval_prepare")
            ent[0] = stamp


def clear_eval_weights() -> None:
    """Drop the prepared inference weights (they are re-made on the next eval forward).  Needed only after parameters were rewritten
    through raw pointers by code outside this package; torch in-place ops and ops.adamw_ / ops.lion_ are noticed by themselves."""
    _EVAL_ARENAS.clear()


def trunk_eval_applies(tok: torch.Tensor) -> bool:
    """Whether ops.trunk_eval is the path for this token tensor: bf16 on the GPU, 12 x 12-token frames, E = 384, not switched off, and at most
    BF_TRUNK_EVAL_MAX_FRAMES frames (default 96).  The whole-frame kernels are built for few frames: measured eval forward, 16x192x192 clips,
    batch 1 / 2 / 4 / 8 / 16: 2.23 / 1.83 / 2.76 / 4.67 / 8.97 ms against 3.20 / 2.37 / 3.02 / 4.09 / 7.71 ms for the stage forwards (eager) --
    from 128 frames on the streaming GEMMs of the training-shaped forward win (one workgroup per CU pays its prologue and epilogue serially)."""
    ok = (tok.is_cuda and tok.dtype == torch.bfloat16 and tok.dim() == 5 and tok.shape[2] * tok.shape[3] == 144 and tok.shape[4] == 384
          and tok.shape[2] <= 16 and tok.shape[3] <= 16 and tok.shape[1] <= 32 and os.environ.get("BF_TRUNK_EVAL", "1") != "0"
          and tok.shape[0] * tok.shape[1] <= int(os.environ.get("BF_TRUNK_EVAL_MAX_FRAMES", "96")))
    if (not ok and tok.is_cuda and tok.dtype == torch.bfloat16 and tok.dim() == 5 and tok.shape[0] * tok.shape[1] <= 96
            and os.environ.get("BF_TRUNK_EVAL", "1") != "0" and not _SLOW_EVAL_WARNED):
        # few frames but not the shape the whole-frame kernels are built for (144-token frames, E = 384): say so once instead of silently
        # running the training-shaped stage forwards (about 1.6x slower per rollout step at batch 1)
        _SLOW_EVAL_WARNED.append(True)
        import warnings
        warnings.warn("bubbleformer_amd: eval forward of a %s token tensor takes the stage-by-stage path; the whole-frame inference kernels "
                      "cover 144-token frames with embed_dim 384 only" % (tuple(tok.shape),), RuntimeWarning, stacklevel=3)
    return ok


_SLOW_EVAL_WARNED: list = []


def _stage_param_shapes(kind: str, E: int, heads: int) -> dict:
    """Element counts the native stage reads from each parameter (layers/attention.py:35-63, 149-197): a shorter tensor would be read
    out of bounds on the GPU."""
    d = E // heads
    base = {"gamma": E, "gamma_att": E, "gamma_mlp": E, "attn_scale_factor": heads, "attn_scale_factor_x": heads, "attn_scale_factor_y": heads,
            "low_freq_scalar": E, "high_freq_scalar": E, "norm1_w": E, "norm1_b": E, "norm2_w": E, "norm2_b": E, "mlp_norm_w": E, "mlp_norm_b": E,
            "input_head_w": 3 * E * E, "input_head_b": 3 * E, "output_head_w": E * E, "output_head_b": E, "qnorm_w": d, "qnorm_b": d,
            "knorm_w": d, "knorm_b": d, "rel_pos_emb": 32 * heads, "fc1_w": 4 * E * E, "fc1_b": 4 * E, "fc2_w": 4 * E * E, "fc2_b": E}
    return base


def _check_stage_params(kind: str, params, E: int, heads: int, device) -> None:
    fields = L.TEMPORAL_FIELDS if kind == "temporal" else L.SPATIAL_FIELDS
    if len(params) != len(fields):
        raise L.BubbleformerHipError(f"{kind} stage: expected {len(fields)} parameters, got {len(params)}")
    want = _stage_param_shapes(kind, E, heads)
    for name, p in zip(fields, params):
        if p is None:
            continue
        if p.device != device:
            raise L.BubbleformerHipError(f"{kind} stage: parameter {name} is on {p.device}, the tokens on {device}")
        if name in want and p.numel() != want[name]:
            raise L.BubbleformerHipError(f"{kind} stage: parameter {name} has {p.numel()} elements, the kernels read {want[name]}")


_EVAL_TOKENS = [0]


def new_eval_token() -> int:
    """A process-unique id for one owner of prepared inference weights (a model instance).  The prepared-weights cache is keyed by it:
    parameter ADDRESSES alone would let a new model that the allocator placed where a freed one lived, with equal version counters,
    find the old model's weights."""
    _EVAL_TOKENS[0] += 1
    return _EVAL_TOKENS[0]


def trunk_eval(tok: torch.Tensor, heads: int, attn_scale: bool, feat_scale: bool, stages, owner: int = 0) -> Optional[torch.Tensor]:
    """Eval forward of all trunk stages in ONE native call (bf_trunk_eval_fwd: whole-frame projection kernels with the InstanceNorms
    inside, nothing saved for a backward).  stages: [(kind, params)] in call order.  The bf16 weight copies and out-projection folds
    are prepared once per set of weights (torch version counters + this package's optimizer epoch) and kept per model.
    Returns None when the path does not apply (not bf16 on the GPU, shape not covered, BF_TRUNK_EVAL=0): the caller then runs the
    stage forwards.  Under HIP-graph capture the preparation must already have happened (utils/rollout.py warms up first).
    owner: ops.new_eval_token() of the caller that owns these parameters (models pass theirs); 0 = keyed by parameter addresses only."""
    if not tok.is_cuda or tok.dtype != torch.bfloat16 or not stages or os.environ.get("BF_TRUNK_EVAL", "1") == "0":
        return None
    tok = tok.contiguous()
    if tok.dim() != 5:
        raise L.BubbleformerHipError("trunk_eval: tokens must be (B, T, h, w, E)")
    B, T, h, w, E = tok.shape
    if heads < 1 or E % heads:
        raise L.BubbleformerHipError(f"trunk_eval: embed dim {E} is not a multiple of {heads} heads")
    d = make_dims(tok.dtype, B, T, h, w, E, heads, attn_scale, feat_scale)
    lib = L.lib()
    n = len(stages)
    kinds = (C.c_int32 * n)(*[0 if kind == "temporal" else 1 for kind, _ in stages])
    plist = [[_f32c(p) for p in params] for _, params in stages]
    for (kind, _), ps in zip(stages, plist):
        _check_stage_params(kind, ps, E, heads, tok.device)
    structs = [(L.TemporalParams if kind == "temporal" else L.SpatialParams)(*[_p(p) for p in ps]) for (kind, _), ps in zip(stages, plist)]
    pp = (C.c_void_p * n)(*[C.addressof(s) for s in structs])
    flat = [p for ps in plist for p in ps if p is not None]
    key = (int(owner), str(tok.device), tuple(p.data_ptr() for p in flat), h, w, E, heads, bool(attn_scale), bool(feat_scale))
    stamp = (_WEIGHTS_EPOCH[0], tuple(p._version for p in flat))
    ent = _EVAL_ARENAS.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None or ent[0] != stamp:
        if capturing and ent is None:
            raise L.BubbleformerHipError("trunk_eval: run one eval forward before capturing it in a HIP graph (the weights are prepared on the first call)")
        nbytes = lib.bf_trunk_eval_weights_bytes(C.byref(d), n, kinds)
        arena = ent[1] if ent is not None else _saved(nbytes, tok.device, "bf_trunk_eval_weights_bytes")
        # re-preparing inside a capture puts the launch INTO the graph (same arena): every replay then re-reads the live parameters
        rc = lib.bf_trunk_eval_prepare(C.byref(d), n, kinds, pp, _p(arena), _stream())
        if rc == 1:
            return None
        L.check(rc, "bf_trunk_eval_prepare")
        if ent is None:
            _evict_eval_arenas()
            ent = _EVAL_ARENAS[key] = [stamp, arena, False, None]
        else:
            ent[0] = stamp
    if capturing:
        ent[2] = True         # a captured graph reads this arena by address: never evicted, refreshed in place (refresh_eval_weights)
    ent[3] = (d, n, kinds, pp, structs, plist)      # what a refresh needs (keeps the parameter tensors and ctypes records alive)
    _EVAL_ARENAS[key] = _EVAL_ARENAS.pop(key)       # most recently used last
    out = torch.empty_like(tok)
    rc = lib.bf_trunk_eval_fwd(C.byref(d), n, kinds, pp, _p(ent[1]), _p(tok), _p(out), _p(scratch_for(d, tok.device)), _stream())
    if rc == 1:
        return None
    L.check(rc, "bf_trunk_eval_fwd")
    return out


class _BlockFn(torch.autograd.Function):
    """Shared driver for the temporal and the axial block."""

    @staticmethod
    def forward(ctx, x, kind, heads, attn_scale, feat_scale, drop_a, drop_b, *params):
        # drop_a / drop_b: per-sample stochastic-depth factors (0 or 1/keep) or None
        _require_gpu(x)
        x = x.contiguous()
        B, T, h, w, E = x.shape
        d = make_dims(x.dtype, B, T, h, w, E, heads, attn_scale, feat_scale)
        lib = L.lib()
        params = [_f32c(p) for p in params]
        drop_a = None if drop_a is None else drop_a.contiguous().float()
        drop_b = None if drop_b is None else drop_b.contiguous().float()
        pre = _PREPARED.pop(_stage_key(kind, params), None) if _PREPARED else None
        if pre is not None and (pre[0] != _dims_key(d) or (pre[2] is None) != (drop_b is None) or
                                (drop_b is not None and pre[2].data_ptr() != drop_b.data_ptr())):
            pre = None                                  # prepared for another shape or another stochastic-depth table: prepare here
        nxt, _CHAIN["next"] = _CHAIN["next"], None
        if nxt is not None and nxt[0] != kind:          # the next stage's opening InstanceNorm rides in this stage's last launch
            pn = _PREPARED.get(nxt)
            if pn is not None and pn[0][:7] == _dims_key(d)[:7]:      # same dtype and token geometry (the stages' own switches may differ)
                dn = L.Dims(*pn[0])
                L.check(lib.bf_stage_chain_next(C.byref(dn), 0 if nxt[0] == "temporal" else 1, C.addressof(pn[3]), _p(pn[1])), "bf_stage_chain_next")
        if kind == "temporal":
            st = pre[3] if pre else L.TemporalParams(*[_p(p) for p in params])
            saved = pre[1] if pre else _saved(lib.bf_temporal_saved_bytes(C.byref(d)), x.device, "bf_temporal_saved_bytes")
            fwd = lib.bf_temporal_fwd
        else:
            st = pre[3] if pre else L.SpatialParams(*[_p(p) for p in params])
            saved = pre[1] if pre else _saved(lib.bf_spatial_saved_bytes(C.byref(d)), x.device, "bf_spatial_saved_bytes")
            fwd = lib.bf_spatial_fwd
        out = torch.empty_like(x)
        lib.bf_stage_prepared(1 if pre else 0)
        if kind == "temporal":
            rc = fwd(C.byref(d), C.byref(st), _p(x), _p(out), _p(saved), _p(scratch_for(d, x.device)), _p(drop_a), _stream())
        else:
            rc = fwd(C.byref(d), C.byref(st), _p(x), _p(out), _p(saved), _p(scratch_for(d, x.device)), _p(drop_a), _p(drop_b), _stream())
        L.check(rc, f"bf_{kind}_fwd")
        # Backward chain (bf_stage_chain_tail): a temporal stage fed by a spatial stage's output remembers that stage -- its backward's last
        # kernel produces that stage's output gradient and can open that stage's backward (the MLP-branch InstanceNorm) in the same launch
        last, _LAST_SPATIAL["v"] = _LAST_SPATIAL["v"], None
        lastt, _LAST_TEMPORAL["v"] = _LAST_TEMPORAL["v"], None
        ctx.prev_spatial = None
        ctx.next_scale = None
        if kind == "temporal" and drop_a is not None and x.dtype == torch.bfloat16:
            _LAST_TEMPORAL["v"] = (out.data_ptr(), drop_a, T)
        if kind == "spatial" and lastt is not None and lastt[0] == x.data_ptr():
            ctx.next_scale = (lastt[1], lastt[2])      # the backward of the temporal stage in front scales this stage's dx by these factors (bf_stage_next_scale)
        if kind == "spatial":
            if x.dtype == torch.bfloat16 and os.environ.get("BF_STAGE_CHAIN", "1") != "0":
                _LAST_SPATIAL["v"] = (out.data_ptr(), (tuple(x.shape), x.dtype), st, saved, drop_b is not None, params)
        elif last is not None and last[0] == x.data_ptr() and last[1] == (tuple(x.shape), x.dtype):
            ctx.prev_spatial = last[2:]
        ctx.drops = (drop_a, drop_b)
        ctx.kind, ctx.cfg = kind, (heads, attn_scale, feat_scale)
        ctx.save_for_backward(x, saved, *[p for p in params if p is not None])
        ctx.mask = [p is not None for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        x, saved, *ps = ctx.saved_tensors
        it = iter(ps)
        params = [next(it) if m else None for m in ctx.mask]
        heads, attn_scale, feat_scale = ctx.cfg
        B, T, h, w, E = x.shape
        d = make_dims(x.dtype, B, T, h, w, E, heads, attn_scale, feat_scale)
        lib = L.lib()
        dout = dout.contiguous()
        gviews, ret, direct = _stage_grads(params)
        if ctx.kind == "temporal":
            st, gs, bwd = L.TemporalParams(*[_p(p) for p in params]), L.TemporalParams(*[_p(g) for g in gviews]), lib.bf_temporal_bwd
        else:
            st, gs, bwd = L.SpatialParams(*[_p(p) for p in params]), L.SpatialParams(*[_p(g) for g in gviews]), lib.bf_spatial_bwd
        dx = torch.empty_like(x)
        drop_a, drop_b = ctx.drops
        if ctx.kind == "temporal":
            ps = getattr(ctx, "prev_spatial", None)
            if ps is not None and direct:       # (the chained norm's partial sums are reduced by the spatial stage's own backward, which must follow)
                L.check(lib.bf_stage_chain_tail(C.byref(ps[0]), _p(ps[1]), 1 if ps[2] else 0), "bf_stage_chain_tail")
            rc = bwd(C.byref(d), C.byref(st), C.byref(gs), _p(x), _p(dout), _p(dx), _p(saved), _p(scratch_for(d, x.device)), _p(drop_a), _stream())
            lib.bf_stage_chain_tail(None, None, 0)
        else:
            ns = getattr(ctx, "next_scale", None)
            if ns is not None and direct:
                lib.bf_stage_next_scale(_p(ns[0]), ns[1])
            rc = bwd(C.byref(d), C.byref(st), C.byref(gs), _p(x), _p(dout), _p(dx), _p(saved), _p(scratch_for(d, x.device)), _p(drop_a),
                     _p(drop_b), _stream())
            lib.bf_stage_next_scale(None, 0)
        L.check(rc, f"bf_{ctx.kind}_bwd")
        if _DEFER["on"]:
            _DEFER["keep"].append((saved, dout, x))
        _stage_done(params, direct)
        return (dx, None, None, None, None, None, None, *ret)


# ------------------------------------------------------------------------------------------------ the training trunk in one call per direction
class _TrunkFn(torch.autograd.Function):
    """All trunk stages of a training step through bf_trunk_train_fwd / bf_trunk_train_bwd: one native call per direction instead of one
    Python -> ctypes round trip (and one autograd node) per stage.  Same kernels, same chain hints, same saved records as the per-stage
    Functions; the per-stage gradient-ready notifications of the data-parallel reducer come through a host callback."""

    @staticmethod
    def forward(ctx, x, cfg, kinds, masks, drops, *flat):
        _require_gpu(x)
        heads, attn_scale, feat_scale = cfg
        x = x.contiguous()
        B, T, h, w, E = x.shape
        d = make_dims(x.dtype, B, T, h, w, E, heads, attn_scale, feat_scale)
        lib = L.lib()
        n = len(kinds)
        it = iter(flat)
        plist = [[_f32c(next(it)) if m else None for m in mask] for mask in masks]
        for kind, ps in zip(kinds, plist):
            _check_stage_params(kind, ps, E, heads, x.device)
        structs = [(L.TemporalParams if kind == "temporal" else L.SpatialParams)(*[_p(p) for p in ps]) for kind, ps in zip(kinds, plist)]
        kinds_c = (C.c_int32 * n)(*[0 if k == "temporal" else 1 for k in kinds])
        pp = (C.c_void_p * n)(*[C.addressof(s) for s in structs])
        nb = {"temporal": lib.bf_temporal_saved_bytes(C.byref(d)), "spatial": lib.bf_spatial_saved_bytes(C.byref(d))}
        if min(nb.values()) < 0:
            L.check(-1, "bf_*_saved_bytes")
        offs, total = [], 0
        for k in kinds:
            offs.append(total)
            total += (nb[k] + 255) // 256 * 256
        arena = torch.empty(total, dtype=torch.uint8, device=x.device)
        sp = (C.c_void_p * n)(*[arena.data_ptr() + o for o in offs])
        acts = torch.empty((max(n - 1, 1),) + tuple(x.shape), dtype=x.dtype, device=x.device)      # the outputs of stages 0 .. n - 2 (inputs of 1 .. n - 1)
        out = torch.empty_like(x)
        ap = (C.c_void_p * n)(*([acts[i].data_ptr() for i in range(n - 1)] + [out.data_ptr()]))
        da = [None if dr is None or dr[0] is None else dr[0].contiguous().float() for dr in drops]
        db = [None if dr is None or len(dr) < 2 or dr[1] is None else dr[1].contiguous().float() for dr in drops]
        dap = (C.c_void_p * n)(*[_p(t) for t in da])
        dbp = (C.c_void_p * n)(*[_p(t) for t in db])
        L.check(lib.bf_trunk_train_fwd(C.byref(d), n, kinds_c, pp, sp, dap, dbp, _p(x), ap, _p(scratch_for(d, x.device)), _stream()), "bf_trunk_train_fwd")
        ctx.cfg, ctx.kinds, ctx.masks, ctx.offs = cfg, kinds, masks, offs
        ctx.drops = (da, db)
        ctx.save_for_backward(x, arena, acts, *flat)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, arena, acts, *flat = ctx.saved_tensors
        heads, attn_scale, feat_scale = ctx.cfg
        kinds, masks = ctx.kinds, ctx.masks
        B, T, h, w, E = x.shape
        d = make_dims(x.dtype, B, T, h, w, E, heads, attn_scale, feat_scale)
        lib = L.lib()
        n = len(kinds)
        it = iter(flat)
        plist = [[next(it) if m else None for m in mask] for mask in masks]
        pstructs, gstructs, rets, directs = [], [], [], []
        for kind, ps in zip(kinds, plist):
            gviews, ret, direct = _stage_grads(ps)
            cls = L.TemporalParams if kind == "temporal" else L.SpatialParams
            pstructs.append(cls(*[_p(p) for p in ps]))
            gstructs.append(cls(*[_p(g) for g in gviews]))
            rets.append(ret)
            directs.append(direct)
        kinds_c = (C.c_int32 * n)(*[0 if k == "temporal" else 1 for k in kinds])
        pp = (C.c_void_p * n)(*[C.addressof(s) for s in pstructs])
        gp = (C.c_void_p * n)(*[C.addressof(s) for s in gstructs])
        sp = (C.c_void_p * n)(*[arena.data_ptr() + o for o in ctx.offs])
        ap = (C.c_void_p * n)(*([acts[i].data_ptr() for i in range(n - 1)] + [None]))
        da, db = ctx.drops
        dap = (C.c_void_p * n)(*[_p(t) for t in da])
        dbp = (C.c_void_p * n)(*[_p(t) for t in db])
        dout = dout.contiguous()
        gbuf = torch.empty((3,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        g3 = (C.c_void_p * 3)(*[gbuf[i].data_ptr() for i in range(3)])
        dx = torch.empty_like(x)
        on_ready = _DIRECT["on_ready"]
        errs = []

        def done(i, _user):          # host callback from inside the native call: stage i's backward is enqueued
            try:
                if directs[i] and on_ready is not None:
                    on_ready([p.data_ptr() for p in plist[i] if p is not None])
            except BaseException as e:      # an exception must not unwind through the C frames
                errs.append(e)

        cb = L.STAGE_DONE_FN(done)
        rc = lib.bf_trunk_train_bwd(C.byref(d), n, kinds_c, pp, gp, sp, dap, dbp, _p(x), ap, _p(dout), g3, _p(dx), _p(scratch_for(d, x.device)),
                                    C.cast(cb, C.c_void_p), None, _stream())
        L.check(rc, "bf_trunk_train_bwd")
        if errs:
            raise errs[0]
        if not all(directs):             # gradients go back through autograd on this stream: nothing may still be in flight on the side stream
            L.check(lib.bf_side_join(_stream()), "bf_side_join")
        if _DEFER["on"]:
            _DEFER["keep"].append((arena, acts, dout, x, gbuf))
        flat_ret = [g for ret, mask in zip(rets, masks) for g, m in zip(ret, mask) if m]
        return (dx, None, None, None, None, *flat_ret)


def trunk_train(tok: torch.Tensor, heads: int, attn_scale: bool, feat_scale: bool, stages) -> Optional[torch.Tensor]:
    """Training forward of all trunk stages in ONE native call (and their backward in one more).  stages: [(kind, params, drops)] in call
    order, drops = None, (drop,) for a temporal stage, (drop_att, drop_mlp) for a spatial one.  Returns None when the per-stage path is
    asked for (BF_TRUNK_NATIVE=0, or BF_STAGE_CHAIN=0: the native call always chains the stage heads) -- the caller then runs the stages."""
    if not tok.is_cuda or not stages or os.environ.get("BF_TRUNK_NATIVE", "1") == "0" or os.environ.get("BF_STAGE_CHAIN", "1") == "0":
        return None
    if tok.dim() != 5:
        raise L.BubbleformerHipError("trunk_train: tokens must be (B, T, h, w, E)")
    kinds = tuple(k for k, _, _ in stages)
    masks = tuple(tuple(p is not None for p in ps) for _, ps, _ in stages)
    flat = [p for _, ps, _ in stages for p in ps if p is not None]
    drops = [dr for _, _, dr in stages]
    return _TrunkFn.apply(tok, (heads, bool(attn_scale), bool(feat_scale)), kinds, masks, drops, *flat)


def temporal_block(x: torch.Tensor, heads: int, attn_scale: bool, params: List[Optional[torch.Tensor]], drop=None) -> torch.Tensor:
    """x: (B, T, h, w, E) tokens.  params in ``_lib.TEMPORAL_FIELDS`` order (None where the reference has no parameter).
    drop: optional [B] stochastic-depth factors (0 or 1/keep) for the attention branch."""
    return _BlockFn.apply(x, "temporal", heads, attn_scale, True, drop, None, *params)


def spatial_block(x: torch.Tensor, heads: int, attn_scale: bool, feat_scale: bool, params: List[Optional[torch.Tensor]], drop_att=None,
                  drop_mlp=None) -> torch.Tensor:
    """x: (B, T, h, w, E) tokens (frames = B*T).  params in ``_lib.SPATIAL_FIELDS`` order.
    drop_att / drop_mlp: optional [B*T] stochastic-depth factors for the attention and the MLP branch."""
    return _BlockFn.apply(x, "spatial", heads, attn_scale, feat_scale, drop_att, drop_mlp, *params)


def drop_path_factors(n: int, drop_prob: float, device) -> torch.Tensor:
    """timm.layers.DropPath semantics (scale_by_keep=True): one Bernoulli(keep) / keep factor per sample of dim 0."""
    keep = 1.0 - drop_prob
    return torch.empty(n, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)


# ------------------------------------------------------------------------------------------------ embed / debed
def _stage_arrays(conv, inw, inb):
    n = L.BF_MAX_STAGES
    arr = lambda xs: (L.fp * n)(*([_p(t) for t in xs] + [None] * (n - len(xs))))
    return arr(conv), arr(inw), arr(inb)


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fluid, compute_dtype, patch, embed_dim, nst, *params):
        # params: conv_w[nst], in_w[nst], in_b[nst], then (film_ln_w, film_ln_b, film_w, film_b) or nothing
        _require_gpu(x)
        x = x.contiguous().float()
        B, T, Cin, H, W = x.shape
        if H % patch or W % patch:
            raise L.BubbleformerHipError(f"input {H}x{W} is not a multiple of the patch size {patch}")
        h, w = H // patch, W // patch
        params = [_f32c(p) for p in params]
        conv, inw, inb, film = params[:nst], params[nst:2 * nst], params[2 * nst:3 * nst], params[3 * nst:]
        nfluid = 0
        if film:
            fluid = fluid.contiguous().float()
            nfluid = fluid.shape[1]
        d = make_dims(compute_dtype, B, T, h, w, embed_dim, 1, patch=patch, cin=Cin, cout=1, nfluid=nfluid)
        lib = L.lib()
        cw, iw, ib = _stage_arrays(conv, inw, inb)
        st = L.EmbedParams(cw, iw, ib, *([_p(t) for t in film] if film else [None] * 4))
        saved = _saved(lib.bf_embed_saved_bytes(C.byref(d)), x.device, "bf_embed_saved_bytes")
        out = torch.empty((B, T, h, w, embed_dim), dtype=compute_dtype, device=x.device)
        L.check(lib.bf_embed_fwd(C.byref(d), C.byref(st), _p(x), _p(fluid) if film else None, _p(out), _p(saved),
                                 _p(scratch_for(d, x.device)), _stream()), "bf_embed_fwd")
        ctx.cfg = (compute_dtype, patch, embed_dim, nst, nfluid, tuple(x.shape))
        ctx.save_for_backward(saved, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        saved, *params = ctx.saved_tensors
        compute_dtype, patch, E, nst, nfluid, xshape = ctx.cfg
        B, T, Cin, H, W = xshape
        h, w = H // patch, W // patch
        d = make_dims(compute_dtype, B, T, h, w, E, 1, patch=patch, cin=Cin, cout=1, nfluid=nfluid)
        lib = L.lib()
        conv, inw, inb, film = params[:nst], params[nst:2 * nst], params[2 * nst:3 * nst], params[3 * nst:]
        gv, ret, direct = _stage_grads(params)
        cw, iw, ib = _stage_arrays(conv, inw, inb)
        st = L.EmbedParams(cw, iw, ib, *([_p(t) for t in film] if film else [None] * 4))
        gcw, giw, gib = _stage_arrays(gv[:nst], gv[nst:2 * nst], gv[2 * nst:3 * nst])
        gs = L.EmbedParams(gcw, giw, gib, *([_p(t) for t in gv[3 * nst:]] if film else [None] * 4))
        dout = dout.contiguous()
        dx = torch.empty(xshape, dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[0] else None
        _joined()
        L.check(lib.bf_embed_bwd(C.byref(d), C.byref(st), C.byref(gs), _p(dout), _p(dx), _p(saved), _p(scratch_for(d, dout.device)),
                                 _stream()), "bf_embed_bwd")
        _stage_done(params, direct)
        return (dx, None, None, None, None, None, *ret)


def embed(x, fluid, compute_dtype, patch, embed_dim, conv_w, in_w, in_b, film_params=()):
    """x: (B, T, C, H, W) fp32 clip -> (B, T, h, w, E) tokens.  film_params = (ln_w, ln_b, lin_w, lin_b) or ()."""
    nst = len(conv_w)
    return _EmbedFn.apply(x, fluid, compute_dtype, patch, embed_dim, nst, *conv_w, *in_w, *in_b, *film_params)


class _DebedFn(torch.autograd.Function):
    """tokens -> (B, T, Cout, H, W) fp32; with ``target`` also the fused relative-L2 loss."""

    @staticmethod
    def forward(ctx, x, target, patch, cout, nst, *params):
        _require_gpu(x)
        x = x.contiguous()
        B, T, h, w, E = x.shape
        params = [_f32c(p) for p in params]
        conv, inw, inb = params[:nst], params[nst:2 * nst - 1], params[2 * nst - 1:]
        d = make_dims(x.dtype, B, T, h, w, E, 1, patch=patch, cin=1, cout=cout)
        lib = L.lib()
        cw, iw, ib = _stage_arrays(conv, inw, inb)
        st = L.DebedParams(cw, iw, ib)
        saved = _saved(lib.bf_debed_saved_bytes(C.byref(d)), x.device, "bf_debed_saved_bytes")
        pred = torch.empty((B, T, cout, h * patch, w * patch), dtype=torch.float32, device=x.device)
        loss = torch.zeros((), dtype=torch.float32, device=x.device)
        if target is not None:
            target = target.contiguous().float()
            if target.shape != pred.shape:
                raise L.BubbleformerHipError(f"target shape {tuple(target.shape)} != prediction shape {tuple(pred.shape)}")
        L.check(lib.bf_debed_fwd(C.byref(d), C.byref(st), _p(x), _p(pred), _p(target), _p(loss) if target is not None else None,
                                 _p(saved), _p(scratch_for(d, x.device)), _stream()), "bf_debed_fwd")
        ctx.cfg = (patch, cout, nst, target is not None)
        ctx.save_for_backward(x, saved, pred, target if target is not None else pred, *params)
        ctx.set_materialize_grads(False)      # an unused output must not cost a zero-filled gradient the size of the prediction
        return pred, loss

    @staticmethod
    def backward(ctx, dpred, dloss):
        x, saved, pred, target, *params = ctx.saved_tensors
        patch, cout, nst, fused = ctx.cfg
        if fused and dpred is not None:
            raise L.BubbleformerHipError("debed_with_loss: a gradient w.r.t. the prediction is not supported beside the fused loss; "
                                         "use debed() and a separate loss for that")
        if (dloss if fused else dpred) is None:
            return (None,) * (5 + len(params))
        B, T, h, w, E = x.shape
        d = make_dims(x.dtype, B, T, h, w, E, 1, patch=patch, cin=1, cout=cout)
        lib = L.lib()
        conv, inw, inb = params[:nst], params[nst:2 * nst - 1], params[2 * nst - 1:]
        gv, ret, direct = _stage_grads(params)
        cw, iw, ib = _stage_arrays(conv, inw, inb)
        st = L.DebedParams(cw, iw, ib)
        gcw, giw, gib = _stage_arrays(gv[:nst], gv[nst:2 * nst - 1], gv[2 * nst - 1:])
        gs = L.DebedParams(gcw, giw, gib)
        dx = torch.empty_like(x)
        if fused:
            # loss path: d(pred) = dloss * coef[f, c] * (pred - target); an explicit dpred on top is not supported here
            scale = dloss.contiguous().float().reshape(1)
            L.check(lib.bf_debed_bwd(C.byref(d), C.byref(st), C.byref(gs), _p(x), None, _p(pred), _p(target), _p(scale), _p(dx),
                                     _p(saved), _p(scratch_for(d, x.device)), _stream()), "bf_debed_bwd")
        else:
            dpred = dpred.contiguous().float()
            L.check(lib.bf_debed_bwd(C.byref(d), C.byref(st), C.byref(gs), _p(x), _p(dpred), None, None, None, _p(dx), _p(saved),
                                     _p(scratch_for(d, x.device)), _stream()), "bf_debed_bwd")
        _stage_done(params, direct)
        return (dx, None, None, None, None, *ret)


def debed(x, patch, cout, conv_w, in_w, in_b):
    """x: (B, T, h, w, E) tokens -> (B, T, Cout, H, W) fp32 prediction."""
    pred, _ = _DebedFn.apply(x, None, patch, cout, len(conv_w), *conv_w, *in_w, *in_b)
    return pred


def debed_with_loss(x, target, patch, cout, conv_w, in_w, in_b):
    """Fused debed + relative-L2 loss (LpLoss d=2, p=2, mean B, mean T, sum C).  Returns (loss, pred); only ``loss``
    carries gradient (pred is produced for logging)."""
    pred, loss = _DebedFn.apply(x, target, patch, cout, len(conv_w), *conv_w, *in_w, *in_b)
    return loss, pred.detach()


# ------------------------------------------------------------------------------------------------ optimizer
class _GeluMlpFn(torch.autograd.Function):
    """fc2(gelu(fc1(x))) on the last dimension as four (forward: two) native GEMMs -- `GeluMLP.forward` used on its own
    (bubbleformer/layers/linear_layers.py:18-25); inside the axial block the same GEMMs run as part of bf_spatial_fwd / bwd."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        _require_gpu(x)
        from . import kernels as K
        dt = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32
        D, Hd = w1.shape[1], w1.shape[0]
        x2 = x.reshape(-1, D).to(dt).contiguous()
        M = x2.shape[0]
        w1c, w2c = w1.to(dt).contiguous(), w2.to(dt).contiguous()
        pre = torch.empty(M, Hd, dtype=dt, device=x.device)
        hid = torch.empty(M, Hd, dtype=dt, device=x.device)
        out = torch.empty(M, D, dtype=dt, device=x.device)
        K.gemm(dt, M, Hd, D, K.operand(x2, D), K.operand(w1c, D), K.epilogue(pre, Hd, bias=_f32c(b1), gelu_out=hid))
        K.gemm(dt, M, D, Hd, K.operand(hid, Hd), K.operand(w2c, Hd), K.epilogue(out, D, bias=_f32c(b2)))
        ctx.save_for_backward(x2, pre, hid, w1c, w2c)
        ctx.shape, ctx.in_dtype = x.shape, x.dtype
        return out.reshape(x.shape).to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        from . import kernels as K
        x2, pre, hid, w1c, w2c = ctx.saved_tensors
        dt = x2.dtype
        M, D = x2.shape
        Hd = pre.shape[1]
        dev = x2.device
        dy = dout.reshape(M, D).to(dt).contiguous()
        dw1 = torch.zeros(Hd, D, dtype=torch.float32, device=dev)
        db1 = torch.zeros(Hd, dtype=torch.float32, device=dev)
        dw2 = torch.zeros(D, Hd, dtype=torch.float32, device=dev)
        db2 = torch.zeros(D, dtype=torch.float32, device=dev)
        dpre = torch.empty(M, Hd, dtype=dt, device=dev)
        dx = torch.empty(M, D, dtype=dt, device=dev)
        sk = max(1, min(64, M // 512))
        XC, AT = L.BF_LAY_XC, L.BF_OUT_ATOMIC_F32
        K.gemm(dt, D, Hd, M, K.operand(dy, D, layout=XC), K.operand(hid, Hd, layout=XC), K.epilogue(dw2, Hd, out_mode=AT, colsum=db2), splitk=sk)
        K.gemm(dt, M, Hd, D, K.operand(dy, D), K.operand(w2c, Hd, layout=XC), K.epilogue(dpre, Hd, aux_mode=L.BF_AUX_DGELU, aux=pre, ld_aux=Hd))
        K.gemm(dt, Hd, D, M, K.operand(dpre, Hd, layout=XC), K.operand(x2, D, layout=XC), K.epilogue(dw1, D, out_mode=AT, colsum=db1), splitk=sk)
        K.gemm(dt, M, D, Hd, K.operand(dpre, Hd), K.operand(w1c, D, layout=XC), K.epilogue(dx, D))
        return dx.reshape(ctx.shape).to(ctx.in_dtype), dw1, db1, dw2, db2


def gelu_mlp(x: torch.Tensor, w1, b1, w2, b2) -> torch.Tensor:
    return _GeluMlpFn.apply(x, w1, b1, w2, b2)


class _FilmFn(torch.autograd.Function):
    """`FiLMMLP.forward` used on its own (bubbleformer/layers/linear_layers.py:63-77): gamma, beta = Linear(LayerNorm(cond)).chunk(2);
    out = gamma * x + beta over x (B, T, C, h, w).  Native pieces: bf_film_net_fwd / bwd for the conditioning network,
    bf_affine_apply for the modulation and its data gradient, and the InstanceNorm-backward reduction (mean 0, rstd 1: its
    per-frame partials are exactly sum(dout) and sum(dout * x)) for d gamma / d beta.  In the model FiLM rides inside bf_embed_fwd."""

    @staticmethod
    def forward(ctx, x, cond, lnw, lnb, W, bias):
        _require_gpu(x)
        lib = L.lib()
        B, T, Cc, h, w = x.shape
        dt = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32
        tok = as_tokens(x.to(dt))                                     # (B, T, h, w, C)
        cond = cond.contiguous().float()
        P = cond.shape[1]
        prm = [_f32c(t) for t in (lnw, lnb, W, bias)]
        gb = torch.empty(2, B, Cc, dtype=torch.float32, device=x.device)          # the kernel's layout: [gamma | beta][B][C]
        chat = torch.empty(B, P, dtype=torch.float32, device=x.device)
        crstd = torch.empty(B, dtype=torch.float32, device=x.device)
        L.check(lib.bf_film_net_fwd(_p(cond), *[_p(t) for t in prm], _p(gb), _p(chat), _p(crstd), B, P, 2 * Cc, _stream()), "bf_film_net_fwd")
        gamma, beta = gb[0], gb[1]
        out = torch.empty_like(tok)
        S = T * h * w
        L.check(lib.bf_affine_apply(_dt(dt), _p(tok), None, _p(gamma), _p(beta), _p(out), B * S, S, Cc, _stream()), "bf_affine_apply")
        ctx.save_for_backward(tok, gamma, chat, *prm)
        ctx.dims = (B, T, Cc, h, w, P, x.dtype)
        return as_reference_layout(out).to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        lib = L.lib()
        tok, gamma, chat, lnw, lnb, W, bias = ctx.saved_tensors
        B, T, Cc, h, w, P, in_dtype = ctx.dims
        dt, dev, S = tok.dtype, tok.device, T * h * w
        dtok = as_tokens(dout.to(dt))
        dx = torch.empty_like(tok)
        L.check(lib.bf_affine_apply(_dt(dt), _p(dtok), None, _p(gamma), None, _p(dx), B * S, S, Cc, _stream()), "bf_affine_apply")
        zero = torch.zeros(B, Cc, dtype=torch.float32, device=dev)
        one = torch.ones(B, Cc, dtype=torch.float32, device=dev)
        ws = torch.zeros(lib.bf_in_ws_floats(_dt(dt), B, S, Cc), dtype=torch.float32, device=dev)
        junk = torch.empty_like(tok)
        L.check(lib.bf_in_bwd(_dt(dt), _p(dtok), _p(tok), None, _p(junk), B, S, Cc, _p(zero), _p(one), _p(one[0].contiguous()), _p(zero[0].contiguous()),
                              None, 1, 0, None, None, None, None, _p(ws), _stream()), "bf_in_bwd")
        part = ws[:B * Cc * 2].view(B, Cc, 2)                         # {sum dout, sum dout * x} per (batch element, channel)
        dgb = torch.stack([part[..., 1], part[..., 0]]).contiguous()          # [d gamma | d beta][B][C]
        gr = [torch.zeros_like(t) for t in (W, bias, lnw, lnb)]
        L.check(lib.bf_film_net_bwd(_p(dgb), _p(chat), _p(lnw), _p(lnb), _p(W), *[_p(t) for t in gr], B, P, 2 * Cc, _stream()), "bf_film_net_bwd")
        dW, dbias, dlnw, dlnb = gr
        return as_reference_layout(dx).to(in_dtype), None, dlnw, dlnb, dW, dbias


def film(x: torch.Tensor, cond: torch.Tensor, lnw, lnb, W, bias) -> torch.Tensor:
    return _FilmFn.apply(x, cond, lnw, lnb, W, bias)


def clip_gather(frames: torch.Tensor, first: torch.Tensor, t0: int, T: int, table, Ho: int, Wo: int) -> torch.Tensor:
    """frames [fields][total_frames][H][W] fp32 (device), first [B] int64 absolute first input frame per sample, table =
    (field ids int32 [C], diff fp32 [C], div fp32 [C]) -> (B, T, C, Ho, Wo) fp32 normalised clips (data/dataset.py)."""
    _require_gpu(frames)
    ids, diff, div = table
    nf, total, H, W = frames.shape
    B, Cn = first.numel(), ids.numel()
    out = torch.empty((B, T, Cn, Ho, Wo), dtype=torch.float32, device=frames.device)
    L.check(L.lib().bf_clip_gather(_p(frames), total * H * W, _p(ids), _p(first), int(t0), _p(diff), _p(div), _p(out), B, T, Cn, H, W, Ho, Wo,
                                   _stream()), "bf_clip_gather")
    return out


def clip_gather_batch(frames: torch.Tensor, idx: torch.Tensor, first_tab: torch.Tensor, T: int, in_table, out_table, Ho: int, Wo: int,
                      fluid_tab: Optional[torch.Tensor] = None, file_tab: Optional[torch.Tensor] = None):
    """One launch for a batch: idx [B] int64 sample numbers (device), first_tab / file_tab per-sample tables (device) -> input clips
    (B, T, Cin, Ho, Wo), target clips (B, T, Cout, Ho, Wo) and, when fluid_tab is given, the (B, P) fluid-parameter rows."""
    _require_gpu(frames)
    (iid, idf, idv), (oid, odf, odv) = in_table, out_table
    nf, total, H, W = frames.shape
    B = idx.numel()
    inp = torch.empty((B, T, iid.numel(), Ho, Wo), dtype=torch.float32, device=frames.device)
    out = torch.empty((B, T, oid.numel(), Ho, Wo), dtype=torch.float32, device=frames.device)
    P = int(fluid_tab.shape[1]) if fluid_tab is not None else 0
    fl = torch.empty((B, P), dtype=torch.float32, device=frames.device) if fluid_tab is not None else None
    L.check(L.lib().bf_clip_gather_batch(_p(frames), total * H * W, _p(idx), first_tab.numel(), _p(first_tab), _p(iid), _p(idf), _p(idv), iid.numel(), T, _p(inp),
                                         _p(oid), _p(odf), _p(odv), oid.numel(), T, _p(out), _p(fluid_tab) if fl is not None else None,
                                         _p(file_tab) if fl is not None else None, P, _p(fl) if fl is not None else None, B, H, W, Ho, Wo,
                                         _stream()), "bf_clip_gather_batch")
    return inp, out, fl


def lion_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, lr: float, betas=(0.9, 0.99), weight_decay: float = 0.0,
          grad_scale: float = 1.0) -> None:
    """Fused Lion over flat fp32 buffers (lion_pytorch.Lion semantics, bubbleformer/modules.py:139-140)."""
    _require_gpu(p)
    L.check(L.lib().bf_lion(_p(p), _p(g), _p(m), p.numel(), float(lr), float(betas[0]), float(betas[1]), float(weight_decay),
                            float(grad_scale), _stream()), "bf_lion")
    _weights_changed()


def adamw_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, betas=(0.9, 0.999),
           eps: float = 1e-8, weight_decay: float = 1e-2, grad_scale: float = 1.0) -> None:
    """Fused AdamW over flat fp32 buffers (torch.optim.AdamW semantics, bubbleformer/modules.py:135-136)."""
    _require_gpu(p)
    L.check(L.lib().bf_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), int(step), float(lr), float(betas[0]), float(betas[1]), float(eps),
                             float(weight_decay), float(grad_scale), _stream()), "bf_adamw")
    _weights_changed()
