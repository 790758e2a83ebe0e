"""Dispatcher-visible operators (``torch.library``) over the C ABI.

``north_star`` words the boundary as "stateless ops taking / returning ``at::Tensor``"; the product binds the C ABI with ``ctypes`` (DESIGN.md
section 1 says why).  The inference path is additionally registered here as ``torch.ops.bubbleformer_amd.*`` custom operators -- schema,
fake-tensor (shape) implementation, CUDA-only kernel -- so that it has an operator name in the profiler, passes ``torch.library.opcheck``
and is an opaque node for ``torch.compile`` instead of a graph break:

  bubbleformer_amd::trunk_eval     all SpaceTimeBlocks of FiLMConditionedAViT / AViT in eval mode (models/axial_vit.py; bf_trunk_eval_fwd)
  bubbleformer_amd::frame_linear   one 1x1 conv / Linear of a block with the InstanceNorm in front of / behind it (bf_frame_linear)

The training path stays on ``torch.autograd.Function`` (ops.py): its stages hand saved records and side-stream state to their backward,
which a functional operator schema has no place for.
"""
from typing import List, Optional

import torch
from torch.library import custom_op

from . import _lib as L
from . import ops

_NT, _NS = len(L.TEMPORAL_FIELDS), len(L.SPATIAL_FIELDS)


@custom_op("bubbleformer_amd::trunk_eval", mutates_args=(), device_types="cuda")
def trunk_eval(tok: torch.Tensor, heads: int, attn_scale: bool, feat_scale: bool, kinds: List[int],
               params: List[Optional[torch.Tensor]], owner: int = 0) -> torch.Tensor:
    """tok (B, T, h, w, E) bf16 tokens; kinds[i] 0 = temporal, 1 = axial stage; params: the stages' parameters, flattened in
    ``_lib.TEMPORAL_FIELDS`` / ``_lib.SPATIAL_FIELDS`` order (None where the reference has no parameter); owner: ops.new_eval_token() of
    the model instance (the key of its prepared-weights cache)."""
    stages, at = [], 0
    for k in kinds:
        n = _NT if k == 0 else _NS
        stages.append(("temporal" if k == 0 else "spatial", list(params[at:at + n])))
        at += n
    if at != len(params):
        raise L.BubbleformerHipError("trunk_eval: parameter list does not match the stage kinds")
    out = ops.trunk_eval(tok, heads, attn_scale, feat_scale, stages, owner)
    if out is None:
        raise L.BubbleformerHipError("trunk_eval: shape not covered (bf16, 12 x 12-token frames, E = 384): check ops.trunk_eval_applies first")
    return out


@trunk_eval.register_fake
def _(tok, heads, attn_scale, feat_scale, kinds, params, owner=0):
    if tok.dim() != 5 or heads < 1 or tok.shape[-1] % heads:
        raise L.BubbleformerHipError("trunk_eval: tokens must be (B, T, h, w, E) with E a multiple of the head count")
    if sum(_NT if k == 0 else _NS for k in kinds) != len(params):
        raise L.BubbleformerHipError("trunk_eval: parameter list does not match the stage kinds")
    return torch.empty_like(tok)


@custom_op("bubbleformer_amd::frame_linear", mutates_args=(), device_types="cuda")
def frame_linear(a: torch.Tensor, w: torch.Tensor, frames: int, tokens_per_frame: int, norm_w: Optional[torch.Tensor] = None,
                 norm_b: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, colscale: Optional[torch.Tensor] = None,
                 colshift: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, gelu: bool = False,
                 out_norm_w: Optional[torch.Tensor] = None, out_norm_b: Optional[torch.Tensor] = None,
                 out_norm_g: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epi(IN(a) @ w^T): see bf_frame_linear in include/bubbleformer_hip.h."""
    from . import kernels as K
    out = K.frame_linear(a, w, frames, tokens_per_frame, norm=(norm_w, norm_b) if norm_w is not None else None, bias=bias, colscale=colscale,
                         colshift=colshift, resid=resid, gelu=gelu,
                         out_norm=(out_norm_w, out_norm_b, out_norm_g) if out_norm_w is not None else None)
    if out is None:
        raise L.BubbleformerHipError("frame_linear: shape not covered (bf16, 144-token frames, N % 32 = 0, K = 384 or K % 64 = 0)")
    return out


@frame_linear.register_fake
def _(a, w, frames, tokens_per_frame, norm_w=None, norm_b=None, bias=None, colscale=None, colshift=None, resid=None, gelu=False,
      out_norm_w=None, out_norm_b=None, out_norm_g=None):
    if a.dim() != 2 or w.dim() != 2 or frames * tokens_per_frame != a.shape[0] or w.shape[1] != a.shape[1]:
        raise L.BubbleformerHipError("frame_linear: A must be [frames * tokens_per_frame][K] and W [N][K]")
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        raise L.BubbleformerHipError("frame_linear: A and W must be bfloat16")
    if resid is not None and tuple(resid.shape) != (a.shape[0], w.shape[0]):
        raise L.BubbleformerHipError("frame_linear: resid must be [tokens][N]")
    return a.new_empty((a.shape[0], w.shape[0]))
