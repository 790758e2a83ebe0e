"""SpaceTimeBlock, AViT and FiLMConditionedAViT (mirror of bubbleformer/models/axial_vit.py) on the HIP stages.

Constructor signatures, sub-module names and ``state_dict`` keys are the reference's
(axial_vit.py:23-46, 85-128, 173-215).  One extra keyword, ``compute_dtype``, selects the activation storage / MFMA type: the default
torch.float32 is the exact-fp32 parity mode (the reference trains in fp32: no ``precision=`` at scripts/train.py:158-172), so matching
the reference is what a caller gets without asking; torch.bfloat16 is the opt-in throughput mode (what bench.py measures, as
BASELINE configs[1] names bf16).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..layers import AttentionBlock, AxialAttentionBlock, FiLMMLP, HMLPDebed, HMLPEmbed
from ._api import register_model

__all__ = ["AViT", "FiLMConditionedAViT", "SpaceTimeBlock"]


class SpaceTimeBlock(nn.Module):
    """Temporal attention along T, then axial attention along W and H + MLP (axial_vit.py:48-65)."""

    def __init__(self, embed_dim: int = 768, num_heads: int = 12, drop_path: float = 0.0, attn_scale: bool = True,
                 feat_scale: bool = True):
        super().__init__()
        self.temporal = AttentionBlock(embed_dim=embed_dim, num_heads=num_heads, drop_path=drop_path, attn_scale=attn_scale)
        self.spatial = AxialAttentionBlock(embed_dim=embed_dim, num_heads=num_heads, drop_path=drop_path, attn_scale=attn_scale,
                                           feat_scale=feat_scale)

    def forward_tokens(self, tok: torch.Tensor, drops=None, chain_to=None) -> torch.Tensor:
        """drops: optional (temporal [B], axial attention [B*T], MLP [B*T]) stochastic-depth factors drawn by the caller.
        chain_to: the temporal block that consumes this block's output (its opening InstanceNorm can ride in this block's last launch)."""
        if tok.is_cuda:
            ops.chain_next(self.spatial.stage_params(), "spatial")      # the axial block's opening norm rides in the temporal out-projection
        tok = self.temporal.forward_tokens(tok) if drops is None else self.temporal.forward_tokens(tok, drops[0])
        if chain_to is not None and tok.is_cuda:
            ops.chain_next(chain_to.stage_params())
        if drops is None:
            return self.spatial.forward_tokens(tok)
        return self.spatial.forward_tokens(tok, drops[1], drops[2])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (B, T, emb, H, W) -> same."""
        return ops.as_reference_layout(self.forward_tokens(ops.as_tokens(x)))


class _AxialBase(nn.Module):
    def _build(self, input_fields, output_fields, patch_size, embed_dim, num_heads, processor_blocks, drop_path, attn_scale,
               feat_scale, compute_dtype):
        self.compute_dtype = compute_dtype if compute_dtype is not None else torch.float32
        self.patch_size = patch_size
        self.embed = HMLPEmbed(patch_size=patch_size, in_channels=input_fields, embed_dim=embed_dim)
        self.dp = np.linspace(0, drop_path, processor_blocks)

    def _finish(self, output_fields, patch_size, embed_dim, num_heads, processor_blocks, attn_scale, feat_scale):
        self.blocks = nn.ModuleList([
            SpaceTimeBlock(embed_dim=embed_dim, num_heads=num_heads, drop_path=self.dp[i], attn_scale=attn_scale, feat_scale=feat_scale)
            for i in range(processor_blocks)
        ])
        self.debed = HMLPDebed(patch_size=patch_size, embed_dim=embed_dim, out_channels=output_fields)

    def _process(self, tok):
        # training mode: the per-sample stochastic-depth factors of ALL blocks come from one draw (2 launches per step
        # instead of 6 per block); each DropPath call of the reference still gets its own independent Bernoulli(keep) samples.
        if not self.training and not torch.is_grad_enabled() and len(self.blocks) > 0 and ops.trunk_eval_applies(tok):
            # inference: all stages in one native call, InstanceNorms inside the whole-frame projection kernels -- through the
            # dispatcher-visible operator torch.ops.bubbleformer_amd.trunk_eval (torch_ops.py)
            from .. import torch_ops  # noqa: F401  (registers the operators)
            b0 = self.blocks[0]
            kinds, params = [], []
            for blk in self.blocks:
                kinds += [0, 1]
                params += list(blk.temporal.stage_params()) + list(blk.spatial.stage_params())
            owner = self.__dict__.get("_bf_eval_owner")
            if owner is None:
                owner = self.__dict__["_bf_eval_owner"] = ops.new_eval_token()      # this instance's prepared-weights cache key
            return torch.ops.bubbleformer_amd.trunk_eval(tok.contiguous(), b0.temporal.num_heads, bool(b0.temporal.attn_scale),
                                                        bool(b0.spatial.feat_scale), kinds, params, owner)
        rates = [float(getattr(blk.temporal.drop_path, "drop_prob", 0.0)) for blk in self.blocks]
        table = None
        if self.training and any(r > 0.0 for r in rates):
            B, F = tok.shape[0], tok.shape[0] * tok.shape[1]
            cache = self.__dict__.setdefault("_keep_cache", {})
            key = (str(tok.device), tuple(rates))
            if key not in cache:
                cache.clear()
                cache[key] = torch.tensor([1.0 - r for r in rates], dtype=torch.float32, device=tok.device)[:, None]
            keep = cache[key]
            table = torch.bernoulli(keep.expand(len(rates), B + 2 * F)).div_(keep)      # two launches: Bernoulli(keep) / keep
        drops = [None] * len(self.blocks)
        if table is not None:
            for i in range(len(self.blocks)):
                if rates[i] > 0.0:
                    drops[i] = (table[i, :B], table[i, B:B + F], table[i, B + F:])
        if tok.is_cuda and self.blocks:      # all stages in one native call per direction (ops.trunk_train)
            b0 = self.blocks[0]
            seq = []
            for i, blk in enumerate(self.blocks):
                dr = drops[i]
                seq.append(("temporal", blk.temporal.stage_params(), None if dr is None else (dr[0],)))
                seq.append(("spatial", blk.spatial.stage_params(), None if dr is None else (dr[1], dr[2])))
            out = ops.trunk_train(tok, b0.temporal.num_heads, b0.temporal.attn_scale, b0.spatial.feat_scale, seq)
            if out is not None:
                return out
        if tok.is_cuda and self.blocks:      # every stage's parameter preparation in one launch per 12 stages (a no-op outside bf16)
            b0 = self.blocks[0]
            stages = []
            for i, blk in enumerate(self.blocks):
                stages.append(("temporal", blk.temporal.stage_params(), None))
                stages.append(("spatial", blk.spatial.stage_params(), drops[i][2] if drops[i] is not None else None))
            ops.prepare_stages(tok, b0.temporal.num_heads, b0.temporal.attn_scale, b0.spatial.feat_scale, stages)
        try:
            for i, blk in enumerate(self.blocks):
                tok = blk.forward_tokens(tok, drops[i], self.blocks[i + 1].temporal if i + 1 < len(self.blocks) else None)
        finally:
            ops.discard_prepared()       # records nobody consumed (an exception above) must not meet a later call with newer weights
        return tok


@register_model("avit")
class AViT(_AxialBase):
    def __init__(self, input_fields: int = 3, output_fields: int = 3, time_window: int = 12, patch_size: int = 16,
                 embed_dim: int = 768, num_heads: int = 12, processor_blocks: int = 12, drop_path: int = 0.2,
                 attn_scale: bool = True, feat_scale: bool = True, compute_dtype=None):
        super().__init__()
        self.drop_path = drop_path
        self._build(input_fields, output_fields, patch_size, embed_dim, num_heads, processor_blocks, drop_path, attn_scale,
                    feat_scale, compute_dtype)
        self._finish(output_fields, patch_size, embed_dim, num_heads, processor_blocks, attn_scale, feat_scale)

    def tokens(self, x):
        return self._process(self.embed.tokens(x, compute_dtype=self.compute_dtype))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (B, T, C, H, W) -> (B, T, C_out, H, W)."""
        return self.debed.from_tokens(self.tokens(x))

    def forward_loss(self, x, target):
        return self.debed.loss_from_tokens(self.tokens(x), target)


@register_model("filmavit")
class FiLMConditionedAViT(_AxialBase):
    def __init__(self, input_fields: int = 3, output_fields: int = 3, time_window: int = 12, patch_size: int = 16,
                 embed_dim: int = 768, num_heads: int = 12, processor_blocks: int = 12, drop_path: int = 0.2,
                 attn_scale: bool = True, feat_scale: bool = True, num_fluid_params: int = 8, compute_dtype=None):
        super().__init__()
        self._build(input_fields, output_fields, patch_size, embed_dim, num_heads, processor_blocks, drop_path, attn_scale,
                    feat_scale, compute_dtype)
        self.film_embed = FiLMMLP(num_fluid_params, embed_dim)
        self._finish(output_fields, patch_size, embed_dim, num_heads, processor_blocks, attn_scale, feat_scale)

    def tokens(self, x, fluid_params):
        tok = self.embed.tokens(x, fluid_params, self.film_embed.film_params(), compute_dtype=self.compute_dtype)
        return self._process(tok)

    def forward(self, x: torch.Tensor, fluid_params: torch.Tensor) -> torch.Tensor:
        """x: (B, T, C, H, W), fluid_params: (B, num_fluid_params) -> (B, T, C_out, H, W)."""
        return self.debed.from_tokens(self.tokens(x, fluid_params))

    def forward_loss(self, x, fluid_params, target):
        """Fused debed + relative-L2 loss (utils/losses.py:67-94 as configured at modules.py:50): (loss, prediction)."""
        return self.debed.loss_from_tokens(self.tokens(x, fluid_params), target)
