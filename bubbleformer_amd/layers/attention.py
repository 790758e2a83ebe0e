"""AttentionBlock (temporal) and AxialAttentionBlock (spatial) -- mirror of bubbleformer/layers/attention.py.

The sub-module tree and parameter names are the reference's (identical ``state_dict``); ``forward`` hands the
parameters to the fused HIP stage (csrc/model.hip).  Inputs/outputs keep the reference's logical shapes
(B, N, E, h, w) / (B, E, h, w); physically they are token-major, so chained blocks exchange memory without copies.
"""
import torch
import torch.nn as nn

from .. import ops
from .linear_layers import GeluMLP
from .positional_encoding import RelativePositionBias


class DropPath(nn.Module):
    """Stochastic depth holder (timm.layers.DropPath at layers/attention.py:64,194).  The mask multiply itself runs inside
    the fused stage (per-sample factor in the out-projection epilogue / the MLP branch's InstanceNorm affine); this module
    only carries the rate and draws the per-sample factors in training mode."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep

    def factors(self, n: int, device):
        if not self.training or self.drop_prob <= 0.0:
            return None
        return ops.drop_path_factors(n, self.drop_prob, device)

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:0.3f}"


def _make_drop_path(p: float):
    return DropPath(p) if p > 0.0 else nn.Identity()


def _factors(mod: nn.Module, n: int, device):
    dp = mod.drop_path
    return dp.factors(n, device) if isinstance(dp, DropPath) else None


def _no_bias(*_):
    """bias_type="none": `self.rel_pos_bias = lambda x, y: None` in the reference (layers/attention.py:58-59) -- the scores get no bias
    term and the block has no embedding table in its state_dict; the stage kernels take a null table pointer for that."""
    return None


def _check_bias_type(bias_type: str) -> None:
    if bias_type == "continuous":
        raise NotImplementedError("bias_type='continuous' (ContinuousPositionBias1D) is not built: no reference config selects it "
                                  "(SURVEY.md section 2 scopes it out); 'rel' (the default) and 'none' are")


def _bias_table(rel_pos_bias):
    return rel_pos_bias.relative_attention_bias.weight if isinstance(rel_pos_bias, RelativePositionBias) else None


def _require_layer_scale(gamma) -> None:
    if gamma is None:       # what `x * self.gamma[None, None, :, None, None]` raises in the reference's forward
        raise TypeError("'NoneType' object is not subscriptable")


class AttentionBlock(nn.Module):
    def __init__(self, embed_dim: int = 768, num_heads: int = 12, drop_path: float = 0, layer_scale_init_value: float = 1e-6,
                 bias_type: str = "rel", attn_scale: bool = True):
        super().__init__()
        _check_bias_type(bias_type)
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.attn_scale = attn_scale
        self.norm1 = nn.InstanceNorm2d(embed_dim, affine=True)
        self.norm2 = nn.InstanceNorm2d(embed_dim, affine=True)
        # layer_scale_init_value <= 0: no parameter, exactly as the reference constructs it (layers/attention.py:41-46) -- and its forward
        # then fails on `self.gamma[None, ...]` (:123), which forward_tokens reproduces
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((embed_dim)), requires_grad=True) if layer_scale_init_value > 0 else None
        self.input_head = nn.Conv2d(embed_dim, 3 * embed_dim, 1)
        self.output_head = nn.Conv2d(embed_dim, embed_dim, 1)
        self.qnorm = nn.LayerNorm(embed_dim // num_heads)
        self.knorm = nn.LayerNorm(embed_dim // num_heads)
        if attn_scale:
            self.attn_scale_factor = nn.Parameter(torch.ones((1, num_heads, 1, 1)), requires_grad=True)
        self.rel_pos_bias = RelativePositionBias(n_heads=num_heads) if bias_type != "none" else _no_bias
        self.drop_path = _make_drop_path(drop_path)

    def stage_params(self):
        _require_layer_scale(self.gamma)
        return [self.gamma, self.attn_scale_factor if self.attn_scale else None, self.norm1.weight, self.norm1.bias,
                self.norm2.weight, self.norm2.bias, self.input_head.weight, self.input_head.bias, self.output_head.weight,
                self.output_head.bias, self.qnorm.weight, self.qnorm.bias, self.knorm.weight, self.knorm.bias, _bias_table(self.rel_pos_bias)]

    def forward_tokens(self, tok: torch.Tensor, drop=None) -> torch.Tensor:
        """drop: explicit [B] stochastic-depth factors (tests); drawn here in training mode when the block has a rate."""
        if drop is None:
            drop = _factors(self, tok.shape[0], tok.device)            # dim 0 of the reference's (B, n, emb, h, w) input
        return ops.temporal_block(tok, self.num_heads, self.attn_scale, self.stage_params(), drop)

    def forward(self, x):
        """x: (B, N, emb, H, W) -> same."""
        return ops.as_reference_layout(self.forward_tokens(ops.as_tokens(x)))


class AxialAttentionBlock(nn.Module):
    def __init__(self, embed_dim=768, num_heads=12, drop_path=0, layer_scale_init_value=1e-6, bias_type="rel", attn_scale=True,
                 feat_scale=True):
        super().__init__()
        _check_bias_type(bias_type)
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.attn_scale = attn_scale
        self.feat_scale = feat_scale
        self.norm1 = nn.InstanceNorm2d(embed_dim, affine=True)
        self.norm2 = nn.InstanceNorm2d(embed_dim, affine=True)
        ls = layer_scale_init_value > 0           # <= 0: no parameters, as the reference (layers/attention.py:155-168); its forward then fails (:309)
        self.gamma_att = nn.Parameter(layer_scale_init_value * torch.ones((embed_dim)), requires_grad=True) if ls else None
        self.gamma_mlp = nn.Parameter(layer_scale_init_value * torch.ones((embed_dim)), requires_grad=True) if ls else None
        self.input_head = nn.Conv2d(embed_dim, 3 * embed_dim, 1)
        self.output_head = nn.Conv2d(embed_dim, embed_dim, 1)
        self.qnorm = nn.LayerNorm(embed_dim // num_heads)
        self.knorm = nn.LayerNorm(embed_dim // num_heads)
        self.rel_pos_bias = RelativePositionBias(n_heads=num_heads) if bias_type != "none" else _no_bias
        if attn_scale:
            self.attn_scale_factor_x = nn.Parameter(torch.ones((1, num_heads, 1, 1)), requires_grad=True)
            self.attn_scale_factor_y = nn.Parameter(torch.ones((1, num_heads, 1, 1)), requires_grad=True)
        if feat_scale:
            self.low_freq_scalar = nn.Parameter(torch.zeros(embed_dim), requires_grad=True)
            self.high_freq_scalar = nn.Parameter(torch.zeros(embed_dim), requires_grad=True)
        self.drop_path = _make_drop_path(drop_path)
        self.mlp = GeluMLP(embed_dim)
        self.mlp_norm = nn.InstanceNorm2d(embed_dim, affine=True)

    def stage_params(self):
        a, f = self.attn_scale, self.feat_scale
        _require_layer_scale(self.gamma_att)
        return [self.gamma_att, self.gamma_mlp, self.attn_scale_factor_x if a else None, self.attn_scale_factor_y if a else None,
                self.low_freq_scalar if f else None, self.high_freq_scalar if f else None, self.norm1.weight, self.norm1.bias,
                self.norm2.weight, self.norm2.bias, self.input_head.weight, self.input_head.bias, self.output_head.weight,
                self.output_head.bias, self.qnorm.weight, self.qnorm.bias, self.knorm.weight, self.knorm.bias,
                _bias_table(self.rel_pos_bias), self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight,
                self.mlp.fc2.bias, self.mlp_norm.weight, self.mlp_norm.bias]

    def forward_tokens(self, tok: torch.Tensor, drop_att=None, drop_mlp=None) -> torch.Tensor:
        """tok: (B, T, h, w, E); frames are independent, so any leading (B, T) split of B*T frames is equivalent.
        Two independent masks over dim 0 = B*T of the reference's (B*T, emb, h, w) input (attention.py:309,317)."""
        nf = tok.shape[0] * tok.shape[1]
        if drop_att is None:
            drop_att = _factors(self, nf, tok.device)
        if drop_mlp is None:
            drop_mlp = _factors(self, nf, tok.device)
        return ops.spatial_block(tok, self.num_heads, self.attn_scale, self.feat_scale, self.stage_params(), drop_att, drop_mlp)

    def forward(self, x):
        """x: (B, emb, H, W) -> same."""
        tok = ops.as_tokens(x.unsqueeze(1))
        return ops.as_reference_layout(self.forward_tokens(tok))[:, 0]
