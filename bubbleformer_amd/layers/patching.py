"""HMLPEmbed / HMLPDebed (mirror of bubbleformer/layers/patching.py) on the HIP stage kernels."""
import math

import torch
import torch.nn as nn

from .. import ops


def _stages(patch_size: int) -> int:
    n = int(math.log2(patch_size))
    assert (n - math.log2(patch_size)) == 0, "Patch size must be a power of 2"
    return n


class HMLPEmbed(nn.Module):
    """log2(P) stages of Conv2d(k2,s2,no bias) + InstanceNorm2d(affine) + GELU (none after the last)."""

    def __init__(self, patch_size: int = 16, in_channels: int = 3, embed_dim: int = 768, compute_dtype=None):
        super().__init__()
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.embed_dim = embed_dim
        self.compute_dtype = compute_dtype
        n = _stages(patch_size)
        layers, cin = [], in_channels
        for i in range(n):
            last = i == n - 1
            cout = embed_dim if (last or n == 1) else embed_dim // 4
            layers.append(nn.Conv2d(cin, cout, kernel_size=2, stride=2, bias=False))
            layers.append(nn.InstanceNorm2d(cout, affine=True))
            if not last:
                layers.append(nn.GELU())
            cin = cout
        self.in_proj = nn.Sequential(*layers)
        self.num_stages = n

    def stage_params(self):
        conv = [self.in_proj[3 * i].weight for i in range(self.num_stages)]
        inw = [self.in_proj[3 * i + 1].weight for i in range(self.num_stages)]
        inb = [self.in_proj[3 * i + 1].bias for i in range(self.num_stages)]
        return conv, inw, inb

    def tokens(self, x5: torch.Tensor, fluid=None, film_params=(), compute_dtype=None) -> torch.Tensor:
        """(B, T, C, H, W) -> (B, T, h, w, E) tokens, optionally FiLM-conditioned."""
        conv, inw, inb = self.stage_params()
        dt = compute_dtype or self.compute_dtype or torch.float32
        return ops.embed(x5, fluid, dt, self.patch_size, self.embed_dim, conv, inw, inb, film_params)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, C, H, W) -> (B, E, h, w) (logical view of token-major memory)."""
        tok = self.tokens(x.unsqueeze(1))
        return ops.as_reference_layout(tok)[:, 0]


class HMLPDebed(nn.Module):
    """log2(P) stages of ConvTranspose2d(k2,s2,no bias), InstanceNorm2d + GELU after all but the last."""

    def __init__(self, patch_size: int = 16, out_channels: int = 3, embed_dim: int = 768):
        super().__init__()
        self.patch_size = patch_size
        self.out_channels = out_channels
        self.embed_dim = embed_dim
        n = _stages(patch_size)
        layers, cin = [], embed_dim
        for i in range(n):
            last = i == n - 1
            cout = out_channels if (last or n == 1) else embed_dim // 4
            layers.append(nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2, bias=False))
            if not last:
                layers.append(nn.InstanceNorm2d(cout, affine=True))
                layers.append(nn.GELU())
            cin = cout
        self.out_proj = nn.Sequential(*layers)
        self.num_stages = n

    def stage_params(self):
        n = self.num_stages
        conv = [self.out_proj[3 * i].weight for i in range(n)]
        inw = [self.out_proj[3 * i + 1].weight for i in range(n - 1)]
        inb = [self.out_proj[3 * i + 1].bias for i in range(n - 1)]
        return conv, inw, inb

    def from_tokens(self, tok: torch.Tensor) -> torch.Tensor:
        """(B, T, h, w, E) -> (B, T, C, H, W) fp32."""
        return ops.debed(tok, self.patch_size, self.out_channels, *self.stage_params())

    def loss_from_tokens(self, tok: torch.Tensor, target: torch.Tensor):
        """Fused debed + relative-L2 loss: returns (loss, prediction)."""
        return ops.debed_with_loss(tok, target, self.patch_size, self.out_channels, *self.stage_params())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, E, h, w) -> (B, C, H, W)."""
        tok = ops.as_tokens(x.unsqueeze(1))
        return self.from_tokens(tok)[:, 0]
