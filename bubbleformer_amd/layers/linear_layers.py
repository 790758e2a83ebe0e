"""Parameter containers mirroring bubbleformer/layers/linear_layers.py.

``GeluMLP`` and ``FiLMMLP`` are consumed by the fused stages (the axial block's MLP GEMM pair, and the patch
embed whose last InstanceNorm absorbs the FiLM scale/shift); they hold the parameters under the reference's names.
Both also run on their own (``ops.gelu_mlp`` / ``ops.film``: native kernels, forward and backward).
"""
import torch
import torch.nn as nn


class GeluMLP(nn.Module):
    def __init__(self, hidden_dim, exp_factor=4.0):
        super().__init__()
        self.fc1 = nn.Linear(hidden_dim, int(hidden_dim * exp_factor))
        self.fc2 = nn.Linear(int(hidden_dim * exp_factor), hidden_dim)
        self.act = nn.GELU()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """fc2(GELU(fc1(x))) on the last dimension (linear_layers.py:18-25), as native GEMMs with the bias / GELU / gelu' epilogues.
        Inside AxialAttentionBlock the pair runs as part of the fused stage (csrc/model.hip: bf_spatial_fwd) instead."""
        from .. import ops
        return ops.gelu_mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)


class SirenMLP(nn.Module):
    """API stub: unused by every model in the reference."""

    def __init__(self, hidden_dim, w0=1.0):
        super().__init__()
        self.fc = nn.Linear(hidden_dim, hidden_dim)
        self.w0 = w0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("SirenMLP is outside the FiLMAViT hot path")


class FiLMMLP(nn.Module):
    def __init__(self, param_dim, embed_dim):
        super().__init__()
        self.film_net = nn.Sequential(
            nn.LayerNorm(param_dim),
            nn.Linear(param_dim, embed_dim * 2),
        )

    def film_params(self):
        return (self.film_net[0].weight, self.film_net[0].bias, self.film_net[1].weight, self.film_net[1].bias)

    def forward(self, x: torch.Tensor, cond) -> torch.Tensor:
        """gamma * x + beta with (gamma, beta) = film_net(cond).chunk(2), x (B, T, C, h, w) (linear_layers.py:63-77), on native kernels.
        In the model the modulation rides inside the fused patch embed (csrc/model.hip: bf_embed_fwd) instead."""
        from .. import ops
        return ops.film(x, cond, *self.film_params())
