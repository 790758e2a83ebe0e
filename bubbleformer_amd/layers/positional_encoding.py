"""Relative position bias tables (mirror of bubbleformer/layers/positional_encoding.py).

``RelativePositionBias`` only OWNS the (num_buckets, heads) embedding; the attention kernels index it directly
with the T5 bucket of (query - key) (csrc/attn.hip: t5_bucket).  ``forward`` materialises the (1, heads, q, k)
tensor for callers that want it (host-side table lookup of the same integer buckets).
"""
import torch
import torch.nn as nn

# one-sided bucket of |offset| for num_buckets=32 (16 per side, 8 exact), max_distance=32 -- the effective value,
# because compute_bias() does not forward the constructor's max_distance (positional_encoding.py:150-154,81).
_T5_ONE_SIDED = (0, 1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 12, 12, 13, 13, 13, 14, 14, 14, 14, 15, 15, 15, 15, 15)


def t5_bucket(query_minus_key: int) -> int:
    n = query_minus_key
    a = -n if n < 0 else n
    b = _T5_ONE_SIDED[a] if a < 32 else 15
    return b + (16 if n < 0 else 0)


class RelativePositionBias(nn.Module):
    def __init__(self, bidirectional: bool = True, num_buckets: int = 32, max_distance: int = 128, n_heads: int = 2):
        super().__init__()
        if not bidirectional or num_buckets != 32:
            raise NotImplementedError("the HIP attention kernels implement the bidirectional 32-bucket table the reference configures")
        self.bidirectional = bidirectional
        self.num_buckets = num_buckets
        self.max_distance = max_distance
        self.n_heads = n_heads
        self.relative_attention_bias = nn.Embedding(self.num_buckets, self.n_heads)

    def bucket_matrix(self, qlen: int, klen: int) -> torch.Tensor:
        return torch.tensor([[t5_bucket(i - j) for j in range(klen)] for i in range(qlen)], dtype=torch.long)

    def forward(self, qlen: int, klen: int) -> torch.Tensor:
        idx = self.bucket_matrix(qlen, klen).to(self.relative_attention_bias.weight.device)
        return self.relative_attention_bias(idx).permute(2, 0, 1).unsqueeze(0)


class ContinuousPositionBias1D(nn.Module):
    """API stub: ``bias_type="continuous"`` is not selected by any reference config."""

    def __init__(self, n_heads: int):
        super().__init__()
        raise NotImplementedError("ContinuousPositionBias1D is outside the FiLMAViT hot path (no reference config uses it)")
