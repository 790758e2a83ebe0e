"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

Name-keyed seeded parameter generator: every
``state_dict`` key gets its own ``torch.Generator`` seeded from crc32(key) ^
seed, so the same values can be regenerated at any model size on any box
without shipping weights, and so a renamed / re-ordered parameter shows up as
a parity failure.  Values are O(1)-perturbed (layer-scale ~0.5, attention scale
factors ~1 +- 0.3, frequency scalars ~ +-0.3) because the reference's default
init (gamma = 1e-6, freq scalars = 0) hides block-level errors.

``param_shapes`` restates the reference's parameter inventory
(/root/reference/bubbleformer/models/axial_vit.py:173-215 and the layer ctors
in bubbleformer/layers/{attention,patching,linear_layers,positional_encoding}.py).
"""
import math
import zlib
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch


def param_shapes(*, input_fields: int, output_fields: int, patch_size: int, embed_dim: int, num_heads: int,
                 processor_blocks: int, num_fluid_params: Optional[int] = None, attn_scale: bool = True,
                 feat_scale: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    E, he = embed_dim, num_heads
    d = E // he
    n = int(math.log2(patch_size))
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    cin = input_fields
    for i in range(n):
        cout = E if (i == n - 1 or n == 1) else E // 4
        s[f"embed.in_proj.{3 * i}.weight"] = (cout, cin, 2, 2)
        s[f"embed.in_proj.{3 * i + 1}.weight"] = (cout,)
        s[f"embed.in_proj.{3 * i + 1}.bias"] = (cout,)
        cin = cout
    if num_fluid_params is not None:
        s["film_embed.film_net.0.weight"] = (num_fluid_params,)
        s["film_embed.film_net.0.bias"] = (num_fluid_params,)
        s["film_embed.film_net.1.weight"] = (2 * E, num_fluid_params)
        s["film_embed.film_net.1.bias"] = (2 * E,)
    for b in range(processor_blocks):
        p = f"blocks.{b}.temporal."
        s[p + "gamma"] = (E,)
        if attn_scale:
            s[p + "attn_scale_factor"] = (1, he, 1, 1)
        for nm in ("norm1", "norm2"):
            s[p + nm + ".weight"] = (E,)
            s[p + nm + ".bias"] = (E,)
        s[p + "input_head.weight"] = (3 * E, E, 1, 1)
        s[p + "input_head.bias"] = (3 * E,)
        s[p + "output_head.weight"] = (E, E, 1, 1)
        s[p + "output_head.bias"] = (E,)
        for nm in ("qnorm", "knorm"):
            s[p + nm + ".weight"] = (d,)
            s[p + nm + ".bias"] = (d,)
        s[p + "rel_pos_bias.relative_attention_bias.weight"] = (32, he)
        p = f"blocks.{b}.spatial."
        s[p + "gamma_att"] = (E,)
        s[p + "gamma_mlp"] = (E,)
        if attn_scale:
            s[p + "attn_scale_factor_x"] = (1, he, 1, 1)
            s[p + "attn_scale_factor_y"] = (1, he, 1, 1)
        if feat_scale:
            s[p + "low_freq_scalar"] = (E,)
            s[p + "high_freq_scalar"] = (E,)
        for nm in ("norm1", "norm2"):
            s[p + nm + ".weight"] = (E,)
            s[p + nm + ".bias"] = (E,)
        s[p + "input_head.weight"] = (3 * E, E, 1, 1)
        s[p + "input_head.bias"] = (3 * E,)
        s[p + "output_head.weight"] = (E, E, 1, 1)
        s[p + "output_head.bias"] = (E,)
        for nm in ("qnorm", "knorm"):
            s[p + nm + ".weight"] = (d,)
            s[p + nm + ".bias"] = (d,)
        s[p + "rel_pos_bias.relative_attention_bias.weight"] = (32, he)
        s[p + "mlp.fc1.weight"] = (4 * E, E)
        s[p + "mlp.fc1.bias"] = (4 * E,)
        s[p + "mlp.fc2.weight"] = (E, 4 * E)
        s[p + "mlp.fc2.bias"] = (E,)
        s[p + "mlp_norm.weight"] = (E,)
        s[p + "mlp_norm.bias"] = (E,)
    cin = E
    for i in range(n):
        last = i == n - 1
        cout = output_fields if (last or n == 1) else E // 4
        s[f"debed.out_proj.{3 * i}.weight"] = (cin, cout, 2, 2)
        if not last:
            s[f"debed.out_proj.{3 * i + 1}.weight"] = (cout,)
            s[f"debed.out_proj.{3 * i + 1}.bias"] = (cout,)
        cin = cout
    return s


def _family(key: str, shape: Tuple[int, ...]):
    """(mean, std) for a key."""
    leaf = key.split(".")[-1]
    if "gamma" in leaf:
        return 0.5, 0.1
    if leaf.startswith("attn_scale_factor"):
        return 1.0, 0.3
    if leaf in ("low_freq_scalar", "high_freq_scalar"):
        return 0.0, 0.3
    if "relative_attention_bias" in key:
        return 0.0, 0.5
    if len(shape) == 1:                              # norm weights / all biases
        if leaf == "weight":
            return 1.0, 0.1
        return 0.0, 0.1
    if "out_proj" in key and len(shape) == 4:        # ConvTranspose2d: (in, out, 2, 2); fan_in = in
        return 0.0, 1.0 / math.sqrt(shape[0])
    fan_in = 1
    for v in shape[1:]:
        fan_in *= v
    return 0.0, 1.0 / math.sqrt(fan_in)


def generate(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for key, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        mean, std = _family(key, shape)
        out[key] = (mean + std * torch.randn(shape, generator=g, dtype=torch.float64)).to(dtype)
    return out


FIELD_STATS = ((-2.37, 1.98), (0.0145, 0.081), (-0.07, 0.49), (0.055, 0.77))   # dfun, temperature, velx, vely


def synthetic_clip(B: int, T: int, C: int, H: int, W: int, seed: int, dtype=torch.float32) -> torch.Tensor:
    """Seeded synthetic clip with the per-field statistics of samples/sample_1.hdf5
    (SURVEY.md section 8d).  Field 1 (temperature) is clamped >= 0."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, T, C, H, W), generator=g, dtype=torch.float64)
    for c in range(C):
        mu, sd = FIELD_STATS[c % 4]
        x[:, :, c] = mu + sd * x[:, :, c]
        if c % 4 == 1:
            x[:, :, c].clamp_(min=0.0)
    return x.to(dtype)


def synthetic_fluid_params(B: int, P: int, seed: int, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn((B, P), generator=g, dtype=torch.float64).to(dtype)
