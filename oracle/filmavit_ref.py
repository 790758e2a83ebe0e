"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (plain PyTorch, fp32/fp64, autograd) of the FiLMAViT forward
path of HPCForge/Bubbleformer, written from the maths of the reference in a
token-major ("channels-last") functional form that operates on a flat
``state_dict``-style mapping of tensors.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product (``bubbleformer_amd``) never does.

Parity pin: ``oracle/gen_golden.py`` imports the real reference from
``/root/reference`` (CPU) and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors
(forward, loss, every parameter gradient, dx).

Reference lines each function follows (relative to /root/reference):
  t5_bucket_table / rel_pos_bias .. bubbleformer/layers/positional_encoding.py:73-161
  instance_norm_tokens ........... torch InstanceNorm2d(affine) at layers/attention.py:39-40,153-154,197; patching.py:45,102
  embed .......................... bubbleformer/layers/patching.py:30-59
  film ........................... bubbleformer/layers/linear_layers.py:57-77
  temporal_block ................. bubbleformer/layers/attention.py:66-124
  spatial_block .................. bubbleformer/layers/attention.py:199-319
  debed .......................... bubbleformer/layers/patching.py:87-115
  filmavit_forward ............... bubbleformer/models/axial_vit.py:217-242
  avit_forward ................... bubbleformer/models/axial_vit.py:129-151
  lp_loss ........................ bubbleformer/utils/losses.py:67-94 (as configured at bubbleformer/modules.py:50)
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

EPS = 1e-5


# --------------------------------------------------------------------------- #
# T5 relative-position buckets (integer work)                                  #
# --------------------------------------------------------------------------- #
def t5_bucket_table(max_offset: int = 32, num_buckets: int = 32, max_distance: int = 32) -> np.ndarray:
    """bucket(|n|) for |n| = 0..max_offset-1 on ONE side (0..num_buckets/2-1).

    positional_encoding.py:106-131.  ``compute_bias`` (150-154) does not pass
    ``max_distance`` so the static default 32 applies, not the ctor's 128.
    The log is evaluated in float32 exactly as torch does (tensor.float() /
    python-int, log, / python-float, * python-int, truncation toward zero).
    """
    half = num_buckets // 2
    max_exact = half // 2
    out = np.zeros(max_offset, dtype=np.int64)
    for n in range(max_offset):
        if n < max_exact:
            out[n] = n
        else:
            v = np.log(np.float32(n) / np.float32(max_exact), dtype=np.float32)
            v = np.float32(v / np.float32(math.log(max_distance / max_exact)))
            v = np.float32(v * np.float32(half - max_exact))
            out[n] = min(max_exact + int(v), half - 1)
    return out


def rel_pos_bucket_matrix(L: int, num_buckets: int = 32) -> np.ndarray:
    """[L, L] bucket index for (query i, key j); key after query -> +num_buckets/2."""
    tab = t5_bucket_table(max(L, 1), num_buckets)
    idx = np.zeros((L, L), dtype=np.int64)
    for i in range(L):
        for j in range(L):
            n = i - j                      # = -(memory - context)
            idx[i, j] = (num_buckets // 2 if n < 0 else 0) + tab[abs(n)]
    return idx


def rel_pos_bias(emb_weight: Tensor, L: int) -> Tensor:
    """[heads, L, L] additive score bias from the (num_buckets, heads) embedding."""
    idx = torch.from_numpy(rel_pos_bucket_matrix(L, emb_weight.shape[0])).to(emb_weight.device)
    return emb_weight[idx].permute(2, 0, 1)


# --------------------------------------------------------------------------- #
# small pieces                                                                 #
# --------------------------------------------------------------------------- #
def instance_norm_tokens(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """x: [F, S, C] tokens of one frame on axis 1.  Biased variance, eps 1e-5."""
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    return (x - mu) / torch.sqrt(var + EPS) * w + b


def gelu(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm_last(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + EPS) * w + b


def _hf_attention(q: Tensor, k: Tensor, v: Tensor, bias: Optional[Tensor], scale_he: Optional[Tensor]) -> Tensor:
    """q,k,v: [..., heads, L, d].  softmax(q k^T d^-1/2 + bias) with the
    high-frequency rescale 1/L + (p - 1/L) * s_head  (attention.py:85-101)."""
    d = q.shape[-1]
    L = q.shape[-2]
    s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    if bias is not None:
        s = s + bias
    p = torch.softmax(s, dim=-1)
    if scale_he is not None:
        # attention.py:95 builds 1/L from a float32 ``torch.ones`` whatever the model dtype,
        # so the constant is fl32(1/L) even in an fp64 model.
        inv = (torch.ones((), dtype=torch.float32) / L).to(p.dtype)
        p = inv + (p - inv) * scale_he.reshape(-1, 1, 1)
    return torch.matmul(p, v)


def _stage_count(patch: int) -> int:
    n = int(math.log2(patch))
    assert 2 ** n == patch, "Patch size must be a power of 2"
    return n


# --------------------------------------------------------------------------- #
# patch embed / debed (k2 s2 (transposed) convs as patch GEMMs)                #
# --------------------------------------------------------------------------- #
def embed(sd: SD, pre: str, x: Tensor, patch: int) -> Tensor:
    """x: [F, C, H, W] -> tokens [F, h, w, E] (channels last)."""
    n = _stage_count(patch)
    t = x.permute(0, 2, 3, 1)                                 # [F,H,W,C]
    for i in range(n):
        wconv = sd[f"{pre}in_proj.{3 * i}.weight"]           # [Co, Ci, 2, 2]
        Fr, H, W, C = t.shape
        p = t.reshape(Fr, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4)   # [F,H/2,W/2,Ci,ky,kx]
        t = torch.einsum("fyxckl,ockl->fyxo", p, wconv)
        Co = t.shape[-1]
        t = instance_norm_tokens(t.reshape(Fr, -1, Co), sd[f"{pre}in_proj.{3 * i + 1}.weight"],
                                 sd[f"{pre}in_proj.{3 * i + 1}.bias"]).reshape(Fr, H // 2, W // 2, Co)
        if i != n - 1:
            t = gelu(t)
    return t


def debed(sd: SD, pre: str, t: Tensor, patch: int) -> Tensor:
    """tokens [F, h, w, E] -> [F, Cout, H, W]."""
    n = _stage_count(patch)
    for i in range(n):
        wt = sd[f"{pre}out_proj.{3 * i}.weight"]               # [Ci, Co, 2, 2]
        Fr, h, w, C = t.shape
        o = torch.einsum("fyxc,cokl->fykxlo", t, wt)           # [F,h,ky,w,kx,Co]
        Co = o.shape[-1]
        t = o.reshape(Fr, 2 * h, 2 * w, Co)
        if i != n - 1:
            t = instance_norm_tokens(t.reshape(Fr, -1, Co), sd[f"{pre}out_proj.{3 * i + 1}.weight"],
                                     sd[f"{pre}out_proj.{3 * i + 1}.bias"]).reshape(Fr, 2 * h, 2 * w, Co)
            t = gelu(t)
    return t.permute(0, 3, 1, 2)


def film(sd: SD, pre: str, t: Tensor, cond: Tensor) -> Tensor:
    """t: [B, T, h, w, E]; cond: [B, P].  gamma * x + beta (NOT 1 + gamma)."""
    c = layer_norm_last(cond, sd[f"{pre}film_net.0.weight"], sd[f"{pre}film_net.0.bias"])
    gb = c @ sd[f"{pre}film_net.1.weight"].t() + sd[f"{pre}film_net.1.bias"]
    E = t.shape[-1]
    g, b = gb[:, :E], gb[:, E:]
    return t * g[:, None, None, None, :] + b[:, None, None, None, :]


# --------------------------------------------------------------------------- #
# processor blocks                                                             #
# --------------------------------------------------------------------------- #
def _split_heads(qkv: Tensor, heads: int):
    """[..., 3E] with channel = head*3d + {q|k|v}*d + e  ->  q,k,v [..., heads, d]."""
    d3 = qkv.shape[-1] // heads
    u = qkv.reshape(*qkv.shape[:-1], heads, d3)
    d = d3 // 3
    return u[..., :d], u[..., d:2 * d], u[..., 2 * d:]


def temporal_block(sd: SD, pre: str, t: Tensor, heads: int, attn_scale: bool = True, drop: Optional[Tensor] = None) -> Tensor:
    """t: [B, T, h, w, E] -> same.  Attention along T for every (b, y, x, head).
    drop: optional [B] stochastic-depth factors (0 or 1/keep) -- timm.layers.DropPath's published behaviour (per-sample
    Bernoulli(keep) mask on dim 0, scaled by 1/keep; the package is absent from the reference tree, so this leg is
    parity-UNPINNED and only cross-checks the HIP path against the same explicit masks)."""
    B, T, h, w, E = t.shape
    xn = instance_norm_tokens(t.reshape(B * T, h * w, E), sd[f"{pre}norm1.weight"], sd[f"{pre}norm1.bias"])
    qkv = xn @ sd[f"{pre}input_head.weight"].reshape(3 * E, E).t() + sd[f"{pre}input_head.bias"]
    q, k, v = _split_heads(qkv.reshape(B, T, h, w, 3 * E), heads)       # [B,T,h,w,he,d]
    q = layer_norm_last(q, sd[f"{pre}qnorm.weight"], sd[f"{pre}qnorm.bias"])
    k = layer_norm_last(k, sd[f"{pre}knorm.weight"], sd[f"{pre}knorm.bias"])
    q, k, v = (z.permute(0, 2, 3, 4, 1, 5) for z in (q, k, v))          # [B,h,w,he,T,d]
    bias = rel_pos_bias(sd[f"{pre}rel_pos_bias.relative_attention_bias.weight"], T)
    sc = sd[f"{pre}attn_scale_factor"].reshape(-1) if attn_scale else None
    o = _hf_attention(q, k, v, bias, sc)                                # [B,h,w,he,T,d]
    o = o.permute(0, 4, 1, 2, 3, 5).reshape(B * T, h * w, E)
    on = instance_norm_tokens(o, sd[f"{pre}norm2.weight"], sd[f"{pre}norm2.bias"])
    y = on @ sd[f"{pre}output_head.weight"].reshape(E, E).t() + sd[f"{pre}output_head.bias"]
    br = (y * sd[f"{pre}gamma"]).reshape(B, T, h, w, E)
    if drop is not None:
        br = br * drop.reshape(B, 1, 1, 1, 1)
    return t + br


def spatial_block(sd: SD, pre: str, t: Tensor, heads: int, attn_scale: bool = True, feat_scale: bool = True,
                  drop_att: Optional[Tensor] = None, drop_mlp: Optional[Tensor] = None) -> Tensor:
    """t: [F, h, w, E] -> same.  Axial attention along w and along h (shared
    q/k/v and shared bias table), averaged; feature scaling; MLP + InstanceNorm."""
    Fr, h, w, E = t.shape
    xn = instance_norm_tokens(t.reshape(Fr, h * w, E), sd[f"{pre}norm1.weight"], sd[f"{pre}norm1.bias"])
    qkv = xn @ sd[f"{pre}input_head.weight"].reshape(3 * E, E).t() + sd[f"{pre}input_head.bias"]
    q, k, v = _split_heads(qkv.reshape(Fr, h, w, 3 * E), heads)         # [F,h,w,he,d]
    q = layer_norm_last(q, sd[f"{pre}qnorm.weight"], sd[f"{pre}qnorm.bias"])
    k = layer_norm_last(k, sd[f"{pre}knorm.weight"], sd[f"{pre}knorm.bias"])
    emb = sd[f"{pre}rel_pos_bias.relative_attention_bias.weight"]
    sx = sd[f"{pre}attn_scale_factor_x"].reshape(-1) if attn_scale else None
    sy = sd[f"{pre}attn_scale_factor_y"].reshape(-1) if attn_scale else None
    # along w: sequences (f, y), layout [F,h,he,w,d]
    ox = _hf_attention(*(z.permute(0, 1, 3, 2, 4) for z in (q, k, v)), rel_pos_bias(emb, w), sx)
    ox = ox.permute(0, 1, 3, 2, 4)                                       # [F,h,w,he,d]
    # along h: sequences (f, x), layout [F,w,he,h,d]
    oy = _hf_attention(*(z.permute(0, 2, 3, 1, 4) for z in (q, k, v)), rel_pos_bias(emb, h), sy)
    oy = oy.permute(0, 3, 1, 2, 4)                                       # [F,h,w,he,d]
    o = ((ox + oy) / 2).reshape(Fr, h * w, E)
    on = instance_norm_tokens(o, sd[f"{pre}norm2.weight"], sd[f"{pre}norm2.bias"])
    y = on @ sd[f"{pre}output_head.weight"].reshape(E, E).t() + sd[f"{pre}output_head.bias"]
    if feat_scale:
        m = y.mean(dim=1, keepdim=True)
        y = y + m * sd[f"{pre}low_freq_scalar"] + (y - m) * sd[f"{pre}high_freq_scalar"]
    br = y * sd[f"{pre}gamma_att"]
    if drop_att is not None:
        br = br * drop_att.reshape(Fr, 1, 1)
    x1 = t.reshape(Fr, h * w, E) + br
    hid = gelu(x1 @ sd[f"{pre}mlp.fc1.weight"].t() + sd[f"{pre}mlp.fc1.bias"])
    z = hid @ sd[f"{pre}mlp.fc2.weight"].t() + sd[f"{pre}mlp.fc2.bias"]
    zn = instance_norm_tokens(z, sd[f"{pre}mlp_norm.weight"], sd[f"{pre}mlp_norm.bias"])
    br2 = zn * sd[f"{pre}gamma_mlp"]
    if drop_mlp is not None:
        br2 = br2 * drop_mlp.reshape(Fr, 1, 1)
    return (x1 + br2).reshape(Fr, h, w, E)


# --------------------------------------------------------------------------- #
# whole models                                                                 #
# --------------------------------------------------------------------------- #
def count_blocks(sd: SD) -> int:
    n = 0
    while f"blocks.{n}.temporal.gamma" in sd:
        n += 1
    return n


def _processor(sd: SD, t: Tensor, heads: int, attn_scale: bool, feat_scale: bool) -> Tensor:
    B, T, h, w, E = t.shape
    for i in range(count_blocks(sd)):
        t = temporal_block(sd, f"blocks.{i}.temporal.", t, heads, attn_scale)
        t = spatial_block(sd, f"blocks.{i}.spatial.", t.reshape(B * T, h, w, E), heads, attn_scale, feat_scale)
        t = t.reshape(B, T, h, w, E)
    return t


def filmavit_forward(sd: SD, x: Tensor, fluid_params: Tensor, *, patch_size: int, num_heads: int,
                     attn_scale: bool = True, feat_scale: bool = True) -> Tensor:
    """x: [B,T,C,H,W], fluid_params: [B,P] -> [B,T,Cout,H,W].  drop_path = 0 / eval."""
    B, T, C, H, W = x.shape
    t = embed(sd, "embed.", x.reshape(B * T, C, H, W), patch_size)
    _, h, w, E = t.shape
    t = film(sd, "film_embed.", t.reshape(B, T, h, w, E), fluid_params)
    t = _processor(sd, t, num_heads, attn_scale, feat_scale)
    y = debed(sd, "debed.", t.reshape(B * T, h, w, E), patch_size)
    return y.reshape(B, T, -1, H, W)


def avit_forward(sd: SD, x: Tensor, *, patch_size: int, num_heads: int,
                 attn_scale: bool = True, feat_scale: bool = True) -> Tensor:
    B, T, C, H, W = x.shape
    t = embed(sd, "embed.", x.reshape(B * T, C, H, W), patch_size)
    _, h, w, E = t.shape
    t = _processor(sd, t.reshape(B, T, h, w, E), num_heads, attn_scale, feat_scale)
    y = debed(sd, "debed.", t.reshape(B * T, h, w, E), patch_size)
    return y.reshape(B, T, -1, H, W)


def lp_loss(pred: Tensor, y: Tensor) -> Tensor:
    """Relative L2 over (H,W) per (b,t,c); mean over B, mean over T, sum over C."""
    num = torch.sqrt(((pred - y) ** 2).sum(dim=(-1, -2)))
    den = torch.sqrt((y ** 2).sum(dim=(-1, -2)))
    return (num / den).mean(dim=0).mean(dim=0).sum()


def lion_step(p: Tensor, g: Tensor, m: Tensor, lr: float, beta1: float = 0.9, beta2: float = 0.99, wd: float = 0.0) -> None:
    """Lion single-tensor update, in place.  PARITY UNPINNED: the reference calls the third-party ``lion_pytorch.Lion``
    (bubbleformer/modules.py:139-140; not vendored, version not pinned in env/*.yaml); this restates the published rule
    (Chen et al. 2023, Algorithm 2; lion_pytorch ``update_fn``): decoupled decay, sign of the beta1-interpolated momentum,
    then the beta2 momentum update."""
    p.mul_(1.0 - lr * wd)
    p.add_(torch.sign(m * beta1 + g * (1.0 - beta1)), alpha=-lr)
    m.mul_(beta2).add_(g, alpha=1.0 - beta2)


def eikonal_loss(phi: Tensor) -> Tensor:
    """utils/losses.py:5-15: mean of (|grad phi| - 1)^2, torch.gradient semantics (spacing 1/32, edge_order 1) written out:
    central differences in the interior, one-sided first differences on the border, along H (dim -2) and W (dim -1)."""
    dx = 1.0 / 32

    def grad(a: Tensor, dim: int) -> Tensor:
        a = a.movedim(dim, -1)
        g = torch.empty_like(a)
        g[..., 1:-1] = (a[..., 2:] - a[..., :-2]) / (2 * dx)
        g[..., 0] = (a[..., 1] - a[..., 0]) / dx
        g[..., -1] = (a[..., -1] - a[..., -2]) / dx
        return g.movedim(-1, dim)
    gy, gx = grad(phi, -2), grad(phi, -1)
    return ((torch.sqrt(gy ** 2 + gx ** 2) - 1.0) ** 2).mean()


def eikonal_l1_per_frame(phi: Tensor) -> Tensor:
    """scripts/inference_autoregressive.ipynb cell 8 (`get_eikonal_loss`): phi (T, H, W) -> (T,) mean over the frame of
    | |grad phi| - 1 |, central differences at dx = 1/32 in the interior, the border taking its neighbour's value (replicate pad)."""
    dx = 1.0 / 32
    H, W = phi.shape[-2:]
    xi = torch.arange(W).clamp(1, W - 2)
    yi = torch.arange(H).clamp(1, H - 2)
    gx = (phi[:, :, xi + 1] - phi[:, :, xi - 1]) / (2 * dx)
    gy = (phi[:, yi + 1, :] - phi[:, yi - 1, :]) / (2 * dx)
    return (torch.sqrt(gx ** 2 + gy ** 2) - 1.0).abs().mean(dim=(1, 2))


def heatflux(dfun, temp, heater_temp: float):
    """utils/heatflux.py:3-38 (numpy arrays (T, 512, 512)): bottom-row flux of the liquid cells over the heater |x| <= 5 on the
    16 x 16 domain at dx = 1/32, 0.054 * (T_heater - T) / (dx * lc) with lc = 0.0007; returns (mean, max) over frames."""
    import numpy as np
    dx, lc = 1 / 32, 0.0007
    xc = -8 + (np.arange(512) + 0.5) * dx
    mask = (xc >= -5.0) & (xc <= 5.0)
    row = (mask[None, :] & (dfun[:, 0, :] < 0)).astype(float) * (heater_temp - temp[:, 0, :])
    fl = (0.054 * row / (dx * lc)).mean(axis=1)
    return float(np.mean(fl)), float(np.max(fl))


def cosine_warmup_lr(step: int, base_lr: float, warmup_iters: int, max_iters: int, eta_min: float = 0.0) -> float:
    """Learning rate in effect for optimizer step number ``step`` (0-based) under the reference's CosineWarmupLR
    (bubbleformer/utils/lr_schedulers.py:4-31: SequentialLR[LambdaLR(step / warmup_iters), CosineAnnealingLR(T_max=max_iters,
    eta_min)], milestone warmup_iters, stepped once per batch, modules.py:153-171).  Closed form of what torch's schedulers
    produce: linear ramp from 0, then the cosine restarted at the milestone (its own step counter starts at 0 there)."""
    if step < warmup_iters:
        return base_lr * step / warmup_iters
    t = step - warmup_iters
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * t / max_iters)) / 2.0


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, wd: float = 1e-2) -> None:
    """torch.optim.AdamW single-tensor update (decoupled decay), in place."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
