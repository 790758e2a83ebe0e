"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from oracle import filmavit_ref as R
from oracle import weights as W
from oracle.gen_golden import VARIANTS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_variant(name):
    spec = VARIANTS[name]
    z = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    return spec, z


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


# Structurally-zero gradient families: softmax is invariant to a
# key-constant shift (knorm.bias) and InstanceNorm removes a per-channel constant
# (mlp.fc2.bias, and output_head.bias when followed directly by mean-removal is NOT
# the case here).  Their true gradient is exactly 0; compare with an absolute bound.
def structurally_zero(key):
    return key.endswith("knorm.bias") or key.endswith("mlp.fc2.bias")


def oracle_run(name, dtype):
    """Run the oracle restatement on a golden variant; returns (pred, loss, dx, grads)."""
    spec, z = load_variant(name)
    cfg = spec["cfg"]
    shapes = W.param_shapes(**cfg)
    sd = {k: v.requires_grad_(True) for k, v in W.generate(shapes, seed=spec["seed"], dtype=dtype).items()}
    x = W.synthetic_clip(spec["B"], spec["T"], cfg["input_fields"], spec["H"], spec["W"], 100 + spec["seed"], dtype)
    y = W.synthetic_clip(spec["B"], spec["T"], cfg["output_fields"], spec["H"], spec["W"], 200 + spec["seed"], dtype)
    x.requires_grad_(True)
    kw = dict(patch_size=cfg["patch_size"], num_heads=cfg["num_heads"], attn_scale=cfg.get("attn_scale", True),
              feat_scale=cfg.get("feat_scale", True))
    if spec["model"] == "filmavit":
        cond = W.synthetic_fluid_params(spec["B"], cfg["num_fluid_params"], 300 + spec["seed"], dtype)
        pred = R.filmavit_forward(sd, x, cond, **kw)
    else:
        pred = R.avit_forward(sd, x, **kw)
    loss = R.lp_loss(pred, y)
    loss.backward()
    return pred.detach(), loss.detach(), x.grad.detach(), {k: v.grad.detach() for k, v in sd.items()}


def fullsize_errors(name, pred, loss, dx, grads):
    """(pred, loss, dx, grads) of any implementation against tests/golden/fullsize_<name>.npz (the REFERENCE in fp64 at a BASELINE
    size, oracle/gen_golden.py: gen_fullsize).  Returns relative errors: loss, per-(b,t,c) prediction moments, sampled prediction /
    dx entries, per-(b,t,c) dx norms, and per parameter the gradient norm and sampled entries (structurally-zero families apart)."""
    from oracle.gen_golden import fullsize_stats
    z = np.load(os.path.join(GOLDEN, f"fullsize_{name}.npz"))
    got = fullsize_stats(pred.detach().cpu(), float(torch.as_tensor(loss).detach()), dx.detach().cpu(), {k: g.detach().cpu() for k, g in grads.items()})
    names = sorted(grads)
    live = np.array([not structurally_zero(k) for k in names])
    per = np.repeat(live, [min(16, grads[k].numel()) for k in names])
    gscale = float(z["grad_l2"].max())
    e = {
        "loss": abs(got["loss"] - z["loss"]) / abs(z["loss"]),
        "pred_samples": rel_l2(got["pred_samples"], z["pred_samples"]),
        "pred_mean": float(np.abs(got["pred_mean"] - z["pred_mean"]).max() / np.abs(z["pred_l2"]).max()),
        "pred_std": float(np.abs(got["pred_std"] / z["pred_std"] - 1).max()),
        "pred_l2": float(np.abs(got["pred_l2"] / z["pred_l2"] - 1).max()),
        "dx_samples": rel_l2(got["dx_samples"], z["dx_samples"]),
        "dx_l2": float(np.abs(got["dx_l2"] / z["dx_l2"] - 1).max()),
        "grad_l2_worst": float(np.abs(got["grad_l2"][live] / z["grad_l2"][live] - 1).max()),
        "grad_l2_all": rel_l2(got["grad_l2"][live], z["grad_l2"][live]),
        "grad_samples": rel_l2(got["grad_samples"][per], z["grad_samples"][per]),
        "grad_zero_families": float(got["grad_l2"][~live].max() / gscale) if (~live).any() else 0.0,
    }
    return e
