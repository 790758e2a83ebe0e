"""Checkpoints in the layout the reference's Lightning runs write and its inference script reads
(bubbleformer/modules.py:57 `save_hyperparameters()`, scripts/inference.py:205-226): a dict with

  "state_dict"        parameter tensors keyed "model.<name>" (the LightningModule holds the network as `self.model`)
  "hyper_parameters"  the module's constructor arguments, incl. "normalization_constants" = (diff_terms, div_terms)
  "global_step"       optimizer steps taken
plus, for resuming the native training step, "optimizer_states" / "lr_schedulers" entries holding the flat AdamW / Lion moments
and the scheduler position (Lightning stores its torch.optim state dicts under those keys; ours are flat buffers, so a checkpoint
written here resumes here, while its "state_dict" loads anywhere the reference's does)."""
import os
from collections import OrderedDict
from typing import Optional

import torch

PREFIX = "model."


def to_reference_state_dict(model: torch.nn.Module) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((PREFIX + k, v.detach().to("cpu", copy=True)) for k, v in model.state_dict().items())


def from_reference_state_dict(sd) -> "OrderedDict[str, torch.Tensor]":
    """Strip Lightning's "model." prefix exactly as scripts/inference.py:222-225 does (`key[6:]`)."""
    out = OrderedDict()
    for k, v in sd.items():
        if not k.startswith(PREFIX):
            raise KeyError(f"checkpoint key {k!r} does not start with {PREFIX!r}")
        out[k[len(PREFIX):]] = v
    return out


def save_checkpoint(path: str, model: torch.nn.Module, hyper_parameters: Optional[dict] = None, normalization_constants=None,
                    train_step=None, global_step: Optional[int] = None, epoch: Optional[int] = None) -> None:
    """One torch.save into `path + ".tmp"`, then os.replace: a kill between two writes (the reference's use case is SLURM pre-emption,
    scripts/train.py:36-67) can never leave a half-written or epoch-less file behind."""
    hp = dict(hyper_parameters or {})
    if normalization_constants is not None:
        hp["normalization_constants"] = normalization_constants
    ckpt = {"state_dict": to_reference_state_dict(model), "hyper_parameters": hp,
            "global_step": int(global_step if global_step is not None else (train_step.step_no if train_step is not None else 0))}
    if train_step is not None:
        ckpt["optimizer_states"] = [{"name": train_step.optimizer, "step": train_step.step_no, "m": train_step.m.detach().cpu(),
                                     "v": None if train_step.v is None else train_step.v.detach().cpu()}]
        ckpt["lr_schedulers"] = [train_step.scheduler.state_dict()] if train_step.scheduler is not None else []
    if epoch is not None:
        ckpt["epoch"] = int(epoch)
    tmp = path + ".tmp"
    torch.save(ckpt, tmp)
    os.replace(tmp, path)


def load_checkpoint(path: str, model: torch.nn.Module, train_step=None, map_location="cpu") -> dict:
    """Loads the weights (into `model`, in place, so a FlatParams re-homing stays valid) and, if given, the training-step state."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = from_reference_state_dict(ckpt["state_dict"])
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    if missing or unexpected:
        raise KeyError(f"checkpoint / model mismatch: missing {missing[:3]}..., unexpected {unexpected[:3]}...")
    with torch.no_grad():
        for k, t in own.items():
            t.copy_(sd[k])
    if train_step is not None and ckpt.get("optimizer_states"):
        st = ckpt["optimizer_states"][0]
        if st["name"] != train_step.optimizer:
            raise ValueError(f"checkpoint optimizer {st['name']!r} != {train_step.optimizer!r}")
        train_step.step_no = int(st["step"])
        train_step.m.copy_(st["m"])
        if train_step.v is not None:
            train_step.v.copy_(st["v"])
        if train_step.scheduler is not None and ckpt.get("lr_schedulers"):
            train_step.scheduler.load_state_dict(ckpt["lr_schedulers"][0])
    from .. import ops
    ops._weights_changed()                    # prepared inference weights (ops.trunk_eval) are re-made from the loaded parameters
    if train_step is not None and hasattr(train_step, "sync_from_rank0"):
        train_step.sync_from_rank0()          # data parallel: every replica continues from rank 0's weights, moments and step count
    return ckpt
