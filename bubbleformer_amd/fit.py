"""The epoch loop the reference's `Trainer.fit(ForecastModule, ...)` runs (scripts/train.py:157-171, bubbleformer/modules.py:76-171),
driving the native training step from device-resident clips.

What is kept from the reference's run configuration:
  * one optimizer step per batch, `CosineWarmupLR` stepped per batch (``interval="step"``, modules.py:153-171) with
    ``max_iters = trainer.estimated_stepping_batches`` = max_epochs x batches per epoch (modules.py:63-68);
  * ``limit_train_batches`` / ``limit_val_batches`` (train.py:167-168: 1000 / 25): an epoch is at most that many batches of a
    freshly shuffled pass over the dataset; validation runs after every epoch, unshuffled, no sanity pass (train.py:169);
  * the last, partial batch is kept (DataLoader's ``drop_last=False``);
  * a checkpoint in the Lightning layout after every epoch (utils/checkpoint.py), from which a run resumes at the next epoch;
  * seeding: the shuffle of epoch e on every rank is ``randperm`` under ``seed + e`` and rank r takes ``[r::world]`` of it after
    padding to a multiple of the world size -- `torch.utils.data.DistributedSampler`, which Lightning injects under DDP
    (SURVEY.md section 8e); default seed 42 (config/default.yaml:2).
Logging back-ends (CSV / wandb), SLURM pre-emption and the model summary are the reference's control plane and are not rebuilt;
`log` receives one dict per step / validation instead.
"""
import math
from typing import Callable, Dict, List, Optional

import torch

from .trainer import TrainStep
from .utils.checkpoint import load_checkpoint, save_checkpoint
from .utils.lr_schedulers import CosineWarmupLR


def epoch_indices(n: int, epoch: int, seed: int = 42, shuffle: bool = True, rank: int = 0, world: int = 1) -> List[int]:
    """Sample order of one rank for one epoch (DistributedSampler semantics, drop_last=False)."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    if world > 1:
        total = math.ceil(n / world) * world
        order = (order + order[:total - n])[:total] if n > 0 else order
        order = order[rank:total:world]
    return order


def batches(order: List[int], batch_size: int, limit: Optional[int]) -> List[List[int]]:
    out = [order[i:i + batch_size] for i in range(0, len(order), batch_size)]
    return out if limit is None else out[:int(limit)]


def fit(model: torch.nn.Module, train_set, val_set=None, *, batch_size: int, max_epochs: int, optimizer: str = "lion", lr: float = 5e-5,
        weight_decay: float = 0.1, warmup_iters: Optional[int] = 1000, eta_min: float = 1e-6, limit_train_batches: Optional[int] = 1000,
        limit_val_batches: Optional[int] = 25, seed: int = 42, rank: int = 0, world: int = 1, checkpoint_path: Optional[str] = None,
        resume_from: Optional[str] = None, hyper_parameters: Optional[dict] = None, log: Optional[Callable[[Dict], None]] = None) -> Dict:
    """Trains `model` (a bubbleformer_amd model on the GPU) on `train_set` (data.BubbleForecast, already normalised).  Defaults are the
    reference's: Lion lr 5e-5 wd 0.1 (config/optim_cfg/lion.yaml), cosine schedule with 1000 warm-up steps to 1e-6
    (config/scheduler_cfg/cosine_warmup.yaml).  ``warmup_iters=None`` runs at a constant learning rate.  Returns the history."""
    dev = next(model.parameters()).device
    store = train_set.device_store(dev)
    vstore = val_set.device_store(dev) if val_set is not None else None
    conditioned = getattr(train_set, "return_fluid_params", False)
    per_epoch = len(batches(epoch_indices(len(train_set), 0, seed, True, rank, world), batch_size, limit_train_batches))
    sched = CosineWarmupLR(lr, warmup_iters, max_epochs * per_epoch, eta_min) if warmup_iters is not None else None
    step = TrainStep(model, lr=lr, weight_decay=weight_decay, optimizer=optimizer, scheduler=sched)
    norm = (train_set.diff_terms, train_set.div_terms)
    hist: Dict[str, list] = {"train_loss": [], "lr": [], "val_loss": [], "epoch_train_loss": []}
    first_epoch = 0
    if resume_from is not None:
        ck = load_checkpoint(resume_from, model, step)
        # an epoch-less file (written by an older version, or by save_checkpoint outside fit) resumes at the epoch its step count implies
        first_epoch = int(ck["epoch"]) + 1 if "epoch" in ck else int(ck.get("global_step", 0)) // max(per_epoch, 1)
    for epoch in range(first_epoch, max_epochs):
        model.train()
        losses = []
        for bi, idx in enumerate(batches(epoch_indices(len(train_set), epoch, seed, True, rank, world), batch_size, limit_train_batches)):
            got = store.gather(idx)
            x, y, c = (got[0], got[1], got[2]) if conditioned else (got[0], got[1], None)
            cur_lr = sched.get_last_lr()[0] if sched is not None else lr
            loss = step(x, c, y)
            losses.append(loss)
            hist["lr"].append(cur_lr)
            if log is not None:
                log({"epoch": epoch, "batch_idx": bi, "global_step": step.step_no, "train_loss": loss, "learning_rate": cur_lr})
        ep = torch.stack(losses).float()
        hist["train_loss"].extend(ep.tolist())
        hist["epoch_train_loss"].append(float(ep.mean()))
        if vstore is not None:
            hist["val_loss"].append(validate(model, val_set, vstore, batch_size, limit_val_batches, rank, world))
            if log is not None:
                log({"epoch": epoch, "val_loss": hist["val_loss"][-1]})
        if checkpoint_path is not None and rank == 0:
            save_checkpoint(checkpoint_path, model, hyper_parameters, norm, step, epoch=epoch)       # one atomic write
    return hist


@torch.no_grad()
def validate(model, val_set, vstore, batch_size: int, limit_val_batches: Optional[int], rank: int = 0, world: int = 1) -> float:
    """Mean over the (limited) validation batches of the training criterion, as `validation_step` logs it on epoch end."""
    was_training = model.training
    model.eval()
    conditioned = getattr(val_set, "return_fluid_params", False)
    tot, n = 0.0, 0
    for idx in batches(epoch_indices(len(val_set), 0, 0, False, rank, world), batch_size, limit_val_batches):
        got = vstore.gather(idx)
        loss, _ = model.forward_loss(got[0], got[2], got[1]) if conditioned else model.forward_loss(got[0], got[1])
        tot += float(loss)
        n += 1
    model.train(was_training)
    return tot / max(n, 1)
