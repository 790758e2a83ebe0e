"""Learning-rate schedule of the reference training step (bubbleformer/utils/lr_schedulers.py:4-31, configured at
bubbleformer/modules.py:153-171 with ``interval="step"``): linear warm-up ``step / warmup_iters`` followed by cosine annealing
to ``eta_min`` over ``max_iters``.

The reference builds it from torch's ``SequentialLR`` around a ``torch.optim`` optimizer; the native training step has fused
optimizer kernels over a flat buffer and no ``torch.optim`` object, so this class keeps the reference's name and constructor
arguments (minus the optimizer) and yields the same sequence of learning rates, one per optimizer step."""
import math


class CosineWarmupLR:
    """Same learning-rate sequence as the reference's ``CosineWarmupLR(SequentialLR)`` (utils/lr_schedulers.py:4-31), stepped per batch
    (modules.py:164-171).  The first argument is what the reference passes -- a ``torch.optim.Optimizer`` (anything with
    ``param_groups``): every group's ``lr`` is then written on construction and on each ``step()``, as torch schedulers do, and
    ``get_last_lr()`` has one entry per group -- or a plain base learning rate (the native ``trainer.TrainStep`` keeps its own fused
    optimizer state and only needs the number)."""

    def __init__(self, optimizer_or_lr, warmup_iters: int, max_iters: int, eta_min: float = 0.0, last_epoch: int = -1):
        self.optimizer = optimizer_or_lr if hasattr(optimizer_or_lr, "param_groups") else None
        if self.optimizer is not None:
            for g in self.optimizer.param_groups:
                g.setdefault("initial_lr", g["lr"])
            self.base_lrs = [float(g["initial_lr"]) for g in self.optimizer.param_groups]
        else:
            self.base_lrs = [float(optimizer_or_lr)]
        self.base_lr = self.base_lrs[0]
        self.warmup_iters, self.max_iters, self.eta_min = int(warmup_iters), int(max_iters), float(eta_min)
        self.last_epoch = last_epoch + 1          # like torch: constructing the scheduler performs the initial step
        self._write()

    def _lr(self, base: float, step: int) -> float:
        if step < self.warmup_iters:
            return base * step / self.warmup_iters
        t = step - self.warmup_iters
        return self.eta_min + (base - self.eta_min) * (1.0 + math.cos(math.pi * t / self.max_iters)) / 2.0

    def lr_at(self, step: int) -> float:
        return self._lr(self.base_lr, step)

    def _write(self) -> None:
        if self.optimizer is not None:
            for g, b in zip(self.optimizer.param_groups, self.base_lrs):
                g["lr"] = self._lr(b, self.last_epoch)

    def get_last_lr(self):
        return [self._lr(b, self.last_epoch) for b in self.base_lrs]

    def step(self) -> None:
        self.last_epoch += 1
        self._write()

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
        self._write()
