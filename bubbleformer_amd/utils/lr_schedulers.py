"""Learning-rate schedule of the reference training step (bubbleformer/utils/lr_schedulers.py:4-31, configured at
bubbleformer/modules.py:153-171 with ``interval="step"``): linear warm-up ``step / warmup_iters`` followed by cosine annealing
to ``eta_min`` over ``max_iters``.

The reference builds it from torch's ``SequentialLR`` around a ``torch.optim`` optimizer; the native training step has fused
optimizer kernels over a flat buffer and no ``torch.optim`` object, so this class keeps the reference's name and constructor
arguments (minus the optimizer) and yields the same sequence of learning rates, one per optimizer step."""
import math


class CosineWarmupLR:
    def __init__(self, base_lr: float, warmup_iters: int, max_iters: int, eta_min: float = 0.0, last_epoch: int = -1):
        self.base_lr, self.warmup_iters, self.max_iters, self.eta_min = float(base_lr), int(warmup_iters), int(max_iters), float(eta_min)
        self.last_epoch = last_epoch + 1          # like torch: constructing the scheduler performs the initial step

    def lr_at(self, step: int) -> float:
        if step < self.warmup_iters:
            return self.base_lr * step / self.warmup_iters
        t = step - self.warmup_iters
        return self.eta_min + (self.base_lr - self.eta_min) * (1.0 + math.cos(math.pi * t / self.max_iters)) / 2.0

    def get_last_lr(self):
        return [self.lr_at(self.last_epoch)]

    def step(self) -> None:
        self.last_epoch += 1

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
