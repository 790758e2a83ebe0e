"""Data-parallel training step for the native FiLMAViT path.

* ``FlatParams``   -- re-homes every parameter (and its gradient) of a module into ONE fp32 buffer each, in
                      registration order, so the optimizer is a single fused kernel (csrc/patch.hip: adamw_kernel)
                      and gradient buckets are contiguous slices.
* ``BucketReducer``-- the data-parallel exchange (reference: Lightning's ``strategy="ddp"``, scripts/train.py:158-172):
                      mean all-reduce of gradient buckets over ``torch.distributed`` (backend "nccl" = RCCL over xGMI),
                      issued per bucket as soon as backward has produced every gradient in it (debed first, then
                      blocks N-1 .. 0, then embed), overlapped with the rest of backward.  One process per GPU.
* ``TrainStep``    -- forward (+ fused relative-L2 loss) -> backward -> bucket wait -> fused AdamW.
Device agnostic where it can be (the reducer and the flat views are tested on CPU with gloo); the model itself
only runs on the GPU.
"""
from typing import List, Optional, Sequence

import os

import torch
import torch.distributed as dist
import torch.nn as nn


class FlatParams:
    def __init__(self, module: nn.Module, align: int = 64):
        params = [p for p in module.parameters() if p.requires_grad]
        self.params = params
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        dev = params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets, self.numel = offs, n
        for p, o in zip(params, offs):
            v = self.flat[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()


class BucketReducer:
    """Mean all-reduce of contiguous slices of a flat gradient buffer, launched from post-accumulate hooks."""

    def __init__(self, flat: FlatParams, bucket_of: Sequence[int], group=None, use_hooks: bool = True, bucket_dtype=None):
        """use_hooks=False: buckets are completed by stage_ready() only (TrainStep's direct-gradient mode -- autograd runs the
        post-accumulate hooks even for the ``None`` gradients that mode returns, which would count every bucket twice).
        bucket_dtype=torch.bfloat16 (or BF_GRAD_BUCKET_DTYPE=bf16): a bucket travels as a bf16 copy -- half the bytes per xGMI link
        (57.8 MB instead of 115.6 MB per step for FiLMAViT-small) for one cast each way; the sum of `world`
        bf16-rounded addends carries a relative error of ~2^-9 per element, the master gradients and the optimizer stay fp32."""
        self.flat, self.group = flat, group
        if bucket_dtype is None and os.environ.get("BF_GRAD_BUCKET_DTYPE", "").lower() in ("bf16", "bfloat16"):
            bucket_dtype = torch.bfloat16
        self.bucket_dtype = bucket_dtype
        self.launch_log = []      # bucket ids in launch order (this step); tests assert the reverse-forward order and one collective per bucket
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        nb = max(bucket_of) + 1
        self.lo = [None] * nb
        self.hi = [None] * nb
        self.count = [0] * nb
        for p, o, b in zip(flat.params, flat.offsets, bucket_of):
            self.lo[b] = o if self.lo[b] is None else min(self.lo[b], o)
            self.hi[b] = o + p.numel() if self.hi[b] is None else max(self.hi[b], o + p.numel())
            self.count[b] += 1
        self.pending = [0] * nb
        self.handles = []
        self.held = None          # direct-gradient mode: a complete bucket waits until the NEXT one is complete (see stage_ready)
        self.enabled = self.world > 1
        self.done = set()         # buckets reduced in this step
        if self.enabled and use_hooks:
            for p, b in zip(flat.params, bucket_of):
                p.register_post_accumulate_grad_hook(self._make_hook(b))

        self.bucket_of_ptr = {p.data_ptr(): b for p, b in zip(flat.params, bucket_of)}

    def _launch(self, b):
        self.done.add(b)
        self.launch_log.append(b)
        g = self.flat.grad[self.lo[b]:self.hi[b]]
        if self.bucket_dtype is not None and self.bucket_dtype != g.dtype:
            buf = g.to(self.bucket_dtype)
            self.handles.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), g, buf))
        else:
            self.handles.append((dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))

    def _arrived(self, b, hold=False):
        self.pending[b] += 1
        if self.pending[b] == self.count[b]:
            self.pending[b] = 0
            if self.enabled:
                if self.held is not None:
                    self._launch(self.held)
                    self.held = None
                if hold:
                    self.held = b
                else:
                    self._launch(b)

    def _make_hook(self, b):
        def hook(_p):
            self._arrived(b)
        return hook

    def stage_ready(self, ptrs):
        """Direct-gradient mode: the stage's kernels (accumulating into the flat buffer) are enqueued on the current stream --
        except possibly its last weight-gradient GEMM, which the library joins inside the next stage call (bf_side_defer).  A
        complete bucket is therefore reduced when the next bucket completes, the last one in flush() after bf_side_join()."""
        for ptr in ptrs:
            self._arrived(self.bucket_of_ptr[ptr], hold=True)

    def flush(self):
        if self.held is not None:
            self._launch(self.held)
            self.held = None

    def wait(self) -> float:
        """Block the current stream on the outstanding buckets; returns the factor the optimizer must apply (1/world)."""
        self.flush()
        if self.enabled:          # buckets nobody completed (a stage that fell back to autograd accumulation, unused parameters):
            for b in range(len(self.count)):      # the backward is over, so they are final; same order on every rank
                if b not in self.done and self.lo[b] is not None:
                    self._launch(b)
        for h, g, buf in self.handles:
            h.wait()
            if buf is not None:
                g.copy_(buf)
        self.handles.clear()
        self.done.clear()
        self.pending = [0] * len(self.pending)
        return 1.0 / self.world

    def begin_step(self):
        self.launch_log = []


def stage_buckets(model: nn.Module, blocks_per_bucket: Optional[int] = None) -> List[int]:
    """Bucket index per parameter: one bucket per top-level stage and per group of `blocks_per_bucket` consecutive processor blocks
    (gradient-ready order is the reverse of this list).  Each bucket is one collective: fewer, larger ones cost less host time
    and fewer stream synchronisations, more of them start the exchange earlier: a bucket is launched one bucket late (see
    BucketReducer.stage_ready), so at the end of the backward two buckets are still travelling.  BF_BLOCKS_PER_BUCKET, default 2
    (19 MB of fp32 gradients per collective for FiLMAViT-small)."""
    if blocks_per_bucket is None:
        blocks_per_bucket = int(os.environ.get("BF_BLOCKS_PER_BUCKET", "2"))
    ids, names = [], {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        parts = name.split(".")
        key = "blocks.%d" % (int(parts[1]) // max(1, blocks_per_bucket)) if parts[0] == "blocks" else parts[0]
        ids.append(names.setdefault(key, len(names)))
    return ids


class TrainStep:
    """optimizer: "adamw" (config/optim_cfg/adamw.yaml; modules.py:135-136) or "lion" (config/optim_cfg/lion.yaml, the reference
    default; modules.py:139-140).  scheduler: optional object with get_last_lr() / step() (utils.lr_schedulers.CosineWarmupLR),
    stepped once per optimizer step like the reference's ``interval="step"`` (modules.py:166-171)."""

    def __init__(self, model: nn.Module, lr: float = 2.5e-4, weight_decay: float = 1e-2, betas=None, eps: float = 1e-8,
                 optimizer: str = "adamw", scheduler=None):
        from . import ops
        if optimizer not in ("adamw", "lion"):
            raise ValueError(f"Optimizer {optimizer} not supported")
        self.ops = ops
        self.model = model
        self.flat = FlatParams(model)
        self.reducer = BucketReducer(self.flat, stage_buckets(model), use_hooks=False)
        self.slots = {p.data_ptr(): p.grad for p in self.flat.params}
        self.optimizer = optimizer
        self.m = torch.zeros_like(self.flat.flat)
        self.v = torch.zeros_like(self.flat.flat) if optimizer == "adamw" else None
        if betas is None:
            betas = (0.9, 0.999) if optimizer == "adamw" else (0.9, 0.99)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.scheduler = scheduler
        self.step_no = 0
        self.sync_from_rank0()

    def sync_from_rank0(self) -> None:
        """Data parallel: every replica starts (and resumes) from rank 0's parameters, optimizer moments and step count -- what
        `DistributedDataParallel` does at construction under Lightning's strategy="ddp" (scripts/train.py:158-172).  Only gradients are
        averaged afterwards, so a rank built from another seed would otherwise diverge silently."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        for t in (self.flat.flat, self.m, self.v):
            if t is not None:
                dist.broadcast(t, src=0)
        n = torch.tensor([self.step_no], dtype=torch.int64, device=self.flat.flat.device)
        dist.broadcast(n, src=0)
        self.step_no = int(n)
        from . import ops
        ops._weights_changed()           # the broadcast wrote the flat buffer behind the parameters' version counters: prepared inference weights are stale

    def __call__(self, x, fluid, target) -> torch.Tensor:
        self.flat.zero_grad()
        self.reducer.begin_step()
        self.ops.set_direct_grad_slots(self.slots, self.reducer.stage_ready, self.reducer.flush)
        self.ops.set_side_defer(True)       # a stage's weight-gradient GEMMs may run into the next stage; joined below
        try:
            loss = self._fwd_bwd(x, fluid, target)
        finally:
            self.ops.set_side_defer(False)
            self.ops.set_direct_grad_slots(None)
        gscale = self.reducer.wait()
        self.step_no += 1
        lr = self.scheduler.get_last_lr()[0] if self.scheduler is not None else self.lr
        if self.optimizer == "adamw":
            self.ops.adamw_(self.flat.flat, self.flat.grad, self.m, self.v, self.step_no, lr, self.betas, self.eps, self.wd, gscale)
        else:
            self.ops.lion_(self.flat.flat, self.flat.grad, self.m, lr, self.betas, self.wd, gscale)
        if self.scheduler is not None:
            self.scheduler.step()
        return loss.detach()

    def _fwd_bwd(self, x, fluid, target):
        loss, _ = self.model.forward_loss(x, fluid, target) if fluid is not None else self.model.forward_loss(x, target)
        loss.backward()
        return loss
