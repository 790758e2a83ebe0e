"""Autoregressive rollout kept on the device (reference loop: scripts/inference.py:239-252).

The reference moves every prediction to the host and back (`pred.squeeze(0).detach().cpu()` ... `inp.cuda()`); here the eval
forward is captured once in a HIP graph and replayed per step with the previous prediction copied into the graph's static input,
so a step is one graph launch and the trajectory never leaves HBM."""
from typing import Callable, List, Optional, Tuple

import torch


def relative_l2_per_step(pred: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """LpLoss(d=2, p=2, reduce_dims=[0, 1], reductions=["mean", "mean"]) of scripts/inference.py:230 on (T, C, H, W) clips."""
    num = (pred - tgt).flatten(-2).norm(dim=-1)
    den = tgt.flatten(-2).norm(dim=-1)
    return (num / den).mean(0, keepdim=True).mean(1, keepdim=True).squeeze()


class GraphedForward:
    """model(x, *extra) in eval / no-grad mode, captured in a HIP graph for one input shape.

    Weights: the captured graph reads the model's PREPARED inference weights (bf16 copies and out-projection folds in a per-model arena,
    ops.trunk_eval) by address.  Before every replay the arena is re-prepared in place if the parameters changed since (torch in-place
    updates, ops.adamw_ / ops.lion_, load_state_dict), so a GraphedForward kept across optimizer steps tracks the live parameters; the
    arena and the scratch buffers the capture saw are pinned for the lifetime of the process (ops.clear_scratch / clear_eval_weights
    release them -- only once the graph is gone)."""

    def __init__(self, model, x: torch.Tensor, *extra: torch.Tensor, warmup: int = 2):
        self.model = model.eval()
        self.static_x = x.clone()
        self.extra = extra
        with torch.no_grad():
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    self.model(self.static_x, *extra)
            torch.cuda.current_stream().wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = self.model(self.static_x, *extra)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        self.static_x.copy_(x)
        owner = self.model.__dict__.get("_bf_eval_owner")
        if owner is not None:
            from .. import ops
            ops.refresh_eval_weights(owner)
        self.graph.replay()
        return self.static_out


def autoregressive_rollout(model, first_input: torch.Tensor, steps: int, *extra: torch.Tensor, use_graph: bool = True,
                           target_fn: Optional[Callable[[int], torch.Tensor]] = None) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """first_input (T, C, H, W) on the device; each step feeds the previous prediction back (scripts/inference.py:242-245).
    Returns (predictions concatenated along time: (steps*T, C, H, W), per-step relative L2 against target_fn(step) if given)."""
    x = first_input.unsqueeze(0).float()
    fwd = GraphedForward(model, x, *extra) if use_graph else None
    preds, errs = [], []
    with torch.no_grad():
        for s in range(steps):
            out = fwd(x) if fwd is not None else model.eval()(x, *extra)
            pred = out.squeeze(0).clone()
            preds.append(pred)
            if target_fn is not None:
                errs.append(relative_l2_per_step(pred, target_fn(s)))
            x = pred.unsqueeze(0)
    return torch.cat(preds, dim=0), errs
