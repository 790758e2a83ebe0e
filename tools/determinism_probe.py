#!/usr/bin/env python3
"""Where does the backward first differ between two runs on identical inputs?  Hooks the gradient entering every trunk stage."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_baseline_configs import _model, _inputs  # noqa: E402
from bubbleformer_amd.models.axial_vit import SpaceTimeBlock  # noqa: E402

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
B, T, H, W, seed = 1, 16, 192, 192, 12
if len(sys.argv) > 2:
    T, H, W = (int(v) for v in sys.argv[2].split("x"))
grads = []
cur = None
orig = SpaceTimeBlock.forward_tokens

def patched(self, tok, drops=None):
    t1 = self.temporal.forward_tokens(tok)
    t1.register_hook(lambda g, n=("temporal-out", id(self)): cur.append((n[0], g.detach().float().clone())))
    t2 = self.spatial.forward_tokens(t1)
    t2.register_hook(lambda g, n=("spatial-out", id(self)): cur.append((n[0], g.detach().float().clone())))
    return t2

SpaceTimeBlock.forward_tokens = patched
runs = []
for rep in range(2):
    m = _model(seed, dt, T)
    x, y, c = (t.cuda() for t in _inputs(B, T, H, W, seed))
    x.requires_grad_(True)
    cur = []
    if os.environ.get("PROBE_UNFUSED") == "1":
        pred = m(x, c)
        num = ((pred - y) ** 2).sum(dim=(-1, -2)).sqrt()
        den = (y ** 2).sum(dim=(-1, -2)).sqrt()
        loss = (num / den).mean(0).mean(0).sum()
    else:
        loss, pred = m.forward_loss(x, c, y)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((cur, x.grad.detach().float().clone()))
a, b = runs
print("hooks per run:", len(a[0]))
for i, ((n1, g1), (n2, g2)) in enumerate(zip(a[0], b[0])):
    d = float((g1 - g2).norm() / g2.norm().clamp_min(1e-30))
    nz = int((g1 != g2).sum())
    print(f"{i:3d} {n1:13s} rel diff {d:.3e}  elements differing {nz} / {g1.numel()}")
print("dx rel diff", float((a[1] - b[1]).norm() / b[1].norm()))
