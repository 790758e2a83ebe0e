#!/usr/bin/env python3
"""Two forward + backward passes on the same inputs (one 16x192x192 sample, bf16): which outputs differ run to run, and by how much."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_baseline_configs import _product  # noqa: E402
from tests.helpers import rel_l2  # noqa: E402
B, T, H, W, seed = 1, 16, 192, 192, 12
a = _product(B, T, H, W, seed, torch.bfloat16)
b = _product(B, T, H, W, seed, torch.bfloat16)
print("pred", rel_l2(a[0], b[0]), "loss", a[1], b[1], "dx", rel_l2(a[2], b[2]))
worst = sorted(((rel_l2(a[3][k], b[3][k]) if float(b[3][k].norm()) > 0 else 0.0, k) for k in a[3]), reverse=True)
print("families differing:", sum(1 for e, _ in worst if e > 0), "of", len(worst))
for e, k in worst[:25]:
    print(f"  {e:.3e}  {k}")
