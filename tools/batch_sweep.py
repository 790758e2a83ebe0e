#!/usr/bin/env python3
"""Training-step rate at the bench shape for several batch sizes (sanity of the size-dependent launch geometry)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from tools.config_bench import train_rate
for B in (1, 3, 8, 16, 24):
    r, ms = train_rate(B, 16, 192, 192, steps=10, warm=3)
    print(f"batch {B}: {r:.1f} samples/s ({ms:.2f} ms/step)")
