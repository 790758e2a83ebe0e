#!/usr/bin/env python3
"""Extra data points beside bench.py (which measures BASELINE configs[1]): configs[3] long-aspect training step and configs[4]
bs-1 autoregressive rollout (eager vs HIP-graph replay).  Usage: python tools/config_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd.models import get_model  # noqa: E402
from bubbleformer_amd.trainer import TrainStep  # noqa: E402
from bubbleformer_amd.utils.rollout import GraphedForward  # noqa: E402

CFG = dict(input_fields=4, output_fields=4, patch_size=16, embed_dim=384, num_heads=6, processor_blocks=12, num_fluid_params=9)
dev = torch.device("cuda")


def train_rate(B, T, H, W, steps=10, warm=3, cfg=CFG):
    torch.manual_seed(0)
    m = get_model("filmavit", time_window=T, drop_path=0.2, compute_dtype=torch.bfloat16, **cfg).to(dev).train()
    step = TrainStep(m)
    x = torch.randn(B, T, 4, H, W, device=dev)
    y = torch.randn(B, T, 4, H, W, device=dev)
    c = torch.randn(B, 9, device=dev)
    for _ in range(warm):
        step(x, c, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(x, c, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return B / dt, dt * 1e3


def rollout_ms(T=16, H=192, W=192, steps=50):
    torch.manual_seed(0)
    m = get_model("filmavit", time_window=T, drop_path=0.0, compute_dtype=torch.bfloat16, **CFG).to(dev).eval()
    x = torch.randn(1, T, 4, H, W, device=dev)
    c = torch.randn(1, 9, device=dev)
    out = {}
    with torch.no_grad():
        for _ in range(3):
            m(x, c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cur = x
        for _ in range(steps):
            cur = m(cur, c)
        torch.cuda.synchronize()
        out["eager"] = (time.perf_counter() - t0) / steps * 1e3
        g = GraphedForward(m, x, c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cur = x
        for _ in range(steps):
            cur = g(cur)
        torch.cuda.synchronize()
        out["graph"] = (time.perf_counter() - t0) / steps * 1e3
    return out


if __name__ == "__main__":
    for B in (1, 2, 4):
        r, ms = train_rate(B, 32, 384, 192)
        print(f"configs[3] train 32x384x192 bs {B}: {r:.1f} samples/s ({ms:.1f} ms/step)")
    r, ms = train_rate(4, 32, 192, 384)
    print(f"configs[3] transposed train 32x192x384 bs 4: {r:.1f} samples/s ({ms:.1f} ms/step)")
    print("configs[4] rollout 16x192x192 bs 1, ms/step:", rollout_ms())
    big = dict(CFG, embed_dim=768, num_heads=12)        # config/model_cfg/film_avit_big.yaml
    r, ms = train_rate(8, 16, 192, 192, cfg=big)
    print(f"film_avit_big train 16x192x192 bs 8: {r:.1f} samples/s ({ms:.1f} ms/step)")
