#!/usr/bin/env python3
"""Eval forward (bf16, 16x192x192 clips) per batch size through the whole-frame inference path (ops.trunk_eval) and through the stage
forwards (BF_TRUNK_EVAL=0).  Usage: python tools/eval_batch_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bubbleformer_amd.models import get_model  # noqa: E402
CFG = dict(input_fields=4, output_fields=4, patch_size=16, embed_dim=384, num_heads=6, processor_blocks=12, num_fluid_params=9)
m = get_model("filmavit", time_window=16, drop_path=0.0, compute_dtype=torch.bfloat16, **CFG).cuda().eval()
for B in (1, 2, 4, 8, 16):
    x = torch.randn(B, 16, 4, 192, 192, device="cuda"); c = torch.randn(B, 9, device="cuda")
    res = {}
    for mode in ("1", "0"):
        os.environ["BF_TRUNK_EVAL"] = mode
        with torch.no_grad():
            for _ in range(3): m(x, c)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 20
            for _ in range(n): m(x, c)
            torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / n * 1e3
    print(f"batch {B:2d}: whole-frame path {res['1']:7.2f} ms   stage forwards {res['0']:7.2f} ms", flush=True)
