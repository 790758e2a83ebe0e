#!/usr/bin/env python3
"""Whole training step captured in a HIP graph vs eager enqueue (bench shape).  Usage: python tools/graph_step.py [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bubbleformer_amd.models import get_model  # noqa: E402
from bubbleformer_amd.trainer import TrainStep  # noqa: E402
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["configs1"]
torch.manual_seed(42)
model = get_model("filmavit", time_window=w["T"], drop_path=bench.DROP_PATH, compute_dtype=torch.bfloat16, **bench.CFG).to(dev).train()
step = TrainStep(model, lr=2.5e-4, weight_decay=1e-2)
x, cond, y = bench.synthetic_batch(w, 42, dev)
for _ in range(5):
    loss = step(x, cond, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step(x, cond, y)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / steps
print(f"eager: {eager * 1e3:.3f} ms/step  {w['batch'] / eager:.1f} samples/s  loss {float(loss):.5f}", flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step(x, cond, y)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gl = step(x, cond, y)
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    g.replay()
torch.cuda.synchronize()
gr = (time.perf_counter() - t0) / steps
print(f"graph: {gr * 1e3:.3f} ms/step  {w['batch'] / gr:.1f} samples/s  loss {float(gl):.5f}", flush=True)
