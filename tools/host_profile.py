#!/usr/bin/env python3
"""Where the host spends its time while it enqueues a training step (the GPU runs ~13.5 ms per step; the host must stay ahead)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench as B
from bubbleformer_amd.models import get_model
from bubbleformer_amd.trainer import TrainStep

dev = torch.device("cuda", 0)
torch.manual_seed(42)
W = B.WORKLOADS["configs1"]
model = get_model("filmavit", time_window=W["T"], drop_path=B.DROP_PATH, compute_dtype=torch.bfloat16, **B.CFG).to(dev).train()
step = TrainStep(model, lr=2.5e-4, weight_decay=1e-2)
x, cond, y = B.synthetic_batch(W, 42, dev)
for _ in range(5):
    step(x, cond, y)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step(x, cond, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, wall {1e3 * (t2 - t0) / n:.2f} ms/step")
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(x, cond, y)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
