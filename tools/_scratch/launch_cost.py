import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from bubbleformer_amd import _lib as L
from bubbleformer_amd.ops import _p, _stream
lib = L.lib()
p = torch.zeros(256, device="cuda"); g = torch.zeros_like(p); m = torch.zeros_like(p); v = torch.zeros_like(p)
st = _stream()
def run(n):
    t0 = time.perf_counter()
    for _ in range(n):
        lib.bf_adamw(_p(p), _p(g), _p(m), _p(v), 256, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6
run(100)
print("tiny launches: host %.2f us each, wall %.2f us each" % run(5000))
pp, gg, mm, vv = _p(p), _p(g), _p(m), _p(v)
t0 = time.perf_counter()
for _ in range(5000):
    lib.bf_adamw(pp, gg, mm, vv, 256, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, st)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("with cached pointers: host %.2f us each" % ((t1 - t0) / 5000 * 1e6))
e = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(2000):
    e.record()
t1 = time.perf_counter()
print("event record %.2f us" % ((t1 - t0) / 2000 * 1e6))
