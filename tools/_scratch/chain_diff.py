import os, sys, torch
sys.path.insert(0, os.getcwd())
from tests.test_gpu_baseline_configs import _inputs, _model
from bubbleformer_amd.trainer import TrainStep
B, T, H, W, seed = 2, 16, 192, 192, 12
x, y, c = (t.cuda() for t in _inputs(B, T, H, W, seed))
def run():
    m = _model(seed, torch.bfloat16, T)
    step = TrainStep(m, lr=0.0, weight_decay=0.0)
    loss = float(step(x, c, y))
    torch.cuda.synchronize()
    return loss, {k: p.grad.detach().clone() for k, p in m.named_parameters()}
l1, g1 = run()
l1b, g1b = run()
os.environ["BF_STAGE_CHAIN"] = "0"
l2, g2 = run()
print(l1, l1b, l2)
rows = []
for k in g1:
    d = float((g1[k] - g2[k]).norm()); n = float(g2[k].norm()); d0 = float((g1[k] - g1b[k]).norm())
    rows.append((d, d / max(n, 1e-30), d0, k))
rows.sort(reverse=True)
for r in rows[:14]: print("%.3e rel %.3e  (run-to-run %.3e)  %s" % r)
tot = sum(r[0] ** 2 for r in rows) ** 0.5
print("total", tot, sum(float(g2[k].norm()) ** 2 for k in g2) ** 0.5)
