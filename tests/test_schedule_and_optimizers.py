"""Training-step harness pieces beside the model: learning-rate schedule (CPU, pinned by the reference class's own output) and the
fused optimizer kernels (GPU, against the oracle restatements)."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cosine_warmup_lr.npz")


def test_cosine_warmup_lr_matches_reference_class():
    """tests/golden/cosine_warmup_lr.npz: learning rates read off the reference's CosineWarmupLR (oracle/gen_golden.py)."""
    from bubbleformer_amd.utils import CosineWarmupLR
    from oracle.filmavit_ref import cosine_warmup_lr
    z = np.load(GOLDEN)
    cases = sorted({k.split("/")[0] for k in z.files})
    assert len(cases) == 3
    for c in cases:
        lr, warm, tmax, eta = z[c + "/params"]
        ref = z[c + "/lr"]
        sch = CosineWarmupLR(lr, int(warm), int(tmax), eta)
        for step, want in enumerate(ref):
            got = sch.get_last_lr()[0]
            assert abs(got - want) <= 1e-12 * max(abs(want), 1e-30) + 1e-18, (c, step, got, want)
            assert abs(cosine_warmup_lr(step, lr, int(warm), int(tmax), eta) - want) <= 1e-12 * max(abs(want), 1e-30) + 1e-18
            sch.step()
    sch = CosineWarmupLR(1.0, 3, 7)
    for _ in range(5):
        sch.step()
    sch2 = CosineWarmupLR(1.0, 3, 7)
    sch2.load_state_dict(sch.state_dict())
    assert sch2.get_last_lr() == sch.get_last_lr()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096, 1003])
def test_fused_lion_matches_oracle(n):
    from bubbleformer_amd import ops
    from oracle.filmavit_ref import lion_step
    g = torch.Generator().manual_seed(9)
    p = torch.randn(n + 4, generator=g)[:n].clone()
    m = 0.1 * torch.randn(n, generator=g)
    pd, md = p.clone().cuda(), m.clone().cuda()
    for step in range(4):
        grad = torch.randn(n, generator=g)
        if step == 2:
            grad[:17] = 0.0
            m[:17] = 0.0
            md[:17] = 0.0                       # sign(0) = 0: parameters only decay there
        lion_step(p, grad, m, lr=5e-5, beta1=0.9, beta2=0.99, wd=0.1)
        ops.lion_(pd, grad.cuda(), md, 5e-5, (0.9, 0.99), 0.1)
    assert torch.allclose(pd.cpu(), p, rtol=2e-6, atol=2e-7)          # fused multiply-add vs two roundings per step
    assert torch.allclose(md.cpu(), m, rtol=1e-5, atol=1e-7)
    # gradient pre-scale (1 / world size of the data-parallel mean) is applied before the sign
    p2, m2 = p.clone(), m.clone()
    grad = torch.randn(n, generator=g)
    lion_step(p2, grad * 0.25, m2, lr=5e-5, wd=0.1)
    ops.lion_(pd, grad.cuda(), md, 5e-5, (0.9, 0.99), 0.1, 0.25)
    assert torch.allclose(pd.cpu(), p2, rtol=2e-6, atol=2e-7)


@pytest.mark.gpu
def test_train_step_with_lion_and_schedule_reduces_loss():
    """The reference's default optimizer and schedule through the native step: loss goes down on a fixed batch, the schedule advances."""
    from bubbleformer_amd.models import get_model
    from bubbleformer_amd.trainer import TrainStep
    from bubbleformer_amd.utils import CosineWarmupLR
    torch.manual_seed(0)
    cfg = dict(input_fields=4, output_fields=4, patch_size=4, embed_dim=64, num_heads=2, processor_blocks=2, num_fluid_params=9)
    model = get_model("filmavit", time_window=4, drop_path=0.0, **cfg).cuda().train()
    sch = CosineWarmupLR(3e-4, warmup_iters=3, max_iters=40, eta_min=1e-6)
    step = TrainStep(model, lr=3e-4, weight_decay=0.1, optimizer="lion", scheduler=sch)
    x = torch.randn(2, 4, 4, 16, 16, device="cuda")
    c = torch.randn(2, 9, device="cuda")
    y = torch.randn(2, 4, 4, 16, 16, device="cuda")
    losses = [float(step(x, c, y)) for _ in range(25)]
    assert sch.last_epoch == 25
    assert losses[-1] < 0.97 * losses[1], losses
