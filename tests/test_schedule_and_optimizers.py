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
    model = get_model("filmavit", time_window=4, drop_path=0.0, compute_dtype=torch.bfloat16, **cfg).cuda().train()
    sch = CosineWarmupLR(3e-4, warmup_iters=3, max_iters=40, eta_min=1e-6)
    step = TrainStep(model, lr=3e-4, weight_decay=0.1, optimizer="lion", scheduler=sch)
    x = torch.randn(2, 4, 4, 16, 16, device="cuda")
    c = torch.randn(2, 9, device="cuda")
    y = torch.randn(2, 4, 4, 16, 16, device="cuda")
    losses = [float(step(x, c, y)) for _ in range(25)]
    assert sch.last_epoch == 25
    assert losses[-1] < 0.97 * losses[1], losses


def test_checkpoint_layout_is_the_reference_one(tmp_path):
    """state_dict keys carry Lightning's "model." prefix (scripts/inference.py:222-225 strips 6 characters), hyper-parameters keep the
    normalisation constants (modules.py:46-57); CPU-only check on a stock nn container with the reference's key names."""
    from bubbleformer_amd.utils import checkpoint as C
    from oracle import weights as Wt
    cfg = dict(input_fields=4, output_fields=4, patch_size=4, embed_dim=16, num_heads=2, processor_blocks=1, num_fluid_params=9)
    sd = Wt.generate(Wt.param_shapes(**cfg), seed=1)

    class Holder(torch.nn.Module):                      # same keys / shapes as the model, no GPU needed
        def __init__(self):
            super().__init__()
            self.p = torch.nn.ParameterDict({k.replace(".", "|"): torch.nn.Parameter(v.clone()) for k, v in sd.items()})

        def state_dict(self, *a, **k):
            return {n.replace("|", "."): t for n, t in self.p.items()}
    m = Holder()
    path = str(tmp_path / "hpc_ckpt_1.ckpt")
    C.save_checkpoint(path, m, hyper_parameters={"model_cfg": {"name": "filmavit", "params": cfg}},
                      normalization_constants=({"dfun": 0.1}, {"dfun": 2.0}), global_step=7)
    raw = torch.load(path, weights_only=False)
    assert set(raw) >= {"state_dict", "hyper_parameters", "global_step"} and raw["global_step"] == 7
    assert all(k.startswith("model.") for k in raw["state_dict"])
    assert raw["hyper_parameters"]["normalization_constants"] == ({"dfun": 0.1}, {"dfun": 2.0})
    stripped = {k[6:]: v for k, v in raw["state_dict"].items()}            # what the reference's inference script does
    assert set(stripped) == set(sd) and all(torch.equal(stripped[k], sd[k]) for k in sd)
    with torch.no_grad():
        for t in m.p.values():
            t.zero_()
    C.load_checkpoint(path, m)
    assert all(torch.equal(t, sd[n.replace("|", ".")]) for n, t in m.p.items())


@pytest.mark.gpu
def test_checkpoint_resumes_the_training_step(tmp_path):
    from bubbleformer_amd.models import get_model
    from bubbleformer_amd.trainer import TrainStep
    from bubbleformer_amd.utils import CosineWarmupLR, checkpoint as C
    cfg = dict(input_fields=4, output_fields=4, patch_size=4, embed_dim=64, num_heads=2, processor_blocks=2, num_fluid_params=9)

    def make():
        torch.manual_seed(0)
        model = get_model("filmavit", time_window=4, drop_path=0.0, **cfg).cuda().train()
        return model, TrainStep(model, lr=1e-3, optimizer="adamw", scheduler=CosineWarmupLR(1e-3, 2, 30, 1e-6))
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(2, 4, 4, 16, 16, device="cuda", generator=g)
    c = torch.randn(2, 9, device="cuda", generator=g)
    y = torch.randn(2, 4, 4, 16, 16, device="cuda", generator=g)
    m1, s1 = make()
    for _ in range(3):
        s1(x, c, y)
    path = str(tmp_path / "resume.ckpt")
    C.save_checkpoint(path, m1, train_step=s1)
    ref = [float(s1(x, c, y)) for _ in range(3)]
    m2, s2 = make()
    C.load_checkpoint(path, m2, train_step=s2)
    assert s2.step_no == 3 and s2.scheduler.last_epoch == 3
    got = [float(s2(x, c, y)) for _ in range(3)]
    assert got == ref                                   # bit-identical continuation (same kernels, same state)
