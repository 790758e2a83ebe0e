"""Constructor branches of the mirrored layers and the scheduler overload (CPU; the GPU halves are in test_gpu_parity.py)."""
import math

import pytest
import torch

from bubbleformer_amd.layers import AttentionBlock, AxialAttentionBlock
from bubbleformer_amd.utils import CosineWarmupLR


def test_bias_type_none_has_no_table_and_continuous_is_declined():
    """layers/attention.py:58-63,174-179 of the reference: "none" -> no bias term (and no embedding table in the state_dict), any other
    string but "continuous" -> the T5 table; "continuous" is the one branch that is not built."""
    for cls in (AttentionBlock, AxialAttentionBlock):
        rel, none, other = cls(64, 2), cls(64, 2, bias_type="none"), cls(64, 2, bias_type="anything")
        assert any("relative_attention_bias" in k for k in rel.state_dict())
        assert set(other.state_dict()) == set(rel.state_dict())
        assert set(none.state_dict()) == {k for k in rel.state_dict() if "rel_pos_bias" not in k}
        assert none.stage_params()[14 if cls is AttentionBlock else 18] is None
        with pytest.raises(NotImplementedError):
            cls(64, 2, bias_type="continuous")


def test_layer_scale_off_constructs_and_fails_in_forward_like_the_reference():
    """layer_scale_init_value <= 0 gives `gamma = None` (attention.py:41-46,155-168) and the reference's own forward then raises
    TypeError on `self.gamma[None, ...]` (:123, :309): same constructor result, same exception type."""
    a = AttentionBlock(64, 2, layer_scale_init_value=0.0)
    b = AxialAttentionBlock(64, 2, layer_scale_init_value=-1.0)
    assert a.gamma is None and b.gamma_att is None and b.gamma_mlp is None
    assert "gamma" not in a.state_dict() and "gamma_att" not in b.state_dict()
    with pytest.raises(TypeError):
        a(torch.zeros(1, 2, 64, 4, 4))
    with pytest.raises(TypeError):
        b(torch.zeros(1, 64, 4, 4))


def test_cosine_warmup_lr_takes_the_optimizer_first_like_the_reference():
    """utils/lr_schedulers.py:13 of the reference: CosineWarmupLR(optimizer, warmup_iters, max_iters, eta_min, last_epoch).  The group's
    lr follows torch's SequentialLR(LambdaLR, CosineAnnealingLR) of the reference class step by step."""
    from torch.optim.lr_scheduler import CosineAnnealingLR, LambdaLR, SequentialLR
    def make():
        return torch.optim.AdamW([torch.nn.Parameter(torch.zeros(3))], lr=2.5e-4)
    o1, o2 = make(), make()
    ours = CosineWarmupLR(o1, 7, 40, eta_min=1e-6)
    ref = SequentialLR(o2, schedulers=[LambdaLR(o2, lr_lambda=lambda step: step / 7), CosineAnnealingLR(o2, T_max=40, eta_min=1e-6)],
                       milestones=[7])
    for _ in range(45):
        assert math.isclose(o1.param_groups[0]["lr"], o2.param_groups[0]["lr"], rel_tol=1e-12, abs_tol=1e-18)
        assert math.isclose(ours.get_last_lr()[0], ref.get_last_lr()[0], rel_tol=1e-12, abs_tol=1e-18)
        o1.step(); o2.step()
        ours.step(); ref.step()
    plain = CosineWarmupLR(2.5e-4, 7, 40, eta_min=1e-6)          # the native TrainStep's form: a base learning rate
    plain.load_state_dict(ours.state_dict())
    assert plain.get_last_lr() == ours.get_last_lr()
