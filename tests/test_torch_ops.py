"""CPU: the dispatcher-visible operators of the inference path are registered with the schemas torch_ops.py documents (no compute here)."""
import torch


def test_operator_schemas():
    from bubbleformer_amd import torch_ops  # noqa: F401
    s = str(torch.ops.bubbleformer_amd.trunk_eval.default._schema)
    assert s.startswith("bubbleformer_amd::trunk_eval(Tensor tok, SymInt heads, bool attn_scale, bool feat_scale, SymInt[] kinds, Tensor?[] params, SymInt owner=0) -> Tensor"), s
    s = str(torch.ops.bubbleformer_amd.frame_linear.default._schema)
    assert "Tensor a, Tensor w, SymInt frames, SymInt tokens_per_frame" in s and s.endswith("-> Tensor"), s


def test_trunk_eval_applies_only_to_the_covered_shape():
    from bubbleformer_amd import ops
    assert not ops.trunk_eval_applies(torch.zeros(1, 4, 12, 12, 384, dtype=torch.bfloat16))        # not on the GPU
    assert not ops.trunk_eval_applies(torch.zeros(1, 4, 12, 12, 384))


def test_eval_tokens_are_unique():
    from bubbleformer_amd import ops
    a, b = ops.new_eval_token(), ops.new_eval_token()
    assert a != b and a > 0 and b > 0
