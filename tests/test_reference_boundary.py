"""The drop-in boundary against the REAL reference: `bubbleformer_amd.install_into_reference()` puts the native
classes into the reference's own registry, so `bubbleformer.models.get_model("filmavit", ...)` -- what `ForecastModule.__init__`
(bubbleformer/modules.py:51-54) and scripts/inference.py:202 call -- returns the native class with the reference class's state_dict
keys and shapes.  Needs the reference checkout (/root/reference: the build container only; skipped on the GPU box, where it does not
exist).  Runs in a child process so that the reference's `bubbleformer` package never enters this test session's module table."""
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CHILD = textwrap.dedent("""
    import sys, types
    sys.dont_write_bytecode = True
    import torch
    timm = types.ModuleType("timm"); layers = types.ModuleType("timm.layers")
    class DropPath(torch.nn.Module):            # timm is not installed here (SURVEY.md section 8c); never executed, only constructed
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__(); self.drop_prob = drop_prob
    layers.DropPath = DropPath; timm.layers = layers
    sys.modules["timm"] = timm; sys.modules["timm.layers"] = layers
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import bubbleformer.models as ref_models
    from bubbleformer.models import axial_vit as ref_axial
    ref_cls = {"filmavit": ref_axial.FiLMConditionedAViT, "avit": ref_axial.AViT}
    assert ref_models.get_model("filmavit", input_fields=4, output_fields=4, time_window=4, patch_size=4, embed_dim=64, num_heads=2,
                                processor_blocks=1, drop_path=0.0, num_fluid_params=9).__class__ is ref_cls["filmavit"]
    import bubbleformer_amd
    from bubbleformer_amd.models import axial_vit as native
    bubbleformer_amd.install_into_reference()
    small = dict(input_fields=4, output_fields=4, time_window=16, patch_size=16, embed_dim=384, num_heads=6, processor_blocks=12,
                 drop_path=0.2, attn_scale=True, feat_scale=True)       # config/model_cfg/film_avit_small.yaml + modules.py:51-53
    tiny = dict(input_fields=3, output_fields=2, time_window=3, patch_size=8, embed_dim=96, num_heads=4, processor_blocks=2, drop_path=0.0,
                attn_scale=False, feat_scale=False)
    for name, extra in (("filmavit", dict(num_fluid_params=9)), ("avit", {})):
        for cfg in (small, tiny):
            m = ref_models.get_model(name.upper() if cfg is tiny else name, **cfg, **extra)       # the registry lower-cases (models/_api.py:35)
            assert type(m) is getattr(native, ref_cls[name].__name__), type(m)
            ref = ref_cls[name](**cfg, **extra)
            a, b = m.state_dict(), ref.state_dict()
            assert list(a.keys()) == list(b.keys()), (name, set(a) ^ set(b))
            assert all(tuple(a[k].shape) == tuple(b[k].shape) and a[k].dtype == b[k].dtype for k in a)
            m.load_state_dict(b)                                      # a reference checkpoint loads as it stands
            assert [n for n, _ in m.named_modules()][:4] == [n for n, _ in ref.named_modules()][:4]
    assert ref_models.list_models() == ["avit", "filmavit", "unet_classic", "unet_modern"]
    print("BOUNDARY-OK")
""")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "bubbleformer")), reason="reference checkout not present (GPU box)")
def test_install_into_reference_registers_native_classes_with_reference_state_dict():
    res = subprocess.run([sys.executable, "-c", CHILD % (REF, REPO)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert res.returncode == 0 and "BOUNDARY-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
