"""bubbleformer_amd -- MI355X-native (gfx950) FiLMAViT forward/backward path behind the
HPCForge/Bubbleformer ``bubbleformer.models`` / ``bubbleformer.layers`` nn.Module API.

    from bubbleformer_amd.models import get_model
    model = get_model("filmavit", **cfg).cuda()

The compute path is hand-written HIP (``csrc/``) behind a C ABI (``include/bubbleformer_hip.h``); there is no
CPU or eager-PyTorch fallback -- importing works anywhere, running needs the built library and a ROCm GPU.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def install_into_reference() -> None:
    """Register the native models in an importable reference checkout's registry so that the reference's own
    ``scripts/train.py`` / ``scripts/inference.py`` pick them up unchanged."""
    import bubbleformer.models._api as ref_api  # the user's reference checkout
    from .models import axial_vit
    ref_api.MODELS["filmavit"] = axial_vit.FiLMConditionedAViT
    ref_api.MODELS["avit"] = axial_vit.AViT
