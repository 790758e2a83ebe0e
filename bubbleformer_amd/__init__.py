"""bubbleformer_amd -- MI355X-native (gfx950) FiLMAViT forward/backward path behind the
HPCForge/Bubbleformer ``bubbleformer.models`` / ``bubbleformer.layers`` nn.Module API.

    from bubbleformer_amd.models import get_model
    model = get_model("filmavit", **cfg).cuda()

The compute path is hand-written HIP (``csrc/``) behind a C ABI (``include/bubbleformer_hip.h``); there is no
CPU or eager-PyTorch fallback -- importing works anywhere, running needs the built library and a ROCm GPU.
"""
import os as _os

# The stage backwards overlap weight-gradient GEMMs on a second HIP stream.  HIP deals streams onto GPU_MAX_HW_QUEUES hardware
# queues round-robin; at the default of 4 the streams RCCL creates land that side stream on the main stream's queue and the overlap
# is lost.  Takes effect only if the HIP runtime has not been initialised yet; an explicit setting wins.
_hwq_preset = "GPU_MAX_HW_QUEUES" in _os.environ
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def _warn_if_hip_already_up() -> None:
    """The setting above is read once, when the HIP runtime initialises.  Imported into a process that has already touched the GPU
    (the normal case when dropped into the reference's scripts/train.py after a torch.cuda call) it silently has no effect and the
    weight-gradient side stream can share a hardware queue with the caller's stream (measured: 589 -> 479 samples/s): say so."""
    import sys as _sys
    import warnings as _warnings
    t = _sys.modules.get("torch")
    if t is not None and not _hwq_preset and t.cuda.is_available() and t.cuda.is_initialized():
        _warnings.warn("bubbleformer_amd: the HIP runtime was initialised before this import, so GPU_MAX_HW_QUEUES=8 cannot take effect; "
                       "export GPU_MAX_HW_QUEUES=8 in the environment (or import bubbleformer_amd before the first torch.cuda call) -- "
                       "with the default of 4 hardware queues the backward loses its two-stream overlap (DESIGN.md section 5)",
                       RuntimeWarning, stacklevel=3)


_warn_if_hip_already_up()

from . import _lib  # noqa: F401,E402

__version__ = "0.1.0"


def install_into_reference() -> None:
    """Register the native models in an importable reference checkout's registry so that the reference's own
    ``scripts/train.py`` / ``scripts/inference.py`` pick them up unchanged."""
    import bubbleformer.models._api as ref_api  # the user's reference checkout
    from .models import axial_vit
    ref_api.MODELS["filmavit"] = axial_vit.FiLMConditionedAViT
    ref_api.MODELS["avit"] = axial_vit.AViT
