"""Physics metrics of a rollout, on the device (reference: bubbleformer/utils/losses.py:5-15, bubbleformer/utils/heatflux.py)."""
import ctypes as C

import torch

from .. import _lib as L
from ..ops import _p, _require_gpu, _stream


def eikonal_loss(phi: torch.Tensor) -> torch.Tensor:
    """phi = SDF tensor (..., H, W): mean over all elements of (|grad phi| - 1)^2 with dx = 1/32 (utils/losses.py:5-15)."""
    _require_gpu(phi)
    phi = phi.contiguous().float()
    H, W = phi.shape[-2:]
    frames = phi.numel() // (H * W)
    acc = torch.zeros(1, dtype=torch.float64, device=phi.device)
    L.check(L.lib().bf_eikonal_sum(_p(phi), frames, H, W, 1.0 / 32, _p(acc), _stream()), "bf_eikonal_sum")
    return (acc / phi.numel()).float().squeeze(0)


def eikonal_l1_per_frame(phi: torch.Tensor) -> torch.Tensor:
    """phi (T, H, W) SDF frames -> (T,) scores of the rollout notebook (`get_eikonal_loss`, scripts/inference_autoregressive.ipynb):
    mean | |grad phi| - 1 | per frame, central differences at dx = 1/32 with replicate-padded borders."""
    _require_gpu(phi)
    if phi.dim() != 3:
        raise ValueError("eikonal_l1_per_frame expects (T, H, W)")
    phi = phi.contiguous().float()
    T, H, W = phi.shape
    out = torch.empty(T, dtype=torch.float32, device=phi.device)
    L.check(L.lib().bf_eikonal_l1_frames(_p(phi), T, H, W, 1.0 / 32, _p(out), _stream()), "bf_eikonal_l1_frames")
    return out


def heatflux(dfun: torch.Tensor, temp: torch.Tensor, heater_temp: float):
    """FC-72 heater heat flux (utils/heatflux.py:3-38): dfun, temp (T, 512, 512) device tensors -> (mean, max) over frames of the
    bottom-row flux.  The reference hard-codes the 16 x 16 domain at dx = 1/32 (512 x 512 cells); so does this."""
    _require_gpu(dfun)
    if tuple(dfun.shape[1:]) != (512, 512) or dfun.shape != temp.shape:
        raise ValueError("heatflux expects (T, 512, 512) fields (utils/heatflux.py:21-33)")
    dfun, temp = dfun.contiguous().float(), temp.contiguous().float()
    T = dfun.shape[0]
    flux = torch.empty(T, dtype=torch.float32, device=dfun.device)
    L.check(L.lib().bf_heatflux_rows(_p(dfun), _p(temp), T, 512 * 512, 512, -8.0, 1.0 / 32, float(heater_temp), 0.0007, _p(flux), _stream()),
            "bf_heatflux_rows")
    return flux.mean(), flux.max()
