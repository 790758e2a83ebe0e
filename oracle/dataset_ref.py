"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy) of the reference's clip supply, ``BubbleForecast`` (bubbleformer/data/dataset.py), on arrays that
have already been read from the trajectory files:

  * ``num_samples``      -- dataset.py:67-71   sum over files of (traj_len - start_time - 2*time_window + 1)
  * ``locate``           -- dataset.py:120-128 sample index -> (file, first input frame)
  * ``norm_constants``   -- dataset.py:73-117  per-field (diff, div) = mean over files of the per-file statistic, div + 1e-8
  * ``nearest_rows``     -- dataset.py:138-145 F.interpolate(mode="nearest") to (h // f, w // f): source index
                            floor(dst * float32(in / out)) clamped to in - 1 (ATen's nearest_idx)
  * ``clip``             -- dataset.py:130-182 input frames [start, start+tw), target frames [start+tw, start+2tw), per field
                            (x - diff) / div in float32, stacked (C, T, H, W) then permuted to (T, C, H, W)

Pinned by the reference's own test contract (data/tests/test_dataset.py: lengths and shapes for every combination of fields,
norm, downsample factor and window on samples/sample_{1,2}.hdf5) in tests/test_clip_supply.py; the values have no golden in the
reference, so the mirror is additionally compared element-wise with this restatement.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np


def num_samples(traj_lens: Sequence[int], start_time: int, time_window: int) -> int:
    return int(sum(t - start_time - 2 * time_window + 1 for t in traj_lens))


def locate(idx: int, traj_lens: Sequence[int], start_time: int, time_window: int) -> Tuple[int, int]:
    per = [t - start_time - 2 * time_window + 1 for t in traj_lens]
    cum = np.cumsum(per)
    file_idx = int(np.searchsorted(cum, idx, side="right"))
    start = idx + start_time - (int(cum[file_idx - 1]) if file_idx > 0 else 0)
    return file_idx, int(start)


def norm_constants(arrays: List[Dict[str, np.ndarray]], fields: Sequence[str], norm: str) -> Tuple[Dict[str, float], Dict[str, float]]:
    diff, div = {}, {}
    for f in fields:
        d, v = [], []
        for a in arrays:
            x = a[f]
            if norm == "std":
                d.append(x.mean()); v.append(x.std())
            elif norm == "minmax":
                d.append(x.min()); v.append(x.max() - x.min())
            elif norm == "tanh":
                d.append((x.max() + x.min()) / 2.0); v.append((x.max() - x.min()) / 2.0)
            elif norm == "none":
                d.append(0.0); v.append(1.0)
            else:
                raise ValueError(f"Unknown normalization type: {norm}")
        diff[f] = np.mean(d).item()
        div[f] = np.mean(v).item() + 1e-8
    return diff, div


def nearest_rows(n_in: int, n_out: int) -> np.ndarray:
    scale = np.float32(n_in) / np.float32(n_out)
    return np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64), n_in - 1)


def clip(arrays: List[Dict[str, np.ndarray]], idx: int, in_fields: Sequence[str], out_fields: Sequence[str], diff: Dict[str, float],
         div: Dict[str, float], start_time: int, time_window: int, downsample_factor: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    traj_lens = [a[in_fields[0]].shape[0] for a in arrays]
    fi, start = locate(idx, traj_lens, start_time, time_window)

    def grab(fields, s0):
        out = []
        for f in fields:
            x = np.asarray(arrays[fi][f][s0:s0 + time_window], dtype=np.float32)
            if downsample_factor > 1:
                _, h, w = x.shape
                x = x[:, nearest_rows(h, h // downsample_factor)][:, :, nearest_rows(w, w // downsample_factor)]
            out.append((x - np.float32(diff[f])) / np.float32(div[f]))
        return np.stack(out).transpose(1, 0, 2, 3).astype(np.float32)

    return grab(in_fields, start), grab(out_fields, start + time_window)
