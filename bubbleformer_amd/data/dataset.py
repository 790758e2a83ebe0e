"""Clip supply for the training step.

``BubbleForecast`` keeps the reference class's constructor, ``__len__``, ``normalize`` and ``__getitem__`` contract
(bubbleformer/data/dataset.py:17-184) but reads the trajectory files with the in-tree HDF5 reader (no h5py), and adds the
MI355X-side path: ``device_store()`` puts every trajectory in HBM once (a BubbleML study is a few GB; one GPU has 288 GB) and
``DeviceClipStore.gather`` builds a whole batch of normalised (input, target) clips with ONE HIP kernel (`bf_clip_gather`,
csrc/patch.hip) -- no host-side slicing, stacking or H2D copy per step.
"""
import json
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from . import hdf5_lite

FLUID_PARAM_KEYS = ("inv_reynolds", "cpgas", "mugas", "rhogas", "thcogas", "stefan", "prandtl")       # dataset.py:165-178, then heater.*


def _fluid_vector(fp: dict) -> List[float]:
    return [fp[k] for k in FLUID_PARAM_KEYS] + [fp["heater"]["nucWaitTime"], fp["heater"]["wallTemp"]]


# (offset, scale) of one file's field from its {mean, std, min, max}; the dataset's constants are the means of these over the files
_NORMS = {
    "none": lambda st: (0.0, 1.0),
    "std": lambda st: (st["mean"], st["std"]),
    "minmax": lambda st: (st["min"], st["max"] - st["min"]),
    "tanh": lambda st: ((st["max"] + st["min"]) / 2.0, (st["max"] - st["min"]) / 2.0),
}


def _host_constants(x: Optional[np.ndarray], norm: str) -> Tuple[float, float]:
    if norm not in _NORMS:
        raise ValueError(f"Unknown normalization type: {norm}")
    if x is None:
        return _NORMS[norm]({})
    need = {"std": ("mean", "std"), "minmax": ("min", "max"), "tanh": ("min", "max")}[norm]
    return _NORMS[norm]({k: getattr(x, k)() for k in need})      # numpy's own reductions in the array's dtype, as the reference's h5py arrays give


def _combine_constants(per_file: Dict[str, List[Tuple[float, float]]]) -> Tuple[Dict, Dict]:
    diff = {f: np.mean([c[0] for c in cs]).item() for f, cs in per_file.items()}
    div = {f: np.mean([c[1] for c in cs]).item() + 1e-8 for f, cs in per_file.items()}
    return diff, div


class BubbleForecast(Dataset):
    """Dataset class for time series forecasting on the BubbleML dataset (reference: bubbleformer/data/dataset.py:17)."""

    def __init__(self, filenames: List[str], input_fields: Optional[List[str]] = None, output_fields: Optional[List[str]] = None,
                 norm: str = "none", downsample_factor: int = 1, time_window: int = 16, start_time: int = 50,
                 return_fluid_params: bool = False):
        super().__init__()
        self.filenames = filenames
        self.input_fields = input_fields if input_fields is not None else ["dfun", "temperature", "velx", "vely"]
        self.output_fields = output_fields if output_fields is not None else ["dfun", "temperature", "velx", "vely"]
        self.norm = norm
        self.downsample_factor = downsample_factor
        self.time_window = time_window
        self.start_time = start_time
        self.data = [hdf5_lite.File(filename, "r") for filename in filenames]
        self.num_trajs = [1 for _ in self.data]
        self.traj_lens = [f[self.input_fields[0]].shape[0] for f in self.data]
        self.input_num_fields = len(self.input_fields)
        self.output_num_fields = len(self.output_fields)
        self.fields = list(set(self.input_fields + self.output_fields))
        self.diff_terms = {k: [] for k in self.fields}
        self.div_terms = {k: [] for k in self.fields}
        self.return_fluid_params = return_fluid_params
        if self.return_fluid_params:
            self.fluid_params = []
            for fname in filenames:
                with open(fname.replace(".hdf5", ".json"), "r", encoding="utf-8") as f:
                    self.fluid_params.append(json.load(f))

    @classmethod
    def from_arrays(cls, trajectories: Sequence[Dict[str, np.ndarray]], fluid_params: Optional[Sequence[dict]] = None, **kw) -> "BubbleForecast":
        """The same dataset over trajectories that are already in memory: one dict {field: array [frames][H][W]} per simulation in
        place of one HDF5 file each (synthetic or pre-loaded studies; bench.py's clip-supply leg).  Keyword arguments as the constructor's."""
        self = cls([], **{k: v for k, v in kw.items() if k != "return_fluid_params"})
        self.data = [dict(t) for t in trajectories]
        self.num_trajs = [1 for _ in self.data]
        self.traj_lens = [t[self.input_fields[0]].shape[0] for t in self.data]
        self.return_fluid_params = fluid_params is not None
        if fluid_params is not None:
            if len(fluid_params) != len(self.data):
                raise ValueError("one fluid-parameter record per trajectory")
            self.fluid_params = list(fluid_params)
        return self

    # ---------------------------------------------------------------- reference contract
    def _per_traj(self) -> List[int]:
        return [n * (t - self.start_time - 2 * self.time_window + 1) for n, t in zip(self.num_trajs, self.traj_lens)]

    def __len__(self) -> int:
        return sum(self._per_traj())

    def normalize(self, diff_terms: Optional[Dict] = None, div_terms: Optional[Dict] = None) -> Tuple[Dict, Dict]:
        """Channel-wise normalisation constants (the reference's contract, dataset.py:73-117): per field, the mean over files of that file's
        (offset, scale) under ``self.norm``, with 1e-8 added to the scale; or the constants handed in (a validation set takes the training
        set's).  This is the HOST path (numpy over the whole field of every file, as the reference computes them): what a CPU-only caller and
        ``norm="none"`` use.  With the trajectories resident in HBM, ``device_store(...).normalize()`` gets the same constants from one
        reduction launch instead of a host pass over every file."""
        if diff_terms is None and div_terms is None:
            per_file = {f: [_host_constants(None if self.norm == "none" else h5[f][...], self.norm) for h5 in self.data] for f in self.fields}
            diff_terms, div_terms = _combine_constants(per_file)
        self.diff_terms = diff_terms
        self.div_terms = div_terms
        return self.diff_terms, self.div_terms

    def locate(self, idx: int) -> Tuple[int, int]:
        """Sample index -> (file index, first input frame) (dataset.py:120-128)."""
        cumulative = np.cumsum(self._per_traj())
        file_idx = int(np.searchsorted(cumulative, idx, side="right"))
        start = idx + self.start_time - (int(cumulative[file_idx - 1]) if file_idx > 0 else 0)
        return file_idx, int(start)

    def _field_clip(self, file_idx: int, field: str, sl: slice) -> torch.Tensor:
        item = torch.from_numpy(np.array(self.data[file_idx][field][sl], dtype=np.float32))
        if self.downsample_factor > 1:
            _, h, w = item.shape
            item = torch.nn.functional.interpolate(item.unsqueeze(1), size=(h // self.downsample_factor, w // self.downsample_factor),
                                                   mode="nearest").squeeze(1)
        return (item - self.diff_terms[field]) / self.div_terms[field]

    def __getitem__(self, idx: int):
        file_idx, start = self.locate(idx)
        tw = self.time_window
        inp = torch.stack([self._field_clip(file_idx, f, slice(start, start + tw)) for f in self.input_fields])          # (C, T, H, W)
        out = torch.stack([self._field_clip(file_idx, f, slice(start + tw, start + 2 * tw)) for f in self.output_fields])
        if self.return_fluid_params:
            fp = torch.tensor(_fluid_vector(self.fluid_params[file_idx]), dtype=torch.float32)
            return inp.float().permute(1, 0, 2, 3), out.float().permute(1, 0, 2, 3), fp
        return inp.float().permute(1, 0, 2, 3), out.float().permute(1, 0, 2, 3)

    # ---------------------------------------------------------------- device-resident path
    def device_store(self, device) -> "DeviceClipStore":
        return DeviceClipStore(self, device)


class DeviceClipStore:
    """Every trajectory of the dataset resident in HBM as one fp32 tensor ``[field][frame][H][W]`` (files concatenated along the
    frame axis); ``gather(indices)`` returns the batch the reference's DataLoader would have collated from ``dataset[i]``:
    ``(B, T, C_in, H', W')``, ``(B, T, C_out, H', W')`` [, ``(B, 9)`` fluid parameters] -- built by one kernel launch."""

    def __init__(self, ds: BubbleForecast, device):
        self.ds = ds
        self.device = torch.empty(0, device=device).device      # indexed ("cuda" -> "cuda:0"): gather() compares index tensors' devices with it
        shapes = {tuple(f[ds.fields[0]].shape[1:]) for f in ds.data}
        if len(shapes) != 1:
            raise ValueError(f"device store needs one spatial resolution per dataset, got {sorted(shapes)}")
        (self.H, self.W), = shapes
        self.fields = sorted(ds.fields)
        self.frame0 = np.concatenate([[0], np.cumsum(ds.traj_lens)]).astype(np.int64)       # first frame of each file on the frame axis
        total = int(self.frame0[-1])
        host = torch.empty((len(self.fields), total, self.H, self.W), dtype=torch.float32)
        for fi, f in enumerate(ds.data):
            for ci, name in enumerate(self.fields):
                host[ci, self.frame0[fi]:self.frame0[fi + 1]] = torch.from_numpy(np.array(f[name][...], dtype=np.float32))
        self.frames = host.to(self.device)
        self.fluid = None
        if ds.return_fluid_params:
            self.fluid = torch.tensor([_fluid_vector(fp) for fp in ds.fluid_params], dtype=torch.float32, device=self.device)
        self._tables()

    def normalize(self) -> Tuple[Dict, Dict]:
        """``BubbleForecast.normalize()`` from the trajectories in HBM: {sum, sum of squares, min, max} of every (field, file) segment by ONE
        reduction launch (bf_field_stats, fp64 accumulation in a fixed order) instead of a host pass over every file, then the reference's
        rule per file and the mean over files (dataset.py:84-117).  Sets the dataset's constants and this store's tables; returns them.
        Against the host path the constants agree to fp32 rounding of the host's own float32 reductions (~1e-7 relative)."""
        from .. import _lib as L
        from ..ops import _p, _stream
        ds = self.ds
        if ds.norm == "none" or self.device.type != "cuda":
            out = ds.normalize()
            self._tables()
            return out
        if ds.norm not in _NORMS:
            raise ValueError(f"Unknown normalization type: {ds.norm}")
        nfile, px = len(ds.traj_lens), self.H * self.W
        total = int(self.frame0[-1])
        begin = [(ci * total + int(self.frame0[fi])) * px for ci in range(len(self.fields)) for fi in range(nfile)]
        length = [int(ds.traj_lens[fi]) * px for _ in self.fields for fi in range(nfile)]
        nseg = len(begin)
        seg_b = torch.tensor(begin, dtype=torch.int64, device=self.device)
        seg_n = torch.tensor(length, dtype=torch.int64, device=self.device)
        out = torch.empty(nseg, 4, dtype=torch.float64, device=self.device)
        ws = torch.empty(L.lib().bf_field_stats_ws_doubles(nseg), dtype=torch.float64, device=self.device)
        L.check(L.lib().bf_field_stats(_p(self.frames), _p(seg_b), _p(seg_n), nseg, _p(out), _p(ws), _stream()), "bf_field_stats")
        st = out.cpu().view(len(self.fields), nfile, 4).numpy()
        n = np.asarray(length, dtype=np.float64).reshape(len(self.fields), nfile)
        mean = st[..., 0] / n
        var = np.maximum(st[..., 1] / n - mean * mean, 0.0)
        per_file = {name: [_NORMS[ds.norm]({"mean": mean[ci, fi], "std": float(np.sqrt(var[ci, fi])), "min": st[ci, fi, 2], "max": st[ci, fi, 3]})
                           for fi in range(nfile)] for ci, name in enumerate(self.fields)}
        ds.diff_terms, ds.div_terms = _combine_constants(per_file)
        self._tables()
        return ds.diff_terms, ds.div_terms

    def _tables(self):
        """Per-output-channel field index and normalisation constants; call again after ``ds.normalize()`` changed them."""
        ds = self.ds
        def tab(names):
            ids = torch.tensor([self.fields.index(n) for n in names], dtype=torch.int32, device=self.device)
            diff = torch.tensor([float(ds.diff_terms[n]) if not isinstance(ds.diff_terms[n], list) else 0.0 for n in names], dtype=torch.float32, device=self.device)
            div = torch.tensor([float(ds.div_terms[n]) if not isinstance(ds.div_terms[n], list) else 1.0 for n in names], dtype=torch.float32, device=self.device)
            return ids, diff, div
        self.in_tab, self.out_tab = tab(ds.input_fields), tab(ds.output_fields)

    def _index_tables(self):
        """Per sample index: absolute first input frame and file index, resident on the device (a few KB): gather() then needs no host
        work per step beyond the launch."""
        ds = self.ds
        key = (len(ds), ds.time_window, ds.start_time)
        if getattr(self, "_tab_key", None) != key:
            loc = [ds.locate(i) for i in range(len(ds))]
            self._first_all = torch.tensor([int(self.frame0[fi]) + st for fi, st in loc], dtype=torch.int64, device=self.device)
            self._file_all = torch.tensor([fi for fi, _ in loc], dtype=torch.int64, device=self.device)
            self._tab_key = key
        return self._first_all, self._file_all

    def gather(self, indices):
        """indices: sample indices as a sequence of ints or an integer tensor -- a DEVICE tensor (e.g. a slice of
        ``torch.randperm(len(ds), device=...)``) keeps the whole step free of host-device synchronisation; host indices are staged
        through pinned memory."""
        from .. import ops
        ds = self.ds
        first_all, file_all = self._index_tables()
        if isinstance(indices, torch.Tensor) and indices.device == self.device:
            idx = indices.to(torch.int64)
        else:
            host = torch.as_tensor(np.asarray(indices, dtype=np.int64) if not isinstance(indices, torch.Tensor) else indices.to(torch.int64).cpu())
            if host.numel() and (int(host.min()) < 0 or int(host.max()) >= len(ds)):
                raise IndexError("sample index out of range")
            idx = host.pin_memory().to(self.device, non_blocking=True) if self.device.type == "cuda" else host.to(self.device)
        tw, f = ds.time_window, ds.downsample_factor
        ho, wo = (self.H // f, self.W // f) if f > 1 else (self.H, self.W)
        if self.device.type == "cuda":       # one launch: both clips and the fluid rows, indexed by sample number on the device
            inp, out, fl = ops.clip_gather_batch(self.frames, idx.contiguous(), first_all, tw, self.in_tab, self.out_tab, ho, wo,
                                                 self.fluid, file_all if self.fluid is not None else None)
            return (inp, out, fl) if self.fluid is not None else (inp, out)
        first = first_all[idx]
        inp = ops.clip_gather(self.frames, first, 0, tw, self.in_tab, ho, wo)
        out = ops.clip_gather(self.frames, first, tw, tw, self.out_tab, ho, wo)
        if self.fluid is not None:
            return inp, out, self.fluid[file_all[idx]]
        return inp, out
