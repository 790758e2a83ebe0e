"""Clip supply: in-tree HDF5 reader, `BubbleForecast` mirror, device-resident clip gather.

Fixtures: tests/golden/samples/sample_{1,2}.hdf5 are the two 50x64x64 trajectories the reference's own dataset test reads
(bubbleformer/data/tests/test_dataset.py:31; data files, copied unchanged).  CPU tests: the reference test's contract for every
parameter combination, element-wise agreement with the oracle restatement (oracle/dataset_ref.py), the field statistics
SURVEY.md section 8c measured on the reference.  GPU tests: `DeviceClipStore.gather` is bit-identical to collating
`dataset[i]` on the host, and feeds a training step.
"""
import json
import os
import shutil

import numpy as np
import pytest
import torch

SAMPLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "samples")
FILES = [os.path.join(SAMPLES, "sample_1.hdf5"), os.path.join(SAMPLES, "sample_2.hdf5")]
ALL = ["dfun", "temperature", "velx", "vely"]


def _arrays():
    from bubbleformer_amd.data import hdf5_lite
    return [{k: f[k][...] for k in ALL} for f in (hdf5_lite.File(p) for p in FILES)]


def test_hdf5_reader_reads_the_sample_trajectories():
    from bubbleformer_amd.data import hdf5_lite
    f = hdf5_lite.File(FILES[0])
    assert sorted(f.keys()) == ALL
    stats = {"dfun": (-2.37, 1.98), "temperature": (0.0145, 0.081), "velx": (-0.07, 0.49), "vely": (0.055, 0.77)}
    for k in ALL:
        d = f[k]
        assert d.shape == (50, 64, 64) and d.dtype == np.float32 and len(d) == 50
        x = d[...]
        assert np.isfinite(x).all()
        assert abs(x.mean() - stats[k][0]) < 0.006 and abs(x.std() - stats[k][1]) < 0.006
        assert np.array_equal(d[5:9], x[5:9])
    with pytest.raises(KeyError):
        f["pressure"]
    with pytest.raises(hdf5_lite.Hdf5Error):
        hdf5_lite.File(os.path.abspath(__file__))            # not an HDF5 file


def test_hdf5_reader_chunked_and_compressed_layouts(tmp_path):
    """What h5py writes under `chunks=` / `compression="gzip"`: chunked layout (B-tree v1 index, edge chunks), deflate, shuffle,
    fletcher32.  tests/golden/hdf5/chunked.hdf5 was written by libhdf5 1.10.6 (make_chunked_fixture.c beside it); values are exact."""
    from bubbleformer_amd.data import BubbleForecast, hdf5_lite
    path = os.path.join(os.path.dirname(SAMPLES), "hdf5", "chunked.hdf5")
    f = hdf5_lite.File(path)
    t, y, x = np.meshgrid(np.arange(7), np.arange(10), np.arange(12), indexing="ij")
    for k, name in enumerate(ALL):
        want = ((131 * t + 17 * y + 3 * x + 1000 * k) * 0.25).astype(np.float32)
        assert f[name].shape == (7, 10, 12) and np.array_equal(f[name][...], want) and np.array_equal(f[name][2:5], want[2:5])
    yy, xx = np.meshgrid(np.arange(5), np.arange(6), indexing="ij")
    assert f["counts"].dtype == np.int16 and np.array_equal(f["counts"][...], (100 * yy - 7 * xx).astype(np.int16))
    d = BubbleForecast([path], norm="none", time_window=2, start_time=1)            # the dataset class reads such files unchanged
    d.normalize()
    a, b = d[0]
    assert tuple(a.shape) == (2, 4, 10, 12) and float(a[0, 0, 0, 0]) == 131 * 1 * 0.25
    # many small chunks: a multi-level chunk B-tree.  Made on the fly where the HDF5 tools exist (they do in the build container).
    repack = shutil.which("h5repack") or ("/opt/conda/bin/h5repack" if os.path.exists("/opt/conda/bin/h5repack") else None)
    if repack:
        import subprocess
        out = str(tmp_path / "s1_chunked.hdf5")
        subprocess.run([repack, "-l", "CHUNK=1x8x8", "-f", "SHUF", "-f", "GZIP=3", FILES[0], out], check=True)
        g, h = hdf5_lite.File(FILES[0]), hdf5_lite.File(out)
        for k in ALL:
            assert np.array_equal(g[k][...], h[k][...])


@pytest.mark.parametrize("input_fields", [["dfun"], ["temperature", "velx", "vely"], ALL])
@pytest.mark.parametrize("output_fields", [["temperature"], ["temperature", "velx", "vely"], ALL])
@pytest.mark.parametrize("norm", ["none", "std", "minmax", "tanh"])
@pytest.mark.parametrize("downsample_factor", [1, 2, 4])
@pytest.mark.parametrize("time_window", [5, 10])
def test_bubblemlforecastdataset(input_fields, output_fields, norm, downsample_factor, time_window):
    """The reference's own test (bubbleformer/data/tests/test_dataset.py:3-54) on the mirror."""
    from bubbleformer_amd.data import BubbleForecast
    dataset = BubbleForecast(filenames=FILES, input_fields=input_fields, output_fields=output_fields, norm=norm,
                             downsample_factor=downsample_factor, time_window=time_window, start_time=5)
    _, _ = dataset.normalize()
    sample = dataset[0]
    assert len(dataset) == 2 * (50 - 5 - 2 * time_window + 1)
    assert sample[0].shape == (time_window, len(input_fields), 64 // downsample_factor, 64 // downsample_factor)
    assert sample[1].shape == (time_window, len(output_fields), 64 // downsample_factor, 64 // downsample_factor)


@pytest.mark.parametrize("norm,ds,tw", [("std", 1, 8), ("minmax", 2, 5), ("tanh", 4, 10), ("none", 1, 16)])
def test_dataset_values_match_oracle_restatement(norm, ds, tw):
    from bubbleformer_amd.data import BubbleForecast
    from oracle import dataset_ref as O
    inf, outf = ["dfun", "temperature", "velx", "vely"], ["temperature", "vely"]
    d = BubbleForecast(FILES, inf, outf, norm=norm, downsample_factor=ds, time_window=tw, start_time=3)
    diff, div = d.normalize()
    arrays = _arrays()
    odiff, odiv = O.norm_constants(arrays, sorted(set(inf + outf)), norm)
    for k in odiff:
        assert diff[k] == pytest.approx(odiff[k], rel=1e-12, abs=1e-12) and div[k] == pytest.approx(odiv[k], rel=1e-12)
    n = len(d)
    assert n == O.num_samples([50, 50], 3, tw)
    per = n // 2
    for idx in (0, 1, per - 1, per, per + 1, n - 1):          # both sides of the file boundary
        assert d.locate(idx) == O.locate(idx, [50, 50], 3, tw)
        a, b = d[idx]
        oa, ob = O.clip(arrays, idx, inf, outf, diff, div, 3, tw, ds)
        assert a.dtype == torch.float32 and tuple(a.shape) == oa.shape
        assert np.array_equal(a.numpy(), oa) and np.array_equal(b.numpy(), ob)
    with pytest.raises(IndexError):
        d[n]


def _with_sidecars(tmp_path):
    files = []
    for i, src in enumerate(FILES):
        dst = tmp_path / os.path.basename(src)
        shutil.copy(src, dst)
        fp = {"inv_reynolds": 0.0042 + i, "cpgas": 0.83, "mugas": 0.023, "rhogas": 0.0083, "thcogas": 0.25, "stefan": 0.5298, "prandtl": 8.4,
              "heater": {"nucWaitTime": 0.4, "wallTemp": 1.0 + 0.1 * i}}
        with open(str(dst).replace(".hdf5", ".json"), "w", encoding="utf-8") as f:
            json.dump(fp, f)
        files.append(str(dst))
    return files


def test_fluid_params_sidecar(tmp_path):
    from bubbleformer_amd.data import BubbleForecast
    d = BubbleForecast(_with_sidecars(tmp_path), norm="none", time_window=4, start_time=2, return_fluid_params=True)
    d.normalize()
    a, b, fp = d[len(d) - 1]
    assert fp.shape == (9,) and fp.dtype == torch.float32
    assert float(fp[0]) == pytest.approx(1.0042) and float(fp[8]) == pytest.approx(1.1)        # second file's parameters


@pytest.mark.gpu
@pytest.mark.parametrize("norm,ds", [("std", 1), ("none", 2), ("tanh", 4)])
def test_device_gather_is_bit_identical_to_host_collate(tmp_path, norm, ds):
    from bubbleformer_amd.data import BubbleForecast
    d = BubbleForecast(_with_sidecars(tmp_path), ALL, ["temperature", "velx"], norm=norm, downsample_factor=ds, time_window=6, start_time=4,
                       return_fluid_params=True)
    d.normalize()
    store = d.device_store("cuda")
    n = len(d)
    idx = [0, n // 2 - 1, n // 2, n - 1, 7]
    inp, out, fp = store.gather(idx)
    host = [d[i] for i in idx]
    assert torch.equal(inp.cpu(), torch.stack([h[0] for h in host]))
    assert torch.equal(out.cpu(), torch.stack([h[1] for h in host]))
    assert torch.equal(fp.cpu(), torch.stack([h[2] for h in host]))
    # indices that already live on the device (a slice of a device-side shuffle): same batch, no host staging
    inp2, out2, fp2 = store.gather(torch.tensor(idx, device="cuda"))
    assert torch.equal(inp2, inp) and torch.equal(out2, out) and torch.equal(fp2, fp)
    with pytest.raises(IndexError):
        store.gather([n])
    # the batch entry point is ONE launch; the two single-clip launches with framework indexing in front give the same bits
    from bubbleformer_amd import ops
    first_all, file_all = store._index_tables()
    dev_idx = torch.tensor(idx, device="cuda")
    ho, wo = inp.shape[-2:]
    a = ops.clip_gather(store.frames, first_all[dev_idx], 0, d.time_window, store.in_tab, ho, wo)
    b = ops.clip_gather(store.frames, first_all[dev_idx], d.time_window, d.time_window, store.out_tab, ho, wo)
    assert torch.equal(a, inp) and torch.equal(b, out) and torch.equal(store.fluid[file_all[dev_idx]], fp)
    # an index out of range on the device path is clamped inside the kernel (it cannot be checked without a host synchronisation)
    bad = store.gather(torch.tensor([n + 5, -3], device="cuda"))
    ok = store.gather(torch.tensor([n - 1, 0], device="cuda"))
    assert all(torch.equal(x, y) for x, y in zip(bad, ok))


@pytest.mark.gpu
@pytest.mark.parametrize("norm", ["std", "minmax", "tanh", "none"])
def test_device_store_normalize_matches_the_host_pass(norm):
    """DeviceClipStore.normalize(): mean / std / min / max of every (field, file) by one reduction launch over the HBM-resident trajectories
    (bf_field_stats) against BubbleForecast.normalize() -- the reference's host pass over every full field of every file
    (bubbleformer/data/dataset.py:84-117) -- and against fp64 statistics of the same arrays; the gathered batch then uses them."""
    from bubbleformer_amd.data import BubbleForecast
    fields = ["dfun", "temperature", "velx", "vely"]
    host = BubbleForecast(FILES, fields, ["temperature", "vely"], norm=norm, time_window=6, start_time=4)
    hdiff, hdiv = host.normalize()
    d = BubbleForecast(FILES, fields, ["temperature", "vely"], norm=norm, time_window=6, start_time=4)
    store = d.device_store("cuda")
    diff, div = store.normalize()
    assert d.diff_terms is diff and d.div_terms is div
    arrays = _arrays()
    for k in fields:
        # min / max are exact; mean / std differ from numpy's float32 reductions by fp32 rounding only
        assert diff[k] == pytest.approx(hdiff[k], rel=2e-6, abs=1e-7) and div[k] == pytest.approx(hdiv[k], rel=2e-6)
        if norm == "std":
            xs = [np.asarray(a[k], dtype=np.float64) for a in arrays]
            assert diff[k] == pytest.approx(np.mean([x.mean() for x in xs]), rel=1e-12, abs=1e-14)
            assert div[k] == pytest.approx(np.mean([x.std() for x in xs]) + 1e-8, rel=1e-9)
        if norm == "minmax":
            assert diff[k] == pytest.approx(np.mean([float(a[k].min()) for a in arrays]), rel=1e-15)      # minima are exact
    again = store.normalize()
    assert again[0] == diff and again[1] == div                     # fixed summation order: the same bits every time
    inp, out = store.gather([0, 5])
    href = BubbleForecast(FILES, fields, ["temperature", "vely"], norm=norm, time_window=6, start_time=4)
    href.normalize(diff, div)
    assert torch.equal(inp.cpu(), torch.stack([href[0][0], href[5][0]])) and torch.equal(out.cpu(), torch.stack([href[0][1], href[5][1]]))


def test_dataset_over_in_memory_trajectories_equals_the_file_backed_one(tmp_path):
    """BubbleForecast.from_arrays (bench.py's clip-supply leg tiles the sample trajectories in memory): same length, same samples as
    the dataset over the files the arrays were read from."""
    import numpy as np
    from bubbleformer_amd.data import BubbleForecast, hdf5_lite
    files = _with_sidecars(tmp_path)
    kw = dict(norm="std", time_window=5, start_time=3)
    a = BubbleForecast(files, return_fluid_params=True, **kw)
    a.normalize()
    trajs, fluid = [], []
    for f in files:
        h = hdf5_lite.File(f)
        trajs.append({k: np.asarray(h[k].array(), dtype=np.float32) for k in ALL})
        h.close()
    b = BubbleForecast.from_arrays(trajs, a.fluid_params, **kw)
    b.normalize()
    assert len(a) == len(b) and a.diff_terms == b.diff_terms and a.div_terms == b.div_terms
    for i in (0, len(a) // 2, len(a) - 1):
        for u, v in zip(a[i], b[i]):
            assert torch.equal(u, v)


@pytest.mark.gpu
def test_training_step_fed_from_the_device_store(tmp_path):
    """plumbing (BASELINE configs[0] flavour): clips gathered on the device from the sample trajectories drive the native step."""
    from bubbleformer_amd.data import BubbleForecast
    from bubbleformer_amd.models import get_model
    from bubbleformer_amd.trainer import TrainStep
    torch.manual_seed(0)
    d = BubbleForecast(_with_sidecars(tmp_path), norm="std", time_window=8, start_time=5, return_fluid_params=True)
    d.normalize()
    store = d.device_store("cuda")
    model = get_model("filmavit", input_fields=4, output_fields=4, time_window=8, patch_size=16, embed_dim=96, num_heads=2, processor_blocks=2,
                      num_fluid_params=9, drop_path=0.0, compute_dtype=torch.bfloat16).cuda().train()
    step = TrainStep(model, lr=1e-3)
    x, y, c = store.gather([0, 30])
    assert x.shape == (2, 8, 4, 64, 64) and c.shape == (2, 9)
    losses = [float(step(x, c, y)) for _ in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
