"""CPU, world_size 2, gloo: the data-parallel exchange (BucketReducer over FlatParams) averages gradients across
ranks exactly as one process over the concatenated batch would -- the N>1 path of bench.py, minus the GPU model."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = nn.Linear(6, 8)
        self.blocks = nn.ModuleList([nn.Sequential(nn.Linear(8, 8), nn.Tanh()) for _ in range(3)])
        self.debed = nn.Linear(8, 2)

    def forward(self, x):
        x = self.embed(x)
        for b in self.blocks:
            x = x + b(x)
        return self.debed(x)


def _worker(rank, world, port, out, bucket_dtype=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bubbleformer_amd.trainer import BucketReducer, FlatParams, stage_buckets
    torch.manual_seed(0)
    model = Toy()
    flat = FlatParams(model)
    buckets = stage_buckets(model, blocks_per_bucket=1)
    assert max(buckets) + 1 == 5          # embed, 3 blocks, debed
    assert max(stage_buckets(model, blocks_per_bucket=2)) + 1 == 4      # embed, blocks {0,1}, block 2, debed
    red = BucketReducer(flat, buckets, bucket_dtype=bucket_dtype)
    g = torch.Generator().manual_seed(123)
    xs = torch.randn(world * 4, 6, generator=g)
    ys = torch.randn(world * 4, 2, generator=g)
    flat.zero_grad()
    loss = ((model(xs[rank * 4:(rank + 1) * 4]) - ys[rank * 4:(rank + 1) * 4]) ** 2).mean()
    loss.backward()
    launched = list(red.launch_log)
    scale = red.wait()
    # gradient-ready order is the reverse of the forward (debed, blocks 2..0, embed): one collective per bucket, in that order
    assert launched == [4, 3, 2, 1, 0], launched
    if rank == 0:
        torch.save({"grad": flat.grad * scale, "xs": xs, "ys": ys}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_reducer_matches_single_process(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    blob = torch.load(out)
    from bubbleformer_amd.trainer import FlatParams
    torch.manual_seed(0)
    model = Toy()
    flat = FlatParams(model)
    loss = ((model(blob["xs"]) - blob["ys"]) ** 2).mean()
    loss.backward()
    assert torch.allclose(flat.grad, blob["grad"], rtol=1e-5, atol=1e-7)


def test_bf16_gradient_buckets(tmp_path):
    """BF_GRAD_BUCKET_DTYPE=bf16 / bucket_dtype=torch.bfloat16: the buckets travel as bf16 copies (half the bytes per xGMI link); the
    averaged gradient agrees with the fp32 exchange to bf16 rounding and stays fp32 in the flat buffer."""
    out = str(tmp_path / "g16.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, torch.bfloat16), nprocs=2, join=True)
    blob = torch.load(out)
    from bubbleformer_amd.trainer import FlatParams
    torch.manual_seed(0)
    model = Toy()
    flat = FlatParams(model)
    ((model(blob["xs"]) - blob["ys"]) ** 2).mean().backward()
    assert blob["grad"].dtype == torch.float32
    err = float((flat.grad - blob["grad"]).norm() / flat.grad.norm())
    assert 0 < err < 8e-3, err


def _sync_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bubbleformer_amd import trainer
    torch.manual_seed(100 + rank)          # every rank builds DIFFERENT weights ...
    model = Toy()
    ts = trainer.TrainStep.__new__(trainer.TrainStep)      # TrainStep's GPU-only pieces are not exercised here
    ts.flat = trainer.FlatParams(model)
    ts.m = torch.full_like(ts.flat.flat, float(rank + 1))
    ts.v = torch.full_like(ts.flat.flat, float(rank + 2))
    ts.step_no = 7 * (rank + 1)
    ts.sync_from_rank0()                   # ... and continues from rank 0's parameters, moments and step count
    torch.save({"flat": ts.flat.flat.clone(), "m": ts.m.clone(), "v": ts.v.clone(), "step": ts.step_no}, out + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_start_from_rank0(tmp_path):
    out = str(tmp_path / "s")
    mp.spawn(_sync_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = torch.load(out + "0"), torch.load(out + "1")
    assert torch.equal(a["flat"], b["flat"]) and torch.equal(a["m"], b["m"]) and torch.equal(a["v"], b["v"])
    assert a["step"] == b["step"] == 7 and float(a["m"][0]) == 1.0 and float(a["v"][0]) == 2.0


def test_flat_params_are_views():
    from bubbleformer_amd.trainer import FlatParams
    m = Toy()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatParams(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    flat.flat.mul_(2.0)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k] * 2)
