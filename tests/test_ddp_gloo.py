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


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bubbleformer_amd.trainer import BucketReducer, FlatParams, stage_buckets
    torch.manual_seed(0)
    model = Toy()
    flat = FlatParams(model)
    buckets = stage_buckets(model, blocks_per_bucket=1)
    assert max(buckets) + 1 == 5          # embed, 3 blocks, debed
    assert max(stage_buckets(model, blocks_per_bucket=2)) + 1 == 4      # embed, blocks {0,1}, block 2, debed
    red = BucketReducer(flat, buckets)
    g = torch.Generator().manual_seed(123)
    xs = torch.randn(world * 4, 6, generator=g)
    ys = torch.randn(world * 4, 2, generator=g)
    flat.zero_grad()
    loss = ((model(xs[rank * 4:(rank + 1) * 4]) - ys[rank * 4:(rank + 1) * 4]) ** 2).mean()
    loss.backward()
    scale = red.wait()
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
