"""GPU, world size 2 (gloo, both ranks on cuda:0): the native training step under data parallelism.

bench.py's N > 1 path: each rank runs TrainStep on its own clips; gradient buckets are all-reduced while the backward is still
being enqueued, and (bf_side_defer) a stage's weight-gradient GEMMs may still be running on the library's side stream when the
stage returns -- a bucket is therefore only reduced once the NEXT bucket is complete.  The test: after one step on two half
batches the summed gradients (x 1/world) equal those of ONE process stepping on the whole batch
(LpLoss means over the batch: utils/losses.py:60-65, so equal per-rank batches give mean-of-means = global mean).
fp32 compute, tolerance 1e-4 (only the accumulation order differs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(input_fields=3, output_fields=3, patch_size=4, embed_dim=128, num_heads=2, processor_blocks=3, num_fluid_params=5)
B, T, H, W = 2, 4, 48, 48            # per rank; 12 x 12 = 144 tokens per frame (the whole-frame GEMM tiles are bf16-only, exercised below too)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(world):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(world * B, T, 3, H, W, generator=g)
    y = torch.randn(world * B, T, 3, H, W, generator=g)
    c = torch.randn(world * B, 5, generator=g)
    return x, c, y


def _model(dtype):
    from bubbleformer_amd.models import get_model
    torch.manual_seed(3)
    m = get_model("filmavit", time_window=T, drop_path=0.0, compute_dtype=dtype, **CFG)
    with torch.no_grad():             # layer scales start at 1e-6 in the reference: give every branch weight so all gradients matter
        for k, p in m.named_parameters():
            if "gamma" in k:
                p.fill_(0.5)
    return m.cuda().train()


def _step(model, x, c, y):
    from bubbleformer_amd.trainer import TrainStep
    step = TrainStep(model, lr=1e-3, weight_decay=1e-2)
    loss = step(x.cuda(), c.cuda(), y.cuda())
    torch.cuda.synchronize()
    return float(loss), step


def _worker(rank, world, port, dtype_name, out, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = getattr(torch, dtype_name)
    x, c, y = _data(world)
    sl = slice(rank * B, (rank + 1) * B)
    model = _model(dtype)
    loss, step = _step(model, x[sl], c[sl], y[sl])
    # one collective per bucket, launched in gradient-ready order: debed first, then the processor-block groups last to first, then
    # (film_embed +) embed -- the reverse of the forward; bucket ids follow registration (= forward) order
    nb = max(step.reducer.bucket_of_ptr.values()) + 1
    log = step.reducer.launch_log
    assert sorted(log) == list(range(nb)) and len(log) == nb, log
    names = {}
    for (k, p_) in model.named_parameters():
        names.setdefault(step.reducer.bucket_of_ptr[p_.data_ptr()], k.split(".")[0])
    order = [names[b] for b in log]
    assert order[0] == "debed" and order[-1] in ("embed", "film_embed") and all(n == "blocks" for n in order[1:-2]), order
    blocks = [b for b in log if names[b] == "blocks"]
    assert blocks == sorted(blocks, reverse=True), log
    if rank == 0:
        torch.save({"grad": (step.flat.grad / world).cpu(), "flat": step.flat.flat.detach().cpu(), "loss": loss}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_two_rank_step_matches_one_process_on_the_whole_batch(tmp_path, dtype_name):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), dtype_name, out), nprocs=2, join=True)
    blob = torch.load(out)
    dtype = getattr(torch, dtype_name)
    x, c, y = _data(2)
    model = _model(dtype)
    _, step = _step(model, x, c, y)
    g1, g2 = step.flat.grad.cpu().double(), blob["grad"].double()
    tol = 1e-4 if dtype == torch.float32 else 3e-2      # bf16: the two runs round different partial sums
    err = float((g1 - g2).norm() / g1.norm())
    if err >= tol:                                      # say where: one line per parameter that is off
        for (k, p_), o in zip(model.named_parameters(), step.flat.offsets):
            a, b = g1[o:o + p_.numel()], g2[o:o + p_.numel()]
            e = float((a - b).norm() / (a.norm() + 1e-30))
            if e > tol:
                print(f"  {k}: rel {e:.3e} |g| {float(a.norm()):.3e}")
    assert err < tol, err
    assert g1.abs().max() > 0 and torch.isfinite(blob["flat"]).all()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI); the one-GPU box runs the gloo variant above")
def test_two_rank_step_over_rccl_one_gpu_per_rank(tmp_path):
    """The same comparison with backend "nccl" (= RCCL), one GPU per rank: pins the one-bucket-late launch order and the side-stream
    joins against RCCL's own stream on hardware."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), "float32", out, "nccl"), nprocs=2, join=True)
    blob = torch.load(out)
    x, c, y = _data(2)
    model = _model(torch.float32)
    _, step = _step(model, x, c, y)
    g1, g2 = step.flat.grad.cpu().double(), blob["grad"].double()
    assert float((g1 - g2).norm() / g1.norm()) < 1e-4


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, c, y = _data(1)
    model = _model(torch.float32)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])      # what Lightning's strategy="ddp" wraps the module in
    pred = ddp(x.cuda(), c.cuda())
    yy = y.cuda()
    loss = (((pred - yy) ** 2).sum(dim=(-1, -2)).sqrt() / (yy ** 2).sum(dim=(-1, -2)).sqrt()).mean(0).mean(0).sum()
    loss.backward()
    torch.cuda.synchronize()
    torch.save({k: p.grad.detach().cpu() for k, p in model.named_parameters()}, out)
    dist.destroy_process_group()


def test_reference_style_ddp_wrapper_in_autograd_mode(tmp_path):
    """outside TrainStep the stages hand their parameter gradients back to autograd, so an UNMODIFIED
    `DistributedDataParallel` wrapper (scripts/train.py:158-172) sees per-parameter hooks fire and produces the gradients of the
    plain backward (1-rank gloo group on this box)."""
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_ddp_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    x, c, y = _data(1)
    model = _model(torch.float32)
    loss, _ = model.forward_loss(x.cuda(), c.cuda(), y.cuda())
    loss.backward()
    from tests.helpers import structurally_zero
    gscale = max(float(p.grad.norm()) for p in model.parameters())
    for k, p in model.named_parameters():
        ref = p.grad.detach().cpu()
        assert got[k] is not None, k
        if structurally_zero(k):                  # exact-zero true gradient: rounding noise on both sides
            assert float(got[k].norm()) <= 1e-5 * gscale, k
        else:
            assert float((got[k].double() - ref.double()).norm()) <= 1e-4 * float(ref.double().norm()), k


# ---------------------------------------------------------------------------------------------- RCCL stream ordering on ONE GPU
SMALL = dict(input_fields=4, output_fields=4, patch_size=16, embed_dim=384, num_heads=6, processor_blocks=12, num_fluid_params=9)
EXACT = ("input_head.weight", "input_head.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")


def _bench_model():
    from bubbleformer_amd.models import get_model
    from oracle import weights as Wt
    m = get_model("filmavit", time_window=16, drop_path=0.0, compute_dtype=torch.bfloat16, **SMALL)
    m.load_state_dict(Wt.generate(Wt.param_shapes(**SMALL), seed=31))
    return m.cuda().train()


def _rccl_one_rank_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    warm = torch.ones(1024, device="cuda")
    dist.all_reduce(warm)                          # communicator set-up (tens of ms) happens here, not inside a timed step
    torch.cuda.synchronize()
    from bubbleformer_amd.trainer import TrainStep
    from oracle import weights as Wt
    Bb = 8                                         # the bench batch: the device, not the host's enqueue rate, paces the step
    x = Wt.synthetic_clip(Bb, 16, 4, 192, 192, 131).cuda()
    y = Wt.synthetic_clip(Bb, 16, 4, 192, 192, 231).cuda()
    c = Wt.synthetic_fluid_params(Bb, 9, 331).cuda()
    res = {}
    for mode in ("warm", "off", "fp32", "bf16", "fp32_again"):      # "warm": RCCL's first collectives on large buffers set channels up (tens of ms); not compared
        model = _bench_model()
        step = TrainStep(model, lr=1e-3, weight_decay=1e-2)
        if mode not in ("off",):
            step.reducer.enabled = True            # a 1-rank group: the all-reduce is the identity, so ANY difference is an ordering bug
            if mode == "bf16":
                step.reducer.bucket_dtype = torch.bfloat16
        marks = []
        if mode not in ("off",):                   # an event on the caller's stream at every bucket launch, one after the side-stream join
            orig = step.reducer._launch
            def launch(b, orig=orig, marks=marks):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((b, e))
                orig(b)
            step.reducer._launch = launch
        step.lr = 0.0                              # warm-up step of THIS model with lr = 0 (AdamW's decay is lr-scaled too: weights unchanged, moments
        step(x, c, y)                              # advance identically in every mode): first-call host costs stay out of the timed step
        step.lr = 1e-3
        del marks[:]
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t0.record()
        loss = step(x, c, y)
        t1 = torch.cuda.Event(enable_timing=True); t1.record()
        torch.cuda.synchronize()
        names = [k for k, _ in model.named_parameters()]
        res[mode] = {"loss": float(loss), "grad": step.flat.grad.detach().cpu().clone(), "flat": step.flat.flat.detach().cpu().clone(),
                     "offsets": list(step.flat.offsets), "names": names, "numel": [p.numel() for p in step.flat.params],
                     "log": list(step.reducer.launch_log), "launch_ms": [(b, t0.elapsed_time(e)) for b, e in marks], "step_ms": t0.elapsed_time(t1)}
    torch.save(res, out)
    dist.destroy_process_group()


def test_rccl_one_rank_exchange_is_ordered_after_the_side_stream(tmp_path):
    """One GPU, backend "nccl" (= RCCL), world size 1, the gradient exchange switched ON at the bench model's width, depth and batch, after a warm-up step with lr = 0:
    * fp32 buckets: every all-reduce is the identity, so the step's gradients and post-step weights must be BIT-identical to a step with
      the exchange off on the families the kernels produce deterministically (the trunk's conv / Linear weights and biases; the others
      to rounding) -- an all-reduce that ran before the library's side stream had written a bucket would copy stale sums back;
    * bf16 buckets: a bucket is cast on the caller's stream at launch and copied back at the end: the result must equal bf16(reference
      gradient) bit for bit on the same families -- this catches a bucket handed to the exchange before its last weight-gradient GEMM
      (deferred joins: a bucket is launched one bucket late) even where the in-place identity would hide it;
    * one collective per bucket, gradient-ready order; the launches (events on the caller's stream) are spread over the backward: the
      first at least 3 ms before the end of the step, all but the last two at least 0.4 ms before it -- they travel under the remaining
      stages' kernels."""
    out = str(tmp_path / "one_rank.pt")
    mp.spawn(_rccl_one_rank_worker, args=(_free_port(), out), nprocs=1, join=True)
    r = torch.load(out)
    ref = r["off"]
    def fam(res, k):
        i = res["names"].index(k)
        o, n = res["offsets"][i], res["numel"][i]
        return res["grad"][o:o + n], res["flat"][o:o + n]
    n_exact = 0
    for mode in ("fp32", "fp32_again", "bf16"):
        got = r[mode]
        assert got["loss"] == ref["loss"]
        for k in ref["names"]:
            g_ref, w_ref = fam(ref, k)
            g, w = fam(got, k)
            exact = k.startswith("blocks.") and k.endswith(EXACT)
            if mode == "bf16":
                want = g_ref.to(torch.bfloat16).float()
                if exact:
                    assert torch.equal(g, want), (mode, k)
                else:
                    assert float((g - want).norm()) <= 2e-2 * float(want.norm()) + 1e-12, (mode, k)
            elif exact:
                assert torch.equal(g, g_ref) and torch.equal(w, w_ref), (mode, k)
                n_exact += 1
            else:
                assert float((g - g_ref).norm()) <= 1e-4 * float(g_ref.norm()) + 1e-12, (mode, k)
                assert float((w - w_ref).norm()) <= 1e-5 * float(w_ref.norm()) + 1e-12, (mode, k)
        nb = len(set(got["log"]))
        assert sorted(got["log"]) == list(range(nb)) and len(got["log"]) == nb, got["log"]
        when = [ms for _, ms in got["launch_ms"]]
        # measured from the END of the step (a slow host in front of the backward must not matter): the first launch at least 3 ms before it (the
        # backward is ~6.5 ms), all but the last two at least 0.4 ms before it
        assert when == sorted(when) and got["step_ms"] - when[0] > 3.0 and got["step_ms"] - when[-3] > 0.4, (got["launch_ms"], got["step_ms"])
    assert n_exact == 2 * 12 * (2 * 2 + 4)
