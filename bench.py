#!/usr/bin/env python3
"""Headline benchmark: FiLMAViT-small training samples/sec, 4 fields, 16x192x192 clip, batch 8 per GPU (BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; step = forward + fused relative-L2 loss + backward + (N>1: RCCL bucketed gradient all-reduce
overlapped with backward) + fused AdamW.  Rank 0 prints ONE JSON line.  After the timed region the same step is run
twice more with per-launch HIP-event timing enabled inside the library to produce the `roofline` object for the
dominant kernel, and (N=1 only) the oracle is timed on the host cores for `cpu_baseline`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues round-robin; with the default of 4, the streams RCCL creates push this
# library's weight-gradient side stream onto the main stream's queue and the two serialise (measured with a 1-rank RCCL group:
# 479 samples/s at 4 queues, 577-584 at 2 / 3 / 6 / 8 / 16).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CFG = dict(input_fields=4, output_fields=4, patch_size=16, embed_dim=384, num_heads=6, processor_blocks=12,
           attn_scale=True, feat_scale=True, num_fluid_params=9)          # config/model_cfg/film_avit_small.yaml
BATCH, T, H, W = 8, 16, 192, 192
DROP_PATH = 0.2                     # film_avit_small.yaml: stochastic depth, rates np.linspace(0, 0.2, 12) over the blocks
FIELD_STATS = ((-2.37, 1.98), (0.0145, 0.081), (-0.07, 0.49), (0.055, 0.77))
# algorithmic work per sample at this shape (fwd+bwd)
FLOPS_PER_SAMPLE = 423.20e9
BYTES_PER_SAMPLE = 2.90e9          # 1,638 U x 1.769 MB (bf16 activations)
BYTES_PER_STEP_PARAMS = 0.69e9     # weights / grads / AdamW state, per step
PEAK_HBM_GBS = 8000.0
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def synthetic_batch(seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    def clip(s):
        gg = torch.Generator(device=device).manual_seed(s)
        t = torch.randn((BATCH, T, 4, H, W), device=device, generator=gg)
        for c, (mu, sd) in enumerate(FIELD_STATS):
            t[:, :, c].mul_(sd).add_(mu)
        t[:, :, 1].clamp_(min=0.0)
        return t
    return clip(seed), torch.randn((BATCH, 9), device=device, generator=g), clip(seed + 1)


def cpu_baseline(threads):
    """The oracle (CPU restatement of the reference path, pinned by tests/golden) as the reported CPU baseline."""
    from oracle import filmavit_ref as R, weights as Wt
    torch.set_num_threads(threads)
    shapes = Wt.param_shapes(**{k: v for k, v in CFG.items()})
    sd = {k: v.requires_grad_(True) for k, v in Wt.generate(shapes, seed=42).items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=2.5e-4, weight_decay=1e-2)
    bs = 1
    x = Wt.synthetic_clip(bs, T, 4, H, W, 42)
    y = Wt.synthetic_clip(bs, T, 4, H, W, 43)
    c = Wt.synthetic_fluid_params(bs, 9, 44)
    def step():
        opt.zero_grad(set_to_none=True)
        loss = R.lp_loss(R.filmavit_forward(sd, x, c, patch_size=16, num_heads=6), y)
        loss.backward()
        opt.step()
    step()
    log("cpu baseline warm-up step done")
    n, t0 = 0, time.perf_counter()
    while n < 3 or (time.perf_counter() - t0 < 10.0 and n < 12):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": bs * n / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"batch 1 clip 16x192x192x4ch fp32, 1 warm-up + {n} timed fwd+loss+bwd+AdamW steps of the oracle (torch CPU, {threads} threads)"}


def eager_gpu_baseline(dev, autocast):
    """the same restatement run eagerly on the MI355X through stock PyTorch-ROCm kernels -- the
    un-accelerated GPU comparator (batch 8, the bench shape, AdamW; fp32 or bf16 autocast).  Checker code, timed, never shipped."""
    from oracle import filmavit_ref as R, weights as Wt
    shapes = Wt.param_shapes(**{k: v for k, v in CFG.items()})
    sd = {k: v.to(dev).requires_grad_(True) for k, v in Wt.generate(shapes, seed=42).items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=2.5e-4, weight_decay=1e-2, fused=True)
    x, c, y = synthetic_batch(42, dev)
    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            pred = R.filmavit_forward(sd, x, c, patch_size=16, num_heads=6)
        loss = R.lp_loss(pred.float(), y)
        loss.backward()
        opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    n, t0 = 5, time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": BATCH * n / dt, "unit": "samples/s", "ms_per_step": dt / n * 1e3, "dtype": "bf16 autocast" if autocast else "f32",
            "kind": "oracle restatement, eager PyTorch-ROCm on the same GPU", "sample": f"batch {BATCH}, 2 warm-up + {n} timed steps"}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def host_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, int(os.environ.get("BENCH_CPU_THREADS", "16"))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager-gpu-baseline", action="store_true", help="also time the oracle eagerly on the GPU (fp32 and bf16 autocast)")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: whatever libraries print there (RCCL's version banner under NCCL_DEBUG=VERSION) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1"
    # BENCH_DEVICE / BENCH_BACKEND exist only to rehearse the multi-rank control flow on a one-GPU box (gloo, every rank on
    # one device); the driver's runs use one GPU per rank and RCCL ("nccl").
    if os.environ.get("BENCH_DEVICE") is not None:
        local = int(os.environ["BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"      # rehearsal: a 1-rank RCCL group with the gradient exchange switched on
    if world > 1 or force_dist:
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from bubbleformer_amd import _lib
    from bubbleformer_amd.models import get_model
    from bubbleformer_amd.trainer import TrainStep
    torch.manual_seed(42)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = get_model("filmavit", time_window=T, drop_path=DROP_PATH, compute_dtype=cdt, **CFG).to(dev).train()
    step = TrainStep(model, lr=2.5e-4, weight_decay=1e-2)            # config/optim_cfg/adamw.yaml
    if force_dist and os.environ.get("BENCH_FORCE_REDUCE", "1") == "1":
        step.reducer.enabled = True
    x, cond, y = synthetic_batch(42 + 1000 * rank, dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        loss = step(x, cond, y)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done, loss", float(loss))
    sync()
    log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(x, cond, y)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    final_loss = float(loss)
    log("timed region: %.3f s for %d steps" % (dt, args.steps))

    # ---- roofline leg: per-launch HIP-event timing (on the launch stream) over two more steps
    h = _lib.lib()
    h.bf_prof_enable(1)
    nprof = 2
    for _ in range(nprof):
        step(x, cond, y)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    n = h.bf_prof_report(buf, len(buf))
    h.bf_prof_enable(0)
    prof = json.loads(buf.value.decode()) if n > 0 else {}
    roofline = None
    if prof:
        tot_ms = sum(v["ms"] for v in prof.values())
        name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])          # dominant kernel = largest share of GPU time
        avg_ms = dom["ms"] / dom["calls"]
        flops, nbytes = dom["flops"] / dom["calls"], dom["bytes"] / dom["calls"]    # ALGORITHMIC work per launch
        peak_tf = PEAK_MFMA_TFLOPS[args.dtype]
        ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
        tflops = flops / (avg_ms * 1e-3) / 1e12
        gbs = nbytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "r01_kernels.json")          # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command
        if os.path.exists(tpath):
            for k in json.load(open(tpath))["kernels"]:
                if k["kernel"] == name and "hbm_bytes_per_launch" in k:
                    traffic = k["hbm_bytes_per_launch"]
        if flops > 0 and flops / max(nbytes, 1.0) >= ridge:
            roofline = {"kernel": name, "bound": "mfma", "achieved": tflops, "peak": peak_tf, "unit": "TFLOP/s", "frac": tflops / peak_tf}
        else:
            roofline = {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}
        roofline.update({"traffic": traffic, "avg_launch_ms": avg_ms, "launches_per_step": dom["calls"] / nprof,
                         "algorithmic_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops, "tflops": tflops,
                         "mfma_frac": tflops / peak_tf, "share_of_gpu_time": dom["ms"] / tot_ms,
                         "kernel_time_share": {k: round(v["ms"] / tot_ms, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:12]},
                         "kernel_avg_us": {k: [round(v["ms"] / v["calls"] * 1e3, 1), v["calls"] // nprof] for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:16]},
                         "gpu_kernel_ms_per_step": tot_ms / nprof})

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    value = BATCH * world * args.steps / dt
    per_gpu = value / world
    out = {
        "metric": "train samples/sec, FiLMAViT 16x192x192x4ch, bs=8 per GPU", "value": value, "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "FiLMAViT-small (E=384, 6 heads, 12 blocks, P=16, 9 fluid params) fwd + relative-L2 loss + bwd + AdamW, "
                               "4-field 16x192x192 clips, batch 8 per GPU (BASELINE.json configs[1])",
                   "global_batch": BATCH * world, "parallelism": f"dp{world}", "drop_path": DROP_PATH},
        "loss": final_loss,
        "roofline": roofline,
        "step_roofline": {"hbm_frac": per_gpu * (BYTES_PER_SAMPLE + BYTES_PER_STEP_PARAMS / BATCH) / (PEAK_HBM_GBS * 1e9),
                          "mfma_frac": per_gpu * FLOPS_PER_SAMPLE / (PEAK_MFMA_TFLOPS[args.dtype] * 1e12),
                          "algorithmic_GB_per_sample": BYTES_PER_SAMPLE / 1e9, "algorithmic_GFLOP_per_sample": FLOPS_PER_SAMPLE / 1e9},
    }
    if world == 1 and not args.no_cpu_baseline:
        log("timing the CPU baseline (oracle) on", host_threads(), "threads")
        out["cpu_baseline"] = cpu_baseline(host_threads())
    if world == 1 and args.eager_gpu_baseline:
        del step, model
        torch.cuda.empty_cache()
        out["eager_gpu_baseline"] = [eager_gpu_baseline(dev, True), eager_gpu_baseline(dev, False)]
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
