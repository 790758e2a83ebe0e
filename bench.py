#!/usr/bin/env python3
"""Headline benchmark: FiLMAViT-small training samples/sec, 4 fields, 16x192x192 clip, batch 8 per GPU (BASELINE.json configs[1]).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config configs1|configs3|configs4]

  --gpus N > 1 without a launcher: this script starts N rank processes itself (one per GPU, RCCL) BEFORE anything touches the GPU
  and relays rank 0's JSON line.  Launched externally (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...) it
  reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment as before.

  --config configs1  (default) the headline: training step at 16x192x192, batch 8 per GPU
           configs3  BASELINE configs[3]: the long-aspect 32x384x192 training step (batch 4 per GPU: 36,864 tokens)
           configs4  BASELINE configs[4]: 200-step autoregressive rollout at batch 1, forward captured in a HIP graph; --record PATH
                     writes the per-step record (relative L2, Eikonal residual, heater heat flux) there
  --data device-store  every timed step draws its batch from HBM-resident trajectories (DeviceClipStore.gather = bf_clip_gather) instead of
                     re-using one resident batch; the default run reports that rate too, beside the headline, as `clip_supply`

One process per GPU; training step = forward + fused relative-L2 loss + backward + (N>1: RCCL bucketed gradient all-reduce
overlapped with backward) + fused AdamW.  Rank 0 prints ONE JSON line.  After the timed region the same step is run twice more with
per-launch HIP-event timing enabled inside the library to produce the `roofline` object for the dominant kernel, and (N=1 only) the
oracle is timed on the host cores for `cpu_baseline`.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues round-robin; with the default of 4, the streams RCCL creates push this
# library's weight-gradient side stream onto the main stream's queue and the two serialise (measured with a 1-rank RCCL group:
# 479 samples/s at 4 queues, 577-584 at 2 / 3 / 6 / 8 / 16).  Must be set before the HIP runtime initialises.
# (Rehearsals that put several ranks on ONE GPU -- BENCH_DEVICE -- keep the default of 4: two processes x 8 queues plus gloo's copy
# streams oversubscribe the device's hardware queues and the ranks stall in their first collectives; measured.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if os.environ.get("BENCH_DEVICE") is not None else "8")

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CFG = dict(input_fields=4, output_fields=4, patch_size=16, embed_dim=384, num_heads=6, processor_blocks=12,
           attn_scale=True, feat_scale=True, num_fluid_params=9)          # config/model_cfg/film_avit_small.yaml
DROP_PATH = 0.2                     # film_avit_small.yaml: stochastic depth, rates np.linspace(0, 0.2, 12) over the blocks
FIELD_STATS = ((-2.37, 1.98), (0.0145, 0.081), (-0.07, 0.49), (0.055, 0.77))
# algorithmic work per sample (fwd+bwd) -- 423.20 GFLOP and 1,638 U x 1.769 MB = 2.90 GB at 16x192x192;
# the 32x384x192 clip has 4x the tokens (1,709 GFLOP, 11.6 GB); the forward alone is 141.22 GFLOP / 546 U = 0.97 GB
WORKLOADS = {
    "configs1": dict(batch=8, T=16, H=192, W=192, flops=423.20e9, bytes=2.90e9, kind="train",
                     metric="train samples/sec, FiLMAViT 16x192x192x4ch, bs=8 per GPU",
                     what="4-field 16x192x192 clips, batch 8 per GPU (BASELINE.json configs[1])"),
    "configs3": dict(batch=4, T=32, H=384, W=192, flops=1709.0e9, bytes=11.6e9, kind="train",
                     metric="train samples/sec, FiLMAViT 32x384x192x4ch long-aspect clip, bs=4 per GPU",
                     what="4-field 32x384x192 long-aspect clips (24x12 tokens per frame, T = 32), batch 4 per GPU (BASELINE.json configs[3])"),
    "configs4": dict(batch=1, T=16, H=192, W=192, flops=141.22e9, bytes=0.97e9, kind="rollout",
                     metric="autoregressive rollout forward steps/sec, FiLMAViT 16x192x192x4ch, bs=1, HIP-graph replay",
                     what="200-step fed-back rollout of one 16x192x192 clip, eval forward captured in a HIP graph (BASELINE.json configs[4])"),
}
BYTES_PER_STEP_PARAMS = 0.69e9     # weights / grads / AdamW state, per training step
PEAK_HBM_GBS = 8000.0
ACHIEVABLE_HBM_GBS = 6290.0        # MI355X_MICROARCH.md: measured float4 copy
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def self_launch(args) -> int:
    """--gpus N > 1 with no launcher: start the N ranks as child processes (never re-exec a process that has touched the GPU; this
    parent never does) and pass rank 0's line through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    logdir = os.environ.get("BENCH_RANK_LOG_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out"))
    try:
        os.makedirs(logdir, exist_ok=True)
    except OSError:
        logdir = None
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0's stderr is the caller's; the other ranks' goes to a file each, so that a first multi-GPU run that fails can be read afterwards
        err = None
        if r > 0 and logdir is not None:
            err = open(os.path.join(logdir, "bench_rank%d.err" % r), "w")
            logs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err))
    # Poll every child: the first one that fails (any non-zero return code -- a rank killed by a signal has a NEGATIVE one) takes the
    # others down (they would otherwise block forever in their next collective, holding their GPUs); an overall limit does the same.
    import threading
    out_box = []
    reader = threading.Thread(target=lambda: out_box.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1500"))
    failed = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            failed = abs(bad[0]) or 1
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            log("self-launch: ranks still running after the limit; terminating them")
            failed = 124
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        log("self-launch: a rank failed (return codes %s); rank stderr files: %s" % ([p.returncode for p in procs], logdir))
    reader.join(timeout=5)
    for f in logs:
        f.close()
    if not failed and out_box:
        sys.stdout.write(out_box[0].decode())
        sys.stdout.flush()
    return failed


def synthetic_batch(w, seed, device):
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    def clip(s):
        gg = torch.Generator(device=device).manual_seed(s)
        t = torch.randn((w["batch"], w["T"], 4, w["H"], w["W"]), device=device, generator=gg)
        for c, (mu, sd) in enumerate(FIELD_STATS):
            t[:, :, c].mul_(sd).add_(mu)
        t[:, :, 1].clamp_(min=0.0)
        return t
    return clip(seed), torch.randn((w["batch"], 9), device=device, generator=g), clip(seed + 1)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, threads):
    """The oracle (CPU restatement of the reference path, pinned by tests/golden) as the reported CPU baseline: the workload's own
    batch (bounded: 1 warm-up + 2 timed steps) and batch 1, fp32, AdamW, on the host cores."""
    import torch
    from oracle import filmavit_ref as R, weights as Wt
    torch.set_num_threads(threads)
    shapes = Wt.param_shapes(**{k: v for k, v in CFG.items()})
    T, H, W = w["T"], w["H"], w["W"]

    def run(bs, nmin, nmax, budget):
        sd = {k: v.requires_grad_(True) for k, v in Wt.generate(shapes, seed=42).items()}
        opt = torch.optim.AdamW(list(sd.values()), lr=2.5e-4, weight_decay=1e-2)
        x = Wt.synthetic_clip(bs, T, 4, H, W, 42)
        y = Wt.synthetic_clip(bs, T, 4, H, W, 43)
        c = Wt.synthetic_fluid_params(bs, 9, 44)
        def step():
            opt.zero_grad(set_to_none=True)
            pred = R.filmavit_forward(sd, x, c, patch_size=16, num_heads=6)
            if w["kind"] == "train":
                R.lp_loss(pred, y).backward()
                opt.step()
        with torch.set_grad_enabled(w["kind"] == "train"):
            step()
            log(f"cpu baseline batch {bs}: warm-up step done")
            n, t0 = 0, time.perf_counter()
            while n < nmin or (time.perf_counter() - t0 < budget and n < nmax):
                step()
                n += 1
                log(f"cpu baseline batch {bs}: step {n} at {time.perf_counter() - t0:.1f} s")
            dt = time.perf_counter() - t0
        return bs * n / dt, n

    unit = "samples/s" if w["kind"] == "train" else "steps/s"
    what = "fwd+loss+bwd+AdamW steps" if w["kind"] == "train" else "eval forward steps"
    v1, n1 = run(1, 3, 12, 8.0)
    out = {"value": v1, "unit": unit, "cores": threads, "host_threads_visible": os.cpu_count(), "kind": "port", "cpu": cpu_model_name(),
           "sample": f"batch 1 clip {T}x{H}x{W}x4ch fp32, 1 warm-up + {n1} timed {what} of the oracle (torch CPU, {threads} threads)"}
    if w["batch"] > 1 and os.environ.get("BENCH_CPU_FULL_BATCH", "1") == "1":
        vb, nb = run(w["batch"], 2, 3, 20.0)
        out.update({"value": vb, "value_batch1": v1,
                    "sample": f"batch {w['batch']} clips {T}x{H}x{W}x4ch fp32 (the workload's own batch), 1 warm-up + {nb} timed {what} of the oracle "
                              f"(torch CPU, {threads} threads); value_batch1: the same at batch 1, {n1} timed steps"})
    return out


def eager_gpu_baseline(w, dev, autocast, nsteps=5, nwarm=2):
    """the same restatement run eagerly on the MI355X through stock PyTorch-ROCm kernels -- the
    un-accelerated GPU comparator (the bench shape, AdamW; fp32 or bf16 autocast).  Checker code, timed, never shipped."""
    import torch
    from oracle import filmavit_ref as R, weights as Wt
    shapes = Wt.param_shapes(**{k: v for k, v in CFG.items()})
    sd = {k: v.to(dev).requires_grad_(True) for k, v in Wt.generate(shapes, seed=42).items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=2.5e-4, weight_decay=1e-2, fused=True)
    x, c, y = synthetic_batch(w, 42, dev)
    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            pred = R.filmavit_forward(sd, x, c, patch_size=16, num_heads=6)
        loss = R.lp_loss(pred.float(), y)
        loss.backward()
        opt.step()
    for _ in range(nwarm):
        step()
    torch.cuda.synchronize()
    n, t0 = nsteps, time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": w["batch"] * n / dt, "unit": "samples/s", "ms_per_step": dt / n * 1e3, "dtype": "bf16 autocast" if autocast else "f32",
            "kind": "oracle restatement, eager PyTorch-ROCm on the same GPU", "sample": f"batch {w['batch']}, {nwarm} warm-up + {n} timed steps"}


def host_threads():
    """Cores this process may actually use: the affinity mask, cut to the cgroup CPU quota (a GPU box shows all 256 host threads in the
    mask but grants a 16-CPU share; 256 torch threads on 16 CPUs take minutes per step), or BENCH_CPU_THREADS."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    cap = os.environ.get("BENCH_CPU_THREADS")
    return max(1, min(n, int(cap)) if cap else n)


def roofline_from(prof, nprof, dtype, config):
    """The `roofline` object for the dominant kernel (largest share of GPU time) from the library's per-launch HIP-event report."""
    if not prof:
        return None
    tot_ms = sum(v["ms"] for v in prof.values())
    name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
    avg_ms = dom["ms"] / dom["calls"]
    flops, nbytes = dom["flops"] / dom["calls"], dom["bytes"] / dom["calls"]    # ALGORITHMIC work per launch
    peak_tf = PEAK_MFMA_TFLOPS[dtype]
    ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
    tflops = flops / (avg_ms * 1e-3) / 1e12
    gbs = nbytes / (avg_ms * 1e-3) / 1e9
    traffic, src = None, None
    tname = {"configs1": "r04_kernels.json", "configs4": "r04_rollout_kernels.json"}.get(config)     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command
    tpath = os.path.join(REPO, "profiles", tname) if tname else None
    if tpath and os.path.exists(tpath):
        for k in json.load(open(tpath))["kernels"]:
            if k["kernel"] == name and "hbm_bytes_per_launch" in k:
                traffic, src = k["hbm_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc passes of this command, committed; not re-measured in this run)" % tname
    if flops > 0 and flops / max(nbytes, 1.0) >= ridge:
        r = {"kernel": name, "bound": "mfma", "achieved": tflops, "peak": peak_tf, "unit": "TFLOP/s", "frac": tflops / peak_tf}
    else:
        r = {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}
    r.update({"traffic": traffic, "traffic_source": src, "avg_launch_ms": avg_ms, "launches_per_step": dom["calls"] / nprof,
              "algorithmic_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops, "tflops": tflops,
              "mfma_frac": tflops / peak_tf, "share_of_gpu_time": dom["ms"] / tot_ms,
              "kernel_time_share": {k: round(v["ms"] / tot_ms, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:12]},
              "kernel_avg_us": {k: [round(v["ms"] / v["calls"] * 1e3, 1), v["calls"] // nprof] for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:16]},
              "gpu_kernel_ms_per_step": tot_ms / nprof})
    return r


def prof_steps(fn, nprof=2):
    import torch
    from bubbleformer_amd import _lib
    h = _lib.lib()
    h.bf_prof_enable(1)
    for _ in range(nprof):
        fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    n = h.bf_prof_report(buf, len(buf))
    h.bf_prof_enable(0)
    return json.loads(buf.value.decode()) if n > 0 else {}


def clip_store_batches(w, dev, rank, world):
    """step index -> (input, fluid parameters, target) drawn from a DeviceClipStore over the reference's two sample trajectories tiled
    to the bench resolution (bubbleformer/data/dataset.py:120-184 semantics: sliding windows, std normalisation, 9 fluid parameters;
    the sample files ship without their .json sidecars, so the parameter records are synthetic).  The index order is a seeded
    shuffle, rank-strided like DistributedSampler's."""
    import numpy as np
    from bubbleformer_amd.data import hdf5_lite
    from bubbleformer_amd.data.dataset import BubbleForecast
    names = ("dfun", "temperature", "velx", "vely")
    trajs = []
    for k in (1, 2):
        f = hdf5_lite.File(os.path.join(REPO, "tests", "golden", "samples", "sample_%d.hdf5" % k))
        t = {}
        for n in names:
            a = np.asarray(f[n].array(), dtype=np.float32)
            ry, rx = -(-w["H"] // a.shape[-2]), -(-w["W"] // a.shape[-1])
            t[n] = np.tile(a, (1, ry, rx))[:, :w["H"], :w["W"]].copy()
        f.close()
        trajs.append(t)
    fluid = [{"inv_reynolds": 0.0042 * (1 + 0.1 * i), "cpgas": 0.83, "mugas": 0.023, "rhogas": 0.0083, "thcogas": 0.25, "stefan": 0.5298,
              "prandtl": 8.4, "heater": {"nucWaitTime": 0.4, "wallTemp": 1.0 + 0.05 * i}} for i in range(len(trajs))]
    ds = BubbleForecast.from_arrays(trajs, fluid, norm="std", time_window=w["T"], start_time=0)
    store = ds.device_store(dev)
    store.normalize()          # mean / std of every field by one reduction launch over the resident trajectories (bf_field_stats)
    import torch
    g = torch.Generator(device=dev).manual_seed(7)
    order = torch.cat([torch.randperm(len(ds), device=dev, generator=g) for _ in range(64)])     # device-resident shuffles: no host work per step
    B = w["batch"]

    def batch(s_):
        at = ((s_ * world + rank) * B) % (order.numel() - B)
        inp, tgt, fp = store.gather(order[at:at + B])
        return inp, fp, tgt
    return batch


def sample_trajectory(w, dev):
    """(frames, 4, H, W) trajectory derived from the reference's own sample file (tests/golden/samples/sample_1.hdf5: 50 frames of
    64 x 64, fields dfun / temperature / velx / vely): tiled to the clip size -- the rollout's target and first input."""
    import numpy as np
    import torch
    from bubbleformer_amd.data import hdf5_lite
    path = os.path.join(REPO, "tests", "golden", "samples", "sample_1.hdf5")
    f = hdf5_lite.File(path)
    fields = [np.asarray(f[k].array(), dtype=np.float32) for k in ("dfun", "temperature", "velx", "vely")]
    f.close()
    traj = torch.from_numpy(np.stack(fields, axis=1))                               # (50, 4, 64, 64)
    ry, rx = -(-w["H"] // traj.shape[-2]), -(-w["W"] // traj.shape[-1])
    return traj.repeat(1, 1, ry, rx)[..., :w["H"], :w["W"]].contiguous().to(dev)


def run_rollout(args, w, dev, dtype, profile=True):
    import torch
    from bubbleformer_amd import _lib
    from bubbleformer_amd.models import get_model
    from bubbleformer_amd.ops import _p, _stream
    from bubbleformer_amd.utils import physics
    from bubbleformer_amd.utils.rollout import GraphedForward, relative_l2_per_step
    torch.manual_seed(42)
    T = w["T"]
    model = get_model("filmavit", time_window=T, drop_path=0.0, compute_dtype=dtype, **CFG).to(dev).eval()
    traj = sample_trajectory(w, dev)
    nfr = traj.shape[0]
    clip_at = lambda s: traj[torch.arange(s * T, (s + 1) * T, device=dev) % nfr]     # the 50 sample frames, cycled
    cond = torch.randn((1, 9), device=dev, generator=torch.Generator(device=dev).manual_seed(44))
    x0 = clip_at(0).unsqueeze(0)
    steps = args.steps
    with torch.no_grad():
        fwd = GraphedForward(model, x0, cond)
        cur = x0
        for _ in range(args.warmup):
            cur = fwd(cur)
        torch.cuda.synchronize()
        # timed: K fed-back steps, the trajectory never leaves HBM (reference loop: scripts/inference.py:239-252)
        preds = torch.empty((steps, T, 4, w["H"], w["W"]), device=dev)
        cur = x0
        t0 = time.perf_counter()
        for s in range(steps):
            cur = fwd(cur)
            preds[s].copy_(cur[0])
            cur = preds[s].unsqueeze(0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = prof_steps(lambda: model(x0, cond)) if profile else {}
        # the record: per fed-back step, relative L2 against the sample-derived target (utils/losses.py:17-94 as inference.py:230 uses it),
        # Eikonal residual of the predicted signed-distance field (utils/losses.py:5-15), heater heat flux of the last frame of the step
        # (utils/heatflux.py:17-38 geometry scaled to this grid: 16-wide domain, dx = 16 / W, FC-72 constants)
        rec = []
        flux = torch.empty(1, device=dev)
        for s in range(steps):
            p = preds[s]
            finite = bool(torch.isfinite(p).all())
            l2 = float(relative_l2_per_step(p, clip_at(s + 1)).mean()) if finite else float("nan")
            eik = float(physics.eikonal_loss(p[:, 0])) if finite else float("nan")
            df, tp = p[-1, 0].contiguous(), p[-1, 1].contiguous()
            _lib.check(_lib.lib().bf_heatflux_rows(_p(df), _p(tp), 1, w["H"] * w["W"], w["W"], -8.0, 16.0 / w["W"], 1.0, 0.0007, _p(flux), _stream()), "bf_heatflux_rows")
            rec.append({"step": s, "rel_l2": l2, "eikonal": eik, "heatflux": float(flux)})
    return dt, prof, rec


def other_config_legs(dev, cdt, dtype_name):
    """Bounded legs of the default one-GPU run (a few seconds together), so that the driver's own record carries the other BASELINE
    configurations and the un-accelerated GPU comparator beside the headline: configs3 (5 timed training steps at 32x384x192, batch 4),
    configs4 (50 HIP-graph replays of the batch-1 rollout step) and the oracle restatement run eagerly on this GPU (bf16 autocast, 3 steps)."""
    import types
    import torch
    from bubbleformer_amd import ops
    from bubbleformer_amd.models import get_model
    from bubbleformer_amd.trainer import TrainStep
    legs = {}

    def roof(w, per_s, kind):
        pb = BYTES_PER_STEP_PARAMS / w["batch"] if kind == "train" else 0.058e9
        return {"hbm_frac": per_s * (w["bytes"] + pb) / (PEAK_HBM_GBS * 1e9), "mfma_frac": per_s * w["flops"] / (PEAK_MFMA_TFLOPS[dtype_name] * 1e12)}

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            legs[name] = fn()
        except Exception as e:      # a leg must never cost the headline line
            legs[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        legs[name]["leg_wall_s"] = round(time.perf_counter() - t0, 2)
        torch.cuda.synchronize()
        ops.clear_scratch()
        torch.cuda.empty_cache()

    def configs3():
        w = WORKLOADS["configs3"]
        torch.manual_seed(42)
        model = get_model("filmavit", time_window=w["T"], drop_path=DROP_PATH, compute_dtype=cdt, **CFG).to(dev).train()
        step = TrainStep(model, lr=2.5e-4, weight_decay=1e-2)
        x, cond, y = synthetic_batch(w, 42, dev)
        for _ in range(2):
            step(x, cond, y)
        torch.cuda.synchronize()
        n, t0 = 5, time.perf_counter()
        for _ in range(n):
            loss = step(x, cond, y)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        v = w["batch"] * n / dt
        return {"metric": w["metric"], "value": v, "unit": "samples/s", "ms_per_step": dt / n * 1e3, "steps": n, "warmup": 2, "loss": float(loss),
                "step_roofline": roof(w, v, "train"), "what": w["what"]}

    def configs4():
        w = WORKLOADS["configs4"]
        a = types.SimpleNamespace(steps=50, warmup=3)
        dt, _prof, rec = run_rollout(a, w, dev, cdt, profile=False)
        v = a.steps / dt
        return {"metric": w["metric"], "value": v, "unit": "steps/s", "ms_per_step": dt / a.steps * 1e3, "steps": a.steps, "warmup": a.warmup,
                "rel_l2_last": rec[-1]["rel_l2"], "step_roofline": roof(w, v, "rollout"), "what": w["what"] + " (random-init weights: plumbing, not physics)"}

    guarded("configs3", configs3)
    guarded("configs4", configs4)
    guarded("eager_gpu_baseline_bf16", lambda: eager_gpu_baseline(WORKLOADS["configs1"], dev, True, nsteps=3, nwarm=1))
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--config", default="configs1", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "device-store"],
                    help="device-store: every timed step draws its batch from a DeviceClipStore (HBM-resident trajectories, bf_clip_gather)")
    ap.add_argument("--record", default=None, help="configs4: write the per-step rollout record (relative L2, Eikonal, heat flux) to this file")
    ap.add_argument("--eager-gpu-baseline", action="store_true", help="also time the oracle eagerly on the GPU (fp32 and bf16 autocast)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the bounded configs3 / configs4 / eager-GPU legs the default one-GPU configs1 run adds as `other_configs`")
    args = ap.parse_args()
    w = WORKLOADS[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    if os.environ.get("BENCH_STACKS_AFTER"):                    # debugging aid: dump every thread's Python stack after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_STACKS_AFTER"]), repeat=False, exit=False)

    # stdout carries exactly one JSON line: whatever libraries print there (RCCL's version banner under NCCL_DEBUG=VERSION) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    # BENCH_DEVICE / BENCH_BACKEND exist only to rehearse the multi-rank control flow on a one-GPU box (gloo, every rank on
    # one device); the driver's runs use one GPU per rank and RCCL ("nccl").
    if os.environ.get("BENCH_DEVICE") is not None:
        local = int(os.environ["BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"      # rehearsal: a 1-rank RCCL group with the gradient exchange switched on
    if world > 1 or force_dist:
        if force_dist and world == 1:
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531")):
                os.environ.setdefault(k, v)
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from bubbleformer_amd.models import get_model
    from bubbleformer_amd.trainer import TrainStep
    torch.manual_seed(42)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rollout_record = None
    clip_supply = None
    if w["kind"] == "rollout":                                   # replicas only: every rank runs its own rollout
        sync()
        dt, prof, rollout_record = run_rollout(args, w, dev, cdt)
        sync()
        final_loss = rollout_record[-1]["rel_l2"]
    else:
        model = get_model("filmavit", time_window=w["T"], drop_path=DROP_PATH, compute_dtype=cdt, **CFG).to(dev).train()
        step = TrainStep(model, lr=2.5e-4, weight_decay=1e-2)            # config/optim_cfg/adamw.yaml
        if force_dist and os.environ.get("BENCH_FORCE_REDUCE", "1") == "1":
            step.reducer.enabled = True
        x, cond, y = synthetic_batch(w, 42 + 1000 * rank, dev)
        for i in range(args.warmup):
            loss = step(x, cond, y)
            if i == 0:
                torch.cuda.synchronize()
                log("first step done, loss", float(loss))
        sync()
        log("warm-up done")

        def timed(batch_fn, nsteps):
            sync()
            t0 = time.perf_counter()
            for s_ in range(nsteps):
                xb, cb, yb = batch_fn(s_)
                ls = step(xb, cb, yb)
            sync()
            return time.perf_counter() - t0, ls

        store_fn = None
        if args.config == "configs1" and (args.data == "device-store" or world == 1):
            store_fn = clip_store_batches(w, dev, rank, world)
        if args.data == "device-store":
            if store_fn is None:
                raise SystemExit("--data device-store is built for configs1 (the 16x192x192 training step)")
            timed(store_fn, min(3, args.steps))                 # first gathers (index tables, allocator)
            dt, loss = timed(store_fn, args.steps)
            dt_syn, _ = timed(lambda s_: (x, cond, y), min(args.steps, 50))
            clip_supply = {"samples_per_s_device_store": w["batch"] * world * args.steps / dt,
                           "samples_per_s_synthetic_same_run": w["batch"] * world * min(args.steps, 50) / dt_syn}
        else:
            dt, loss = timed(lambda s_: (x, cond, y), args.steps)
            if store_fn is not None:                             # the clip supply in a timed loop beside the resident-batch figure (bounded: 50 steps)
                n2 = min(args.steps, 50)
                timed(store_fn, 3)
                dt2, _ = timed(store_fn, n2)
                clip_supply = {"samples_per_s_device_store": w["batch"] * world * n2 / dt2, "steps": n2,
                               "samples_per_s_synthetic_same_run": w["batch"] * world * args.steps / dt}
        if clip_supply is not None:
            clip_supply["ratio"] = clip_supply["samples_per_s_device_store"] / clip_supply["samples_per_s_synthetic_same_run"]
            clip_supply["what"] = ("each step: DeviceClipStore.gather(8 shuffled sample indices) -> (input, target, fluid parameters) -> TrainStep; "
                                   "trajectories: tests/golden/samples/sample_{1,2}.hdf5 tiled 3 x 3 to 192 x 192, 38 sliding windows")
        final_loss = float(loss)
        # ---- roofline leg: per-launch HIP-event timing (on the launch stream) over two more steps
        prof = prof_steps(lambda: step(x, cond, y))
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    log("timed region: %.3f s for %d steps" % (dt, args.steps))
    roofline = roofline_from(prof, 2, args.dtype, args.config)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    value = w["batch"] * world * args.steps / dt
    per_gpu = value / world
    param_bytes = BYTES_PER_STEP_PARAMS / w["batch"] if w["kind"] == "train" else 0.058e9      # rollout: the bf16 weights are read once per step
    bytes_per_sample = w["bytes"] + param_bytes
    out = {
        "metric": w["metric"], "value": value, "unit": "samples/s" if w["kind"] == "train" else "steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "FiLMAViT-small (E=384, 6 heads, 12 blocks, P=16, 9 fluid params) "
                               + ("fwd + relative-L2 loss + bwd + AdamW, " if w["kind"] == "train" else "eval forward fed back on itself, ") + w["what"],
                   "name": args.config, "global_batch": w["batch"] * world, "parallelism": f"dp{world}" if w["kind"] == "train" else f"replicas x{world}",
                   "drop_path": DROP_PATH if w["kind"] == "train" else 0.0},
        "loss": final_loss,
        "roofline": roofline,
        "step_roofline": {"hbm_frac": per_gpu * bytes_per_sample / (PEAK_HBM_GBS * 1e9),
                          "hbm_frac_of_achievable_6.29TBs": per_gpu * bytes_per_sample / (ACHIEVABLE_HBM_GBS * 1e9),
                          "mfma_frac": per_gpu * w["flops"] / (PEAK_MFMA_TFLOPS[args.dtype] * 1e12),
                          "algorithmic_GB_per_sample": w["bytes"] / 1e9, "algorithmic_GFLOP_per_sample": w["flops"] / 1e9},
    }
    if rollout_record is not None:
        out["rollout_record"] = {"steps": len(rollout_record), "first": rollout_record[0], "last": rollout_record[-1]}
        if args.record:           # opt-in: profiling passes of this command must not rewrite a committed record
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.record)), exist_ok=True)
                json.dump({"config": out["config"], "dtype": args.dtype, "ms_per_step": ms, "record": rollout_record}, open(args.record, "w"), indent=0)
                out["rollout_record"]["file"] = args.record
            except OSError as e:
                out["rollout_record"]["error"] = str(e)
    if clip_supply is not None:
        out["clip_supply"] = clip_supply
        if args.data == "device-store":
            out["data"] = "device-store: batches gathered per step from HBM-resident trajectories (the reference's sample files tiled to 192 x 192, std-normalised; DeviceClipStore.gather = bf_clip_gather)"
    if world == 1 and args.config == "configs1" and args.dtype == "bf16" and not args.no_other_configs and not force_dist:
        log("bounded legs: configs3, configs4, eager GPU comparator")
        del step, model, x, cond, y
        from bubbleformer_amd import ops as _ops
        torch.cuda.synchronize()
        _ops.clear_scratch()
        torch.cuda.empty_cache()
        out["other_configs"] = other_config_legs(dev, cdt, args.dtype)
        step = model = None
    if world == 1 and not args.no_cpu_baseline:
        log("timing the CPU baseline (oracle) on", host_threads(), "threads")
        out["cpu_baseline"] = cpu_baseline(w, host_threads())
    if world == 1 and args.eager_gpu_baseline and w["kind"] == "train":
        step = model = None
        torch.cuda.empty_cache()
        out["eager_gpu_baseline"] = [eager_gpu_baseline(w, dev, True), eager_gpu_baseline(w, dev, False)]
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
