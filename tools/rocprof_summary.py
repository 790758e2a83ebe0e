#!/usr/bin/env python3
"""Turns rocprofv3 CSV output into the small summaries committed under profiles/.

  python tools/rocprof_summary.py <trace_dir> [<pmc_fetch_dir> <pmc_write_dir>] --steps N --out profiles/rNN

* <trace_dir>:  rocprofv3 --kernel-trace --stats --output-format csv -d <trace_dir> -- python bench.py ...
* <pmc_*_dir>:  rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md section HBM prescribes).
  HBM bytes per launch = 2 * FETCH_SIZE (gfx950 counts a wide coalesced read at half its bytes) + WRITE_SIZE, in KiB units.
Kernel names are normalised to the names the library's own launch profiler uses (bench.py `roofline.kernel`).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def normalise(name: str) -> str:
    m = re.search(r"gemm_kernelI(DF16b|f)Lb([01])ELb([01])ELb([01])ELb([01])ELi(\d)ELi(\d)ELi(\d+)E", name)
    if m:
        t, ax, bx, ap, bp, _, tm, nt = m.groups()
        return "gemm_kernel<%s,%s,%s,pro%s,tm%s,w%d>" % ("bf16" if t == "DF16b" else "f32", "xc" if ax == "1" else "kc", "xc" if bx == "1" else "kc",
                                                        "A" if ap == "1" else "B" if bp == "1" else "0", tm, int(nt) // 64)
    m = re.search(r"gemm_kernel<(__hip_bfloat16|float), (true|false), (true|false), (true|false), (true|false), \d, (\d), (\d+)>", name)
    if m:
        t, ax, bx, ap, bp, tm, nt = m.groups()
        return "gemm_kernel<%s,%s,%s,pro%s,tm%s,w%d>" % ("bf16" if t != "float" else "f32", "xc" if ax == "true" else "kc", "xc" if bx == "true" else "kc",
                                                        "A" if ap == "true" else "B" if bp == "true" else "0", tm, int(nt) // 64)
    m = re.search(r"gemm_kernel<bool _Accum, bool, E, (true|false), (true|false), (true|false), \d, (\d), (\d+)>", name)
    if m:   # rocprofv3's demangler garbles the A-outer-contiguous (dW) instantiations; AXC = true there
        bx, ap, bp, tm, nt = m.groups()
        return "gemm_kernel<bf16,xc,%s,pro%s,tm%s,w%d>" % ("xc" if bx == "true" else "kc", "A" if ap == "true" else "B" if bp == "true" else "0", tm,
                                                          int(nt) // 64)
    m = re.search(r"gemm_wide_kernel<(true|false), (true|false), (true|false), (true|false), (\d)>", name)
    if m:
        ax, bx, ap, bp, tm = m.groups()
        return "gemm_wide_kernel<bf16,%s,%s,pro%s,tm%s>" % ("xc" if ax == "true" else "kc", "xc" if bx == "true" else "kc",
                                                           "A" if ap == "true" else "B" if bp == "true" else "0", tm)
    m = re.search(r"stream_gemm_kernel<(\d), (true|false), (true|false)>", name) or re.search(r"stream_gemm_kernelILi(\d)ELb([01])ELb([01])E", name)
    if m:
        aux, g2, _ = m.groups()
        return "stream_gemm<%s>" % ("gelu2" if g2 in ("true", "1") else {"0": "plain", "1": "add", "2": "dgelu"}[aux])
    m = re.search(r"stream_pp_kernel<(\d), (true|false), (true|false)>", name) or re.search(r"stream_pp_kernelILi(\d)ELb([01])ELb([01])E", name)
    if m:
        aux, g2, _ = m.groups()
        return "stream_pp<%s>" % ("gelu2" if g2 in ("true", "1") else {"0": "plain", "1": "add", "2": "dgelu"}[aux])
    m = re.search(r"gemm_pair_kernel<(\d)>", name) or re.search(r"gemm_pair_kernelILi(\d)E", name)
    if m:      # MODE 0: data gradient + InstanceNorm backward; 1: plain data gradient (+ residual) -- the library's profiler names it <add>; 2: MODE 0 + the chained second norm
        # 3: the forward twin (out-projection + the next stage's norm1); 4: fc2 + MLP-branch norm (+ the next stage's norm1)
        return {"0": "gemm_pair<inbwd>", "2": "gemm_pair<inbwd,chain>", "3": "gemm_pair<fwd,chain>", "4": "gemm_pair<fwd,norm>", "5": "gemm_pair<inbwd,scaled>"}.get(m.group(1), "gemm_pair<add>")
    m = re.search(r"tokred_pp_reduce_kernel", name)
    if m:
        return "tokred_reduce_kernel"
    m = re.search(r"tokred_pp_kernel<(\d)>", name) or re.search(r"tokred_pp_kernelILi(\d)E", name)
    if m:
        return "tokred_pp_kernel<%dx192,h32,ring4>" % (64 * int(m.group(1)))
    if "tokred_kernel" in name:
        return "tokred_kernel<128x128,bk64,slots3>"
    m = re.search(r"frame_res_kernel<(\d), (\d), (true|false)>", name) or re.search(r"frame_res_kernelILi(\d)ELi(\d)ELb([01])E", name)
    if m:
        ntc, _, norm = m.groups()
        return "frame_linear<res,bn%d%s>" % (32 * int(ntc), ",norm" if norm in ("true", "1") else "")
    m = re.search(r"frame_ring_kernel<(\d), (\d)>", name) or re.search(r"frame_ring_kernelILi(\d)ELi(\d)E", name)
    if m:
        return "frame_linear<ring,bn%d,in>" % (32 * int(m.group(1)))      # the model uses the streamed form for fc2 + InstanceNorm only
    if "tokred_reduce_kernel" in name:
        return "tokred_reduce_kernel"
    # round-3 embed / debed kernels (gather_gemm.hip, embed_tail.hip, patch.hip)
    m = re.search(r"gather_gemm_kernel<\d, \d, (true|false), (true|false)>", name)
    if m:
        return "gather_gemm<%s>" % ("gelu,rebuilt" if m.group(2) == "true" else "gelu" if m.group(1) == "true" else "plain")
    m = re.search(r"scatter_gemm_kernel<\d, \d, (true|false)>", name)
    if m:
        return "scatter_gemm<%s>" % ("gelu" if m.group(1) == "true" else "plain")
    m = re.search(r"gather_wgrad_kernel<(true|false), (true|false), (true|false)>", name)
    if m:
        return "gather_wgrad<%s>" % ("fine gelu,rebuilt" if m.group(3) == "true" else "fine gelu" if m.group(1) == "true" else "coarse gelu" if m.group(2) == "true" else "plain")
    m = re.search(r"debed_last_inbwd_kernel<\d, (\d)>", name)
    if m:
        return "debed_last_bwd<%s>" % ("stats" if m.group(1) == "1" else "apply")
    m = re.search(r"tokred_narrow_kernel<\d, (true|false)>", name)
    if m:
        return "tokred_narrow<%s>" % ("gelu" if m.group(1) == "true" else "plain")
    if "debed_last_bwd_kernel" in name:
        return "patch16_kernel (embed_first / debed_last_bwd)"      # one kernel: the K = 16 contraction at either end of the model
    for key in ("gather_wgrad_reduce_kernel", "embed_tail_bwd_kernel", "embed_tail_frame_kernel", "embed_tail_sum_kernel", "tokred_narrow_reduce_kernel", "dl_slice_sum_kernel",
                "dl_param_reduce_kernel"):
        if key in name:
            return key
    if "gemm_inbwd_frames_kernel" in name:
        return "gemm_inbwd_frames<bf16>"
    for key in ("frame_scale_kernel", "stage_param_reduce_kernel", "stage_prep_kernel", "clip_gather_kernel", "eikonal_kernel", "heatflux_kernel",
                "lion_kernel", "in_bwd_slice_kernel", "in_stats_slice_kernel", "in_stats_merge_kernel", "in_slice_sum_kernel", "frame_table_kernel", "frame_wcolsum_kernel",
                "cast4_kernel", "lploss_finalize_kernel", "fill_kernel"):
        if key in name:
            return key
    for key in ("attn_fwd_axial_mfma", "attn_bwd_mfma", "attn_fwd_mfma", "attn_ws_reduce", "in_bwd_kernel", "in_stats_kernel", "in_param_reduce_kernel", "affine_apply_kernel",
                "colsum_kernel", "adamw_kernel", "wprep_multi_kernel", "wprep_kernel", "outproj_finalize_kernel", "outproj_prep_kernel", "stage_prep_multi_kernel", "debed_last_bwd_kernel", "debed_last_kernel", "pm2nchw_kernel", "nchw2pm_kernel",
                "im2col_kernel", "film_net_bwd_kernel", "film_net_fwd_kernel", "wgrad_unprep_kernel", "attn_fwd_kernel", "attn_bwd_kernel"):
        if key in name:
            return key
    return name[:80]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("pmc_fetch", nargs="?")
    ap.add_argument("pmc_write", nargs="?")
    ap.add_argument("--steps", type=int, default=0, help="total steps the profiled command ran (warm-up + timed + profiler leg); 0 = the number of optimizer launches in the trace")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    stats = glob.glob(os.path.join(a.trace, "**", "*kernel_stats.csv"), recursive=True)[0]
    agg = collections.OrderedDict()
    total = 0.0
    for r in csv.DictReader(open(stats)):
        k = normalise(r["Name"])
        e = agg.setdefault(k, {"calls": 0, "total_ns": 0.0, "rocprof_names": []})
        e["calls"] += int(r["Calls"])
        e["total_ns"] += float(r["TotalDurationNs"])
        e["rocprof_names"].append(r["Name"][:120])
        total += float(r["TotalDurationNs"])
    if a.steps <= 0:
        a.steps = max(1, sum(e["calls"] for k, e in agg.items() if k in ("adamw_kernel", "lion_kernel")))
    traffic = {}
    if a.pmc_fetch and a.pmc_write:
        for d, ctr in ((a.pmc_fetch, "FETCH_SIZE"), (a.pmc_write, "WRITE_SIZE")):
            f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != ctr:
                    continue
                t = traffic.setdefault(normalise(r["Kernel_Name"]), {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
                t[ctr][0] += 1
                t[ctr][1] += float(r["Counter_Value"])
    rows = []
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1]["total_ns"]):
        row = {"kernel": k, "calls_per_step": e["calls"] / a.steps, "avg_us": e["total_ns"] / e["calls"] / 1e3,
               "ms_per_step": e["total_ns"] / a.steps / 1e6, "share": e["total_ns"] / total}
        if k in traffic:
            fe, wr = traffic[k]["FETCH_SIZE"], traffic[k]["WRITE_SIZE"]
            fkb = fe[1] / max(1, fe[0])
            wkb = wr[1] / max(1, wr[0])
            row.update(fetch_size_kib_per_launch=fkb, write_size_kib_per_launch=wkb, hbm_bytes_per_launch=(2.0 * fkb + wkb) * 1024.0)
        rows.append(row)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"gpu_ms_per_step": total / a.steps / 1e6, "kernels": rows}, open(a.out + "_kernels.json", "w"), indent=1)
    with open(a.out + "_kernel_stats.md", "w") as f:
        f.write("| kernel | calls/step | avg us | ms/step | share | HBM MB/launch (2*FETCH+WRITE) |\n|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            f.write("| `%s` | %.1f | %.1f | %.3f | %.1f%% | %s |\n" % (r["kernel"], r["calls_per_step"], r["avg_us"], r["ms_per_step"], 100 * r["share"],
                                                                      ("%.1f" % (r["hbm_bytes_per_launch"] / 1e6)) if "hbm_bytes_per_launch" in r else ""))
        f.write("\nGPU kernel time per step: %.3f ms\n" % (total / a.steps / 1e6))
    print(open(a.out + "_kernel_stats.md").read())


if __name__ == "__main__":
    main()
