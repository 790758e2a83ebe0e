#!/usr/bin/env python3
"""Two-queue timeline of a training step from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`):
per hardware queue the busy time, the idle gaps (and which kernel pair they sit between), per-kernel time, and how long the
main-queue kernels take when they run alone vs under a side-queue kernel.  Steps are delimited by the optimizer kernel.

  python tools/timeline.py <kernel_trace.csv> [--out profiles/rNN_timeline.md]
"""
import argparse
import bisect
import collections
import csv
import re


def short(n):
    m = re.search(r"gemm_kernelIDF16bLb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d)ELi(\d)ELi(\d+)", n)
    if m:
        return "gemm<bf16,%s,%s,pro%s,tm%s>" % ("xc" if m.group(1) == "1" else "kc", "xc" if m.group(2) == "1" else "kc",
                                                "A" if m.group(3) == "1" else "B" if m.group(4) == "1" else "0", m.group(6))
    m = re.search(r"gemm_kernel<bool _Accum, bool, E, (true|false), (true|false), (true|false)", n)
    if m:
        return "gemm<bf16,xc,%s,pro%s>" % ("xc" if m.group(1) == "true" else "kc", "A" if m.group(2) == "true" else "B" if m.group(3) == "true" else "0")
    m = re.search(r"stream_gemm_kernel<(\d), (true|false)", n) or re.search(r"stream_gemm_kernelILi(\d)ELb([01])", n)
    if m:
        return "stream_gemm<%s>" % ("gelu2" if m.group(2) in ("true", "1") else {"0": "plain", "1": "add", "2": "dgelu"}[m.group(1)])
    m = re.search(r"stream_pp_kernel<(\d), (true|false)", n) or re.search(r"stream_pp_kernelILi(\d)ELb([01])", n)
    if m:
        return "stream_pp<%s>" % ("gelu2" if m.group(2) in ("true", "1") else {"0": "plain", "1": "add", "2": "dgelu"}[m.group(1)])
    m = re.search(r"gemm_pair_kernel<(\d)>", n)
    if m:
        return {"0": "gemm_pair<inbwd>", "2": "gemm_pair<inbwd,chain>", "3": "gemm_pair<fwd,chain>", "4": "gemm_pair<fwd,norm>", "5": "gemm_pair<inbwd,scaled>"}.get(m.group(1), "gemm_pair<add>")
    if "tokred_pp_reduce" in n:
        return "tokred_reduce"
    for key in ("gather_wgrad_reduce", "gather_wgrad", "gather_gemm", "scatter_gemm", "embed_tail_bwd", "embed_tail_frame", "embed_tail_sum", "debed_last_inbwd", "tokred_narrow_reduce",
                "tokred_narrow", "dl_slice_sum", "dl_param_reduce"):
        if key in n:
            return key
    for key in ("tokred_pp_kernel", "tokred_reduce", "tokred_kernel", "gemm_inbwd_frames", "attn_fwd_axial_mfma", "attn_bwd_mfma", "attn_fwd_mfma", "in_bwd_slice", "in_stats_slice", "in_stats_merge", "in_slice_sum", "in_bwd_kernel",
                "in_stats_kernel", "in_param_reduce", "stage_param_reduce", "stage_prep_multi", "stage_prep", "frame_scale", "frame_table", "adamw", "outproj_finalize",
                "wgrad_unprep", "wprep", "debed_last_bwd", "debed_last", "pm2nchw", "nchw2pm", "im2col", "film_net_bwd", "film_net_fwd", "fillBufferAligned", "copyBuffer", "lploss"):
        if key in n:
            return key
    return re.sub(r"^void ", "", n)[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--out")
    ap.add_argument("--anatomy", help="also write ONE steady-state step launch by launch (queue, start, duration, workgroups) to this file")
    a = ap.parse_args()
    raw = list(csv.DictReader(open(a.trace)))
    rows = [(int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
             int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
             // max(1, int(r.get("Workgroup_Size_X", 1) or 1) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))) for r in raw]
    rows.sort(key=lambda r: r[1])
    opt = [i for i, r in enumerate(rows) if r[3] == "adamw"]
    first, last = opt[len(opt) // 3], opt[-3]                     # steady-state steps
    if a.anatomy:
        mid = opt[len(opt) // 2]
        one = rows[mid + 1:opt[len(opt) // 2 + 1] + 1]
        t0 = one[0][1]
        qs = sorted({r[0] for r in one}, key=lambda q: -sum(1 for r in one if r[0] == q))
        with open(a.anatomy, "w") as f:
            f.write("# one steady-state step, launch by launch: queue (0 = caller's stream), start us, duration us, workgroups, kernel\n")
            for r in one:
                f.write("%d %9.1f %8.1f %7d  %s\n" % (qs.index(r[0]), (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[4], r[3]))
    seg = rows[first + 1:last + 1]
    nsteps = opt.index(last) - opt.index(first)
    lines = []
    P = lines.append
    P("# Timeline of the training step (%d steady-state steps of `bench.py`, rocprofv3 kernel trace)\n" % nsteps)
    P("wall time per step: %.0f us\n" % ((seg[-1][2] - seg[0][1]) / nsteps / 1e3))
    queues = sorted({r[0] for r in seg})
    main_q = max(queues, key=lambda q: sum(1 for r in seg if r[0] == q))
    for q in queues:
        s = [r for r in seg if r[0] == q]
        busy = sum(r[2] - r[1] for r in s) / nsteps / 1e3
        gaps = collections.defaultdict(lambda: [0, 0.0])
        prev = None
        for r in s:
            if prev is not None and r[1] > prev[2]:
                g = gaps[prev[3] + " -> " + r[3]]
                g[0] += 1
                g[1] += (r[1] - prev[2]) / 1e3
            prev = r if prev is None or r[2] > prev[2] else prev
        tot = sum(v[1] for v in gaps.values()) / nsteps
        P("## queue %d (%s): %.0f launches/step, busy %.0f us/step, idle between kernels %.0f us/step\n" %
          (q, "caller's stream" if q == main_q else "library side stream", len(s) / nsteps, busy, tot))
        P("| idle us/step | gaps/step | avg us | between |\n|---|---|---|---|")
        for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:10]:
            P("| %.1f | %.1f | %.1f | %s |" % (v[1] / nsteps, v[0] / nsteps, v[1] / v[0], k))
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in s:
            agg[r[3]][0] += 1
            agg[r[3]][1] += (r[2] - r[1]) / 1e3
        P("\n| kernel | launches/step | avg us | us/step |\n|---|---|---|---|")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            P("| %s | %.1f | %.1f | %.0f |" % (k, v[0] / nsteps, v[1] / v[0], v[1] / nsteps))
        P("")
    side = sorted([r for r in seg if r[0] != main_q], key=lambda r: r[1])
    st = [r[1] for r in side]
    agg = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for r in seg:
        if r[0] != main_q:
            continue
        i = bisect.bisect_left(st, r[1])
        ov = 0
        for j in range(max(0, i - 3), min(len(side), i + 6)):
            lo, hi = max(r[1], side[j][1]), min(r[2], side[j][2])
            ov += max(0, hi - lo)
        frac = ov / max(1, r[2] - r[1])
        d = (r[2] - r[1]) / 1e3
        if frac < 0.1:
            agg[r[3]][0] += 1; agg[r[3]][1] += d
        elif frac > 0.7:
            agg[r[3]][2] += 1; agg[r[3]][3] += d
    P("## main-queue kernels alone vs under a side-queue kernel\n")
    P("| kernel | alone: launches/step | avg us | overlapped: launches/step | avg us |\n|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:10]:
        P("| %s | %.1f | %.1f | %.1f | %.1f |" % (k, v[0] / nsteps, v[1] / max(1, v[0]), v[2] / nsteps, v[3] / max(1, v[2])))
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
