#!/usr/bin/env python3
"""Build-time register audit (run by __graft_entry__.build() on the CPU box).

The ring / pair / stream / token-reduction GEMM families order their LDS-DMA traffic with hand-counted `s_waitcnt vmcnt(N)`:
a register spill inside such a kernel adds scratch loads / stores to the same counter and silently changes what a counted wait
covers (EXPERIMENTS.md records a fault from exactly that).  This script reads the per-kernel resource remarks the Makefile leaves
in csrc/build/<file>.remarks (hipcc -Rpass-analysis=kernel-resource-usage) and FAILS when a kernel compiled from a source file
that uses counted waits (`wait_vm<`, or `s_waitcnt vmcnt` in inline asm) reports spilled VGPRs or scratch.  Spills in other
kernels (compiler-ordered waits only) are reported as warnings.  --table prints every kernel (VGPRs, spills, scratch, LDS,
occupancy).

usage: python tools/register_audit.py [--table] [--json out.json]"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bubbleformer_amd", "csrc")
BUILD = os.path.join(CSRC, "build")

FIELDS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
          "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names) + "\n", capture_output=True, text=True).stdout.split("\n")
        return [o if o else n for o, n in zip(out, names)]
    except OSError:
        return names


def short(name):
    if name.startswith("_Z"):      # c++filt does not know the bf16 mangling (DF16b): kernel name + integer / bool template arguments by hand
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name) or re.match(r"_Z(\d+)", name)
        if m:
            n = int(m.group(1))
            base, rest = name[m.end():m.end() + n], name[m.end() + n:]
            args = []
            if rest.startswith("I"):
                for t, v in re.findall(r"L([ib])(\d+)E", rest.split("EEv")[0] + "E"):
                    args.append(("true" if v == "1" else "false") if t == "b" else v)
            return base + ("<" + ", ".join(args) + ">" if args else "")
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<.*?>)?)\(", name)
    return m.group(1) if m else name.split("(")[0]


def parse(path):
    kernels, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"mangled": t.split(":", 1)[1].strip()}
            kernels.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.rsplit(":", 1)
            if k.strip() in FIELDS:
                cur[FIELDS[k.strip()]] = int(v)
    return kernels


def counted_wait_files():
    out = set()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip"):
            s = open(os.path.join(CSRC, f)).read()
            if "wait_vm<" in s or re.search(r'asm[^;]*s_waitcnt vmcnt', s):
                out.add(f[:-4])
    return out


def audit():
    if not os.path.isdir(BUILD):
        raise SystemExit("register_audit: %s is missing -- run make first" % BUILD)
    counted = counted_wait_files()
    rows, bad, warn = [], [], []
    for f in sorted(os.listdir(BUILD)):
        if not f.endswith(".remarks"):
            continue
        stem = f[:-8]
        ks = parse(os.path.join(BUILD, f))
        names = demangle([k["mangled"] for k in ks])
        for k, n in zip(ks, names):
            k["file"], k["kernel"], k["counted_waits"] = stem, short(n), stem in counted
            rows.append(k)
            spilled = k.get("vgpr_spill", 0) > 0 or k.get("scratch", 0) > 0
            if spilled:
                (bad if k["counted_waits"] else warn).append(k)
    missing = [s for s in counted if not os.path.exists(os.path.join(BUILD, s + ".remarks"))]
    return rows, bad, warn, missing


def table(rows):
    out = ["| file | kernel | VGPR | AGPR | spilled VGPR | scratch B/lane | LDS B | waves/SIMD | counted vmcnt |", "|---|---|---|---|---|---|---|---|---|"]
    for k in sorted(rows, key=lambda r: (r["file"], r["kernel"])):
        out.append("| %s | `%s` | %d | %d | %d | %d | %d | %d | %s |" % (k["file"], k["kernel"], k.get("vgpr", 0), k.get("agpr", 0), k.get("vgpr_spill", 0),
                                                                        k.get("scratch", 0), k.get("lds", 0), k.get("occ", 0), "yes" if k["counted_waits"] else ""))
    return "\n".join(out)


def main(argv):
    rows, bad, warn, missing = audit()
    if "--table" in argv:
        print(table(rows))
    if "--json" in argv:
        json.dump(rows, open(argv[argv.index("--json") + 1], "w"), indent=1)
    if missing:
        print("register_audit: no remarks for %s (stale build directory? run `make clean all`)" % ", ".join(sorted(missing)))
        return 2
    for k in warn:
        print("register_audit: warning %s.hip `%s`: %d spilled VGPRs, %d B/lane scratch (no counted waits in that file)" % (
            k["file"], k["kernel"], k.get("vgpr_spill", 0), k.get("scratch", 0)))
    for k in bad:
        print("register_audit: FAIL %s.hip `%s`: %d spilled VGPRs, %d B/lane scratch beside hand-counted vmcnt waits" % (
            k["file"], k["kernel"], k.get("vgpr_spill", 0), k.get("scratch", 0)))
    n_counted = sum(1 for k in rows if k["counted_waits"])
    print("register_audit: %d kernels, %d in files with counted vmcnt waits, %d failures" % (len(rows), n_counted, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
