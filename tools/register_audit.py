#!/usr/bin/env python3
"""Build-time register audit (run by __graft_entry__.build() on the CPU box).

The ring / pair / stream / token-reduction GEMM families order their LDS-DMA traffic with hand-counted `s_waitcnt vmcnt(N)`:
a register spill inside such a kernel adds scratch loads / stores to the same counter and silently changes what a counted wait
covers (EXPERIMENTS.md records a fault from exactly that).  This script reads the per-kernel resource remarks the Makefile leaves
in csrc/build/<file>.remarks (hipcc -Rpass-analysis=kernel-resource-usage) and FAILS when a kernel compiled from a source file
that uses counted waits (`wait_vm<`, or `s_waitcnt vmcnt` in inline asm) reports spilled VGPRs or scratch.  Spills in other
kernels (compiler-ordered waits only) are reported as warnings.  It also reads the device assembly the Makefile leaves in
csrc/build/<file>.s and FAILS when hipcc-issued loads sit inside a loop that waits for LDS-DMA by count (`foreign_loads`), or when a
packed-fp32 instruction takes its LOW lane's src1 / src2 from the HIGH half of a register pair (`packed_high_select`).  --table
prints every kernel (VGPRs, spills, scratch, LDS, occupancy).

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
        t = re.sub(r"^\S+:\d+:\d+:\s+", "", m.group(1))      # (with -save-temps the location follows the word "remark:")
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


def foreign_loads(path):
    """Kernels of one device assembly file in which hipcc-issued loads (outside ;;#ASMSTART .. ;;#ASMEND) sit inside a LOOP that also
    holds an LDS-DMA load and a hand-counted `s_waitcnt vmcnt(N)`, N > 0, from inline asm.  Such a loop orders its DMA by counting
    vector-memory operations; a load the compiler put there is an operation the count does not know -- hipcc waits for it by ITS
    count, which does not know the DMAs: a load placed BEHIND a DMA loosens the hand count by one.  (The round-4 kernel that prompted the
    rule, tokred_narrow_kernel<.., true>, turned out to be wrong for another reason -- packed_high_select below; its loads sat in front of
    the DMAs and only made the counts stricter.)  Returns [(kernel, first offending line)]."""
    out = []
    text = open(path, errors="replace").read()
    for m in re.finditer(r"^(_Z\S+):[ \t]*(?:;.*)?$", text, re.M):
        end = text.find(".Lfunc_end", m.end())
        if end < 0:
            continue
        lines = text[m.end():end].split("\n")
        in_asm, flags = False, []          # per line: (is_asm, text)
        for l in lines:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            flags.append((in_asm, t))
            if t.startswith(";;#ASMEND"):
                in_asm = False
        # hipcc annotates every basic block with the innermost natural loop it belongs to ("=>This Inner Loop Header", "in Loop: Header=BBx_y"):
        # group the lines by that loop, so that an outer loop's own blocks (per-segment operand fetches on a queue it drains itself --
        # tokred_narrow_kernel<.., true> after the fix) are not mistaken for the counted loop inside it
        by_loop, cur = {}, None
        for i, (asm, t) in enumerate(flags):
            bm = re.match(r"^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)(.*)$", t)
            if bm and not asm:
                rest = bm.group(2)
                for _, t2 in flags[i + 1:i + 4]:          # a nested header's annotation continues on comment lines ("Parent Loop ..." / "=> This Inner Loop Header")
                    if not t2.startswith(";") or t2.startswith(";;#") or t2.startswith("; %bb."):
                        break
                    rest += " " + t2
                hm = re.search(r"in Loop: Header=(BB\d+_\d+)", rest)
                cur = bm.group(1) if (bm.group(1) and "Loop Header" in rest) else (hm.group(1) if hm else None)
                continue
            if cur is not None:
                by_loop.setdefault(cur, []).append((asm, t))
        hit = None
        for body in by_loop.values():
            dma = any(asm and re.match(r"^(global_load_lds|buffer_load\S* .*\blds\b)", t) for asm, t in body)
            counted = any(asm and re.match(r"^s_waitcnt vmcnt\(([1-9]\d*)\)", t) for asm, t in body)
            if not (dma and counted):
                continue
            for asm, t in body:
                if not asm and re.match(r"^(global_load_|buffer_load_|flat_load_|scratch_load_)", t) and "lds" not in t:
                    hit = t
                    break
            if hit:
                break
        if hit:
            out.append((m.group(1), hit))
    return out


def packed_high_select(path):
    """Kernels of one device assembly file that hold a packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose
    LOW lane takes src1 or src2 from the HIGH register of the pair (`op_sel:[_,1,_]` / `op_sel:[_,_,1]`).  Round 4, at the ISA level
    (EXPERIMENTS.md): tokred_narrow_kernel<6, true> held four `v_pk_fma_f32 .. v[2:3], v[8:9] op_sel:[0,1,1]` -- the one channel
    block computed with them came out 1-2 % off, differently every run, whenever two workgroups shared a CU; the same binary with only
    those four instructions replaced (two v_fma_f32, or the high halves copied to a free pair and the default selectors) is exact and
    reproducible.  hipcc picks the form when two neighbouring scalars of a register array are broadcast; no kernel of the build needs it.
    Returns [(kernel, instruction)]."""
    out = []
    text = open(path, errors="replace").read()
    for m in re.finditer(r"^(_Z\S+):[ \t]*(?:;.*)?$", text, re.M):
        end = text.find(".Lfunc_end", m.end())
        if end < 0:
            continue
        for l in text[m.end():end].split("\n"):
            t = l.strip()
            sm = re.match(r"^v_pk_(?:fma|mul|add)_f32 .*\bop_sel:\[([01]),([01])(?:,([01]))?\]", t)
            if sm and (sm.group(2) == "1" or sm.group(3) == "1"):
                out.append((m.group(1), t.split("//")[0].strip()))
                break
    return out


def audit_packed_high_select():
    bad = []
    for f in sorted(os.listdir(BUILD)):
        if f.endswith(".s"):
            for k, line in packed_high_select(os.path.join(BUILD, f)):
                bad.append({"file": f[:-2], "kernel": short(demangle([k])[0]), "line": line})
    return bad


def audit_foreign_loads():
    bad = []
    for f in sorted(os.listdir(BUILD)):
        if f.endswith(".s"):
            for k, line in foreign_loads(os.path.join(BUILD, f)):
                bad.append({"file": f[:-2], "kernel": short(demangle([k])[0]), "line": line})
    return bad


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
    foreign = audit_foreign_loads()
    for k in foreign:
        print("register_audit: FAIL %s.hip `%s`: a compiler-issued load (`%s`) inside a loop that waits for LDS-DMA by count" % (k["file"], k["kernel"], k["line"]))
    packed = audit_packed_high_select()
    for k in packed:
        print("register_audit: FAIL %s.hip `%s`: packed fp32 op whose low lane reads the high half of src1 / src2 (`%s`)" % (k["file"], k["kernel"], k["line"]))
    bad = bad + foreign + packed
    n_counted = sum(1 for k in rows if k["counted_waits"])
    print("register_audit: %d kernels, %d in files with counted vmcnt waits, %d failures" % (len(rows), n_counted, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
