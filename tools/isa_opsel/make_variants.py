#!/usr/bin/env python3
"""ISA-level experiment behind EXPERIMENTS.md (round 4, "A wrong kernel found by asking for bit-reproducibility").

Takes the device assembly of gemm_tokred.hip as it was BEFORE commit 3eb5101 (check that file out of history, compile it with
`hipcc --offload-arch=gfx950 -O3 -save-temps=obj -c`, pass the `*-hip-amdgcn-amd-amdhsa-gfx950.s` here), writes variants that differ only
in the `v_pk_fma_f32 .. op_sel:[0,1,1]` instructions, assembles each to a code object, and builds harness.hip, which loads them with
hipModuleLoad and runs tokred_narrow_kernel<6, true> on seeded data against an fp64 host reference:

    python tools/isa_opsel/make_variants.py old_gemm_tokred.s out_dir
    gpurun -- 'cd out_dir && ./harness base.hsaco scalar.hsaco lowcopy.hsaco nop.hsaco drained.hsaco unfused.hsaco'
    GRID=256 ./harness base.hsaco        # one workgroup per CU: exact
    LDSB=102400 ./harness base.hsaco     # 512 workgroups, but one fits a CU: exact
(GRID must stay <= 512: the slab the harness allocates has 512 slices.)"""
import os
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"
SITE = re.compile(r"v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,1\]")


def scalar(m):
    d0, d1, a0, a1, b0, b1, c0, c1 = [int(x) for x in m.groups()]
    return "v_fma_f32 v%d, v%d, v%d, v%d\n\tv_fma_f32 v%d, v%d, v%d, v%d" % (d0, a0, b1, c1, d1, a1, b1, c1)


def lowcopy(m):       # v[54:57] are dead at the four sites of the <6, true> instantiation (checked by hand in that listing)
    d0, d1, a0, a1, b0, b1, c0, c1 = [int(x) for x in m.groups()]
    if (b0, c0) != (2, 8):
        return m.group(0)
    return "v_mov_b32_e32 v54, v%d\n\tv_mov_b32_e32 v56, v%d\n\ts_nop 1\n\tv_pk_fma_f32 v[%d:%d], v[%d:%d], v[54:55], v[56:57] op_sel_hi:[1,0,0]" % (b1, c1, d0, d1, a0, a1)


def unfused(m):
    d0, d1, a0, a1, b0, b1, c0, c1 = [int(x) for x in m.groups()]
    return ("v_pk_mul_f32 v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[0,1]\n\ts_nop 1\n\tv_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[0,1]"
            % (d0, d1, a0, a1, b0, b1, d0, d1, d0, d1, c0, c1))


VARIANTS = {"base": lambda m: m.group(0), "scalar": scalar, "lowcopy": lowcopy, "unfused": unfused,
            "nop": lambda m: "s_nop 4\n\t" + m.group(0) + "\n\ts_nop 4",
            "drained": lambda m: "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t" + m.group(0)}


def main(src, out):
    os.makedirs(out, exist_ok=True)
    text = open(src).read()
    print("%d op_sel:[0,1,1] sites" % len(SITE.findall(text)))
    for name, fn in VARIANTS.items():
        s = os.path.join(out, name + ".s")
        open(s, "w").write(SITE.sub(fn, text))
        subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", s[:-2] + ".o"])
        subprocess.check_call([LLVM + "/ld.lld", "-shared", s[:-2] + ".o", "-o", s[:-2] + ".hsaco"])
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-o", os.path.join(out, "harness"), os.path.join(here, "harness.hip")])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
