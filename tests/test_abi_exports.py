"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/bubbleformer_hip.h declares;
the product path refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(REPO, "include", "bubbleformer_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from bubbleformer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    h = _lib.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(h, n), n
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert h.bf_abi_version() == 1


def test_register_audit_no_spills_beside_counted_waits():
    """tools/register_audit.py over the compiler's per-kernel resource remarks of the in-tree build: no kernel compiled from a source file
    with hand-counted `s_waitcnt vmcnt` waits (ring / pair / stream / token-reduction GEMM families) may spill or use scratch."""
    from bubbleformer_amd import _lib
    from tools import register_audit
    if not os.path.isdir(register_audit.BUILD) or not any(f.endswith(".remarks") for f in os.listdir(register_audit.BUILD)):
        _lib.build()
    rows, bad, warn, missing = register_audit.audit()
    assert not missing, missing
    assert len(rows) > 100 and sum(1 for k in rows if k["counted_waits"]) >= 30
    assert {"gemm_pair_kernel<0>", "gemm_pair_kernel<1>", "gemm_pair_kernel<2>", "tokred_pp_kernel<3>"} <= {k["kernel"] for k in rows}
    assert not bad, [(k["kernel"], k.get("vgpr_spill"), k.get("scratch")) for k in bad]


def test_no_cpu_fallback():
    from bubbleformer_amd import _lib
    from bubbleformer_amd.models import get_model
    m = get_model("filmavit", input_fields=4, output_fields=4, time_window=2, patch_size=4, embed_dim=64, num_heads=1,
                  processor_blocks=1, drop_path=0.0, num_fluid_params=9)
    with pytest.raises(_lib.BubbleformerHipError):
        m(torch.randn(1, 2, 4, 8, 8), torch.randn(1, 9))


def test_registry_contract():
    from bubbleformer_amd.models import get_model, list_models, register_model
    assert list_models() == ["avit", "filmavit"]
    with pytest.raises(KeyError):
        get_model("nope")
    with pytest.raises(ValueError):
        register_model("avit")(object)


def test_state_dict_matches_reference_inventory():
    from bubbleformer_amd.models import get_model
    from oracle import weights as W
    cfg = dict(input_fields=4, output_fields=4, patch_size=16, embed_dim=384, num_heads=6, processor_blocks=12, num_fluid_params=9)
    m = get_model("filmavit", time_window=16, drop_path=0.0, **cfg)
    shapes = W.param_shapes(**cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys()) and len(sd) == 506
    assert sum(v.numel() for v in sd.values()) == 28906602
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)


def test_register_audit_flags_compiler_loads_inside_a_counted_dma_loop(tmp_path):
    """The second build-time rule (tools/register_audit.py: foreign_loads): a load hipcc issued itself inside the innermost loop that
    orders LDS-DMA by a hand-counted `s_waitcnt vmcnt(N)` fails the build; the same load in an enclosing loop's own blocks does not
    (the shape of tokred_narrow_kernel<.., true> before and after the round-4 fix)."""
    from tools import register_audit
    dma = ";;#ASMSTART\n\tglobal_load_lds_dwordx4 v[18:19], off\n;;#ASMEND\n"
    wait = ";;#ASMSTART\n\ts_waitcnt vmcnt(7)\n;;#ASMEND\n"
    bad = ("_Z3badv:                                ; @_Z3badv\n.LBB0_1:                                ; =>This Inner Loop Header: Depth=1\n"
           "\tglobal_load_dword v2, v[22:23], off\n" + dma + wait + "\ts_cbranch_vccnz .LBB0_1\n.Lfunc_end0:\n")
    good = ("_Z4goodv:                               ; @_Z4goodv\n.LBB1_1:                                ; =>This Loop Header: Depth=1\n"
            "\tglobal_load_dword v2, v[22:23], off\n;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n;;#ASMEND\n"
            ".LBB1_2:                                ;   Parent Loop BB1_1 Depth=1\n                                        ; =>  This Inner Loop Header: Depth=2\n"
            + dma + wait + "\ts_cbranch_vccnz .LBB1_2\n; %bb.3:                                ;   in Loop: Header=BB1_1 Depth=1\n\ts_cbranch_vccnz .LBB1_1\n.Lfunc_end1:\n")
    f = tmp_path / "k.s"
    f.write_text(bad + good)
    nested_bad = ("_Z6nestedv:                             ; @_Z6nestedv\n.LBB2_1:                                ; =>This Loop Header: Depth=1\n"
                  ".LBB2_2:                                ;   Parent Loop BB2_1 Depth=1\n                                        ; =>  This Inner Loop Header: Depth=2\n"
                  "\tglobal_load_dword v2, v[22:23], off\n" + dma + wait + "\ts_cbranch_vccnz .LBB2_2\n\ts_cbranch_vccnz .LBB2_1\n.Lfunc_end2:\n")
    f.write_text(bad + good + nested_bad)
    hits = register_audit.foreign_loads(str(f))
    assert [k for k, _ in hits] == ["_Z3badv", "_Z6nestedv"], hits


def test_register_audit_flags_packed_fp32_ops_with_the_low_lane_on_a_high_half(tmp_path):
    """The third build-time rule (tools/register_audit.py: packed_high_select): `v_pk_fma_f32 .. op_sel:[0,1,1]` (low lane from the high
    registers of src1 / src2) fails the build -- the form measured wrong in tokred_narrow_kernel<6, true> (EXPERIMENTS.md, round 4);
    the default selectors, the high-lane selectors and a src0 swap do not."""
    from tools import register_audit
    f = tmp_path / "k.s"
    f.write_text("_Z3badv:                                ; @_Z3badv\n\tv_pk_fma_f32 v[26:27], v[26:27], v[2:3], v[8:9] op_sel:[0,1,1]\n.Lfunc_end0:\n"
                 "_Z4bad2v:                               ; @_Z4bad2v\n\tv_pk_mul_f32 v[26:27], v[26:27], v[2:3] op_sel:[0,1]\n.Lfunc_end1:\n"
                 "_Z4goodv:                               ; @_Z4goodv\n\tv_pk_fma_f32 v[34:35], v[22:23], v[2:3], v[8:9] op_sel_hi:[1,0,0]\n"
                 "\tv_pk_fma_f32 v[136:137], v[150:151], v[220:221], v[136:137] op_sel:[1,0,0] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9]\n.Lfunc_end2:\n")
    assert [k for k, _ in register_audit.packed_high_select(str(f))] == ["_Z3badv", "_Z4bad2v"]
