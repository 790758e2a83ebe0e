"""ctypes binding of libbubbleformer_hip.so (the C ABI declared in include/bubbleformer_hip.h).

There is NO fallback: if the library is missing, or a call fails, this raises.  PyTorch is used only for
device memory (caching allocator), streams and autograd plumbing; every pointer handed to the library is a
raw device address.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbubbleformer_hip.so")
CSRC = os.path.join(_HERE, "csrc")

BF_DTYPE_F32, BF_DTYPE_BF16 = 0, 1
BF_LAY_KC, BF_LAY_XC = 0, 1
BF_PRO_NONE, BF_PRO_AFFINE, BF_PRO_AFFINE_GELU, BF_PRO_GELU = 0, 1, 2, 3
BF_AUX_NONE, BF_AUX_ADD, BF_AUX_DGELU = 0, 1, 2
BF_OUT_STORE, BF_OUT_STORE_F32, BF_OUT_ATOMIC_F32 = 0, 1, 2
BF_MAX_STAGES = 5
BF_LOSS_LIMBS = 5          # int64 limbs per relative-L2 partial sum (include/bubbleformer_hip.h)

vp, fp, i32, i64, f32 = C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float


class Operand(C.Structure):
    _fields_ = [("p", vp), ("ld", i64), ("layout", i32), ("seglen", i32), ("segstride", i64), ("gw", i32), ("gh", i32),
                ("gc", i32), ("pro", i32), ("sc", fp), ("sh", fp), ("rows_per_frame", i32), ("nch", i32)]


class Epilogue(C.Structure):
    _fields_ = [("bias", fp), ("colscale", fp), ("colshift", fp), ("aux_mode", i32), ("aux", vp), ("ld_aux", i64),
                ("out_mode", i32), ("c", vp), ("ldc", i64), ("seglen", i32), ("segstride", i64), ("gw", i32), ("gh", i32),
                ("gc", i32), ("gelu_out", vp), ("colsum", fp), ("rowscale", fp), ("rows_per_group", i32)]


class Dims(C.Structure):
    _fields_ = [(n, i32) for n in ("dtype", "B", "T", "h", "w", "E", "heads", "attn_scale", "feat_scale", "patch", "cin",
                                   "cout", "nfluid")]


TEMPORAL_FIELDS = ("gamma", "attn_scale_factor", "norm1_w", "norm1_b", "norm2_w", "norm2_b", "input_head_w", "input_head_b",
                   "output_head_w", "output_head_b", "qnorm_w", "qnorm_b", "knorm_w", "knorm_b", "rel_pos_emb")
SPATIAL_FIELDS = ("gamma_att", "gamma_mlp", "attn_scale_factor_x", "attn_scale_factor_y", "low_freq_scalar",
                  "high_freq_scalar", "norm1_w", "norm1_b", "norm2_w", "norm2_b", "input_head_w", "input_head_b",
                  "output_head_w", "output_head_b", "qnorm_w", "qnorm_b", "knorm_w", "knorm_b", "rel_pos_emb", "fc1_w", "fc1_b",
                  "fc2_w", "fc2_b", "mlp_norm_w", "mlp_norm_b")


class TemporalParams(C.Structure):
    _fields_ = [(n, fp) for n in TEMPORAL_FIELDS]


class SpatialParams(C.Structure):
    _fields_ = [(n, fp) for n in SPATIAL_FIELDS]


class FrameNorm(C.Structure):
    """bf_frame_norm: an InstanceNorm folded into a frame-pair GEMM launch (bf_gemm_fwd_frames)."""
    _fields_ = [("w", fp), ("b", fp), ("g", fp), ("gdiv", i32), ("mean", fp), ("rstd", fp), ("sc", fp), ("sh", fp), ("resid", vp), ("out", vp)]


class EmbedParams(C.Structure):
    _fields_ = [("conv_w", fp * BF_MAX_STAGES), ("in_w", fp * BF_MAX_STAGES), ("in_b", fp * BF_MAX_STAGES), ("film_ln_w", fp),
                ("film_ln_b", fp), ("film_w", fp), ("film_b", fp)]


class DebedParams(C.Structure):
    _fields_ = [("conv_w", fp * BF_MAX_STAGES), ("in_w", fp * BF_MAX_STAGES), ("in_b", fp * BF_MAX_STAGES)]


STAGE_DONE_FN = C.CFUNCTYPE(None, C.c_int, C.c_void_p)      # bf_stage_done_fn
P = C.POINTER
# name -> (restype, argtypes); mirrors include/bubbleformer_hip.h one to one
SIGNATURES = {
    "bf_last_error": (C.c_char_p, []),
    "bf_abi_version": (C.c_int, []),
    "bf_prof_enable": (None, [C.c_int]),
    "bf_debug_force_generic_attn": (None, [C.c_int]),
    "bf_debug_tokred_fold": (None, [C.c_int]),
    "bf_side_defer": (None, [C.c_int]),
    "bf_prep_stages": (C.c_int, [C.POINTER(Dims), C.c_int, C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp]),
    "bf_stage_prepared": (None, [C.c_int]),
    "bf_trunk_train_fwd": (C.c_int, [C.POINTER(Dims), C.c_int, C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp,
                                     C.POINTER(vp), vp, vp]),
    "bf_trunk_train_bwd": (C.c_int, [C.POINTER(Dims), C.c_int, C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                     C.POINTER(vp), vp, C.POINTER(vp), vp, C.POINTER(vp), vp, vp, vp, vp, vp]),
    "bf_field_stats_ws_doubles": (i64, [C.c_int]),
    "bf_field_stats": (C.c_int, [fp, vp, vp, C.c_int, vp, vp, vp]),
    "bf_stage_chain_next": (C.c_int, [C.POINTER(Dims), C.c_int, vp, vp]),
    "bf_stage_next_scale": (C.c_int, [fp, C.c_int]),
    "bf_gemm_fwd_frames": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, fp, fp, fp, fp, C.c_int, vp, vp, C.c_int,
                                     C.POINTER(FrameNorm), C.POINTER(FrameNorm), vp]),
    "bf_trunk_eval_weights_bytes": (i64, [C.POINTER(Dims), C.c_int, C.POINTER(C.c_int32)]),
    "bf_trunk_eval_prepare": (C.c_int, [C.POINTER(Dims), C.c_int, C.POINTER(C.c_int32), C.POINTER(vp), vp, vp]),
    "bf_trunk_eval_fwd": (C.c_int, [C.POINTER(Dims), C.c_int, C.POINTER(C.c_int32), C.POINTER(vp), vp, vp, vp, vp, vp]),
    "bf_frame_linear": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, fp, fp, fp, fp, fp, vp, i64, C.c_int, fp, fp, fp,
                                  vp, i64, fp, fp, vp, i64, vp]),
    "bf_side_join": (C.c_int, [vp]),
    "bf_prof_report": (C.c_int, [C.c_char_p, C.c_int]),
    "bf_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, P(Operand), P(Operand), P(Epilogue), C.c_int, vp]),
    "bf_gemm_tokred": (C.c_int, [C.c_int, C.c_int, C.c_int, i64, vp, i64, vp, i64, fp, C.c_int, fp, fp, i64, vp]),
    "bf_gemm_tokred_ws_floats": (i64, [C.c_int, C.c_int, i64]),
    "bf_gemm_inbwd_frames": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, vp, vp, vp, C.c_int, fp, fp, fp, fp, fp, C.c_int, vp]),
    "bf_gemm_inbwd_frames_chain": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, vp, vp, vp, C.c_int, fp, fp, fp, fp, fp, C.c_int,
                                             vp, vp, fp, fp, fp, fp, C.c_int, fp, vp]),
    "bf_in_stats": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, fp, fp, fp, C.c_int, fp, fp, fp, fp, fp, fp, vp]),
    "bf_in_ws_floats": (i64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "bf_affine_apply": (C.c_int, [C.c_int, vp, vp, fp, fp, vp, i64, C.c_int, C.c_int, vp]),
    "bf_in_bwd": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp, C.c_int, C.c_int, fp, fp,
                            fp, fp, fp, vp]),
    "bf_colsum": (C.c_int, [C.c_int, vp, i64, C.c_int, fp, fp, vp]),
    "bf_attn_axial_fwd": (C.c_int, [C.c_int, vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, fp, vp]),
    "bf_attn_axial_norm_fwd": (C.c_int, [C.c_int, vp, vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp,
                                         fp, vp]),
    "bf_attn_fwd": (C.c_int, [C.c_int, vp, vp, i64, C.c_int, i64, i64, i64, i64, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, f32,
                              C.c_int, vp]),
    "bf_attn_bwd": (C.c_int, [C.c_int, vp, vp, vp, i64, C.c_int, i64, i64, i64, i64, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp,
                              fp, fp, fp, fp, fp, fp, f32, C.c_int, fp, i64, vp]),
    "bf_im2col_nchw": (C.c_int, [C.c_int, fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_col2im_nchw": (C.c_int, [C.c_int, vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_pm2nchw": (C.c_int, [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_debed_last_bwd": (C.c_int, [C.c_int, fp, fp, fp, fp, fp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_debed_last_bwd_norm": (C.c_int, [C.c_int, fp, fp, fp, fp, fp, vp, vp, vp, fp, fp, fp, fp, vp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, fp, C.c_int64, vp]),
    "bf_gather_gemm": (C.c_int, [C.c_int, vp, vp, C.c_int, fp, fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_scatter_gemm": (C.c_int, [C.c_int, vp, vp, C.c_int, fp, fp, vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_gather_wgrad_ws_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "bf_gather_wgrad": (C.c_int, [C.c_int, vp, vp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int64, vp]),
    "bf_gather_gemm_rebuilt": (C.c_int, [C.c_int, vp, vp, vp, C.c_int, fp, fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_gather_wgrad_rebuilt": (C.c_int, [C.c_int, vp, vp, vp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int64, vp]),
    "bf_stage_chain_head": (C.c_int, [vp, vp, vp]),
    "bf_stage_chain_tail": (C.c_int, [vp, vp, C.c_int]),
    "bf_embed_tail_ws_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bf_embed_tail_bwd": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    fp, C.c_int64, vp]),
    "bf_embed_first": (C.c_int, [C.c_int, fp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, vp]),
    "bf_in_stats_merge_slices": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, C.c_int, fp, fp, fp, fp, fp, fp, vp]),
    "bf_debed_last": (C.c_int, [C.c_int, vp, fp, fp, vp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_lploss_finalize": (C.c_int, [fp, C.c_int, C.c_int, fp, fp, vp]),
    "bf_nchw2pm": (C.c_int, [C.c_int, fp, fp, fp, fp, fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_wprep": (C.c_int, [C.c_int, C.c_int, fp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "bf_wgrad_unprep": (C.c_int, [C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_film_net_fwd": (C.c_int, [fp, fp, fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]),
    "bf_film_net_bwd": (C.c_int, [fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]),
    "bf_adamw": (C.c_int, [fp, fp, fp, fp, i64, C.c_int, f32, f32, f32, f32, f32, f32, vp]),
    "bf_lion": (C.c_int, [fp, fp, fp, i64, f32, f32, f32, f32, f32, vp]),
    "bf_eikonal_sum": (C.c_int, [fp, i64, C.c_int, C.c_int, f32, vp, vp]),
    "bf_eikonal_l1_frames": (C.c_int, [fp, i64, C.c_int, C.c_int, f32, fp, vp]),
    "bf_heatflux_rows": (C.c_int, [fp, fp, i64, i64, C.c_int, f32, f32, f32, f32, fp, vp]),
    "bf_clip_gather": (C.c_int, [fp, i64, vp, vp, C.c_int, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_clip_gather_batch": (C.c_int, [fp, i64, vp, i64, vp, vp, fp, fp, C.c_int, C.c_int, fp, vp, fp, fp, C.c_int, C.c_int, fp, fp, vp, C.c_int, fp,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "bf_temporal_saved_bytes": (i64, [P(Dims)]),
    "bf_spatial_saved_bytes": (i64, [P(Dims)]),
    "bf_embed_saved_bytes": (i64, [P(Dims)]),
    "bf_debed_saved_bytes": (i64, [P(Dims)]),
    "bf_scratch_bytes": (i64, [P(Dims)]),
    "bf_temporal_fwd": (C.c_int, [P(Dims), P(TemporalParams), vp, vp, vp, vp, fp, vp]),
    "bf_temporal_bwd": (C.c_int, [P(Dims), P(TemporalParams), P(TemporalParams), vp, vp, vp, vp, vp, fp, vp]),
    "bf_spatial_fwd": (C.c_int, [P(Dims), P(SpatialParams), vp, vp, vp, vp, fp, fp, vp]),
    "bf_spatial_bwd": (C.c_int, [P(Dims), P(SpatialParams), P(SpatialParams), vp, vp, vp, vp, vp, fp, fp, vp]),
    "bf_embed_fwd": (C.c_int, [P(Dims), P(EmbedParams), fp, fp, vp, vp, vp, vp]),
    "bf_embed_bwd": (C.c_int, [P(Dims), P(EmbedParams), P(EmbedParams), vp, fp, vp, vp, vp]),
    "bf_debed_fwd": (C.c_int, [P(Dims), P(DebedParams), vp, fp, fp, fp, vp, vp, vp]),
    "bf_debed_bwd": (C.c_int, [P(Dims), P(DebedParams), P(DebedParams), vp, fp, fp, fp, fp, vp, vp, vp, vp]),
}

_lib = None


class BubbleformerHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-j8", "-C", CSRC], capture_output=True, text=True)
    if verbose or res.returncode:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode:
        raise BubbleformerHipError("building libbubbleformer_hip.so failed")
    return LIB_PATH


def lib():
    """Load the native library (once).  Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BubbleformerHipError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
                "bubbleformer_amd has no CPU / eager fallback.")
        # PyTorch-ROCm ships its own libamdhip64; make sure THAT runtime instance is the one this library binds to
        # (two HIP runtimes in one process do not share devices, streams or allocations).
        import torch  # noqa: F401
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(tl):
            C.CDLL(tl, mode=C.RTLD_GLOBAL)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().bf_last_error().decode(errors="replace")
        if rc == 1 and not msg.startswith("declined"):      # a refusal that left no text of its own (the last-error text may be stale)
            msg = "declined: the entry point does not cover this shape / dtype / workspace (nothing was launched)"
        raise BubbleformerHipError(f"{what} failed (code {rc}): {msg}")
