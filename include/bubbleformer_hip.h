/*
 * bubbleformer_hip.h -- C ABI of libbubbleformer_hip.so (gfx950 / MI355X).
 *
 * The reference (HPCForge/Bubbleformer) has no FFI: its hot path is a Python
 * nn.Module tree that dispatches stock ATen ops.  This header is the boundary a
 * native replacement binds instead: plain device pointers, sizes and a HIP
 * stream -- no torch types.  Every entry point is stream-ordered, performs no
 * allocation and no device-wide synchronisation (hipGraph-capturable), and
 * returns 0 on success or a negative code (bf_last_error() gives the text).
 *
 * Tensors.  Activations are token-major / channels-last: [N, C] with
 * N = ((b*T + t)*h + y)*w + x.  `dtype` selects the activation storage type
 * (BF_DTYPE_F32 = exact-fp32 parity mode on f32 MFMA; BF_DTYPE_BF16 = bf16
 * storage + bf16 MFMA, fp32 accumulate and fp32 statistics).  Parameters and
 * parameter gradients are always fp32 in the reference's state_dict layout.
 *
 * Reference interface each group of entry points replaces (file:line relative to
 * the reference repository root):
 *   bf_gemm .................. nn.Conv2d 1x1 / nn.Linear / k2s2 (transposed) conv GEMMs:
 *                              layers/attention.py:78,121,210,299; linear_layers.py:25;
 *                              layers/patching.py:36-44,92-100 (stages with C >= 8)
 *   bf_in_stats / bf_in_bwd .. nn.InstanceNorm2d(affine): layers/attention.py:77,120,208,298,316
 *   bf_frame_linear / bf_trunk_eval_*  the same layers' forward in eval mode, whole-frame tiles (scripts/inference.py:239-252)
 *   bf_gemm_tokred ........... weight gradients of the same layers (autograd); bf_gemm_inbwd_frames: their data gradient + InstanceNorm backward
 *   bf_attn_fwd / bf_attn_bwd  AttentionBlock.forward attention core: layers/attention.py:80-119 (and each axial pass)
 *   bf_attn_axial_fwd / bf_attn_axial_norm_fwd  AxialAttentionBlock.forward attention core: layers/attention.py:212-297
 *   bf_embed_first_* ......... HMLPEmbed stage 0 (Conv2d k2s2 on NCHW input): layers/patching.py:36-44
 *   bf_debed_last_* .......... HMLPDebed last stage + LpLoss: layers/patching.py:92-100, utils/losses.py:67-94
 *   bf_film_* ................ FiLMMLP.forward: layers/linear_layers.py:63-77
 *   bf_adamw ................. torch.optim.AdamW as configured at bubbleformer/modules.py:135-136
 *   bf_clip_gather ........... BubbleForecast.__getitem__ + DataLoader collate for a batch of clips: bubbleformer/data/dataset.py:120-182
 *   bf_eikonal_sum / bf_heatflux_rows .. eikonal_loss utils/losses.py:5-15, heatflux utils/heatflux.py:3-38
 *   bf_lion .................. lion_pytorch.Lion (the reference's default optimizer) at bubbleformer/modules.py:139-140
 */
#ifndef BUBBLEFORMER_HIP_H
#define BUBBLEFORMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bf_stream_t; /* hipStream_t */

enum { BF_DTYPE_F32 = 0, BF_DTYPE_BF16 = 1 };

/* operand memory layouts for bf_gemm */
enum { BF_LAY_KC = 0, /* [outer][k]   : reduction index contiguous      */
       BF_LAY_XC = 1  /* [k][outer]   : outer index contiguous          */ };
/* prologues applied to an operand element while it is staged (fp32 math) */
enum { BF_PRO_NONE = 0, BF_PRO_AFFINE = 1, BF_PRO_AFFINE_GELU = 2, BF_PRO_GELU = 3 };
/* what the epilogue combines with the accumulator */
enum { BF_AUX_NONE = 0, BF_AUX_ADD = 1 /* += aux[m][n] */, BF_AUX_DGELU = 2 /* *= gelu'(aux[m][n]) */ };
enum { BF_OUT_STORE = 0 /* store activation dtype */, BF_OUT_ATOMIC_F32 = 2 /* atomicAdd into fp32 */,
       BF_OUT_STORE_F32 = 1 /* store fp32 */ };

/* One GEMM operand.  Memory "rows" are tokens (or weight rows); "columns" are
 * channels.  address(row, col) = rowbase(row) + (col / seglen) * segstride + col % seglen,
 * rowbase(row) = row * ld, or, when gw > 0, the top-left pixel of a 2x2 patch:
 * row = (f*gh + y)*gw + x  ->  ((f*2*gh + 2*y) * 2*gw + 2*x) * gc   (k2s2 patch gather / scatter). */
typedef struct bf_operand {
    const void* p;          /* activation dtype (weights: dtype of the GEMM) */
    int64_t ld;
    int32_t layout;         /* BF_LAY_* */
    int32_t seglen;         /* 0 = no segmentation */
    int64_t segstride;
    int32_t gw, gh, gc;     /* 0 = plain rows */
    int32_t pro;            /* BF_PRO_* */
    const float* sc;        /* [frames][nch] scale  (AFFINE*) */
    const float* sh;        /* [frames][nch] shift (NULL = 0) */
    int32_t rows_per_frame; /* frame = row / rows_per_frame   */
    int32_t nch;            /* channel = col % nch            */
} bf_operand;

typedef struct bf_epilogue {
    const float* bias;      /* [N] or NULL: v += bias[n]                    */
    const float* colscale;  /* [N] or NULL: v  = v * colscale[n] + colshift[n] */
    const float* colshift;  /* [N] or NULL                                   */
    int32_t aux_mode;       /* BF_AUX_* */
    const void* aux;        /* activation dtype, [M][ld_aux] */
    int64_t ld_aux;
    int32_t out_mode;       /* BF_OUT_* */
    void* c;                /* output */
    int64_t ldc;
    int32_t seglen;         /* output scatter (same addressing as bf_operand) */
    int64_t segstride;
    int32_t gw, gh, gc;
    void* gelu_out;         /* optional second output (activation dtype, same addressing as c): gelu(value stored to c) */
    float* colsum;          /* token-reduction form only (A outer-contiguous): colsum[m] += sum_k A[k][m] (bias gradient) */
    const float* rowscale;  /* optional per-row-group factor applied to the value before aux / store (stochastic depth) */
    int32_t rows_per_group; /* group = row / rows_per_group */
} bf_epilogue;

/* C[M,N] (+)= epi( sum_k pro(A)[m,k] * pro(B)[n,k] ).  splitk > 1 requires BF_OUT_ATOMIC_F32. */
int bf_gemm(int dtype, int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E,
            int splitk, bf_stream_t stream);

/* Data-gradient GEMM fused with the InstanceNorm backward that consumes it (whole-frame row tiles, gemm_frame.hip):
 *   dy = A[M][K] @ B[K][N] (both row-major), frames of S consecutive rows; per (frame, column):
 *   out = rstd*w*(dy - s1/S - xhat*s2/S) [+ add],  s1 = sum dy, s2 = sum dy*xhat, xhat = (x - mean)*rstd;  ws[(f*N+n)*2..] = {s1, s2}.
 * Replaces the conv1x1 backward + InstanceNorm2d backward pair of layers/attention.py:77-78,120-121,208-210,298-299 (autograd).
 * Returns 0 when done, 1 when the shape is not covered (bf16, S = 144, M % 144 = N % 128 = K % 64 = 0): the caller then runs
 * bf_gemm + bf_in_bwd; < 0 on error. */
int bf_gemm_inbwd_frames(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                         const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                         const float* fscale /* optional: dy of frame f is multiplied by fscale[f / fdiv] (stochastic depth) */, int fdiv,
                         bf_stream_t stream);
/* ... and the backward of a SECOND InstanceNorm applied to `out` in the same launch: the norm whose output gradient `out` is (in the model: the
 * MLP-branch norm of the spatial stage in front of a temporal stage, autograd of layers/attention.py:305-317 behind :77-78).  cz: that norm's
 * input rows, cmean / crstd its statistics [frames][N], cw its weight, cg an optional post scale [frames / cgdiv][N] (layer scale, or the
 * stochastic-depth table).  cdz = crstd cw cg (out - (s1 + xh s2) / S), xh = (cz - cmean) crstd; partials {s1, s2} to cws (ws layout; cws may
 * be null: the sums are then not kept).  cz / cdz are [M][N] with row stride N (the x / out stride).
 * Returns 1 (nothing launched) where the two-frames-per-tile kernel does not apply (S != 144, odd frame counts, fp32). */
int bf_gemm_inbwd_frames_chain(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                               const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                               const float* fscale, int fdiv, const void* cz, void* cdz, const float* cmean, const float* crstd,
                               const float* cw, const float* cg, int cgdiv, float* cws, bf_stream_t stream);

/* The forward twin: out[M][N] = lin(A[M][K] @ Bt[K][N]) [+ add], lin(v) = ((v + bias) * colscale + colshift) * fscale[frame / fdiv], frames of
 * S = 144 consecutive rows, followed IN THE SAME LAUNCH by up to two InstanceNorm2d over the frame (layers/attention.py:77,208 norm1;
 * 312-317 the MLP-branch norm + layer scale + residual): n1 (optional): y = resid + g * IN(out), written to n1->out; n2 (optional): the NEXT
 * stage's opening norm of the last tensor written, n2->out = IN(.) * w + b (no g, no resid).  Each leaves mean / rstd / sc / sh [frames][N]
 * as bf_in_stats does.  Bt is the weight TRANSPOSED ([in][out], N contiguous).  Replaces bf_gemm + bf_in_stats (+ bf_affine_apply) -- one
 * read of the activation and one launch per norm less -- with identical bits.  Returns 0 when done, 1 when the shape is not covered
 * (bf16, S = 144, M % 288 = N % 128 = K % 64 = 0, K >= 128, 16-byte aligned operands), < 0 on error. */
typedef struct bf_frame_norm {
    const float *w, *b;             /* affine [N] */
    const float* g; int32_t gdiv;   /* optional post scale [frames / gdiv][N] */
    float *mean, *rstd, *sc, *sh;   /* [frames][N] */
    const void* resid;              /* optional [M][N] */
    void* out;                      /* [M][N] */
} bf_frame_norm;
int bf_gemm_fwd_frames(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* Bt, int64_t ldb, const float* bias,
                       const float* colscale, const float* colshift, const float* fscale, int fdiv, const void* add, void* out, int S,
                       const bf_frame_norm* n1, const bf_frame_norm* n2, bf_stream_t stream);

/* Token-reduction GEMM (weight gradient of a 1x1 conv / Linear: autograd of layers/attention.py:78,121,210,299, linear_layers.py:18-25):
 *   out[Nout][Kin] = (accumulate ? out : 0) + sum_tok dy[tok][Nout] * x[tok][Kin],  colsum[Nout] likewise + sum_tok dy[tok][:]  (optional)
 * dy [M][ldy], x [M][ldx] token-major bf16.  Few long token slices, one 128 x 128 tile x slice per workgroup, partial tiles to fp32
 * slabs in `ws` (bf_gemm_tokred_ws_floats), summed in slice order by a second launch: bit-reproducible, no float atomics.
 * Returns 0 when done, 1 when the shape is not covered (bf16, Nout % 128 = Kin % 128 = M % 64 = 0): the caller then uses bf_gemm's
 * outer-contiguous form; < 0 on error. */
int bf_gemm_tokred(int dtype, int Nout, int Kin, int64_t M, const void* dy, int64_t ldy, const void* x, int64_t ldx, float* out,
                   int accumulate, float* colsum, float* ws, int64_t ws_floats, bf_stream_t stream);
int64_t bf_gemm_tokred_ws_floats(int Nout, int Kin, int64_t M);

/* Whole-frame forward projection for inference (frame_fwd.hip): out[f*S + s][:] = epi( IN(A)[f*S + s][:K] @ W[:N][:K]^T ), one workgroup
 * per (frame of S = 144 tokens, block of output columns).  Replaces, in ONE launch each, the pairs the reference runs as separate modules:
 *   norm_w/norm_b != NULL: nn.InstanceNorm2d(affine) in front of the 1x1 conv (layers/attention.py:77-78 / 208-210 norm1 + input_head,
 *     120-121 norm2 + output_head); K = 384 only (the frame's operand is LDS-resident);
 *   en_w/en_b/en_g != NULL: the InstanceNorm behind fc2 with layer scale and residual, out = resid + en_g * IN(A @ W^T + bias)
 *     (layers/attention.py:316-322);
 *   otherwise v = acc + bias; v = v * colscale + colshift (if given); v += resid (if given); v = gelu(v) (if gelu);
 *   out_n != NULL: a second output, the NEXT layer's InstanceNorm of `out` (its norm1: out_n = IN(out) * next_w + next_b, statistics
 *     over the stored bf16 values) -- the tile holds whole-frame columns of `out`, so the next projection needs no norm in front.
 * A [frames*S][lda], W [N][ldw] (K-contiguous), resid / out [frames*S][ld*], all bf16.  Statistics are summed in bf_in_stats' order and
 * the products in bf_gemm's: results equal the separate launches bit for bit.  Returns 0 when done, 1 when the shape is not covered
 * (bf16, S = 144, N % 32 = 0, K = 384 or (no norm in front) K % 64 = 0), < 0 on error. */
int bf_frame_linear(int dtype, int frames, int S, int K, int N, const void* A, int64_t lda, const void* W, int64_t ldw,
                    const float* norm_w, const float* norm_b, const float* bias, const float* colscale, const float* colshift,
                    const void* resid, int64_t ldr, int gelu, const float* en_w, const float* en_b, const float* en_g,
                    void* out, int64_t ldo,
                    const float* next_w, const float* next_b, void* out_n, int64_t ldn, /* optional second output, see above */
                    bf_stream_t stream);

/* ---------------------------------------------------------------- kernel-level entry points (unit-testable) */

/* InstanceNorm statistics over the S tokens of each frame, per channel (two-pass, fp32):
 * mean/rstd [frames][C]; sc = rstd*w (*g), sh = (b - mean*rstd*w) (*g + gb): the affine the consumer GEMM applies.
 * g/gb (optional): [frames/gdiv][C] post scale / shift (FiLM gamma/beta per batch element, or a layer scale). */
int bf_in_stats(int dtype, const void* x, int frames, int S, int C, const float* w, const float* b, const float* g, int gdiv,
                const float* gb, float* mean, float* rstd, float* sc, float* sh,
                float* ws /* optional, bf_in_ws_floats(): long frames are reduced in slices (frames x slices workgroups) */,
                bf_stream_t stream);
/* floats of workspace bf_in_stats / bf_in_bwd want for this problem (2*frames*C, plus slice partials for long frames) */
int64_t bf_in_ws_floats(int dtype, int frames, int S, int C);
/* out = [resid +] z * sc[f,c] + sh[f,c]   (sh may be NULL = 0) */
int bf_affine_apply(int dtype, const void* z, const void* resid, const float* sc, const float* sh, void* out, int64_t nrows,
                    int S, int C, bf_stream_t stream);
/* backward of y = act(xhat*w + b) [*g]; dx = ... [+ add]; dw/db/dg/dgb accumulate (fp32 atomics) */
int bf_in_bwd(int dtype, const void* dy, const void* x, const void* add, void* dx, int frames, int S, int C, const float* mean,
              const float* rstd, const float* w, const float* b, const float* g, int gdiv, int gelu, float* dw, float* db,
              float* dg, float* dgb, float* ws /* optional, bf_in_ws_floats(): per-frame partials + reduce instead of atomics */,
              bf_stream_t stream);
/* out[c] += scale[c] * sum_rows x[row][c] */
int bf_colsum(int dtype, const void* x, int64_t nrows, int C, const float* scale, float* out, bf_stream_t stream);

/* Strided small-sequence attention on the head-interleaved QKV tensor [N][3E] (channel = head*3d + {q,k,v}*d + e):
 * token(l) of sequence s = (s / inner) * outer_stride + (s % inner) * inner_stride + l * tok_stride.
 * q/k LayerNorm(d) + q k^T d^-1/2 + T5 bias (emb [32][heads] or NULL) + softmax + high-frequency rescale
 * (hscale [heads] or NULL) + P V.  out [N][E] = (accumulate ? out : 0) + out_scale * result. */
int bf_attn_fwd(int dtype, const void* qkv, void* out, int64_t nseq, int L, int64_t inner, int64_t outer_stride,
                int64_t inner_stride, int64_t tok_stride, int heads, int d, const float* qw, const float* qb, const float* kw,
                const float* kb, const float* emb, const float* hscale, float out_scale, int accumulate, bf_stream_t stream);
/* AxialAttentionBlock's attention core (layers/attention.py:212-297): attention along W with hscale_x, along H with hscale_y, both on
 * the shared q/k LayerNorm and T5 table, out = (xx + xy) / 2.  qkv [frames*h*w][3E] token-major, out [frames*h*w][E].  One launch
 * where h, w <= 16 (bf16), otherwise the two bf_attn_fwd passes; the results are bit-identical either way. */
int bf_attn_axial_fwd(int dtype, const void* qkv, void* out, int64_t frames, int h, int w, int heads, int d, const float* qw,
                      const float* qb, const float* kw, const float* kb, const float* emb, const float* hscale_x,
                      const float* hscale_y, bf_stream_t stream);
/* The same followed, in the same launch, by AxialAttentionBlock.norm2 (InstanceNorm2d over the frame, layers/attention.py:298): also
 * writes out_n = (out - mean) * rstd * norm_w + norm_b and mean / rstd / sc / sh [frames][E] exactly as bf_in_stats lays them out.
 * Returns 0 when done, 1 when the one-launch form does not cover the shape (h, w <= 16, bf16): the caller then runs bf_attn_axial_fwd
 * and bf_in_stats + bf_affine_apply. */
int bf_attn_axial_norm_fwd(int dtype, const void* qkv, void* out, void* out_n, int64_t frames, int h, int w, int heads, int d,
                           const float* qw, const float* qb, const float* kw, const float* kb, const float* emb, const float* hscale_x,
                           const float* hscale_y, const float* norm_w, const float* norm_b, float* mean, float* rstd, float* sc,
                           float* sh, bf_stream_t stream);
/* Backward of bf_attn_fwd.  accumulate: 0 = dqkv is written, 1 = added to.  Two passes over the SAME tokens (the axial block's W and H
 * attention, layers/attention.py:218-277, share one q / k LayerNorm) may run that LayerNorm's backward -- linear in its incoming gradient --
 * once: first pass accumulate = 2 (q / k columns of dqkv receive the raw gradient with respect to the LayerNorm outputs; no LayerNorm
 * parameter sums), second pass accumulate = 5 (adds the raw values in front of its LayerNorm backward, v as with 1).  bf16 MFMA path only. */
int bf_attn_bwd(int dtype, const void* qkv, const void* dout, void* dqkv, int64_t nseq, int L, int64_t inner,
                int64_t outer_stride, int64_t inner_stride, int64_t tok_stride, int heads, int d, const float* qw,
                const float* qb, const float* kw, const float* kb, const float* emb, const float* hscale, float* dqw,
                float* dqb, float* dkw, float* dkb, float* demb, float* dhscale, float out_scale, int accumulate,
                float* ws, int64_t ws_floats, /* optional workspace for the block-partial parameter gradients (NULL: atomics) */
                bf_stream_t stream);

/* test hook: route bf16 attention through the generic fp32-VALU kernel instead of the MFMA kernel */
/* Weight-gradient GEMMs run on a library-owned side stream that every stage joins before it returns.  bf_side_defer(1) lets a
 * temporal / spatial backward return with its LAST weight-gradient GEMM still in flight: the next stage call on the stream joins it
 * (before it reuses what that GEMM reads), as does bf_side_join().  A caller that opts in must call bf_side_join(stream) before it
 * reads parameter gradients outside the library (optimizer step, all-reduce of the stage's bucket).  Default: off. */
void bf_side_defer(int on);
int bf_side_join(bf_stream_t stream);
void bf_debug_force_generic_attn(int on);
/* test hook: 0 = every weight-gradient slab sum runs as a launch of its own right behind its GEMM; 1 (default) = inside the stage backwards it
 * rides in the NEXT token-reduction launch (extra workgroups) -- the results are bit-identical either way */
void bf_debug_tokred_fold(int on);

int bf_im2col_nchw(int dtype, const float* x, void* out, int frames, int C, int H, int W, int Kp, bf_stream_t stream);
int bf_col2im_nchw(int dtype, const void* g, float* dx, int frames, int C, int H, int W, int Kp, bf_stream_t stream);
/* lossbuf (here, in bf_debed_last and bf_lploss_finalize): [frames][Co][2][BF_LOSS_LIMBS] int64, zeroed by the caller: the sums of (pred - y)^2
 * and of y^2 as integers -- base-2^48 digits in units of 2^-112, one 64-bit integer atomic per non-zero digit of a partial.  Integer adds are
 * order independent, so the loss and its gradient are bit-reproducible run to run; the limbs cover partial sums in [2^-112, 2^127), i.e. the
 * range of the reference's float sums (utils/losses.py:79-89): un-normalised fields and nearly converged numerators included. */
#define BF_LOSS_LIMBS 5
int bf_pm2nchw(const float* pm, float* pred, const float* y, float* lossbuf, int frames, int Co, int h, int w, int Np,
               bf_stream_t stream);
/* Last HMLPDebed stage in one pass (layers/patching.py:92-104: InstanceNorm affine + GELU on the input rows, ConvTranspose2d(k=2,s=2)
 * to the NCHW prediction) with the relative-L2 partial sums of bf_pm2nchw; act [frames*h*w][Ci], sc/sh [frames][Ci], wc [Ci][Np] with
 * n = co*4 + ky*2 + kx.  Returns 1 (nothing launched) for shapes it does not take: fp32, Np != 16, Co > 4, Ci % 32, Ci > 128, w % 16. */
int bf_debed_last(int dtype, const void* act, const float* sc, const float* sh, const void* wc, float* pred, const float* y,
                  float* lossbuf, int frames, int Ci, int Co, int h, int w, int Np, bf_stream_t stream);
/* ... and its backward: dpm [frames*h*w][Np] (bf_nchw2pm's output, kept for the weight-gradient GEMM) and dact[p][ci] = sum_n dpm[p][n]*wc[ci][n]
 * in one pass.  Same declined shapes (returns 1). */
int bf_debed_last_bwd(int dtype, const float* dpred, const float* pred, const float* y, const float* coef, const float* gscale,
                      const void* wc, void* dpm, void* dact, int frames, int Ci, int Co, int h, int w, int Np, bf_stream_t stream);
/* ... and with the InstanceNorm + GELU in front of the stage folded in (layers/patching.py:92-104 under autograd): dpm as above, and
 * dx [frames*h*w][Ci] = the gradient of the RAW map ymap in front of that InstanceNorm (mean / rstd [frames][Ci], affine in_w / in_b) --
 * two passes over ymap around the frame-wide sums, the rank-16 gradient map itself is never stored.  d_in_w / d_in_b are accumulated
 * (optional).  ws: bf_in_ws_floats(dtype, frames, h*w, Ci) floats.  Same declined shapes (returns 1), also for a workspace too small. */
int bf_debed_last_bwd_norm(int dtype, const float* dpred, const float* pred, const float* y, const float* coef, const float* gscale,
                           const void* wc, void* dpm, const void* ymap, const float* mean, const float* rstd, const float* in_w,
                           const float* in_b, void* dx, float* d_in_w, float* d_in_b, int frames, int Ci, int Co, int h, int w, int Np,
                           float* ws, int64_t ws_floats, bf_stream_t stream);
/* First HMLPEmbed stage (layers/patching.py:30-48: Conv2d(k=2, s=2, bias=False) on the NCHW fp32 clip) in one pass: patches [P][Kp]
 * (k = c*4 + ky*2 + kx, what bf_im2col_nchw writes; kept for the weight gradient) and y0[p][co] = sum_k patches[p][k] * wc[co][k].
 * Same kernel and declined shapes as bf_debed_last_bwd (returns 1). */
int bf_embed_first(int dtype, const float* x, const void* wc, void* patches, void* y0, int frames, int C0, int cin, int h2, int w2,
                   int Kp, float* stat_part, bf_stream_t stream);
/* y0 may be NULL when stat_part is given: the map is then not stored at all -- its consumers rebuild its rows from `patches` and `wc`
 * (bf_gather_gemm_rebuilt, bf_gather_wgrad_rebuilt, bf_embed_tail_bwd with y0 = NULL), which is 226 MB less to write and three reads of it
 * less per training step at the bench shape. */
/* stat_part (optional): {mean, centred second moment} of y0 per 256-row slice, [frames][ceil(h2*w2/256)][C0][2]; finished by
 * bf_in_stats_merge_slices(..., rows = 256, ws) with ws + 2*frames*C0 == stat_part (the layout bf_in_stats uses for long frames). */
int bf_in_stats_merge_slices(int dtype, int frames, int S, int C, int rows, const float* w, const float* b, const float* g, int gdiv,
                             const float* gb, float* mean, float* rstd, float* sc, float* sh, float* ws, bf_stream_t stream);
/* The tail of the HMLPEmbed backward when the input needs no gradient (layers/patching.py:24-56 under autograd): the stage-1 data
 * gradient dy1 [frames*gh1*gw1][C1] times w1c [C1][4*C0] (the stage-1 weight, columns (2*ky + kx)*C0 + c), GELU', the stage-0 InstanceNorm
 * backward and the stage-0 weight gradient in one pass that never writes the [frames*4*gh1*gw1][C0] gradient map.  y0: raw stage-0
 * output, or NULL (its rows are then rebuilt from the patch rows: y0 = W0 . patch); patches [..][Kp] (bf_embed_first), w0c [C0][Kp], sc / sh / mean / rstd [frames][C0] of the stage-0 InstanceNorm, in_w its
 * weight.  dwprep [C0][Kp] is WRITTEN (bf_wgrad_unprep folds it into the gradient), d_in_w / d_in_b [C0] are accumulated (optional).
 * ws: bf_embed_tail_ws_floats floats.  Returns 1 (nothing launched) for shapes it does not take: fp32, C0 != 96, C1 not in {96, 192}, Kp != 16,
 * gw1 % 16, gh1*gw1 % 128, or a workspace that is too small. */
/* The 2x2 / stride-2 stages at 96 channels as one streaming GEMM: out[p][n] = sum_{q, c} f(map[pixel(p, q)][c]) * W[(q, c)][n], p over the
 * coarse grid [frames][gh][gw], pixel(p, q) = (2y + q/2, 2x + q%2) of the fine grid [frames][2gh][2gw][C0].  f = GELU(x * sc + sh)
 * (sc / sh [frames][C0]: the HMLPEmbed convolutions after the first, layers/patching.py:30-56) or the identity (sc = sh = NULL: the data
 * gradients of the HMLPDebed transposed convolutions).  w: [4*C0][N] (w_kn = 1) or [N][4*C0] (w_kn = 0).  Returns 1 (nothing launched)
 * for shapes it does not take: fp32, C0 != 96, N != 96, gh*gw % 32. */
int bf_gather_gemm(int dtype, const void* map, const void* w, int w_kn, const float* sc, const float* sh, void* out, int frames, int gh,
                   int gw, int C0, int N, bf_stream_t stream);
/* ... and the transposed direction (HMLPDebed's ConvTranspose2d(k=2, s=2) stages, layers/patching.py:80-104): map[pixel(p, q)][c] =
 * sum_k f(a[p][k]) * w[q*C0 + c][k] (w_kn = 0; or w [K][4*C0] with w_kn = 1), a [frames*gh*gw][K], map [frames][2gh][2gw][C0].  stat_part (optional): {mean, centred second moment} of the
 * map as stored per 128-pixel slice, [frames][gh*gw/32][C0][2]; finished by bf_in_stats_merge_slices(..., rows = 128, ws) with
 * ws + 2*frames*C0 == stat_part.  Returns 1 (nothing launched) for: fp32, K != 96, C0 != 96, gh*gw % 32. */
int bf_scatter_gemm(int dtype, const void* a, const void* w, int w_kn, const float* sc, const float* sh, void* map, float* stat_part, int frames,
                    int gh, int gw, int K, int C0, bf_stream_t stream);
/* ... and their weight gradients: dW[(q, c)][k] = sum_p ff(fine[pixel(p, q)][c]) * fc(coarse[p][k]), fine [frames][2gh][2gw][C0], coarse
 * [frames*gh*gw][Kc]; at most one side through GELU(x * sc + sh) ([frames][96] each; the embed stages transform the fine side, the debed
 * stages the coarse side).  out is WRITTEN: [4*C0][Kc] (transposed = 0) or [Kc][4*C0] (transposed = 1); per-workgroup slabs summed in a
 * fixed order (bit-reproducible).  ws: bf_gather_wgrad_ws_floats floats for the preferred launch (anything from frames * 36864 floats
 * up is accepted: fewer, longer runs).  Returns 1 (nothing launched) for: fp32, C0 != 96, Kc != 96,
 * gh*gw % 32, both sides transformed, a workspace too small. */
int64_t bf_gather_wgrad_ws_floats(int frames, int gh, int gw);
int bf_gather_wgrad(int dtype, const void* fine, const void* coarse, const float* fsc, const float* fsh, const float* csc, const float* csh,
                    float* out, int transposed, int frames, int gh, int gw, int C0, int Kc, float* ws, int64_t ws_floats, bf_stream_t stream);
/* bf_gather_gemm / bf_gather_wgrad with the fine map given as its factors, map[pixel][c] = sum_k patches[pixel][k] * w0c[c][k] (Kp = 16: the
 * output of bf_embed_first), rebuilt per tile and rounded to bf16 like a stored map; the fine side is transformed by GELU(x * sc + sh). */
int bf_gather_gemm_rebuilt(int dtype, const void* patches, const void* w0c, const void* w, int w_kn, const float* sc, const float* sh, void* out,
                           int frames, int gh, int gw, int C0, int N, bf_stream_t stream);
int bf_gather_wgrad_rebuilt(int dtype, const void* patches, const void* w0c, const void* coarse, const float* fsc, const float* fsh, float* out,
                            int transposed, int frames, int gh, int gw, int C0, int Kc, float* ws, int64_t ws_floats, bf_stream_t stream);
int64_t bf_embed_tail_ws_floats(int frames, int gh1, int gw1, int C0, int Kp);
int bf_embed_tail_bwd(int dtype, const void* dy1, const void* w1c, const void* y0, const void* patches, const void* w0c, const float* sc,
                      const float* sh, const float* mean, const float* rstd, const float* in_w, float* dwprep, float* d_in_w, float* d_in_b,
                      int frames, int gh1, int gw1, int C1, int C0, int Kp, float* ws, int64_t ws_floats, bf_stream_t stream);
int bf_lploss_finalize(const float* lossbuf, int frames, int Co, float* loss, float* coef, bf_stream_t stream);
int bf_nchw2pm(int dtype, const float* dpred, const float* pred, const float* y, const float* coef, const float* gscale,
               void* dpm, int frames, int Co, int h, int w, int Np, bf_stream_t stream);
int bf_wprep(int dtype, int mode, const float* src, void* dst, int R, int K, int Kp, bf_stream_t stream);
int bf_wgrad_unprep(int mode, const float* gsrc, float* gdst, int R, int K, int Kp, int transposed, bf_stream_t stream);
/* gb: [2][B][E] (gamma block, beta block) */
int bf_film_net_fwd(const float* cond, const float* lnw, const float* lnb, const float* W, const float* bias, float* gb,
                    float* chat, float* crstd, int B, int P, int E2, bf_stream_t stream);
int bf_film_net_bwd(const float* dgb, const float* chat, const float* lnw, const float* lnb, const float* W, float* dW,
                    float* dbias, float* dlnw, float* dlnb, int B, int P, int E2, bf_stream_t stream);
int bf_adamw(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float beta1, float beta2, float eps,
             float wd, float gscale, bf_stream_t stream);
/* Batch of normalised clips from device-resident trajectories src [fields][frames][H][W] (fp32):
 * out[b][t][c][yo][xo] = (src[field[c]][first[b] + t0 + t][ys][xs] - diff[c]) / div[c], (ys, xs) = nearest-neighbour source pixel of
 * (yo, xo) when Ho x Wo < H x W (F.interpolate(mode="nearest") index rule), identity otherwise.  out is (B, T, C, Ho, Wo) fp32. */
int bf_clip_gather(const float* src, int64_t field_stride, const int32_t* field, const int64_t* first, int t0, const float* diff,
                   const float* div, float* out, int B, int T, int C, int H, int W, int Ho, int Wo, bf_stream_t stream);
/* The same for a whole training batch in one launch, indexed by SAMPLE number on the device: idx [B] (int64; clamped to 0 .. nsamples - 1 in the
 * kernel, so a bad index reads another sample, never past a table), first_tab / file_tab [nsamples] = absolute first input frame and file of every sample.  in_out (B, Tin, Cin, Ho, Wo) = frames
 * first .. first + Tin - 1 of the input fields, out_out (B, Tout, Cout, Ho, Wo) = the Tout frames behind them of the output fields (each with
 * its own field ids and constants), fluid_out (B, P) = fluid_tab[file_tab[idx[b]]] (optional: fluid_out may be null).
 * Replaces BubbleForecast.__getitem__ + the DataLoader's collate for a batch (bubbleformer/data/dataset.py:120-182). */
int bf_clip_gather_batch(const float* src, int64_t field_stride, const int64_t* idx, int64_t nsamples, const int64_t* first_tab, const int32_t* in_field,
                         const float* in_diff, const float* in_div, int Cin, int Tin, float* in_out, const int32_t* out_field,
                         const float* out_diff, const float* out_div, int Cout, int Tout, float* out_out, const float* fluid_tab,
                         const int64_t* file_tab, int P, float* fluid_out, int B, int H, int W, int Ho, int Wo, bf_stream_t stream);
/* Normalisation statistics of device-resident trajectories (BubbleForecast.normalize, bubbleformer/data/dataset.py:74-117: mean / std / min /
 * max of every full field of every file): segment i = seg_len[i] floats at src + seg_begin[i] (both arrays on the DEVICE);
 * out[i] = {sum, sum of squares, min, max} in fp64, summed in a fixed order (bit-reproducible).  ws: bf_field_stats_ws_doubles(nseg) doubles. */
int64_t bf_field_stats_ws_doubles(int nseg);
int bf_field_stats(const float* src, const int64_t* seg_begin, const int64_t* seg_len, int nseg, double* out, double* ws, bf_stream_t stream);
/* Rollout physics metrics (scripts/inference.py; utils/losses.py:5-15, utils/heatflux.py:17-38).
 * bf_eikonal_sum: *out (fp64, caller zeroes) += sum over frames x H x W of (|grad phi| - 1)^2, gradients as torch.gradient(spacing=dx).
 * bf_heatflux_rows: flux[t] = mean over the W bottom-row cells of [|x_c| <= 5 and dfun < 0] * (heater_temp - temp) * 0.054 / (dx * lc),
 * x_c = x_min + (i + 0.5) * dx; dfun / temp point at row 0 of frame 0, frames are frame_stride elements apart. */
int bf_eikonal_sum(const float* phi, int64_t frames, int H, int W, float dx, double* out, bf_stream_t stream);
/* out[frame] = mean over the frame of | |grad phi| - 1 |, central differences at spacing dx, borders replicate their neighbour's
 * gradient: `get_eikonal_loss` of scripts/inference_autoregressive.ipynb (the rollout notebook's per-time-step SDF score). */
int bf_eikonal_l1_frames(const float* phi, int64_t frames, int H, int W, float dx, float* out, bf_stream_t stream);
int bf_heatflux_rows(const float* dfun, const float* temp, int64_t frames, int64_t frame_stride, int W, float x_min, float dx,
                     float heater_temp, float lc, float* flux, bf_stream_t stream);
/* Lion: p *= 1 - lr*wd; p -= lr*sign(beta1*m + (1-beta1)*g); m = beta2*m + (1-beta2)*g   (g is multiplied by gscale first) */
int bf_lion(float* p, const float* g, float* m, int64_t n, float lr, float beta1, float beta2, float wd, float gscale,
            bf_stream_t stream);

/* ---------------------------------------------------------------- stage-level entry points (what the nn.Modules call) */

typedef struct bf_dims {
    int32_t dtype;              /* BF_DTYPE_* */
    int32_t B, T, h, w;         /* batch, frames per clip, token grid */
    int32_t E, heads;           /* embed dim, attention heads (d = E / heads) */
    int32_t attn_scale, feat_scale;
    int32_t patch;              /* patch size (power of two) */
    int32_t cin, cout;          /* input / output fields */
    int32_t nfluid;             /* FiLM parameters, 0 = unconditioned AViT */
} bf_dims;

/* Parameters of one AttentionBlock (layers/attention.py:25-64) in state_dict order; the gradient struct mirrors it. */
typedef struct bf_temporal_params {
    float *gamma, *attn_scale_factor, *norm1_w, *norm1_b, *norm2_w, *norm2_b, *input_head_w, *input_head_b,
          *output_head_w, *output_head_b, *qnorm_w, *qnorm_b, *knorm_w, *knorm_b, *rel_pos_emb;
} bf_temporal_params;
/* Parameters of one AxialAttentionBlock (layers/attention.py:137-197) */
typedef struct bf_spatial_params {
    float *gamma_att, *gamma_mlp, *attn_scale_factor_x, *attn_scale_factor_y, *low_freq_scalar, *high_freq_scalar,
          *norm1_w, *norm1_b, *norm2_w, *norm2_b, *input_head_w, *input_head_b, *output_head_w, *output_head_b,
          *qnorm_w, *qnorm_b, *knorm_w, *knorm_b, *rel_pos_emb, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *mlp_norm_w, *mlp_norm_b;
} bf_spatial_params;
/* HMLPEmbed (+ FiLMMLP): up to 5 stages (patch <= 32) */
#define BF_MAX_STAGES 5
typedef struct bf_embed_params {
    float* conv_w[BF_MAX_STAGES];
    float* in_w[BF_MAX_STAGES];
    float* in_b[BF_MAX_STAGES];
    float *film_ln_w, *film_ln_b, *film_w, *film_b;   /* NULL for AViT */
} bf_embed_params;
typedef struct bf_debed_params {
    float* conv_w[BF_MAX_STAGES];
    float* in_w[BF_MAX_STAGES];    /* last stage unused */
    float* in_b[BF_MAX_STAGES];
} bf_debed_params;

/* Bytes of the per-stage activation record the forward fills and the backward reads, and of the transient scratch
 * arena any stage call needs (forward or backward).  Both are plain device allocations owned by the caller. */
int64_t bf_temporal_saved_bytes(const bf_dims* d);
int64_t bf_spatial_saved_bytes(const bf_dims* d);
int64_t bf_embed_saved_bytes(const bf_dims* d);
int64_t bf_debed_saved_bytes(const bf_dims* d);
int64_t bf_scratch_bytes(const bf_dims* d);

/* Parameter preparation (bf16 weight copies, out-projection fold, MLP-branch stochastic-depth table) of n trunk stages in one launch per
 * 12 stages, written into each stage's `saved` record; kinds[i] 0 = temporal (params[i]: bf_temporal_params*), 1 = spatial
 * (bf_spatial_params*); drop_mlp[i]: that spatial stage's MLP-branch factors or NULL.  bf_stage_prepared(1) just before a stage forward
 * tells it to use them instead of launching its own preparation (the flag is consumed by that call).  Returns 1 in fp32 mode. */
int bf_prep_stages(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, void* const* saved,
                   const float* const* drop_mlp, bf_stream_t stream);
void bf_stage_prepared(int on);
/* Inference forward of n trunk stages in one call (scripts/inference.py:239-252 runs FiLMConditionedAViT.forward under no_grad, one clip
 * at a time; the stages are SpaceTimeBlock's temporal and axial blocks, models/axial_vit.py:18-60).  Nothing is saved for a backward, the
 * InstanceNorms run inside the whole-frame projection kernels (bf_frame_linear) and the bf16 weight copies / out-projection folds live in
 * a caller-owned arena (bf_trunk_eval_weights_bytes) that bf_trunk_eval_prepare fills once per set of weights: call it again after the
 * parameters change.  kinds / params as bf_prep_stages; x, out [N][E]; scratch: bf_scratch_bytes.  No stochastic depth (eval).
 * bf_trunk_eval_prepare / _fwd return 0 when done, 1 when the shape is not covered (bf16, 12 x 12-token frames, E = 384: the caller then
 * runs the stage forwards), < 0 on error. */
int64_t bf_trunk_eval_weights_bytes(const bf_dims* dims, int n, const int32_t* kinds);
int bf_trunk_eval_prepare(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, void* weights, bf_stream_t stream);
int bf_trunk_eval_fwd(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, const void* weights, const void* x,
                      void* out, void* scratch, bf_stream_t stream);
/* The n trunk stages of a TRAINING step in one native call per direction (the reference runs them as a Python loop over SpaceTimeBlock,
 * models/axial_vit.py:234-235 -> :58-63): kinds / params as bf_prep_stages; saved[i]: stage i's record (bf_temporal_saved_bytes /
 * bf_spatial_saved_bytes); acts[i]: stage i's output [N][E] (acts[n - 1] is the trunk's output; the backward reads acts[0 .. n - 2] as stage
 * inputs); drop_a[i] / drop_b[i]: the stage's stochastic-depth factors or NULL (the arrays themselves may be NULL).  The forward prepares
 * the weights (bf_prep_stages) and chains every stage's opening InstanceNorm into the launch in front of it.  Backward: grads[i] mirrors
 * params[i] (gradients accumulate); dout = d(acts[n - 1]), dx = d(x); gbuf3: three [N][E] buffers for the gradients between stages;
 * stage_done(i, user) (optional) is called on the host after stage i's backward has been enqueued. */
typedef void (*bf_stage_done_fn)(int stage, void* user);
int bf_trunk_train_fwd(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, void* const* saved,
                       const float* const* drop_a, const float* const* drop_b, const void* x, void* const* acts, void* scratch, bf_stream_t stream);
int bf_trunk_train_bwd(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, const void* const* grads, void* const* saved,
                       const float* const* drop_a, const float* const* drop_b, const void* x, void* const* acts, const void* dout,
                       void* const* gbuf3, void* dx, void* scratch, bf_stage_done_fn stage_done, void* user, bf_stream_t stream);
/* x, out, dout, dx: [N][E] activations.  Gradients ACCUMULATE into `g` (zero it first). */
/* Chained stage heads (optional, forward only): arm the NEXT temporal stage's opening InstanceNorm (norm1 of next_p, written into next_saved)
 * to be computed by the tail of the spatial stage called next, whose output it normalises -- one launch and one read of the activation
 * less per block pair; bf_temporal_fwd on next_saved then skips its own norm1.  Results are bit-identical to the unchained calls.
 * NULL arguments disarm.  A spatial stage that cannot chain (fp32 frames longer than the register cache, ...) leaves the norm to the
 * temporal stage as usual. */
int bf_stage_chain_head(const bf_dims* dims, const bf_temporal_params* next_p, void* next_saved);
/* ... in either direction: next_kind 0 = a temporal stage follows (next_params: bf_temporal_params*), 1 = a spatial stage follows
 * (bf_spatial_params*).  The stage forward called next folds that stage's opening InstanceNorm into its last GEMM's launch
 * (bf_gemm_fwd_frames: the temporal stage's out-projection, the spatial stage's fc2 + MLP-branch norm) where the frame-pair kernel covers
 * the shape, or into its last InstanceNorm launch; the stage on next_saved then skips its own norm1.  Bit-identical to the unchained calls. */
int bf_stage_chain_next(const bf_dims* dims, int next_kind, const void* next_params, void* next_saved);
/* The mirror image in the backward: the temporal stage's last kernel produces the output gradient of the spatial stage in front of it (prev_p,
 * prev_saved; has_drop_mlp: its MLP branch carried stochastic-depth factors), whose backward opens with its MLP-branch InstanceNorm -- armed
 * here, that norm's backward is applied by the temporal backward called next (bf_gemm_inbwd_frames_chain) and bf_spatial_bwd on prev_saved
 * skips it.  bf16, 144-token frames, an even number of frames; otherwise nothing changes.  NULL arguments disarm. */
int bf_stage_chain_tail(const bf_spatial_params* prev_p, const void* prev_saved, int has_drop_mlp);
/* Stochastic depth in the backward (optional, deferred-side-work mode): `factors` [frames / fdiv] are what the NEXT stage backward to be called
 * (a temporal stage) multiplies its incoming gradient by -- the gradient the spatial stage backward called next produces.  Armed, that stage's
 * last kernel also writes the scaled copy and the temporal backward, given the same dout and drop pointers, uses it instead of scaling in a
 * launch of its own.  The flag is consumed by the next bf_spatial_bwd; NULL disarms. */
int bf_stage_next_scale(const float* factors, int fdiv);
/* Stochastic depth (timm DropPath at layers/attention.py:123,309,317): `drop*` are the per-sample factors (0 or 1/keep) the
 * caller drew -- [B] for the temporal block (dim 0 = batch), [B*T] each for the two branches of the axial block -- or NULL. */
int bf_temporal_fwd(const bf_dims* d, const bf_temporal_params* p, const void* x, void* out, void* saved, void* scratch,
                    const float* drop, bf_stream_t s);
int bf_temporal_bwd(const bf_dims* d, const bf_temporal_params* p, const bf_temporal_params* g, const void* x, const void* dout,
                    void* dx, void* saved, void* scratch, const float* drop, bf_stream_t s);
int bf_spatial_fwd(const bf_dims* d, const bf_spatial_params* p, const void* x, void* out, void* saved, void* scratch,
                   const float* drop_att, const float* drop_mlp, bf_stream_t s);
int bf_spatial_bwd(const bf_dims* d, const bf_spatial_params* p, const bf_spatial_params* g, const void* x, const void* dout,
                   void* dx, void* saved, void* scratch, const float* drop_att, const float* drop_mlp, bf_stream_t s);
/* x: (B*T, cin, H, W) fp32 clip; fluid: [B][nfluid] fp32 or NULL; out: [N][E].  dx_in (optional): d(loss)/d(clip). */
int bf_embed_fwd(const bf_dims* d, const bf_embed_params* p, const float* x, const float* fluid, void* out, void* saved, void* scratch, bf_stream_t s);
int bf_embed_bwd(const bf_dims* d, const bf_embed_params* p, const bf_embed_params* g, const void* dout, float* dx_in, void* saved,
                 void* scratch, bf_stream_t s);
/* x: [N][E]; pred: (B*T, cout, H, W) fp32.  If target != NULL the relative-L2 loss (utils/losses.py:67-94 as
 * configured at modules.py:50) is fused: loss[0] and the per-(frame, channel) backward coefficients are produced. */
int bf_debed_fwd(const bf_dims* d, const bf_debed_params* p, const void* x, float* pred, const float* target, float* loss,
                 void* saved, void* scratch, bf_stream_t s);
/* dpred: explicit (B*T, cout, H, W) fp32 gradient, or NULL to use the fused loss backward (pred, target, loss_scale[0] or 1). */
int bf_debed_bwd(const bf_dims* d, const bf_debed_params* p, const bf_debed_params* g, const void* x, const float* dpred,
                 const float* pred, const float* target, const float* loss_scale, void* dx, void* saved, void* scratch,
                 bf_stream_t s);

/* Optional per-launch HIP-event timing on the launch stream (bench.py's roofline leg); off by default. */
void bf_prof_enable(int on);
int bf_prof_report(char* buf, int n);   /* JSON {kernel: {calls, ms, flops, bytes}}; bytes written or -1 */

const char* bf_last_error(void);
int bf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
