// Layout movers at the two ends of the model, FiLM, the relative-L2 loss and weight preparation.
//   im2col_nchw / col2im_nchw : (B*T, C, H, W) fp32 clip  <->  2x2 patch rows [P][Kp] (k = (c, ky, kx))
//   pm2nchw (+ loss partials)  : last debed GEMM output [P][Np] fp32 (n = (co, ky, kx)) -> (B*T, Co, H, W) fp32
//   nchw2pm (+ loss backward)  : d(pred) -> patch-major rows for the debed backward GEMMs
//   wprep / wgrad_unprep       : fp32 state_dict weights -> GEMM operand layout/dtype and back (gradients)
//   film_net fwd/bwd           : LayerNorm(P) + Linear(P, 2E) on B rows
//   adamw                      : fused flat-buffer AdamW
// All are bandwidth-bound; each thread moves 8..16 contiguous bytes where the layout allows.
#include "bf_common.h"
#include "param_reduce.h"

namespace {
constexpr int NT = 256;

// ---------------------------------------------------------------------------- im2col (stage 0 of HMLPEmbed)
// x: [F][C][H][W] fp32.  rows p = (f, y, x) over the (H/2, W/2) grid; row = Kp elements, k = c*4 + ky*2 + kx, zero padded.
template <typename T>
__global__ void __launch_bounds__(NT) im2col_kernel(const float* __restrict__ x, T* __restrict__ out, int C, int H, int W,
                                                   int Kp, long P) {
    const int W2 = W / 2, H2 = H / 2;
    const long total = P * C;   // one thread per (pixel, channel): reads 2 x float2, writes 4 elements
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int xo = (int)(i % W2);
        long t = i / W2;
        const int c = (int)(t % C);        // channel inner to x so that a wave reads contiguous image rows
        t /= C;
        const int yo = (int)(t % H2);
        const long f = t / H2;
        const float* src = x + ((f * C + c) * H + 2 * yo) * (long)W + 2 * xo;
        const float2 r0 = *reinterpret_cast<const float2*>(src);
        const float2 r1 = *reinterpret_cast<const float2*>(src + W);
        const long p = (f * H2 + yo) * W2 + xo;
        T* dst = out + p * Kp + c * 4;
        dst[0] = from_f<T>(r0.x); dst[1] = from_f<T>(r0.y); dst[2] = from_f<T>(r1.x); dst[3] = from_f<T>(r1.y);
    }
}
template <typename T>
__global__ void __launch_bounds__(NT) zero_pad_cols_kernel(T* __restrict__ out, int K, int Kp, long P) {
    const int padw = Kp - K;
    const long total = P * padw;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT)
        out[(i / padw) * Kp + K + (i % padw)] = from_f<T>(0.f);
}
// dx[f][c][2y+ky][2x+kx] = g[p][c*4 + ky*2 + kx]
template <typename T>
__global__ void __launch_bounds__(NT) col2im_kernel(const T* __restrict__ g, float* __restrict__ dx, int C, int H, int W, int Kp, long P) {
    const int W2 = W / 2, H2 = H / 2;
    const long total = P * C;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int xo = (int)(i % W2);
        long t = i / W2;
        const int c = (int)(t % C);
        t /= C;
        const int yo = (int)(t % H2);
        const long f = t / H2;
        const long p = (f * H2 + yo) * W2 + xo;
        const T* src = g + p * Kp + c * 4;
        float* dst = dx + ((f * C + c) * H + 2 * yo) * (long)W + 2 * xo;
        *reinterpret_cast<float2*>(dst) = make_float2(to_f(src[0]), to_f(src[1]));
        *reinterpret_cast<float2*>(dst + W) = make_float2(to_f(src[2]), to_f(src[3]));
    }
}

// ---------------------------------------------------------------------------- loss partial sums, order-independent
// The relative-L2 sums of a (frame, channel) come from dozens of workgroups.  Added as floats they depend on arrival order, and that
// last-bit noise reaches coef = 1 / (F sqrt(num) sqrt(den)), i.e. EVERY element of the loss gradient: in bf16 mode a few rounding flips
// there cascade through the K = 384 contractions of the backward until ~12 % of a gradient tensor's elements differ by an ulp between
// two runs on identical inputs (1.3 % in d(clip); measured).  So the partial sums are added as INTEGERS (exact, associative): a sum is
// LOSS_LIMBS 64-bit limbs holding base-2^48 digits of the value in units of 2^-112; a partial (an fp32 number: 24 significant bits) is cut
// into its digits exactly and each non-zero digit is one 64-bit integer atomic.  Covers every partial in [2^-112, 2^127) -- the whole
// useful fp32 range: un-normalised fields (norm="none" is the dataset default, bubbleformer/data/dataset.py:25) sum to 1e15 and beyond, a
// nearly converged numerator to 1e-12 and below; the reference's float sums (utils/losses.py:79-89) reach both, and a single 2^-32 limb
// (rounds 1-3) returned NaN above 2.7e8 and flushed below 2.3e-10.  lossbuf is [frames][Co][2][LOSS_LIMBS] int64, zeroed by the caller.
// A non-finite, negative or >= 2^127 partial poisons the sum (finalize returns NaN, as the float sum would give inf / NaN).
constexpr int LOSS_LIMBS = BF_LOSS_LIMBS;
constexpr long long LOSS_POISON = 1LL << 62;
__device__ __forceinline__ void loss_accum(float* lossbuf, long slot, float v) {
    unsigned long long* limb = reinterpret_cast<unsigned long long*>(lossbuf) + slot * LOSS_LIMBS;
    if (!(v >= 0.f) || !(v < 1.7014118346046923e38f)) { atomicAdd(limb + LOSS_LIMBS - 1, (unsigned long long)LOSS_POISON); return; }      // NaN, negative, >= 2^127
    double x = (double)v;                                  // exact; every step below is exact too (x has 24 significant bits)
#pragma unroll
    for (int k = LOSS_LIMBS - 1; k >= 1; --k) {
        const double unit = __builtin_ldexp(1.0, 48 * k - 112);
        const long long dgt = __double2ll_rd(x / unit);    // < 2^48
        if (dgt != 0) { atomicAdd(limb + k, (unsigned long long)dgt); x -= (double)dgt * unit; }
    }
    const long long d0 = __double2ll_rn(x * __builtin_ldexp(1.0, 112));      // what is left below 2^-64 (rounded to the 2^-112 grid)
    if (d0 != 0) atomicAdd(limb, (unsigned long long)d0);
}
__device__ __forceinline__ float loss_read(const float* lossbuf, long slot) {
    const long long* limb = reinterpret_cast<const long long*>(lossbuf) + slot * LOSS_LIMBS;
    if (limb[LOSS_LIMBS - 1] >= LOSS_POISON / 2 || limb[LOSS_LIMBS - 1] < 0) return __builtin_nanf("");
    double t = 0.0;
#pragma unroll
    for (int k = LOSS_LIMBS - 1; k >= 0; --k) t += (double)limb[k] * __builtin_ldexp(1.0, 48 * k - 112);      // same integers in -> same bits out
    return (float)t;
}

// ---------------------------------------------------------------------------- pm2nchw + loss partials
// pm: [P][Np] fp32, n = co*4 + ky*2 + kx; pred: [F][Co][H][W]; grid (blocks over h*w pixels, F)
// lossbuf[f][co][0] += sum (pred - y)^2, [1] += sum y^2   (integer limbs, see loss_accum)
__global__ void __launch_bounds__(NT) pm2nchw_kernel(const float* __restrict__ pm, float* __restrict__ pred, const float* __restrict__ y,
                                                    float* __restrict__ lossbuf, int Co, int h, int w, int Np) {
    __shared__ float red[NT / 64][2];
    const int f = blockIdx.y;
    const int H = 2 * h, W = 2 * w;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int co = 0; co < Co; ++co) {
        float num = 0.f, den = 0.f;
        for (int px = blockIdx.x * NT + threadIdx.x; px < h * w; px += gridDim.x * NT) {
            const int xo = px % w, yo = px / w;
            const float4 v = *reinterpret_cast<const float4*>(pm + ((long)f * h * w + px) * Np + co * 4);
            const long o = (((long)f * Co + co) * H + 2 * yo) * W + 2 * xo;
            *reinterpret_cast<float2*>(pred + o) = make_float2(v.x, v.y);
            *reinterpret_cast<float2*>(pred + o + W) = make_float2(v.z, v.w);
            if (y) {
                const float2 y0 = *reinterpret_cast<const float2*>(y + o);
                const float2 y1 = *reinterpret_cast<const float2*>(y + o + W);
                num += (v.x - y0.x) * (v.x - y0.x) + (v.y - y0.y) * (v.y - y0.y) + (v.z - y1.x) * (v.z - y1.x) + (v.w - y1.y) * (v.w - y1.y);
                den += y0.x * y0.x + y0.y * y0.y + y1.x * y1.x + y1.y * y1.y;
            }
        }
        if (y) {
            num = wave_sum(num); den = wave_sum(den);
            __syncthreads();
            if (lane == 0) { red[wave][0] = num; red[wave][1] = den; }
            __syncthreads();
            if (threadIdx.x == 0) {
                float a = 0.f, b = 0.f;
                for (int i = 0; i < NT / 64; ++i) { a += red[i][0]; b += red[i][1]; }
                loss_accum(lossbuf, ((long)f * Co + co) * 2, a);
                loss_accum(lossbuf, ((long)f * Co + co) * 2 + 1, b);
            }
        }
    }
}
// ---------------------------------------------------------------------------- last HMLPDebed stage in one pass (bf16)
// pred[f][co][2y+ky][2x+kx] = sum_ci gelu(act[p][ci] * sc[f][ci] + sh[f][ci]) * wc[ci][co*4 + ky*2 + kx]   (layers/patching.py:92-104:
// the last ConvTranspose2d(k=2, s=2) after InstanceNorm + GELU), plus the relative-L2 partial sums of pm2nchw_kernel.
// K = Ci <= 128 and N = 16: a 128-wide GEMM tile is 7/8 padding and each workgroup's life is a cold prologue and an epilogue; here a wave
// streams 16-row groups straight into MFMA operand registers (rows = 16 consecutive pixels of one image row), the normalisation and
// GELU are applied there, the 16 x 16 result is already (co, ky, kx) x pixel, and it leaves as float2 rows of the NCHW prediction --
// no patch-major fp32 intermediate and no second pass.  The MFMA sequence over k is the GEMM path's (32 at a time, ascending).
constexpr int DL_GPW = 16;      // 16-row groups per wave (contiguous: one frame, so the loss partials stay in registers)
template <int KS>
__global__ void __launch_bounds__(NT) debed_last_kernel(const bf16* __restrict__ act, const float* __restrict__ sc, const float* __restrict__ sh,
                                                       const bf16* __restrict__ wc, float* __restrict__ pred, const float* __restrict__ y,
                                                       float* __restrict__ lossbuf, int Co, int h, int w) {
    constexpr int Ci = 32 * KS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int f = blockIdx.y;
    const int GF = h * w / 16;                                      // groups per frame
    const int g0 = (blockIdx.x * (NT / 64) + wave) * DL_GPW;
    if (g0 >= GF) return;
    const int g1 = min(g0 + DL_GPW, GF);
    const int H = 2 * h, W = 2 * w;
    // weights as the first MFMA operand: lane holds wc[k = 32 s + 8 lg + j][n = li]
    bf16x8 wf[KS];
    float a_sc[KS][8], a_sh[KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * s + 8 * lg + j;
            wf[s][j] = wc[k * 16 + li];
            a_sc[s][j] = sc[(long)f * Ci + k];
            a_sh[s][j] = sh[(long)f * Ci + k];
        }
    }
    const bf16* rows = act + ((long)f * h * w + li) * Ci + 8 * lg;
    bf16x8 cur[KS], nxt[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) cur[s] = *reinterpret_cast<const bf16x8*>(rows + (long)g0 * 16 * Ci + 32 * s);
    float num = 0.f, den = 0.f;
    const bool live = lg < Co;
    for (int g = g0; g < g1; ++g) {
        if (g + 1 < g1) {
#pragma unroll
            for (int s = 0; s < KS; ++s) nxt[s] = *reinterpret_cast<const bf16x8*>(rows + (long)(g + 1) * 16 * Ci + 32 * s);
        }
        const int px = g * 16 + li, yo = px / w, xo = px - yo * w;
        const long o = (((long)f * Co + lg) * H + 2 * yo) * W + 2 * xo;
        float2 y0 = make_float2(0.f, 0.f), y1 = y0;
        if (y && live) { y0 = *reinterpret_cast<const float2*>(y + o); y1 = *reinterpret_cast<const float2*>(y + o + W); }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (bf16)gelu_fast((float)cur[s][j] * a_sc[s][j] + a_sh[s][j]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], a, acc, 0, 0, 0);      // acc[j] = out[n = 4 lg + j][pixel li]
        }
        if (live) {
            *reinterpret_cast<float2*>(pred + o) = make_float2(acc[0], acc[1]);
            *reinterpret_cast<float2*>(pred + o + W) = make_float2(acc[2], acc[3]);
            if (y) {
                num += (acc[0] - y0.x) * (acc[0] - y0.x) + (acc[1] - y0.y) * (acc[1] - y0.y) + (acc[2] - y1.x) * (acc[2] - y1.x) + (acc[3] - y1.y) * (acc[3] - y1.y);
                den += y0.x * y0.x + y0.y * y0.y + y1.x * y1.x + y1.y * y1.y;
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
    }
    if (y) {        // the 16 lanes of a k-group share (f, co): one pair of atomics per wave and output channel
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) { num += __shfl_xor(num, m, 64); den += __shfl_xor(den, m, 64); }
        if (li == 0 && live) {
            loss_accum(lossbuf, ((long)f * Co + lg) * 2, num);
            loss_accum(lossbuf, ((long)f * Co + lg) * 2 + 1, den);
        }
    }
}
// The same stage backwards: dpm[p][co*4 + ky*2 + kx] = d pred[f][co][2y+ky][2x+kx] (given, or coef[f][co] * gscale * (pred - y) of the fused
// loss), written for the weight-gradient GEMM, and dact[p][ci] = sum_n dpm[p][n] * wc[ci][n] in the same pass (K = 16: one
// v_mfma_f32_16x16x16_bf16 per 16 input channels).  The weight rows are dealt to the MFMA tiles so that a lane ends up with runs of 8
// consecutive input channels of its pixel and the four lanes of a pixel with 32 (tile pair q, lane group lg  <->  ci = 32 q + 8 lg + ..):
// each store instruction writes 64 contiguous bytes of 16 rows, a wave 16 complete rows (BF_DL_PERM=0: 4T consecutive channels per
// lane, 16-byte pieces 8T bytes apart -- 0.13 % slower end to end).  Replaces bf_nchw2pm + a 128-wide GEMM tile with K = 16.
template <int T>
__global__ void __launch_bounds__(NT) debed_last_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ pred, const float* __restrict__ y,
                                                           const float* __restrict__ coef, const float* __restrict__ gscale,
                                                           const bf16* __restrict__ wc, bf16* __restrict__ dpm, bf16* __restrict__ dact,
                                                           int Co, int h, int w, float* __restrict__ part, int perm) {
    constexpr int Ci = 16 * T;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int f = blockIdx.y;
    const int GF = h * w / 16;
    const int g0 = (blockIdx.x * (NT / 64) + wave) * DL_GPW;
    if (g0 >= GF) return;
    const int g1 = min(g0 + DL_GPW, GF);
    const int H = 2 * h, W = 2 * w;
    const bool live = lg < Co;
    s16x4 wf[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        // lane (pixel li, group lg) ends up with output channel chan(k) in o[k], k = 4 t + j:
        //   perm 0: chan = 4T lg + k            (4T consecutive channels per lane: a store instruction writes 16-byte pieces 8T bytes apart)
        //   perm 1: chan = 32 (k / 8) + 8 lg + k % 8   (store instruction q writes the 64 contiguous bytes [64 q, 64 q + 64) of each row)
        const int ci = perm ? 32 * (t >> 1) + 8 * (li >> 2) + 4 * (t & 1) + (li & 3) : 4 * T * (li >> 2) + 4 * t + (li & 3);
        wf[t] = *reinterpret_cast<const s16x4*>(wc + ci * 16 + 4 * lg);
    }
    const float cf = (!dpred && live) ? coef[(long)f * Co + lg] * (gscale ? gscale[0] : 1.f) : 0.f;
    auto fetch = [&](int g, float2 (&r)[4]) {
        const int px = g * 16 + li, yo = px / w, xo = px - yo * w;
        const long o = (((long)f * Co + lg) * H + 2 * yo) * W + 2 * xo;
        if (!live) { r[0] = r[1] = r[2] = r[3] = make_float2(0.f, 0.f); return; }
        if (dpred) { r[0] = *reinterpret_cast<const float2*>(dpred + o); r[1] = *reinterpret_cast<const float2*>(dpred + o + W); r[2] = r[3] = make_float2(0.f, 0.f); }
        else {
            r[0] = *reinterpret_cast<const float2*>(pred + o); r[1] = *reinterpret_cast<const float2*>(pred + o + W);
            r[2] = *reinterpret_cast<const float2*>(y + o); r[3] = *reinterpret_cast<const float2*>(y + o + W);
        }
    };
    float2 cur[4], nxt[4];
    fetch(g0, cur);
    float s1[4 * T], s2[4 * T];       // optional InstanceNorm statistics of the rows written (this wave's 16 DL_GPW rows = one slice)
#pragma unroll
    for (int k = 0; k < 4 * T; ++k) s1[k] = s2[k] = 0.f;
    for (int g = g0; g < g1; ++g) {
        if (g + 1 < g1) fetch(g + 1, nxt);
        float v[4];
        if (dpred) { v[0] = cur[0].x; v[1] = cur[0].y; v[2] = cur[1].x; v[3] = cur[1].y; }
        else { v[0] = cf * (cur[0].x - cur[2].x); v[1] = cf * (cur[0].y - cur[2].y); v[2] = cf * (cur[1].x - cur[3].x); v[3] = cf * (cur[1].y - cur[3].y); }
        const bf16x4v b = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        const long p = (long)f * h * w + g * 16 + li;
        *reinterpret_cast<bf16x4v*>(dpm + p * 16 + 4 * lg) = b;
        const s16x4 bs = __builtin_bit_cast(s16x4, b);
        bf16 o[4 * T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wf[t], bs, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[4 * t + j] = (bf16)acc[j];
        }
        if (part) {
#pragma unroll
            for (int k = 0; k < 4 * T; ++k) { const float v2 = (float)o[k]; s1[k] += v2; s2[k] += v2 * v2; }      // of the values as stored
        }
        if (dact) {                                     // (bf_embed_first without a map: only the patch rows and the statistics leave)
            bf16* dst = dact + p * Ci + (perm ? 8 * lg : 4 * T * lg);
            const int qs = perm ? 32 : 8;
#pragma unroll
            for (int q = 0; q < 4 * T / 8; ++q) {
                bf16x8 o8;
#pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] = o[8 * q + j];
                *reinterpret_cast<bf16x8*>(dst + qs * q) = o8;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
    }
    if (part) {      // {slice mean, centred second moment} in in_stats_slice_kernel's layout: in_stats_merge_kernel finishes the frame
        const float n = (float)((g1 - g0) * 16);
        const int nsl = (GF + DL_GPW - 1) / DL_GPW, sl = g0 / DL_GPW;
        float2* o2 = reinterpret_cast<float2*>(part) + ((long)f * nsl + sl) * Ci;
#pragma unroll
        for (int k = 0; k < 4 * T; ++k) {
            const int chan = perm ? 32 * (k >> 3) + 8 * lg + (k & 7) : 4 * T * lg + k;
            float a = s1[k], b = s2[k];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
            const float mu = a / n;
            if (li == 0) o2[chan] = make_float2(mu, fmaxf(b - a * mu, 0.f));
        }
    }
}
// The last debed stage backwards TOGETHER with the InstanceNorm + GELU in front of it (layers/patching.py:92-104 under autograd).  The
// gradient of the 96-channel map at full patch-grid resolution (226 MB at the bench shape) has rank 16 -- dact = dpm @ wc^T -- so neither
// pass stores it: PASS 1 forms dpm (written: the weight gradient wants it), multiplies dact by gelu'(z) in registers and leaves the
// InstanceNorm backward's two sums per 256-row slice; PASS 2 rebuilds dact from the 16-wide rows and writes
// dx = rstd w (dd - s1/S - xh s2/S) directly.  Against debed_last_bwd_kernel + the two-phase sliced InstanceNorm backward: one write
// (226 MB) and two reads (452 MB) of the gradient map less.  Lane = (pixel li, channel runs 32 q + 8 lg .. +7): 16-byte loads of the
// activation map, 16-byte stores of dx.
struct DlInb {
    const bf16* ymap;                   // raw output of the stage in front [P][Ci]
    const float *mean, *rstd, *w, *b;   // its InstanceNorm: [frames][Ci], [Ci]
    const float* tot;                   // PASS 2: {s1, s2} per (frame, channel)
    float inv_s;                        // 1 / rows per frame
};
template <int T, int PASS>
__global__ void __launch_bounds__(NT) debed_last_inbwd_kernel(const float* __restrict__ dpred, const float* __restrict__ pred, const float* __restrict__ y,
                                                             const float* __restrict__ coef, const float* __restrict__ gscale,
                                                             const bf16* __restrict__ wc, bf16* __restrict__ dpm, bf16* __restrict__ dx,
                                                             int Co, int h, int w, float* __restrict__ part, DlInb nb) {
    constexpr int Ci = 16 * T;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
    __shared__ float4 k1[Ci], k2[Ci];       // {sc, sh, rstd, -mean rstd} and {rstd w, s1 / S, s2 / S, -}
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int f = blockIdx.y;
    for (int c = threadIdx.x; c < Ci; c += NT) {
        const long o = (long)f * Ci + c;
        const float r = nb.rstd[o], m = nb.mean[o], a = r * nb.w[c];
        k1[c] = make_float4(a, fmaf(-m, a, nb.b[c]), r, -m * r);
        if (PASS == 2) k2[c] = make_float4(a, nb.tot[2 * o] * nb.inv_s, nb.tot[2 * o + 1] * nb.inv_s, 0.f);
    }
    __syncthreads();
    const int GF = h * w / 16;
    const int g0 = (blockIdx.x * (NT / 64) + wave) * DL_GPW;
    if (g0 >= GF) return;
    const int g1 = min(g0 + DL_GPW, GF);
    const int H = 2 * h, W = 2 * w;
    const bool live = lg < Co;
    s16x4 wf[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ci = 32 * (t >> 1) + 8 * (li >> 2) + 4 * (t & 1) + (li & 3);      // lane (pixel li, group lg) gets channel 32 (k / 8) + 8 lg + k % 8 in slot k = 4 t + j
        wf[t] = *reinterpret_cast<const s16x4*>(wc + ci * 16 + 4 * lg);
    }
    const float cf = (PASS == 1 && !dpred && live) ? coef[(long)f * Co + lg] * (gscale ? gscale[0] : 1.f) : 0.f;
    struct Row { float2 r[4]; bf16x8 yv[T / 2]; bf16x4v d; };
    auto fetch = [&](int g, Row& o) __attribute__((always_inline)) {
        const long p = (long)f * h * w + g * 16 + li;
#pragma unroll
        for (int q = 0; q < T / 2; ++q) o.yv[q] = *reinterpret_cast<const bf16x8*>(nb.ymap + p * Ci + 8 * lg + 32 * q);
        if (PASS == 2) { o.d = *reinterpret_cast<const bf16x4v*>(dpm + p * 16 + 4 * lg); return; }
        const int px = g * 16 + li, yo = px / w, xo = px - yo * w;
        const long oo = (((long)f * Co + lg) * H + 2 * yo) * W + 2 * xo;
        if (!live) { o.r[0] = o.r[1] = o.r[2] = o.r[3] = make_float2(0.f, 0.f); return; }
        if (dpred) { o.r[0] = *reinterpret_cast<const float2*>(dpred + oo); o.r[1] = *reinterpret_cast<const float2*>(dpred + oo + W); o.r[2] = o.r[3] = make_float2(0.f, 0.f); }
        else {
            o.r[0] = *reinterpret_cast<const float2*>(pred + oo); o.r[1] = *reinterpret_cast<const float2*>(pred + oo + W);
            o.r[2] = *reinterpret_cast<const float2*>(y + oo); o.r[3] = *reinterpret_cast<const float2*>(y + oo + W);
        }
    };
    Row cur, nxt, nx2;                  // two 16-pixel groups ahead: a wave has few registers left for anything but bytes in flight
    fetch(g0, cur);
    if (g0 + 1 < g1) fetch(g0 + 1, nxt);
    float s1[PASS == 1 ? 4 * T : 1], s2[PASS == 1 ? 4 * T : 1];
    if (PASS == 1) {
#pragma unroll
        for (int k = 0; k < 4 * T; ++k) s1[k] = s2[k] = 0.f;
    }
    for (int g = g0; g < g1; ++g) {
        if (g + 2 < g1) fetch(g + 2, nx2);
        const long p = (long)f * h * w + g * 16 + li;
        bf16x4v b;
        if (PASS == 1) {
            float v[4];
            if (dpred) { v[0] = cur.r[0].x; v[1] = cur.r[0].y; v[2] = cur.r[1].x; v[3] = cur.r[1].y; }
            else { v[0] = cf * (cur.r[0].x - cur.r[2].x); v[1] = cf * (cur.r[0].y - cur.r[2].y); v[2] = cf * (cur.r[1].x - cur.r[3].x); v[3] = cf * (cur.r[1].y - cur.r[3].y); }
            b = bf16x4v{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *reinterpret_cast<bf16x4v*>(dpm + p * 16 + 4 * lg) = b;
        } else b = cur.d;
        const s16x4 bs = __builtin_bit_cast(s16x4, b);
        bf16x8 o8[T / 2];
        // the per-channel constants are loop invariant: left alone the compiler keeps all 4 * 4T (+ 3 * 4T) of them in registers, which
        // halves the waves a SIMD can hold; an offset it cannot see through makes them 16-byte LDS reads per group instead
        int kofs = 8 * lg;
        asm volatile("" : "+v"(kofs));
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wf[t], bs, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * t + j;
                const float4 c1 = k1[32 * (k >> 3) + kofs + (k & 7)];
                const float yv = (float)cur.yv[k >> 3][k & 7];
                const float dd = acc[j] * dgelu_fast(fmaf(yv, c1.x, c1.y)), xh = fmaf(yv, c1.z, c1.w);
                if (PASS == 1) { s1[k] += dd; s2[k] = fmaf(dd, xh, s2[k]); }
                else {
                    const float4 c2 = k2[32 * (k >> 3) + kofs + (k & 7)];
                    o8[k >> 3][k & 7] = (bf16)(c2.x * (dd - c2.y - xh * c2.z));
                }
            }
        }
        if (PASS == 2) {
#pragma unroll
            for (int q = 0; q < T / 2; ++q) *reinterpret_cast<bf16x8*>(dx + p * Ci + 8 * lg + 32 * q) = o8[q];
        }
        cur = nxt;
        nxt = nx2;
    }
    if (PASS == 1) {      // {sum dd, sum dd xh} of this wave's slice: part[(f * nsl + sl) * Ci + c]
        const int nsl = (GF + DL_GPW - 1) / DL_GPW, sl = g0 / DL_GPW;
        float2* o2 = reinterpret_cast<float2*>(part) + ((long)f * nsl + sl) * Ci;
#pragma unroll
        for (int k = 0; k < 4 * T; ++k) {
            float a = s1[k], b2 = s2[k];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b2 += __shfl_xor(b2, m, 64); }
            if (li == 0) o2[32 * (k >> 3) + 8 * lg + (k & 7)] = make_float2(a, b2);
        }
    }
}
// tot[f][c] = sum over slices, in slice order.  grid (ceil(C / 64), frames), 256 threads = 64 channels x 4 slice lanes
__global__ void __launch_bounds__(NT) dl_slice_sum_kernel(const float* __restrict__ part, int C, int nsl, float* __restrict__ tot) {
    __shared__ float red[2][4][64];
    const int l = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 64 + l, f = blockIdx.y;
    const bool cv = c < C;
    const float2* pp = reinterpret_cast<const float2*>(part) + (long)f * nsl * C + (cv ? c : 0);
    float a = 0.f, b = 0.f;
    if (cv)
        for (int sl = q; sl < nsl; sl += 4) { const float2 v = pp[(long)sl * C]; a += v.x; b += v.y; }
    red[0][q][l] = a; red[1][q][l] = b;
    __syncthreads();
    if (q != 0 || !cv) return;
    reinterpret_cast<float2*>(tot)[(long)f * C + c] = make_float2((red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]),
                                                                 (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]));
}
__global__ void __launch_bounds__(64 * BF_RED_FL) dl_param_reduce_kernel(InReduceJob j) {
    __shared__ float red[5][BF_RED_FL][64];
    in_reduce_block(j, blockIdx.x, blockIdx.y, red);
}
// loss = sum_c mean_f sqrt(num/den);  coef[f][c] = 1 / (F * sqrt(num) * sqrt(den))   (single block)
__global__ void lploss_finalize_kernel(const float* __restrict__ lossbuf, int F, int Co, float* __restrict__ loss, float* __restrict__ coef) {
    __shared__ float red[NT];
    float acc = 0.f;
    for (int i = threadIdx.x; i < F * Co; i += NT) {
        const float num = loss_read(lossbuf, 2 * i), den = loss_read(lossbuf, 2 * i + 1);
        acc += sqrtf(num) / sqrtf(den);
        if (coef) coef[i] = 1.0f / ((float)F * sqrtf(num) * sqrtf(den));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)F;
}
// dpm[p][n] = dpred[f][co][2y+ky][2x+kx]   with dpred either given, or coef[f][co] * gscale * (pred - y)
template <typename T>
__global__ void __launch_bounds__(NT) nchw2pm_kernel(const float* __restrict__ dpred, const float* __restrict__ pred, const float* __restrict__ y,
                                                    const float* __restrict__ coef, const float* __restrict__ gscale, T* __restrict__ dpm,
                                                    int Co, int h, int w, int Np, long P) {
    const int H = 2 * h, W = 2 * w;
    const int nq = Np / 4;   // groups of 4 columns (one co each; groups >= Co are zero padding)
    const long total = P * nq;
    const float gs = gscale ? gscale[0] : 1.f;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int co = (int)(i % nq);
        const long p = i / nq;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (co < Co) {
            const int xo = (int)(p % w);
            const long t = p / w;
            const int yo = (int)(t % h);
            const long f = t / h;
            const long o = ((f * Co + co) * H + 2 * yo) * (long)W + 2 * xo;
            if (dpred) {
                const float2 a = *reinterpret_cast<const float2*>(dpred + o), b = *reinterpret_cast<const float2*>(dpred + o + W);
                v0 = a.x; v1 = a.y; v2 = b.x; v3 = b.y;
            } else {
                const float cf = coef[f * Co + co] * gs;
                const float2 a = *reinterpret_cast<const float2*>(pred + o), b = *reinterpret_cast<const float2*>(pred + o + W);
                const float2 c = *reinterpret_cast<const float2*>(y + o), d = *reinterpret_cast<const float2*>(y + o + W);
                v0 = cf * (a.x - c.x); v1 = cf * (a.y - c.y); v2 = cf * (b.x - d.x); v3 = cf * (b.y - d.y);
            }
        }
        T* dst = dpm + p * Np + co * 4;
        dst[0] = from_f<T>(v0); dst[1] = from_f<T>(v1); dst[2] = from_f<T>(v2); dst[3] = from_f<T>(v3);
    }
}

// ---------------------------------------------------------------------------- weight preparation
// modes: 0 cast/pad rows: [R][K] -> [R][Kp]
//        1 conv k2s2  [Co][Ci][2][2] -> [Co][(ky,kx,ci)]
//        2 convT k2s2 [Ci][Co][2][2] -> [(ky,kx,co)][Ci]
template <typename T>
__global__ void __launch_bounds__(NT) wprep_kernel(const float* __restrict__ src, T* __restrict__ dst, int mode, int R, int K, int Kp) {
    const long total = (long)R * Kp;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int r = (int)(i / Kp), k = (int)(i % Kp);
        float v = 0.f;
        if (mode == 0) { if (k < K) v = src[(long)r * K + k]; }
        else if (mode == 1) {          // R = Co, K = Kp = 4*Ci ; k = (ky*2+kx)*Ci + ci
            const int Ci = K / 4, q = k / Ci, ci = k % Ci;
            v = src[((long)r * Ci + ci) * 4 + q];
        } else {                       // mode 2: R = 4*Co rows r = q*Co + co ; K = Ci
            const int Co = R / 4, q = r / Co, co = r % Co;
            v = src[((long)k * Co + co) * 4 + q];
        }
        dst[i] = from_f<T>(v);
    }
}
// all stages of an embed / debed call in ONE launch (blockIdx.y = stage)
struct WprepJobs { const float* src[BF_MAX_STAGES]; void* dst[BF_MAX_STAGES]; int mode[BF_MAX_STAGES], R[BF_MAX_STAGES], K[BF_MAX_STAGES], Kp[BF_MAX_STAGES]; };
template <typename T>
__global__ void __launch_bounds__(NT) wprep_multi_kernel(WprepJobs j) {
    const int s = blockIdx.y;
    const float* __restrict__ src = j.src[s];
    T* __restrict__ dst = reinterpret_cast<T*>(j.dst[s]);
    const int mode = j.mode[s], R = j.R[s], K = j.K[s], Kp = j.Kp[s];
    const long total = (long)R * Kp;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int r = (int)(i / Kp), k = (int)(i % Kp);
        float v = 0.f;
        if (mode == 0) { if (k < K) v = src[(long)r * K + k]; }
        else if (mode == 1) { const int Ci = K / 4, q = k / Ci, ci = k % Ci; v = src[((long)r * Ci + ci) * 4 + q]; }
        else { const int Co = R / 4, q = r / Co, co = r % Co; v = src[((long)k * Co + co) * 4 + q]; }
        dst[i] = from_f<T>(v);
    }
}
// gradient of the prepared operand (fp32, prepared layout) accumulated into the state_dict layout
__global__ void __launch_bounds__(NT) wgrad_unprep_kernel(const float* __restrict__ gsrc, float* __restrict__ gdst, int mode, int R, int K, int Kp,
                                                         int transposed) {
    const long total = (long)R * K;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int r = (int)(i / K), k = (int)(i % K);
        // transposed: gsrc is [Kp_rows = k][R] (the GEMM produced the transpose)
        const float v = transposed ? gsrc[(long)k * R + r] : gsrc[(long)r * Kp + k];
        long o;
        if (mode == 0) o = (long)r * K + k;
        else if (mode == 1) { const int Ci = K / 4, q = k / Ci, ci = k % Ci; o = ((long)r * Ci + ci) * 4 + q; }
        else { const int Co = R / 4, q = r / Co, co = r % Co; o = ((long)k * Co + co) * 4 + q; }
        gdst[o] += v;
    }
}

// ---------------------------------------------------------------------------- FiLM net (B rows; one block)
// gb[b][0:E] = gamma, gb[b][E:2E] = beta ;  c = LN(cond) ; gb = c @ W^T + bias
__global__ void film_net_fwd_kernel(const float* __restrict__ cond, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                    const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ gb,
                                    float* __restrict__ chat, float* __restrict__ crstd, int B, int P, int E2) {
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        float mu = 0.f;
        for (int i = 0; i < P; ++i) mu += cond[b * P + i];
        mu /= (float)P;
        float var = 0.f;
        for (int i = 0; i < P; ++i) { const float t = cond[b * P + i] - mu; var += t * t; }
        const float r = rsqrtf(var / (float)P + BF_IN_EPS);
        if (threadIdx.x == 0) crstd[b] = r;
        for (int i = threadIdx.x; i < P; i += blockDim.x) chat[b * P + i] = (cond[b * P + i] - mu) * r;
        for (int o = threadIdx.x; o < E2; o += blockDim.x) {
            float acc = bias[o];
            for (int i = 0; i < P; ++i) acc += ((cond[b * P + i] - mu) * r * lnw[i] + lnb[i]) * W[(long)o * P + i];
            gb[(long)(o / (E2 / 2)) * B * (E2 / 2) + (long)b * (E2 / 2) + (o % (E2 / 2))] = acc;   // [2][B][E]
        }
    }
}
// dgb: [B][2E] (dgamma | dbeta).  dW[o][i] += sum_b dgb[b][o] c[b][i]; dbias[o] += sum_b dgb;
// dc[b][i] = sum_o dgb[b][o] W[o][i]; dlnw[i] += sum_b dc*chat; dlnb[i] += sum_b dc   (d cond is not needed)
__global__ void film_net_bwd_kernel(const float* __restrict__ dgb_, const float* __restrict__ chat, const float* __restrict__ lnw,
                                    const float* __restrict__ lnb, const float* __restrict__ W, float* __restrict__ dW,
                                    float* __restrict__ dbias, float* __restrict__ dlnw, float* __restrict__ dlnb, int B, int P, int E2) {
    const int E = E2 / 2;
    // fp64 accumulation: the LayerNorm(P) gradients are sums of 2E signed terms that largely cancel, and the kernel is tiny
    const int nmain = ((E2 + 63) / 64);
    if ((int)blockIdx.x >= nmain) {
        // P more workgroups (64 threads), one per LayerNorm input i: dlnw[i] / dlnb[i], each a sum over ALL outputs o and samples b -- lane l
        // takes o = l, l + 64, ... in order and the 64 partial sums meet in a fixed butterfly: one writer per value, no float atomics (they
        // used to arrive from every wave in arrival order: the two gradients differed from run to run)
        {
            const int i = (int)blockIdx.x - nmain;
            double sw = 0.0, sbias = 0.0;
            for (int o = threadIdx.x; o < E2; o += 64) {
                const double w = W[(long)o * P + i];
                for (int b = 0; b < B; ++b) {
                    const double dg = (double)dgb_[(long)(o / E) * B * E + (long)b * E + (o % E)];
                    sw += dg * w * chat[b * P + i];
                    sbias += dg * w;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { sw += __shfl_xor(sw, off, 64); sbias += __shfl_xor(sbias, off, 64); }
            if (threadIdx.x == 0) { dlnw[i] += (float)sw; dlnb[i] += (float)sbias; }
        }
        return;
    }
    // one thread per output row o of the Linear: dW[o][:] and dbias[o]
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= E2) return;
    double sb = 0.0;
    for (int b = 0; b < B; ++b) sb += (double)dgb_[(long)(o / E) * B * E + (long)b * E + (o % E)];
    dbias[o] += (float)sb;
    for (int i = 0; i < P; ++i) {
        double acc = 0.0;
        for (int b = 0; b < B; ++b)
            acc += (double)dgb_[(long)(o / E) * B * E + (long)b * E + (o % E)] * ((double)chat[b * P + i] * lnw[i] + lnb[i]);
        dW[(long)o * P + i] += (float)acc;
    }
}

// ---------------------------------------------------------------------------- AdamW (torch.optim.AdamW semantics)
__global__ void __launch_bounds__(NT) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                                                  float bc1, float sqrt_bc2, float gscale) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
#define BF_ADAM1(X)                                                        \
        { const float gr = gg.X * gscale;                                   \
          pp.X *= (1.f - lr * wd);                                          \
          mm.X = b1 * mm.X + (1.f - b1) * gr;                               \
          vv.X = b2 * vv.X + (1.f - b2) * gr * gr;                          \
          pp.X -= (lr / bc1) * mm.X / (sqrtf(vv.X) / sqrt_bc2 + eps); }
        BF_ADAM1(x) BF_ADAM1(y) BF_ADAM1(z) BF_ADAM1(w)
#undef BF_ADAM1
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
        const long i = n4 * 4 + threadIdx.x;
        const float gr = g[i] * gscale;
        float pp = p[i] * (1.f - lr * wd);
        const float mm = b1 * m[i] + (1.f - b1) * gr;
        const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        pp -= (lr / bc1) * mm / (sqrtf(vv) / sqrt_bc2 + eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

int grid_for(long total) { return (int)std::max<long>(1, std::min<long>((total + NT - 1) / NT, 256L * 16)); }
}  // namespace

extern "C" int bf_im2col_nchw(int dtype, const float* x, void* out, int frames, int C, int H, int W, int Kp, bf_stream_t stream) {
    BF_REQUIRE(x && out && frames > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Kp >= 4 * C, "bf_im2col_nchw: bad arguments");
    const long P = (long)frames * (H / 2) * (W / 2);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BF_DTYPE_BF16) {
        hipLaunchKernelGGL(im2col_kernel<bf16>, dim3(grid_for(P * C)), dim3(NT), 0, st, x, (bf16*)out, C, H, W, Kp, P);
        if (Kp > 4 * C) hipLaunchKernelGGL(zero_pad_cols_kernel<bf16>, dim3(grid_for(P * (Kp - 4 * C))), dim3(NT), 0, st, (bf16*)out, 4 * C, Kp, P);
    } else {
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(P * C)), dim3(NT), 0, st, x, (float*)out, C, H, W, Kp, P);
        if (Kp > 4 * C) hipLaunchKernelGGL(zero_pad_cols_kernel<float>, dim3(grid_for(P * (Kp - 4 * C))), dim3(NT), 0, st, (float*)out, 4 * C, Kp, P);
    }
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_col2im_nchw(int dtype, const void* g, float* dx, int frames, int C, int H, int W, int Kp, bf_stream_t stream) {
    BF_REQUIRE(g && dx && frames > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && Kp >= 4 * C, "bf_col2im_nchw: bad arguments");
    const long P = (long)frames * (H / 2) * (W / 2);
    if (dtype == BF_DTYPE_BF16) hipLaunchKernelGGL(col2im_kernel<bf16>, dim3(grid_for(P * C)), dim3(NT), 0, (hipStream_t)stream, (const bf16*)g, dx, C, H, W, Kp, P);
    else hipLaunchKernelGGL(col2im_kernel<float>, dim3(grid_for(P * C)), dim3(NT), 0, (hipStream_t)stream, (const float*)g, dx, C, H, W, Kp, P);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_pm2nchw(const float* pm, float* pred, const float* y, float* lossbuf, int frames, int Co, int h, int w, int Np,
                          bf_stream_t stream) {
    BF_REQUIRE(pm && pred && frames > 0 && Co > 0 && h > 0 && w > 0 && Np >= 4 * Co && Np % 4 == 0, "bf_pm2nchw: bad arguments");
    BF_REQUIRE(!y || lossbuf, "bf_pm2nchw: loss buffer missing");
    dim3 grid(std::max(1, std::min(bf_cdiv((long)h * w, NT), 64)), frames);
    hipLaunchKernelGGL(pm2nchw_kernel, grid, dim3(NT), 0, (hipStream_t)stream, pm, pred, y, lossbuf, Co, h, w, Np);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_debed_last(int dtype, const void* act, const float* sc, const float* sh, const void* wc, float* pred, const float* y,
                             float* lossbuf, int frames, int Ci, int Co, int h, int w, int Np, bf_stream_t stream) {
    BF_REQUIRE(act && sc && sh && wc && pred && frames > 0 && Ci > 0 && Co > 0 && h > 0 && w > 0, "bf_debed_last: bad arguments");
    BF_REQUIRE(!y || lossbuf, "bf_debed_last: loss buffer missing");
    // shapes the streaming kernel does not take (the caller falls back to GEMM + bf_pm2nchw): not an error
    if (dtype != BF_DTYPE_BF16 || Np != 16 || Co > 4 || Ci % 32 != 0 || Ci > 128 || w % 16 != 0) return 1;
    static const bool off = bf_knob("BF_DEBED_LAST", 1) == 0;
    if (off) return 1;
    const int GF = h * w / 16;
    dim3 grid(bf_cdiv(GF, (NT / 64) * DL_GPW), frames);
    hipStream_t st = (hipStream_t)stream;
    const double P = (double)frames * h * w;
    BfProfScope prof(st, "debed_last", 2.0 * P * Ci * 16, P * (2.0 * Ci + 16.0 * Co * (y ? 2 : 1)));
#define DL(KS) hipLaunchKernelGGL(debed_last_kernel<KS>, grid, dim3(NT), 0, st, (const bf16*)act, sc, sh, (const bf16*)wc, pred, y, lossbuf, Co, h, w)
    switch (Ci / 32) { case 1: DL(1); break; case 2: DL(2); break; case 3: DL(3); break; default: DL(4); break; }
#undef DL
    BF_CHECK_LAUNCH();
    return 0;
}

static int debed_last_bwd_launch(int dtype, const float* dpred, const float* pred, const float* y, const float* coef, const float* gscale,
                                 const void* wc, void* dpm, void* dact, int frames, int Ci, int Co, int h, int w, int Np, float* part,
                                 bf_stream_t stream) {
    BF_REQUIRE(wc && dpm && (dact || part) && (dpred || (pred && y && coef)) && frames > 0 && Ci > 0 && Co > 0 && h > 0 && w > 0, "bf_debed_last_bwd: bad arguments");
    if (dtype != BF_DTYPE_BF16 || Np != 16 || Co > 4 || Ci % 32 != 0 || Ci > 128 || w % 16 != 0) return 1;      // caller keeps bf_nchw2pm + GEMM
    static const bool off = bf_knob("BF_DEBED_LAST_BWD", 1) == 0;
    if (off) return 1;
    const int GF = h * w / 16;
    dim3 grid(bf_cdiv(GF, (NT / 64) * DL_GPW), frames);
    hipStream_t st = (hipStream_t)stream;
    static const int perm = bf_knob("BF_DL_PERM", 1);
    const double P = (double)frames * h * w;
    BfProfScope prof(st, "patch16", 2.0 * P * Ci * 16, P * (2.0 * Ci + 32.0 + 16.0 * Co * (dpred ? 1 : 2)));
#define DLB(T) hipLaunchKernelGGL(debed_last_bwd_kernel<T>, grid, dim3(NT), 0, st, dpred, pred, y, coef, gscale, (const bf16*)wc, (bf16*)dpm, (bf16*)dact, Co, h, w, part, perm)
    switch (Ci / 32) { case 1: DLB(2); break; case 2: DLB(4); break; case 3: DLB(6); break; default: DLB(8); break; }
#undef DLB
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_debed_last_bwd(int dtype, const float* dpred, const float* pred, const float* y, const float* coef, const float* gscale,
                                 const void* wc, void* dpm, void* dact, int frames, int Ci, int Co, int h, int w, int Np, bf_stream_t stream) {
    return debed_last_bwd_launch(dtype, dpred, pred, y, coef, gscale, wc, dpm, dact, frames, Ci, Co, h, w, Np, nullptr, stream);
}

// ... with the InstanceNorm + GELU in front of the stage folded in (debed_last_inbwd_kernel): dpm as above, dx [P][Ci] = the gradient of the
// RAW map ymap in front of that InstanceNorm, d_in_w / d_in_b accumulated.  ws: bf_in_ws_floats(dtype, frames, h*w, Ci) floats.
// Returns 1 (nothing launched) for the shapes bf_debed_last_bwd declines, Ci % 32, or a workspace that does not hold 256-row slices.
extern "C" int bf_debed_last_bwd_norm(int dtype, const float* dpred, const float* pred, const float* y, const float* coef, const float* gscale,
                                      const void* wc, void* dpm, const void* ymap, const float* mean, const float* rstd, const float* in_w,
                                      const float* in_b, void* dx, float* d_in_w, float* d_in_b, int frames, int Ci, int Co, int h, int w, int Np,
                                      float* ws, int64_t ws_floats, bf_stream_t stream) {
    if (dtype != BF_DTYPE_BF16 || Np != 16 || Co > 4 || Ci % 32 != 0 || Ci > 128 || w % 16 != 0)
        return bf_decline("bf_debed_last_bwd_norm covers bf16, Np = 16, Co <= 4, Ci a multiple of 32 up to 128, w a multiple of 16");
    static const bool off = bf_knob("BF_DEBED_LAST_NORM", 1) == 0;
    if (off) return bf_decline("bf_debed_last_bwd_norm: switched off (BF_DEBED_LAST_NORM=0)");
    BF_REQUIRE(wc && dpm && ymap && mean && rstd && in_w && in_b && dx && ws && (dpred || (pred && y && coef)) && frames > 0 && Co > 0 && h > 0 && w > 0,
               "bf_debed_last_bwd_norm: bad arguments");
    const int GF = h * w / 16, nsl = bf_cdiv(GF, DL_GPW);
    if (ws_floats < (int64_t)2 * frames * Ci * (1 + nsl))
        return bf_decline("bf_debed_last_bwd_norm: workspace smaller than 2 * frames * Ci * (1 + slices) floats (bf_in_ws_floats)");
    float* tot = ws;
    float* part = ws + (size_t)2 * frames * Ci;
    dim3 grid(bf_cdiv(GF, (NT / 64) * DL_GPW), frames);
    hipStream_t st = (hipStream_t)stream;
    const double P = (double)frames * h * w;
    const DlInb nb1{(const bf16*)ymap, mean, rstd, in_w, in_b, nullptr, 0.f};
    const DlInb nb2{(const bf16*)ymap, mean, rstd, in_w, in_b, tot, 1.0f / (float)(h * w)};
    {
        BfProfScope prof(st, "debed_last_bwd<stats>", 2.0 * P * Ci * 16, P * (2.0 * Ci + 32.0 + 16.0 * Co * (dpred ? 1 : 2)));
#define DLN(T) hipLaunchKernelGGL((debed_last_inbwd_kernel<T, 1>), grid, dim3(NT), 0, st, dpred, pred, y, coef, gscale, (const bf16*)wc, (bf16*)dpm, (bf16*)dx, Co, h, w, part, nb1)
        switch (Ci / 32) { case 1: DLN(2); break; case 2: DLN(4); break; case 3: DLN(6); break; default: DLN(8); break; }
#undef DLN
        BF_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(dl_slice_sum_kernel, dim3(bf_cdiv(Ci, 64), frames), dim3(NT), 0, st, (const float*)part, Ci, nsl, tot);
    BF_CHECK_LAUNCH();
    {
        BfProfScope prof(st, "debed_last_bwd<apply>", 2.0 * P * Ci * 16, P * (4.0 * Ci + 32.0));
#define DLN(T) hipLaunchKernelGGL((debed_last_inbwd_kernel<T, 2>), grid, dim3(NT), 0, st, nullptr, nullptr, nullptr, nullptr, nullptr, (const bf16*)wc, (bf16*)dpm, (bf16*)dx, Co, h, w, nullptr, nb2)
        switch (Ci / 32) { case 1: DLN(2); break; case 2: DLN(4); break; case 3: DLN(6); break; default: DLN(8); break; }
#undef DLN
        BF_CHECK_LAUNCH();
    }
    if (d_in_w || d_in_b) {
        const InReduceJob j{tot, frames, Ci, in_w, in_b, nullptr, 1, d_in_w, d_in_b, nullptr, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(dl_param_reduce_kernel, dim3(bf_cdiv(Ci, 64), bf_cdiv(frames, j.rdiv())), dim3(64 * BF_RED_FL), 0, st, j);
        BF_CHECK_LAUNCH();
    }
    return 0;
}

// The first HMLPEmbed stage is the same contraction (layers/patching.py:30-48, Conv2d(k=2, s=2, bias=False) on the NCHW clip): rows of 2x2
// patches of x (k = c*4 + ky*2 + kx) times conv weight [C0][16]; `patches` is kept for the weight-gradient GEMM.
// `stat_part` (optional): per 256-row slice {mean, centred second moment} of y0 as stored, [frames][ceil(h2*w2/256)][C0][2] floats -- the
// partials bf_in_stats_merge_slices turns into the stage's InstanceNorm statistics, so that y0 is not read again for them.
extern "C" int bf_embed_first(int dtype, const float* x, const void* wc, void* patches, void* y0, int frames, int C0, int cin, int h2, int w2,
                              int Kp, float* stat_part, bf_stream_t stream) {
    BF_REQUIRE(x && wc && patches && (y0 || stat_part), "bf_embed_first: null pointer");      // y0 may be null when its statistics are all that is wanted of it
    static const bool off = bf_knob("BF_EMBED_FIRST", 1) == 0;
    if (off) return 1;
    return debed_last_bwd_launch(dtype, x, nullptr, nullptr, nullptr, nullptr, wc, patches, y0, frames, C0, cin, h2, w2, Kp, stat_part, stream);
}

extern "C" int bf_lploss_finalize(const float* lossbuf, int frames, int Co, float* loss, float* coef, bf_stream_t stream) {
    BF_REQUIRE(lossbuf && loss && frames > 0 && Co > 0, "bf_lploss_finalize: bad arguments");
    hipLaunchKernelGGL(lploss_finalize_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, lossbuf, frames, Co, loss, coef);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_nchw2pm(int dtype, const float* dpred, const float* pred, const float* y, const float* coef, const float* gscale,
                          void* dpm, int frames, int Co, int h, int w, int Np, bf_stream_t stream) {
    BF_REQUIRE(dpm && (dpred || (pred && y && coef)) && Np >= 4 * Co && Np % 4 == 0, "bf_nchw2pm: bad arguments");
    const long P = (long)frames * h * w;
    if (dtype == BF_DTYPE_BF16) hipLaunchKernelGGL(nchw2pm_kernel<bf16>, dim3(grid_for(P * (Np / 4))), dim3(NT), 0, (hipStream_t)stream, dpred, pred, y, coef, gscale, (bf16*)dpm, Co, h, w, Np, P);
    else hipLaunchKernelGGL(nchw2pm_kernel<float>, dim3(grid_for(P * (Np / 4))), dim3(NT), 0, (hipStream_t)stream, dpred, pred, y, coef, gscale, (float*)dpm, Co, h, w, Np, P);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_wprep(int dtype, int mode, const float* src, void* dst, int R, int K, int Kp, bf_stream_t stream) {
    BF_REQUIRE(src && dst && R > 0 && K > 0 && Kp >= K && mode >= 0 && mode <= 2, "bf_wprep: bad arguments");
    BF_REQUIRE(mode == 0 || Kp == K, "bf_wprep: padding only with mode 0");
    if (dtype == BF_DTYPE_BF16) hipLaunchKernelGGL(wprep_kernel<bf16>, dim3(grid_for((long)R * Kp)), dim3(NT), 0, (hipStream_t)stream, src, (bf16*)dst, mode, R, K, Kp);
    else hipLaunchKernelGGL(wprep_kernel<float>, dim3(grid_for((long)R * Kp)), dim3(NT), 0, (hipStream_t)stream, src, (float*)dst, mode, R, K, Kp);
    BF_CHECK_LAUNCH();
    return 0;
}

// bf_wprep for n <= BF_MAX_STAGES weights at once (same modes and argument meaning per entry)
int bf_wprep_multi(int dtype, int n, const int* mode, const float* const* src, void* const* dst, const int* R, const int* K, const int* Kp, hipStream_t st) {
    BF_REQUIRE(n >= 1 && n <= BF_MAX_STAGES && mode && src && dst && R && K && Kp, "bf_wprep_multi: bad arguments");
    WprepJobs j;
    long most = 0;
    for (int i = 0; i < n; ++i) {
        BF_REQUIRE(src[i] && dst[i] && R[i] > 0 && K[i] > 0 && Kp[i] >= K[i] && mode[i] >= 0 && mode[i] <= 2 && (mode[i] == 0 || Kp[i] == K[i]), "bf_wprep_multi: bad entry");
        j.src[i] = src[i]; j.dst[i] = dst[i]; j.mode[i] = mode[i]; j.R[i] = R[i]; j.K[i] = K[i]; j.Kp[i] = Kp[i];
        most = std::max(most, (long)R[i] * Kp[i]);
    }
    const dim3 grid(std::min<long>(bf_cdiv(most, NT), 256), n);
    if (dtype == BF_DTYPE_BF16) hipLaunchKernelGGL(wprep_multi_kernel<bf16>, grid, dim3(NT), 0, st, j);
    else hipLaunchKernelGGL(wprep_multi_kernel<float>, grid, dim3(NT), 0, st, j);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_wgrad_unprep(int mode, const float* gsrc, float* gdst, int R, int K, int Kp, int transposed, bf_stream_t stream) {
    BF_REQUIRE(gsrc && gdst && R > 0 && K > 0 && Kp >= K && mode >= 0 && mode <= 2, "bf_wgrad_unprep: bad arguments");
    hipLaunchKernelGGL(wgrad_unprep_kernel, dim3(grid_for((long)R * K)), dim3(NT), 0, (hipStream_t)stream, gsrc, gdst, mode, R, K, Kp, transposed);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_film_net_fwd(const float* cond, const float* lnw, const float* lnb, const float* W, const float* bias, float* gb,
                               float* chat, float* crstd, int B, int P, int E2, bf_stream_t stream) {
    BF_REQUIRE(cond && lnw && lnb && W && bias && gb && chat && crstd && B > 0 && P > 0 && E2 > 0, "bf_film_net_fwd: bad arguments");
    hipLaunchKernelGGL(film_net_fwd_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, cond, lnw, lnb, W, bias, gb, chat, crstd, B, P, E2);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_film_net_bwd(const float* dgb, const float* chat, const float* lnw, const float* lnb, const float* W, float* dW,
                               float* dbias, float* dlnw, float* dlnb, int B, int P, int E2, bf_stream_t stream) {
    BF_REQUIRE(dgb && chat && lnw && lnb && W && dW && dbias && dlnw && dlnb, "bf_film_net_bwd: bad arguments");
    hipLaunchKernelGGL(film_net_bwd_kernel, dim3(bf_cdiv(E2, 64) + P), dim3(64), 0, (hipStream_t)stream, dgb, chat, lnw, lnb, W, dW, dbias, dlnw, dlnb, B, P, E2);
    BF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------- physics metrics of a rollout
// Eikonal residual of a signed-distance field (utils/losses.py:5-15): torch.gradient(edge_order=1, spacing=dx) along H and W
// (central differences inside, one-sided at the borders), out += sum over pixels of (|grad phi| - 1)^2   (caller divides by the count)
__global__ void __launch_bounds__(NT) eikonal_kernel(const float* __restrict__ phi, long frames, int H, int W, float inv_dx, double* __restrict__ out) {
    __shared__ double red[NT / 64];
    const long total = frames * H * W;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float* p = phi + i;
        float gy, gx;
        if (H == 1) gy = 0.f;
        else if (y == 0) gy = (p[W] - p[0]) * inv_dx;
        else if (y == H - 1) gy = (p[0] - p[-W]) * inv_dx;
        else gy = (p[W] - p[-W]) * (0.5f * inv_dx);
        if (W == 1) gx = 0.f;
        else if (x == 0) gx = (p[1] - p[0]) * inv_dx;
        else if (x == W - 1) gx = (p[0] - p[-1]) * inv_dx;
        else gx = (p[1] - p[-1]) * (0.5f * inv_dx);
        const float r = sqrtf(gy * gy + gx * gx) - 1.f;
        acc += (double)(r * r);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < NT / 64; ++i) t += red[i];
        atomicAdd(out, t);
    }
}
// The rollout notebook's Eikonal score (scripts/inference_autoregressive.ipynb, `get_eikonal_loss`): per frame, the mean of
// | |grad phi| - 1 | with central differences at spacing dx in the interior and the border taking its neighbour's gradient
// (replicate padding).  One workgroup per frame.
__global__ void __launch_bounds__(NT) eikonal_l1_kernel(const float* __restrict__ phi, int H, int W, float inv_2dx, float* __restrict__ out) {
    __shared__ double red[NT / 64];
    const float* f = phi + (long)blockIdx.x * H * W;
    double acc = 0.0;
    for (int i = threadIdx.x; i < H * W; i += NT) {
        const int x = i % W, y = i / W;
        const int xi = min(max(x, 1), W - 2), yi = min(max(y, 1), H - 2);
        const float gx = (f[y * W + xi + 1] - f[y * W + xi - 1]) * inv_2dx;
        const float gy = (f[(yi + 1) * W + x] - f[(yi - 1) * W + x]) * inv_2dx;
        acc += (double)fabsf(sqrtf(gx * gx + gy * gy) - 1.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < NT / 64; ++i) t += red[i];
        out[blockIdx.x] = (float)(t / ((double)H * W));
    }
}
// Heater heat flux of FC-72 pool boiling per frame (utils/heatflux.py:17-38): bottom row y = 0 of a W-column grid spanning
// x in [x_min, x_min + W*dx); flux[t] = mean_x( [ |x| <= 5 and dfun < 0 ] * (heater_temp - temp) ) * 0.054 / (dx * lc)
__global__ void __launch_bounds__(64) heatflux_kernel(const float* __restrict__ dfun, const float* __restrict__ temp, long frame_stride, int W,
                                                     float x_min, float dx, float heater_temp, float coef, float* __restrict__ flux) {
    const float* d = dfun + (long)blockIdx.x * frame_stride;
    const float* t = temp + (long)blockIdx.x * frame_stride;
    double acc = 0.0;
    for (int x = threadIdx.x; x < W; x += 64) {
        const double xc = (double)x_min + ((double)x + 0.5) * (double)dx;
        if (xc >= -5.0 && xc <= 5.0 && d[x] < 0.f) acc += (double)(heater_temp - t[x]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) flux[blockIdx.x] = (float)(acc / (double)W * (double)coef);
}

extern "C" int bf_eikonal_sum(const float* phi, int64_t frames, int H, int W, float dx, double* out, bf_stream_t stream) {
    BF_REQUIRE(phi && out && frames > 0 && H > 0 && W > 0 && dx > 0.f, "bf_eikonal_sum: bad arguments");
    hipLaunchKernelGGL(eikonal_kernel, dim3(grid_for(frames * H * W)), dim3(NT), 0, (hipStream_t)stream, phi, (long)frames, H, W, 1.f / dx, out);
    BF_CHECK_LAUNCH();
    return 0;
}
extern "C" int bf_eikonal_l1_frames(const float* phi, int64_t frames, int H, int W, float dx, float* out, bf_stream_t stream) {
    BF_REQUIRE(phi && out && frames > 0 && H >= 3 && W >= 3 && dx > 0.f, "bf_eikonal_l1_frames: bad arguments (central differences need >= 3 points per axis)");
    hipLaunchKernelGGL(eikonal_l1_kernel, dim3((unsigned)frames), dim3(NT), 0, (hipStream_t)stream, phi, H, W, 0.5f / dx, out);
    BF_CHECK_LAUNCH();
    return 0;
}
extern "C" int bf_heatflux_rows(const float* dfun, const float* temp, int64_t frames, int64_t frame_stride, int W, float x_min, float dx,
                                float heater_temp, float lc, float* flux, bf_stream_t stream) {
    BF_REQUIRE(dfun && temp && flux && frames > 0 && W > 0 && dx > 0.f && lc > 0.f, "bf_heatflux_rows: bad arguments");
    hipLaunchKernelGGL(heatflux_kernel, dim3((unsigned)frames), dim3(64), 0, (hipStream_t)stream, dfun, temp, (long)frame_stride, W, x_min, dx,
                       heater_temp, 0.054f / (dx * lc), flux);
    BF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------- field statistics of device-resident trajectories
// The normalisation constants of the dataset (bubbleformer/data/dataset.py:74-117: mean / std / min / max of every full field of every file,
// which the reference reads through h5py and reduces on the host) from the trajectories ALREADY resident in HBM: one launch over all
// (field, file) segments.  Pass 1: every workgroup sweeps a contiguous share of one segment (16-byte loads) and leaves {sum, sum of squares,
// min, max} in fp64; pass 2: one wave per segment adds the workgroup rows in row order (bit-reproducible; no atomics).
constexpr int FS_ROWS = 64;      // workgroups per segment
__global__ void __launch_bounds__(NT) field_stats_kernel(const float* __restrict__ src, const long* __restrict__ seg_begin, const long* __restrict__ seg_len,
                                                        double* __restrict__ part) {
    __shared__ double red[NT / 64][4];
    const int seg = blockIdx.y;
    const float* p = src + seg_begin[seg];
    const long n = seg_len[seg];
    const long per = ((n + FS_ROWS - 1) / FS_ROWS + 3) & ~3L;                      // a multiple of 4 floats: whole 16-byte groups when the segment is aligned
    const long lo = (long)blockIdx.x * per, hi = min(n, lo + per);
    double s1 = 0.0, s2 = 0.0, mn = 1.0 / 0.0, mx = -1.0 / 0.0;
    auto take = [&](float v) { const double d = (double)v; s1 += d; s2 += d * d; mn = fmin(mn, d); mx = fmax(mx, d); };
    const bool vec = (((uintptr_t)p) & 15) == 0;
    long i = lo + 4L * threadIdx.x;
    if (vec)
        for (; i + 3 < hi; i += 4L * NT) { const float4 v = *reinterpret_cast<const float4*>(p + i); take(v.x); take(v.y); take(v.z); take(v.w); }
    else
        for (; i + 3 < hi; i += 4L * NT) { take(p[i]); take(p[i + 1]); take(p[i + 2]); take(p[i + 3]); }
    for (long j = i; j < hi && j < i + 4; ++j) take(p[j]);                            // the share's ragged end (at most one thread has one)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); mn = fmin(mn, __shfl_xor(mn, o, 64)); mx = fmax(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0) { double* r = red[threadIdx.x >> 6]; r[0] = s1; r[1] = s2; r[2] = mn; r[3] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, c = 1.0 / 0.0, e = -1.0 / 0.0;
        for (int w = 0; w < NT / 64; ++w) { a += red[w][0]; b += red[w][1]; c = fmin(c, red[w][2]); e = fmax(e, red[w][3]); }
        double* o = part + ((long)seg * FS_ROWS + blockIdx.x) * 4;
        o[0] = a; o[1] = b; o[2] = c; o[3] = e;
    }
}
__global__ void __launch_bounds__(64) field_stats_finish_kernel(const double* __restrict__ part, double* __restrict__ out) {
    const int seg = blockIdx.x;
    if (threadIdx.x != 0) return;
    double a = 0.0, b = 0.0, c = 1.0 / 0.0, e = -1.0 / 0.0;
    for (int r = 0; r < FS_ROWS; ++r) { const double* q = part + ((long)seg * FS_ROWS + r) * 4; a += q[0]; b += q[1]; c = fmin(c, q[2]); e = fmax(e, q[3]); }
    out[seg * 4] = a; out[seg * 4 + 1] = b; out[seg * 4 + 2] = c; out[seg * 4 + 3] = e;
}
extern "C" int64_t bf_field_stats_ws_doubles(int nseg) { return nseg > 0 ? (int64_t)nseg * FS_ROWS * 4 : 0; }
extern "C" int bf_field_stats(const float* src, const int64_t* seg_begin, const int64_t* seg_len, int nseg, double* out, double* ws, bf_stream_t stream) {
    BF_REQUIRE(src && seg_begin && seg_len && out && ws && nseg > 0 && nseg <= 65535, "bf_field_stats: bad arguments");
    hipLaunchKernelGGL(field_stats_kernel, dim3(FS_ROWS, (unsigned)nseg), dim3(NT), 0, (hipStream_t)stream, src, (const long*)seg_begin, (const long*)seg_len, ws);
    BF_CHECK_LAUNCH();
    hipLaunchKernelGGL(field_stats_finish_kernel, dim3((unsigned)nseg), dim3(64), 0, (hipStream_t)stream, (const double*)ws, out);
    BF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------- clip gather (device-resident trajectories -> batch)
// out[b][t][c][yo][xo] = (src[field[c]][first[b] + t0 + t][ys(yo)][xs(xo)] - diff[c]) / div[c]
// ys / xs: identity, or torch's F.interpolate(mode="nearest") source index floor(dst * float(in / out)) clamped to in - 1
// (bubbleformer/data/dataset.py:138-148).  One thread per 4 output pixels of a row; reads of a full-resolution row are 16-byte.
__global__ void __launch_bounds__(NT) clip_gather_kernel(const float* __restrict__ src, long field_stride, const int* __restrict__ field,
                                                        const long* __restrict__ first, int t0, const float* __restrict__ diff,
                                                        const float* __restrict__ dv, float* __restrict__ out, int B, int T, int C, int H, int W,
                                                        int Ho, int Wo) {
    const int wq = (Wo + 3) / 4;
    const long total = (long)B * T * C * Ho * wq;
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    const bool ident = Ho == H && Wo == W;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int xq = (int)(i % wq);
        long r = i / wq;
        const int yo = (int)(r % Ho); r /= Ho;
        const int c = (int)(r % C); r /= C;
        const int t = (int)(r % T);
        const int b = (int)(r / T);
        const int ys = ident ? yo : min((int)floorf((float)yo * sy), H - 1);
        const float* row = src + (long)field[c] * field_stride + ((first[b] + t0 + t) * H + ys) * (long)W;
        float* dst = out + ((((long)b * T + t) * C + c) * Ho + yo) * (long)Wo + 4 * xq;
        const float d = diff[c], q = dv[c];
        if (ident && (W & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(row + 4 * xq);
            *reinterpret_cast<float4*>(dst) = make_float4((v.x - d) / q, (v.y - d) / q, (v.z - d) / q, (v.w - d) / q);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xo = 4 * xq + j;
                if (xo < Wo) {
                    const int xs = ident ? xo : min((int)floorf((float)xo * sx), W - 1);
                    dst[j] = (row[xs] - d) / q;
                }
            }
        }
    }
}

extern "C" int bf_clip_gather(const float* src, int64_t field_stride, const int32_t* field, const int64_t* first, int t0,
                              const float* diff, const float* div, float* out, int B, int T, int C, int H, int W, int Ho, int Wo,
                              bf_stream_t stream) {
    BF_REQUIRE(src && field && first && diff && div && out, "bf_clip_gather: null pointer");
    BF_REQUIRE(B > 0 && T > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Ho <= H && Wo <= W && t0 >= 0, "bf_clip_gather: bad sizes");
    BF_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0), "bf_clip_gather: buffers must be 16-byte aligned");
    const long total = (long)B * T * C * Ho * ((Wo + 3) / 4);
    hipLaunchKernelGGL(clip_gather_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, src, (long)field_stride, (const int*)field,
                       (const long*)first, t0, diff, div, out, B, T, C, H, W, Ho, Wo);
    BF_CHECK_LAUNCH();
    return 0;
}

// A training batch in ONE launch: the input clips (frames first .. first + Tin - 1), the target clips (the Tout frames behind them) and the
// per-sample fluid-parameter rows, all indexed by SAMPLE number on the device (first_tab / file_tab: absolute first frame and file of every
// sample of the dataset) -- what took two launches and three index kernels of the host framework (first_tab[idx], file_tab[idx], fluid[...]).
struct ClipSeg { const int* field; const float* diff; const float* dv; float* out; int T, C, t0; };
__global__ void __launch_bounds__(NT) clip_gather_batch_kernel(const float* __restrict__ src, long field_stride, const long* __restrict__ idx, long nsamples,
                                                              const long* __restrict__ first_tab, ClipSeg a, ClipSeg b,
                                                              const float* __restrict__ fluid_tab, const long* __restrict__ file_tab, int P,
                                                              float* __restrict__ fluid_out, int B, int H, int W, int Ho, int Wo) {
    const int wq = (Wo + 3) / 4;
    const long per_a = (long)a.T * a.C * Ho * wq, per_b = (long)b.T * b.C * Ho * wq, per = per_a + per_b;
    const long total = (long)B * per;
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    const bool ident = Ho == H && Wo == W;
    if (fluid_out && blockIdx.x == 0)
        for (int i = threadIdx.x; i < B * P; i += NT) fluid_out[i] = fluid_tab[file_tab[min(max(idx[i / P], 0L), nsamples - 1)] * P + i % P];
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int bb = (int)(i / per);
        long r = i - (long)bb * per;
        const bool second = r >= per_a;
        if (second) r -= per_a;
        const ClipSeg& sg = second ? b : a;
        const int xq = (int)(r % wq); r /= wq;
        const int yo = (int)(r % Ho); r /= Ho;
        const int c = (int)(r % sg.C);
        const int t = (int)(r / sg.C);
        const int ys = ident ? yo : min((int)floorf((float)yo * sy), H - 1);
        const long smp = min(max(idx[bb], 0L), nsamples - 1);      // an index out of range reads a valid sample, never past a table
        const float* row = src + (long)sg.field[c] * field_stride + ((first_tab[smp] + sg.t0 + t) * H + ys) * (long)W;
        float* dst = sg.out + ((((long)bb * sg.T + t) * sg.C + c) * Ho + yo) * (long)Wo + 4 * xq;
        const float d = sg.diff[c], q = sg.dv[c];
        if (ident && (W & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(row + 4 * xq);
            *reinterpret_cast<float4*>(dst) = make_float4((v.x - d) / q, (v.y - d) / q, (v.z - d) / q, (v.w - d) / q);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xo = 4 * xq + j;
                if (xo < Wo) {
                    const int xs = ident ? xo : min((int)floorf((float)xo * sx), W - 1);
                    dst[j] = (row[xs] - d) / q;
                }
            }
        }
    }
}
extern "C" int bf_clip_gather_batch(const float* src, int64_t field_stride, const int64_t* idx, int64_t nsamples, const int64_t* first_tab, const int32_t* in_field,
                                    const float* in_diff, const float* in_div, int Cin, int Tin, float* in_out, const int32_t* out_field,
                                    const float* out_diff, const float* out_div, int Cout, int Tout, float* out_out, const float* fluid_tab,
                                    const int64_t* file_tab, int P, float* fluid_out, int B, int H, int W, int Ho, int Wo, bf_stream_t stream) {
    BF_REQUIRE(src && idx && first_tab && in_field && in_diff && in_div && in_out && out_field && out_diff && out_div && out_out,
               "bf_clip_gather_batch: null pointer");
    BF_REQUIRE(nsamples > 0 && B > 0 && Tin > 0 && Tout > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Ho <= H && Wo <= W, "bf_clip_gather_batch: bad sizes");
    BF_REQUIRE(!fluid_out || (fluid_tab && file_tab && P > 0), "bf_clip_gather_batch: the fluid rows need their table, the file table and P > 0");
    BF_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)in_out % 16 == 0) && ((uintptr_t)out_out % 16 == 0), "bf_clip_gather_batch: buffers must be 16-byte aligned");
    const ClipSeg a{(const int*)in_field, in_diff, in_div, in_out, Tin, Cin, 0}, b{(const int*)out_field, out_diff, out_div, out_out, Tout, Cout, Tin};
    const long total = (long)B * ((long)Tin * Cin + (long)Tout * Cout) * Ho * ((Wo + 3) / 4);
    hipLaunchKernelGGL(clip_gather_batch_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, src, (long)field_stride, (const long*)idx, (long)nsamples,
                       (const long*)first_tab, a, b, fluid_tab, (const long*)file_tab, P, fluid_out, B, H, W, Ho, Wo);
    BF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------- Lion (Chen et al. 2023, "Symbolic Discovery of Optimization
// Algorithms"; the update lion_pytorch.Lion applies at bubbleformer/modules.py:139-140):
//   p *= 1 - lr*wd;  p -= lr * sign(b1*m + (1-b1)*g);  m = b2*m + (1-b2)*g
__global__ void __launch_bounds__(NT) lion_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, long n, float lr,
                                                 float b1, float b2, float wd, float gscale) {
    const long n4 = n / 4;
    auto sgn = [](float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); };
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
#define BF_LION1(X)                                                        \
        { const float gr = gg.X * gscale;                                   \
          pp.X = pp.X * (1.f - lr * wd) - lr * sgn(b1 * mm.X + (1.f - b1) * gr); \
          mm.X = b2 * mm.X + (1.f - b2) * gr; }
        BF_LION1(x) BF_LION1(y) BF_LION1(z) BF_LION1(w)
#undef BF_LION1
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
        const long i = n4 * 4 + threadIdx.x;
        const float gr = g[i] * gscale;
        p[i] = p[i] * (1.f - lr * wd) - lr * sgn(b1 * m[i] + (1.f - b1) * gr);
        m[i] = b2 * m[i] + (1.f - b2) * gr;
    }
}

extern "C" int bf_lion(float* p, const float* g, float* m, int64_t n, float lr, float beta1, float beta2, float wd, float gscale,
                       bf_stream_t stream) {
    BF_REQUIRE(p && g && m && n > 0, "bf_lion: bad arguments");
    BF_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0), "bf_lion: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(lion_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, (hipStream_t)stream, p, g, m, (long)n, lr, beta1, beta2, wd, gscale);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_adamw(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float beta1, float beta2,
                        float eps, float wd, float gscale, bf_stream_t stream) {
    BF_REQUIRE(p && g && m && v && n > 0 && step >= 1, "bf_adamw: bad arguments");
    BF_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "bf_adamw: buffers must be 16-byte aligned");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float sbc2 = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, beta1, beta2, eps, wd, bc1, sbc2, gscale);
    BF_CHECK_LAUNCH();
    return 0;
}
