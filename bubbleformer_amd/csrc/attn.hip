// Small-sequence attention for the factored space-time blocks (L <= 32 along T, W or H).
//
// One wavefront owns one (sequence, head) problem: q/k/v rows are pulled into LDS as fp32
// (16-byte global chunks, each token row contributes 3*d contiguous elements of the
// head-interleaved QKV tensor), q/k LayerNorm, q k^T * d^-1/2 + T5 bias, softmax and the
// high-frequency rescale 1/L + (p - 1/L) * s_head run wave-locally, then P V.
// A "sequence" is described by strides so the same kernel walks T (temporal block),
// W or H (axial block): token(l) = (s / inner) * outer_stride + (s % inner) * inner_stride + l * tok_stride.
// The backward recomputes P from q/k (nothing but QKV is saved) and produces dQKV plus the
// LayerNorm / bias-table / scale-factor gradients (block-reduced, then fp32 atomics).
//
// This is the generic fp32-VALU form (any d <= 128, any L <= 32, both dtypes).
#include "bf_common.h"

namespace {

constexpr int NT = 256;           // at most 4 waves per block; fewer when a problem's LDS plan is large
__device__ __forceinline__ void wave_sync() {
    // the wave's own LDS traffic is in order; this only pins the compiler's ordering of it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
constexpr int LMAX = 32;
constexpr int DMAX = 128;

// one-sided T5 bucket for |offset| (num_buckets 32 -> 16 per side, max_exact 8, max_distance 32);
// restated from the reference formula, checked against the reference's tables in the tests.
__device__ __forceinline__ int t5_bucket(int n) {
    const int a = n < 0 ? -n : n;
    int b;
    if (a < 8) b = a;
    else if (a < 10) b = 8;
    else if (a < 12) b = 9;
    else if (a < 14) b = 10;
    else if (a < 16) b = 11;
    else if (a < 20) b = 12;
    else if (a < 23) b = 13;
    else if (a < 27) b = 14;
    else b = 15;
    return b + (n < 0 ? 16 : 0);   // n = query - key; key after query -> upper half
}

struct SeqGeo {
    long nseq; int L; long inner; long outer_stride; long inner_stride; long tok_stride;
};
__device__ __forceinline__ long seq_base(const SeqGeo& g, long s) {
    return (s / g.inner) * g.outer_stride + (s % g.inner) * g.inner_stride;
}

struct AttnParams {
    const float *qw, *qb, *kw, *kb;   // [d]
    const float* emb;                 // [32][heads] or null
    const float* hscale;              // [heads] or null
};
struct AttnGrads {
    float *dqw, *dqb, *dkw, *dkb, *demb, *dhscale;
};

template <typename T>
__device__ __forceinline__ void load_rows(const T* __restrict__ src, long row_stride, float* __restrict__ dst, int L,
                                          int width, int ldd, int dsplit, long tok0, long tok_stride, int lane, float mul) {
    // copies L rows of `width` contiguous elements into dst; element e of row l goes to
    // dst[(e / dsplit) * L * ldd + l * ldd + e % dsplit]  (splits q|k|v into three [L][ldd] planes)
    constexpr int CH = Chunk<T>::N;
    const int cpr = width / CH;
    for (int i = lane; i < L * cpr; i += 64) {
        const int l = i / cpr, e0 = (i % cpr) * CH;
        Chunk<T> v;
        v.load(src + (tok0 + l * tok_stride) * row_stride + e0);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int e = e0 + j;
            dst[(e / dsplit) * L * ldd + l * ldd + (e % dsplit)] = v.get(j) * mul;
        }
    }
}

// LayerNorm over d for 2L rows (q rows then k rows), in place -> xhat; rstd kept per row.
__device__ __forceinline__ void ln_rows(float* q, float* k, float* rstd, int L, int d, int ldd, int lane) {
    if (lane < 2 * L) {
        float* row = (lane < L) ? (q + lane * ldd) : (k + (lane - L) * ldd);
        float mu = 0.f;
        for (int e = 0; e < d; ++e) mu += row[e];
        mu /= (float)d;
        float var = 0.f;
        for (int e = 0; e < d; ++e) { const float t = row[e] - mu; var += t * t; }
        const float r = rsqrtf(var / (float)d + BF_IN_EPS);
        for (int e = 0; e < d; ++e) row[e] = (row[e] - mu) * r;
        rstd[lane] = r;
    }
}

// scores -> P (softmax) in sP, A (rescaled) in sA (may alias sP when P itself is not needed later)
__device__ __forceinline__ void scores_softmax(const float* qh, const float* kh, const AttnParams& p, int head, int heads,
                                               int L, int d, int ldd, float* sP, float* sA, int lane) {
    const int lds = L + 1;
    const float scale = rsqrtf((float)d);
    for (int t = lane; t < L * L; t += 64) {
        const int i = t / L, j = t % L;
        float acc = 0.f;
        for (int e = 0; e < d; ++e) {
            const float qn = qh[i * ldd + e] * p.qw[e] + p.qb[e];
            const float kn = kh[j * ldd + e] * p.kw[e] + p.kb[e];
            acc += qn * kn;
        }
        float s = acc * scale;
        if (p.emb) s += p.emb[t5_bucket(i - j) * heads + head];
        sP[i * lds + j] = s;
    }
    wave_sync();
    if (lane < L) {
        float* row = sP + lane * lds;
        float m = -INFINITY;
        for (int j = 0; j < L; ++j) m = fmaxf(m, row[j]);
        float sum = 0.f;
        for (int j = 0; j < L; ++j) { const float e = __expf(row[j] - m); row[j] = e; sum += e; }
        const float inv = 1.f / sum;
        const float invL = 1.0f / (float)L;
        const float hs = p.hscale ? p.hscale[head] : 1.f;
        for (int j = 0; j < L; ++j) {
            const float pr = row[j] * inv;
            row[j] = pr;
            sA[lane * lds + j] = p.hscale ? (invL + (pr - invL) * hs) : pr;
        }
    }
    wave_sync();
}

template <typename T>
__global__ void __launch_bounds__(NT) attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, SeqGeo g, int heads, int d,
                                                     AttnParams p, float out_scale, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = g.L, ldd = d + 1, E = heads * d;
    const int per_wave = 3 * L * ldd + L * (L + 1) + 2 * L;
    float* q = smem + wave * per_wave;
    float* k = q + L * ldd;
    float* v = k + L * ldd;
    float* sA = v + L * ldd;
    float* rstd = sA + L * (L + 1);
    const long nprob = g.nseq * heads;
    // LN'd q/k affine is folded into the score loop, so stage qn = xhat*w+b once instead: do it in place
    const int WPB = blockDim.x >> 6;
    for (long pr = (long)blockIdx.x * WPB + wave; pr < nprob; pr += (long)gridDim.x * WPB) {
        const long s = pr / heads;
        const int head = (int)(pr % heads);
        const long tok0 = seq_base(g, s);
        load_rows<T>(qkv + head * 3 * d, 3L * E, q, L, 3 * d, ldd, d, tok0, g.tok_stride, lane, 1.f);
        wave_sync();
        ln_rows(q, k, rstd, L, d, ldd, lane);
        wave_sync();
        scores_softmax(q, k, p, head, heads, L, d, ldd, sA, sA, lane);
        for (int e = lane; e < d; e += 64) {
            for (int i = 0; i < L; ++i) {
                float acc = 0.f;
                for (int j = 0; j < L; ++j) acc += sA[i * (L + 1) + j] * v[j * ldd + e];
                const long o = (tok0 + i * g.tok_stride) * E + head * d + e;
                float r = acc * out_scale;
                if (accumulate) r += to_f(out[o]);
                out[o] = from_f<T>(r);
            }
        }
        wave_sync();
    }
}

template <typename T>
__global__ void __launch_bounds__(NT) attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                     SeqGeo g, int heads, int d, AttnParams p, AttnGrads gr, float out_scale,
                                                     int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float s_demb[32 * 16];     // [bucket][head] (heads <= 16), block-reduced
    __shared__ float s_dhs[16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = g.L, ldd = d + 1, E = heads * d, lds = L + 1;
    const int per_wave = 6 * L * ldd + 2 * L * lds + 2 * L;
    float* q = smem + wave * per_wave;      // xhat_q
    float* k = q + L * ldd;                 // xhat_k
    float* v = k + L * ldd;
    float* dO = v + L * ldd;                // dout rows (scaled)
    float* dqn = dO + L * ldd;
    float* dkn = dqn + L * ldd;
    float* sP = dkn + L * ldd;
    float* sA = sP + L * lds;               // A, later dS
    float* rstd = sA + L * lds;
    for (int i = threadIdx.x; i < 32 * 16; i += blockDim.x) s_demb[i] = 0.f;
    if (threadIdx.x < 16) s_dhs[threadIdx.x] = 0.f;
    __syncthreads();
    // per-lane LayerNorm parameter gradient accumulators (columns lane, lane + 64)
    float a_qw[2] = {0.f, 0.f}, a_qb[2] = {0.f, 0.f}, a_kw[2] = {0.f, 0.f}, a_kb[2] = {0.f, 0.f};
    const float scale = rsqrtf((float)d);
    const long nprob = g.nseq * heads;
    const int WPB = blockDim.x >> 6;
    for (long pr = (long)blockIdx.x * WPB + wave; pr < nprob; pr += (long)gridDim.x * WPB) {
        const long s = pr / heads;
        const int head = (int)(pr % heads);
        const long tok0 = seq_base(g, s);
        load_rows<T>(qkv + head * 3 * d, 3L * E, q, L, 3 * d, ldd, d, tok0, g.tok_stride, lane, 1.f);
        load_rows<T>(dout + head * d, (long)E, dO, L, d, ldd, d, tok0, g.tok_stride, lane, out_scale);
        wave_sync();
        ln_rows(q, k, rstd, L, d, ldd, lane);
        wave_sync();
        scores_softmax(q, k, p, head, heads, L, d, ldd, sP, sA, lane);
        // dV[j][e] = sum_i A[i][j] dO[i][e]   -> stored into dkn temporarily? no: write straight to global later; keep in regs
        // dA[i][j] = sum_e dO[i][e] v[j][e]
        const float hs = p.hscale ? p.hscale[head] : 1.f;
        const float invL = 1.0f / (float)L;
        float dhs_part = 0.f;
        // stage dA in dqn's storage region is unsafe (dqn written later) -> use sA after reading A for dV.
        // 1) dV into dkn plane (it is free until dkn is produced); written to global at the end.
        for (int e = lane; e < d; e += 64)
            for (int j = 0; j < L; ++j) {
                float acc = 0.f;
                for (int i = 0; i < L; ++i) acc += sA[i * lds + j] * dO[i * ldd + e];
                dqn[j * ldd + e] = acc;          // dV parked in the dqn plane
            }
        wave_sync();
        // write dV now (frees the plane)
        {
            constexpr int CH = Chunk<T>::N;
            const int cpr = d / CH;
            for (int i = lane; i < L * cpr; i += 64) {
                const int l = i / cpr, e0 = (i % cpr) * CH;
                T* dst = dqkv + (tok0 + l * g.tok_stride) * 3L * E + head * 3 * d + 2 * d + e0;
                Chunk<T> o, old;
                if (accumulate) old.load(dst);
#pragma unroll
                for (int j = 0; j < CH; ++j) o.set(j, dqn[l * ldd + e0 + j] + (accumulate ? old.get(j) : 0.f));
                o.store(dst);
            }
        }
        wave_sync();
        // 2) dA -> dP -> dS (into sA)
        for (int t = lane; t < L * L; t += 64) {
            const int i = t / L, j = t % L;
            float acc = 0.f;
            for (int e = 0; e < d; ++e) acc += dO[i * ldd + e] * v[j * ldd + e];
            if (p.hscale) { dhs_part += (sP[i * lds + j] - invL) * acc; acc *= hs; }
            sA[i * lds + j] = acc;               // dP
        }
        wave_sync();
        if (lane < L) {
            float dot = 0.f;
            for (int j = 0; j < L; ++j) dot += sP[lane * lds + j] * sA[lane * lds + j];
            for (int j = 0; j < L; ++j) sA[lane * lds + j] = sP[lane * lds + j] * (sA[lane * lds + j] - dot);   // dS
        }
        wave_sync();
        // bias-table gradient: Toeplitz -> one LDS atomic per (i, j)
        if (gr.demb)
            for (int t = lane; t < L * L; t += 64) {
                const int i = t / L, j = t % L;
                atomicAdd(&s_demb[t5_bucket(i - j) * 16 + head], sA[i * lds + j]);
            }
        if (p.hscale) {
            dhs_part = wave_sum(dhs_part);
            if (lane == 0 && gr.dhscale) atomicAdd(&s_dhs[head], dhs_part);
        }
        // 3) dqn[i][e] = scale * sum_j dS[i][j] kn[j][e];  dkn[j][e] = scale * sum_i dS[i][j] qn[i][e]
        for (int e = lane; e < d; e += 64) {
            const float kw = p.kw[e], kb = p.kb[e], qw = p.qw[e], qb = p.qb[e];
            for (int i = 0; i < L; ++i) {
                float a1 = 0.f, a2 = 0.f;
                for (int j = 0; j < L; ++j) {
                    a1 += sA[i * lds + j] * (k[j * ldd + e] * kw + kb);
                    a2 += sA[j * lds + i] * (q[j * ldd + e] * qw + qb);
                }
                dqn[i * ldd + e] = a1 * scale;
                dkn[i * ldd + e] = a2 * scale;
            }
        }
        wave_sync();
        // LayerNorm parameter grads (per column) ...
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int e = lane + 64 * c;
            if (e < d) {
                float sw = 0.f, sb = 0.f, tw = 0.f, tb = 0.f;
                for (int i = 0; i < L; ++i) {
                    sw += dqn[i * ldd + e] * q[i * ldd + e]; sb += dqn[i * ldd + e];
                    tw += dkn[i * ldd + e] * k[i * ldd + e]; tb += dkn[i * ldd + e];
                }
                a_qw[c] += sw; a_qb[c] += sb; a_kw[c] += tw; a_kb[c] += tb;
            }
        }
        wave_sync();
        // ... and LayerNorm input grads, row per lane, in place: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dn * w
        if (lane < 2 * L) {
            const bool isq = lane < L;
            const int r = isq ? lane : lane - L;
            float* dn = (isq ? dqn : dkn) + r * ldd;
            const float* xh = (isq ? q : k) + r * ldd;
            const float* w = isq ? p.qw : p.kw;
            float m1 = 0.f, m2 = 0.f;
            for (int e = 0; e < d; ++e) { const float gg = dn[e] * w[e]; m1 += gg; m2 += gg * xh[e]; }
            m1 /= (float)d; m2 /= (float)d;
            const float rs = rstd[lane];
            for (int e = 0; e < d; ++e) dn[e] = rs * (dn[e] * w[e] - m1 - xh[e] * m2);
        }
        wave_sync();
        // write dq, dk
        {
            constexpr int CH = Chunk<T>::N;
            const int cpr = d / CH;
            for (int i = lane; i < 2 * L * cpr; i += 64) {
                const int part = i / (L * cpr);
                const int l = (i / cpr) % L, e0 = (i % cpr) * CH;
                const float* src = (part ? dkn : dqn) + l * ldd + e0;
                T* dst = dqkv + (tok0 + l * g.tok_stride) * 3L * E + head * 3 * d + part * d + e0;
                Chunk<T> o, old;
                if (accumulate) old.load(dst);
#pragma unroll
                for (int j = 0; j < CH; ++j) o.set(j, src[j] + (accumulate ? old.get(j) : 0.f));
                o.store(dst);
            }
        }
        wave_sync();
    }
    // flush parameter gradients
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int e = lane + 64 * c;
        if (e < d && gr.dqw) { atomicAdd(gr.dqw + e, a_qw[c]); atomicAdd(gr.dqb + e, a_qb[c]); atomicAdd(gr.dkw + e, a_kw[c]); atomicAdd(gr.dkb + e, a_kb[c]); }
    }
    __syncthreads();
    if (gr.demb)
        for (int i = threadIdx.x; i < 32 * heads; i += blockDim.x) {
            const float val = s_demb[(i / heads) * 16 + (i % heads)];
            if (val != 0.f) atomicAdd(gr.demb + i, val);
        }
    if (gr.dhscale && threadIdx.x < heads) atomicAdd(gr.dhscale + threadIdx.x, s_dhs[threadIdx.x]);
}

// waves per block so the dynamic LDS plan fits; raises the kernel's dynamic-LDS limit when needed
template <typename K>
int plan_waves(K kernel, size_t floats_per_wave, int* wpb, size_t* shm) {
    int w = 4;
    while (w > 1 && (size_t)w * floats_per_wave * sizeof(float) > 64 * 1024) w >>= 1;
    *wpb = w;
    *shm = (size_t)w * floats_per_wave * sizeof(float);
    if (*shm > 156 * 1024) return bf_fail_msg("attention: L*d too large for the LDS plan", __FILE__, __LINE__);
    if (*shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)*shm);
        if (e != hipSuccess) return bf_fail(e, __FILE__, __LINE__);
    }
    return 0;
}

bool g_force_generic = false;

int check_geo(const char* who, int dtype, int heads, int d, int L) {
    const int ch = dtype == BF_DTYPE_BF16 ? 8 : 4;
    if (heads < 1 || heads > 16) return bf_fail_msg("attention: heads must be in 1..16", who, 0);
    if (d < ch || d > DMAX || d % ch) return bf_fail_msg("attention: head dim must be a multiple of the 16-byte chunk and <= 128", who, 0);
    if (L < 1 || L > LMAX) return bf_fail_msg("attention: sequence length must be in 1..32", who, 0);
    return 0;
}

}  // namespace

// attn_mfma.hip
int bf_attn_fwd_mfma(const void* qkv, void* out, long nseq, int L, long inner, long outer_stride, long inner_stride, long tok_stride, int heads,
                     int d, const float* qw, const float* qb, const float* kw, const float* kb, const float* emb, const float* hscale,
                     float out_scale, int accumulate, hipStream_t st);
int bf_attn_bwd_mfma(const void* qkv, const void* dout, void* dqkv, long nseq, int L, long inner, long outer_stride, long inner_stride,
                     long tok_stride, int heads, int d, const float* qw, const float* qb, const float* kw, const float* kb, const float* emb,
                     const float* hscale, float* dqw, float* dqb, float* dkw, float* dkb, float* demb, float* dhscale, float out_scale,
                     int accumulate, float* ws, long ws_floats, int* rows_out, hipStream_t st);
int bf_attn_axial_fwd_mfma(const void* qkv, void* out, int frames, int h, int w, int heads, int d, const float* qw, const float* qb, const float* kw,
                           const float* kb, const float* emb, const float* hscale_x, const float* hscale_y, const float* nw, const float* nb,
                           void* out_n, float* mean, float* rstd, float* sc, float* sh, hipStream_t st);
static bool use_mfma(int dtype, int d) { return !g_force_generic && dtype == BF_DTYPE_BF16 && d % 32 == 0 && d <= 128; }
// library-internal: whether bf_attn_bwd takes the raw-gradient modes (accumulate 2 / 5: the LayerNorm backward once for two passes)
bool bf_attn_raw_modes(int dtype, int d) { return use_mfma(dtype, d); }

extern "C" void bf_debug_force_generic_attn(int on) { g_force_generic = on != 0; }

extern "C" int bf_attn_fwd(int dtype, const void* qkv, void* out, int64_t nseq, int L, int64_t inner, int64_t outer_stride,
                           int64_t inner_stride, int64_t tok_stride, int heads, int d, const float* qw, const float* qb,
                           const float* kw, const float* kb, const float* emb, const float* hscale, float out_scale,
                           int accumulate, bf_stream_t stream) {
    BF_REQUIRE(qkv && out && qw && qb && kw && kb && nseq > 0 && inner > 0, "bf_attn_fwd: bad arguments");
    if (int rc = check_geo("bf_attn_fwd", dtype, heads, d, L)) return rc;
    if (use_mfma(dtype, d)) {
        BfProfScope prof((hipStream_t)stream, "attn_fwd", 4.0 * nseq * heads * L * L * d, (double)nseq * heads * L * d * 2.0 * (accumulate ? 5.0 : 4.0));
        return bf_attn_fwd_mfma(qkv, out, nseq, L, inner, outer_stride, inner_stride, tok_stride, heads, d, qw, qb, kw, kb, emb, hscale, out_scale,
                                accumulate, (hipStream_t)stream);
    }
    SeqGeo g{nseq, L, inner, outer_stride, inner_stride, tok_stride};
    AttnParams p{qw, qb, kw, kb, emb, hscale};
    const size_t fpw = 3 * L * (d + 1) + L * (L + 1) + 2 * L;
    int WPB; size_t shm;
    if (int rc = dtype == BF_DTYPE_BF16 ? plan_waves(attn_fwd_kernel<bf16>, fpw, &WPB, &shm) : plan_waves(attn_fwd_kernel<float>, fpw, &WPB, &shm)) return rc;
    const long nprob = nseq * heads;
    BfProfScope prof((hipStream_t)stream, "attn_fwd", 4.0 * nprob * L * L * d, (double)nprob * L * d * bf_esize(dtype) * (accumulate ? 5.0 : 4.0));
    const int grid = (int)std::min<long>((nprob + WPB - 1) / WPB, 256 * 8);
    if (dtype == BF_DTYPE_BF16)
        hipLaunchKernelGGL(attn_fwd_kernel<bf16>, dim3(grid), dim3(WPB * 64), shm, (hipStream_t)stream, (const bf16*)qkv, (bf16*)out, g, heads, d, p, out_scale, accumulate);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(grid), dim3(WPB * 64), shm, (hipStream_t)stream, (const float*)qkv, (float*)out, g, heads, d, p, out_scale, accumulate);
    BF_CHECK_LAUNCH();
    return 0;
}

// Both axial passes of AxialAttentionBlock (along W, then along H, averaged: layers/attention.py:212-297) on QKV [frames*h*w][3E].
// One launch where the fused kernel covers the shape, otherwise the two bf_attn_fwd calls it is bit-identical to.
extern "C" int bf_attn_axial_fwd(int dtype, const void* qkv, void* out, int64_t frames, int h, int w, int heads, int d, const float* qw,
                                 const float* qb, const float* kw, const float* kb, const float* emb, const float* hscale_x,
                                 const float* hscale_y, bf_stream_t stream) {
    BF_REQUIRE(qkv && out && qw && qb && kw && kb && frames > 0 && h > 0 && w > 0, "bf_attn_axial_fwd: bad arguments");
    if (use_mfma(dtype, d) && frames < (1L << 24)) {
        BfProfScope prof((hipStream_t)stream, "attn_fwd", 4.0 * frames * heads * h * w * (h + w) * d, (double)frames * h * w * heads * d * 2.0 * 4.0);
        const int rc = bf_attn_axial_fwd_mfma(qkv, out, (int)frames, h, w, heads, d, qw, qb, kw, kb, emb, hscale_x, hscale_y, nullptr, nullptr, nullptr,
                                              nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
        if (rc <= 0) return rc;
    }
    const long S = (long)h * w;
    int rc = bf_attn_fwd(dtype, qkv, out, frames * h, w, 1, w, 0, 1, heads, d, qw, qb, kw, kb, emb, hscale_x, 0.5f, 0, stream);
    if (rc) return rc;
    return bf_attn_fwd(dtype, qkv, out, frames * w, h, w, S, 1, w, heads, d, qw, qb, kw, kb, emb, hscale_y, 0.5f, 1, stream);
}

// bf_attn_axial_fwd followed by the InstanceNorm2d of AxialAttentionBlock.norm2 (layers/attention.py:298) in the same launch: also
// writes out_n = (out - mean) * rstd * w + b and mean / rstd / sc / sh [frames][E] as bf_in_stats does.  Returns 1 when the one-launch
// form does not cover the shape (the caller then runs bf_attn_axial_fwd and bf_in_stats / bf_affine_apply).
extern "C" int bf_attn_axial_norm_fwd(int dtype, const void* qkv, void* out, void* out_n, int64_t frames, int h, int w, int heads, int d,
                                      const float* qw, const float* qb, const float* kw, const float* kb, const float* emb,
                                      const float* hscale_x, const float* hscale_y, const float* norm_w, const float* norm_b, float* mean,
                                      float* rstd, float* sc, float* sh, bf_stream_t stream) {
    BF_REQUIRE(qkv && out && out_n && qw && qb && kw && kb && norm_w && norm_b && mean && rstd && sc && sh && frames > 0 && h > 0 && w > 0,
               "bf_attn_axial_norm_fwd: bad arguments");
    static const bool off = bf_knob("BF_ATTN_AXIAL_NORM", 1) == 0;
    if (off || !use_mfma(dtype, d) || frames >= (1L << 24)) return 1;
    BfProfScope prof((hipStream_t)stream, "attn_fwd", 4.0 * frames * heads * h * w * (h + w) * d, (double)frames * h * w * heads * d * 2.0 * 5.0);
    return bf_attn_axial_fwd_mfma(qkv, out, (int)frames, h, w, heads, d, qw, qb, kw, kb, emb, hscale_x, hscale_y, norm_w, norm_b, out_n, mean, rstd, sc,
                                  sh, (hipStream_t)stream);
}

static int attn_bwd_impl(int dtype, const void* qkv, const void* dout, void* dqkv, int64_t nseq, int L, int64_t inner,
                         int64_t outer_stride, int64_t inner_stride, int64_t tok_stride, int heads, int d, const float* qw,
                         const float* qb, const float* kw, const float* kb, const float* emb, const float* hscale,
                         float* dqw, float* dqb, float* dkw, float* dkb, float* demb, float* dhscale, float out_scale,
                         int accumulate, float* ws, int64_t ws_floats, int* rows_out, bf_stream_t stream) {
    if (rows_out) *rows_out = 0;
    BF_REQUIRE(qkv && dout && dqkv && qw && qb && kw && kb && nseq > 0 && inner > 0, "bf_attn_bwd: bad arguments");
    if (int rc = check_geo("bf_attn_bwd", dtype, heads, d, L)) return rc;
    BF_REQUIRE(accumulate == 0 || accumulate == 1 || ((accumulate == 2 || accumulate == 5) && use_mfma(dtype, d)),
               "bf_attn_bwd: accumulate must be 0 / 1 (or 2 / 5, the raw-gradient pair of passes, on the bf16 MFMA path)");
    if (use_mfma(dtype, d)) {
        BfProfScope prof((hipStream_t)stream, "attn_bwd", 10.0 * nseq * heads * L * L * d, (double)nseq * heads * L * d * 2.0 * (accumulate ? 10.0 : 7.0));
        return bf_attn_bwd_mfma(qkv, dout, dqkv, nseq, L, inner, outer_stride, inner_stride, tok_stride, heads, d, qw, qb, kw, kb, emb, hscale, dqw,
                                dqb, dkw, dkb, demb, dhscale, out_scale, accumulate, ws, (long)ws_floats, rows_out, (hipStream_t)stream);
    }
    SeqGeo g{nseq, L, inner, outer_stride, inner_stride, tok_stride};
    AttnParams p{qw, qb, kw, kb, emb, hscale};
    AttnGrads gr{dqw, dqb, dkw, dkb, demb, dhscale};
    const size_t fpw = 6 * L * (d + 1) + 2 * L * (L + 1) + 2 * L;
    int WPB; size_t shm;
    if (int rc = dtype == BF_DTYPE_BF16 ? plan_waves(attn_bwd_kernel<bf16>, fpw, &WPB, &shm) : plan_waves(attn_bwd_kernel<float>, fpw, &WPB, &shm)) return rc;
    const long nprob = nseq * heads;
    BfProfScope prof((hipStream_t)stream, "attn_bwd", 10.0 * nprob * L * L * d, (double)nprob * L * d * bf_esize(dtype) * (accumulate ? 10.0 : 7.0));
    const int grid = (int)std::min<long>((nprob + WPB - 1) / WPB, 256 * 4);
    if (dtype == BF_DTYPE_BF16)
        hipLaunchKernelGGL(attn_bwd_kernel<bf16>, dim3(grid), dim3(WPB * 64), shm, (hipStream_t)stream, (const bf16*)qkv, (const bf16*)dout, (bf16*)dqkv, g, heads, d, p, gr, out_scale, accumulate);
    else
        hipLaunchKernelGGL(attn_bwd_kernel<float>, dim3(grid), dim3(WPB * 64), shm, (hipStream_t)stream, (const float*)qkv, (const float*)dout, (float*)dqkv, g, heads, d, p, gr, out_scale, accumulate);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_attn_bwd(int dtype, const void* qkv, const void* dout, void* dqkv, int64_t nseq, int L, int64_t inner,
                           int64_t outer_stride, int64_t inner_stride, int64_t tok_stride, int heads, int d, const float* qw,
                           const float* qb, const float* kw, const float* kb, const float* emb, const float* hscale,
                           float* dqw, float* dqb, float* dkw, float* dkb, float* demb, float* dhscale, float out_scale,
                           int accumulate, float* ws, int64_t ws_floats, bf_stream_t stream) {
    return attn_bwd_impl(dtype, qkv, dout, dqkv, nseq, L, inner, outer_stride, inner_stride, tok_stride, heads, d, qw, qb, kw, kb, emb, hscale, dqw,
                         dqb, dkw, dkb, demb, dhscale, out_scale, accumulate, ws, ws_floats, nullptr, stream);
}
// Same, but the parameter-gradient rows the MFMA kernel leaves in ws are NOT reduced: *rows = number of rows for a later
// AttnReduceJob (0 when the kernel accumulated the parameter gradients itself: fp32 / generic path, or no workspace).
int bf_attn_bwd_partials(int dtype, const void* qkv, const void* dout, void* dqkv, int64_t nseq, int L, int64_t inner, int64_t outer_stride,
                         int64_t inner_stride, int64_t tok_stride, int heads, int d, const float* qw, const float* qb, const float* kw,
                         const float* kb, const float* emb, const float* hscale, float* dqw, float* dqb, float* dkw, float* dkb, float* demb,
                         float* dhscale, float out_scale, int accumulate, float* ws, int64_t ws_floats, int* rows, hipStream_t stream) {
    return attn_bwd_impl(dtype, qkv, dout, dqkv, nseq, L, inner, outer_stride, inner_stride, tok_stride, heads, d, qw, qb, kw, kb, emb, hscale, dqw,
                         dqb, dkw, dkb, demb, dhscale, out_scale, accumulate, ws, ws_floats, rows, (bf_stream_t)stream);
}
