// InstanceNorm (per frame, per channel over the frame's tokens) statistics, backward, and the
// element-wise affine / residual kernels of the token-major FiLMAViT path.
// All statistics and reductions are fp32 whatever the activation dtype; global access is in 16-byte
// chunks (8 bf16 / 4 f32) so a 64-channel slab row is one 128-/256-byte coalesced segment.
#include "bf_common.h"
#include "param_reduce.h"

namespace {

constexpr int NT = 256;
constexpr int CPB = 64;  // channels per block

template <typename T, int CPB_ = CPB, int NT_ = NT> struct Geo {
    static constexpr int CH = Chunk<T>::N;
    static constexpr int LC = CPB_ / CH;   // chunk lanes per row
    static constexpr int RG = NT_ / LC;    // row groups (a power of two: reduce_rows halves it)
};

// block-wide reduction of CH per-thread partials over the RG row groups; result valid in every thread.
// When the chunk lanes tile a wavefront (LC a power of two <= 32) the row groups inside a wave are folded with lane permutes
// (row rotate by 8, then the gfx950 row / half swaps for lanes ^16 and ^32 -- VALU only), and the waves meet once in LDS: two
// barriers per reduction instead of two per tree level.
__device__ __forceinline__ float lane_xor_add(float v, int o) {
    if (o == 8) return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned u = __float_as_uint(v);
    const u2 r = o == 16 ? __builtin_amdgcn_permlane16_swap(u, u, false, false) : __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <typename T, int NV, int CPB_ = CPB, int NT_ = NT>
__device__ __forceinline__ void reduce_rows(float (&v)[NV][Chunk<T>::N], float* sm) {
    constexpr int CH = Chunk<T>::N, LC = Geo<T, CPB_, NT_>::LC, RG = Geo<T, CPB_, NT_>::RG;
    static_assert((RG & (RG - 1)) == 0 && LC * RG == NT_, "row groups must be a power of two and fill the block");
    const int tid = threadIdx.x, lc = tid % LC, rg = tid / LC;
    if constexpr ((LC == 8 || LC == 16 || LC == 32) && NT_ % 64 == 0) {
        constexpr int NW = NT_ / 64;
        const int wave = tid >> 6;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                float x = v[q][j];
                if (LC <= 8) x = lane_xor_add(x, 8);
                if (LC <= 16) x = lane_xor_add(x, 16);
                x = lane_xor_add(x, 32);
                v[q][j] = x;
            }
            __syncthreads();                       // previous use of sm is over
            if ((tid & 63) < LC) {
#pragma unroll
                for (int j = 0; j < CH; ++j) sm[(wave * LC + lc) * CH + j] = v[q][j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += sm[(w * LC + lc) * CH + j];
                v[q][j] = t;
            }
        }
        __syncthreads();
        return;
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CH; ++j) sm[(rg * LC + lc) * CH + j] = v[q][j];
        __syncthreads();
        for (int s = RG / 2; s > 0; s >>= 1) {
            if (rg < s) {
#pragma unroll
                for (int j = 0; j < CH; ++j) sm[(rg * LC + lc) * CH + j] += sm[((rg + s) * LC + lc) * CH + j];
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) v[q][j] = sm[lc * CH + j];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------ stats
// grid: (frames, ceil(C / 64)).  Two-pass (mean, then centred second moment) -- no E[x^2]-E[x]^2.
constexpr int MAXR = 6;   // rows a thread may keep in registers (frames of up to RG * MAXR tokens are read from memory once)

// The next InstanceNorm in line, applied to the rows this kernel has just produced (fused apply only): a stage that ends in
// out = resid + x * sc + sh is followed by a stage that opens with xn = InstanceNorm(out) -- the frame is still in registers, so the second
// statistics and xn cost one more pair of reductions instead of a launch and a read of `out`.  Same rows per thread, same reduction tree,
// same formulae as a separate launch on `out`: bit-identical results.
struct InChain { const float *w, *b; float *mean, *rstd, *sc, *sh; void* xn; };

template <typename T, bool CACHED>
__global__ void __launch_bounds__(NT) in_stats_kernel(const T* __restrict__ x, int S, int C, const float* __restrict__ w,
                                                     const float* __restrict__ b, const float* __restrict__ g, int gdiv,
                                                     const float* __restrict__ gb, float* __restrict__ mean,
                                                     float* __restrict__ rstd, float* __restrict__ sc, float* __restrict__ sh,
                                                     const T* __restrict__ resid, T* __restrict__ out, InChain ch) {
    constexpr int CH = Chunk<T>::N, LC = Geo<T>::LC, RG = Geo<T>::RG;
    __shared__ float sm[NT * CH];
    const int f = blockIdx.x, c0 = blockIdx.y * CPB;
    const int tid = threadIdx.x, lc = tid % LC, rg = tid / LC;
    const int c = c0 + lc * CH;
    const bool cv = c < C;
    const T* xf = x + (long)f * S * C + c;
    float acc[1][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[0][j] = 0.f;
    Chunk<T> keep[CACHED ? MAXR : 1];
    if constexpr (CACHED) {
#pragma unroll
        for (int q = 0; q < MAXR; ++q) {
            const int s = rg + RG * q;
            if (cv && s < S) {
                keep[q].load(xf + (long)s * C);
#pragma unroll
                for (int j = 0; j < CH; ++j) acc[0][j] += keep[q].get(j);
            }
        }
    } else if (cv) {
        for (int s = rg; s < S; s += RG) {
            Chunk<T> v;
            v.load(xf + (long)s * C);
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[0][j] += v.get(j);
        }
    }
    reduce_rows<T, 1>(acc, sm);
    float mu[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { mu[j] = acc[0][j] / (float)S; acc[0][j] = 0.f; }
    if constexpr (CACHED) {
#pragma unroll
        for (int q = 0; q < MAXR; ++q) {
            if (cv && rg + RG * q < S) {
#pragma unroll
                for (int j = 0; j < CH; ++j) { const float d = keep[q].get(j) - mu[j]; acc[0][j] = fmaf(d, d, acc[0][j]); }      // explicit fma: frame_fwd.hip sums the same way, bit for bit
            }
        }
    } else if (cv) {
        for (int s = rg; s < S; s += RG) {
            Chunk<T> v;
            v.load(xf + (long)s * C);
#pragma unroll
            for (int j = 0; j < CH; ++j) { const float d = v.get(j) - mu[j]; acc[0][j] = fmaf(d, d, acc[0][j]); }
        }
    }
    reduce_rows<T, 1>(acc, sm);
    float aa[CH], ss[CH];
    if (cv && (rg == 0 || (CACHED && out))) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float r = rsqrtf(acc[0][j] / (float)S + BF_IN_EPS);
            const long o = (long)f * C + c + j;
            float a = r * w[c + j];
            float s0 = fmaf(-mu[j], a, b[c + j]);
            if (g) {                      // optional per-(frame group, channel) post scale/shift (FiLM, layer scale)
                const long gi = (long)(f / gdiv) * C + c + j;
                const float gg = g[gi];
                a *= gg;
                s0 = fmaf(s0, gg, gb ? gb[gi] : 0.f);
            }
            aa[j] = a; ss[j] = s0;
            if (rg == 0) { mean[o] = mu[j]; rstd[o] = r; sc[o] = a; sh[o] = s0; }
        }
    }
    if constexpr (CACHED) {
        if (out && cv) {          // fused apply: out = resid + x * sc + sh from the rows still in registers (one launch and one read of x less)
#pragma unroll
            for (int q = 0; q < MAXR; ++q) {
                const int s = rg + RG * q;
                if (s < S) {
                    const long off = ((long)f * S + s) * C + c;
                    Chunk<T> rr, oo;
                    if (resid) rr.load(resid + off);
#pragma unroll
                    for (int j = 0; j < CH; ++j) oo.set(j, fmaf(keep[q].get(j), aa[j], ss[j]) + (resid ? rr.get(j) : 0.f));
                    oo.store(out + off);
                    keep[q] = oo;                   // (chain) the rows as stored
                }
            }
        }
        if (out && ch.xn) {       // block-uniform: every thread takes part in the reductions
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[0][j] = 0.f;
#pragma unroll
            for (int q = 0; q < MAXR; ++q) {
                if (cv && rg + RG * q < S) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) acc[0][j] += keep[q].get(j);
                }
            }
            reduce_rows<T, 1>(acc, sm);
            float mu2[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) { mu2[j] = acc[0][j] / (float)S; acc[0][j] = 0.f; }
#pragma unroll
            for (int q = 0; q < MAXR; ++q) {
                if (cv && rg + RG * q < S) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) { const float d = keep[q].get(j) - mu2[j]; acc[0][j] = fmaf(d, d, acc[0][j]); }
                }
            }
            reduce_rows<T, 1>(acc, sm);
            if (cv) {
                float a2[CH], s2[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float r = rsqrtf(acc[0][j] / (float)S + BF_IN_EPS);
                    const long o = (long)f * C + c + j;
                    a2[j] = r * ch.w[c + j];
                    s2[j] = fmaf(-mu2[j], a2[j], ch.b[c + j]);
                    if (rg == 0) { ch.mean[o] = mu2[j]; ch.rstd[o] = r; ch.sc[o] = a2[j]; ch.sh[o] = s2[j]; }
                }
#pragma unroll
                for (int q = 0; q < MAXR; ++q) {
                    const int s = rg + RG * q;
                    if (s < S) {
                        Chunk<T> oo;
#pragma unroll
                        for (int j = 0; j < CH; ++j) oo.set(j, fmaf(keep[q].get(j), a2[j], s2[j]) + 0.f);
                        oo.store(reinterpret_cast<T*>(ch.xn) + ((long)f * S + s) * C + c);
                    }
                }
            }
        }
    }
}

// Long frames (S > RG * MAXR tokens: the embed / debed resolutions) are cut into slices of RG * MAXR tokens so that the grid
// is frames x slices x channel blocks instead of frames x channel blocks (which left most CUs idle and every wave with one
// load in flight).  A slice is read ONCE into registers; slices are merged exactly (Chan et al. pairwise update):
//   part[(f * nsl + sl) * C + c] = {slice mean, slice centred second moment}
// CPB_ channels per workgroup of NT_ threads: 64 / 256 by default, 96 / 192 when C is a multiple of 96 (the E/4 = 96-channel
// embed / debed maps: no half-empty second channel block, whole 192-byte rows per row group)
template <typename T, int CPB_, int NT_>
__global__ void __launch_bounds__(NT_) in_stats_slice_kernel(const T* __restrict__ x, int S, int C, int nsl, float* __restrict__ part) {
    constexpr int CH = Chunk<T>::N, LC = Geo<T, CPB_, NT_>::LC, RG = Geo<T, CPB_, NT_>::RG;
    __shared__ float sm[NT_ * CH];
    const int f = blockIdx.x / nsl, sl = blockIdx.x % nsl, c0 = blockIdx.y * CPB_;
    const int tid = threadIdx.x, lc = tid % LC, rg = tid / LC;
    const int c = c0 + lc * CH;
    const bool cv = c < C;
    const int s0 = sl * (RG * MAXR), n = min(S - s0, RG * MAXR);
    const T* xf = x + ((long)f * S + s0) * C + c;
    float acc[1][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[0][j] = 0.f;
    Chunk<T> keep[MAXR];
#pragma unroll
    for (int q = 0; q < MAXR; ++q) {
        const int r = rg + RG * q;
        if (cv && r < n) keep[q].load(xf + (long)r * C); else keep[q].zero();
    }
#pragma unroll
    for (int q = 0; q < MAXR; ++q)
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[0][j] += keep[q].get(j);
    reduce_rows<T, 1, CPB_, NT_>(acc, sm);
    float mu[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { mu[j] = acc[0][j] / (float)n; acc[0][j] = 0.f; }
#pragma unroll
    for (int q = 0; q < MAXR; ++q) {
        if (rg + RG * q < n) {
#pragma unroll
            for (int j = 0; j < CH; ++j) { const float d = keep[q].get(j) - mu[j]; acc[0][j] += d * d; }
        }
    }
    reduce_rows<T, 1, CPB_, NT_>(acc, sm);
    if (cv && rg == 0) {
        float* o = part + (((long)f * nsl + sl) * C + c) * 2;
#pragma unroll
        for (int j = 0; j < CH; ++j) { o[2 * j] = mu[j]; o[2 * j + 1] = acc[0][j]; }
    }
}
// grid (ceil(C/64), frames), 256 threads = 64 channels x 4 slice lanes: merge the slices, then the same outputs as in_stats_kernel
__global__ void __launch_bounds__(NT) in_stats_merge_kernel(const float* __restrict__ part, int frames, int S, int C, int nsl, int rows,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* __restrict__ g, int gdiv, const float* __restrict__ gb,
                                                           float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ sc,
                                                           float* __restrict__ sh) {
    __shared__ float red[2][4][64];
    const int l = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 64 + l, f = blockIdx.y;
    const bool cv = c < C;
    const float2* pp = reinterpret_cast<const float2*>(part) + (long)f * nsl * C + (cv ? c : 0);
    float tot = 0.f;
    if (cv)
        for (int sl = q; sl < nsl; sl += 4) tot += pp[(long)sl * C].x * (float)min(rows, S - sl * rows);
    red[0][q][l] = tot;
    __syncthreads();
    const float mu = (red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l]) / (float)S;
    float m2 = 0.f;
    if (cv)
        for (int sl = q; sl < nsl; sl += 4) {
            const float2 v = pp[(long)sl * C];
            const float dm = v.x - mu;
            m2 += v.y + dm * dm * (float)min(rows, S - sl * rows);
        }
    red[1][q][l] = m2;
    __syncthreads();
    if (q != 0 || !cv) return;
    m2 = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    const long i = (long)f * C + c;
    const float r = rsqrtf(m2 / (float)S + BF_IN_EPS);
    float a = r * w[c];
    float s0 = b[c] - mu * a;
    if (g) {
        const long gi = (long)(f / gdiv) * C + c;
        const float gg = g[gi];
        a *= gg;
        s0 = s0 * gg + (gb ? gb[gi] : 0.f);
    }
    mean[i] = mu; rstd[i] = r; sc[i] = a; sh[i] = s0;
}

// ------------------------------------------------------------------------------------ apply
// out = [resid +] z * sc[f, c] + sh[f, c]      (grid-stride over 16-byte chunks)
template <typename T>
__global__ void __launch_bounds__(NT) affine_apply_kernel(const T* __restrict__ z, const T* __restrict__ resid,
                                                         const float* __restrict__ sc, const float* __restrict__ sh,
                                                         T* __restrict__ out, long nrows, int S, int C) {
    constexpr int CH = Chunk<T>::N;
    const int cpr = C / CH;
    const long total = nrows * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long row = i / cpr;
        const int c = (int)(i % cpr) * CH;
        const long f = row / S;
        Chunk<T> v, r, o;
        v.load(z + row * C + c);
        if (resid) r.load(resid + row * C + c);
        float a[CH], b[CH];
#pragma unroll
        for (int j = 0; j < CH; j += 4) {        // 16-byte aligned: C and c are multiples of the chunk
            const float4 a4 = *reinterpret_cast<const float4*>(sc + f * C + c + j);
            const float4 b4 = sh ? *reinterpret_cast<const float4*>(sh + f * C + c + j) : float4{0.f, 0.f, 0.f, 0.f};
            a[j] = a4.x; a[j + 1] = a4.y; a[j + 2] = a4.z; a[j + 3] = a4.w;
            b[j] = b4.x; b[j + 1] = b4.y; b[j + 2] = b4.z; b[j + 3] = b4.w;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float t = fmaf(v.get(j), a[j], b[j]);
            if (resid) t += r.get(j);
            o.set(j, t);
        }
        o.store(out + row * C + c);
    }
}

// out = z * m[(row / S) / fdiv]: one scalar per frame group (the stochastic-depth factor of a branch gradient)
template <typename T>
__global__ void __launch_bounds__(NT) frame_scale_kernel(const T* __restrict__ z, const float* __restrict__ m, int fdiv, T* __restrict__ out,
                                                        long nrows, int S, int C) {
    constexpr int CH = Chunk<T>::N;
    const int cpr = C / CH;
    const long total = nrows * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long row = i / cpr;
        const float f = m[(row / S) / fdiv];
        Chunk<T> v, o;
        v.load(z + i * CH);
#pragma unroll
        for (int j = 0; j < CH; ++j) o.set(j, v.get(j) * f);
        o.store(out + i * CH);
    }
}

// ------------------------------------------------------------------------------------ backward
// y = act( xhat * w + b ) [* g],  xhat = (x - mean) * rstd, act = identity | GELU.
// Given dy: s1 = sum_s dyn, s2 = sum_s dyn * xhat  (dyn = dy * act'),  per (frame, channel)
//   dx = rstd * w * g * (dyn - s1/S - xhat * s2/S) [+ add]
//   dw += g * s2, db += g * s1, dg += w * s2 + b * s1, dgb += s1     (fp32 atomics over frames)
template <typename T, bool GELU, bool CACHED>
__global__ void __launch_bounds__(NT) in_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ add,
                                                   T* __restrict__ dx, int S, int C, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, const float* __restrict__ w,
                                                   const float* __restrict__ b, const float* __restrict__ g, int gdiv,
                                                   float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dg,
                                                   float* __restrict__ dgb, float* __restrict__ ws) {
    constexpr int CH = Chunk<T>::N, LC = Geo<T>::LC, RG = Geo<T>::RG;
    __shared__ float sm[NT * CH];
    const int f = blockIdx.x, c0 = blockIdx.y * CPB;
    const int tid = threadIdx.x, lc = tid % LC, rg = tid / LC;
    const int c = c0 + lc * CH;
    const bool cv = c < C;
    const long base = (long)f * S * C + c;
    float mu[CH], rs[CH], ww[CH], bb[CH], gg[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        mu[j] = cv ? mean[(long)f * C + c + j] : 0.f;
        rs[j] = cv ? rstd[(long)f * C + c + j] : 0.f;
        ww[j] = cv ? w[c + j] : 0.f;
        bb[j] = cv ? b[c + j] : 0.f;
        gg[j] = (cv && g) ? g[(long)(f / gdiv) * C + c + j] : 1.f;
    }
    float acc[2][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[0][j] = acc[1][j] = 0.f;
    // a (frame, 64-channel) slab of up to RG * MAXR tokens is held in registers between the reduction and the apply pass
    Chunk<T> kd[CACHED ? MAXR : 1], kx[CACHED ? MAXR : 1];
    auto accumulate = [&](const Chunk<T>& d, const Chunk<T>& v) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float xh = (v.get(j) - mu[j]) * rs[j];
            float dd = d.get(j);
            if (GELU) dd *= dgelu_t<T>(xh * ww[j] + bb[j]);
            acc[0][j] += dd;
            acc[1][j] += dd * xh;
        }
    };
    auto apply = [&](const Chunk<T>& d, const Chunk<T>& v, long off) {
        Chunk<T> a, o;
        if (add) a.load(add + off);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float xh = (v.get(j) - mu[j]) * rs[j];
            float dd = d.get(j);
            if (GELU) dd *= dgelu_t<T>(xh * ww[j] + bb[j]);
            float t = rs[j] * ww[j] * gg[j] * (dd - (acc[0][j] + xh * acc[1][j]) / (float)S);
            if (add) t += a.get(j);
            o.set(j, t);
        }
        o.store(dx + off);
    };
    if constexpr (CACHED) {
#pragma unroll
        for (int q = 0; q < MAXR; ++q) {
            const int s = rg + RG * q;
            if (cv && s < S) { kd[q].load(dy + base + (long)s * C); kx[q].load(x + base + (long)s * C); accumulate(kd[q], kx[q]); }
        }
    } else if (cv) {
        for (int s = rg; s < S; s += RG) {
            Chunk<T> d, v;
            d.load(dy + base + (long)s * C);
            v.load(x + base + (long)s * C);
            accumulate(d, v);
        }
    }
    reduce_rows<T, 2>(acc, sm);
    if constexpr (CACHED) {
#pragma unroll
        for (int q = 0; q < MAXR; ++q) {
            const int s = rg + RG * q;
            if (cv && s < S) apply(kd[q], kx[q], base + (long)s * C);
        }
    } else if (cv) {
        for (int s = rg; s < S; s += RG) {
            Chunk<T> d, v;
            d.load(dy + base + (long)s * C);
            v.load(x + base + (long)s * C);
            apply(d, v, base + (long)s * C);
        }
    }
    if (cv && rg == 0 && ws) {          // per-frame partials; in_param_reduce_kernel sums them (no same-address atomics)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            ws[((long)f * C + c + j) * 2] = acc[0][j];
            ws[((long)f * C + c + j) * 2 + 1] = acc[1][j];
        }
    } else if (cv && rg == 0) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (dw) atomicAdd(dw + c + j, gg[j] * acc[1][j]);
            if (db) atomicAdd(db + c + j, gg[j] * acc[0][j]);
            if (dg) atomicAdd(dg + (long)(f / gdiv) * C + c + j, ww[j] * acc[1][j] + bb[j] * acc[0][j]);
            if (dgb) atomicAdd(dgb + (long)(f / gdiv) * C + c + j, acc[0][j]);
        }
    }
}

// Long frames, as for the statistics: phase 0 reduces {s1, s2} per slice into part[(f * nsl + sl) * C + c], in_slice_sum_kernel
// adds the slices into ws[f][c], phase 1 applies over the same slices (grid = frames * slices x channel blocks).
// A slice is BREP batches of RG * MAXR rows: the per-channel constants (mean, rstd, w, b, totals, scale: ~7 x CH loads per thread)
// are fetched once per slice, which matters because a batch is only MAXR chunks of payload per thread.
constexpr int BREP = 4;
template <typename T, bool GELU, int PHASE, int CPB_, int NT_>
__global__ void __launch_bounds__(NT_) in_bwd_slice_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ add,
                                                         T* __restrict__ dx, int S, int C, int nsl, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ w,
                                                         const float* __restrict__ b, const float* __restrict__ g, int gdiv,
                                                         float* __restrict__ part, const float* __restrict__ tot) {
    constexpr int CH = Chunk<T>::N, LC = Geo<T, CPB_, NT_>::LC, RG = Geo<T, CPB_, NT_>::RG;
    __shared__ float sm[NT_ * CH];
    const int f = blockIdx.x / nsl, sl = blockIdx.x % nsl, c0 = blockIdx.y * CPB_;
    const int tid = threadIdx.x, lc = tid % LC, rg = tid / LC;
    const int c = c0 + lc * CH;
    const bool cv = c < C;
    const int s0 = sl * (RG * MAXR * BREP), n = min(S - s0, RG * MAXR * BREP);
    const long base = ((long)f * S + s0) * C + c;
    float mu[CH], rs[CH], ww[CH], bb[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        mu[j] = cv ? mean[(long)f * C + c + j] : 0.f;
        rs[j] = cv ? rstd[(long)f * C + c + j] : 0.f;
        ww[j] = cv ? w[c + j] : 0.f;
        bb[j] = cv ? b[c + j] : 0.f;
    }
    float acc[2][CH], t1[CH], t2[CH], gg[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        acc[0][j] = acc[1][j] = 0.f;
        t1[j] = (PHASE == 1 && cv) ? tot[((long)f * C + c + j) * 2] / (float)S : 0.f;
        t2[j] = (PHASE == 1 && cv) ? tot[((long)f * C + c + j) * 2 + 1] / (float)S : 0.f;
        gg[j] = (PHASE == 1 && cv && g) ? g[(long)(f / gdiv) * C + c + j] : 1.f;
    }
    for (int it = 0; it < BREP; ++it) {
        const int r0 = it * (RG * MAXR);
        if (r0 >= n) break;
        Chunk<T> kd[MAXR], kx[MAXR], ka[MAXR];
#pragma unroll
        for (int q = 0; q < MAXR; ++q) {            // all loads of the batch in flight before the first use
            const int r = r0 + rg + RG * q;
            if (cv && r < n) {
                kd[q].load(dy + base + (long)r * C);
                kx[q].load(x + base + (long)r * C);
                if (PHASE == 1 && add) ka[q].load(add + base + (long)r * C);
            } else { kd[q].zero(); kx[q].zero(); }
        }
#pragma unroll
        for (int q = 0; q < MAXR; ++q) {
            const int r = r0 + rg + RG * q;
            if (cv && r < n) {
                Chunk<T> o;
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float xh = (kx[q].get(j) - mu[j]) * rs[j];
                    float dd = kd[q].get(j);
                    if (GELU) dd *= dgelu_t<T>(xh * ww[j] + bb[j]);
                    if (PHASE == 0) { acc[0][j] += dd; acc[1][j] += dd * xh; }
                    else {
                        float t = rs[j] * ww[j] * gg[j] * (dd - t1[j] - xh * t2[j]);
                        if (add) t += ka[q].get(j);
                        o.set(j, t);
                    }
                }
                if (PHASE == 1) o.store(dx + base + (long)r * C);
            }
        }
    }
    if constexpr (PHASE == 0) {
        reduce_rows<T, 2, CPB_, NT_>(acc, sm);
        if (cv && rg == 0) {
            float* o = part + (((long)f * nsl + sl) * C + c) * 2;
#pragma unroll
            for (int j = 0; j < CH; ++j) { o[2 * j] = acc[0][j]; o[2 * j + 1] = acc[1][j]; }
        }
    }
}
// grid (ceil(C/64), frames), 256 threads = 64 channels x 4 slice lanes
__global__ void __launch_bounds__(NT) in_slice_sum_kernel(const float* __restrict__ part, int C, int nsl, float* __restrict__ tot) {
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
    reinterpret_cast<float2*>(tot)[(long)f * C + c] = make_float2(red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l],
                                                                 red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l]);
}

// parameter gradients from the per-frame partials ws[f][c] = {s1, s2}: see param_reduce.h
__global__ void __launch_bounds__(64 * BF_RED_FL) in_param_reduce_kernel(InReduceJob j) {
    __shared__ float red[5][BF_RED_FL][64];
    in_reduce_block(j, blockIdx.x, blockIdx.y, red);
}

// ------------------------------------------------------------------------------------ column sums
// out[c] += scale[c] * sum_rows x[row][c]      grid: (row blocks, ceil(C/64))
template <typename T>
__global__ void __launch_bounds__(NT) colsum_kernel(const T* __restrict__ x, long nrows, int C, long rows_per_block,
                                                   const float* __restrict__ scale, float* __restrict__ out) {
    constexpr int CH = Chunk<T>::N, LC = Geo<T>::LC, RG = Geo<T>::RG;
    __shared__ float sm[NT * CH];
    const int tid = threadIdx.x, lc = tid % LC, rg = tid / LC;
    const int c = blockIdx.y * CPB + lc * CH;
    const bool cv = c < C;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(nrows, r0 + rows_per_block);
    float acc[1][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[0][j] = 0.f;
    if (cv)
        for (long r = r0 + rg; r < r1; r += RG) {
            Chunk<T> v;
            v.load(x + r * C + c);
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[0][j] += v.get(j);
        }
    reduce_rows<T, 1>(acc, sm);
    if (cv && rg == 0) {
#pragma unroll
        for (int j = 0; j < CH; ++j) atomicAdd(out + c + j, acc[0][j] * (scale ? scale[c + j] : 1.f));
    }
}

template <typename T>
int chunk_ok(int C) { return C % Chunk<T>::N == 0; }

}  // namespace

// frames longer than the register cache of the one-workgroup-per-frame kernels are cut into slices; multiples of 96 channels take
// the 96-channel x 192-thread geometry (whole 192-byte rows), everything else 64 x 256
constexpr int WCPB = 96, WNT = 192;
struct SliceCfg { bool sliced, wide; int rows; };
SliceCfg slice_cfg(int dtype, int S, int C) {
    const bool bf = dtype == BF_DTYPE_BF16;
    const int cached = bf ? Geo<bf16>::RG * MAXR : Geo<float>::RG * MAXR;
    SliceCfg c;
    c.sliced = S > cached;
    c.wide = C % WCPB == 0;
    c.rows = c.wide ? (bf ? Geo<bf16, WCPB, WNT>::RG : Geo<float, WCPB, WNT>::RG) * MAXR : cached;
    return c;
}

extern "C" int64_t bf_in_ws_floats(int dtype, int frames, int S, int C) {
    const SliceCfg c = slice_cfg(dtype, S, C);
    const int64_t fc2 = (int64_t)2 * frames * C;
    return c.sliced ? fc2 * (1 + bf_cdiv(S, c.rows)) : fc2;
}

// statistics, optionally followed by out = resid + x * sc + sh in the same kernel (short frames); *applied tells the caller
static int in_stats_impl(int dtype, const void* x, int frames, int S, int C, const float* w, const float* b,
                         const float* g, int gdiv, const float* gb, float* mean, float* rstd, float* sc, float* sh,
                         float* ws, const void* resid, void* out, bool* applied, bf_stream_t stream, const InChain* chain = nullptr,
                         bool* chained = nullptr) {
    if (chained) *chained = false;
    if (applied) *applied = false;
    BF_REQUIRE(x && w && b && mean && rstd && sc && sh, "bf_in_stats: null pointer");
    BF_REQUIRE(frames > 0 && S > 0 && C > 0, "bf_in_stats: empty");
    dim3 grid(frames, bf_cdiv(C, CPB));
    if (gdiv < 1) gdiv = 1;
    BfProfScope prof((hipStream_t)stream, "in_stats", 0.0, (double)frames * S * C * bf_esize(dtype));
    {
        const SliceCfg cfg = slice_cfg(dtype, S, C);
        const int rows = cfg.rows;
        if (ws && cfg.sliced) {                   // long frames: slices + exact merge
            BF_REQUIRE(C % (dtype == BF_DTYPE_BF16 ? 8 : 4) == 0, "bf_in_stats: C must be a multiple of the 16-byte chunk");
            const int nsl = bf_cdiv(S, rows);
            BF_REQUIRE((long)frames * nsl < 2147483647L, "bf_in_stats: grid too large");
            float* part = ws + (size_t)2 * frames * C;
            if (cfg.wide) {
                dim3 sg(frames * nsl, C / WCPB);
                if (dtype == BF_DTYPE_BF16) hipLaunchKernelGGL((in_stats_slice_kernel<bf16, WCPB, WNT>), sg, dim3(WNT), 0, (hipStream_t)stream, (const bf16*)x, S, C, nsl, part);
                else hipLaunchKernelGGL((in_stats_slice_kernel<float, WCPB, WNT>), sg, dim3(WNT), 0, (hipStream_t)stream, (const float*)x, S, C, nsl, part);
            } else {
                dim3 sg(frames * nsl, bf_cdiv(C, CPB));
                if (dtype == BF_DTYPE_BF16) hipLaunchKernelGGL((in_stats_slice_kernel<bf16, CPB, NT>), sg, dim3(NT), 0, (hipStream_t)stream, (const bf16*)x, S, C, nsl, part);
                else hipLaunchKernelGGL((in_stats_slice_kernel<float, CPB, NT>), sg, dim3(NT), 0, (hipStream_t)stream, (const float*)x, S, C, nsl, part);
            }
            BF_CHECK_LAUNCH();
            hipLaunchKernelGGL(in_stats_merge_kernel, dim3(bf_cdiv(C, 64), frames), dim3(NT), 0, (hipStream_t)stream, (const float*)part, frames, S, C,
                               nsl, rows, w, b, g, gdiv, gb, mean, rstd, sc, sh);
            BF_CHECK_LAUNCH();
            return 0;
        }
    }
    const int cached_rows = dtype == BF_DTYPE_BF16 ? Geo<bf16>::RG * MAXR : Geo<float>::RG * MAXR;
    const bool fuse = out != nullptr && S <= cached_rows;      // the apply needs the frame in registers
    const void* RS = fuse ? resid : nullptr;
    void* OU = fuse ? out : nullptr;
    if (applied) *applied = fuse;
    InChain ch{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (chain && fuse && C % CPB == 0) { ch = *chain; if (chained) *chained = true; }      // whole 64-channel blocks only: the chained reductions are block-uniform
    if (dtype == BF_DTYPE_BF16) {
        BF_REQUIRE(chunk_ok<bf16>(C), "bf_in_stats: C must be a multiple of 8 (bf16)");
        if (S <= Geo<bf16>::RG * MAXR) hipLaunchKernelGGL((in_stats_kernel<bf16, true>), grid, dim3(NT), 0, (hipStream_t)stream, (const bf16*)x, S, C, w, b, g, gdiv, gb, mean, rstd, sc, sh, (const bf16*)RS, (bf16*)OU, ch);
        else hipLaunchKernelGGL((in_stats_kernel<bf16, false>), grid, dim3(NT), 0, (hipStream_t)stream, (const bf16*)x, S, C, w, b, g, gdiv, gb, mean, rstd, sc, sh, (const bf16*)RS, (bf16*)OU, ch);
    } else {
        BF_REQUIRE(chunk_ok<float>(C), "bf_in_stats: C must be a multiple of 4 (f32)");
        if (S <= Geo<float>::RG * MAXR) hipLaunchKernelGGL((in_stats_kernel<float, true>), grid, dim3(NT), 0, (hipStream_t)stream, (const float*)x, S, C, w, b, g, gdiv, gb, mean, rstd, sc, sh, (const float*)RS, (float*)OU, ch);
        else hipLaunchKernelGGL((in_stats_kernel<float, false>), grid, dim3(NT), 0, (hipStream_t)stream, (const float*)x, S, C, w, b, g, gdiv, gb, mean, rstd, sc, sh, (const float*)RS, (float*)OU, ch);
    }
    BF_CHECK_LAUNCH();
    return 0;
}

// Statistics from slice partials a producer kernel already wrote (bf_embed_first): ws + 2*frames*C holds {mean, M2} per `rows`-row slice.
// Returns 1 when the workspace bf_in_ws_floats sizes for (S, C) would not hold that many slices.
extern "C" int bf_in_stats_merge_slices(int dtype, int frames, int S, int C, int rows, const float* w, const float* b, const float* g, int gdiv,
                                        const float* gb, float* mean, float* rstd, float* sc, float* sh, float* ws, bf_stream_t stream) {
    BF_REQUIRE(w && b && mean && rstd && sc && sh && ws && frames > 0 && S > 0 && C > 0 && rows > 0, "bf_in_stats_merge_slices: bad arguments");
    const SliceCfg cfg = slice_cfg(dtype, S, C);
    if (!cfg.sliced || bf_cdiv(S, rows) > bf_cdiv(S, cfg.rows)) return 1;
    if (gdiv < 1) gdiv = 1;
    hipLaunchKernelGGL(in_stats_merge_kernel, dim3(bf_cdiv(C, 64), frames), dim3(NT), 0, (hipStream_t)stream, (const float*)(ws + (size_t)2 * frames * C), frames,
                       S, C, bf_cdiv(S, rows), rows, w, b, g, gdiv, gb, mean, rstd, sc, sh);
    BF_CHECK_LAUNCH();
    return 0;
}

extern "C" int bf_in_stats(int dtype, const void* x, int frames, int S, int C, const float* w, const float* b,
                           const float* g, int gdiv, const float* gb, float* mean, float* rstd, float* sc, float* sh,
                           float* ws, bf_stream_t stream) {
    return in_stats_impl(dtype, x, frames, S, C, w, b, g, gdiv, gb, mean, rstd, sc, sh, ws, nullptr, nullptr, nullptr, stream);
}
// statistics of x and out = resid + x * sc + sh: one kernel when a frame fits the register cache, else statistics + bf_affine_apply
int bf_in_stats_apply(int dtype, const void* x, int frames, int S, int C, const float* w, const float* b, const float* g, int gdiv, const float* gb,
                      float* mean, float* rstd, float* sc, float* sh, float* ws, const void* resid, void* out, hipStream_t stream) {
    bool applied = false;
    if (int rc = in_stats_impl(dtype, x, frames, S, C, w, b, g, gdiv, gb, mean, rstd, sc, sh, ws, resid, out, &applied, (bf_stream_t)stream)) return rc;
    if (applied) return 0;
    return bf_affine_apply(dtype, x, resid, sc, sh, out, (int64_t)frames * S, S, C, (bf_stream_t)stream);
}

// ... and, where the fused apply runs, the NEXT InstanceNorm (affine nw / nb) of `out` in the same launch: its statistics and
// nxn = InstanceNorm(out).  *chained tells the caller whether that happened (else it runs the second norm itself).
int bf_in_stats_apply_chain(int dtype, const void* x, int frames, int S, int C, const float* w, const float* b, const float* g, int gdiv, const float* gb,
                            float* mean, float* rstd, float* sc, float* sh, float* ws, const void* resid, void* out, const float* nw, const float* nb,
                            float* nmean, float* nrstd, float* nsc, float* nsh, void* nxn, bool* chained, hipStream_t stream) {
    bool applied = false;
    const InChain ch{nw, nb, nmean, nrstd, nsc, nsh, nxn};
    if (int rc = in_stats_impl(dtype, x, frames, S, C, w, b, g, gdiv, gb, mean, rstd, sc, sh, ws, resid, out, &applied, (bf_stream_t)stream, &ch, chained)) return rc;
    if (applied) return 0;
    return bf_affine_apply(dtype, x, resid, sc, sh, out, (int64_t)frames * S, S, C, (bf_stream_t)stream);
}

extern "C" int bf_affine_apply(int dtype, const void* z, const void* resid, const float* sc, const float* sh, void* out,
                               int64_t nrows, int S, int C, bf_stream_t stream) {
    BF_REQUIRE(z && sc && out && nrows > 0 && S > 0, "bf_affine_apply: bad arguments");
    const int ch = dtype == BF_DTYPE_BF16 ? 8 : 4;
    BF_REQUIRE(C % ch == 0, "bf_affine_apply: C must be a multiple of the 16-byte chunk");
    const long total = nrows * (C / ch);
    BfProfScope prof((hipStream_t)stream, "affine_apply", 0.0, (double)nrows * C * bf_esize(dtype) * (resid ? 3.0 : 2.0));
    const int grid = (int)std::min<long>((total + NT - 1) / NT, 256 * 16);
    if (dtype == BF_DTYPE_BF16)
        hipLaunchKernelGGL(affine_apply_kernel<bf16>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, (const bf16*)z, (const bf16*)resid, sc, sh, (bf16*)out, (long)nrows, S, C);
    else
        hipLaunchKernelGGL(affine_apply_kernel<float>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, (const float*)z, (const float*)resid, sc, sh, (float*)out, (long)nrows, S, C);
    BF_CHECK_LAUNCH();
    return 0;
}

int bf_frame_scale(int dtype, const void* z, const float* m, int fdiv, void* out, long nrows, int S, int C, hipStream_t st) {
    BF_REQUIRE(z && m && out && nrows > 0 && S > 0 && fdiv > 0, "bf_frame_scale: bad arguments");
    const int ch = dtype == BF_DTYPE_BF16 ? 8 : 4;
    BF_REQUIRE(C % ch == 0, "bf_frame_scale: C must be a multiple of the 16-byte chunk");
    const long total = nrows * (C / ch);
    BfProfScope prof(st, "frame_scale", 0.0, (double)nrows * C * bf_esize(dtype) * 2.0);
    const int grid = (int)std::min<long>((total + NT - 1) / NT, 256 * 16);
    if (dtype == BF_DTYPE_BF16) hipLaunchKernelGGL(frame_scale_kernel<bf16>, dim3(grid), dim3(NT), 0, st, (const bf16*)z, m, fdiv, (bf16*)out, nrows, S, C);
    else hipLaunchKernelGGL(frame_scale_kernel<float>, dim3(grid), dim3(NT), 0, st, (const float*)z, m, fdiv, (float*)out, nrows, S, C);
    BF_CHECK_LAUNCH();
    return 0;
}

static int in_bwd_impl(int dtype, const void* dy, const void* x, const void* add, void* dx, int frames, int S, int C,
                       const float* mean, const float* rstd, const float* w, const float* b, const float* g, int gdiv,
                       int gelu, float* dw, float* db, float* dg, float* dgb, float* ws, bool reduce, bf_stream_t stream) {
    BF_REQUIRE(dy && x && dx && mean && rstd && w && b, "bf_in_bwd: null pointer");
    const int ch = dtype == BF_DTYPE_BF16 ? 8 : 4;
    BF_REQUIRE(C % ch == 0, "bf_in_bwd: C must be a multiple of the 16-byte chunk");
    dim3 grid(frames, bf_cdiv(C, CPB));
    if (gdiv < 1) gdiv = 1;
    hipStream_t st = (hipStream_t)stream;
    BfProfScope prof(st, "in_bwd", 0.0, (double)frames * S * C * bf_esize(dtype) * (add ? 4.0 : 3.0));
    const SliceCfg cfg = slice_cfg(dtype, S, C);
    if (ws && cfg.sliced) {                       // long frames: slice reduce -> sum -> slice apply
        const int nsl = bf_cdiv(S, cfg.rows * BREP);
        BF_REQUIRE((long)frames * nsl < 2147483647L, "bf_in_bwd: grid too large");
        float* part = ws + (size_t)2 * frames * C;
#define GOS(T, G, CP, TH)                                                                                                                \
    do {                                                                                                                                  \
        dim3 sg(frames * nsl, bf_cdiv(C, CP));                                                                                            \
        hipLaunchKernelGGL((in_bwd_slice_kernel<T, G, 0, CP, TH>), sg, dim3(TH), 0, st, (const T*)dy, (const T*)x, (const T*)add, (T*)dx, S, C, nsl, mean, rstd, w, b, g, gdiv, part, (const float*)ws); \
        hipLaunchKernelGGL(in_slice_sum_kernel, dim3(bf_cdiv(C, 64), frames), dim3(NT), 0, st, (const float*)part, C, nsl, ws); \
        hipLaunchKernelGGL((in_bwd_slice_kernel<T, G, 1, CP, TH>), sg, dim3(TH), 0, st, (const T*)dy, (const T*)x, (const T*)add, (T*)dx, S, C, nsl, mean, rstd, w, b, g, gdiv, part, (const float*)ws); \
    } while (0)
#define GOS2(T, G) do { if (cfg.wide) GOS(T, G, WCPB, WNT); else GOS(T, G, CPB, NT); } while (0)
        if (dtype == BF_DTYPE_BF16) { if (gelu) GOS2(bf16, true); else GOS2(bf16, false); }
        else { if (gelu) GOS2(float, true); else GOS2(float, false); }
#undef GOS2
#undef GOS
        BF_CHECK_LAUNCH();
    } else {
#define GO(T, G)                                                                                                                         \
    do {                                                                                                                                  \
        if (S <= Geo<T>::RG * MAXR) hipLaunchKernelGGL((in_bwd_kernel<T, G, true>), grid, dim3(NT), 0, st, (const T*)dy, (const T*)x, (const T*)add, (T*)dx, S, C, mean, rstd, w, b, g, gdiv, dw, db, dg, dgb, ws); \
        else hipLaunchKernelGGL((in_bwd_kernel<T, G, false>), grid, dim3(NT), 0, st, (const T*)dy, (const T*)x, (const T*)add, (T*)dx, S, C, mean, rstd, w, b, g, gdiv, dw, db, dg, dgb, ws); \
    } while (0)
    if (dtype == BF_DTYPE_BF16) { if (gelu) GO(bf16, true); else GO(bf16, false); }
    else { if (gelu) GO(float, true); else GO(float, false); }
#undef GO
    BF_CHECK_LAUNCH();
    }
    if (ws && reduce) {
        const InReduceJob j{ws, frames, C, w, b, g, gdiv, dw, db, dg, dgb, nullptr, nullptr};
        hipLaunchKernelGGL(in_param_reduce_kernel, dim3(bf_cdiv(C, 64), bf_cdiv(frames, j.rdiv())), dim3(64 * BF_RED_FL), 0, st, j);
        BF_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int bf_in_bwd(int dtype, const void* dy, const void* x, const void* add, void* dx, int frames, int S, int C,
                         const float* mean, const float* rstd, const float* w, const float* b, const float* g, int gdiv,
                         int gelu, float* dw, float* db, float* dg, float* dgb, float* ws, bf_stream_t stream) {
    return in_bwd_impl(dtype, dy, x, add, dx, frames, S, C, mean, rstd, w, b, g, gdiv, gelu, dw, db, dg, dgb, ws, true, stream);
}
// data gradient only: the per-frame partials stay in ws (required) for a later InReduceJob (model.hip runs a stage's jobs in one launch)
int bf_in_bwd_partials(int dtype, const void* dy, const void* x, const void* add, void* dx, int frames, int S, int C, const float* mean,
                       const float* rstd, const float* w, const float* b, const float* g, int gdiv, int gelu, float* ws, hipStream_t stream) {
    BF_REQUIRE(ws, "bf_in_bwd_partials: workspace required");
    return in_bwd_impl(dtype, dy, x, add, dx, frames, S, C, mean, rstd, w, b, g, gdiv, gelu, nullptr, nullptr, nullptr, nullptr, ws, false, (bf_stream_t)stream);
}

extern "C" int bf_colsum(int dtype, const void* x, int64_t nrows, int C, const float* scale, float* out, bf_stream_t stream) {
    BF_REQUIRE(x && out && nrows > 0 && C > 0, "bf_colsum: bad arguments");
    const int ch = dtype == BF_DTYPE_BF16 ? 8 : 4;
    BF_REQUIRE(C % ch == 0, "bf_colsum: C must be a multiple of the 16-byte chunk");
    const int cb = bf_cdiv(C, CPB);
    long rpb = std::max<long>(64, (nrows * cb + 1023) / 1024);   // ~1024 blocks
    dim3 grid(bf_cdiv(nrows, rpb), cb);
    BfProfScope prof((hipStream_t)stream, "colsum", 0.0, (double)nrows * C * bf_esize(dtype));
    if (dtype == BF_DTYPE_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)stream, (const bf16*)x, (long)nrows, C, rpb, scale, out);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, (const float*)x, (long)nrows, C, rpb, scale, out);
    BF_CHECK_LAUNCH();
    return 0;
}
