// The tail of the patch-embedding backward in one pass (layers/patching.py:24-56, HMLPEmbed: conv 2x2/s2 -> InstanceNorm -> GELU per stage).
//
// Behind the stage-1 data gradient sit, in the reference's autograd order: GELU', the stage-0 InstanceNorm backward and the stage-0
// weight gradient -- and nothing else when the input needs no gradient.  The stage-0 maps are the largest tensors of the whole step
// (F x 96 x 96 x 96 channels: 226 MB in bf16 at the bench shape); the unfused path writes the gradient map, reads it and the
// activation map twice for the InstanceNorm backward (two phases around a frame-wide reduction), writes it again, and reads it a third
// time for the 16-wide weight gradient: 1.7 GB of HBM traffic, 0.5 ms at the very end of the step with nothing left to overlap.
//
// Only SUMS over pixels leave that chain: with dd = dact * gelu'(z) (z = y0 * sc + sh, the forward's own affine),
//     s1[f][c] = sum dd                      -> d in_b,   t1 = s1 / S
//     s2[f][c] = sum dd * xh                 -> d in_w,   t2 = s2 / S          (xh = (y0 - mean) * rstd)
//     dW0[c][k] = sum_f rstd w (G[f][c][k] - t1 P1[f][k] - t2 PX[f][c][k])
//     G = sum dd * patch_k,   P1 = sum patch_k,   PX = sum xh * patch_k = rstd (W0 M2 - mean P1),   M2 = sum patch_j patch_k
// (y0 = W0 . patch, the stage-0 convolution has no bias).  So the data-gradient GEMM keeps its output tile in registers, multiplies by
// gelu'(z) there and contracts it with the patch rows on the matrix cores: the gradient map is never written, the activation map is
// read once, and what reaches memory is 8 KB of partial sums per workgroup.
//
// Kernel: a workgroup = 4 waves on one (frame, 2x2 position q, run of rows); a wave owns 32-row tiles of the stage-1 grid and all
// C0 = 96 columns of its q.  dy1 rows come straight from global memory as MFMA A operands (16-byte loads, prefetched one tile ahead),
// the [C1][C0] slice of the stage-1 weight sits in LDS for the whole run (B operands by transposing reads), the y0 / patch rows of the
// tile pass through a wave-private LDS tile so that the transposing read returns them in the accumulator layout (lane = channel,
// four consecutive pixels) -- which is also the MFMA A-operand layout of the pixel contraction, so dd goes from the accumulators
// into the next MFMA without leaving the lane.  Waves never meet between the weight load and the final reduction.
#include "bf_common.h"
#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// sum over the lane quad {l, l+16, l+32, l+48} (v_permlane16_swap / v_permlane32_swap), in every lane
__device__ __forceinline__ float quad_sum(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// transposing read of a 4-row x 16-col block of a bf16 LDS tile: lane i16 of the 16-lane group gets column c0 + i16 of rows r0..r0+3
__device__ __forceinline__ s16x4 tr4(const bf16* tile, int ld, int r0, int c0, int lane) {
    const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tile + (r0 + q) * ld + c0 + 4 * p));
}
__device__ __forceinline__ bf16x8 cat(s16x4 lo, s16x4 hi) {
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ float bf_bits_f(short b) { return __uint_as_float(((unsigned)(unsigned short)b) << 16); }
__device__ __forceinline__ short f_bf_bits(float x) { return __builtin_bit_cast(short, (bf16)x); }

struct TailArgs {
    const bf16 *dy1, *w1, *y0, *patches, *w0;     // y0 may be null: the stage-0 rows are then rebuilt from the patch rows (y0 = W0 . patch)
    const float *sc, *sh, *mean, *rstd;
    float* part;
    int F, gh1, gw1, tpw, cpf;        // stage-1 grid, 32-row tiles per wave, row runs per frame
};

constexpr int TWAVES = 4;
template <int NCB> constexpr int tail_npart() { return 16 * NCB * 16 + 256 + 16 + 2 * 16 * NCB; }      // G | M2 | P1 | s1 | s2
// the weight slice + per wave {y0 tile (YMAP only), patch tile}; the final reduction (4 rows of partials) reuses the region
template <int NCB, int NKS, bool YMAP> constexpr int tail_tiles_bytes() {
    const int tiles = (32 * NKS * (16 * NCB + 8) + TWAVES * ((YMAP ? 32 * (16 * NCB + 8) : 0) + 32 * 24)) * 2, red = TWAVES * tail_npart<NCB>() * 4;
    return tiles > red ? tiles : red;
}
template <int NCB, int NKS, bool YMAP> constexpr int tail_lds_bytes() { return tail_tiles_bytes<NCB, NKS, YMAP>() + 16 * NCB * 16; }

template <int NCB, int NKS, bool YMAP>
__global__ void __launch_bounds__(64 * TWAVES, 2) embed_tail_bwd_kernel(TailArgs a) {
    constexpr int C0 = 16 * NCB, C1 = 32 * NKS, LDW = C0 + 8, LDY = C0 + 8, LDP = 24, CPR = C0 / 8;
    constexpr int NPART = tail_npart<NCB>();
    extern __shared__ __attribute__((aligned(16))) char smem_tail[];
    bf16* Wt = reinterpret_cast<bf16*>(smem_tail);                                   // [C1][LDW]: this q's columns of the stage-1 weight
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16* ytile = Wt + C1 * LDW + wave * ((YMAP ? 32 * LDY : 0) + 32 * LDP);         // [32][LDY] raw stage-0 rows of the tile (YMAP)
    bf16* ptile = ytile + (YMAP ? 32 * LDY : 0);                                     // [32][LDP] their patch rows
    const int i16 = lane & 15, g = lane >> 4;
    // (run, q) with q fastest; workgroups of one XCD (blockIdx % 8 under round-robin placement; speed only) take consecutive ones, so
    // the four q of a run read their dy1 rows from the same L2
    const int nwg = gridDim.x;
    const int logical = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
    const int q = logical & 3, unit = logical >> 2;
    const int f = unit / a.cpf, run = unit - f * a.cpf;
    const int qy = q >> 1, qx = q & 1;
    const unsigned gw1 = (unsigned)a.gw1;
    const long S1 = (long)a.gh1 * a.gw1;
    for (int t = tid; t < C1 * CPR; t += 64 * TWAVES) {
        const int k = t / CPR, ch = t - k * CPR;
        *reinterpret_cast<bf16x8*>(Wt + k * LDW + 8 * ch) = *reinterpret_cast<const bf16x8*>(a.w1 + (long)k * (4 * C0) + q * C0 + 8 * ch);
    }
    // per-(frame, channel) constants {sc, sh, rstd, -mean * rstd}: one 16-byte LDS read per channel block and tile instead of 4 NCB live registers
    float4* cst = reinterpret_cast<float4*>(smem_tail + tail_tiles_bytes<NCB, NKS, YMAP>());
    for (int c = tid; c < C0; c += 64 * TWAVES) {
        const long o = (long)f * C0 + c;
        const float r = a.rstd[o];
        cst[c] = make_float4(a.sc[o], a.sh[o], r, -a.mean[o] * r);
    }
    __syncthreads();

    // where this lane's 16-byte pieces of the tile sit: y0 piece j -> row (lane >> 2) + 16 (j / (NCB / 2)), 8-channel group
    // 4 (j % (NCB / 2)) + (lane & 3) -- 64-byte runs per row, and every offset is one lane base plus a constant; patch piece -> (row, half)
    static_assert(NCB % 2 == 0, "a row is NCB / 2 pieces of 4 lanes");
    const int yl0 = (lane >> 2) * LDY + 8 * (lane & 3), yg0 = 2 * (lane >> 2) * C0 + 8 * (lane & 3);
    const int prow = lane >> 1, ph = lane & 1;
    const int plds = prow * LDP + 8 * ph, poff = 2 * (prow & 15) * 16 + 8 * ph;
    const bool phi = prow >= 16;

    f32x4 accG[NCB], accM2 = {0.f, 0.f, 0.f, 0.f}, accP1 = {0.f, 0.f, 0.f, 0.f};
    float s1[NCB], s2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { accG[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; s1[cb] = s2[cb] = 0.f; }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;

    // !YMAP: y0[pixel][c] = sum_k patch[pixel][k] W0[c][k] per tile, one v_mfma_f32_16x16x16_bf16 per (row block, channel block): A = the
    // patch rows (lane = pixel, k = 4g..4g+3: an 8-byte load), B = this lane's row of W0 -- the result lands in the accumulator layout
    s16x4 w0f[YMAP ? 1 : NCB], pa[2];
    if (!YMAP) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) w0f[cb] = *reinterpret_cast<const s16x4*>(a.w0 + (16 * cb + i16) * 16 + 4 * g);
    }
    bf16x8 af[2][NKS];
    u32x4 yr[YMAP ? NCB : 1], pr;
    // a 16-row block of the stage-1 grid lies inside one image row (gw1 % 16 == 0): its stage-0 pixels of position q are 2 apart
    auto pixel0 = [&](int tt, int rb) __attribute__((always_inline)) -> long {
        const unsigned pl = 32u * (unsigned)tt + 16u * (unsigned)rb, y = pl / gw1, x0 = pl - y * gw1;
        return ((long)f * (2 * a.gh1) + 2 * y + qy) * (2L * a.gw1) + 2 * x0 + qx;
    };
    auto issue_rows = [&](int tt) __attribute__((always_inline)) {
        const long p0 = pixel0(tt, 0), p1 = pixel0(tt, 1);
        if (YMAP) {
#pragma unroll
            for (int j = 0; j < NCB; ++j) yr[j] = *reinterpret_cast<const u32x4*>(a.y0 + (j < NCB / 2 ? p0 : p1) * C0 + yg0 + 32 * (j % (NCB / 2)));
        } else {
            pa[0] = *reinterpret_cast<const s16x4*>(a.patches + (p0 + 2 * i16) * 16 + 4 * g);
            pa[1] = *reinterpret_cast<const s16x4*>(a.patches + (p1 + 2 * i16) * 16 + 4 * g);
        }
        pr = *reinterpret_cast<const u32x4*>(a.patches + (phi ? p1 : p0) * 16 + poff);
    };
    auto issue_a = [&](int tt) __attribute__((always_inline)) {
        const bf16* base = a.dy1 + ((long)f * S1 + 32L * tt + i16) * C1 + 8 * g;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int s = 0; s < NKS; ++s) af[rb][s] = *reinterpret_cast<const bf16x8*>(base + (long)rb * 16 * C1 + 32 * s);
    };
    const int tt0 = (run * TWAVES + wave) * a.tpw;
    issue_rows(tt0);
    issue_a(tt0);
    for (int t = 0; t < a.tpw; ++t) {
        const int tt = tt0 + t;
        if (YMAP) {
#pragma unroll
            for (int j = 0; j < NCB; ++j) *reinterpret_cast<u32x4*>(ytile + yl0 + (j / (NCB / 2)) * 16 * LDY + 32 * (j % (NCB / 2))) = yr[j];
        }
        *reinterpret_cast<u32x4*>(ptile + plds) = pr;
        const s16x4 pa0 = pa[0], pa1 = pa[1];
        wsync();
        if (t + 1 < a.tpw) issue_rows(tt + 1);            // in flight under the products and the epilogue below
        // ---- dact[pixel][c] = sum_co dy1[pixel][co] W1[co][(q, c)]: lane (c = 16 cb + i16) holds pixels 16 rb + 4 g + r
        f32x4 acc[2][NCB];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKS; ++s)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const bf16x8 bw = cat(tr4(Wt, LDW, 32 * s + 8 * g, 16 * cb, lane), tr4(Wt, LDW, 32 * s + 8 * g + 4, 16 * cb, lane));
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rb][s], bw, acc[rb][cb], 0, 0, 0);
            }
        if (t + 1 < a.tpw) issue_a(tt + 1);
        // ---- epilogue.  MFMA k-slot (g, jj) := pixel 4g + jj of block 0 (jj < 4) / pixel 4g + jj - 4 of block 1: the accumulators of
        // the two row blocks ARE the A operand of the pixel contraction, and the transposing read of the patch tile is its B operand
        const bf16x8 pf = cat(tr4(ptile, LDP, 4 * g, 0, lane), tr4(ptile, LDP, 16 + 4 * g, 0, lane));
        accM2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, pf, accM2, 0, 0, 0);
        accP1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, accP1, 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float yy[2][4];
            if (YMAP) {
                const s16x4 y0v = tr4(ytile, LDY, 4 * g, 16 * cb, lane), y1v = tr4(ytile, LDY, 16 + 4 * g, 16 * cb, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) { yy[0][r] = bf_bits_f(y0v[r]); yy[1][r] = bf_bits_f(y1v[r]); }
            } else {      // rounded to bf16 like the rows the forward normalised
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                const f32x4 m0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa0, w0f[cb], z4, 0, 0, 0), m1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa1, w0f[cb], z4, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { yy[0][r] = (float)(bf16)m0[r]; yy[1][r] = (float)(bf16)m1[r]; }
            }
            const float4 k4 = cst[16 * cb + i16];
            s16x8 ddp;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = yy[rb][r];
                    const float dd = acc[rb][cb][r] * dgelu_fast(fmaf(y, k4.x, k4.y));
                    s1[cb] += dd;
                    s2[cb] = fmaf(dd, fmaf(y, k4.z, k4.w), s2[cb]);
                    ddp[4 * rb + r] = f_bf_bits(dd);
                }
            accG[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ddp), pf, accG[cb], 0, 0, 0);
        }
        wsync();                                          // the tile's reads are done before the next tile's rows overwrite it
    }
    // ---- the four waves' sums -> one row of partials per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_tail) + wave * NPART;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(16 * cb + 4 * g + r) * 16 + i16] = accG[cb][r];       // accG: lane (k = i16) holds channels 16 cb + 4 g + r
        const float t1 = quad_sum(s1[cb]), t2 = quad_sum(s2[cb]);
        if (g == 0) { red[C0 * 16 + 272 + 16 * cb + i16] = t1; red[C0 * 16 + 272 + C0 + 16 * cb + i16] = t2; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[C0 * 16 + (4 * g + r) * 16 + i16] = accM2[r];
    if (g == 0) red[C0 * 16 + 256 + i16] = accP1[0];
    __syncthreads();
    const float* all = reinterpret_cast<const float*>(smem_tail);
    float* dst = a.part + (long)logical * NPART;
    for (int e = tid; e < NPART; e += 64 * TWAVES) dst[e] = (all[e] + all[NPART + e]) + (all[2 * NPART + e] + all[3 * NPART + e]);
}

// One frame: its workgroups' partials summed in a fixed order, then the frame's term of the stage-0 weight gradient (see the header)
template <int NCB>
__global__ void __launch_bounds__(256) embed_tail_frame_kernel(const float* __restrict__ part, int nper, const bf16* __restrict__ w0c, int Kp,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ in_w, float invS0, float* __restrict__ slab,
                                                               float* __restrict__ sums) {
    constexpr int C0 = 16 * NCB, NPART = tail_npart<NCB>();
    __shared__ float sm[NPART];
    const int f = blockIdx.x;
    const float* src = part + (long)f * nper * NPART;
    for (int e = threadIdx.x; e < NPART; e += 256) {      // up to 16 partial rows in flight (a load-then-add loop pays a round trip per row)
        float v = 0.f;
        for (int b0 = 0; b0 < nper; b0 += 16) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = b0 + u < nper ? src[(long)(b0 + u) * NPART + e] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) v += t[u];
        }
        sm[e] = v;
    }
    __syncthreads();
    const float* G = sm;
    const float* M2 = sm + C0 * 16;
    const float* P1 = M2 + 256;
    const float* S1v = P1 + 16;
    const float* S2v = S1v + C0;
    for (int e = threadIdx.x; e < C0 * 16; e += 256) {
        const int c = e >> 4, k = e & 15;
        const float rs = rstd[(long)f * C0 + c], mu = mean[(long)f * C0 + c];
        float px = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) px = fmaf((float)w0c[(long)c * Kp + j], M2[j * 16 + k], px);
        px = rs * (px - mu * P1[k]);
        slab[(long)f * C0 * 16 + e] = rs * in_w[c] * (G[e] - S1v[c] * invS0 * P1[k] - S2v[c] * invS0 * px);
    }
    for (int c = threadIdx.x; c < 2 * C0; c += 256) sums[(long)f * 2 * C0 + c] = S1v[c];      // s1 | s2 are adjacent
}
// frames summed in order: dW0 (prepared layout [C0][Kp]) is written, d in_b / d in_w are accumulated
__global__ void __launch_bounds__(256) embed_tail_sum_kernel(const float* __restrict__ slab, const float* __restrict__ sums, int F, int C0, int Kp,
                                                             float* __restrict__ dwprep, float* __restrict__ d_in_w, float* __restrict__ d_in_b) {
    const int e = blockIdx.x * 256 + threadIdx.x, n = C0 * 16;
    if (e >= n + 2 * C0) return;
    const float* src = e < n ? slab + e : sums + (e - n);
    const long stride = e < n ? n : 2 * C0;
    float v = 0.f;
    int r = 0;
    for (; r + 16 <= F; r += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = src[(long)(r + u) * stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) v += t[u];
    }
    for (; r < F; ++r) v += src[(long)r * stride];
    if (e < n) dwprep[(long)(e >> 4) * Kp + (e & 15)] = v;
    else if (e - n < C0) { if (d_in_b) d_in_b[e - n] += v; }
    else if (d_in_w) d_in_w[e - n - C0] += v;
}

}  // namespace

// 32-row tiles per wave (at most 6: a workgroup's partials cost 8 KB of traffic, a run of 768 rows amortises them) and runs per frame
static bool tail_plan(long S1, int* tpw, int* cpf) {
    if (S1 <= 0 || S1 % (32 * TWAVES)) return false;
    const int wtiles = (int)(S1 / 32 / TWAVES);
    *tpw = 1;
    for (int t = 6; t >= 1; --t)
        if (wtiles % t == 0) { *tpw = t; break; }
    *cpf = wtiles / *tpw;
    return true;
}

extern "C" int64_t bf_embed_tail_ws_floats(int F, int gh1, int gw1, int C0, int Kp) {
    int tpw, cpf;
    if (F <= 0 || gh1 <= 0 || gw1 <= 0 || C0 != 96 || Kp != 16 || !tail_plan((long)gh1 * gw1, &tpw, &cpf)) return 0;
    return (int64_t)F * cpf * 4 * tail_npart<6>() + (int64_t)F * C0 * 16 + (int64_t)F * 2 * C0;
}

// 0 = done (dwprep = the stage-0 weight gradient in the prepared [C0][Kp] layout, d_in_w / d_in_b accumulated), 1 = shape not covered
extern "C" int bf_embed_tail_bwd(int dtype, const void* dy1, const void* w1c, const void* y0, const void* patches, const void* w0c,
                                 const float* sc, const float* sh, const float* mean, const float* rstd, const float* in_w, float* dwprep,
                                 float* d_in_w, float* d_in_b, int F, int gh1, int gw1, int C1, int C0, int Kp, float* ws, int64_t ws_floats,
                                 bf_stream_t stream) {
    if (dtype != BF_DTYPE_BF16 || C0 != 96 || (C1 != 96 && C1 != 192) || Kp != 16) return 1;
    if (F <= 0 || gh1 <= 0 || gw1 <= 0 || gw1 % 16) return 1;
    const long S1 = (long)gh1 * gw1;
    int tpw, cpf;
    if (!tail_plan(S1, &tpw, &cpf)) return 1;
    BF_REQUIRE(dy1 && w1c && patches && w0c && sc && sh && mean && rstd && in_w && dwprep && ws, "bf_embed_tail_bwd: null pointer");
    BF_REQUIRE((((uintptr_t)dy1 | (uintptr_t)w1c | (uintptr_t)y0 | (uintptr_t)patches) & 15) == 0, "bf_embed_tail_bwd: operands must be 16-byte aligned");
    const long nwg = (long)F * cpf * 4;
    BF_REQUIRE(nwg < (1L << 30), "bf_embed_tail_bwd: grid too large");
    constexpr int NPART = tail_npart<6>();
    const int64_t need = nwg * NPART + (int64_t)F * C0 * 16 + (int64_t)F * 2 * C0;
    if (ws_floats < need) return 1;
    hipStream_t st = (hipStream_t)stream;
    float* part = ws;
    float* slab = part + nwg * NPART;
    float* sums = slab + (size_t)F * C0 * 16;
    TailArgs a{(const bf16*)dy1, (const bf16*)w1c, (const bf16*)y0, (const bf16*)patches, (const bf16*)w0c, sc, sh, mean, rstd, part, F, gh1, gw1, tpw, cpf};
    {
        const double rows = (double)F * S1;
        BfProfScope prof(st, "embed_tail_bwd", 2.0 * rows * C1 * 4 * C0 + 2.0 * rows * 4 * (C0 + 2) * 16, rows * C1 * 2.0 + rows * 4 * (C0 + 16) * 2.0);
#define BF_TAIL_GO(NKS, YM)                                                                                                                \
        do {                                                                                                                              \
            constexpr int lds = tail_lds_bytes<6, NKS, YM>();                                                                             \
            static BfPerDeviceOnce attr_once; bool& attr_done = attr_once.flag();                                                                                                \
            if (!attr_done) {                                                                                                             \
                hipError_t e_ = hipFuncSetAttribute((const void*)embed_tail_bwd_kernel<6, NKS, YM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
                if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);                                                             \
                attr_done = true;                                                                                                         \
            }                                                                                                                             \
            hipLaunchKernelGGL((embed_tail_bwd_kernel<6, NKS, YM>), dim3((unsigned)nwg), dim3(64 * TWAVES), lds, st, a);                  \
        } while (0)
        if (y0) { if (C1 == 96) BF_TAIL_GO(3, true); else BF_TAIL_GO(6, true); }
        else { if (C1 == 96) BF_TAIL_GO(3, false); else BF_TAIL_GO(6, false); }
#undef BF_TAIL_GO
        BF_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL((embed_tail_frame_kernel<6>), dim3(F), dim3(256), 0, st, (const float*)part, 4 * cpf, (const bf16*)w0c, Kp, mean, rstd, in_w,
                       1.0f / (float)(4 * S1), slab, sums);
    BF_CHECK_LAUNCH();
    hipLaunchKernelGGL(embed_tail_sum_kernel, dim3(bf_cdiv(C0 * 16 + 2 * C0, 256)), dim3(256), 0, st, (const float*)slab, (const float*)sums, F, C0, Kp,
                       dwprep, d_in_w, d_in_b);
    BF_CHECK_LAUNCH();
    return 0;
}
