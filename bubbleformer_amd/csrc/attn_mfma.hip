// bf16 MFMA small-sequence attention (L <= 32, head dim d in {32, 64, 96, 128}) for the factored space-time blocks.
//
// One wavefront owns one (sequence, head) problem; everything but V (forward) / Qn, Kn, dO (backward) stays in
// registers.  v_mfma_f32_16x16x32_bf16 is used in the orientation that leaves the softmax axis (keys) inside a
// lane quad {l, l+16, l+32, l+48}:
//     S^T[j][i]  = sum_e Kn[j][e] Qn[i][e]          A = Kn rows, B = Qn rows (both straight from global, 16-byte loads)
//       -> lane (i = l & 15) holds keys j = 4*(l >> 4) + r, r = 0..3 (per 16-key block)
//     O^T[e][i]  = sum_j V[j][e] A[i][j]            A = V^T via ds_read_b64_tr_b16 on the LDS V tile, B = the P registers as they stand:
//       MFMA k-slot (g, jj) := key 4g + jj (block 0, jj < 4) / key 16 + 4g + jj - 4 (block 1, jj >= 4)
//       -> lane holds 4 consecutive channels of one query: 8-byte stores.
// q/k LayerNorm runs in the operand layout (a row's 64 channels sit in the 4 lanes of a quad: two __shfl_xor).
// Backward recomputes P, forms dA^T the same way (A = V rows, B = dO rows), then
//     dQn^T[e][i] = sum_j Kn[j][e] dS[i][j]         A = Kn^T (tr read), B = dS registers
//     dKn^T[e][j] = sum_i Qn[i][e] dS[i][j]         A = Qn^T (tr read), B = dS^T through a small LDS transpose
//     dV^T[e][j]  = sum_i dO[i][e] A[i][j]          A = dO^T (tr read), B = A^T through the same transpose
// and finishes LayerNorm backward in the operand layout after one LDS re-layout (16-byte global stores).
#include "bf_common.h"
#include "param_reduce.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

// All outstanding vector-memory operations complete, as an s_waitcnt the compiler's wait-count pass models (an inline-asm wait is opaque to
// it).  In front of a loop that prefetches the next problem's rows: the pass merges the loop header's pending-load state from the preheader
// and the back edge, and with the first problem's loads still pending there it counts every use of `cur` at the top of the body against
// them -- vmcnt(7), vmcnt(6), ... right behind the eight NEW loads, i.e. a full memory round trip per problem and no look-ahead at all.
__device__ __forceinline__ void drain_vm() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0), expcnt / lgkmcnt untouched
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int t5b(int n) {   // n = query - key  (see attn.hip)
    const int a = n < 0 ? -n : n;
    int b;
    if (a < 8) b = a; else if (a < 10) b = 8; else if (a < 12) b = 9; else if (a < 14) b = 10; else if (a < 16) b = 11;
    else if (a < 20) b = 12; else if (a < 23) b = 13; else if (a < 27) b = 14; else b = 15;
    return b + (n < 0 ? 16 : 0);
}
// Reductions over a lane quad {l, l+16, l+32, l+48} with the gfx950 row-swap VALU ops (no LDS crossbar round trip):
// v_permlane16_swap(a, b) exchanges the odd 16-lane rows of a with the even rows of b, v_permlane32_swap the upper half of a
// with the lower half of b; with a = b = v the two results are v and its xor-16 / xor-32 partner in every lane.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float quad_sum(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// total of a 16-lane row in every lane of the row: xor-1 / xor-2 quad permutes, then the half-row and row mirrors (DPP, no LDS)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}
__device__ __forceinline__ float quad_max(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

struct Geo { long nseq; int L; long inner, outer_stride, inner_stride, tok_stride; };
// Where problem pr = sequence * heads + head starts.  pr is wave-uniform (the wave index comes through readfirstlane), so this is scalar
// work; 32-bit on purpose: a 64-bit division by a run-time value is a ~60-instruction routine, and the kernels used to run four of them
// per problem in VECTOR registers (the host checks nseq * heads < 2^31).
struct Prob { long tok0; int head; };
__device__ __forceinline__ Prob locate(const Geo& g, int heads, long pr) {
    const unsigned p = (unsigned)pr, hd = (unsigned)heads, inner = (unsigned)g.inner;
    const unsigned s = p / hd, head = p - s * hd;
    const unsigned q = s / inner, r = s - q * inner;
    return Prob{(long)q * g.outer_stride + (long)r * g.inner_stride, (int)head};
}
// Totals over the 16 lanes of a row of sixteen values, value v landing in lane v of the row: four halving exchange steps (mirror,
// half-mirror, xor 2, xor 1 -- one DPP add per surviving value: 8 + 4 + 2 + 1) instead of sixteen full row sums and sixteen selects.
// In each step a lane keeps the half of its values whose index bit matches its own lane bit and receives its partner's copy of it.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float keep, float send) {
    return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_scatter_sum(const float (&p)[16], int i16) {
    // The two wide steps (8 + 4 survivors) as bank-masked DPP adds: lanes whose index bit is 0 take own + partner's copy of the low half of the
    // values, lanes whose bit is 1 the same of the high half -- both halves of a result register written by one v_add_f32_dpp each, no select,
    // no separate v_mov_dpp (the compiler forms neither: bank masks have no builtin).  Bits 3 and 2 of the lane index are bank-aligned (a bank =
    // four lanes); bits 1 and 0 are not, so the two narrow steps (2 + 1 survivors) stay selects.  s_nop: a DPP read of a VGPR a VALU op has
    // just written needs two wait states (five after an EXEC write), and the compiler's hazard pass does not see inside the block.
    float a0, a1, a2, a3, a4, a5, a6, a7, b[4];
#define BF_M1(a, lo, hi) "v_add_f32_dpp " a ", " lo ", " lo " row_mirror row_mask:0xf bank_mask:0x3\n\t" \
                         "v_add_f32_dpp " a ", " hi ", " hi " row_mirror row_mask:0xf bank_mask:0xc\n\t"
#define BF_M2(b, lo, hi) "v_add_f32_dpp " b ", " lo ", " lo " row_half_mirror row_mask:0xf bank_mask:0x5\n\t" \
                         "v_add_f32_dpp " b ", " hi ", " hi " row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
    asm("s_nop 4\n\t"
        BF_M1("%4", "%12", "%20") BF_M1("%5", "%13", "%21") BF_M1("%6", "%14", "%22") BF_M1("%7", "%15", "%23")
        BF_M1("%8", "%16", "%24") BF_M1("%9", "%17", "%25") BF_M1("%10", "%18", "%26") BF_M1("%11", "%19", "%27")
        BF_M2("%0", "%4", "%8") BF_M2("%1", "%5", "%9") BF_M2("%2", "%6", "%10") BF_M2("%3", "%7", "%11")
        : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]),
          "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]));
#undef BF_M1
#undef BF_M2
    const bool b1 = i16 & 2, b0 = i16 & 1;
    float c[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) c[j] = dpp_add<0x4E>(b1 ? b[2 + j] : b[j], b1 ? b[j] : b[2 + j]);           // quad_perm [2,3,0,1]: partner i ^ 2
    return dpp_add<0xB1>(b0 ? c[1] : c[0], b0 ? c[0] : c[1]);                                                // quad_perm [1,0,3,2]: partner i ^ 1
}
struct Par { const float *qw, *qb, *kw, *kb, *emb, *hscale; };
struct Grd { float *dqw, *dqb, *dkw, *dkb, *demb, *dhscale; };

// transposing read of a 4-row x 16-col block of a bf16 LDS tile (row stride ld elements): lane i16 of the 16-lane group gets
// column c0 + i16 of rows r0..r0+3
__device__ __forceinline__ s16x4 tr4(const bf16* tile, int ld, int r0, int c0, int lane) {
    const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tile + (r0 + q) * ld + c0 + 4 * p));
}
__device__ __forceinline__ bf16x8 cat(s16x4 lo, s16x4 hi) {
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ short bfbits(float x) { return __builtin_bit_cast(short, (bf16)x); }

// Load one 16-row block of q (part 0) / k (1) / v (2) rows as fp32 in the operand layout:
// lane (i = l & 15, g = l >> 4) gets channels 32*s + 8*g + jj of row i.
template <int KS>
__device__ __forceinline__ void load_rows_f32(const bf16* __restrict__ base, long row_stride, int col0, long tok0, long tok_stride, int L, int blk,
                                              int lane, float (&x)[KS][8]) {
    const int i = blk * 16 + (lane & 15), g = lane >> 4;
    const int ic = i < L ? i : L - 1;
    const bf16* p = base + (tok0 + ic * tok_stride) * row_stride + col0 + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + 32 * s);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[s][j] = (float)v[j];
    }
}
// LayerNorm statistics of the row held by a lane quad; x becomes xhat, returns rstd
template <int KS>
__device__ __forceinline__ float ln_quad(float (&x)[KS][8], int d) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < KS; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[a][j];
    const float mu = quad_sum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int a = 0; a < KS; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = x[a][j] - mu; v += t * t; }
    const float r = rsqrtf(quad_sum(v) / (float)d + BF_IN_EPS);
#pragma unroll
    for (int a = 0; a < KS; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[a][j] = (x[a][j] - mu) * r;
    return r;
}
template <int KS>
__device__ __forceinline__ void affine_frag(const float (&xh)[KS][8], const float* __restrict__ w, const float* __restrict__ b, float mul, int lane,
                                            bf16x8 (&f)[KS]) {
    const int g = lane >> 4;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = 32 * s + 8 * g + j;
            f[s][j] = (bf16)((xh[s][j] * w[e] + b[e]) * mul);
        }
}

// scores^T for all (key block, query block) pairs -> softmax over keys -> P (and the rescaled A) in registers.
// sc[jb][ib][r]: lane (i = 16*ib + (l & 15)) x key j = 16*jb + 4*(l >> 4) + r.  Everything lane-constant is precomputed by the caller
// (the kernels run this per problem in VALU-bound loops): the T5
// bucket offsets eidx of this lane's (query, key) pairs, the key mask as an additive 0 / -inf, the (query, key) validity mask as a
// 0 / 1 factor.  emb is an LDS table [32][16] (zeros when the block has no bias table): no branch per element.
template <int NB, int KS>
__device__ __forceinline__ void scores_softmax_pre(const bf16x8 (&kf)[NB][KS], const bf16x8 (&qf)[NB][KS], const float* emb, const int (&eidx)[NB][NB][4],
                                                   const float (&mneg)[NB][NB][4], const float (&mk)[NB][NB][4], const float* hscale, int head, int L,
                                                   float (&P)[NB][NB][4], float (&A)[NB][NB][4]) {
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jb][s], qf[ib][s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) P[jb][ib][r] = acc[r] + emb[eidx[jb][ib][r] + head] + mneg[jb][ib][r];
        }
    const float invL = 1.0f / (float)L;
    const float hs = hscale ? hscale[head] : 1.f;
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
        float m = -INFINITY;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, P[jb][ib][r]);
        m = quad_max(m);
        float sum = 0.f;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __expf(P[jb][ib][r] - m); P[jb][ib][r] = e; sum += e; }
        const float inv = 1.f / quad_sum(sum);
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = P[jb][ib][r] * inv;
                P[jb][ib][r] = pr;
                A[jb][ib][r] = (hscale ? invL + (pr - invL) * hs : pr) * mk[jb][ib][r];
            }
    }
}
// the lane constants scores_softmax_pre takes, for sequences of length L
template <int NB>
__device__ __forceinline__ void lane_masks(int L, int lane, float (&mk)[NB][NB][4], float (&mneg)[NB][NB][4], int (&eidx)[NB][NB][4]) {
    const int gq = lane >> 4, i16 = lane & 15;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ib + i16, j = 16 * jb + 4 * gq + r;
                mk[jb][ib][r] = (i < L && j < L) ? 1.f : 0.f;
                mneg[jb][ib][r] = j < L ? 0.f : -INFINITY;
                eidx[jb][ib][r] = t5b(i - j) * 16;
            }
}
// registers (key blocks x 4) of one query block -> the B operand whose k-slot (g, jj) is key 4g+jj / 16+4g+jj-4
template <int NB>
__device__ __forceinline__ bf16x8 pack_keys(const float (&X)[NB][NB][4], int ib) {
    s16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = bfbits(X[0][ib][j]); r[4 + j] = NB > 1 ? bfbits(X[NB - 1][ib][j]) : (short)0; }
    return __builtin_bit_cast(bf16x8, r);
}
// A operand = tile^T for that k-slot order: rows 4g..4g+3 of block 0 and of block 1, columns 16t + (l & 15)
template <int NB>
__device__ __forceinline__ bf16x8 tr_keys(const bf16* tile, int ld, int t, int lane) {
    const int g = lane >> 4;
    const s16x4 lo = tr4(tile, ld, 4 * g, 16 * t, lane);
    s16x4 hi = {0, 0, 0, 0};
    if (NB > 1) hi = tr4(tile, ld, 16 + 4 * g, 16 * t, lane);
    return cat(lo, hi);
}
// A operand = tile^T in natural k order (k = row 8g .. 8g+7).  A 16-row tile (NB == 1) has no rows 16..31: those
// k-slots read (in-bounds, finite) rows again; the B operand it meets holds zeros in the same k-slots (columns 16..31 of the A^T / dS^T tiles).
template <int NB>
__device__ __forceinline__ bf16x8 tr_nat(const bf16* tile, int ld, int t, int lane) {
    const int g = lane >> 4;
    const int r0 = NB == 1 ? 8 * (g & 1) : 8 * g;
    return cat(tr4(tile, ld, r0, 16 * t, lane), tr4(tile, ld, r0 + 4, 16 * t, lane));
}

// stage L rows of `width` channels (global, 16-byte chunks) into a bf16 LDS tile [16*NB][ld]; rows >= L are zeroed
template <int NB>
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ base, long row_stride, int col0, long tok0, long tok_stride, int L, int d,
                                           bf16* tile, int ld, int lane) {
    const int cpr = d / 8;
    for (int c = lane; c < 16 * NB * cpr; c += 64) {
        const int row = c / cpr, e0 = (c % cpr) * 8;
        bf16x8 v;
        if (row < L) v = *reinterpret_cast<const bf16x8*>(base + (tok0 + row * tok_stride) * row_stride + col0 + e0);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
        *reinterpret_cast<bf16x8*>(tile + row * ld + e0) = v;
    }
}

// q / k / v rows of one problem in the MFMA operand layout: lane (i = l & 15, g = l >> 4) holds channels 32*s + 8*g .. +7 of row i
template <int NB, int KS> struct FwdRows { bf16x8 q[NB][KS], k[NB][KS], v[NB][KS]; };
template <int NB, int KS>
__device__ __forceinline__ void load_fwd_rows(FwdRows<NB, KS>& r, const bf16* __restrict__ qkv, const Geo& g, int heads, const Prob& at, int lane) {
    constexpr int D = 32 * KS;
    const int E = heads * D;
    const int head = at.head;
    const long tok0 = at.tok0;
    const int gq = lane >> 4, i16 = lane & 15;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = b * 16 + i16, ic = i < g.L ? i : g.L - 1;      // rows >= L: clamped duplicates, masked / zeroed later
        const bf16* rp = qkv + (tok0 + ic * g.tok_stride) * 3L * E + head * 3 * D + 8 * gq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            r.q[b][ks] = *reinterpret_cast<const bf16x8*>(rp + 32 * ks);
            r.k[b][ks] = *reinterpret_cast<const bf16x8*>(rp + D + 32 * ks);
            r.v[b][ks] = *reinterpret_cast<const bf16x8*>(rp + 2 * D + 32 * ks);
        }
    }
}

// Forward: persistent waves, one (sequence, head) problem at a time.  Every global read of a problem is issued at its top (and
// the next problem's q/k/v rows before that), the LayerNorm affine / T5 table / head scales are staged once per workgroup in
// LDS: a problem is then one round trip instead of six dependent ones.
template <int NB, int KS>
__global__ void __launch_bounds__(256) attn_fwd_mfma(const bf16* __restrict__ qkv, bf16* __restrict__ out, Geo g, int heads, Par p, float out_scale,
                                                     int accumulate) {
    constexpr int D = 32 * KS, LD = D + 16, NT16 = D / 16;
    constexpr bool PREFETCH = NB * KS <= 4;
    extern __shared__ __attribute__((aligned(16))) bf16 smem_fwd[];
    __shared__ __attribute__((aligned(16))) float s_par[4 * 32 * KS];   // qw | qb | kw | kb
    __shared__ float s_emb[32 * 16];
    __shared__ float s_hsc[16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;      // wave-uniform: scalar problem bookkeeping
    bf16* vt = smem_fwd + wave * (16 * NB * LD);
    for (int i = threadIdx.x; i < 32 * 16; i += blockDim.x) {
        const int t = i >> 4, h = i & 15;
        s_emb[i] = (p.emb && h < heads) ? p.emb[t * heads + h] : 0.f;
    }
    if (threadIdx.x < 16) s_hsc[threadIdx.x] = (p.hscale && (int)threadIdx.x < heads) ? p.hscale[threadIdx.x] : 1.f;
    for (int i = threadIdx.x; i < 4 * D; i += blockDim.x) {
        const int q = i / D, e = i % D;
        s_par[i] = (q == 0 ? p.qw : q == 1 ? p.qb : q == 2 ? p.kw : p.kb)[e];
    }
    __syncthreads();
    const int E = heads * D, L = g.L;
    const float scale = rsqrtf((float)D);
    const long nprob = g.nseq * heads;
    const int gq = lane >> 4, i16 = lane & 15;
    const long pstep = (long)gridDim.x * wpb;
    long pr = (long)blockIdx.x * wpb + wave;
    FwdRows<NB, KS> cur, nxt;
    Prob at = locate(g, heads, pr < nprob ? pr : 0), at_next = at;
    if (pr < nprob) load_fwd_rows<NB, KS>(cur, qkv, g, heads, at, lane);
    if (PREFETCH) drain_vm();
    float mk[NB][NB][4], mneg[NB][NB][4];
    int eidx[NB][NB][4];
    lane_masks<NB>(L, lane, mk, mneg, eidx);
    for (; pr < nprob; pr += pstep) {
        const int head = at.head;
        const long tok0 = at.tok0;
        const bool more = pr + pstep < nprob;
        if (more) at_next = locate(g, heads, pr + pstep);
        if (PREFETCH && more) load_fwd_rows<NB, KS>(nxt, qkv, g, heads, at_next, lane);
        bf16x4 old[NB][NT16];
        if (accumulate) {
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                const int i = 16 * ib + i16, ic = i < L ? i : L - 1;
#pragma unroll
                for (int t = 0; t < NT16; ++t)
                    old[ib][t] = *reinterpret_cast<const bf16x4*>(out + (tok0 + ic * g.tok_stride) * (long)E + head * D + 16 * t + 4 * gq);
            }
        }
        bf16x8 qf[NB][KS], kf[NB][KS];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float x[KS][8];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[ks][j] = (float)cur.q[b][ks][j];
            ln_quad<KS>(x, D);
            affine_frag<KS>(x, s_par, s_par + D, scale, lane, qf[b]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[ks][j] = (float)cur.k[b][ks][j];
            ln_quad<KS>(x, D);
            affine_frag<KS>(x, s_par + 2 * D, s_par + 3 * D, 1.f, lane, kf[b]);
            const int row = b * 16 + i16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {      // V tile for the transposing read; rows >= L are zeros
                bf16x8 zv = cur.v[b][ks];
                if (row >= L) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) zv[j] = (bf16)0.f;
                }
                *reinterpret_cast<bf16x8*>(vt + row * LD + 32 * ks + 8 * gq) = zv;
            }
        }
        float P[NB][NB][4], A[NB][NB][4];
        scores_softmax_pre<NB, KS>(kf, qf, s_emb, eidx, mneg, mk, p.hscale ? s_hsc : nullptr, head, L, P, A);
        wsync();   // V tile visible to the wave
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
            const bf16x8 pa = pack_keys<NB>(A, ib);
            const int i = 16 * ib + i16;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_keys<NB>(vt, LD, t, lane), pa, o, 0, 0, 0);
                if (i < L) {
                    bf16* dst = out + (tok0 + i * g.tok_stride) * (long)E + head * D + 16 * t + 4 * gq;
                    float v[4] = {o[0] * out_scale, o[1] * out_scale, o[2] * out_scale, o[3] * out_scale};
                    if (accumulate) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)old[ib][t][r];
                    }
                    const bf16x4 w4 = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(dst) = w4;
                }
            }
        }
        wsync();   // before the next problem overwrites the V tile
        if (PREFETCH) cur = nxt;
        else if (more) load_fwd_rows<NB, KS>(cur, qkv, g, heads, at_next, lane);
        at = at_next;
    }
}

// Axial attention forward, both passes in one launch (layers/attention.py:212-297: along W, then along H, (xx + xy) / 2).  A workgroup
// owns one (frame, head): its waves first run the h row sequences and leave 0.5 * result in an LDS tile [h*w][D] (rounded to bf16
// exactly as the two-launch form rounds its intermediate), then the w column sequences, which add their half and write the output
// once.  QKV is read from HBM once (the second pass hits L2), the output is never read back: the two-launch form re-reads QKV and
// read-modify-writes `out`.  Results are bit-identical to bf_attn_fwd(W, out_scale 0.5) + bf_attn_fwd(H, 0.5, accumulate).
// NORM: the InstanceNorm that follows the attention (norm2, layers/attention.py:298) runs here as well -- a (frame, head) tile holds
// all h*w tokens of its d channels, i.e. everything a per-(frame, channel) statistic needs: two-pass mean / variance over the LDS tile,
// then `out` and the normalised `out_n` (the out-projection's operand) are written together; the separate statistics launch is gone.
struct AxNorm { const float* w; const float* b; bf16* out_n; float* mean; float* rstd; float* sc; float* sh; };
template <int KS, bool NORM, int WPB = 4>      // WPB waves per workgroup: 4 (two workgroups per CU, large batches) or 12 (inference: one round per pass)
__global__ void __launch_bounds__(64 * WPB) attn_fwd_axial_mfma(const bf16* __restrict__ qkv, bf16* __restrict__ out, int frames, int h, int w, int heads, Par p,
                                                           const float* __restrict__ hscale_y, AxNorm nrm) {
    constexpr int NB = 1, D = 32 * KS, LD = D + 16, NT16 = D / 16, DP = D + 8;
    extern __shared__ __attribute__((aligned(16))) bf16 smem_ax[];
    __shared__ __attribute__((aligned(16))) float s_par[4 * 32 * KS];   // qw | qb | kw | kb
    __shared__ float s_emb[32 * 16];
    __shared__ float s_hsx[16], s_hsy[16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;      // wave-uniform: scalar sequence bookkeeping
    bf16* vt = smem_ax + wave * (16 * LD);
    bf16* ot = smem_ax + wpb * (16 * LD);                                 // [h * w][DP]
    for (int i = threadIdx.x; i < 32 * 16; i += blockDim.x) {
        const int t = i >> 4, hd = i & 15;
        s_emb[i] = (p.emb && hd < heads) ? p.emb[t * heads + hd] : 0.f;
    }
    if (threadIdx.x < 16) {
        s_hsx[threadIdx.x] = (p.hscale && (int)threadIdx.x < heads) ? p.hscale[threadIdx.x] : 1.f;
        s_hsy[threadIdx.x] = (hscale_y && (int)threadIdx.x < heads) ? hscale_y[threadIdx.x] : 1.f;
    }
    for (int i = threadIdx.x; i < 4 * D; i += blockDim.x) {
        const int q = i / D, e = i % D;
        s_par[i] = (q == 0 ? p.qw : q == 1 ? p.qb : q == 2 ? p.kw : p.kb)[e];
    }
    __syncthreads();
    const int E = heads * D, S = h * w;
    const float scale = rsqrtf((float)D);
    const int gq = lane >> 4, i16 = lane & 15;
    const Geo gW{(long)frames * h, w, 1, w, 0, 1}, gH{(long)frames * w, h, w, S, 1, w};
    const int RW = (h + wpb - 1) / wpb, RH = (w + wpb - 1) / wpb;         // rounds per phase (uniform over the waves: barriers)
    float mkW[NB][NB][4], mnegW[NB][NB][4], mkH[NB][NB][4], mnegH[NB][NB][4];      // lane constants of the two passes (sequence lengths w and h)
    int eidxW[NB][NB][4], eidxH[NB][NB][4];
    lane_masks<NB>(w, lane, mkW, mnegW, eidxW);
    lane_masks<NB>(h, lane, mkH, mnegH, eidxH);
    const long ntile = (long)frames * heads;
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int frame = (int)((unsigned)tile / (unsigned)heads), head = (int)((unsigned)tile - (unsigned)frame * (unsigned)heads);      // (< 2^31 tiles: host-checked)
        for (int k = 0; k < RW + RH; ++k) {
            const bool isH = k >= RW;
            const int idx = wave + (isH ? k - RW : k) * wpb;
            const int L = isH ? h : w;
            if (k == RW) __syncthreads();                                   // every row sequence has left its half in the tile
            if (idx < (isH ? w : h)) {
                const Geo& g = isH ? gH : gW;
                const long seq = (long)frame * (isH ? w : h) + idx;
                FwdRows<NB, KS> cur;
                load_fwd_rows<NB, KS>(cur, qkv, g, heads, locate(g, heads, seq * heads + head), lane);
                bf16x8 qf[NB][KS], kf[NB][KS];
                {
                    float x[KS][8];
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[ks][j] = (float)cur.q[0][ks][j];
                    ln_quad<KS>(x, D);
                    affine_frag<KS>(x, s_par, s_par + D, scale, lane, qf[0]);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[ks][j] = (float)cur.k[0][ks][j];
                    ln_quad<KS>(x, D);
                    affine_frag<KS>(x, s_par + 2 * D, s_par + 3 * D, 1.f, lane, kf[0]);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {      // V tile for the transposing read; rows >= L are zeros
                        bf16x8 zv = cur.v[0][ks];
                        if (i16 >= L) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) zv[j] = (bf16)0.f;
                        }
                        *reinterpret_cast<bf16x8*>(vt + i16 * LD + 32 * ks + 8 * gq) = zv;
                    }
                }
                float P[NB][NB][4], A[NB][NB][4];
                if (isH) scores_softmax_pre<NB, KS>(kf, qf, s_emb, eidxH, mnegH, mkH, hscale_y ? s_hsy : nullptr, head, L, P, A);
                else scores_softmax_pre<NB, KS>(kf, qf, s_emb, eidxW, mnegW, mkW, p.hscale ? s_hsx : nullptr, head, L, P, A);
                wsync();   // V tile visible to the wave
                const bf16x8 pa = pack_keys<NB>(A, 0);
#pragma unroll
                for (int t = 0; t < NT16; ++t) {
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_keys<NB>(vt, LD, t, lane), pa, o, 0, 0, 0);
                    if (i16 < L) {
                        const int tok = isH ? i16 * w + idx : idx * w + i16;                 // token inside the frame
                        bf16* cell = ot + tok * DP + 16 * t + 4 * gq;
                        float v[4] = {o[0] * 0.5f, o[1] * 0.5f, o[2] * 0.5f, o[3] * 0.5f};
                        if (isH) {
                            const bf16x4 old = *reinterpret_cast<const bf16x4*>(cell);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
                        }
                        const bf16x4 w4 = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        if (isH && !NORM) *reinterpret_cast<bf16x4*>(out + ((long)frame * S + tok) * E + head * D + 16 * t + 4 * gq) = w4;
                        else *reinterpret_cast<bf16x4*>(cell) = w4;
                    }
                }
                wsync();   // before the next problem overwrites the V tile
            }
        }
        __syncthreads();                                                   // NORM: the tile holds the frame's output; else: it is free again
        if constexpr (NORM) {
            constexpr int CPR = D / 8, RGN = 256 / CPR;                     // 16-byte chunks per token row, row groups
            float* red = reinterpret_cast<float*>(ot + ((S * DP + 7) & ~7)); // [RGN][D] partial sums, then [2][D] mean | rstd
            float* stat = red + RGN * D;
            const int cg = threadIdx.x % CPR, rg = threadIdx.x / CPR;
            const bool act = rg < RGN;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            if (act)
                for (int r = rg; r < S; r += RGN) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(ot + r * DP + cg * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
                }
            if (act) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[rg * D + cg * 8 + j] = acc[j];
            }
            __syncthreads();
            if ((int)threadIdx.x < D) {
                float t = 0.f;
                for (int g2 = 0; g2 < RGN; ++g2) t += red[g2 * D + threadIdx.x];
                stat[threadIdx.x] = t / (float)S;
            }
            __syncthreads();
            float mu[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { mu[j] = stat[cg * 8 + j]; acc[j] = 0.f; }
            if (act)
                for (int r = rg; r < S; r += RGN) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(ot + r * DP + cg * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float dlt = (float)v[j] - mu[j]; acc[j] += dlt * dlt; }
                }
            if (act) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[rg * D + cg * 8 + j] = acc[j];
            }
            __syncthreads();
            if ((int)threadIdx.x < D) {
                float t = 0.f;
                for (int g2 = 0; g2 < RGN; ++g2) t += red[g2 * D + threadIdx.x];
                const float r = rsqrtf(t / (float)S + BF_IN_EPS);
                const int c = head * D + threadIdx.x;
                const long o = (long)frame * E + c;
                const float a = r * nrm.w[c], m = stat[threadIdx.x];
                stat[D + threadIdx.x] = r;
                nrm.mean[o] = m; nrm.rstd[o] = r; nrm.sc[o] = a; nrm.sh[o] = nrm.b[c] - m * a;
            }
            __syncthreads();
            if (act) {
                float aa[8], ss[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = head * D + cg * 8 + j;
                    aa[j] = stat[D + cg * 8 + j] * nrm.w[c];
                    ss[j] = nrm.b[c] - mu[j] * aa[j];
                }
                for (int r = rg; r < S; r += RGN) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(ot + r * DP + cg * 8);
                    bf16x8 nv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) nv[j] = (bf16)((float)v[j] * aa[j] + ss[j]);
                    const long go = ((long)frame * S + r) * E + head * D + cg * 8;
                    *reinterpret_cast<bf16x8*>(out + go) = v;
                    *reinterpret_cast<bf16x8*>(nrm.out_n + go) = nv;
                }
            }
            __syncthreads();                                               // the tile is free for the next (frame, head)
        }
    }
}

// The q / k / v / dO rows of one problem exactly as the MFMA operand layout wants them: lane (i = l & 15, g = l >> 4) holds
// channels 32*s + 8*g .. +7 of row i.  The backward loads the NEXT problem's rows while it works on the current one: a wave runs
// ~20 dependent phases per problem, and with only two waves per SIMD every exposed global round trip is paid in full.
template <int NB, int KS> struct RawRows { bf16x8 q[NB][KS], k[NB][KS], v[NB][KS], d[NB][KS]; };
template <int NB, int KS> struct OldRows { bf16x8 q[NB][KS], k[NB][KS], v[NB][KS]; };

template <int NB, int KS>
__device__ __forceinline__ void load_raw(RawRows<NB, KS>& r, const bf16* __restrict__ qkv, const bf16* __restrict__ dout, const Geo& g, int heads, const Prob& at,
                                         int lane) {
    constexpr int D = 32 * KS;
    const int E = heads * D;
    const int head = at.head;
    const long tok0 = at.tok0;
    const int gq = lane >> 4, i16 = lane & 15;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = b * 16 + i16, ic = i < g.L ? i : g.L - 1;      // rows >= L: clamped duplicates, masked / never stored later
        const bf16* rp = qkv + (tok0 + ic * g.tok_stride) * 3L * E + head * 3 * D + 8 * gq;
        const bf16* dp = dout + (tok0 + ic * g.tok_stride) * (long)E + head * D + 8 * gq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            r.q[b][ks] = *reinterpret_cast<const bf16x8*>(rp + 32 * ks);
            r.k[b][ks] = *reinterpret_cast<const bf16x8*>(rp + D + 32 * ks);
            r.v[b][ks] = *reinterpret_cast<const bf16x8*>(rp + 2 * D + 32 * ks);
            r.d[b][ks] = *reinterpret_cast<const bf16x8*>(dp + 32 * ks);
        }
    }
}

// ACC >= 0: the accumulate mode as a compile-time constant (the training shapes: every `if (raw_in)` / `if (accumulate)` on a run-time flag
// is otherwise compiled as compute-both-and-select, ~100 VALU instructions per problem in a VALU-bound loop); ACC < 0: the run-time argument.
template <int NB, int KS, int ACC>
__global__ void __launch_bounds__(256, ((NB == 1 && KS <= 2) ? 2 : 1)) attn_bwd_mfma(const bf16* __restrict__ qkv, const bf16* __restrict__ dout, bf16* __restrict__ dqkv, Geo g,
                                                     int heads, Par p, Grd gr, float out_scale, int accumulate_rt, float* __restrict__ ws) {
    constexpr int D = 32 * KS, LD = D + 16, NT16 = D / 16, R = 16 * NB, LDP = 32 + 8;
    constexpr bool PREFETCH = NB * KS <= 4;      // register budget: 16 * NB * KS VGPRs for the look-ahead rows
    extern __shared__ __attribute__((aligned(16))) bf16 smem_bwd[];
    __shared__ float s_demb[32 * 16];
    __shared__ float s_dhs[16];
    // parameter copies: every problem reads the q/k LayerNorm affine, the T5 bias table and the head scale; from global memory each
    // of those reads is a dependent round trip in the middle of the problem
    __shared__ __attribute__((aligned(16))) float s_par[4 * 32 * KS];   // qw | qb | kw | kb
    __shared__ float s_emb[32 * 16];
    __shared__ float s_hsc[16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;      // provably wave-uniform: problem bookkeeping in SGPRs
    // per-wave: Qn, Kn, dO tiles [R][LD] bf16; A^T and dS^T tiles [R][LDP] bf16 ([key j][query i])
    constexpr int PER_WAVE = 3 * R * LD + 2 * R * LDP;
    static_assert(3 * R * LD * 2 >= 4 * D * 4, "the parameter-gradient flush must fit in the operand tiles");
    bf16* qn_t = smem_bwd + wave * PER_WAVE;
    bf16* kn_t = qn_t + R * LD;
    bf16* do_t = kn_t + R * LD;
    bf16* at_t = do_t + R * LD;
    bf16* ds_t = at_t + R * LDP;
    float* relay = reinterpret_cast<float*>(qn_t);      // end of the kernel only: this wave's LayerNorm parameter sums on their way to the block total
    for (int i = threadIdx.x; i < 32 * 16; i += blockDim.x) {
        s_demb[i] = 0.f;
        const int t = i >> 4, h = i & 15;
        s_emb[i] = (p.emb && h < heads) ? p.emb[t * heads + h] : 0.f;
    }
    if (threadIdx.x < 16) { s_dhs[threadIdx.x] = 0.f; s_hsc[threadIdx.x] = (p.hscale && (int)threadIdx.x < heads) ? p.hscale[threadIdx.x] : 1.f; }
    for (int i = threadIdx.x; i < 4 * D; i += blockDim.x) {
        const int q = i / D, e = i % D;
        s_par[i] = (q == 0 ? p.qw : q == 1 ? p.qb : q == 2 ? p.kw : p.kb)[e];
    }
    __syncthreads();
    const int E = heads * D, L = g.L;
    const float scale = rsqrtf((float)D);
    const long nprob = g.nseq * heads;
    const int gq = lane >> 4, i16 = lane & 15;
    // accumulate: bit 0 = add what dqkv holds; bit 1 ("raw out") = leave the q / k gradients BEFORE the LayerNorm backward (the gradient with
    // respect to the LayerNorm outputs) and skip that backward and its parameter sums; bit 2 ("raw in", with bit 0) = the q / k values dqkv holds
    // are such raw gradients: they are added in front of the LayerNorm backward.  The LayerNorm backward is linear in its incoming gradient, so
    // the two axial passes over the same tokens (layers/attention.py:218-277) run it ONCE, on the sum: pass W with 2, pass H with 1 | 4.
    const int mode = ACC >= 0 ? ACC : accumulate_rt;
    const bool raw_out = mode & 2, raw_in = mode & 4, accumulate = mode & 1;

    // q/k LayerNorm parameter gradients: value v = ((2 * part + {dw: 0, db: 1}) * KS + ks) * 8 + j of channel group gq.  Each
    // problem's 16-row totals (DPP row reduction, VALU only) are deposited in lane (v & 15) of the row, slot v >> 4: 2 * KS
    // accumulator registers per lane instead of 32 * KS, which is what lets two waves share a SIMD.
    constexpr int NACC = 2 * KS;
    float a_ln[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) a_ln[k] = 0.f;
    // When the problem stride is a multiple of `heads` every problem of this wave belongs to ONE head: the T5-bias and head-scale
    // gradients then accumulate in registers (a lane's (query, key) pairs are fixed) and reach LDS once, at the end.  LDS float
    // atomics cost ~500 cycles per wave instruction on gfx950: they must stay out of the per-problem path.
    const long pstep = (long)gridDim.x * wpb;
    const bool one_head = pstep % heads == 0;
    float a_emb[NB][NB][4], a_dhs = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) a_emb[jb][ib][r] = 0.f;
    long pr = (long)blockIdx.x * wpb + wave;
    RawRows<NB, KS> cur, nxt;
    Prob at = locate(g, heads, pr < nprob ? pr : 0), at_next = at;
    if (pr < nprob) load_raw<NB, KS>(cur, qkv, dout, g, heads, at, lane);
    if (PREFETCH) drain_vm();
    // lane-constant masks and T5 bucket offsets of this lane's (query, key) pairs: once, not per problem
    float mk[NB][NB][4], mneg[NB][NB][4];
    int eidx[NB][NB][4];
    lane_masks<NB>(L, lane, mk, mneg, eidx);
    if (NB == 1) {      // k-slots 16..31 of the natural-order operands must read zeros: columns 16..31 of the two small tiles, which nothing else writes
        for (int c = lane; c < 16 * 16; c += 64) { at_t[(c >> 4) * LDP + 16 + (c & 15)] = (bf16)0.f; ds_t[(c >> 4) * LDP + 16 + (c & 15)] = (bf16)0.f; }
    }
    for (; pr < nprob; pr += pstep) {
        const int head = at.head;
        const long tok0 = at.tok0;
        const bool more = pr + pstep < nprob;
        if (more) at_next = locate(g, heads, pr + pstep);
        if (PREFETCH && more) load_raw<NB, KS>(nxt, qkv, dout, g, heads, at_next, lane);
        // gradient rows this problem accumulates into (second axial pass): issued now, consumed at the very end
        OldRows<NB, KS> old;
        if (accumulate) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = b * 16 + i16, rc = row < L ? row : L - 1;
                const bf16* ob = dqkv + (tok0 + rc * g.tok_stride) * 3L * E + head * 3 * D;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    old.q[b][ks] = *reinterpret_cast<const bf16x8*>(ob + 32 * ks + 8 * gq);
                    old.k[b][ks] = *reinterpret_cast<const bf16x8*>(ob + D + 32 * ks + 8 * gq);
                    old.v[b][ks] = *reinterpret_cast<const bf16x8*>(ob + 2 * D + 32 * ks + 8 * gq);
                }
                if (raw_in && row >= L) {      // a clamped copy of row L - 1: must not reach the LayerNorm parameter sums
                    const bf16x8 z8 = {};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) { old.q[b][ks] = z8; old.k[b][ks] = z8; }
                }
            }
        }
        float xq[NB][KS][8], xk[NB][KS][8], rq[NB], rk[NB];
        bf16x8 qf[NB][KS], kf[NB][KS], vf[NB][KS], df[NB][KS];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) { xq[b][ks][j] = (float)cur.q[b][ks][j]; xk[b][ks][j] = (float)cur.k[b][ks][j]; }
            rq[b] = ln_quad<KS>(xq[b], D);
            affine_frag<KS>(xq[b], s_par, s_par + D, scale, lane, qf[b]);
            rk[b] = ln_quad<KS>(xk[b], D);
            affine_frag<KS>(xk[b], s_par + 2 * D, s_par + 3 * D, 1.f, lane, kf[b]);
            // Qn (with the d^-1/2 fold), Kn and dO tiles for the transposed operands
            const int row = b * 16 + i16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                vf[b][ks] = cur.v[b][ks];
                df[b][ks] = cur.d[b][ks];
                // rows >= L hold copies of row L - 1 (finite): every product that reduces over them meets A = dS = 0 there (the masks mk).
                // Column order of the tiles: this lane's channels 32 ks + 8 g + j go to column 16 (2 ks + (j >> 2)) + 4 g + (j & 3).  The
                // products below put tile column 16 t + 4 g + r in register r of lane group g, so with this order a lane gets dQ / dK / dV
                // for exactly the channels whose xhat it holds (t = 2 ks + (j >> 2), r = j & 3): the LayerNorm backward needs no re-layout
                // through LDS, and the gradient rows leave as 16-byte stores.
#define BF_HALVES(tile, v) { \
                    *reinterpret_cast<bf16x4*>(tile + row * LD + 32 * ks + 4 * gq) = __builtin_shufflevector(v, v, 0, 1, 2, 3); \
                    *reinterpret_cast<bf16x4*>(tile + row * LD + 32 * ks + 16 + 4 * gq) = __builtin_shufflevector(v, v, 4, 5, 6, 7); }
                BF_HALVES(qn_t, qf[b][ks]) BF_HALVES(kn_t, kf[b][ks]) BF_HALVES(do_t, df[b][ks])
#undef BF_HALVES
            }
        }
        float P[NB][NB][4], A[NB][NB][4], dA[NB][NB][4];
        scores_softmax_pre<NB, KS>(kf, qf, s_emb, eidx, mneg, mk, p.hscale ? s_hsc : nullptr, head, L, P, A);
        // dA^T[j][i] = sum_e V[j][e] dO[i][e] * out_scale
        const float hs = s_hsc[head];
        const float invL = 1.0f / (float)L;
        float dhs = 0.f;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[jb][ks], df[ib][ks], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[r] * out_scale * mk[jb][ib][r];
                    if (p.hscale) { dhs += (P[jb][ib][r] - invL) * v; v *= hs; }
                    dA[jb][ib][r] = v;     // now dP
                }
            }
        // dS = P * (dP - sum_j P dP)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
            float dot = 0.f;
#pragma unroll
            for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) dot += P[jb][ib][r] * dA[jb][ib][r];
            dot = quad_sum(dot);
#pragma unroll
            for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ib + i16, j = 16 * jb + 4 * gq + r;
                    const float v = P[jb][ib][r] * (dA[jb][ib][r] - dot) * mk[jb][ib][r];
                    dA[jb][ib][r] = v;     // now dS
                    if (one_head) a_emb[jb][ib][r] += v;
                    else if (gr.demb && i < L && j < L) atomicAdd(&s_demb[eidx[jb][ib][r] + head], v);
                    // (A of a query row >= L is already zero: scores_softmax_pre masks it)
                    // transposed copies [key j][query i] for the products that reduce over queries
                    at_t[j * LDP + i] = (bf16)A[jb][ib][r];
                    ds_t[j * LDP + i] = (bf16)v;
                }
        }
        if (one_head) a_dhs += dhs;
        else if (p.hscale && gr.dhscale) {
            dhs = wave_sum(dhs);
            if (lane == 0) atomicAdd(&s_dhs[head], dhs);
        }
        wsync();
        // ---- dV^T[e][j] = sum_i dO[i][e] A[i][j]   and   dKn^T[e][j] = sum_i Qn[i][e] dS[i][j]      (k = query i, natural order)
        // ---- dQn^T[e][i] = sum_j Kn[j][e] dS[i][j]                                                (k-slots in key order)
        float dq[NB][NT16][4], dk[NB][NT16][4];
        f32x4 dv_even[NB];
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            const bf16x8 doT = tr_nat<NB>(do_t, LD, t, lane);
            const bf16x8 qnT = tr_nat<NB>(qn_t, LD, t, lane);
            const bf16x8 knT = tr_keys<NB>(kn_t, LD, t, lane);
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                // B operands: row (key) j = 16*jb + (l & 15), 8 consecutive queries 8g..8g+7
                const bf16x8 aB = *reinterpret_cast<const bf16x8*>(at_t + (16 * jb + i16) * LDP + 8 * gq);
                const bf16x8 sB = *reinterpret_cast<const bf16x8*>(ds_t + (16 * jb + i16) * LDP + 8 * gq);
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT, aB, z, 0, 0, 0);
                const f32x4 dkk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qnT, sB, z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) dk[jb][t][r] = dkk[r];     // Qn tile carries d^-1/2 already
                if (!(t & 1)) dv_even[jb] = dv;
                else {      // dV: lane (key j, channels 32 ks + 8 g .. + 7), ks = t >> 1 -> 16-byte store
                    const int j = 16 * jb + i16, ks = t >> 1;
                    if (j < L) {
                        bf16x8 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float lo = dv_even[jb][r] * out_scale, hi = dv[r] * out_scale;
                            if (accumulate) { lo += (float)old.v[jb][ks][r]; hi += (float)old.v[jb][ks][4 + r]; }
                            o[r] = (bf16)lo; o[4 + r] = (bf16)hi;
                        }
                        *reinterpret_cast<bf16x8*>(dqkv + (tok0 + j * g.tok_stride) * 3L * E + head * 3 * D + 2 * D + 32 * ks + 8 * gq) = o;
                    }
                }
            }
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 dqq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(knT, pack_keys<NB>(dA, ib), z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) dq[ib][t][r] = dqq[r] * scale;
            }
        }
        // dq[b][2 ks + (j >> 2)][j & 3] / dk[..] are this lane's gradients for channel 32 ks + 8 g + j of row 16 b + (l & 15) (see the tile writes)
        if (raw_out) {      // the first of two passes over these tokens: the gradients with respect to the LayerNorm outputs, as they stand
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = 16 * b + i16;
                if (row < L) {
                    bf16* dst = dqkv + (tok0 + row * g.tok_stride) * 3L * E + head * 3 * D + 8 * gq;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        bf16x8 oq, ok;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float vq = dq[b][2 * ks + (j >> 2)][j & 3], vk = dk[b][2 * ks + (j >> 2)][j & 3];
                            if (accumulate) { vq += (float)old.q[b][ks][j]; vk += (float)old.k[b][ks][j]; }
                            oq[j] = (bf16)vq; ok[j] = (bf16)vk;
                        }
                        *reinterpret_cast<bf16x8*>(dst + 32 * ks) = oq;
                        *reinterpret_cast<bf16x8*>(dst + D + 32 * ks) = ok;
                    }
                }
            }
        } else
        // ---- LayerNorm backward, q then k: two independent register-only chains
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const float* w = part == 0 ? s_par : s_par + 2 * D;
            float pw[KS][8], pb[KS][8];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) pw[ks][j] = pb[ks][j] = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = 16 * b + i16;
                float dn[KS][8];
                float m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xh = part == 0 ? xq[b][ks][j] : xk[b][ks][j];
                        float d0 = part == 0 ? dq[b][2 * ks + (j >> 2)][j & 3] : dk[b][2 * ks + (j >> 2)][j & 3];      // rows >= L: exactly zero (dS carries the masks)
                        if (raw_in) d0 += (float)(part == 0 ? old.q[b][ks][j] : old.k[b][ks][j]);      // the other pass's raw gradient (zeroed for rows >= L above)
                        pw[ks][j] += d0 * xh;
                        pb[ks][j] += d0;
                        const float gg = d0 * w[32 * ks + 8 * gq + j];
                        dn[ks][j] = gg;
                        m1 += gg; m2 += gg * xh;
                    }
                }
                m1 = quad_sum(m1) / (float)D;
                m2 = quad_sum(m2) / (float)D;
                const float rs = part == 0 ? rq[b] : rk[b];
                if (row < L) {
                    bf16* dst = dqkv + (tok0 + row * g.tok_stride) * 3L * E + head * 3 * D + part * D + 8 * gq;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        bf16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float xh = part == 0 ? xq[b][ks][j] : xk[b][ks][j];
                            float v = rs * (dn[ks][j] - m1 - xh * m2);
                            if (accumulate && !raw_in) v += (float)(part == 0 ? old.q[b][ks][j] : old.k[b][ks][j]);
                            o[j] = (bf16)v;
                        }
                        *reinterpret_cast<bf16x8*>(dst + 32 * ks) = o;
                    }
                }
            }
            {   // value v = 16 part KS + (dw: 8 ks + j | db: 8 KS + 8 ks + j) ends in lane v & 15 of the row, slot v >> 4: KS groups of sixteen
                float flat[16 * KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { flat[8 * ks + j] = pw[ks][j]; flat[8 * KS + 8 * ks + j] = pb[ks][j]; }
#pragma unroll
                for (int gi = 0; gi < KS; ++gi) {
                    float grp[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) grp[q] = flat[16 * gi + q];
                    a_ln[part * KS + gi] += row16_scatter_sum(grp, i16);
                }
            }
        }
        wsync();   // this problem's transposed reads of the tiles sit in front of the next problem's writes
        if (PREFETCH) cur = nxt;
        else if (more) load_raw<NB, KS>(cur, qkv, dout, g, heads, at_next, lane);
        at = at_next;
    }
    // ---- flush parameter gradients.  Thousands of waves adding to the same few hundred addresses serialise at the
    // memory side, so: block reduce in LDS -> ONE row of plain stores per block into the workspace (summed by
    // attn_ws_reduce), or atomics when no workspace is given.
    if (one_head && pr - pstep >= 0) {        // register-held T5-bias / head-scale gradients of this wave's head
        const int head = (int)(((long)blockIdx.x * wpb + wave) % heads);
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ib + i16, j = 16 * jb + 4 * gq + r;
                    if (gr.demb && i < L && j < L) atomicAdd(&s_demb[t5b(i - j) * 16 + head], a_emb[jb][ib][r]);
                }
        if (p.hscale && gr.dhscale) {
            a_dhs = wave_sum(a_dhs);
            if (lane == 0) atomicAdd(&s_dhs[head], a_dhs);
        }
    }
#pragma unroll
    for (int k = 0; k < NACC; ++k) {     // plain stores into this wave's (now idle) tile area; the waves are summed below
        const int v = 16 * k + i16, q = v / (8 * KS), ks = (v >> 3) % KS, j = v & 7;
        relay[q * D + 32 * ks + 8 * gq + j] = a_ln[k];
    }
    __syncthreads();
    const int nvals = 4 * D + 32 * heads + heads;
    for (int i = threadIdx.x; i < nvals; i += blockDim.x) {
        float val;
        float* dst;
        if (i < 4 * D) {
            val = 0.f;
            for (int w = 0; w < wpb; ++w) val += reinterpret_cast<const float*>(smem_bwd + w * PER_WAVE)[i];
            const int q = i / D, e = i % D; dst = (q == 0 ? gr.dqw : q == 1 ? gr.dqb : q == 2 ? gr.dkw : gr.dkb); if (dst) dst += e; }
        else if (i < 4 * D + 32 * heads) { const int t = i - 4 * D; val = s_demb[(t / heads) * 16 + (t % heads)]; dst = gr.demb ? gr.demb + t : nullptr; }
        else { const int t = i - 4 * D - 32 * heads; val = s_dhs[t]; dst = gr.dhscale ? gr.dhscale + t : nullptr; }
        if (ws) ws[(long)blockIdx.x * nvals + i] = val;
        else if (dst && val != 0.f) atomicAdd(dst, val);
    }
}

// dst += sum over workspace rows (param_reduce.h)
__global__ void __launch_bounds__(64 * BF_RED_FL) attn_ws_reduce(AttnReduceJob j) {
    __shared__ float red[1][BF_RED_FL][64];
    attn_reduce_block(j, blockIdx.x, blockIdx.y, gridDim.y, red);
}

template <typename K>
int set_lds(K kernel, size_t shm) {
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return bf_fail(e, __FILE__, __LINE__);
    }
    return 0;
}

template <int NB, int KS>
int go_fwd(const bf16* qkv, bf16* out, Geo g, int heads, Par p, float out_scale, int accumulate, hipStream_t st) {
    constexpr int D = 32 * KS;
    const int wpb = 4;
    const size_t shm = (size_t)wpb * 16 * NB * (D + 16) * sizeof(bf16);
    const long nprob = g.nseq * heads;
    BF_REQUIRE(nprob < (1L << 31) && g.inner < (1L << 31), "attention: problem count must fit 31 bits");
    // measured (tools/attn_bench.py): 2 workgroups per CU of look-ahead waves beat 8 of plain ones (18.7 vs 21.0 us at the bench shape)
    static const int bpc = bf_knob("BF_ATTN_FWD_BPC", 2);
    const int grid = (int)std::min<long>((nprob + wpb - 1) / wpb, 256L * bpc);
    hipLaunchKernelGGL((attn_fwd_mfma<NB, KS>), dim3(grid), dim3(wpb * 64), shm, st, qkv, out, g, heads, p, out_scale, accumulate);
    BF_CHECK_LAUNCH();
    return 0;
}
template <int NB, int KS, int ACC>
int go_bwd_mode(const bf16* qkv, const bf16* dout, bf16* dqkv, Geo g, int heads, Par p, Grd gr, float out_scale, int accumulate, float* ws,
                long ws_floats, int* rows_out, hipStream_t st) {
    constexpr int D = 32 * KS, R = 16 * NB;
    const int wpb = NB == 1 ? 4 : 2;
    const size_t shm = (size_t)wpb * (3 * R * (D + 16) + 2 * R * 40) * sizeof(bf16);
    if (int rc = set_lds(attn_bwd_mfma<NB, KS, ACC>, shm)) return rc;
    const long nprob = g.nseq * heads;
    BF_REQUIRE(nprob < (1L << 31) && g.inner < (1L << 31), "attention: problem count must fit 31 bits");
    const int nvals = 4 * D + 32 * heads + heads;
    // persistent waves: one resident set of workgroups (what the register / LDS budget admits per CU), each wave loops over problems
    static int resident = 0;       // per instantiation
    if (!resident) {
        int dev = 0, cus = 256, per_cu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_bwd_mfma<NB, KS, ACC>, wpb * 64, shm) != hipSuccess || per_cu < 1) per_cu = 1;
        static const int bpc = bf_knob("BF_ATTN_BWD_BPC", 0);
        if (bpc > 0) per_cu = bpc;
        resident = cus * per_cu;
    }
    long grid = std::min<long>((nprob + wpb - 1) / wpb, (long)resident);
    {   // keep (grid * wpb) a multiple of heads when that costs at most a few workgroups: one head per wave (see the kernel)
        long a = wpb, b = heads;
        while (b) { const long t = a % b; a = b; b = t; }      // a = gcd(wpb, heads)
        const long m = heads / a;
        if (grid >= 8 * m) grid -= grid % m;
    }
    if (ws && ws_floats < grid * nvals) { grid = ws_floats / nvals; if (grid < 1) ws = nullptr; }
    hipLaunchKernelGGL((attn_bwd_mfma<NB, KS, ACC>), dim3((int)grid), dim3(wpb * 64), shm, st, qkv, dout, dqkv, g, heads, p, gr, out_scale, accumulate, ws);
    BF_CHECK_LAUNCH();
    if (rows_out) { *rows_out = ws ? (int)grid : 0; return 0; }       // the caller reduces the workspace rows later (AttnReduceJob)
    if (ws) {
        const AttnReduceJob j{ws, (int)grid, D, heads, gr.dqw, gr.dqb, gr.dkw, gr.dkb, gr.demb, gr.dhscale};
        hipLaunchKernelGGL(attn_ws_reduce, dim3(bf_cdiv(nvals, 64), 1), dim3(64 * BF_RED_FL), 0, st, j);      // one row slice: one writer per value
        BF_CHECK_LAUNCH();
    }
    return 0;
}
template <int NB, int KS>
int go_bwd(const bf16* qkv, const bf16* dout, bf16* dqkv, Geo g, int heads, Par p, Grd gr, float out_scale, int accumulate, float* ws,
           long ws_floats, int* rows_out, hipStream_t st) {
#define BF_MODE(A) return go_bwd_mode<NB, KS, A>(qkv, dout, dqkv, g, heads, p, gr, out_scale, accumulate, ws, ws_floats, rows_out, st)
    if constexpr (NB == 1 && KS == 2) {      // FiLMAViT-small's head width, sequences of <= 16 tokens: one instantiation per mode
        switch (accumulate) { case 0: BF_MODE(0); case 1: BF_MODE(1); case 2: BF_MODE(2); case 5: BF_MODE(5); default: break; }
    }
    BF_MODE(-1);
#undef BF_MODE
}

}  // namespace

// dispatched from bf_attn_fwd / bf_attn_bwd (attn.hip) for bf16 with d in {32, 64, 96, 128}
int bf_attn_fwd_mfma(const void* qkv, void* out, long nseq, int L, long inner, long outer_stride, long inner_stride, long tok_stride, int heads,
                     int d, const float* qw, const float* qb, const float* kw, const float* kb, const float* emb, const float* hscale,
                     float out_scale, int accumulate, hipStream_t st) {
    Geo g{nseq, L, inner, outer_stride, inner_stride, tok_stride};
    Par p{qw, qb, kw, kb, emb, hscale};
    const int nb = L <= 16 ? 1 : 2, ks = d / 32;
#define GO(NB, KS) if (nb == NB && ks == KS) return go_fwd<NB, KS>((const bf16*)qkv, (bf16*)out, g, heads, p, out_scale, accumulate, st)
    GO(1, 1); GO(1, 2); GO(1, 3); GO(1, 4); GO(2, 1); GO(2, 2); GO(2, 3); GO(2, 4);
#undef GO
    return bf_fail_msg("bf_attn_fwd_mfma: unsupported shape", __FILE__, __LINE__);
}
// both axial passes in one launch (see attn_fwd_axial_mfma); 1 = shape not covered (the caller runs the two passes)
int bf_attn_axial_fwd_mfma(const void* qkv, void* out, int frames, int h, int w, int heads, int d, const float* qw, const float* qb, const float* kw,
                           const float* kb, const float* emb, const float* hscale_x, const float* hscale_y, const float* nw, const float* nb,
                           void* out_n, float* mean, float* rstd, float* sc, float* sh, hipStream_t st) {
    static const bool off = bf_knob("BF_ATTN_AXIAL_FUSED", 1) == 0;
    if (off || h > 16 || w > 16 || h < 1 || w < 1 || d % 32 || d > 128 || heads > 16) return 1;
    Par p{qw, qb, kw, kb, emb, hscale_x};
    const long ntile = (long)frames * heads;
    // few (frame, head) tiles (inference at batch 1-2: 96-192 on 256 CUs): 12 waves per workgroup run the h row sequences in ONE round and
    // the w column sequences in one more instead of three each -- the launch is a chain of dependent ~2 us rounds, not throughput
    static const int wpb_env = bf_knob("BF_ATTN_AXIAL_WPB", 0);
    const int ks = d / 32, wpb = (wpb_env == 4 || wpb_env == 12) ? wpb_env : (ntile <= 256 ? 12 : 4);
    const bool norm = nw != nullptr;
    const size_t tile = ((size_t)h * w * (d + 8) + 7) & ~(size_t)7;
    const size_t shm = ((size_t)wpb * 16 * (d + 16) + tile) * sizeof(bf16) + (norm ? ((size_t)(256 / (d / 8)) + 2) * d * sizeof(float) : 0);
    const int grid = (int)std::min<long>(ntile, 256L * 3);
    const AxNorm nrm{nw, nb, (bf16*)out_n, mean, rstd, sc, sh};
#define GO2(KS, NRM, W) { if (int rc = set_lds(attn_fwd_axial_mfma<KS, NRM, W>, shm)) return rc; \
        hipLaunchKernelGGL((attn_fwd_axial_mfma<KS, NRM, W>), dim3(grid), dim3(W * 64), shm, st, (const bf16*)qkv, (bf16*)out, frames, h, w, heads, p, hscale_y, nrm); }
#define GO(KS) if (ks == KS) { \
        if (norm) { if (wpb == 12) GO2(KS, true, 12) else GO2(KS, true, 4) } \
        else { if (wpb == 12) GO2(KS, false, 12) else GO2(KS, false, 4) } }
    GO(1) GO(2) GO(3) GO(4)
#undef GO
#undef GO2
    BF_CHECK_LAUNCH();
    return 0;
}
int bf_attn_bwd_mfma(const void* qkv, const void* dout, void* dqkv, long nseq, int L, long inner, long outer_stride, long inner_stride,
                     long tok_stride, int heads, int d, const float* qw, const float* qb, const float* kw, const float* kb, const float* emb,
                     const float* hscale, float* dqw, float* dqb, float* dkw, float* dkb, float* demb, float* dhscale, float out_scale,
                     int accumulate, float* ws, long ws_floats, int* rows_out, hipStream_t st) {
    Geo g{nseq, L, inner, outer_stride, inner_stride, tok_stride};
    Par p{qw, qb, kw, kb, emb, hscale};
    Grd gr{dqw, dqb, dkw, dkb, demb, dhscale};
    const int nb = L <= 16 ? 1 : 2, ks = d / 32;
#define GO(NB, KS) if (nb == NB && ks == KS) return go_bwd<NB, KS>((const bf16*)qkv, (const bf16*)dout, (bf16*)dqkv, g, heads, p, gr, out_scale, accumulate, ws, ws_floats, rows_out, st)
    GO(1, 1); GO(1, 2); GO(1, 3); GO(1, 4); GO(2, 1); GO(2, 2); GO(2, 3); GO(2, 4);
#undef GO
    return bf_fail_msg("bf_attn_bwd_mfma: unsupported shape", __FILE__, __LINE__);
}
