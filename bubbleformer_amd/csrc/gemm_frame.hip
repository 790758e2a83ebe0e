// Whole-frame row tiles: a data-gradient GEMM whose output goes straight through the InstanceNorm backward.
//
//   dy[M][N] = A[M][K] @ B[K][N]                       (A = gradient of a projection's output, B = its weight [out][in])
//   per (frame f, column n), S = 144 rows per frame:    s1 = sum_s dy, s2 = sum_s dy * xhat, xhat = (x - mean) * rstd
//   out = rstd * w * (dy - s1/S - xhat * s2/S) [+ add]   ws[(f*N + n)*2 ..] = {s1, s2}   (parameter-gradient partials)
//
// InstanceNorm2d over a 12 x 12 token frame reduces over exactly 144 consecutive rows of the token-major activation, so a
// 144-row block tile holds every row a (frame, channel) statistic needs: the column sums are taken from the MFMA accumulators
// (DPP row reduction over the 16 lanes that hold a column, one LDS exchange between the three row groups of waves) and the
// separate InstanceNorm-backward pass over dy / x / dx (bf_in_bwd: 3-4 activation sweeps and a launch) disappears.
// Replaces bf_gemm + bf_in_bwd for the four projections per block whose input is an InstanceNorm (layers/attention.py:77-78,
// 120-121, 208-210, 298-299); any other shape returns 1 and the caller runs the two-kernel path.
//
// 144 x 128 x 64 tiles, 6 waves (3 x 2; 48 x 64 per wave = 3 x 4 MFMA tiles), two LDS buffers per operand, one barrier per K-step,
// <= 168 VGPRs and 80 KB of LDS so that two workgroups share a CU.  bf16 only (the fp32 parity mode keeps the separate kernels).
#include "gemm_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

int bf_gemm_pair_scaled(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, void* out2, const float* rowfac,
                        int rows_per_group, hipStream_t st);

namespace {
using namespace bfgemm;

constexpr int FM = 144, FN = 128, FK = 64, FNT = 384;
constexpr int A_EL = FM * FK, B_EL = FK * FN, BUF_EL = A_EL + B_EL;      // 34,816 bytes per buffer
constexpr int XLD = 136;                      // staged x / add rows: 272 bytes, so the 16 rows an MFMA-layout 8-byte read touches sit 4 banks apart
constexpr int TILE_EL = FM * XLD;             // 39,168 bytes
constexpr int RED_OFF = 2 * TILE_EL * 2;      // byte offset of the cross-wave exchange [3][2][128] floats
constexpr int LDS_BYTES = RED_OFF + 3 * 2 * FN * 4;                       // 81,408: two workgroups per CU
static_assert(2 * BUF_EL * 2 <= RED_OFF, "operand buffers must not reach the exchange area");

struct FrameArgs {
    const bf16* A; long lda; const bf16* B; long ldb; int N, nk, nt;
    const bf16* x; const bf16* add; bf16* out; long ldx;
    const float* mean; const float* rstd; const float* w; float* ws;
    const float* fscale; int fdiv;      // optional per-frame-group factor on dy (stochastic depth: the branch gradient of frame f is scaled by fscale[f / fdiv])
};

__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

__global__ void __launch_bounds__(FNT, 3) gemm_inbwd_frames_kernel(FrameArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int seq = xcd_remap(blockIdx.x, gridDim.x);
    const int fidx = seq / a.nt;                        // the tile's frame
    const int m0 = fidx * FM, n0 = (seq - fidx * a.nt) * FN;

    // ---- operand staging: A tile [144][64] (3 chunks per thread), B tile [64][128] (1024 chunks: the third only for tid < 256)
    const bf16* ga[3]; int la[3];
    const bf16* gb[3]; int lb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = tid + FNT * i;
        const int row = c >> 3, c8 = (c & 7) << 3;
        ga[i] = a.A + (long)(m0 + row) * a.lda + c8;
        la[i] = lds_off<bf16, false, FK>(row, c8);
        const int kr = (c >> 4) & 63, d8 = (c & 15) << 3;
        gb[i] = a.B + (long)kr * a.ldb + n0 + d8;
        lb[i] = A_EL + lds_off<bf16, true, FN>(kr, d8);
    }
    const bool b2 = tid < FK * FN / 8 - 2 * FNT;
    Chunk<bf16> ra[3], rb[3];
    auto issue = [&](int kt) {
        const long ka = (long)kt * FK, kb = (long)kt * FK * a.ldb;
#pragma unroll
        for (int i = 0; i < 3; ++i) ra[i].load(ga[i] + ka);
        rb[0].load(gb[0] + kb);
        rb[1].load(gb[1] + kb);
        if (b2) rb[2].load(gb[2] + kb);
    };
    auto commit = [&](int buf) {
        bf16* base = lds + buf * BUF_EL;
#pragma unroll
        for (int i = 0; i < 3; ++i) ra[i].store(base + la[i]);
        rb[0].store(base + lb[0]);
        rb[1].store(base + lb[1]);
        if (b2) rb[2].store(base + lb[2]);
    };
    f32x4 acc[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int buf) {
        const bf16* cA = lds + buf * BUF_EL;
        const bf16* cB = cA + A_EL;
#pragma unroll
        for (int kk = 0; kk < FK; kk += 32) {
            bf16x8 fa[3], fb[4];
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] = frag_bf16<false, FK>(cA, wm * 48 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = frag_bf16<true, FN>(cB, wn * 64 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };

    issue(0);
    commit(0);
    if (a.nk > 1) issue(1);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt + 1 < a.nk; ++kt) {
        mma(cur);
        commit(cur ^ 1);                         // nobody reads the other buffer until the barrier below
        if (kt + 2 < a.nk) issue(kt + 2);
        __syncthreads();
        cur ^= 1;
    }
    // last K tile: the frame's x (and residual-gradient) tile is fetched while it multiplies
    Chunk<bf16> cx[6], cd[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = tid + FNT * i;
        const long off = (long)(m0 + (c >> 4)) * a.ldx + n0 + ((c & 15) << 3);
        cx[i].load(a.x + off);
    }
    mma(cur);
    if (a.add) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = tid + FNT * i;
            cd[i].load(a.add + (long)(m0 + (c >> 4)) * a.ldx + n0 + ((c & 15) << 3));
        }
    }
    __syncthreads();                             // operand tiles are dead: LDS becomes the x / add tiles
    bf16* tx = lds;
    bf16* td = lds + TILE_EL;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = tid + FNT * i;
        const int o = (c >> 4) * XLD + ((c & 15) << 3);
        cx[i].store(tx + o);
        if (a.add) cd[i].store(td + o);
    }
    __syncthreads();

    if (a.fscale) {                              // dy of this frame carries its stochastic-depth factor (one frame = one tile)
        const float m = a.fscale[fidx / a.fdiv];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= m;
    }
    // ---- column sums over the frame.  Lane (li, lg) holds row li, columns 4 lg .. 4 lg + 3 of each 16 x 16 tile.
    float* red = reinterpret_cast<float*>(smem + RED_OFF);            // [wm][s1 | s2][128]
    const int colw = wn * 64 + 4 * lg;
    const long pbase = (long)fidx * a.N + n0 + colw;
    float mu[4][4], rs[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 m4 = *reinterpret_cast<const float4*>(a.mean + pbase + j * 16);
        const float4 r4 = *reinterpret_cast<const float4*>(a.rstd + pbase + j * 16);
        mu[j][0] = m4.x; mu[j][1] = m4.y; mu[j][2] = m4.z; mu[j][3] = m4.w;
        rs[j][0] = r4.x; rs[j][1] = r4.y; rs[j][2] = r4.z; rs[j][3] = r4.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float p1[4] = {0.f, 0.f, 0.f, 0.f}, p2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bf16x4 xv = *reinterpret_cast<const bf16x4*>(tx + (wm * 48 + i * 16 + li) * XLD + colw + j * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = ((float)xv[r] - mu[j][r]) * rs[j][r];
                p1[r] += acc[i][j][r];
                p2[r] += acc[i][j][r] * xh;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { p1[r] = row16_sum(p1[r]); p2[r] = row16_sum(p2[r]); }
        if (li == 0) {
            *reinterpret_cast<float4*>(red + (wm * 2 + 0) * FN + colw + j * 16) = make_float4(p1[0], p1[1], p1[2], p1[3]);
            *reinterpret_cast<float4*>(red + (wm * 2 + 1) * FN + colw + j * 16) = make_float4(p2[0], p2[1], p2[2], p2[3]);
        }
    }
    __syncthreads();

    // ---- dx in the MFMA layout, written over the x tile (each lane reads and writes only its own elements)
    constexpr float invS = 1.f / (float)FM;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4 u = *reinterpret_cast<const float4*>(red + (g * 2 + 0) * FN + colw + j * 16);
            const float4 v = *reinterpret_cast<const float4*>(red + (g * 2 + 1) * FN + colw + j * 16);
            s1[0] += u.x; s1[1] += u.y; s1[2] += u.z; s1[3] += u.w;
            s2[0] += v.x; s2[1] += v.y; s2[2] += v.z; s2[3] += v.w;
        }
        if (wm == 0 && li == 0 && a.ws) {        // per-frame partials for the affine-parameter gradients (param_reduce.h)
            float* wp = a.ws + (pbase + j * 16) * 2;
            *reinterpret_cast<float4*>(wp) = make_float4(s1[0], s2[0], s1[1], s2[1]);
            *reinterpret_cast<float4*>(wp + 4) = make_float4(s1[2], s2[2], s1[3], s2[3]);
        }
        const float4 w4 = *reinterpret_cast<const float4*>(a.w + n0 + colw + j * 16);
        const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int o = (wm * 48 + i * 16 + li) * XLD + colw + j * 16;
            const bf16x4 xv = *reinterpret_cast<const bf16x4*>(tx + o);
            bf16x4 dv;
            if (a.add) dv = *reinterpret_cast<const bf16x4*>(td + o);
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = ((float)xv[r] - mu[j][r]) * rs[j][r];
                float t = rs[j][r] * ww[r] * (acc[i][j][r] - (s1[r] + xh * s2[r]) * invS);
                if (a.add) t += (float)dv[r];
                ov[r] = (bf16)t;
            }
            *reinterpret_cast<bf16x4*>(tx + o) = ov;
        }
    }
    __syncthreads();
    // ---- whole rows out: 16 lanes x 16 bytes = one 256-byte row segment
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = tid + FNT * i;
        Chunk<bf16> v;
        v.load(tx + (c >> 4) * XLD + ((c & 15) << 3));
        v.store(a.out + (long)(m0 + (c >> 4)) * a.ldx + n0 + ((c & 15) << 3));
    }
}


// ================================================================================================ frame-pair kernel (LDS-DMA, ping-pong)
// The same product for an even number of 144-token frames, restructured around what bounds it on the MI355X: a CU takes in
// ~50-70 GB/s from L2 however the loads are issued, so the tile is chosen for FLOP per staged byte and the schedule so that the
// matrix pipe never waits for the staging:
//  * one workgroup = TWO frames x 128 output columns (288 x 128, 90 FLOP per staged byte against 69 for 144 x 128), one per CU,
//    (M / 288) x (N / 128) of them: 192 at the bench shape -- the launch leaves a quarter of the chip to the weight-gradient kernels
//    of the side stream, whose workgroups need a whole CU's LDS as well;
//  * 8 waves as 2 x 4: wave (g, c) owns frame g x 32 columns = 9 x 2 MFMA tiles with WHOLE frame columns, so the InstanceNorm column
//    sums never leave the wave (9 register adds + a 16-lane DPP reduction) and the epilogue needs no LDS;
//  * both operands go global -> LDS by DMA (no staging registers) into a ring of three 52 KB K-step slots: A chunk [288][64] in the
//    K-contiguous swizzled image, weight chunk [64][128] in the transposing-read image (gemm_common.h: lds_off); step s + 2 is issued
//    while step s is read, each wave waits with a counted vmcnt for its own pieces of step s + 1 only;
//  * per K-step a wave runs a LOAD segment (18 + 4 fragment reads, its 6-7 DMA pieces, the counted wait) and a COMPUTE segment
//    (36 MFMA 16x16x32 = 576 matrix-pipe cycles), each closed by a raw s_barrier; waves 4-7 (frame 1) run one barrier behind waves
//    0-3 (frame 0): on every SIMD one wave multiplies while its partner loads (cdna_hip_programming.md section 5, the 8-phase
//    template's stagger; MI355X_MICROARCH.md "Two waves per SIMD");
//  * epilogue from registers: two neighbouring accumulator tiles are exchanged between lane rows (v_permlane16_swap), after which a
//    lane owns 8 consecutive columns of 9 rows: 16-byte loads of x / the residual gradient and 16-byte stores (64-byte row segments).
// MODE 0: InstanceNorm backward behind the product (bf_gemm_inbwd_frames); MODE 1: out = product (+ add) (bf_gemm_pair_try: fc1's
// data gradient, any 288-row tiling).
constexpr int PM = 288, PN = 128, PK = 64;
constexpr int PA_BYTES = PM * PK * 2, PB_BYTES = PK * PN * 2, PSLOT_BYTES = PA_BYTES + PB_BYTES, PNSLOT = 3;      // 36,864 + 16,384 = 53,248; x 3 = 159,744
constexpr int PA_PIECES = PA_BYTES / 1024, PB_PIECES = PB_BYTES / 1024;                                           // 36, 16

struct FwdNorm { const float *w, *b, *g; int gdiv; float *mean, *rstd, *sc, *sh; const bf16* resid; bf16* out; };
struct PairArgs {
    const bf16* A; long lda; const bf16* B; long ldb; int N, nk, nt;
    const bf16* x; const bf16* add; bf16* out; long ldx;
    const float* mean; const float* rstd; const float* w; float* ws;
    const float* fscale; int fdiv;
    bf16* out2;                         // MODE 1, optional: out2 = out * fscale[row / fdiv] (the stochastic-depth-scaled copy the next kernels read)
    int whole;                          // MODE 0: 1 = the tile is ONE frame of 288 tokens (24 x 12 grids: BASELINE configs[3]) instead of two of 144
    // MODE 0, optional chain (two-frame tiles only): `out` is also the incoming gradient of ANOTHER InstanceNorm (statistics cmean / crstd of
    // cz, affine weight cw, optional post scale cg[(frame / cgdiv)][column]) whose backward is the next kernel in line -- it is applied here,
    // to the rows as stored: cdz = crstd cw cg (out - (s1 + xh s2) / S), partials {s1, s2} to cws (the layout of ws)
    const bf16* cz; bf16* cdz; const float *cmean, *crstd, *cw, *cg; int cgdiv; float* cws;
    // MODE 3 (the forward twin): out = lin(product) [+ add], lin(v) = ((v + bias) * cs + ch) * fscale[frame / fdiv]; then up to two InstanceNorms
    // of the rows as stored, over the whole frame columns a wave holds: n1 (optional: y = resid + g * IN(out), the MLP-branch norm behind fc2) and
    // n2 (optional: the NEXT stage's opening norm of the last tensor written, xn = IN(y) -- statistics, sc / sh and the normalised copy)
    const float *bias, *cs, *ch;
    FwdNorm n1, n2;
    // MODE 0 / 2, optional: a second copy of `out` scaled per frame group, out_s = out * fscale_s[frame / fdiv_s] -- the stochastic-depth factor the
    // NEXT stage's backward applies to this gradient before anything else reads it (one elementwise launch and a read of `out` less)
    bf16* out_s; const float* fscale_s; int fdiv_s;
};

// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a per-lane 32-bit byte offset: one offset register serves every piece of a wave
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// ---- InstanceNorm statistics of the frame columns a wave of the frame-pair kernel holds, summed in EXACTLY the order in_stats_kernel (norm.hip)
// sums a 144-token frame -- so that a norm folded into a producer's epilogue leaves the bits a separate launch would.  There thread (rg, lc) of 32
// row groups adds its rows rg, rg + 32, .. in order, the row groups meet by a butterfly over rg bits 0, 1, 2 inside a wave, and the four waves
// (rg >> 3) are added in wave order.  Here lane li holds rows 16 i + li: the even i are row group li, the odd i row group li + 16.
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float stats_tree(float pe, float po, bool lo) {
    pe = pe + dpp_get<0xB1>(pe); po = po + dpp_get<0xB1>(po);        // rg bit 0 (lane xor 1)
    pe = pe + dpp_get<0x4E>(pe); po = po + dpp_get<0x4E>(po);        // rg bit 1 (lane xor 2)
    pe = pe + dpp_get<0x141>(pe); po = po + dpp_get<0x141>(po);      // rg bit 2 (the other quad of the half row)
    const float eo = dpp_get<0x140>(pe), oo = dpp_get<0x140>(po);    // the other half row's totals
    float t = 0.f;
    t += lo ? pe : eo;                                               // wave 0: row groups 0..7
    t += lo ? eo : pe;                                               // wave 1: 8..15
    t += lo ? po : oo;                                               // wave 2: 16..23
    t += lo ? oo : po;                                               // wave 3: 24..31
    return t;
}
__device__ __forceinline__ float bfq(const uint4& v, int q) { return (float)__builtin_bit_cast(bf16x8, v)[q]; }
// mean and 1 / sqrt(var + eps) of 8 columns over the frame's 144 rows (y[i]: row 16 i + li, packed bf16), two passes
__device__ __forceinline__ void frame_stats(const uint4 (&y)[9], bool lo, float (&mu)[8], float (&r)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float pe = 0.f, po = 0.f;
#pragma unroll
        for (int i = 0; i < 9; i += 2) pe += bfq(y[i], q);
#pragma unroll
        for (int i = 1; i < 9; i += 2) po += bfq(y[i], q);
        mu[q] = stats_tree(pe, po, lo) / 144.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float pe = 0.f, po = 0.f;
#pragma unroll
        for (int i = 0; i < 9; i += 2) { const float d = bfq(y[i], q) - mu[q]; pe = fmaf(d, d, pe); }
#pragma unroll
        for (int i = 1; i < 9; i += 2) { const float d = bfq(y[i], q) - mu[q]; po = fmaf(d, d, po); }
        r[q] = rsqrtf(stats_tree(pe, po, lo) / 144.f + BF_IN_EPS);
    }
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

template <int MODE, int GRP>
__device__ __forceinline__ void pair_body(const PairArgs& a, unsigned char* smem, int lane, int w4, int m0, int n0, int fidx0) {
    constexpr bool INBWD = MODE == 0 || MODE == 2 || MODE == 5;      // the InstanceNorm backward behind the product (x rows staged in LDS); 5 = 0 + the scaled second copy
    constexpr bool FWD = MODE == 3 || MODE == 4;        // the forward twin (4: with the MLP-branch norm n1 -- its own instantiation, so that profiles tell the two apart)
    constexpr int PG = GRP == 0 ? 7 : 6;            // DMA pieces per wave per K-step: A pieces {w, w+8, w+16, w+24} (+ 32 + w for waves 0-3), B pieces {w, w+8}
    const int wave = GRP * 4 + w4;
    const unsigned ring = lds_addr(smem);
    // ---- DMA geometry.  A piece p = rows 8p .. 8p+7 of the [288][64] chunk; lane -> row 8p + (lane >> 3), LDS 16-byte chunk (lane & 7)
    // holds global chunk (lane & 7) ^ ((row >> 1) & 7); p = wave + 8t, so (row >> 1) & 7 = (4 (wave & 1) + (lane >> 4)) & 7 for every t.
    const unsigned voffA = (unsigned)(((long)(lane >> 3) * a.lda + 8 * ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7))) * 2);
    // B piece p = rows 4p .. 4p+3 of the [64][128] chunk; lane -> row 4p + (lane >> 4), LDS chunk (lane & 15) holds global chunk
    // (lane & 15) ^ (2 key(row)), key(r) = (r & 3) | ((r >> 1) & 4); p = wave + 8t: key = (lane >> 4) | ((2 wave + (lane >> 5)) & 4)
    const int bkey = (lane >> 4) | ((2 * wave + (lane >> 5)) & 4);
    const unsigned voffB = (unsigned)(((long)(lane >> 4) * a.ldb + 8 * ((lane & 15) ^ (bkey << 1))) * 2);
    const bf16* sA = a.A + (long)(m0 + 8 * wave) * a.lda;                  // piece t: + 64 t rows; K-step: + 64 elements
    const bf16* sB = a.B + (long)(4 * wave) * a.ldb + n0;                  // piece t: + 32 t rows; K-step: + 64 rows
    const long pieceA = 64 * a.lda, pieceB = 32 * a.ldb, stepB = 64 * a.ldb;
    auto issue = [&](int slot) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring + (unsigned)slot * (unsigned)PSLOT_BYTES + (unsigned)wave * 1024u);
#pragma unroll
        for (int t = 0; t < 4; ++t) glds16_s(sA + t * pieceA, voffA, dst + (unsigned)t * 8192u);
        if constexpr (GRP == 0) glds16_s(sA + 4 * pieceA, voffA, dst + 32768u);
#pragma unroll
        for (int t = 0; t < 2; ++t) glds16_s(sB + t * pieceB, voffB, dst + (unsigned)PA_BYTES + (unsigned)t * 8192u);
        sA += PK; sB += stepB;
    };

    f32x4 acc[9][2];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue operands: lane (li, lg) ends up with rows 16 i + li of its frame, columns c8 .. c8 + 7 of the wave's 32-column strip
    const int li = lane & 15, lg = lane >> 4;
    const int c8 = (lg & 1) * 16 + (lg >> 1) * 8;
    const int col0 = n0 + 32 * w4 + c8;
    const long row0 = (long)m0 + 144 * GRP + li;
    // MODE 0: the frame's x rows [144][128] go by DMA into a ring slot that is free during the last K-step (frame 0: the slot of step
    // nk - 2, frame 1: the slot of step nk - 3 -- both were read by every wave before this wave's last load segment starts); no register
    // is written asynchronously.  Piece p = w4 + 4t (t < 9) = rows 4p .. 4p+3 of the frame, 256 bytes each.
    const unsigned voffX = (unsigned)(((long)(lane >> 4) * a.ldx + 8 * (lane & 15)) * 2);
    const bf16* sX = INBWD ? a.x + ((long)m0 + 144 * GRP + 4 * w4) * a.ldx + n0 : nullptr;
    const long pieceX = 16 * a.ldx;
    int xslot = 0;

    const int nk = a.nk;
    // ---- prologue: steps 0 and 1 in flight, step 0 landed for every wave
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 1) wait_vm<PG>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if constexpr (GRP == 1) __builtin_amdgcn_s_barrier();          // one barrier behind waves 0-3 from here on

    int slot = 0;
    // epilogue operands that come by plain loads (the residual gradient rows, the frame's statistics and the norm weight): requested at
    // the top of the LAST K-step, so that they travel under its fragment reads and MFMAs (the last step is a separate instance of the
    // step body: nothing is carried through the loop in registers)
    uint4 ad[9];
    float mu[8], rs[8], ww[8];
    const int fidx = (INBWD && a.whole) ? fidx0 : fidx0 + GRP;
    float rsf = 1.f;                                   // MODE 3: the frame's stochastic-depth factor
    const long pbase = MODE != 1 ? (long)fidx * a.N + col0 : 0;
    auto kstep = [&](auto last_tag, int s) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        // ======== load segment: fragments of step s; DMA of step s + 2 into the slot that held step s - 1 (every wave finished reading
        // it before the barrier that ended ITS load segment of s - 1); the counted wait for this wave's pieces of step s + 1
        const bf16* cA = reinterpret_cast<const bf16*>(smem + (size_t)slot * PSLOT_BYTES);
        const bf16* cB = reinterpret_cast<const bf16*>(smem + (size_t)slot * PSLOT_BYTES + PA_BYTES);
        if constexpr (LAST) {
            if (MODE != 2 && MODE != 5 && a.add) {
#pragma unroll
                for (int i = 0; i < 9; ++i) ad[i] = *reinterpret_cast<const uint4*>(a.add + (row0 + 16 * i) * a.ldx + col0);
            }
            if constexpr (FWD) {                        // mu / rs / ww hold bias / column scale / column shift here
#pragma unroll
                for (int q = 0; q < 8; ++q) { mu[q] = 0.f; rs[q] = 1.f; ww[q] = 0.f; }
                if (a.bias) load8(a.bias + col0, mu);
                if (a.cs) { load8(a.cs + col0, rs); load8(a.ch + col0, ww); }
                if (a.fscale) rsf = a.fscale[fidx / a.fdiv];
            }
            if constexpr (INBWD) {
                const float4 m0 = *reinterpret_cast<const float4*>(a.mean + pbase), m1 = *reinterpret_cast<const float4*>(a.mean + pbase + 4);
                const float4 r0 = *reinterpret_cast<const float4*>(a.rstd + pbase), r1 = *reinterpret_cast<const float4*>(a.rstd + pbase + 4);
                const float4 w0 = *reinterpret_cast<const float4*>(a.w + col0), w1 = *reinterpret_cast<const float4*>(a.w + col0 + 4);
                mu[0] = m0.x; mu[1] = m0.y; mu[2] = m0.z; mu[3] = m0.w; mu[4] = m1.x; mu[5] = m1.y; mu[6] = m1.z; mu[7] = m1.w;
                rs[0] = r0.x; rs[1] = r0.y; rs[2] = r0.z; rs[3] = r0.w; rs[4] = r1.x; rs[5] = r1.y; rs[6] = r1.z; rs[7] = r1.w;
                ww[0] = w0.x; ww[1] = w0.y; ww[2] = w0.z; ww[3] = w0.w; ww[4] = w1.x; ww[5] = w1.y; ww[6] = w1.z; ww[7] = w1.w;
            }
        }
        bf16x8 fa[2][9], fb[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 9; ++i) fa[kk][i] = frag_bf16<false, PK>(cA, 144 * GRP + 16 * i, 32 * kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[kk][j] = frag_bf16<true, PN>(cB, 32 * w4 + 16 * j, 32 * kk, lane);
        }
        const int sl2 = slot == 0 ? 2 : slot - 1;
        if constexpr (!LAST) {
            if (s + 2 < nk) { issue(sl2); wait_vm<PG>(); }
            else wait_vm<0>();
        } else if constexpr (INBWD) {               // last step: the x rows land under its MFMAs (waited for before the barrier that ends them)
            xslot = GRP == 0 ? sl2 : (slot == PNSLOT - 1 ? 0 : slot + 1);
            const unsigned dst = __builtin_amdgcn_readfirstlane(ring + (unsigned)xslot * (unsigned)PSLOT_BYTES + (unsigned)w4 * 1024u);
#pragma unroll
            for (int t = 0; t < 9; ++t) glds16_s(sX + t * pieceX, voffX, dst + (unsigned)t * 4096u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the reads are done before the barrier: the slot may be refilled after it
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ======== compute segment
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (LAST && INBWD && GRP == 1) wait_vm<0>();           // this wave's x pieces: visible to the workgroup after the barrier below
        __builtin_amdgcn_s_barrier();
        slot = slot == PNSLOT - 1 ? 0 : slot + 1;
    };
    for (int s = 0; s + 1 < nk; ++s) kstep(std::false_type{}, s);
    kstep(std::true_type{}, nk - 1);
    if constexpr (GRP == 0) {
        if constexpr (INBWD) wait_vm<0>();                          // ... and likewise for waves 0-3
        __builtin_amdgcn_s_barrier();                               // pairs with the extra barrier of waves 4-7: everybody's x pieces have landed
    }

    constexpr bool chain = MODE == 2;      // its own instantiation: the code below costs the unchained kernel 13 spilled registers and 1.5 us per launch
    if constexpr (chain || MODE == 5) {    // the chained form has no registers for the residual-gradient rows during the last K-step (they spilled, beside
        // hand-counted vmcnt): it asks for them here, and they travel under the column-sum pass below instead of under the MFMAs
        if (a.add) {
#pragma unroll
            for (int i = 0; i < 9; ++i) ad[i] = *reinterpret_cast<const uint4*>(a.add + (row0 + 16 * i) * a.ldx + col0);
        }
    }
    // ---- epilogue.  acc[i][j]: row 16 i + li, columns 16 j + 4 lg .. + 3.  After exchanging the odd lane rows of tile 0 with the even
    // lane rows of tile 1 a lane holds 8 consecutive columns at c8.
    float v[9][8];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][0][r]), __float_as_uint(acc[i][1][r]), false, false);
            v[i][r] = __uint_as_float(sw[0]); v[i][4 + r] = __uint_as_float(sw[1]);
        }
    if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            bf16x8 o;
            if (a.add) {
                const bf16x8 d8 = __builtin_bit_cast(bf16x8, ad[i]);
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)(v[i][q] + (float)d8[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)v[i][q];
            }
            *reinterpret_cast<bf16x8*>(a.out + (row0 + 16 * i) * a.ldx + col0) = o;
            if (a.out2) {                            // the scaled copy is the scaled ROUNDED value: what a separate pass over `out` would write
                const float m = a.fscale[(row0 + 16 * i) / a.fdiv];
                bf16x8 o2;
#pragma unroll
                for (int q = 0; q < 8; ++q) o2[q] = (bf16)((float)o[q] * m);
                *reinterpret_cast<bf16x8*>(a.out2 + (row0 + 16 * i) * a.ldx + col0) = o2;
            }
        }
    } else if constexpr (FWD) {
        // ---- the forward twin.  Stage A: the linear epilogue of the streaming kernels (same expressions: same bits), rows rounded and stored
        const bool lo = li < 8;
        const bool late_resid = MODE == 4 && a.n1.resid && !a.add;      // the second norm's residual rows travel under stage A and the statistics
        if (late_resid) {
#pragma unroll
            for (int i = 0; i < 9; ++i) ad[i] = *reinterpret_cast<const uint4*>(a.n1.resid + (row0 + 16 * i) * a.ldx + col0);
        }
        uint4 y[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            bf16x8 o;
            if (a.add) {
                const bf16x8 d8 = __builtin_bit_cast(bf16x8, ad[i]);
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)epi_lin_add(v[i][q], mu[q], rs[q], ww[q], rsf, a.cs != nullptr, (float)d8[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)epi_lin(v[i][q], mu[q], rs[q], ww[q], rsf, a.cs != nullptr);
            }
            *reinterpret_cast<bf16x8*>(a.out + (row0 + 16 * i) * a.ldx + col0) = o;
            y[i] = __builtin_bit_cast(uint4, o);
        }
        // ---- stage B: y = resid + g * InstanceNorm(out) over the frame (statistics of the rows as stored)
        if constexpr (MODE == 4) {
            float m1[8], r1[8], aa[8], ss[8];
            load8(a.n1.w + col0, aa); load8(a.n1.b + col0, ss);
            float gg[8];
            if (a.n1.g) load8(a.n1.g + (long)(fidx / a.n1.gdiv) * a.N + col0, gg);
            if (a.n1.resid && !late_resid) {
#pragma unroll
                for (int i = 0; i < 9; ++i) ad[i] = *reinterpret_cast<const uint4*>(a.n1.resid + (row0 + 16 * i) * a.ldx + col0);
            }
            frame_stats(y, lo, m1, r1);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float s0 = r1[q] * aa[q];
                float t0 = fmaf(-m1[q], s0, ss[q]);
                if (a.n1.g) { s0 *= gg[q]; t0 = fmaf(t0, gg[q], 0.f); }
                aa[q] = s0; ss[q] = t0;
            }
            if (li == 0) { store8(a.n1.mean + pbase, m1); store8(a.n1.rstd + pbase, r1); store8(a.n1.sc + pbase, aa); store8(a.n1.sh + pbase, ss); }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const bf16x8 d8 = __builtin_bit_cast(bf16x8, ad[i]);
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)(fmaf(bfq(y[i], q), aa[q], ss[q]) + (a.n1.resid ? (float)d8[q] : 0.f));
                *reinterpret_cast<bf16x8*>(a.n1.out + (row0 + 16 * i) * a.ldx + col0) = o;
                y[i] = __builtin_bit_cast(uint4, o);
            }
        }
        // ---- stage C: the next stage's opening norm of the rows just stored
        if (a.n2.w) {
            float m2[8], r2[8], aa[8], ss[8];
            load8(a.n2.w + col0, aa); load8(a.n2.b + col0, ss);
            frame_stats(y, lo, m2, r2);
#pragma unroll
            for (int q = 0; q < 8; ++q) { aa[q] = r2[q] * aa[q]; ss[q] = fmaf(-m2[q], aa[q], ss[q]); }
            if (li == 0) { store8(a.n2.mean + pbase, m2); store8(a.n2.rstd + pbase, r2); store8(a.n2.sc + pbase, aa); store8(a.n2.sh + pbase, ss); }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)(fmaf(bfq(y[i], q), aa[q], ss[q]) + 0.f);
                *reinterpret_cast<bf16x8*>(a.n2.out + (row0 + 16 * i) * a.ldx + col0) = o;
            }
        }
    } else {
        if (a.fscale) {                              // dy of this frame carries its stochastic-depth factor
            const float m = a.fscale[fidx / a.fdiv];
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[i][q] *= m;
        }
        bf16x8 xr[9];                                // this lane's rows 16 i + li, columns 32 w4 + c8 .. + 7 of the staged frame
        {
            const unsigned char* xt = smem + (size_t)xslot * PSLOT_BYTES + (size_t)li * 256 + (size_t)(32 * w4 + c8) * 2;
#pragma unroll
            for (int i = 0; i < 9; ++i) xr[i] = *reinterpret_cast<const bf16x8*>(xt + (size_t)i * 4096);
        }
        float s1[8], s2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const bf16x8 x8 = xr[i];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float xh = ((float)x8[q] - mu[q]) * rs[q];
                s1[q] += v[i][q];
                s2[q] += v[i][q] * xh;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1[q] = row16_sum(s1[q]); s2[q] = row16_sum(s2[q]); }
        if (a.whole) {                               // one 288-token frame per tile: the two wave groups hold its two halves -- exchange the
            // half sums through the other ring slot (every slot is free here; the x rows sit in xslot), group 0's first for both: same bits
            const int cslot = slot == 0 ? PNSLOT - 1 : slot - 1;      // the last step's operand slot: neither group's x slot
            float* ex = reinterpret_cast<float*>(smem + (size_t)cslot * PSLOT_BYTES) + ((GRP * 4 + w4) * 4 + lg) * 16;
            if (li == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { ex[q] = s1[q]; ex[8 + q] = s2[q]; }
            }
            __syncthreads();
            const float* e0 = reinterpret_cast<const float*>(smem + (size_t)cslot * PSLOT_BYTES) + ((0 * 4 + w4) * 4 + lg) * 16;
            const float* e1 = e0 + 4 * 4 * 16;
#pragma unroll
            for (int q = 0; q < 8; ++q) { s1[q] = e0[q] + e1[q]; s2[q] = e0[8 + q] + e1[8 + q]; }
        }
        if (li == 0 && a.ws && (GRP == 0 || !a.whole)) {      // per-frame partials for the affine-parameter gradients (param_reduce.h)
            float* wp = a.ws + pbase * 2;
#pragma unroll
            for (int q = 0; q < 8; q += 2) *reinterpret_cast<float4*>(wp + 2 * q) = make_float4(s1[q], s2[q], s1[q + 1], s2[q + 1]);
        }
        const float invS = a.whole ? 1.f / 288.f : 1.f / 144.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const bf16x8 x8 = xr[i];
            const bf16x8 d8 = __builtin_bit_cast(bf16x8, ad[i]);
            bf16x8 o;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float xh = ((float)x8[q] - mu[q]) * rs[q];
                float t = rs[q] * ww[q] * (v[i][q] - (s1[q] + xh * s2[q]) * invS);
                if (a.add) t += (float)d8[q];
                o[q] = (bf16)t;
            }
            *reinterpret_cast<bf16x8*>(a.out + (row0 + 16 * i) * a.ldx + col0) = o;
            if constexpr (MODE == 5) {               // the scaled copy is the scaled ROUNDED value: what a separate pass over `out` would write
                const float m2 = a.fscale_s[fidx / a.fdiv_s];
                bf16x8 o2;
#pragma unroll
                for (int q = 0; q < 8; ++q) o2[q] = (bf16)((float)o[q] * m2);
                *reinterpret_cast<bf16x8*>(a.out_s + (row0 + 16 * i) * a.ldx + col0) = o2;
            }
        }
        if constexpr (chain) {      // the next InstanceNorm backward in line, on the rows just stored (144-token frames: whole frame columns sit in this wave)
            // its rows are requested only now: before this point the kernel has no registers to park them in (248 of 256 in use), and a
            // spill next to hand-counted vmcnt is not an option; the ~2 us of exposed latency cost less than the launch they replace
            // (the rows just stored are read back -- same lane, same addresses, L2-hot -- rather than kept)
            __builtin_amdgcn_sched_barrier(0);      // ... and the scheduler must not lift these loads into the epilogue above
            uint4 zr[9];
            bf16x8 ov[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                zr[i] = *reinterpret_cast<const uint4*>(a.cz + (row0 + 16 * i) * a.ldx + col0);
                ov[i] = *reinterpret_cast<const bf16x8*>(a.out + (row0 + 16 * i) * a.ldx + col0);
            }
            const long cb = (long)fidx * a.N + col0;
            float m3[8], r3[8], k3[8];
            {
                const float4 m0 = *reinterpret_cast<const float4*>(a.cmean + cb), m1 = *reinterpret_cast<const float4*>(a.cmean + cb + 4);
                const float4 r0 = *reinterpret_cast<const float4*>(a.crstd + cb), r1 = *reinterpret_cast<const float4*>(a.crstd + cb + 4);
                const float4 w0 = *reinterpret_cast<const float4*>(a.cw + col0), w1 = *reinterpret_cast<const float4*>(a.cw + col0 + 4);
                m3[0] = m0.x; m3[1] = m0.y; m3[2] = m0.z; m3[3] = m0.w; m3[4] = m1.x; m3[5] = m1.y; m3[6] = m1.z; m3[7] = m1.w;
                r3[0] = r0.x; r3[1] = r0.y; r3[2] = r0.z; r3[3] = r0.w; r3[4] = r1.x; r3[5] = r1.y; r3[6] = r1.z; r3[7] = r1.w;
                k3[0] = w0.x; k3[1] = w0.y; k3[2] = w0.z; k3[3] = w0.w; k3[4] = w1.x; k3[5] = w1.y; k3[6] = w1.z; k3[7] = w1.w;
                if (a.cg) {
                    const long gbase = (long)(fidx / a.cgdiv) * a.N + col0;
                    const float4 g0 = *reinterpret_cast<const float4*>(a.cg + gbase), g1 = *reinterpret_cast<const float4*>(a.cg + gbase + 4);
                    k3[0] *= g0.x; k3[1] *= g0.y; k3[2] *= g0.z; k3[3] *= g0.w; k3[4] *= g1.x; k3[5] *= g1.y; k3[6] *= g1.z; k3[7] *= g1.w;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) k3[q] *= r3[q];
            }
            float c1[8], c2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { c1[q] = 0.f; c2[q] = 0.f; }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const bf16x8 z8 = __builtin_bit_cast(bf16x8, zr[i]);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float dd = (float)ov[i][q], xh = ((float)z8[q] - m3[q]) * r3[q];
                    c1[q] += dd;
                    c2[q] += dd * xh;
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { c1[q] = row16_sum(c1[q]); c2[q] = row16_sum(c2[q]); }
            if (li == 0 && a.cws) {
                float* wp = a.cws + cb * 2;
#pragma unroll
                for (int q = 0; q < 8; q += 2) *reinterpret_cast<float4*>(wp + 2 * q) = make_float4(c1[q], c2[q], c1[q + 1], c2[q + 1]);
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const bf16x8 z8 = __builtin_bit_cast(bf16x8, zr[i]);
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float dd = (float)ov[i][q], xh = ((float)z8[q] - m3[q]) * r3[q];
                    o[q] = (bf16)(k3[q] * (dd - (c1[q] + xh * c2[q]) * (1.f / 144.f)));
                }
                *reinterpret_cast<bf16x8*>(a.cdz + (row0 + 16 * i) * a.ldx + col0) = o;
            }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(512) gemm_pair_kernel(PairArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seq = xcd_remap(blockIdx.x, gridDim.x);       // the column blocks of a frame pair run back to back on one XCD: its rows come through one L2
    const int fp = seq / a.nt;
    const int m0 = fp * PM, n0 = (seq - fp * a.nt) * PN;
    const int f0 = ((MODE == 0 || MODE == 2 || MODE == 5) && a.whole) ? fp : 2 * fp;
    if (wave < 4) pair_body<MODE, 0>(a, smem, lane, wave, m0, n0, f0);
    else pair_body<MODE, 1>(a, smem, lane, wave - 4, m0, n0, f0);
}

template <int MODE>
int launch_pair(const PairArgs& a, int M, hipStream_t st) {
    static BfPerDeviceOnce attr_once; bool& attr_done = attr_once.flag();
    if (!attr_done) {
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pair_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, PNSLOT * PSLOT_BYTES);
        if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_pair_kernel<MODE>, dim3((unsigned)((M / PM) * a.nt)), dim3(512), PNSLOT * PSLOT_BYTES, st, a);
    BF_CHECK_LAUNCH();
    return 0;
}
bool pair_shape_ok(int M, int N, int K, int64_t lda, int64_t ldb) {
    static const bool off = bf_knob("BF_PAIR", 1) == 0;      // A/B against the kernels it replaces
    if (off) return false;
    return M > 0 && M % PM == 0 && N > 0 && N % PN == 0 && K >= PK && K % PK == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
           (long)PM * lda * 2 < (1L << 31) && (long)PK * ldb * 2 < (1L << 31);
}

}  // namespace

struct InbwdChain { const void* z; void* dz; const float *mean, *rstd, *w, *g; int gdiv; float* ws; };
struct InbwdScaled { void* out_s; const float* f; int fdiv; };
static int inbwd_frames_impl2(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                              const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                              const float* fscale, int fdiv, const InbwdChain* ch, const InbwdScaled* sc2, bf_stream_t stream);
static int inbwd_frames_impl(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                             const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                             const float* fscale, int fdiv, const InbwdChain* ch, bf_stream_t stream);
extern "C" int bf_gemm_inbwd_frames(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                                    const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                                    const float* fscale, int fdiv, bf_stream_t stream) {
    return inbwd_frames_impl(dtype, M, N, K, A, lda, B, ldb, x, add, out, S, mean, rstd, w, ws, fscale, fdiv, nullptr, stream);
}
// ... with the backward of a SECOND InstanceNorm applied to `out` in the same launch (the norm whose output gradient `out` is: the spatial
// stage's MLP-branch norm behind a temporal stage's norm1): dz = crstd cw cg (out - (s1 + xh s2) / S), xh = (z - cmean) crstd, partials {s1, s2}
// to cws.  Returns 1 (nothing launched) where the frame-pair kernel with two 144-token frames per tile does not apply.
extern "C" int bf_gemm_inbwd_frames_chain(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                                          const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                                          const float* fscale, int fdiv, const void* cz, void* cdz, const float* cmean, const float* crstd,
                                          const float* cw, const float* cg, int cgdiv, float* cws, bf_stream_t stream) {
    BF_REQUIRE(cz && cdz && cmean && crstd && cw, "bf_gemm_inbwd_frames_chain: null pointer");
    const InbwdChain ch{cz, cdz, cmean, crstd, cw, cg, cgdiv > 0 ? cgdiv : 1, cws};
    return inbwd_frames_impl(dtype, M, N, K, A, lda, B, ldb, x, add, out, S, mean, rstd, w, ws, fscale, fdiv, &ch, stream);
}
static int inbwd_frames_impl(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                             const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                             const float* fscale, int fdiv, const InbwdChain* ch, bf_stream_t stream) {
    return inbwd_frames_impl2(dtype, M, N, K, A, lda, B, ldb, x, add, out, S, mean, rstd, w, ws, fscale, fdiv, ch, nullptr, stream);
}
// library-internal (model.hip): bf_gemm_inbwd_frames with a second, per-frame-group scaled copy of `out` (out_s = out * f[frame / fdiv]); returns 1
// (nothing launched) where the frame-pair kernel does not take the shape -- the caller then runs the plain entry point and scales separately
int bf_gemm_inbwd_frames_scaled(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                                const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                                const float* fscale, int fdiv, void* out_s, const float* f_s, int fdiv_s, hipStream_t stream) {
    if (!out_s || !f_s || (((uintptr_t)out_s) & 15)) return 1;
    const InbwdScaled sc2{out_s, f_s, fdiv_s > 0 ? fdiv_s : 1};
    return inbwd_frames_impl2(dtype, M, N, K, A, lda, B, ldb, x, add, out, S, mean, rstd, w, ws, fscale, fdiv, nullptr, &sc2, (bf_stream_t)stream);
}
static int inbwd_frames_impl2(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                              const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                              const float* fscale, int fdiv, const InbwdChain* ch, const InbwdScaled* sc2, bf_stream_t stream) {
    BF_REQUIRE(A && B && x && out && mean && rstd && w, "bf_gemm_inbwd_frames: null pointer");
    static const bool off = bf_knob("BF_FUSE_INBWD", 1) == 0;
    if (off || dtype != BF_DTYPE_BF16 || (S != FM && S != PM) || M <= 0 || M % S || N <= 0 || N % FN || K < FK || K % FK || lda % 8 || ldb % 8) return 1;
    if (pair_shape_ok(M, N, K, lda, ldb) && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)x | (uintptr_t)add | (uintptr_t)out) & 15) == 0) {
        PairArgs a;      // an even number of 144-token frames (two per workgroup) or 288-token frames (one per workgroup): LDS-DMA ping-pong
        a.whole = S == PM;
        a.A = (const bf16*)A; a.lda = lda; a.B = (const bf16*)B; a.ldb = ldb; a.N = N; a.nk = K / PK; a.nt = N / PN;
        a.x = (const bf16*)x; a.add = (const bf16*)add; a.out = (bf16*)out; a.ldx = N;
        a.mean = mean; a.rstd = rstd; a.w = w; a.ws = ws; a.fscale = fscale; a.fdiv = fdiv > 0 ? fdiv : 1; a.out2 = nullptr;
        a.cz = nullptr; a.cdz = nullptr; a.cmean = a.crstd = a.cw = a.cg = nullptr; a.cgdiv = 1; a.cws = nullptr;
        a.bias = a.cs = a.ch = nullptr; memset(&a.n1, 0, sizeof(a.n1)); memset(&a.n2, 0, sizeof(a.n2));
        a.out_s = nullptr; a.fscale_s = nullptr; a.fdiv_s = 1;
        if (sc2) { a.out_s = (bf16*)sc2->out_s; a.fscale_s = sc2->f; a.fdiv_s = sc2->fdiv; }
        if (ch) {
            if (a.whole || (((uintptr_t)ch->z | (uintptr_t)ch->dz) & 15)) return 1;
            a.cz = (const bf16*)ch->z; a.cdz = (bf16*)ch->dz; a.cmean = ch->mean; a.crstd = ch->rstd; a.cw = ch->w; a.cg = ch->g; a.cgdiv = ch->gdiv; a.cws = ch->ws;
        }
        BfProfScope prof((hipStream_t)stream, ch ? "gemm_pair<inbwd,chain>" : sc2 ? "gemm_pair<inbwd,scaled>" : "gemm_pair<inbwd>", 2.0 * M * N * K,
                         2.0 * ((double)M * K + (double)N * K + (double)M * N * ((add ? 3 : 2) + (ch ? 2 : 0) + (sc2 ? 1 : 0))));
        if (sc2) { if (ch) return 1; return launch_pair<5>(a, M, (hipStream_t)stream); }
        return ch ? launch_pair<2>(a, M, (hipStream_t)stream) : launch_pair<0>(a, M, (hipStream_t)stream);
    }
    if (ch || sc2) return 1;
    if (S != FM) return 1;
    static BfPerDeviceOnce attr_once;
    if (bool& done = attr_once.flag(); !done) {
        const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_inbwd_frames_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (attr != hipSuccess) return bf_fail(attr, __FILE__, __LINE__);
        done = true;
    }
    hipStream_t st = (hipStream_t)stream;
    FrameArgs a;
    a.A = (const bf16*)A; a.lda = lda; a.B = (const bf16*)B; a.ldb = ldb; a.N = N; a.nk = K / FK; a.nt = N / FN;
    a.x = (const bf16*)x; a.add = (const bf16*)add; a.out = (bf16*)out; a.ldx = N;
    a.mean = mean; a.rstd = rstd; a.w = w; a.ws = ws;
    a.fscale = fscale; a.fdiv = fdiv > 0 ? fdiv : 1;
    BfProfScope prof(st, "gemm_inbwd_frames<bf16>", 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K + (double)M * N * (add ? 3 : 2)));
    hipLaunchKernelGGL(gemm_inbwd_frames_kernel, dim3((unsigned)((M / FM) * a.nt)), dim3(FNT), LDS_BYTES, st, a);
    BF_CHECK_LAUNCH();
    return 0;
}

// out[M][N] = A[M][K] @ B[K][N] (+ add) on the frame-pair kernel: A K-contiguous, B outer-contiguous (a weight [out][in] used as the
// data gradient's operand), 288-row tiles.  0 = handled, 1 = not covered, < 0 = error.
int bf_gemm_pair_try(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, hipStream_t st) {
    return bf_gemm_pair_scaled(M, N, K, A, B, E, nullptr, nullptr, 1, st);
}
// ... with an optional second output out2[m][:] = C[m][:] * rowfac[m / rows_per_group] (library-internal: the gradient entering a branch
// under stochastic depth, written by the kernel that produces the gradient instead of by a pass of its own)
int bf_gemm_pair_scaled(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, void* out2, const float* rowfac,
                        int rows_per_group, hipStream_t st) {
    if (A->layout != BF_LAY_KC || B->layout != BF_LAY_XC || A->pro != BF_PRO_NONE || B->pro != BF_PRO_NONE) return 1;
    if (A->gw > 0 || A->seglen > 0 || B->gw > 0 || B->seglen > 0 || E->gw > 0 || E->seglen > 0) return 1;
    if (E->out_mode != BF_OUT_STORE || E->colsum || E->bias || E->colscale || E->rowscale || E->gelu_out) return 1;
    if (E->aux_mode != BF_AUX_NONE && E->aux_mode != BF_AUX_ADD) return 1;
    if (!pair_shape_ok(M, N, K, A->ld, B->ld) || K < 2 * PK || E->ldc != N || (E->aux_mode == BF_AUX_ADD && E->ld_aux != N)) return 1;
    if (((uintptr_t)A->p | (uintptr_t)B->p | (uintptr_t)E->c | (uintptr_t)E->aux) & 15) return 1;
    PairArgs a;
    a.A = (const bf16*)A->p; a.lda = A->ld; a.B = (const bf16*)B->p; a.ldb = B->ld; a.N = N; a.nk = K / PK; a.nt = N / PN;
    a.x = nullptr; a.add = E->aux_mode == BF_AUX_ADD ? (const bf16*)E->aux : nullptr; a.out = (bf16*)E->c; a.ldx = N;
    a.mean = nullptr; a.rstd = nullptr; a.w = nullptr; a.ws = nullptr; a.fscale = rowfac; a.fdiv = rows_per_group > 0 ? rows_per_group : 1;
    a.out2 = rowfac ? (bf16*)out2 : nullptr; a.whole = 0;
    a.cz = nullptr; a.cdz = nullptr; a.cmean = a.crstd = a.cw = a.cg = nullptr; a.cgdiv = 1; a.cws = nullptr;
    a.bias = a.cs = a.ch = nullptr; memset(&a.n1, 0, sizeof(a.n1)); memset(&a.n2, 0, sizeof(a.n2));
    a.out_s = nullptr; a.fscale_s = nullptr; a.fdiv_s = 1;
    if (a.out2 && ((uintptr_t)out2 & 15)) return 1;
    BfProfScope prof(st, a.add ? "gemm_pair<add>" : "gemm_pair<plain>", 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K + (double)M * N * (a.add ? 2 : 1)));
    return launch_pair<1>(a, M, st);
}

// The forward twin of bf_gemm_inbwd_frames: out[M][N] = lin(A[M][K] @ Bt[K][N]) [+ add] on the frame-pair kernel (two 144-token frames x 128
// columns per workgroup), followed -- in the same launch, from the registers that hold whole frame columns -- by up to two InstanceNorms of the
// rows as stored: n1 (y = resid + g * IN(out): the MLP-branch norm behind fc2, layers/attention.py:312-317) and n2 (the NEXT stage's opening
// norm of the last tensor written: statistics, sc / sh and the normalised operand, layers/attention.py:77,208).  Statistics are summed in
// in_stats_kernel's order and the products in the streaming kernels': the results equal bf_gemm + bf_in_stats (+ apply) bit for bit.
// Returns 0 when done, 1 when the shape is not covered (bf16, S = 144, an even number of frames, N % 128 = K % 64 = 0, K >= 128), < 0 on error.
extern "C" int bf_gemm_fwd_frames(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* Bt, int64_t ldb, const float* bias,
                                  const float* colscale, const float* colshift, const float* fscale, int fdiv, const void* add, void* out, int S,
                                  const bf_frame_norm* n1, const bf_frame_norm* n2, bf_stream_t stream) {
    static const bool off = bf_knob("BF_FUSE_FWD_NORM", 1) == 0;
    if (off || dtype != BF_DTYPE_BF16 || S != FM || !pair_shape_ok(M, N, K, lda, ldb) || K < 2 * PK) return 1;
    BF_REQUIRE(A && Bt && out, "bf_gemm_fwd_frames: null pointer");
    BF_REQUIRE(!colscale == !colshift, "bf_gemm_fwd_frames: column scale and shift come together");
    if (((uintptr_t)A | (uintptr_t)Bt | (uintptr_t)add | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)colscale | (uintptr_t)colshift) & 15) return 1;
    PairArgs a;
    memset(&a, 0, sizeof(a));
    a.A = (const bf16*)A; a.lda = lda; a.B = (const bf16*)Bt; a.ldb = ldb; a.N = N; a.nk = K / PK; a.nt = N / PN;
    a.add = (const bf16*)add; a.out = (bf16*)out; a.ldx = N;
    a.fscale = fscale; a.fdiv = fdiv > 0 ? fdiv : 1; a.cgdiv = 1;
    a.bias = bias; a.cs = colscale; a.ch = colshift;
    auto take = [&](const bf_frame_norm* n, FwdNorm& o, bool first) -> int {
        if (!n) return 0;
        BF_REQUIRE(n->w && n->b && n->mean && n->rstd && n->sc && n->sh && n->out, "bf_gemm_fwd_frames: incomplete norm record");
        if (((uintptr_t)n->w | (uintptr_t)n->b | (uintptr_t)n->g | (uintptr_t)n->mean | (uintptr_t)n->rstd | (uintptr_t)n->sc | (uintptr_t)n->sh |
             (uintptr_t)n->resid | (uintptr_t)n->out) & 15) return 1;
        if (!first && (n->g || n->resid)) return 1;      // the chained norm is a plain one
        o.w = n->w; o.b = n->b; o.g = n->g; o.gdiv = n->gdiv > 0 ? n->gdiv : 1; o.mean = n->mean; o.rstd = n->rstd; o.sc = n->sc; o.sh = n->sh;
        o.resid = (const bf16*)n->resid; o.out = (bf16*)n->out;
        return 0;
    };
    if (int rc = take(n1, a.n1, true)) return rc;
    if (int rc = take(n2, a.n2, false)) return rc;
    BfProfScope prof((hipStream_t)stream, n1 ? (n2 ? "gemm_pair<fwd,norm,chain>" : "gemm_pair<fwd,norm>") : (n2 ? "gemm_pair<fwd,chain>" : "gemm_pair<fwd>"),
                     2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K + (double)M * N * (1 + (add ? 1 : 0) + (n1 ? (n1->resid ? 2 : 1) : 0) + (n2 ? 1 : 0))));
    return n1 ? launch_pair<4>(a, M, (hipStream_t)stream) : launch_pair<3>(a, M, (hipStream_t)stream);
}
