// Whole-frame forward projections for INFERENCE on gfx950 (eval forward of the trunk: scripts/inference.py:239-252 drives
// FiLMConditionedAViT.forward one clip at a time; layers/attention.py:77-78,120-121,208-210,298-299,316, linear_layers.py:18-25).
//
// At batch 1 a clip is 16 frames x 144 tokens x 384 channels: every kernel of the training-shaped path (statistics pass, streaming GEMM
// with a 96 KB resident weight block, 128 x 128 tile GEMM on 54 workgroups) is a 10-25 us launch that moves a few MB -- the step is the
// sum of ~160 such launches.  Here ONE workgroup owns all 144 token rows of a frame for a block of output columns, so that
//  * the InstanceNorm in FRONT of a projection (norm1 before QKV, norm2 before the temporal out-projection) runs inside the projection's
//    launch: the frame's 144 x 384 operand (110 KB) is resident in LDS -- the MI355X CU has 160 KB -- each wave owns one 64-channel block
//    of it, reduces the statistics in the same order as in_stats_kernel (norm.hip; the results are bit-identical) and normalises in place;
//  * the InstanceNorm BEHIND fc2 (out = x1 + gamma_mlp * IN(z), layers/attention.py:316-322) runs in fc2's epilogue: the tile holds
//    whole-frame columns of z;
//  * operands arrive by LDS-DMA (global_load_lds_dwordx4, gemm_common.h) in [rows][64] blocks with the usual chunk swizzle, waits are
//    counted (vmcnt) and per 64-deep K-block; weight blocks that do not fit beside the frame go through a ring of slots.
// 6 waves = 3 (rows, 48 each) x 2 (columns): per wave 3 x NTC MFMA 16x16x32 tiles.  The K order and the operand -> lane mapping are
// those of every other GEMM here, so accumulators match the tile / streaming kernels bit for bit.
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

namespace {
using namespace bfgemm;

constexpr int FS = 144;                 // token rows of a frame (12 x 12 patches)
constexpr int ABLK = FS * 64;           // elements of one [144][64] operand block (18 KB)
constexpr int NW = 6, NTHR = 64 * NW;

struct FrameArgs {
    const bf16* A; long lda; const bf16* W; long ldw; bf16* out; long ldo;
    int N, KB;                                            // output columns; 64-deep K blocks
    const float* nw; const float* nb;                     // InstanceNorm in front (resident form)
    const float* bias; const float* cs; const float* ch;  // v = acc + bias; v = v * cs + ch
    const bf16* resid; long ldr;                          // v += resid
    int gelu;                                             // v = gelu(v)
    const float* en_w; const float* en_b; const float* en_g;     // InstanceNorm behind: out = resid + en_g * IN(bf16(acc + bias))
    const float* xw; const float* xb; bf16* xn; long ldx;        // second output: the NEXT layer's InstanceNorm of `out` (its norm1), xn = IN(out) * xw + xb
};

__device__ __forceinline__ void wait_vm_n(int n) {      // n is wave-uniform; a smaller count than asked for is always safe (in-order retirement)
    if (n >= 16) { wait_vm<16>(); return; }
    switch (n) {
        case 0: wait_vm<0>(); break;   case 1: wait_vm<1>(); break;   case 2: wait_vm<2>(); break;   case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break;   case 5: wait_vm<5>(); break;   case 6: wait_vm<6>(); break;   case 7: wait_vm<7>(); break;
        case 8: wait_vm<8>(); break;   case 9: wait_vm<9>(); break;   case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
        case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; default: wait_vm<15>(); break;
    }
}
// 8-byte load the compiler does not count (it would drain the DMA queue at the first use): completion by the caller's wait
__device__ __forceinline__ void gload8(uint2& dst, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }

// v + (v of lane ^ o), o = 8 / 16 / 32: VALU only (norm.hip's reduction steps)
__device__ __forceinline__ float lane_xor_add(float v, int o) {
    if (o == 8) return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned u = __float_as_uint(v);
    const u2 r = o == 16 ? __builtin_amdgcn_permlane16_swap(u, u, false, false) : __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rg8_sum(float v) { return lane_xor_add(lane_xor_add(lane_xor_add(v, 8), 16), 32); }

// one DMA piece: rows 8p .. 8p+7 of a [rows][64] block; lane -> row 8p + (lane >> 3), LDS chunk (lane & 7) = global chunk
// (lane & 7) ^ ((row >> 1) & 7) of that row (lds_off<bf16, false, 64>)
__device__ __forceinline__ void dma_piece(const bf16* g0, long ld, unsigned lds_blk, int p, int lane) {
    const int r = 8 * p + (lane >> 3);
    glds16(g0 + (long)r * ld + 8 * ((lane & 7) ^ ((r >> 1) & 7)), lds_blk + (unsigned)p * 1024u);
}

// One frame's column statistics in in_stats_kernel's summation order (norm.hip: 32 row groups of rows rg + 32 q, eight row groups per
// wave reduced by lane ^ 8, 16, 32, the four waves added in order).  A lane holds row group residue `lane >> 3` of 8 channels:
// x[w4][qq][j] = row 8 w4 + (lane >> 3) + 32 qq (w4 < 4; qq < 5 where that row is < 144), channel j.  Returns mean and 1/sqrt(var + eps).
template <int NJ>
__device__ __forceinline__ void frame_stats(const float (&x)[4][5][NJ], float (&mu)[NJ], float (&rs)[NJ]) {
    float tot[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) tot[j] = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float p = 0.f;
#pragma unroll
            for (int qq = 0; qq < 5; ++qq)
                if (w4 < 2 || qq < 4) p += x[w4][qq][j];
            tot[j] += rg8_sum(p);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { mu[j] = tot[j] / (float)FS; tot[j] = 0.f; }
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float p = 0.f;
#pragma unroll
            for (int qq = 0; qq < 5; ++qq)
                if (w4 < 2 || qq < 4) { const float d = x[w4][qq][j] - mu[j]; p = fmaf(d, d, p); }
            tot[j] += rg8_sum(p);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) rs[j] = rsqrtf(tot[j] / (float)FS + BF_IN_EPS);
}

// Column statistics of a staged [144][LDZ] fp32 tile (values already rounded to the storage type), 8 columns per wave-task in
// frame_stats' order, -> tab[c] = scale, tab[BN + c] = shift of the InstanceNorm affine (w, b, optional layer scale g).  Whole workgroup;
// the caller brackets it with barriers.
template <int BN, int LDZ>
__device__ __forceinline__ void tile_col_stats(const float* zs, float* tab, const float* __restrict__ w, const float* __restrict__ b,
                                               const float* __restrict__ g, int n0, int wave, int lane) {
    const int rgp = lane >> 3, c8 = lane & 7;
    for (int task = wave; task < BN / 8; task += NW) {
        const int c = 8 * task + c8;
        float x[4][5][1];
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4)
#pragma unroll
            for (int qq = 0; qq < 5; ++qq) x[w4][qq][0] = (w4 < 2 || qq < 4) ? zs[(8 * w4 + rgp + 32 * qq) * LDZ + c] : 0.f;
        float mu[1], rs[1];
        frame_stats<1>(x, mu, rs);
        if (rgp == 0) {
            float aa = rs[0] * w[n0 + c];
            float s0 = fmaf(-mu[0], aa, b[n0 + c]);
            if (g) { const float gg = g[n0 + c]; aa *= gg; s0 = fmaf(s0, gg, 0.f); }
            tab[c] = aa; tab[BN + c] = s0;
        }
    }
}

// Epilogue of both kernel forms, from registers: acc[i][j][r] = row wm*48 + 16 i + li, column wn*16*NTC + 16 j + 4 lg + r of the tile.
//   en_w:  z = bf16(acc + bias); v = IN(z) * en_w * en_g + ... + resid     (statistics over the tile's whole-frame columns)
//   else:  v = acc + bias; v = v * cs + ch; v += resid; v = gelu(v)        (each step if given)
//   xn:    second output bf16(IN(bf16(v)) * xw + xb): the next layer's norm1 of this tensor, again whole-frame columns
// The operand images are dead when this runs; the staging tile and the 2 x BN table overlay them (the callers' LDS is large enough).
template <int NTC>
__device__ __forceinline__ void frame_epilogue(const FrameArgs& a, f32x4 (&acc)[3][NTC], const bf16x4 (&rx)[3][NTC], char* smem, int f, int n0,
                                               int wave, int lane) {
    constexpr int BN = 32 * NTC, LDZ = BN + 1;
    const int wm = wave % 3, wn = wave / 3, li = lane & 15, lg = lane >> 4;
    float* zs = reinterpret_cast<float*>(smem);          // [144][LDZ]
    float* tab = zs + FS * LDZ;                          // [2][BN]
    auto stage = [&]() {                                 // acc -> zs (values are storage-rounded already)
#pragma unroll
        for (int j = 0; j < NTC; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) zs[(wm * 48 + i * 16 + li) * LDZ + wn * (16 * NTC) + j * 16 + 4 * lg + r] = acc[i][j][r];
    };
    if (a.en_w) {
#pragma unroll
        for (int j = 0; j < NTC; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)       // the tensor in_stats_kernel reads in the separate-launch path is bf16
                    acc[i][j][r] = (float)(bf16)(acc[i][j][r] + (a.bias ? a.bias[n0 + wn * (16 * NTC) + j * 16 + 4 * lg + r] : 0.f));
        lds_barrier();                                   // every wave is done with the operand images
        stage();
        lds_barrier();
        tile_col_stats<BN, LDZ>(zs, tab, a.en_w, a.en_b, a.en_g, n0, wave, lane);
        lds_barrier();
    }
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
        const int c = wn * (16 * NTC) + j * 16 + 4 * lg, n = n0 + c;
        float cb[4], cs[4], ch[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { cb[r] = a.bias ? a.bias[n + r] : 0.f; cs[r] = a.cs ? a.cs[n + r] : 1.f; ch[r] = a.cs ? a.ch[n + r] : 0.f; }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const long row = (long)f * FS + wm * 48 + i * 16 + li;
            float v[4];
            if (a.en_w) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[i][j][r], tab[c + r], tab[BN + c + r]) + (float)rx[i][j][r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][j][r] + cb[r];
                    if (a.cs) v[r] = fmaf(v[r], cs[r], ch[r]);
                    if (a.resid) v[r] = v[r] + (float)rx[i][j][r];
                    if (a.gelu) v[r] = gelu_fast(v[r]);
                }
            }
            const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *reinterpret_cast<bf16x4*>(a.out + row * a.ldo + n) = o;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (float)o[r];
        }
    }
    if (a.xn) {
        lds_barrier();                                   // operand images (or the first statistics round's reads) are done
        stage();
        lds_barrier();
        tile_col_stats<BN, LDZ>(zs, tab, a.xw, a.xb, nullptr, n0, wave, lane);
        lds_barrier();
#pragma unroll
        for (int j = 0; j < NTC; ++j) {
            const int c = wn * (16 * NTC) + j * 16 + 4 * lg;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const long row = (long)f * FS + wm * 48 + i * 16 + li;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16)(fmaf(acc[i][j][r], tab[c + r], tab[BN + c + r]) + 0.f);
                *reinterpret_cast<bf16x4*>(a.xn + row * a.ldx + n0 + c) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Resident form, K = 384: the frame's whole operand (six [144][64] blocks) sits in LDS; weight blocks [BN][64] in NSB slots.
template <int NTC, int NSB, bool NORM>
__global__ void __launch_bounds__(NTHR) frame_res_kernel(FrameArgs a) {
    constexpr int KB = 6, BN = 32 * NTC, PB = BN / 8, BBLK = BN * 64;
    static_assert(NSB >= 2 && NSB <= KB, "weight ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Aimg = reinterpret_cast<bf16*>(smem);
    bf16* Bimg = Aimg + KB * ABLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % 3, wn = wave / 3;
    const int li = lane & 15, lg = lane >> 4;
    const int ncb = a.N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int f = bid / ncb, n0 = (bid % ncb) * BN;
    const bf16* Af = a.A + (long)f * FS * a.lda;
    const bf16* Wb = a.W + (long)n0 * a.ldw;
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds_addr(Aimg)), ldsB = __builtin_amdgcn_readfirstlane(lds_addr(Bimg));

    int issued = 0;
    auto issueA = [&](int kb) {
#pragma unroll
        for (int t = 0; t < 3; ++t) dma_piece(Af + kb * 64, a.lda, ldsA + (unsigned)kb * (unsigned)(ABLK * 2), wave + NW * t, lane);
        issued += 3;
    };
    auto issueB = [&](int kb) {
        for (int p = wave; p < PB; p += NW) { dma_piece(Wb + kb * 64, a.ldw, ldsB + (unsigned)(kb % NSB) * (unsigned)(BBLK * 2), p, lane); ++issued; }
    };
    int need[KB + NSB];
    if constexpr (NORM) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) issueA(kb);
        const int markA = issued;
#pragma unroll
        for (int kb = 0; kb < NSB; ++kb) { issueB(kb); need[kb] = issued; }
#pragma unroll
        for (int kb = NSB; kb < KB; ++kb) need[kb] = issued;
        // ---- InstanceNorm of the frame, in place: wave w owns K-block w (channels 64 w .. 64 w + 63)
        wait_vm_n(issued - markA);
        lds_barrier();
        bf16* blk = Aimg + wave * ABLK;
        const int rgp = lane >> 3, c8 = lane & 7;
        float x[4][5][8];
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4)
#pragma unroll
            for (int qq = 0; qq < 5; ++qq)
                if (w4 < 2 || qq < 4) {
                    const int r = 8 * w4 + rgp + 32 * qq;
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(blk + lds_off<bf16, false, 64>(r, 8 * c8));
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[w4][qq][j] = (float)v[j];
                }
        float mu[8], rs[8], aa[8], ss[8];
        frame_stats<8>(x, mu, rs);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 64 * wave + 8 * c8 + j;
            aa[j] = rs[j] * a.nw[c];
            ss[j] = fmaf(-mu[j], aa[j], a.nb[c]);
        }
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4)
#pragma unroll
            for (int qq = 0; qq < 5; ++qq)
                if (w4 < 2 || qq < 4) {
                    const int r = 8 * w4 + rgp + 32 * qq;
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)(fmaf(x[w4][qq][j], aa[j], ss[j]) + 0.f);
                    *reinterpret_cast<bf16x8*>(blk + lds_off<bf16, false, 64>(r, 8 * c8)) = o;
                }
    } else {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) { issueA(kb); if (kb < NSB) issueB(kb); need[kb] = issued; }
    }
    // residual rows of this lane's accumulator tiles: in flight under the whole product
    uint2 rr[3][NTC];
    if (a.resid) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < NTC; ++j)
                gload8(rr[i][j], a.resid + ((long)f * FS + wm * 48 + i * 16 + li) * a.ldr + n0 + wn * (16 * NTC) + j * 16 + 4 * lg);
        issued += 3 * NTC;
    }

    f32x4 acc[3][NTC];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        wait_vm_n(issued - need[kb]);
        lds_barrier();                                   // block kb has landed for every wave (and, NORM, is normalised); slot (kb - 1) % NSB is free
        if (kb >= 1 && kb + NSB - 1 < KB) { issueB(kb + NSB - 1); need[kb + NSB - 1] = issued; }
        const bf16* cA = Aimg + kb * ABLK;
        const bf16* cB = Bimg + (kb % NSB) * BBLK;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 32) {
            bf16x8 fa[3], fb[NTC];
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] = frag_bf16<false, 64>(cA, wm * 48 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < NTC; ++j) fb[j] = frag_bf16<false, 64>(cB, wn * (16 * NTC) + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < NTC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }
    // ---- epilogue
    wait_vm<0>();
    bf16x4 rx[3][NTC];
    if (a.resid) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < NTC; ++j) { asm volatile("" : "+v"(rr[i][j].x), "+v"(rr[i][j].y)); rx[i][j] = __builtin_bit_cast(bf16x4, rr[i][j]); }
    }
    __builtin_amdgcn_sched_barrier(0);
    frame_epilogue<NTC>(a, acc, rx, smem, f, n0, wave, lane);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Streamed product of one frame tile: acc += A[144 rows][64 KB] @ W[BN rows][64 KB]^T, both operands through a ring of NS slots
// {[144][64] | [BN][64]} filled by LDS-DMA, NS - 1 blocks in flight, counted vmcnt + one raw barrier per 64-deep K-block.
template <int NTC, int NS>
__device__ __forceinline__ void ring_product(const bf16* Af, long lda, const bf16* Wb, long ldw, int KB, char* smem, f32x4 (&acc)[3][NTC], int wave, int lane) {
    constexpr int BN = 32 * NTC, PB = BN / 8, BBLK = BN * 64, SLOT = ABLK + BBLK;
    bf16* ring = reinterpret_cast<bf16*>(smem);
    const int wm = wave % 3, wn = wave / 3;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const int cntB = (PB - wave + NW - 1) / NW;          // weight pieces this wave issues per block (wave-uniform)
    const int grp = 3 + (cntB > 0 ? cntB : 0);
    auto issue = [&](int kb) {
        const unsigned s = lds0 + (unsigned)(kb % NS) * (unsigned)(SLOT * 2);
#pragma unroll
        for (int t = 0; t < 3; ++t) dma_piece(Af + (long)kb * 64, lda, s, wave + NW * t, lane);
        for (int p = wave; p < PB; p += NW) dma_piece(Wb + (long)kb * 64, ldw, s + (unsigned)(ABLK * 2), p, lane);
    };
    int last = -1;                                       // newest block issued
    for (int kb = 0; kb < NS && kb < KB; ++kb) { issue(kb); last = kb; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < KB; ++kb) {
        wait_vm_n((last - kb) * grp);
        lds_barrier();                                   // block kb has landed for every wave; the slot block kb - 1 used is free
        if (kb >= 1 && last + 1 < KB) { ++last; issue(last); }
        const bf16* cA = ring + (size_t)(kb % NS) * SLOT;
        const bf16* cB = cA + ABLK;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 32) {
            bf16x8 fa[3], fb[NTC];
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] = frag_bf16<false, 64>(cA, wm * 48 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < NTC; ++j) fb[j] = frag_bf16<false, 64>(cB, wn * (16 * NTC) + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < NTC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }
}

// Streamed form (fc2: K = 4E); optional InstanceNorm behind.
template <int NTC, int NS>
__global__ void __launch_bounds__(NTHR) frame_ring_kernel(FrameArgs a) {
    constexpr int BN = 32 * NTC, SLOT = ABLK + BN * 64;
    static_assert(FS * (BN + 1) * 4 + 2 * BN * 4 <= NS * SLOT * 2, "epilogue staging must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % 3, wn = wave / 3;
    const int li = lane & 15, lg = lane >> 4;
    const int ncb = a.N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int f = bid / ncb, n0 = (bid % ncb) * BN;
    f32x4 acc[3][NTC];
    ring_product<NTC, NS>(a.A + (long)f * FS * a.lda, a.lda, a.W + (long)n0 * a.ldw, a.ldw, a.KB, smem, acc, wave, lane);
    // ---- epilogue
    bf16x4 rx[3][NTC];
    if (a.resid) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < NTC; ++j)
                rx[i][j] = *reinterpret_cast<const bf16x4*>(a.resid + ((long)f * FS + wm * 48 + i * 16 + li) * a.ldr + n0 + wn * (16 * NTC) + j * 16 + 4 * lg);
    }
    frame_epilogue<NTC>(a, acc, rx, smem, f, n0, wave, lane);
}

template <typename Kern>
int launch_frame(Kern k, bool& attr_done, size_t lds, unsigned grid, const FrameArgs& a, hipStream_t st) {
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bf_fail(e, __FILE__, __LINE__);
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR), lds, st, a);
    BF_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Returns 0 when done, 1 when the shape is not covered (the caller then runs the statistics + GEMM launches of the training path), < 0 on error.
extern "C" int bf_frame_linear(int dtype, int frames, int S, int K, int N, const void* A, int64_t lda, const void* W, int64_t ldw,
                               const float* norm_w, const float* norm_b, const float* bias, const float* colscale, const float* colshift,
                               const void* resid, int64_t ldr, int gelu, const float* en_w, const float* en_b, const float* en_g,
                               void* out, int64_t ldo, const float* next_w, const float* next_b, void* out_n, int64_t ldn, bf_stream_t stream) {
    if (dtype != BF_DTYPE_BF16 || S != FS || frames < 1 || N % 32 || K % 64 || K < 128) return 1;
    const bool res = K == 384 && !en_w;
    if (!res && (norm_w || K / 64 < 2)) return 1;                       // the frame's operand is resident only at K = 384
    BF_REQUIRE(A && W && out, "bf_frame_linear: null pointer");
    BF_REQUIRE((norm_w == nullptr) == (norm_b == nullptr) && (colscale == nullptr) == (colshift == nullptr), "bf_frame_linear: norm / column tables come in pairs");
    BF_REQUIRE(!en_w || (en_b && en_g && resid && !colscale && !gelu), "bf_frame_linear: the InstanceNorm behind needs en_b, en_g and the residual");
    BF_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldo % 4 == 0 && (!resid || ldr % 4 == 0), "bf_frame_linear: leading dimensions must keep 16-byte chunks whole");
    BF_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 7) == 0 && ((uintptr_t)resid & 7) == 0, "bf_frame_linear: operands must be 16-byte aligned");
    FrameArgs a;
    a.A = (const bf16*)A; a.lda = lda; a.W = (const bf16*)W; a.ldw = ldw; a.out = (bf16*)out; a.ldo = ldo; a.N = N; a.KB = K / 64;
    a.nw = norm_w; a.nb = norm_b; a.bias = bias; a.cs = colscale; a.ch = colshift; a.resid = (const bf16*)resid; a.ldr = ldr; a.gelu = gelu;
    a.en_w = en_w; a.en_b = en_b; a.en_g = en_g;
    BF_REQUIRE(!out_n || (next_w && next_b && ldn % 4 == 0 && ((uintptr_t)out_n & 7) == 0), "bf_frame_linear: the second output needs the next norm's weight and bias");
    a.xw = next_w; a.xb = next_b; a.xn = (bf16*)out_n; a.ldx = ldn;
    hipStream_t st = (hipStream_t)stream;
    // widest column block that still gives every CU a workgroup (a frame's operand is shared by the N / BN workgroups of the frame)
    static const int force = bf_knob("BF_FRAME_NTC", 0);
    int ntc = 1;
    for (int c : {3, 2}) if (N % (32 * c) == 0 && (long)frames * (N / (32 * c)) >= 180) { ntc = c; break; }
    if (force >= 1 && force <= 3 && N % (32 * force) == 0) ntc = force;
    const unsigned grid = (unsigned)(frames * (N / (32 * ntc)));
    static thread_local char pname[64];
    snprintf(pname, sizeof(pname), "frame_linear<%s,bn%d%s>", res ? "res" : "ring", 32 * ntc, norm_w ? ",norm" : en_w ? ",in" : "");
    BfProfScope prof(st, pname, 2.0 * frames * FS * (double)N * K, ((double)frames * FS * (K + N * (resid ? 2.0 : 1.0)) + (double)N * K) * 2.0);
#define FR_RES(NTC, NSB, NORM) do { static BfPerDeviceOnce once; return launch_frame(frame_res_kernel<NTC, NSB, NORM>, once.flag(), (size_t)(6 * ABLK + NSB * 32 * NTC * 64) * 2, grid, a, st); } while (0)
#define FR_RING(NTC, NS) do { static BfPerDeviceOnce once; return launch_frame(frame_ring_kernel<NTC, NS>, once.flag(), (size_t)NS * (ABLK + 32 * NTC * 64) * 2, grid, a, st); } while (0)
    if (res) {
        if (norm_w) { if (ntc == 3) FR_RES(3, 4, true); else if (ntc == 2) FR_RES(2, 6, true); else FR_RES(1, 6, true); }
        else { if (ntc == 3) FR_RES(3, 4, false); else if (ntc == 2) FR_RES(2, 6, false); else FR_RES(1, 6, false); }
    } else {
        if (ntc == 3) FR_RING(3, 5); else if (ntc == 2) FR_RING(2, 6); else FR_RING(1, 7);
    }
#undef FR_RES
#undef FR_RING
    return 0;
}

