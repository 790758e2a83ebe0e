// Token-reduction GEMM for gfx950: the weight gradient of every 1x1 conv / Linear of the trunk,
//
//     out[Nout][Kin] (+)= sum_tok dy[tok][Nout] * x[tok][Kin]           (autograd of layers/attention.py:78,121,210,299,
//                                                                         linear_layers.py:18-25; both operands token-major)
//
// The reduction runs over tens of thousands of tokens and the result is a few hundred KB, so the split over workgroups is a split
// of the TOKENS, and what that costs is the partial results: (#workgroups) x (tile bytes) of fp32 leave the chip whatever the tile
// shape.  So: FEW workgroups, each as efficient as a CU can be, beside the caller's data-gradient kernels (the launch deliberately
// does not fill the chip).  The partial tile goes to a slab with plain 16-byte stores and a second small kernel adds the slabs in
// slice order: bit-reproducible run to run, no float atomics.
//
// Main kernel (tokred_pp_kernel): one workgroup = one 384 x 192 output tile x one token slice -- 128 FLOP per staged byte, the ratio
// of a 256 x 256 tile, in a shape that divides every trunk width (E, 3E, 4E with E a multiple of 384) -- one workgroup per CU
// (144 KB of LDS), 8 waves as 4 x 2 with 96 x 96 outputs each (36 accumulator tiles of 16 x 16).  Schedule (cdna_hip_programming.md
// section 5, the 8-phase template's ping-pong, and MI355X_MICROARCH.md "Two waves per SIMD"):
//  * the unit of work is a HALF-step of 32 tokens.  Operand rows go global -> LDS by DMA (global_load_lds_dwordx4, no staging
//    registers) into a ring of four 36 KB half-buffers: while half H is read, halves H+1 .. H+3 are landing or in flight; each wave
//    waits with a counted s_waitcnt vmcnt for exactly its own pieces of half H+1 and never drains the queue inside the loop;
//  * per half a wave runs a LOAD segment (24 ds_read_b64_tr_b16 for 6 + 6 fragments, its 4-5 DMA pieces of half H+3, the counted wait)
//    and a COMPUTE segment (36 MFMA 16x16x32 = 576 matrix-pipe cycles), each closed by a raw s_barrier.  Waves 4-7 run one barrier
//    behind waves 0-3, so on every SIMD one wave computes while its partner loads: the matrix pipe alternates between the two and
//    never waits for LDS reads or DMA issue (with all eight waves in the same phase, the earlier kernel's step was the SUM of its parts);
//  * LDS image: sub-chunks [32 tokens][64 channels] with 128-byte rows whose 32-byte segments are XORed with a row key, so that the
//    eight rows a half-wave of the transposing read touches fall on eight different bank groups (no conflicts); the DMA writes LDS
//    lane-linearly, so the same permutation is applied to each lane's SOURCE address.  A wave's fragments all sit at one column
//    position of their sub-chunks (output tiles are dealt to the waves interleaved), i.e. one address register per operand side and
//    immediates for the rest; its DMA pieces are the same rows of consecutive sub-chunks: one source pointer, immediates again.
//  * the bias gradient colsum(dy) comes out of the same pass: three extra MFMAs per half against an all-ones operand (column block 0 only).
// Shapes outside that tiling (tiny test models) take the 128 x 128 kernel below (3-slot ring, one barrier per 64-token step).
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

namespace {
using namespace bfgemm;

// ================================================================================================ ping-pong kernel: 384 x 192 or 192 x 192 tiles
constexpr int PTN = 192;                      // output tile, Kin side (3 sub-chunks of 64 channels)
constexpr int HR = 32;                        // tokens per half-step
constexpr int NSB = PTN / 64;
constexpr int SUBB = HR * 64 * 2;             // bytes of one sub-chunk [32][64] (4 KB = 4 DMA pieces of 8 rows)
constexpr int NBUF = 4;                       // ring of half-buffers
// NI = 16-row output tiles per wave on the Nout side = 64-channel sub-chunks of dy per half: 6 -> 384 x 192 tiles (128 FLOP per staged
// byte, 144 KB of LDS), 3 -> 192 x 192 (96 FLOP per byte, 96 KB): the same slab traffic (that is set by the slice count alone), twice
// as many, half as long workgroups.
template <int NI> struct PPGeom {
    static constexpr int TM = 64 * NI, HALFB = (NI + NSB) * SUBB, TILES = NI * 6, TILE_FLOATS = TM * PTN;
};

// LDS-DMA with an immediate byte offset (one source pointer serves the same rows of several sub-chunks).  The instruction adds its
// immediate to BOTH addresses -- the global source and the LDS destination (M0 + offset + 16 * lane) -- so M0 is given the
// destination minus the offset.
template <int OFF>
__device__ __forceinline__ void glds16_off(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst - (unsigned)OFF), "n"(OFF) : "memory");
}

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__device__ __forceinline__ bf16x8 tr_frag(unsigned addr) {       // addr: LDS byte address of the lane's "lo" block row; "hi" = 4 rows on
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uintptr_t)addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uintptr_t)(addr + 512u));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// GRP 0: waves 0-3 (start in the load segment), GRP 1: waves 4-7 (one barrier behind).  Same program, different DMA pieces.
template <int NI, int GRP>
__device__ __forceinline__ void tokred_pp_body(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B, long ldb,
                                               float* __restrict__ slab_tile, float* __restrict__ cslab_row, int nh, bool do_cs,
                                               unsigned ring_lds, int lane, int w4) {
    using Gm = PPGeom<NI>;
    constexpr int HALFB = Gm::HALFB;
    // DMA pieces per wave per half: the NI + 3 sub-chunks are dealt to the two wave groups (same 8 rows of each): 5 + 4 or 3 + 3
    constexpr int PG = NI == 6 ? (GRP == 0 ? 5 : 4) : 3;
    constexpr int NCS = 3;                    // column-sum accumulators per wave (NI = 6: three of its six Nout tiles each; NI = 3: waves 0-3 take all three)
    const int wm = w4, wn = GRP;              // 4 x 2 waves; output tiles dealt interleaved: Nout tiles wm + 4i, Kin tiles wn + 2j
    // ---- DMA geometry: piece = 8 token rows x 128 bytes of one sub-chunk; lane -> row 8 w4 + (lane >> 3), LDS 16-byte chunk (lane & 7),
    // which holds global chunk (lane & 7) ^ (key(row) << 1), key(r) = ((r >> 1) & 1) | (((r >> 3) & 1) << 1)
    const int dr = 8 * w4 + (lane >> 3);
    const int dkey = ((dr >> 1) & 1) | (((dr >> 3) & 1) << 1);
    const int dch = (lane & 7) ^ (dkey << 1);
    const bf16* pA = A + (long)dr * lda + 8 * dch;
    const bf16* pB = B + (long)dr * ldb + 8 * dch;
    const long stepA = (long)HR * lda, stepB = (long)HR * ldb;
    auto issue = [&](int buf) {
        const unsigned base = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)buf * (unsigned)HALFB + (unsigned)w4 * 1024u);
        if constexpr (NI == 6 && GRP == 0) {
            glds16_off<0>(pA, base + 0 * SUBB); glds16_off<128>(pA, base + 1 * SUBB); glds16_off<256>(pA, base + 2 * SUBB);
            glds16_off<384>(pA, base + 3 * SUBB); glds16_off<512>(pA, base + 4 * SUBB);
            pA += stepA;
        } else if constexpr (NI == 6) {
            glds16_off<640>(pA, base + 5 * SUBB);
            glds16_off<0>(pB, base + 6 * SUBB); glds16_off<128>(pB, base + 7 * SUBB); glds16_off<256>(pB, base + 8 * SUBB);
            pA += stepA; pB += stepB;
        } else if constexpr (GRP == 0) {
            glds16_off<0>(pA, base + 0 * SUBB); glds16_off<128>(pA, base + 1 * SUBB); glds16_off<256>(pA, base + 2 * SUBB);
            pA += stepA;
        } else {
            glds16_off<0>(pB, base + 3 * SUBB); glds16_off<128>(pB, base + 4 * SUBB); glds16_off<256>(pB, base + 5 * SUBB);
            pB += stepB;
        }
    };
    // ---- fragment addresses (byte offsets inside a half-buffer): lane 16 g + 4 q + p reads block row 8 g + q (lo) / + 4 (hi),
    // columns pos + 4 p .. + 3 of its tile; key(row) = ((q >> 1) & 1) | ((g & 1) << 1), the same for lo and hi
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int fkey = ((q >> 1) & 1) | ((g & 1) << 1);
    const unsigned rowb = (unsigned)(8 * g + q) * 128u + (unsigned)(p & 1) * 8u;
    const unsigned offA = rowb + (unsigned)(((2 * wm + (p >> 1)) ^ (fkey << 1)) * 16);
    const unsigned offB0 = rowb + (unsigned)(((2 * wn + (p >> 1)) ^ (fkey << 1)) * 16) + (unsigned)(NI * SUBB);
    const unsigned offB1 = rowb + (unsigned)(((2 * wn + 4 + (p >> 1)) ^ (fkey << 1)) * 16) + (unsigned)(NI * SUBB);

    f32x4 acc[NI][6];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 cs[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) cs[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int c = 0; c < 8; ++c) ones[c] = (bf16)1.0f;

    // ---- prologue: three halves in flight, the first one landed for every wave
    issue(0);
    if (nh > 1) issue(1);
    if (nh > 2) issue(2);
    if (nh > 2) wait_vm<2 * PG>(); else if (nh > 1) wait_vm<PG>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if constexpr (GRP == 1) __builtin_amdgcn_s_barrier();          // one barrier behind waves 0-3 from here on

    int buf = 0;
    for (int h = 0; h < nh; ++h) {
        // ======== load segment: fragments of half h, DMA of half h + 3 (its buffer held half h - 1: every wave finished reading it
        // before the barrier that ended ITS load segment of h - 1), then the counted wait for this wave's pieces of half h + 1
        const unsigned hb = ring_lds + (unsigned)buf * (unsigned)HALFB;
        bf16x8 fa[NI], fb[6];
#pragma unroll
        for (int i = 0; i < NI; ++i) fa[i] = tr_frag(hb + offA + (unsigned)(i * SUBB));
#pragma unroll
        for (int j = 0; j < 6; ++j) fb[j] = tr_frag(hb + ((j & 1) ? offB1 : offB0) + (unsigned)((j >> 1) * SUBB));
        if (h + 3 < nh) issue((buf + 3) & 3);
        const int younger = min(h + 3, nh - 1) - (h + 1);           // halves issued after h + 1 (0 .. 2; negative at the very end)
        if (younger >= 2) wait_vm<2 * PG>(); else if (younger == 1) wait_vm<PG>(); else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the reads are done before the barrier: the buffer may be refilled after it
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ======== compute segment
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        if (do_cs) {                   // column sums of dy against an all-ones operand
#pragma unroll
            for (int c = 0; c < NCS; ++c) {
                if constexpr (NI == 6) cs[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[NCS * GRP + c], cs[c], 0, 0, 0);
                else if constexpr (GRP == 0) cs[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[c], cs[c], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        buf = (buf + 1) & 3;
    }
    if constexpr (GRP == 0) __builtin_amdgcn_s_barrier();          // pairs with the extra barrier of waves 4-7

    // ---- partial tile -> slab in FRAGMENT order ([wave][i][j][lane] float4: 1 KB contiguous per wave-instruction); the reduce
    // kernel undoes the order when it writes the result
    const int wave = GRP * 4 + w4;
    float* so = slab_tile + ((size_t)wave * Gm::TILES * 64 + lane) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
            *reinterpret_cast<float4*>(so + (size_t)(i * 6 + j) * 256) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    if (do_cs && lane < 16) {
        if constexpr (NI == 6) {
#pragma unroll
            for (int c = 0; c < NCS; ++c) cslab_row[16 * (wm + 4 * (NCS * GRP + c)) + lane] = cs[c][0];
        } else if constexpr (GRP == 0) {
#pragma unroll
            for (int c = 0; c < NCS; ++c) cslab_row[16 * (wm + 4 * c) + lane] = cs[c][0];
        }
    }
}

// The slab sum of the PREVIOUS token-reduction GEMM, carried by extra workgroups of this launch (bf_gemm_tokred_deferred): the slabs were
// complete when this launch started (stream order), so no hand-off protocol is needed; the sum runs in slice order exactly as
// tokred_pp_reduce_kernel's (same bits) while the other workgroups multiply, instead of in a launch of its own between two GEMMs.
struct FoldRed { const float* slab; const float* cslab; int nslice, ntiles, tiles_k, Nout, Kin; float* out; float* colsum; int accumulate; int nwg; };
template <int NI>
__device__ __forceinline__ void fold_reduce(const FoldRed& r, int b) {
    using Gm = PPGeom<NI>;
    constexpr int T4 = Gm::TILE_FLOATS / 4;
    const long total = (long)r.ntiles * T4;
    const size_t sstride = (size_t)r.ntiles * Gm::TILE_FLOATS;
    for (long e = (long)b * 512 + threadIdx.x; e < total; e += (long)r.nwg * 512) {
        const int tile = (int)(e / T4), q = (int)(e - (long)tile * T4);
        const int lane = q & 63, t = q >> 6, wave = t / Gm::TILES, ij = t - wave * Gm::TILES, i = ij / 6, j = ij - i * 6;
        const int wm = wave & 3, wn = wave >> 2;
        const int nout = (tile / r.tiles_k) * Gm::TM + 16 * (wm + 4 * i) + (lane & 15);
        const int kin = (tile % r.tiles_k) * PTN + 16 * (wn + 2 * j) + 4 * (lane >> 4);
        const float* src = r.slab + (size_t)tile * Gm::TILE_FLOATS + (size_t)q * 4;
        float4* dst = reinterpret_cast<float4*>(r.out + (size_t)nout * r.Kin + kin);
        float4 a = r.accumulate ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < r.nslice; s0 += 8) {          // eight slices in flight (a load-then-add loop pays a memory round trip per slice)
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s0 + u < r.nslice ? *reinterpret_cast<const float4*>(src + (size_t)(s0 + u) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < r.nslice) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        *dst = a;
    }
    if (r.colsum)
        for (long m = (long)b * 512 + threadIdx.x; m < r.Nout; m += (long)r.nwg * 512) {
            float a = r.accumulate ? r.colsum[m] : 0.f;
            for (int s2 = 0; s2 < r.nslice; ++s2) a += r.cslab[(size_t)s2 * r.Nout + m];
            r.colsum[m] = a;
        }
}

template <int NI>
__global__ void __launch_bounds__(512) tokred_pp_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B, long ldb,
                                                      float* __restrict__ slab, float* __restrict__ cslab, int Nout, int halves_total,
                                                      int halves_per, int tiles_k, int ntiles, int main_wgs, FoldRed red) {
    using Gm = PPGeom<NI>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= main_wgs) { fold_reduce<NI>(red, (int)blockIdx.x - main_wgs); return; }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // contiguous runs of the (slice, tile) sequence per XCD: the tiles of a slice read the same token rows through one L2
    const int seq = xcd_remap(blockIdx.x, main_wgs);
    const int slice = seq / ntiles, tile = seq - slice * ntiles;
    const int n0 = (tile / tiles_k) * Gm::TM, c0 = (tile % tiles_k) * PTN;
    const int h_beg = slice * halves_per;
    const int nh = min(halves_per, halves_total - h_beg);
    const bf16* a0 = A + (long)h_beg * HR * lda + n0;
    const bf16* b0 = B + (long)h_beg * HR * ldb + c0;
    float* slab_tile = slab + ((size_t)slice * ntiles + tile) * Gm::TILE_FLOATS;
    const bool do_cs = cslab != nullptr && c0 == 0;
    float* cslab_row = do_cs ? cslab + (size_t)slice * Nout + n0 : nullptr;
    const unsigned ring_lds = lds_addr(smem);
    if (wave < 4) tokred_pp_body<NI, 0>(a0, lda, b0, ldb, slab_tile, cslab_row, nh, do_cs, ring_lds, lane, wave);
    else tokred_pp_body<NI, 1>(a0, lda, b0, ldb, slab_tile, cslab_row, nh, do_cs, ring_lds, lane, wave - 4);
}

// out (+)= sum over slices of the fragment-ordered slabs, in slice order; one float4 of one tile per thread, the slices' loads issued
// together (a load-then-add loop paid a memory round trip per slice).  Column sums likewise from cslab.
template <int NI, int MAXS>
__global__ void __launch_bounds__(256) tokred_pp_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ cslab, int nslice,
                                                               int ntiles, int tiles_k, int Nout, int Kin, float* __restrict__ out,
                                                               float* __restrict__ colsum, int accumulate) {
    using Gm = PPGeom<NI>;
    constexpr int T4 = Gm::TILE_FLOATS / 4;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < (long)ntiles * T4) {
        const int tile = (int)(e / T4), r = (int)(e - (long)tile * T4);
        const int lane = r & 63, t = r >> 6, wave = t / Gm::TILES, ij = t - wave * Gm::TILES, i = ij / 6, j = ij - i * 6;
        const int wm = wave & 3, wn = wave >> 2;
        const int nout = (tile / tiles_k) * Gm::TM + 16 * (wm + 4 * i) + (lane & 15);
        const int kin = (tile % tiles_k) * PTN + 16 * (wn + 2 * j) + 4 * (lane >> 4);
        const size_t sstride = (size_t)ntiles * Gm::TILE_FLOATS;
        const float* src = slab + (size_t)tile * Gm::TILE_FLOATS + (size_t)r * 4;
        float4 v[MAXS];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) v[s] = s < nslice ? *reinterpret_cast<const float4*>(src + s * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4* dst = reinterpret_cast<float4*>(out + (size_t)nout * Kin + kin);
        float4 a = accumulate ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < nslice) { a.x += v[s].x; a.y += v[s].y; a.z += v[s].z; a.w += v[s].w; }
        *dst = a;
    }
    if (colsum && e < Nout) {
        float v[MAXS];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) v[s] = s < nslice ? cslab[(size_t)s * Nout + e] : 0.f;
        float a = accumulate ? colsum[e] : 0.f;
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < nslice) a += v[s];
        colsum[e] = a;
    }
}

// ================================================================================================ narrow form: Kin = 16 (the first embed stage's weight gradient)
// dW[C][16] = sum_p dy[p][C] * x[p][16] over ~1.2 M pixel rows (HMLPEmbed stage 0, layers/patching.py:36-44: Conv2d k2s2 on 4 fields = 16
// patch values; autograd of it).  Six MFMAs per 32 pixels against 7 KB of operands: a pure stream.  On the 128 x 128 split-K tile kernel
// 7/8 of the weight tile was padding and the launch took 287 us at the END of the step, when nothing else is left to overlap with.  Here
// every wave is its own pipeline: it pulls 32-pixel tiles (dy rows and patch rows are contiguous in memory: 6 + 1 DMA pieces, lane-linear
// LDS image) into a private two-tile ring, reads them back transposed (ds_read_b64_tr_b16: the reduction index of both operands is the
// row) and keeps dW[C][16] in C/16 accumulator tiles.  No barrier in the loop; partial results are summed per workgroup through LDS, written
// to a slab row per workgroup and added in workgroup order by a tiny second kernel: bit-reproducible, no float atomics.
constexpr int NW_TILE = 32;                         // pixels per tile
// PRO: the wide operand is an activation map that enters as gelu(y * sc[frame][c] + sh[frame][c]) (the InstanceNorm + GELU in front of
// the last debed stage, layers/patching.py:92-104: that stage's weight gradient, with the roles of the operands swapped) -- applied to the
// fragments in registers, 8 pixels of one channel per lane, the frame's scale / shift reloaded only when a wave's tiles cross a frame.
template <int CT, bool PRO>                         // CT = C / 16 column tiles of dy (6 for the 96-channel stages)
__global__ void __launch_bounds__(256) tokred_narrow_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, long tiles_total, float* __restrict__ slab,
                                                          const float* __restrict__ psc, const float* __restrict__ psh, int tiles_per_frame) {
    constexpr int C = 16 * CT, DYB = NW_TILE * C * 2, XB = NW_TILE * 16 * 2, TB_ = DYB + XB;      // bytes of one staged tile: dy rows | patch rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long nwaves = (long)gridDim.x * 4, w = (long)blockIdx.x * 4 + wave;
    const long t_beg = tiles_total * w / nwaves, t_end = tiles_total * (w + 1) / nwaves;
    char* mine = smem + (size_t)wave * 2 * TB_;
    const unsigned lds0 = lds_addr(mine);
    // DMA: a tile's dy rows are NW_TILE * C * 2 contiguous bytes (DYB / 1024 pieces), its patch rows 1 KB (one piece)
    auto issue = [&](long t, int slot) {
        const char* gd = reinterpret_cast<const char*>(dy) + t * DYB + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot * (unsigned)TB_);
#pragma unroll
        for (int pce = 0; pce < DYB / 1024; ++pce) glds16(gd + pce * 1024, dst + (unsigned)pce * 1024u);
        glds16(reinterpret_cast<const char*>(x) + t * XB + lane * 16, dst + (unsigned)DYB);
    };
    constexpr int PCS = DYB / 1024 + 1;
    f32x4 acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transposing reads: lane 16 g + 4 q + p supplies row 8 g + q (lo) / 8 g + 4 + q (hi), columns c0 + 4 p .. + 3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const unsigned offD = (unsigned)((8 * g + q) * (C * 2) + 8 * pp), offX = (unsigned)(DYB + (8 * g + q) * 32 + 8 * pp);
    auto trf = [&](unsigned addr, unsigned hi_off) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uintptr_t)addr);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uintptr_t)(addr + hi_off));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    };
    // The tile loop waits by COUNT for its own LDS-DMA pieces, so nothing else may sit in the vector-memory queue while it runs: the
    // frame's scale / shift (ordinary loads to registers) are fetched BETWEEN loops -- a wave's tile range is cut at frame boundaries, and
    // each segment starts with an empty queue (drain, load, drain) and restarts the two-slot pipeline.  (The earlier form, loads inside the
    // loop, produced a second channel block 1-2 % off and different from run to run: traced at the ISA level to the packed fp32 form hipcc
    // chose for that block's scale / shift -- v_pk_fma_f32 .. op_sel:[0,1,1], wrong with two workgroups on a CU; EXPERIMENTS.md round 4 --
    // which tools/register_audit.py now refuses, as it refuses compiler loads inside a counted loop.)
    float scv[CT], shv[CT];
    for (long s_beg = t_beg; s_beg < t_end;) {
        long s_end = t_end;
        if constexpr (PRO) {
            const long f = s_beg / tiles_per_frame;          // wave-uniform
            s_end = min(t_end, (f + 1) * (long)tiles_per_frame);
            wait_vm<0>();
#pragma unroll
            for (int i = 0; i < CT; ++i) { scv[i] = psc[f * C + 16 * i + (lane & 15)]; shv[i] = psh ? psh[f * C + 16 * i + (lane & 15)] : 0.f; }
#pragma unroll
            for (int i = 0; i < CT; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(scv[i]), "+v"(shv[i]) :: "memory");      // operands: the loads cannot sink below
        }
        issue(s_beg, 0);
        int slot = 0;
        for (long t = s_beg; t < s_end; ++t) {
            if (t + 1 < s_end) { issue(t + 1, slot ^ 1); wait_vm<PCS>(); } else wait_vm<0>();
            const unsigned tb = lds0 + (unsigned)slot * (unsigned)TB_;
            const bf16x8 fx = trf(tb + offX, 4 * 32);
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                bf16x8 fd = trf(tb + offD + (unsigned)(i * 32), 4 * C * 2);
                if constexpr (PRO) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) fd[j] = (bf16)gelu_fast(fmaf((float)fd[j], scv[i], shv[i]));
                }
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx, fd, acc[i], 0, 0, 0);      // acc[i][r]: dW[16 i + (lane & 15)][4 (lane >> 4) + r]
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads of this slot are done before the next iteration's DMA may refill it
            slot ^= 1;
        }
        s_beg = s_end;
    }
    // ---- the workgroup's four partial results -> one slab row [C][16], summed in wave order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                // [4 waves][C][16]
#pragma unroll
    for (int i = 0; i < CT; ++i)
        *reinterpret_cast<float4*>(red + ((size_t)wave * C + 16 * i + (lane & 15)) * 16 + 4 * (lane >> 4)) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    __syncthreads();
    for (int e = tid; e < C * 16; e += 256) {
        float v = red[e];
#pragma unroll
        for (int k = 1; k < 4; ++k) v += red[(size_t)k * C * 16 + e];
        slab[(size_t)blockIdx.x * C * 16 + e] = v;
    }
}
// out[C][ldo] (+)= sum over the workgroup rows of the slab, in workgroup order
__global__ void __launch_bounds__(256) tokred_narrow_reduce_kernel(const float* __restrict__ slab, int rows, int C, float* __restrict__ out, int ldo, int accumulate,
                                                                   int transposed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C * 16) return;
    float v = 0.f;
    int r = 0;
    for (; r + 16 <= rows; r += 16) {          // sixteen rows in flight (a load-then-add loop pays a memory round trip per row: 98 us for 256 rows)
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = slab[(size_t)(r + u) * C * 16 + e];
#pragma unroll
        for (int u = 0; u < 16; ++u) v += t[u];
    }
    for (; r < rows; ++r) v += slab[(size_t)r * C * 16 + e];
    float* o = transposed ? out + (size_t)(e & 15) * ldo + (e >> 4) : out + (size_t)(e >> 4) * ldo + (e & 15);      // transposed: out[16][C]
    *o = accumulate ? *o + v : v;
}

// ================================================================================================ 128 x 128 kernel (other shapes)
constexpr int TB = 128;            // output tile edge
constexpr int BK = 64;             // tokens per K-step
constexpr int NSLOT = 3;           // ring slots: two chunks in flight while one is multiplied

__global__ void __launch_bounds__(512) tokred_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B, long ldb,
                                                   float* __restrict__ slab, float* __restrict__ cslab, int Nout, int Kin,
                                                   int steps_total, int steps_per, int tiles_k, int ntiles) {
    constexpr int WN = 4;                              // 2 x 4 waves of 64 x 32
    constexpr int TMI = 4, TNI = 2;                    // 16 x 16 MFMA tiles per wave
    constexpr int CHUNK = BK * TB;                     // elements of one [64][128] sub-chunk
    constexpr int PB = 2;                              // 1-KiB DMA pieces per wave per operand per step
    constexpr int G = 2 * PB;                          // DMA instructions per thread per K-step
    constexpr int SLOT = 2 * CHUNK;                    // elements of one ring slot: a chunk of dy, a chunk of x
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* ring = reinterpret_cast<bf16*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform
    const int seq = xcd_remap(blockIdx.x, gridDim.x);      // contiguous runs of the (slice, tile) sequence per XCD
    const int slice = seq / ntiles, tile = seq - slice * ntiles;
    const int n0 = (tile / tiles_k) * TB, c0 = (tile % tiles_k) * TB;
    const int s_beg = slice * steps_per;
    const int steps = min(steps_per, steps_total - s_beg);

    // DMA source addresses: piece p of a [64][128] sub-chunk = its rows 4p .. 4p+3; lane -> row 4p + (lane >> 4), LDS 16-byte chunk
    // (lane & 15), which holds global chunk (lane & 15) ^ (2 * key(row)) of that row (lds_off<bf16, true, 128>)
    const bf16* pa[PB];
    const bf16* pb[PB];
#pragma unroll
    for (int t = 0; t < PB; ++t) {
        const int p = wave * PB + t, r = 4 * p + (lane >> 4);
        const int key = (r & 3) | ((r >> 1) & 4);
        pa[t] = A + ((long)s_beg * BK + r) * lda + n0 + 8 * ((lane & 15) ^ (key << 1));
        pb[t] = B + ((long)s_beg * BK + r) * ldb + c0 + 8 * ((lane & 15) ^ (key << 1));
    }
    const long stepa = (long)BK * lda, stepb = (long)BK * ldb;
    const unsigned ring_lds = lds_addr(ring);
    auto issue = [&](int slot) {
        const unsigned sa = ring_lds + (unsigned)slot * (unsigned)(SLOT * 2);
#pragma unroll
        for (int t = 0; t < PB; ++t) {
            glds16(pa[t], __builtin_amdgcn_readfirstlane(sa + (unsigned)(wave * PB + t) * 1024u));
            pa[t] += stepa;
        }
#pragma unroll
        for (int t = 0; t < PB; ++t) {
            glds16(pb[t], __builtin_amdgcn_readfirstlane(sa + (unsigned)(CHUNK * 2) + (unsigned)(wave * PB + t) * 1024u));
            pb[t] += stepb;
        }
    };

    const int wm = wave / WN, wn = wave % WN;
    f32x4 acc[TMI][TNI];
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};     // column-sum accumulator: row tile wn of this wave
    const bool do_cs = cslab != nullptr && c0 == 0;
    bf16x8 ones;
#pragma unroll
    for (int q = 0; q < 8; ++q) ones[q] = (bf16)1.0f;

#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s)
        if (s < steps) issue(s);

    int cur = 0, nxt = NSLOT - 1;
    for (int s = 0; s < steps; ++s) {
        // chunk s has landed once at most the younger chunks' DMAs are outstanding (vmcnt counts in issue order)
        const int younger = min(NSLOT - 2, steps - 1 - s);
        if (younger == NSLOT - 2) wait_vm<(NSLOT - 2) * G>();
        else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // every wave's pieces of chunk s are visible; nobody reads slot (s-1) % NSLOT any more
        if (s + NSLOT - 1 < steps) issue(nxt);
        const bf16* cA = ring + (size_t)cur * SLOT;
        const bf16* cB = cA + CHUNK;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            bf16x8 fa[TMI], fb[TNI];
#pragma unroll
            for (int i = 0; i < TMI; ++i) fa[i] = frag_bf16<true, TB>(cA, wm * (16 * TMI) + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < TNI; ++j) fb[j] = frag_bf16<true, TB>(cB, wn * (16 * TNI) + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < TMI; ++i)
#pragma unroll
                for (int j = 0; j < TNI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < TMI; ++i)
                    if (wn == i) cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], cs, 0, 0, 0);
            }
        }
        cur = cur + 1 == NSLOT ? 0 : cur + 1;
        nxt = nxt + 1 == NSLOT ? 0 : nxt + 1;
    }

    // partial tile -> slab[slice][Nout][Kin] (plain 16-byte stores; lane = one row, 4 consecutive columns per MFMA tile)
    const int li = lane & 15, lg = lane >> 4;
    float* so = slab + ((size_t)slice * Nout + n0 + wm * (16 * TMI) + li) * Kin + c0 + wn * (16 * TNI) + 4 * lg;
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j)
            *reinterpret_cast<float4*>(so + (size_t)(i * 16) * Kin + j * 16) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    if (do_cs && lg == 0) cslab[(size_t)slice * Nout + n0 + wm * (16 * TMI) + wn * 16 + li] = cs[0];
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slab[s][i] in slice order; colsum likewise from cslab.  The slices' loads are issued together.
__global__ void __launch_bounds__(256) tokred_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ cslab, int nslice, long n,
                                                            int Nout, float* __restrict__ out, float* __restrict__ colsum, int accumulate) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = accumulate ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < nslice; s0 += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = s0 + q < nslice ? reinterpret_cast<const float4*>(slab + (size_t)(s0 + q) * n)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (s0 + q < nslice) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
        }
        reinterpret_cast<float4*>(out)[i] = a;
    }
    if (colsum) {
        for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < Nout; m += (long)gridDim.x * 256) {
            float a = accumulate ? colsum[m] : 0.f;
            for (int s = 0; s < nslice; ++s) a += cslab[(size_t)s * Nout + m];
            colsum[m] = a;
        }
    }
}

constexpr int MAX_SLICES = 16;

}  // namespace

extern "C" int64_t bf_gemm_tokred_ws_floats(int Nout, int Kin, int64_t M) {
    if (Nout <= 0 || Kin <= 0 || M <= 0) return 0;
    return (int64_t)MAX_SLICES * ((int64_t)Nout * Kin + Nout);          // up to 16 slices of the result and of the column sums
}

// Returns 0 when done, 1 when the shape is not covered (the caller then runs bf_gemm's token-reduction form), < 0 on error.
// ---- deferred slab sums (library-internal, model.hip): bf_gemm_tokred_deferred leaves its slab sum PENDING -- the next deferred call on the
// same stream carries it in extra workgroups of its own launch (FoldRed), bf_gemm_tokred_flush runs what is left as a launch of its own.
// The caller alternates nothing: the two halves of `ws` are used in turn, so a pending sum's slabs are never the next launch's target.
namespace {
struct PendingRed { bool on = false; FoldRed r{}; int NI = 0; };
PendingRed g_pending[64];
bool g_half[64];
PendingRed& pending_slot() { int dev = 0; (void)hipGetDevice(&dev); return g_pending[(dev >= 0 && dev < 64) ? dev : 0]; }
bool& half_slot() { int dev = 0; (void)hipGetDevice(&dev); return g_half[(dev >= 0 && dev < 64) ? dev : 0]; }
constexpr int FOLD_WGS = 40;      // extra workgroups that carry a pending sum (they hold a CU's LDS like the others: 128 + 40 of 256)
}  // namespace
static int tokred_impl(int dtype, int Nout, int Kin, int64_t M, const void* dy, int64_t ldy, const void* x, int64_t ldx, float* out,
                       int accumulate, float* colsum, float* ws, int64_t ws_floats, bool defer, bf_stream_t stream);
int bf_gemm_tokred_flush(hipStream_t st) {
    PendingRed& p = pending_slot();
    if (!p.on) return 0;
    p.on = false;
    const FoldRed& r = p.r;
    const long n = (long)r.Nout * r.Kin;
    const unsigned rblocks = (unsigned)bf_cdiv(std::max<long>(n / 4, r.Nout), 256);
    BfProfScope prof(st, "tokred_reduce_kernel", 0.0, (double)(r.nslice + 1 + (r.accumulate ? 1 : 0)) * n * 4.0);
    if (p.NI == 6) {
        if (r.nslice <= 8) hipLaunchKernelGGL((tokred_pp_reduce_kernel<6, 8>), dim3(rblocks), dim3(256), 0, st, r.slab, r.cslab, r.nslice, r.ntiles, r.tiles_k, r.Nout, r.Kin, r.out, r.colsum, r.accumulate);
        else hipLaunchKernelGGL((tokred_pp_reduce_kernel<6, MAX_SLICES>), dim3(rblocks), dim3(256), 0, st, r.slab, r.cslab, r.nslice, r.ntiles, r.tiles_k, r.Nout, r.Kin, r.out, r.colsum, r.accumulate);
    } else {
        if (r.nslice <= 8) hipLaunchKernelGGL((tokred_pp_reduce_kernel<3, 8>), dim3(rblocks), dim3(256), 0, st, r.slab, r.cslab, r.nslice, r.ntiles, r.tiles_k, r.Nout, r.Kin, r.out, r.colsum, r.accumulate);
        else hipLaunchKernelGGL((tokred_pp_reduce_kernel<3, MAX_SLICES>), dim3(rblocks), dim3(256), 0, st, r.slab, r.cslab, r.nslice, r.ntiles, r.tiles_k, r.Nout, r.Kin, r.out, r.colsum, r.accumulate);
    }
    BF_CHECK_LAUNCH();
    return 0;
}
namespace { bool g_fold_off = false; }
// test hook: run every slab sum as a launch of its own right behind its GEMM (the deferred entry point then behaves like bf_gemm_tokred)
extern "C" void bf_debug_tokred_fold(int on) { g_fold_off = on == 0; }
bool bf_gemm_tokred_pending() { return pending_slot().on; }
const float* bf_gemm_tokred_pending_out() { const PendingRed& p = pending_slot(); return p.on ? p.r.out : nullptr; }      // whose sum is pending
// as bf_gemm_tokred, but the slab sum may stay pending (see above); returns 0 / 1 / < 0 likewise.  A shape outside the ping-pong tiling, or a
// workspace too small for two regions, is summed at once (after whatever was pending).
int bf_gemm_tokred_deferred(int dtype, int Nout, int Kin, int64_t M, const void* dy, int64_t ldy, const void* x, int64_t ldx, float* out,
                            int accumulate, float* colsum, float* ws, int64_t ws_floats, hipStream_t stream) {
    return tokred_impl(dtype, Nout, Kin, M, dy, ldy, x, ldx, out, accumulate, colsum, ws, ws_floats, true, (bf_stream_t)stream);
}
extern "C" int bf_gemm_tokred(int dtype, int Nout, int Kin, int64_t M, const void* dy, int64_t ldy, const void* x, int64_t ldx, float* out,
                              int accumulate, float* colsum, float* ws, int64_t ws_floats, bf_stream_t stream) {
    return tokred_impl(dtype, Nout, Kin, M, dy, ldy, x, ldx, out, accumulate, colsum, ws, ws_floats, false, stream);
}
static int tokred_impl(int dtype, int Nout, int Kin, int64_t M, const void* dy, int64_t ldy, const void* x, int64_t ldx, float* out,
                       int accumulate, float* colsum, float* ws, int64_t ws_floats, bool defer, bf_stream_t stream) {
    if (dtype != BF_DTYPE_BF16) return 1;
    static const int skip_env = bf_knob("BF_TOKRED_SKIP", 0);      // timing experiment (results wrong): the step without its weight-gradient GEMMs
    if (skip_env) return 0;
    const bool pp = Nout % 192 == 0 && Kin % PTN == 0 && M % HR == 0 && M >= 4 * HR;
    if (!pp && (Nout % TB || Kin % TB || M % BK || M < BK)) return 1;
    if (ldy % 8 || ldx % 8) return 1;
    BF_REQUIRE(dy && x && out && ws, "bf_gemm_tokred: null pointer");
    BF_REQUIRE(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)ws & 15) == 0,
               "bf_gemm_tokred: operands must be 16-byte aligned");
    BF_REQUIRE(ldy >= Nout && ldx >= Kin, "bf_gemm_tokred: leading dimensions smaller than the row length");
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)Nout * Kin;
    // Token slices.  More slices = more, shorter workgroups and more slab traffic (2 x slices x |out|, whatever the tile shape).
    // BF_TOKRED_SLICES / BF_TOKRED_TILE (384 or 192 rows) override the defaults.
    // Workgroups per launch: the kernel runs on the side stream beside the caller's data-gradient kernels (192 one-per-CU workgroups,
    // gemm_frame.hip); both queues end up equally long, so what counts is the sum of their CU-time.  Measured in the step (targets of
    // 48 / 64 / 80 / 96 / 128 workgroups: 681 / 693 / 701 / 707 / 699 samples/s before the embed / debed rework; after it, same box,
    // 80 / 96 / 112 / 128 / 160 / 192 / 256: 765 / 768 / 768 / 771 / 767 / 766 / 766): 128, from which the slice count follows per shape
    // (QKV 12 tiles of 192 x 192 x 10, fc1 / fc2 16 x 8, out-projection 4 x 16).
    static const int slices_env = bf_knob("BF_TOKRED_SLICES", 0);      // sweeps: a fixed slice count / tile height / workgroup target
    static const int tile_env = bf_knob("BF_TOKRED_TILE", 0);
    static const int wgs_env = bf_knob("BF_TOKRED_WGS", 128);
    if (pp) {
        const long halves = M / HR;
        const int tiles_k0 = Kin / PTN;
        // 384-row tiles stage 25 % fewer bytes per FLOP, 192-row tiles give twice the workgroups to place beside the caller's kernels: in the
        // step 192 wins (same box: 762 against 759 mixed and 755 all-384 samples/s); BF_TOKRED_TILE=384 keeps the other form reachable
        const bool big = Nout % 384 == 0 && tile_env == 384;
        const int tm = big ? 384 : 192;
        int nslice = slices_env > 0 ? slices_env : std::max(1, wgs_env / ((Nout / tm) * tiles_k0));
        nslice = (int)std::max<long>(1, std::min<long>({(long)nslice, (long)MAX_SLICES, halves / 4}));
        const int halves_per = bf_cdiv(halves, nslice);
        const int ns = bf_cdiv(halves, halves_per);                // slices that actually have tokens
        BF_REQUIRE(ws_floats >= (int64_t)ns * (n + Nout), "bf_gemm_tokred: workspace too small (bf_gemm_tokred_ws_floats)");
        const int tiles_k = Kin / PTN, ntiles = (Nout / tm) * tiles_k;
        // deferred form: this launch's slabs go to one half of the workspace while a pending sum reads the other
        static const bool fold_on = bf_knob("BF_TOKRED_FOLD", 1) != 0;
        const int64_t half_floats = (ws_floats / 2) & ~(int64_t)3;
        const bool fold = defer && fold_on && !g_fold_off && (int64_t)ns * (n + Nout) <= half_floats;
        PendingRed& pend = pending_slot();
        if (!fold && pend.on) { const int frc = bf_gemm_tokred_flush(st); if (frc) return frc; }      // (also: the immediate form may be about to overwrite its slabs)
        if (fold && pend.on && pend.NI != (big ? 6 : 3)) { const int frc = bf_gemm_tokred_flush(st); if (frc) return frc; }
        bool& hf = half_slot();
        float* region = ws;
        if (fold) { region = ws + (hf ? half_floats : 0); hf = !hf; }
        float* slab = region;
        float* cslab = colsum ? region + (size_t)ns * n : nullptr;
        const unsigned rblocks = (unsigned)bf_cdiv(std::max<long>(n / 4, Nout), 256);
        FoldRed red{};
        static const int fold_wgs = bf_knob("BF_TOKRED_FOLD_WGS", FOLD_WGS);
        if (fold && pend.on) { red = pend.r; red.nwg = fold_wgs; }
        const unsigned extra = red.nwg > 0 ? (unsigned)red.nwg : 0u;
#define BF_PP_GO(NIV)                                                                                                                     \
        do {                                                                                                                              \
            {                                                                                                                             \
                BfProfScope prof(st, NIV == 6 ? "tokred_pp_kernel<384x192,h32,ring4>" : "tokred_pp_kernel<192x192,h32,ring4>",           \
                                 2.0 * Nout * Kin * (double)M, (double)M * (Nout + Kin) * 2.0 + (double)n * 4.0);                         \
                static BfPerDeviceOnce attr_once; bool& attr_done = attr_once.flag();                                                                                            \
                constexpr int lds_bytes = NBUF * PPGeom<NIV>::HALFB;                                                                      \
                if (!attr_done) {                                                                                                         \
                    hipError_t e_ = hipFuncSetAttribute((const void*)tokred_pp_kernel<NIV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
                    if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);                                                         \
                    attr_done = true;                                                                                                     \
                }                                                                                                                         \
                hipLaunchKernelGGL(tokred_pp_kernel<NIV>, dim3((unsigned)(ns * ntiles) + extra), dim3(512), lds_bytes, st, (const bf16*)dy, (long)ldy, \
                                   (const bf16*)x, (long)ldx, slab, cslab, Nout, (int)halves, halves_per, tiles_k, ntiles, ns * ntiles, red); \
                BF_CHECK_LAUNCH();                                                                                                        \
            }                                                                                                                             \
            if (fold) {                                                                                                                   \
                pend.on = true; pend.NI = NIV;                                                                                            \
                pend.r = FoldRed{slab, cslab, ns, ntiles, tiles_k, Nout, Kin, out, colsum, accumulate, 0};                                \
                break;                                                                                                                    \
            }                                                                                                                             \
            BfProfScope prof(st, "tokred_reduce_kernel", 0.0, (double)(ns + 1 + (accumulate ? 1 : 0)) * n * 4.0);                         \
            if (ns <= 8) hipLaunchKernelGGL((tokred_pp_reduce_kernel<NIV, 8>), dim3(rblocks), dim3(256), 0, st, slab, cslab, ns, ntiles, tiles_k, Nout, Kin, out, colsum, accumulate); \
            else hipLaunchKernelGGL((tokred_pp_reduce_kernel<NIV, MAX_SLICES>), dim3(rblocks), dim3(256), 0, st, slab, cslab, ns, ntiles, tiles_k, Nout, Kin, out, colsum, accumulate); \
            BF_CHECK_LAUNCH();                                                                                                            \
        } while (0)
        if (big) BF_PP_GO(6); else BF_PP_GO(3);
#undef BF_PP_GO
        return 0;
    }
    { const int frc = bf_gemm_tokred_flush(st); if (frc) return frc; }      // (a pending sum's slabs live in the workspace this form is about to use)
    const long steps = M / BK;
    int nslice = slices_env > 0 ? slices_env : 8;
    nslice = (int)std::max<long>(1, std::min<long>({(long)nslice, (long)MAX_SLICES, steps}));
    BF_REQUIRE(ws_floats >= (int64_t)nslice * (n + Nout), "bf_gemm_tokred: workspace too small (bf_gemm_tokred_ws_floats)");
    const int steps_per = bf_cdiv(steps, nslice);
    const int ns = bf_cdiv(steps, steps_per);                  // slices that actually have tokens
    const int tiles_k = Kin / TB, ntiles = (Nout / TB) * tiles_k;
    float* slab = ws;
    float* cslab = colsum ? ws + (size_t)ns * n : nullptr;
    {
        BfProfScope prof(st, "tokred_kernel<128x128,bk64,slots3>", 2.0 * Nout * Kin * (double)M, (double)M * (Nout + Kin) * 2.0 + (double)n * 4.0);
        static BfPerDeviceOnce attr_once; bool& attr_done = attr_once.flag();
        constexpr int lds_bytes = NSLOT * 2 * BK * TB * 2;
        if (!attr_done) {
            hipError_t e_ = hipFuncSetAttribute((const void*)tokred_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);
            attr_done = true;
        }
        hipLaunchKernelGGL(tokred_kernel, dim3((unsigned)(ns * ntiles)), dim3(512), lds_bytes, st, (const bf16*)dy, (long)ldy, (const bf16*)x,
                           (long)ldx, slab, cslab, Nout, Kin, (int)steps, steps_per, tiles_k, ntiles);
        BF_CHECK_LAUNCH();
    }
    {
        BfProfScope prof(st, "tokred_reduce_kernel", 0.0, (double)(ns + 1 + (accumulate ? 1 : 0)) * n * 4.0);
        const int blocks = (int)std::min<long>(512, (n / 4 + 255) / 256);
        hipLaunchKernelGGL(tokred_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, cslab, ns, n, Nout, out, colsum, accumulate);
        BF_CHECK_LAUNCH();
    }
    return 0;
}

// dW[C][16 of ldo] (+)= act(wide[P][C])^T narrow[P][16] -- or its transpose out[16][C of ldo] -- both operands dense row-major bf16
// (row lengths C and 16), P a multiple of 32.  act = identity, or gelu(wide * sc[frame][c] + sh[frame][c]) with rows_per_frame rows per
// frame (a multiple of 32).  Library-internal (the first embed / last debed stage's weight gradient); 0 = handled, 1 = shape not covered.
// ws: 256 * C * 16 floats.
int bf_tokred_narrow(int dtype, int C, int64_t P, const void* wide, const void* narrow, float* out, int ldo, int accumulate, int transposed,
                     const float* sc, const float* sh, int64_t rows_per_frame, float* ws, int64_t ws_floats, hipStream_t st) {
    static const bool off = bf_knob("BF_TOKRED_NARROW", 1) == 0;
    if (off || dtype != BF_DTYPE_BF16 || (C != 96 && C != 64 && C != 32) || P <= 0 || P % NW_TILE || ldo < (transposed ? C : 16)) return 1;
    if (((uintptr_t)wide | (uintptr_t)narrow | (uintptr_t)ws) & 15) return 1;
    if (sc && (rows_per_frame <= 0 || rows_per_frame % NW_TILE)) return 1;
    // two workgroups per CU where the fragments take the InstanceNorm + GELU in registers: that form is VALU-bound and a wave alternates
    // between waiting for its tile and transforming it (measured 190 us with one wave per SIMD)
    const int wgs = sc ? 512 : 256;
    if (ws_floats < (int64_t)wgs * C * 16) return 1;
    const long tiles = P / NW_TILE;
    const int tpf = sc ? (int)(rows_per_frame / NW_TILE) : 1;
    BfProfScope prof(st, sc ? "tokred_narrow_kernel<gelu>" : "tokred_narrow_kernel", 2.0 * C * 16 * (double)P, (double)P * (C + 16) * 2.0);
#define BF_NARROW_GO(CTV)                                                                                                               \
    do {                                                                                                                                \
        constexpr int lds_bytes = 4 * 2 * (NW_TILE * 16 * CTV * 2 + NW_TILE * 16 * 2);                                                  \
        static_assert(lds_bytes >= 4 * 16 * CTV * 16 * 4, "the reduction buffer aliases the tile rings");                               \
        if (sc) hipLaunchKernelGGL((tokred_narrow_kernel<CTV, true>), dim3(wgs), dim3(256), lds_bytes, st, (const bf16*)wide, (const bf16*)narrow, tiles, ws, sc, sh, tpf); \
        else hipLaunchKernelGGL((tokred_narrow_kernel<CTV, false>), dim3(wgs), dim3(256), lds_bytes, st, (const bf16*)wide, (const bf16*)narrow, tiles, ws, sc, sh, tpf); \
    } while (0)
    if (C == 96) BF_NARROW_GO(6); else if (C == 64) BF_NARROW_GO(4); else BF_NARROW_GO(2);
#undef BF_NARROW_GO
    BF_CHECK_LAUNCH();
    hipLaunchKernelGGL(tokred_narrow_reduce_kernel, dim3(bf_cdiv(C * 16, 256)), dim3(256), 0, st, ws, wgs, C, out, ldo, accumulate, transposed);
    BF_CHECK_LAUNCH();
    return 0;
}
