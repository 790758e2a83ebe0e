// Token-reduction GEMM for gfx950: the weight gradient of every 1x1 conv / Linear of the trunk,
//
//     out[Nout][Kin] (+)= sum_tok dy[tok][Nout] * x[tok][Kin]           (autograd of layers/attention.py:78,121,210,299,
//                                                                         linear_layers.py:18-25; both operands token-major)
//
// The reduction runs over tens of thousands of tokens and the result is a few hundred KB, so the split over workgroups is a split
// of the TOKENS.  What that costs is the partial results: (#workgroups) x (tile bytes) of fp32 leave the chip whatever the tile
// shape.  This kernel therefore uses FEW, LONG slices (8): one workgroup = one 128 x 128 (or 384 x 128) output tile x one token slice,
// 72-288 workgroups per launch -- the launch deliberately does not fill the chip; it runs beside the data-gradient kernels of the
// caller's stream -- and its partial tile goes to a slab with plain 16-byte stores.  A second tiny kernel adds the slabs in slice
// order: the result is bit-reproducible run to run (no float atomics anywhere) and the slab traffic is 2 x 4 x |out| instead of
// the split-K atomics' 9-19 x |out|.
//
// Pipeline (cdna_hip_programming.md section 5, "glds ... counted vmcnt ... raw s_barrier"): operand chunks of 64 tokens x 128
// channels go global -> LDS directly (global_load_lds_dwordx4, no staging registers) into a ring of NSLOT slots; NSLOT-1 chunks are
// in flight while one is multiplied; one raw s_barrier per K-step with a counted s_waitcnt vmcnt, never 0 inside the loop.  The LDS
// image is the swizzled [k][128] tile of gemm_common.h (conflict-free ds_read_b64_tr_b16); the DMA writes LDS lane-linearly, so the
// swizzle is applied to each lane's SOURCE address (a permutation of the 16-byte chunks inside a 256-byte row).
// The bias gradient colsum(dy) comes out of the same pass: one extra MFMA per K-step against an all-ones operand.
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

namespace {
using namespace bfgemm;

constexpr int TB = 128;            // output tile edge
constexpr int BK = 64;             // tokens per K-step

// MT = 128-row sub-chunks of the Nout side of a tile: 1 -> 128 x 128 tile, 2 x 4 waves of 64 x 32; 3 -> 384 x 128 tile, 4 x 2 waves of
// 96 x 64 (one and a half times the FLOPs per DMA byte: a CU takes in ~55 GB/s from L2, which is what bounds a step -- measured: the
// 128 x 128 form ran 0.6 us per 32 KB step whatever the ring depth).  8 waves either way.
template <int NSLOT, int MT, int BKT>      // BKT: tokens per K-step (64, or 32 so that four 32 KB slots of the tall tile fit the LDS)
__global__ void __launch_bounds__(512) tokred_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B, long ldb,
                                                   float* __restrict__ slab, float* __restrict__ cslab, int Nout, int Kin,
                                                   int steps_total, int steps_per, int tiles_k, int ntiles, int nslice, int mode) {
    constexpr int TM = 128 * MT;                       // tile rows (Nout side)
    constexpr int WM = MT == 1 ? 2 : 4, WN = 8 / WM;   // waves along Nout / Kin
    constexpr int TMI = TM / WM / 16, TNI = TB / WN / 16;      // 16 x 16 MFMA tiles per wave: 4 x 2 or 6 x 4
    constexpr int CHUNK = BKT * TB;                    // elements of one [BKT][128] sub-chunk
    constexpr int PPC = BKT / 4;                       // 1-KiB DMA pieces per sub-chunk
    constexpr int PA = PPC * MT / 8, PB = PPC / 8;     // pieces per wave per step: Nout side, Kin side
    constexpr int G = PA + PB;                         // DMA instructions per thread per K-step
    constexpr int SLOT = (MT + 1) * CHUNK;             // elements of one ring slot: MT sub-chunks [64][128] of dy, one of x
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* ring = reinterpret_cast<bf16*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform
    int tile, slice;
    if (mode == 1) {            // slice s on the workgroups whose id is 2s mod 8 (one XCD per slice under round-robin placement; speed only)
        const int x = blockIdx.x & 7;
        if ((x & 1) || (x >> 1) >= nslice) return;
        slice = x >> 1; tile = blockIdx.x >> 3;
        if (tile >= ntiles) return;
    } else {                    // contiguous runs of the (slice, tile) sequence per XCD: a slice's tiles share two L2s
        const int seq = xcd_remap(blockIdx.x, gridDim.x);
        slice = seq / ntiles; tile = seq - slice * ntiles;
    }
    const int n0 = (tile / tiles_k) * TM, c0 = (tile % tiles_k) * TB;
    const int s_beg = slice * steps_per;
    const int steps = min(steps_per, steps_total - s_beg);

    // DMA source addresses: piece p of a [64][128] sub-chunk = its rows 4p .. 4p+3; lane -> row 4p + (lane >> 4), LDS 16-byte chunk
    // (lane & 15), which holds global chunk (lane & 15) ^ (2 * key(row)) of that row (lds_off<bf16, true, 128>)
    const bf16* pa[PA];
    const bf16* pb[PB];
    unsigned la[PA];                                   // LDS byte offset of each Nout-side piece inside the slot
#pragma unroll
    for (int t = 0; t < PA; ++t) {
        const int pg = wave * PA + t, sub = pg / PPC, p = pg % PPC, r = 4 * p + (lane >> 4);
        const int key = (r & 3) | ((r >> 1) & 4);
        pa[t] = A + ((long)s_beg * BKT + r) * lda + n0 + sub * TB + 8 * ((lane & 15) ^ (key << 1));
        la[t] = (unsigned)(sub * CHUNK * 2 + p * 1024);
    }
#pragma unroll
    for (int t = 0; t < PB; ++t) {
        const int p = wave * PB + t, r = 4 * p + (lane >> 4);
        const int key = (r & 3) | ((r >> 1) & 4);
        pb[t] = B + ((long)s_beg * BKT + r) * ldb + c0 + 8 * ((lane & 15) ^ (key << 1));
    }
    const long stepa = (long)BKT * lda, stepb = (long)BKT * ldb;
    const unsigned ring_lds = lds_addr(ring);
    auto issue = [&](int slot) {
        const unsigned sa = ring_lds + (unsigned)slot * (unsigned)(SLOT * 2);
#pragma unroll
        for (int t = 0; t < PA; ++t) {
            glds16(pa[t], __builtin_amdgcn_readfirstlane(sa + la[t]));
            pa[t] += stepa;
        }
#pragma unroll
        for (int t = 0; t < PB; ++t) {
            glds16(pb[t], __builtin_amdgcn_readfirstlane(sa + (unsigned)(MT * CHUNK * 2) + (unsigned)(wave * PB + t) * 1024u));
            pb[t] += stepb;
        }
    };

    const int wm = wave / WN, wn = wave % WN;
    f32x4 acc[TMI][TNI];
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int CSN = TMI / WN;             // column-sum accumulators per wave: row tiles wn, wn + WN, ...
    f32x4 cs[CSN];
#pragma unroll
    for (int q = 0; q < CSN; ++q) cs[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_cs = cslab != nullptr && c0 == 0;
    bf16x8 ones;
#pragma unroll
    for (int q = 0; q < 8; ++q) ones[q] = (bf16)1.0f;

    // NSLOT-1 chunks in flight before the first one is needed
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s)
        if (s < steps) issue(s);

    int cur = 0, nxt = NSLOT - 1;
    for (int s = 0; s < steps; ++s) {
        // chunk s has landed once at most the younger chunks' DMAs are outstanding (vmcnt counts in issue order)
        const int younger = min(NSLOT - 2, steps - 1 - s);
        if (NSLOT > 2 && younger == NSLOT - 2) wait_vm<(NSLOT > 2 ? NSLOT - 2 : 0) * G>();
        else if (NSLOT > 3 && younger == 1) wait_vm<G>();
        else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // every wave's pieces of chunk s are visible; nobody reads slot (s-1) % NSLOT any more
        if (s + NSLOT - 1 < steps) issue(nxt);
        const bf16* cA = ring + (size_t)cur * SLOT;
        const bf16* cB = cA + MT * CHUNK;
#pragma unroll
        for (int kk = 0; kk < BKT; kk += 32) {
            bf16x8 fa[TMI], fb[TNI];
#pragma unroll
            for (int i = 0; i < TMI; ++i) {
                const int r = wm * (16 * TMI) + i * 16;                       // row of the tile: sub-chunk r / 128, row r % 128 inside it
                fa[i] = frag_bf16<true, TB>(cA + (r >> 7) * CHUNK, r & 127, kk, lane);
            }
#pragma unroll
            for (int j = 0; j < TNI; ++j) fb[j] = frag_bf16<true, TB>(cB, wn * (16 * TNI) + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < TMI; ++i)
#pragma unroll
                for (int j = 0; j < TNI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            if (do_cs) {          // column sums of dy (bias gradient): row tiles wn, wn + WN, .. of this wave against an all-ones operand
#pragma unroll
                for (int q = 0; q < CSN; ++q)
#pragma unroll
                    for (int i = 0; i < TMI; ++i)
                        if (wn + q * WN == i) cs[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], cs[q], 0, 0, 0);
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
    if (do_cs && lg == 0) {
#pragma unroll
        for (int q = 0; q < CSN; ++q) cslab[(size_t)slice * Nout + n0 + wm * (16 * TMI) + (wn + q * WN) * 16 + li] = cs[q][0];
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slab[s][i] in slice order; colsum likewise from cslab.  The slices' loads are issued together
// (batches of 8 independent 16-byte loads per thread) and added in slice order: a loop of load-then-add ran at one memory round trip per
// slice (24 us per launch at 8 slices for 18 MB).
__global__ void __launch_bounds__(256) tokred_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ cslab, int nslice, long n,
                                                            int Nout, float* __restrict__ out, float* __restrict__ colsum, int accumulate, int cs_mode) {
    if (colsum && cs_mode == 0 && blockIdx.x == 0) {      // single-block column sum (the default, see the host side)
        for (int m = threadIdx.x; m < Nout; m += 256) {
            float a = accumulate ? colsum[m] : 0.f;
            for (int s = 0; s < nslice; ++s) a += cslab[(size_t)s * Nout + m];
            colsum[m] = a;
        }
    }
    if (cs_mode == 0) colsum = nullptr;
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
    if (colsum) {       // spread over the LAST blocks of the grid (one block doing all Nout columns was the launch's long pole: 6 x 8 dependent loads)
        const long t = ((long)gridDim.x - 1 - blockIdx.x) * 256 + threadIdx.x;
        for (long m = t; m < Nout; m += (long)gridDim.x * 256) {
            float a = accumulate ? colsum[m] : 0.f;
            for (int s0 = 0; s0 < nslice; s0 += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = s0 + q < nslice ? cslab[(size_t)(s0 + q) * Nout + m] : 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (s0 + q < nslice) a += v[q];
            }
            colsum[m] = a;
        }
    }
}

int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

int pick_slices(long steps, int dflt) {
    static const int env = env_int("BF_TOKRED_SLICES", 0);
    int ns = env > 0 ? env : dflt;
    ns = ns < 1 ? 1 : ns > 16 ? 16 : ns;
    if (ns > steps) ns = (int)steps;
    return ns;
}

}  // namespace

extern "C" int64_t bf_gemm_tokred_ws_floats(int Nout, int Kin, int64_t M) {
    if (Nout <= 0 || Kin <= 0 || M <= 0) return 0;
    return (int64_t)16 * ((int64_t)Nout * Kin + Nout);          // up to 16 slices of the result and of the column sums
}

// Returns 0 when done, 1 when the shape is not covered (the caller then runs bf_gemm's token-reduction form), < 0 on error.
extern "C" int bf_gemm_tokred(int dtype, int Nout, int Kin, int64_t M, const void* dy, int64_t ldy, const void* x, int64_t ldx, float* out,
                              int accumulate, float* colsum, float* ws, int64_t ws_floats, bf_stream_t stream) {
    static const int enabled = env_int("BF_TOKRED", 1);
    if (!enabled || dtype != BF_DTYPE_BF16) return 1;
    if (Nout % TB || Kin % TB || M % BK || M < BK || ldy % 8 || ldx % 8) return 1;
    BF_REQUIRE(dy && x && out && ws, "bf_gemm_tokred: null pointer");
    BF_REQUIRE(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)ws & 15) == 0,
               "bf_gemm_tokred: operands must be 16-byte aligned");
    // BF_TOKRED_TALL=1: 384 x 128 tiles (1.5 x the FLOPs per DMA byte, a third as many tiles), 32-token steps so that FOUR 32 KB slots
    // fit the LDS.  Measured and NOT the default: a workgroup reaches 49 % MFMA utilisation instead of 40 %, not 1.5 x (both forms
    // sit at ~45 GB/s of DMA per CU behind one barrier per step), so on a third as many workgroups a launch takes 82-88 us instead of
    // 57 alone; in the training step the side stream then outlasts the backward's critical path: 558 (6 slices) / 581 (8) samples/s
    // against 632 with 128 x 128 tiles, although the caller's own kernels run faster beside it (data gradient 54 us instead of 66).
    static const int tall_env = env_int("BF_TOKRED_TALL", 0);
    const bool tall = tall_env != 0 && Nout % 384 == 0;
    const int tm = tall ? 384 : TB, bkt = tall ? 32 : BK;
    const long steps = M / bkt;
    // 8 token slices (216-288 workgroups for the trunk's shapes; 4 until the round-2 kernels shifted the balance of the two queues: measured
    // on the final tree 4 / 6 / 8 / 10 / 12 / 16 slices -> 649 / 662 / 670 / 634-642 / 626-630 / 619 samples/s)
    // (slice counts chosen per shape for 200-256 workgroups -- 9 / 16 / 7 for the QKV / out-projection / MLP shapes: 670-673, no better)
    const int nslice = pick_slices(steps, tall ? (Nout / 384 >= 3 ? 6 : 8) : 8);
    const long n = (long)Nout * Kin;
    BF_REQUIRE(ws_floats >= (int64_t)nslice * (n + Nout), "bf_gemm_tokred: workspace too small (bf_gemm_tokred_ws_floats)");
    const int steps_per = bf_cdiv(steps, nslice);
    const int ns = bf_cdiv(steps, steps_per);                  // slices that actually have tokens
    const int tiles_k = Kin / TB, ntiles = (Nout / tm) * tiles_k;
    float* slab = ws;
    float* cslab = colsum ? ws + (size_t)ns * n : nullptr;
    hipStream_t st = (hipStream_t)stream;
    static const int mode_env = env_int("BF_TOKRED_MODE", 0);
    const int mode = (mode_env == 1 && ntiles <= 32 && ns <= 4) ? 1 : 0;
    const unsigned grid = mode == 1 ? 8u * (unsigned)ntiles : (unsigned)(ns * ntiles);
    static const int nslot_env = env_int("BF_TOKRED_SLOTS", 0);
    const int nslot = tall ? 4 : (nslot_env >= 2 && nslot_env <= 4 ? nslot_env : 3);
    {
        static thread_local char pname[64];
        snprintf(pname, sizeof(pname), "tokred_kernel<%dx128,bk%d,slots%d>", tm, bkt, nslot);
        BfProfScope prof(st, pname, 2.0 * Nout * Kin * (double)M, (double)M * (Nout + Kin) * 2.0 + (double)n * 4.0);
#define BF_TOKRED_GO(NSLOT, MTV, BKV)                                                                                                     \
        do {                                                                                                                              \
            static bool attr_done = false;                                                                                                \
            constexpr int lds_bytes = NSLOT * (MTV + 1) * BKV * TB * 2;                                                                   \
            if (!attr_done) {                                                                                                             \
                hipError_t e_ = hipFuncSetAttribute((const void*)tokred_kernel<NSLOT, MTV, BKV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
                if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);                                                             \
                attr_done = true;                                                                                                         \
            }                                                                                                                             \
            hipLaunchKernelGGL((tokred_kernel<NSLOT, MTV, BKV>), dim3(grid), dim3(512), lds_bytes, st, (const bf16*)dy, (long)ldy, (const bf16*)x,  \
                               (long)ldx, slab, cslab, Nout, Kin, (int)steps, steps_per, tiles_k, ntiles, ns, mode);                      \
        } while (0)
        if (tall) BF_TOKRED_GO(4, 3, 32);
        else if (nslot == 2) BF_TOKRED_GO(2, 1, 64);
        else if (nslot == 4) BF_TOKRED_GO(4, 1, 64);
        else BF_TOKRED_GO(3, 1, 64);
#undef BF_TOKRED_GO
        BF_CHECK_LAUNCH();
    }
    {
        BfProfScope prof(st, "tokred_reduce_kernel", 0.0, (double)(ns + 1 + (accumulate ? 1 : 0)) * n * 4.0);
        const int blocks = (int)std::min<long>(512, (n / 4 + 255) / 256);
        // BF_TOKRED_CS=1: the column sums spread over the grid with batched loads -- the reduce drops from 15.5 to 7.6 us alone and from 20 to
        // 10.6 us in the step, and the STEP gets slower (658-660 vs 667-669 samples/s, three A/B pairs on two boxes; tokred_kernel itself
        // 56 -> 63 us beside the main queue).  An explicit pause of 5 / 10 / 20 us after the reduce is no substitute (664 / 661 / 630): the
        // step's two queues sit at an operating point that the side queue's exact timing decides.  Default: the single-block form.
        static const int cs_mode = env_int("BF_TOKRED_CS", 0);
        hipLaunchKernelGGL(tokred_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, cslab, ns, n, Nout, out, colsum, accumulate, cs_mode);
        BF_CHECK_LAUNCH();
    }
    return 0;
}
