// Token-reduction GEMM for gfx950: the weight gradient of every 1x1 conv / Linear of the trunk,
//
//     out[Nout][Kin] (+)= sum_tok dy[tok][Nout] * x[tok][Kin]           (autograd of layers/attention.py:78,121,210,299,
//                                                                         linear_layers.py:18-25; both operands token-major)
//
// The reduction runs over tens of thousands of tokens and the result is a few hundred KB, so the split over workgroups is a split
// of the TOKENS.  What that costs is the partial results: (#workgroups) x (tile bytes) of fp32 leave the chip whatever the tile
// shape.  This kernel therefore uses FEW, LONG slices (4 by default): one workgroup = one 128 x 128 output tile x one token slice,
// 36-144 workgroups per launch -- the launch deliberately does not fill the chip; it runs beside the data-gradient kernels of the
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
constexpr int CHUNK = BK * TB;     // elements of one operand chunk (16 KB)

// WN = waves along the Kin axis: 4 -> 8 waves (2 x 4), 64 x 32 per wave; 2 -> 4 waves (2 x 2), 64 x 64 per wave
template <int NSLOT, int WN>
__global__ void __launch_bounds__(128 * WN) tokred_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B, long ldb,
                                                         float* __restrict__ slab, float* __restrict__ cslab, int Nout, int Kin,
                                                         int steps_total, int steps_per, int tiles_k, int ntiles, int nslice, int mode) {
    constexpr int NW = 2 * WN, NT = 64 * NW;
    constexpr int TN = 8 / WN;                 // 16-column MFMA tiles per wave along Kin
    constexpr int PPW = 16 / NW;               // 1-KiB DMA pieces per wave per operand chunk
    constexpr int G = 2 * PPW;                 // DMA instructions per thread per K-step
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
    const int n0 = (tile / tiles_k) * TB, c0 = (tile % tiles_k) * TB;
    const int s_beg = slice * steps_per;
    const int steps = min(steps_per, steps_total - s_beg);

    // DMA source addresses: piece p = rows 4p .. 4p+3 of the chunk; lane -> row 4p + (lane >> 4), LDS 16-byte chunk (lane & 15),
    // which holds global chunk (lane & 15) ^ (2 * key(row)) of that row (lds_off<bf16, true, 128>)
    const bf16* pa[PPW];
    const bf16* pb[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        const int p = wave * PPW + t, r = 4 * p + (lane >> 4);
        const int key = (r & 3) | ((r >> 1) & 4);
        const int ch = (lane & 15) ^ (key << 1);
        pa[t] = A + ((long)s_beg * BK + r) * lda + n0 + 8 * ch;
        pb[t] = B + ((long)s_beg * BK + r) * ldb + c0 + 8 * ch;
    }
    const long stepa = (long)BK * lda, stepb = (long)BK * ldb;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(lds_addr(ring) + (unsigned)(wave * PPW) * 1024u);      // this wave's first piece
    auto issue = [&](int slot) {
        const unsigned sa = ring_lds + (unsigned)slot * (unsigned)(2 * CHUNK * 2);
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            glds16(pa[t], sa + t * 1024u);
            pa[t] += stepa;
        }
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            glds16(pb[t], sa + (unsigned)(CHUNK * 2) + t * 1024u);
            pb[t] += stepb;
        }
    };

    const int wm = wave / WN, wn = wave % WN;
    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int CSN = 4 / WN;               // column-sum accumulators per wave
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
        if (younger == NSLOT - 2) wait_vm<(NSLOT - 2) * G>();
        else if (NSLOT > 3 && younger == 1) wait_vm<G>();
        else if (NSLOT > 4 && younger == 2) wait_vm<2 * G>();
        else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // every wave's pieces of chunk s are visible; nobody reads slot (s-1) % NSLOT any more
        if (s + NSLOT - 1 < steps) issue(nxt);
        const bf16* cA = ring + (size_t)cur * (2 * CHUNK);
        const bf16* cB = cA + CHUNK;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            bf16x8 fa[4], fb[TN];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = frag_bf16<true, TB>(cA, wm * 64 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = frag_bf16<true, TB>(cB, wn * (16 * TN) + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            if (do_cs) {          // column sums of dy (bias gradient): row groups wn, wn + WN, .. of this wave's 64 rows against an all-ones operand
#pragma unroll
                for (int q = 0; q < CSN; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (wn + q * WN == i) cs[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], cs[q], 0, 0, 0);
            }
        }
        cur = cur + 1 == NSLOT ? 0 : cur + 1;
        nxt = nxt + 1 == NSLOT ? 0 : nxt + 1;
    }

    // partial tile -> slab[slice][Nout][Kin] (plain 16-byte stores; lane = one row, 4 consecutive columns per MFMA tile)
    const int li = lane & 15, lg = lane >> 4;
    float* so = slab + ((size_t)slice * Nout + n0 + wm * 64 + li) * Kin + c0 + wn * (16 * TN) + 4 * lg;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            *reinterpret_cast<float4*>(so + (size_t)(i * 16) * Kin + j * 16) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    if (do_cs && lg == 0) {
#pragma unroll
        for (int q = 0; q < CSN; ++q) cslab[(size_t)slice * Nout + n0 + wm * 64 + (wn + q * WN) * 16 + li] = cs[q][0];
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slab[s][i] in slice order; colsum likewise from cslab
__global__ void __launch_bounds__(256) tokred_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ cslab, int nslice, long n,
                                                            int Nout, float* __restrict__ out, float* __restrict__ colsum, int accumulate) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = accumulate ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < nslice; ++s) {
            const float4 v = reinterpret_cast<const float4*>(slab + (size_t)s * n)[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4*>(out)[i] = a;
    }
    if (colsum && blockIdx.x == 0) {
        for (int m = threadIdx.x; m < Nout; m += 256) {
            float a = accumulate ? colsum[m] : 0.f;
            for (int s = 0; s < nslice; ++s) a += cslab[(size_t)s * Nout + m];
            colsum[m] = a;
        }
    }
}

int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

int pick_slices(int Nout, int Kin, long steps) {
    static const int max_slices = env_int("BF_TOKRED_SLICES", 4);
    int ns = max_slices < 1 ? 1 : max_slices > 8 ? 8 : max_slices;
    if (ns > steps) ns = (int)steps;
    return ns;
}

}  // namespace

extern "C" int64_t bf_gemm_tokred_ws_floats(int Nout, int Kin, int64_t M) {
    if (Nout <= 0 || Kin <= 0 || M <= 0) return 0;
    return (int64_t)8 * ((int64_t)Nout * Kin + Nout);           // up to 8 slices of the result and of the column sums
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
    const long steps = M / BK;
    const int nslice = pick_slices(Nout, Kin, steps);
    const long n = (long)Nout * Kin;
    BF_REQUIRE(ws_floats >= (int64_t)nslice * (n + Nout), "bf_gemm_tokred: workspace too small (bf_gemm_tokred_ws_floats)");
    const int steps_per = bf_cdiv(steps, nslice);
    const int ns = bf_cdiv(steps, steps_per);                  // slices that actually have tokens
    const int tiles_k = Kin / TB, ntiles = (Nout / TB) * tiles_k;
    float* slab = ws;
    float* cslab = colsum ? ws + (size_t)ns * n : nullptr;
    hipStream_t st = (hipStream_t)stream;
    static const int mode_env = env_int("BF_TOKRED_MODE", 0);
    const int mode = (mode_env == 1 && ntiles <= 32 && ns <= 4) ? 1 : 0;
    const unsigned grid = mode == 1 ? 8u * (unsigned)ntiles : (unsigned)(ns * ntiles);
    static const int nslot_env = env_int("BF_TOKRED_SLOTS", 3);
    {
        static thread_local char pname[64];
        snprintf(pname, sizeof(pname), "tokred_kernel<slots%d>", nslot_env);
        BfProfScope prof(st, pname, 2.0 * Nout * Kin * (double)M, (double)M * (Nout + Kin) * 2.0 + (double)n * 4.0);
#define BF_TOKRED_GO(NSLOT)                                                                                                               \
        do {                                                                                                                              \
            static bool attr_done = false;                                                                                                \
            constexpr int lds_bytes = NSLOT * 2 * CHUNK * 2;                                                                              \
            if (!attr_done) {                                                                                                             \
                hipError_t e_ = hipFuncSetAttribute((const void*)tokred_kernel<NSLOT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
                if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);                                                             \
                attr_done = true;                                                                                                         \
            }                                                                                                                             \
            hipLaunchKernelGGL((tokred_kernel<NSLOT, 4>), dim3(grid), dim3(512), lds_bytes, st, (const bf16*)dy, (long)ldy, (const bf16*)x,  \
                               (long)ldx, slab, cslab, Nout, Kin, (int)steps, steps_per, tiles_k, ntiles, ns, mode);                      \
        } while (0)
        if (nslot_env == 2) BF_TOKRED_GO(2);
        else if (nslot_env == 4) BF_TOKRED_GO(4);
        else BF_TOKRED_GO(3);
#undef BF_TOKRED_GO
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
