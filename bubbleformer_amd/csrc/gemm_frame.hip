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

}  // namespace

extern "C" int bf_gemm_inbwd_frames(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                                    const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                                    const float* fscale, int fdiv, bf_stream_t stream) {
    BF_REQUIRE(A && B && x && out && mean && rstd && w, "bf_gemm_inbwd_frames: null pointer");
    static const bool off = []() { const char* v = getenv("BF_FUSE_INBWD"); return v && atoi(v) == 0; }();
    if (off || dtype != BF_DTYPE_BF16 || S != FM || M <= 0 || M % FM || N <= 0 || N % FN || K < FK || K % FK || lda % 8 || ldb % 8) return 1;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_inbwd_frames_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return bf_fail(attr, __FILE__, __LINE__);
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
