// out[p][n] = sum_{q, c} f(map[pixel(p, q)][c]) * W[(q, c)][n] for the 2x2/stride-2 stages at E/4 = 96 channels: the HMLPEmbed
// convolutions after the first (layers/patching.py:30-56; f = GELU(InstanceNorm affine), folded into the operand) and the data
// gradients of the HMLPDebed transposed convolutions (layers/patching.py:92-104 under autograd; f = identity).  p runs over the coarse
// grid [F][gh][gw], pixel(p, q) = (2y + q / 2, 2x + q % 2) on the fine grid, K = 4 * 96 = 384, N = 96.
//
// These are the GEMMs over the largest maps of the step (226 MB at the bench shape) with only 96 output columns: 43 FLOP per byte of
// the map -- HBM-bound at any decent matrix rate -- and a tile-per-workgroup kernel that stages both operands through registers ran
// them at ~2.3 TB/s.  Here the whole [96][384] weight sits in LDS (75 KB: two workgroups per CU) and is the MFMA A operand (plain
// 16-byte LDS reads); a wave owns 32 coarse rows, fetches the map rows STRAIGHT into B-operand registers (lane = coarse row, 8
// consecutive channels: 16-byte loads, one 2x2 position ahead of the products), applies f there -- each element exactly once --
// and leaves through v_permlane16_swap with 8 consecutive output columns per lane (16-byte stores).  Waves never meet after the
// weight load.
#include "bf_common.h"
#include <algorithm>
#include <stdint.h>

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
constexpr int GW = 4;                                   // waves per workgroup
// transposing read of a 4-row x 16-col block of a bf16 LDS tile: lane i16 of the 16-lane group gets column c0 + i16 of rows r0..r0+3
__device__ __forceinline__ s16x4 tr4g(const bf16* tile, int ld, int r0, int c0, int lane) {
    const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tile + (r0 + q) * ld + c0 + 4 * p));
}
__device__ __forceinline__ bf16x8 cat8(s16x4 lo, s16x4 hi) {
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

struct GatherArgs {
    const bf16 *patches, *w0; // REB: the fine map is W0 . patch ([pixels][16] x [C0][16]) and is rebuilt per tile instead of read
    const bf16* map;          // [F][2 gh][2 gw][C0]
    const bf16* w;            // kn: [4 C0][N] (n contiguous), else [N][4 C0] (k contiguous)
    bf16* out;                // [F gh gw][N]
    const float *sc, *sh;     // [F][C0] (PRO) : f(x) = gelu(x sc + sh)
    int w_kn, F, gh, gw, tiles;
};

template <int NCB, int NNB> constexpr int gather_lds_bytes() { return 16 * NNB * (64 * NCB + 8) * 2; }

template <int NCB, int NNB, bool PRO, bool REB = false>
__global__ void __launch_bounds__(64 * GW, 2) gather_gemm_kernel(GatherArgs a) {
    constexpr int C0 = 16 * NCB, N = 16 * NNB, K = 4 * C0, LDK = K + 8, NS = C0 / 32;
    static_assert(C0 % 32 == 0 && NNB % 2 == 0, "32-channel slabs, column blocks in pairs");
    extern __shared__ __attribute__((aligned(16))) char smem_gg[];
    bf16* Wt = reinterpret_cast<bf16*>(smem_gg);                     // [N][LDK]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    if (a.w_kn) {
        for (int t = tid; t < K * (N / 8); t += 64 * GW) {          // 16-byte pieces of a k-row, scattered down a column of Wt
            const int k = t / (N / 8), n0 = 8 * (t - k * (N / 8));
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.w + (long)k * N + n0);
#pragma unroll
            for (int j = 0; j < 8; ++j) Wt[(n0 + j) * LDK + k] = v[j];
        }
    } else {
        for (int t = tid; t < N * (K / 8); t += 64 * GW) {
            const int n = t / (K / 8), k0 = 8 * (t - n * (K / 8));
            *reinterpret_cast<bf16x8*>(Wt + n * LDK + k0) = *reinterpret_cast<const bf16x8*>(a.w + (long)n * K + k0);
        }
    }
    __syncthreads();
    // this wave's run of 32-row tiles (contiguous: mostly one frame, whose affine then stays in registers)
    const int nw = gridDim.x * GW, wv = blockIdx.x * GW + wave;
    const int t_beg = (int)((long)a.tiles * wv / nw), t_end = (int)((long)a.tiles * (wv + 1) / nw);
    if (t_beg >= t_end) return;
    const unsigned gw = (unsigned)a.gw, tpf = (unsigned)(a.gh * a.gw) / 32u;
    float csc[PRO ? NS : 1][8], csh[PRO ? NS : 1][8];
    int cf = -1;
    // B-operand registers of one 2x2 position: [row block][slab], lane (coarse row i16, channels 32 s + 8 g ..)
    // REB: only the 8 bytes of the pixel's patch row this lane multiplies (k = 4g..4g+3) travel; the map rows come out of one
    // v_mfma_f32_16x16x16_bf16 per 16 channels (A = this lane's row of W0, B = the patch rows) as 4 consecutive channels per lane, so
    // slab s takes its k-slots in the order {32 s + 4 g + j} U {32 s + 16 + 4 g + j} -- the weight operand is read in the same order
    constexpr int NR = REB ? 1 : NS;
    bf16x8 cur[2][NR], nxt[2][NR];
    s16x4 w0f[REB ? NCB : 1];
    if (REB) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) w0f[cb] = *reinterpret_cast<const s16x4*>(a.w0 + (16 * cb + i16) * 16 + 4 * g);
    }
    auto issue = [&](int tt, int q, bf16x8 (&dst)[2][NR]) __attribute__((always_inline)) {
        const unsigned f = (unsigned)tt / tpf, tl = (unsigned)tt - f * tpf;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const unsigned pl = 32u * tl + 16u * rb + (unsigned)i16, y = pl / gw, x = pl - y * gw;     // per lane: a 16-row block may wrap an image row
            const long pix = ((long)f * (2 * a.gh) + 2 * y + (q >> 1)) * (2L * a.gw) + 2 * x + (q & 1);
            if (REB) {
                const s16x4 v = *reinterpret_cast<const s16x4*>(a.patches + pix * 16 + 4 * g);
                dst[rb][0] = __builtin_bit_cast(bf16x8, s16x8{v[0], v[1], v[2], v[3], 0, 0, 0, 0});
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s) dst[rb][s] = *reinterpret_cast<const bf16x8*>(a.map + pix * C0 + 32 * s + 8 * g);
            }
        }
    };
    issue(t_beg, 0, cur);
    for (int tt = t_beg; tt < t_end; ++tt) {
        if (PRO) {
            const int f = (int)((unsigned)tt / tpf);
            if (f != cf) {
                cf = f;
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int c = REB ? 32 * s + 16 * (j >> 2) + 4 * g + (j & 3) : 32 * s + 8 * g + j;
                        csc[s][j] = a.sc[(long)f * C0 + c]; csh[s][j] = a.sh[(long)f * C0 + c];
                    }
            }
        }
        f32x4 acc[NNB][2];
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) { acc[nb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[nb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3) issue(tt, q + 1, nxt);
            else if (tt + 1 < t_end) issue(tt + 1, 0, nxt);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                bf16x8 fb[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    if (REB) {
                        const s16x8 pv = __builtin_bit_cast(s16x8, cur[rb][0]);
                        const s16x4 pb = {pv[0], pv[1], pv[2], pv[3]};
                        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                        const f32x4 lo = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0f[2 * s], pb, z4, 0, 0, 0), hi = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0f[2 * s + 1], pb, z4, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {      // rounded to bf16 like the rows the statistics were taken of
                            fb[rb][j] = (bf16)gelu_fast(fmaf((float)(bf16)lo[j], csc[s][j], csh[s][j]));
                            fb[rb][4 + j] = (bf16)gelu_fast(fmaf((float)(bf16)hi[j], csc[s][4 + j], csh[s][4 + j]));
                        }
                    } else if (PRO) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) fb[rb][j] = (bf16)gelu_fast(fmaf((float)cur[rb][s][j], csc[s][j], csh[s][j]));
                    } else fb[rb] = cur[rb][s];
                }
#pragma unroll
                for (int nb = 0; nb < NNB; ++nb) {
                    const bf16* wrow = Wt + (16 * nb + i16) * LDK + C0 * q + 32 * s;
                    const bf16x8 aw = REB ? cat8(*reinterpret_cast<const s16x4*>(wrow + 4 * g), *reinterpret_cast<const s16x4*>(wrow + 16 + 4 * g))
                                          : *reinterpret_cast<const bf16x8*>(wrow + 8 * g);
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) acc[nb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw, fb[rb], acc[nb][rb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);          // or the scheduler lifts every slab's 6 weight reads to the top of the tile: 288 registers
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int s = 0; s < NR; ++s) cur[rb][s] = nxt[rb][s];
        }
        // acc[nb][rb]: coarse row 16 rb + i16, columns 16 nb + 4 g .. +3.  Exchanging the odd lane rows of block 2 pp with the even lane
        // rows of block 2 pp + 1 leaves 8 consecutive columns at 32 pp + 16 (g & 1) + 8 (g >> 1)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            bf16* dst = a.out + ((long)tt * 32 + 16 * rb + i16) * N + (g & 1) * 16 + (g >> 1) * 8;
#pragma unroll
            for (int pp = 0; pp < NNB / 2; ++pp) {
                bf16x8 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * pp][rb][r]), __float_as_uint(acc[2 * pp + 1][rb][r]), false, false);
                    o[r] = (bf16)__uint_as_float(sw[0]); o[4 + r] = (bf16)__uint_as_float(sw[1]);
                }
                *reinterpret_cast<bf16x8*>(dst + 32 * pp) = o;
            }
        }
    }
}

// The transposed direction (HMLPDebed's ConvTranspose2d(k=2, s=2) stages, layers/patching.py:80-104): map[pixel(p, q)][c] = sum_k f(a[p][k]) *
// W[(q, c)][k], K = 96, 4 x 96 output columns scattered to the 2x2 positions.  Same organisation: the [384][96] weight in LDS, a wave
// fetches and transforms its 32 coarse rows once and sweeps the four positions; 16-byte stores of 8 consecutive channels.  Optionally
// leaves the InstanceNorm statistics of what it stored as {mean, centred second moment} per tile (128 fine pixels) and channel, in the
// slice layout bf_in_stats_merge_slices finishes -- the 226 MB map is not read again for them.
struct ScatterArgs {
    const bf16* a;            // [F gh gw][K]
    const bf16* w;            // [4 C0][K] (k contiguous), or kn: [K][4 C0]
    bf16* map;                // [F][2 gh][2 gw][C0]
    const float *sc, *sh;     // [F][K] (PRO)
    float* part;              // optional: [F][tiles per frame][C0][2]
    int w_kn, F, gh, gw, tiles;
};
template <int NKB, int NCB> constexpr int scatter_lds_bytes() { return 4 * 16 * NCB * (32 * NKB + 8) * 2; }

__device__ __forceinline__ float row16_total(float v) {      // sum over the 16 lanes of a row, in every lane (DPP: xor 1, xor 2, half mirror, mirror)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

template <int NKB, int NCB, bool PRO>
__global__ void __launch_bounds__(64 * GW, 2) scatter_gemm_kernel(ScatterArgs a) {
    constexpr int K = 32 * NKB, C0 = 16 * NCB, LDK = K + 8;
    static_assert(NCB % 2 == 0, "column blocks in pairs");
    extern __shared__ __attribute__((aligned(16))) char smem_sg[];
    bf16* Wt = reinterpret_cast<bf16*>(smem_sg);                     // [4 C0][LDK]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    if (a.w_kn) {
        for (int t = tid; t < K * (4 * C0 / 8); t += 64 * GW) {      // 16-byte pieces of a k-row, scattered down a column of Wt
            const int k = t / (4 * C0 / 8), n0 = 8 * (t - k * (4 * C0 / 8));
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.w + (long)k * (4 * C0) + n0);
#pragma unroll
            for (int j = 0; j < 8; ++j) Wt[(n0 + j) * LDK + k] = v[j];
        }
    } else {
        for (int t = tid; t < 4 * C0 * (K / 8); t += 64 * GW) {
            const int n = t / (K / 8), k0 = 8 * (t - n * (K / 8));
            *reinterpret_cast<bf16x8*>(Wt + n * LDK + k0) = *reinterpret_cast<const bf16x8*>(a.w + (long)n * K + k0);
        }
    }
    __syncthreads();
    const int nw = gridDim.x * GW, wv = blockIdx.x * GW + wave;
    const int t_beg = (int)((long)a.tiles * wv / nw), t_end = (int)((long)a.tiles * (wv + 1) / nw);
    if (t_beg >= t_end) return;
    const unsigned gw = (unsigned)a.gw, tpf = (unsigned)(a.gh * a.gw) / 32u;
    const int c8 = (g & 1) * 16 + (g >> 1) * 8;                      // + 32 pp: the 8 consecutive channels this lane stores
    bf16x8 cur[2][NKB], nxt[2][NKB];
    auto issue = [&](int tt, bf16x8 (&dst)[2][NKB]) __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int s = 0; s < NKB; ++s) dst[rb][s] = *reinterpret_cast<const bf16x8*>(a.a + ((long)tt * 32 + 16 * rb + i16) * K + 32 * s + 8 * g);
    };
    issue(t_beg, cur);
    for (int tt = t_beg; tt < t_end; ++tt) {
        const unsigned f = (unsigned)tt / tpf, tl = (unsigned)tt - f * tpf;
        if (tt + 1 < t_end) issue(tt + 1, nxt);
        bf16x8 fb[2][NKB];
#pragma unroll
        for (int s = 0; s < NKB; ++s) {
            if (PRO) {
                float cs[8], ch[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { cs[j] = a.sc[(long)f * K + 32 * s + 8 * g + j]; ch[j] = a.sh[(long)f * K + 32 * s + 8 * g + j]; }
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int j = 0; j < 8; ++j) fb[rb][s][j] = (bf16)gelu_fast(fmaf((float)cur[rb][s][j], cs[j], ch[j]));
            } else { fb[0][s] = cur[0][s]; fb[1][s] = cur[1][s]; }
        }
        long pix0[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const unsigned pl = 32u * tl + 16u * rb + (unsigned)i16, y = pl / gw, x = pl - y * gw;      // per lane: a 16-row block may wrap an image row
            pix0[rb] = ((long)f * (2 * a.gh) + 2 * y) * (2L * a.gw) + 2 * x;
        }
        float ssum[NCB / 2][8], ssq[NCB / 2][8];
#pragma unroll
        for (int pp = 0; pp < NCB / 2; ++pp)
#pragma unroll
            for (int j = 0; j < 8; ++j) ssum[pp][j] = ssq[pp][j] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 acc[NCB][2];
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) { acc[nb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[nb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int s = 0; s < NKB; ++s)
#pragma unroll
                for (int nb = 0; nb < NCB; ++nb) {
                    const bf16x8 aw = *reinterpret_cast<const bf16x8*>(Wt + (C0 * q + 16 * nb + i16) * LDK + 32 * s + 8 * g);
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) acc[nb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw, fb[rb][s], acc[nb][rb], 0, 0, 0);
                }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                bf16* dst = a.map + (pix0[rb] + (long)(q >> 1) * (2L * a.gw) + (q & 1)) * C0 + c8;
#pragma unroll
                for (int pp = 0; pp < NCB / 2; ++pp) {
                    bf16x8 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * pp][rb][r]), __float_as_uint(acc[2 * pp + 1][rb][r]), false, false);
                        o[r] = (bf16)__uint_as_float(sw[0]); o[4 + r] = (bf16)__uint_as_float(sw[1]);
                    }
                    *reinterpret_cast<bf16x8*>(dst + 32 * pp) = o;
                    if (a.part) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const float v = (float)o[j]; ssum[pp][j] += v; ssq[pp][j] = fmaf(v, v, ssq[pp][j]); }      // of the values as stored
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (a.part) {      // this tile = one slice of 128 fine pixels
            float2* o2 = reinterpret_cast<float2*>(a.part) + ((long)f * tpf + tl) * C0 + c8;
#pragma unroll
            for (int pp = 0; pp < NCB / 2; ++pp)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t1 = row16_total(ssum[pp][j]), t2 = row16_total(ssq[pp][j]);
                    const float mu = t1 * (1.0f / 128.0f);
                    if (i16 == 0) o2[32 * pp + j] = make_float2(mu, fmaxf(t2 - t1 * mu, 0.f));
                }
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int s = 0; s < NKB; ++s) cur[rb][s] = nxt[rb][s];
    }
}

// The weight gradients of the same stages: dW[(q, c)][k] = sum_p ff(fine[pixel(p, q)][c]) * fc(coarse[p][k]) -- the fine map gathered,
// the coarse rows plain, either side optionally through GELU(x * sc + sh) (the embed stages transform the fine side, the debed stages the
// coarse side).  A reduction over hundreds of thousands of rows into a 384 x 96 result: 8 waves split the result (wave = 2x2 position
// x half of the fine channels: 3 x 6 accumulator tiles), so every fine element is fetched and transformed by exactly one wave; the 32
// coarse rows of a step are staged once per workgroup.  Both operands need the row index as the MFMA k-slot, i.e. transposed tiles:
// they pass through LDS (wave-private for the fine side) and come back through ds_read_b64_tr_b16.  One barrier per step (the coarse
// tile is double-buffered), global loads one step ahead.  Partials: one [384][96] fp32 slab per workgroup, summed in a fixed order.
constexpr int WW = 8;                                   // waves per workgroup
struct WgradArgs {
    const bf16 *patches, *w0;                           // FREB: fine = W0 . patch, rebuilt per step (see gather_gemm_kernel)
    const bf16 *fine, *coarse;
    const float *fsc, *fsh, *csc, *csh;                 // [F][96] each, or null
    float* slab;                                        // [workgroups][384][96]
    int F, gh, gw, rpf, steps;                          // runs per frame, 32-row steps per run
};
template <bool FPRO, bool CPRO, bool FREB = false>
__global__ void __launch_bounds__(64 * WW, 4) gather_wgrad_kernel(WgradArgs a) {
    constexpr int C = 96, HC = 48, LDF = HC + 8, LDC = C + 8;
    __shared__ __attribute__((aligned(16))) bf16 ctile[2][32 * LDC];
    __shared__ __attribute__((aligned(16))) bf16 ftile_all[WW][32 * LDF];
    __shared__ float2 kf[C], kc[C];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int q = wave >> 1, half = wave & 1;
    bf16* ftile = ftile_all[wave];
    const int f = blockIdx.x / a.rpf, run = blockIdx.x - f * a.rpf;
    if (FPRO) for (int c = tid; c < C; c += 64 * WW) kf[c] = make_float2(a.fsc[(long)f * C + c], a.fsh[(long)f * C + c]);
    if (CPRO) for (int c = tid; c < C; c += 64 * WW) kc[c] = make_float2(a.csc[(long)f * C + c], a.csh[(long)f * C + c]);
    __syncthreads();
    const unsigned gw = (unsigned)a.gw;
    // fine pieces of this wave: row lane / 2, channels 48 half + 8 (2 j + lane % 2), j = 0..2; coarse piece of this thread (tid < 384)
    const int frow = lane >> 1, fc0 = 8 * (lane & 1);
    const int crow = tid / 12, cch = 8 * (tid - 12 * crow);
    const bool cth = tid < 32 * 12;
    bf16x8 fr[3], cr;
    s16x4 pa[2], w0f[3];                                // FREB: patch rows of the two 16-row blocks (lane = row i16, k = 4g..), this lane's rows of W0
    if (FREB) {
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) w0f[cb] = *reinterpret_cast<const s16x4*>(a.w0 + (HC * half + 16 * cb + i16) * 16 + 4 * g);
    }
    auto issue = [&](int st) __attribute__((always_inline)) {
        const unsigned tl = (unsigned)(run * a.steps + st);                      // 32-row tile inside the frame
        if (FREB) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const unsigned pl = 32u * tl + 16u * rb + (unsigned)i16, y = pl / gw, x = pl - y * gw;
                pa[rb] = *reinterpret_cast<const s16x4*>(a.patches + (((long)f * (2 * a.gh) + 2 * y + (q >> 1)) * (2L * a.gw) + 2 * x + (q & 1)) * 16 + 4 * g);
            }
        } else {
            const unsigned pl = 32u * tl + (unsigned)frow, y = pl / gw, x = pl - y * gw;
            const bf16* src = a.fine + (((long)f * (2 * a.gh) + 2 * y + (q >> 1)) * (2L * a.gw) + 2 * x + (q & 1)) * C + HC * half + fc0;
#pragma unroll
            for (int j = 0; j < 3; ++j) fr[j] = *reinterpret_cast<const bf16x8*>(src + 16 * j);
        }
        if (cth) cr = *reinterpret_cast<const bf16x8*>(a.coarse + ((long)f * a.gh * a.gw + 32L * tl + crow) * C + cch);
    };
    f32x4 acc[3][6];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int kb = 0; kb < 6; ++kb) acc[cb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(0);
    for (int st = 0; st < a.steps; ++st) {
        bf16* ct = ctile[st & 1];
        if (FREB) {      // D[c][pixel] = W0 rows x patch rows: lane (pixel i16, channels 16 cb + 4 g .. +3) -> an 8-byte piece of the tile row
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4w;
            int ko = HC * half + 4 * g;              // an offset the compiler cannot see through: the 12 constant pairs stay in LDS, not in 24 registers
            asm volatile("" : "+v"(ko));
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    const f32x4 y4 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0f[cb], pa[rb], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    bf16x4w o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float2 k2 = kf[ko + 16 * cb + e];
                        o[e] = (bf16)gelu_fast(fmaf((float)(bf16)y4[e], k2.x, k2.y));
                    }
                    *reinterpret_cast<bf16x4w*>(ftile + (16 * rb + i16) * LDF + 16 * cb + 4 * g) = o;
                    __builtin_amdgcn_sched_barrier(0);      // one block at a time: six results in flight would not fit beside the 72 accumulators
                }
        } else
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bf16x8 v = fr[j];
            if (FPRO) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float2 k2 = kf[HC * half + fc0 + 16 * j + e]; v[e] = (bf16)gelu_fast(fmaf((float)v[e], k2.x, k2.y)); }
            }
            *reinterpret_cast<bf16x8*>(ftile + frow * LDF + fc0 + 16 * j) = v;
        }
        if (cth) {
            bf16x8 v = cr;
            if (CPRO) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float2 k2 = kc[cch + e]; v[e] = (bf16)gelu_fast(fmaf((float)v[e], k2.x, k2.y)); }
            }
            *reinterpret_cast<bf16x8*>(ct + crow * LDC + cch) = v;
        }
        __syncthreads();
        if (st + 1 < a.steps) issue(st + 1);
        // k-slot (g, jj) := row 4g + jj (jj < 4) / row 16 + 4g + jj - 4: the same order on both operands
        bf16x8 fa[3];
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) fa[cb] = cat8(tr4g(ftile, LDF, 4 * g, 16 * cb, lane), tr4g(ftile, LDF, 16 + 4 * g, 16 * cb, lane));
#pragma unroll
        for (int kb = 0; kb < 6; ++kb) {
            const bf16x8 fbk = cat8(tr4g(ct, LDC, 4 * g, 16 * kb, lane), tr4g(ct, LDC, 16 + 4 * g, 16 * kb, lane));
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) acc[cb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cb], fbk, acc[cb][kb], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // this wave's tile reads before its next tile writes
        __builtin_amdgcn_wave_barrier();
    }
    // acc[cb][kb]: lane (k = 16 kb + i16) holds fine channels 16 cb + 4 g + r of this wave's half
    float* dst = a.slab + (long)blockIdx.x * (4 * C * C) + (long)(q * C + HC * half) * C;
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int kb = 0; kb < 6; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(16 * cb + 4 * g + r) * C + 16 * kb + i16] = acc[cb][kb][r];
}
// out = sum over slabs, in slab order; transposed: out[k][(q, c)] (ldo = 384), else out[(q, c)][k] (ldo = 96)
__global__ void __launch_bounds__(256) gather_wgrad_reduce_kernel(const float* __restrict__ slab, int rows, float* __restrict__ out, int transposed) {
    // 64 elements x 4 row lanes per block: each lane sums every fourth slab (16 loads in flight), the four lanes are added in a fixed order
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6, n = 4 * 96 * 96;
    const int e = blockIdx.x * 64 + l;
    float v = 0.f;
    if (e < n) {
        int r = q;
        for (; r + 60 < rows; r += 64) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = slab[(long)(r + 4 * u) * n + e];
#pragma unroll
            for (int u = 0; u < 16; ++u) v += t[u];
        }
        for (; r < rows; r += 4) v += slab[(long)r * n + e];
    }
    red[q][l] = v;
    __syncthreads();
    if (q != 0 || e >= n) return;
    v = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    const int qc = e / 96, k = e - 96 * qc;
    out[transposed ? (long)k * 384 + qc : (long)e] = v;
}

int gg_cus() {
    static const int n = []() {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 8) return 256;
        return c;
    }();
    return n;
}

}  // namespace

static int gather_gemm_launch(int dtype, const void* map, const void* patches, const void* w0c, const void* w, int w_kn, const float* sc, const float* sh,
                              void* out, int F, int gh, int gw, int C0, int N, bf_stream_t stream) {
    if (dtype != BF_DTYPE_BF16 || C0 != 96 || N != 96) return 1;
    if (F <= 0 || gh <= 0 || gw <= 0 || ((long)gh * gw) % 32) return 1;
    static const bool off = bf_knob("BF_GATHER_GEMM", 1) == 0;
    if (off) return 1;
    const bool reb = map == nullptr;
    BF_REQUIRE((map || (patches && w0c && sc)) && w && out && (!sc == !sh), "bf_gather_gemm: bad arguments");
    BF_REQUIRE((((uintptr_t)map | (uintptr_t)patches | (uintptr_t)w0c | (uintptr_t)w | (uintptr_t)out) & 15) == 0, "bf_gather_gemm: operands must be 16-byte aligned");
    const long tiles = (long)F * gh * gw / 32;
    BF_REQUIRE(tiles < (1L << 30), "bf_gather_gemm: too many rows");
    hipStream_t st = (hipStream_t)stream;
    const int cus = gg_cus();
    // as few workgroups as give every wave the same number of tiles as a full grid would (each loads the whole weight first)
    const long rounds = (tiles + 2L * cus * GW - 1) / (2L * cus * GW), nwaves = (tiles + rounds - 1) / rounds;
    const int grid = (int)((nwaves + GW - 1) / GW);
    GatherArgs a{(const bf16*)patches, (const bf16*)w0c, (const bf16*)map, (const bf16*)w, (bf16*)out, sc, sh, w_kn, F, gh, gw, (int)tiles};
    constexpr int lds = gather_lds_bytes<6, 6>();
    const double rows = (double)F * gh * gw;
    BfProfScope prof(st, reb ? "gather_gemm<gelu,rebuilt>" : sc ? "gather_gemm<gelu>" : "gather_gemm<plain>", 2.0 * rows * 4 * C0 * N,
                     rows * ((reb ? 4.0 * 16 : 4.0 * C0) + N) * 2.0);
#define BF_GG_GO(PRO, REB)                                                                                                                 \
    do {                                                                                                                                  \
        static BfPerDeviceOnce attr_once; bool& attr_done = attr_once.flag();                                                                                                    \
        if (!attr_done) {                                                                                                                 \
            hipError_t e_ = hipFuncSetAttribute((const void*)gather_gemm_kernel<6, 6, PRO, REB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);                                                                 \
            attr_done = true;                                                                                                             \
        }                                                                                                                                 \
        hipLaunchKernelGGL((gather_gemm_kernel<6, 6, PRO, REB>), dim3(grid), dim3(64 * GW), lds, st, a);                                  \
    } while (0)
    if (reb) BF_GG_GO(true, true); else if (sc) BF_GG_GO(true, false); else BF_GG_GO(false, false);
#undef BF_GG_GO
    BF_CHECK_LAUNCH();
    return 0;
}
// 0 = done, 1 = shape not covered (nothing launched)
extern "C" int bf_gather_gemm(int dtype, const void* map, const void* w, int w_kn, const float* sc, const float* sh, void* out, int F, int gh,
                              int gw, int C0, int N, bf_stream_t stream) {
    BF_REQUIRE(map, "bf_gather_gemm: null map");
    return gather_gemm_launch(dtype, map, nullptr, nullptr, w, w_kn, sc, sh, out, F, gh, gw, C0, N, stream);
}
// the same with the fine map given as its factors: map[pixel][c] = sum_k patches[pixel][k] * w0c[c][k] (Kp = 16: the first HMLPEmbed stage),
// rebuilt per tile and rounded to bf16 like a stored map -- the 2x2 stage behind bf_embed_first without reading (or needing) its output
extern "C" int bf_gather_gemm_rebuilt(int dtype, const void* patches, const void* w0c, const void* w, int w_kn, const float* sc, const float* sh, void* out,
                                      int F, int gh, int gw, int C0, int N, bf_stream_t stream) {
    BF_REQUIRE(patches && w0c && sc && sh, "bf_gather_gemm_rebuilt: null pointer");
    return gather_gemm_launch(dtype, nullptr, patches, w0c, w, w_kn, sc, sh, out, F, gh, gw, C0, N, stream);
}

// 0 = done, 1 = shape not covered (nothing launched).  stat_part (optional): the slice partials, 128-row slices, at ws + 2*frames*C0 of the
// workspace bf_in_stats_merge_slices(..., rows = 128, ws) takes
extern "C" int bf_scatter_gemm(int dtype, const void* a, const void* w, int w_kn, const float* sc, const float* sh, void* map, float* stat_part, int F,
                               int gh, int gw, int K, int C0, bf_stream_t stream) {
    if (dtype != BF_DTYPE_BF16 || C0 != 96 || K != 96) return 1;
    if (F <= 0 || gh <= 0 || gw <= 0 || ((long)gh * gw) % 32) return 1;
    static const bool off = bf_knob("BF_SCATTER_GEMM", 1) == 0;
    if (off) return 1;
    BF_REQUIRE(a && w && map && (!sc == !sh), "bf_scatter_gemm: bad arguments");
    BF_REQUIRE((((uintptr_t)a | (uintptr_t)w | (uintptr_t)map) & 15) == 0, "bf_scatter_gemm: operands must be 16-byte aligned");
    const long tiles = (long)F * gh * gw / 32;
    BF_REQUIRE(tiles < (1L << 30), "bf_scatter_gemm: too many rows");
    hipStream_t st = (hipStream_t)stream;
    const int cus = gg_cus();
    // as few workgroups as give every wave the same number of tiles as a full grid would (each loads the whole weight first)
    const long rounds = (tiles + 2L * cus * GW - 1) / (2L * cus * GW), nwaves = (tiles + rounds - 1) / rounds;
    const int grid = (int)((nwaves + GW - 1) / GW);
    ScatterArgs sa{(const bf16*)a, (const bf16*)w, (bf16*)map, sc, sh, stat_part, w_kn, F, gh, gw, (int)tiles};
    constexpr int lds = scatter_lds_bytes<3, 6>();
    const double rows = (double)F * gh * gw;
    BfProfScope prof(st, sc ? "scatter_gemm<gelu>" : "scatter_gemm<plain>", 2.0 * rows * 4 * C0 * K, rows * (4.0 * C0 + K) * 2.0);
#define BF_SG_GO(PRO)                                                                                                                      \
    do {                                                                                                                                  \
        static BfPerDeviceOnce attr_once; bool& attr_done = attr_once.flag();                                                                                                    \
        if (!attr_done) {                                                                                                                 \
            hipError_t e_ = hipFuncSetAttribute((const void*)scatter_gemm_kernel<3, 6, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);                                                                 \
            attr_done = true;                                                                                                             \
        }                                                                                                                                 \
        hipLaunchKernelGGL((scatter_gemm_kernel<3, 6, PRO>), dim3(grid), dim3(64 * GW), lds, st, sa);                                     \
    } while (0)
    if (sc) BF_SG_GO(true); else BF_SG_GO(false);
#undef BF_SG_GO
    BF_CHECK_LAUNCH();
    return 0;
}

// runs per frame: a divisor of the 32-row tiles per frame, about two workgroups per CU in all, as many as the workspace holds slabs for
static int wgrad_runs(int F, int tpf, int64_t ws_floats) {
    int rpf = 0;
    for (int r = 1; r <= tpf; ++r)
        if (tpf % r == 0 && (long)F * r <= 512 && (int64_t)F * r * (4 * 96 * 96) <= ws_floats) rpf = r;
    return rpf;
}
// the workspace of the preferred launch; any workspace of at least frames * 36864 floats is accepted (fewer, longer runs)
extern "C" int64_t bf_gather_wgrad_ws_floats(int F, int gh, int gw) {
    if (F <= 0 || gh <= 0 || gw <= 0 || ((long)gh * gw) % 32) return 0;
    return (int64_t)F * wgrad_runs(F, (int)((long)gh * gw / 32), INT64_MAX) * (4 * 96 * 96);
}
// dW[(q, c)][k] (transposed = 0, [384][96]) or dW[k][(q, c)] (transposed = 1, [96][384]) is WRITTEN.  0 = done, 1 = shape not covered.
static int gather_wgrad_launch(int dtype, const void* fine, const void* patches, const void* w0c, const void* coarse, const float* fsc, const float* fsh,
                               const float* csc, const float* csh, float* out, int transposed, int F, int gh, int gw, int C0, int Kc, float* ws,
                               int64_t ws_floats, bf_stream_t stream);
extern "C" int bf_gather_wgrad(int dtype, const void* fine, const void* coarse, const float* fsc, const float* fsh, const float* csc, const float* csh,
                               float* out, int transposed, int F, int gh, int gw, int C0, int Kc, float* ws, int64_t ws_floats, bf_stream_t stream) {
    BF_REQUIRE(fine, "bf_gather_wgrad: null pointer");
    return gather_wgrad_launch(dtype, fine, nullptr, nullptr, coarse, fsc, fsh, csc, csh, out, transposed, F, gh, gw, C0, Kc, ws, ws_floats, stream);
}
// the fine side given as its factors (fine[pixel][c] = sum_k patches[pixel][k] * w0c[c][k], Kp = 16) and transformed by GELU(x * fsc + fsh)
extern "C" int bf_gather_wgrad_rebuilt(int dtype, const void* patches, const void* w0c, const void* coarse, const float* fsc, const float* fsh, float* out,
                                       int transposed, int F, int gh, int gw, int C0, int Kc, float* ws, int64_t ws_floats, bf_stream_t stream) {
    BF_REQUIRE(patches && w0c && fsc && fsh, "bf_gather_wgrad_rebuilt: null pointer");
    return gather_wgrad_launch(dtype, nullptr, patches, w0c, coarse, fsc, fsh, nullptr, nullptr, out, transposed, F, gh, gw, C0, Kc, ws, ws_floats, stream);
}
static int gather_wgrad_launch(int dtype, const void* fine, const void* patches, const void* w0c, const void* coarse, const float* fsc, const float* fsh,
                               const float* csc, const float* csh, float* out, int transposed, int F, int gh, int gw, int C0, int Kc, float* ws,
                               int64_t ws_floats, bf_stream_t stream) {
    if (dtype != BF_DTYPE_BF16 || C0 != 96 || Kc != 96) return 1;
    if (F <= 0 || gh <= 0 || gw <= 0 || ((long)gh * gw) % 32) return 1;
    if ((fsc && csc) || (!fsc != !fsh) || (!csc != !csh)) return 1;
    static const bool off = bf_knob("BF_GATHER_WGRAD", 1) == 0;
    if (off) return 1;
    BF_REQUIRE((fine || (patches && w0c)) && coarse && out && ws, "bf_gather_wgrad: null pointer");
    BF_REQUIRE((((uintptr_t)fine | (uintptr_t)patches | (uintptr_t)w0c | (uintptr_t)coarse) & 15) == 0, "bf_gather_wgrad: operands must be 16-byte aligned");
    const int tpf = (int)((long)gh * gw / 32);
    const int rpf = wgrad_runs(F, tpf, ws_floats);
    if (rpf < 1) return 1;
    const long nwg = (long)F * rpf;
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a{(const bf16*)patches, (const bf16*)w0c, (const bf16*)fine, (const bf16*)coarse, fsc, fsh, csc, csh, ws, F, gh, gw, rpf, tpf / rpf};
    {
        const double rows = (double)F * gh * gw;
        BfProfScope prof(st, !fine ? "gather_wgrad<fine gelu,rebuilt>" : fsc ? "gather_wgrad<fine gelu>" : csc ? "gather_wgrad<coarse gelu>" : "gather_wgrad<plain>",
                         2.0 * rows * 384 * 96, rows * ((fine ? 384 : 64) + 96) * 2.0);
        if (!fine) hipLaunchKernelGGL((gather_wgrad_kernel<true, false, true>), dim3((unsigned)nwg), dim3(64 * WW), 0, st, a);
        else if (fsc) hipLaunchKernelGGL((gather_wgrad_kernel<true, false>), dim3((unsigned)nwg), dim3(64 * WW), 0, st, a);
        else if (csc) hipLaunchKernelGGL((gather_wgrad_kernel<false, true>), dim3((unsigned)nwg), dim3(64 * WW), 0, st, a);
        else hipLaunchKernelGGL((gather_wgrad_kernel<false, false>), dim3((unsigned)nwg), dim3(64 * WW), 0, st, a);
        BF_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(gather_wgrad_reduce_kernel, dim3(bf_cdiv(4 * 96 * 96, 64)), dim3(256), 0, st, (const float*)ws, (int)nwg, out, transposed);
    BF_CHECK_LAUNCH();
    return 0;
}
