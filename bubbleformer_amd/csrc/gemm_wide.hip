// Wide-tile bf16 MFMA GEMM for gfx950: 128 (or 64) x 384 output tile per 8-wave workgroup, one workgroup per CU.
//
// Why this shape.  On MI355X a CU streams ~70 GB/s from its XCD's L2 and ~25 GB/s from HBM, against ~9.8 TFLOP/s of bf16
// MFMA: a 128 x 128 x K tile moves 64 FLOP per operand byte and is bandwidth-bound long before the matrix pipe is busy.
// A 128 x 384 tile covers ALL output columns of the E = 384 projections (so the activation operand is read from HBM
// exactly once), a third / a quarter of the 3E / 4E ones, and raises the intensity to 96 FLOP/B; the weight panel it
// re-reads is L2-resident.  For the token-reduction (dW) form the same tile makes the 384-wide activation the shared
// operand.  Operands are staged global -> registers (two K-tiles in flight) -> prologue -> LDS, LDS is double-buffered so a
// K-step has ONE barrier: while a wave multiplies tile t, its SIMD partner is still writing tile t+1.
//
//   waves 2 (M) x 4 (N); wave tile (16 TM) x 96 = TM x 6 MFMA 16x16x32 tiles; BK = 64.
// Same operand / prologue / epilogue contract as gemm.hip (plain rows only: no k2s2 gather or scatter here).
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {
using namespace bfgemm;

constexpr int NTW = 512, BKW = 64, BNW = 384, TNW = 6;

// one operand tile [TR][TC] bf16 handled by 512 threads in 16-byte chunks
template <int TR, int TC, bool XC, bool PRO>
struct WStager {
    static constexpr int NCH = TR * TC / 8 / NTW;
    static constexpr int LDT = TC;
    static constexpr int CPR = (TC <= 128 ? TC : 128) / 8;       // chunks per (sub-)row
    static constexpr int PER = TC <= 128 ? NCH : NCH / (TC / 128);   // chunks per thread per 128-column sub-tile
    static constexpr int RSTEP = NTW / CPR;
    bf16x8 data[2][NCH];      // two K-tiles in flight
    unsigned valid[2];
    int k0[2];
    int voff[NCH];            // element offset of chunk i relative to the K-tile origin (host guarantees 31 bits)
    int loff[NCH];            // LDS element offset
    int fidx[(!XC && PRO) ? NCH : 1];
    int rb, cb;               // tile-local row / column of chunk 0
    unsigned ok;              // bit i: fixed coordinate (KC: row, XC: column) of chunk i in range

    static __device__ __forceinline__ constexpr int rstep(int i) { return RSTEP * (i % PER); }
    static __device__ __forceinline__ constexpr int cstep(int i) { return 128 * (i / PER); }

    __device__ __forceinline__ void setup(const OpDev& op, int outer0, int nouter, int tid) {
        ok = 0u;
        rb = tid / CPR; cb = (tid % CPR) * 8;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = rb + rstep(i), c = cb + cstep(i);
            loff[i] = lds_off<bf16, XC, LDT>(r, c);
            if constexpr (!XC) {
                const int row = outer0 + r;
                const bool v = row < nouter;
                ok |= (v ? 1u : 0u) << i;
                voff[i] = v ? row * (int)op.ld + c : 0;
                if constexpr (PRO) fidx[i] = row / op.rpf;
            } else {
                const int col = outer0 + c;
                const bool v = col < nouter;
                ok |= (v ? 1u : 0u) << i;
                voff[i] = r * (int)op.ld + (v ? col : 0);
            }
        }
    }
    template <int S>
    __device__ __forceinline__ void issue(const OpDev& op, int kt, int kend) {
        k0[S] = kt; valid[S] = 0u;
        const bf16* tile = reinterpret_cast<const bf16*>(op.p) + (XC ? (long)kt * op.ld : (long)kt);   // wave-uniform
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const bool in = XC ? (kt + rb + rstep(i) < kend) : (kt + cb < kend);
            if (in && ((ok >> i) & 1u)) { valid[S] |= 1u << i; data[S][i] = *reinterpret_cast<const bf16x8*>(tile + voff[i]); }
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) data[S][i][j] = (bf16)0.f;
            }
        }
    }
    template <int S>
    __device__ __forceinline__ void commit(const OpDev& op, const ProTab& tab, bf16* lds, int outer0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            bf16x8 v = data[S][i];
            if (PRO && op.pro != BF_PRO_NONE && ((valid[S] >> i) & 1u)) {
                float sc[8], sh[8];
                if (op.pro != BF_PRO_GELU) {
                    int f, ch;
                    if constexpr (!XC) { f = fidx[PRO ? i : 0]; ch = k0[S] + cb; }
                    else { f = (int)((unsigned)(k0[S] + rb + rstep(i)) / (unsigned)op.rpf); ch = outer0 + cb + cstep(i); }
                    if (ch >= op.nch) ch %= op.nch;
                    const long o = tab.ok ? (long)(f - tab.f_lo) * tab.cw + (ch - tab.c_lo) : (long)f * op.nch + ch;
#pragma unroll
                    for (int j = 0; j < 8; j += 4) {
                        const float4 a4 = *reinterpret_cast<const float4*>(tab.sc + o + j);
                        const float4 b4 = tab.sh ? *reinterpret_cast<const float4*>(tab.sh + o + j) : float4{0.f, 0.f, 0.f, 0.f};
                        sc[j] = a4.x; sc[j + 1] = a4.y; sc[j + 2] = a4.z; sc[j + 3] = a4.w;
                        sh[j] = b4.x; sh[j + 1] = b4.y; sh[j + 2] = b4.z; sh[j + 3] = b4.w;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = (float)v[j];
                    if (op.pro != BF_PRO_GELU) x = x * sc[j] + sh[j];
                    if (op.pro != BF_PRO_AFFINE) x = gelu_f(x);
                    v[j] = (bf16)x;
                }
            }
            *reinterpret_cast<bf16x8*>(lds + loff[i]) = v;
        }
    }
};

template <bool AXC, bool BXC, bool APRO, bool BPRO, int TM>
__global__ void __launch_bounds__(NTW) gemm_wide_kernel(int M, int N, int K, OpDev A, OpDev B, EpiDev E, int kper, int mt, int nt) {
    constexpr int BM = 32 * TM;
    constexpr int LDA = AXC ? BM : BKW, LDB = BXC ? BNW : BKW;
    constexpr int A_ELEMS = BM * BKW, B_ELEMS = BNW * BKW;
    constexpr bool ANYPRO = APRO || BPRO;
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    bf16* lA[2] = {reinterpret_cast<bf16*>(smem_w), reinterpret_cast<bf16*>(smem_w) + A_ELEMS + B_ELEMS};
    bf16* lB[2] = {lA[0] + A_ELEMS, lA[1] + A_ELEMS};
    float* ltab = reinterpret_cast<float*>(reinterpret_cast<bf16*>(smem_w) + 2 * (A_ELEMS + B_ELEMS));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int seq = xcd_remap(blockIdx.x, gridDim.x);
    const int zt = seq / (mt * nt), rem = seq - zt * (mt * nt);
    const int m0 = (rem / nt) * BM;
    const int n0 = (rem % nt) * BNW;
    const int kbeg = zt * kper;
    const int kend = min(K, kbeg + kper);

    WStager<AXC ? BKW : BM, AXC ? BM : BKW, AXC, APRO> sa;
    WStager<BXC ? BKW : BNW, BXC ? BNW : BKW, BXC, BPRO> sb;
    sa.setup(A, m0, M, tid);
    sb.setup(B, n0, N, tid);

    f32x4 acc[TM][TNW];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (kbeg < kend) { sa.template issue<0>(A, kbeg, kend); sb.template issue<0>(B, kbeg, kend); }
    if (kbeg + BKW < kend) { sa.template issue<1>(A, kbeg + BKW, kend); sb.template issue<1>(B, kbeg + BKW, kend); }

    ProTab ta{}, tb{};
    if constexpr (APRO) {
        if constexpr (AXC) ta = stage_table(A, kbeg, kend - 1, m0, BM, ltab, ltab + TAB, tid, NTW);
        else ta = stage_table(A, m0, min(M, m0 + BM) - 1, 0, A.nch, ltab, ltab + TAB, tid, NTW);
    }
    if constexpr (BPRO) {
        if constexpr (BXC) tb = stage_table(B, kbeg, kend - 1, n0, BNW, ltab, ltab + TAB, tid, NTW);
        else tb = stage_table(B, n0, min(N, n0 + BNW) - 1, 0, B.nch, ltab, ltab + TAB, tid, NTW);
    }
    if constexpr (ANYPRO) __syncthreads();

    auto compute = [&](const bf16* a, const bf16* b) {
#pragma unroll
        for (int kk = 0; kk < BKW; kk += 32) {
            bf16x8 fa[TM], fb[TNW];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = frag_bf16<AXC, LDA>(a, wm * (16 * TM) + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < TNW; ++j) fb[j] = frag_bf16<BXC, LDB>(b, wn * 96 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };
    // tile t lives in LDS buffer t & 1 and came through register stage t & 1
    int k0 = kbeg;
    if (k0 < kend) {
        sa.template commit<0>(A, ta, lA[0], m0);
        sb.template commit<0>(B, tb, lB[0], n0);
        if (k0 + 2 * BKW < kend) { sa.template issue<0>(A, k0 + 2 * BKW, kend); sb.template issue<0>(B, k0 + 2 * BKW, kend); }
    }
    __syncthreads();
#define BF_WIDE_STEP(CUR, NXT)                                                                                       \
    if (k0 < kend) {                                                                                                 \
        if (k0 + BKW < kend) {                                                                                       \
            sa.template commit<NXT>(A, ta, lA[NXT], m0);                                                       \
            sb.template commit<NXT>(B, tb, lB[NXT], n0);                                                       \
            if (k0 + 3 * BKW < kend) { sa.template issue<NXT>(A, k0 + 3 * BKW, kend); sb.template issue<NXT>(B, k0 + 3 * BKW, kend); } \
        }                                                                                                            \
        compute(lA[CUR], lB[CUR]);                                                                                   \
        __syncthreads();                                                                                             \
        k0 += BKW;                                                                                                   \
    }
    while (k0 < kend) {
        BF_WIDE_STEP(0, 1)
        BF_WIDE_STEP(1, 0)
    }
#undef BF_WIDE_STEP

    // ------------------------------------------------------------------ epilogue (row-major through LDS, same contract as gemm.hip)
    epilogue_rows<bf16, TM, TNW, 2, 4, AXC>(acc, E, M, N, m0, n0, reinterpret_cast<float*>(smem_w), tid);
}

template <typename Kern>
int launch_wide(Kern kern, int tm, int M, int N, int K, const OpDev& a, const OpDev& b, const EpiDev& e, int kper, int splitk, hipStream_t st) {
    const int bm = 32 * tm;
    const int mt = bf_cdiv(M, bm), nt = bf_cdiv(N, BNW);
    const size_t shm = (size_t)2 * (bm * BKW + BNW * BKW) * sizeof(bf16) + 2 * TAB * sizeof(float);
    static thread_local const void* configured[16];
    static thread_local int nconf = 0;
    bool seen = false;
    for (int i = 0; i < nconf; ++i) seen |= configured[i] == (const void*)kern;
    if (!seen) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (err != hipSuccess) return bf_fail(err, __FILE__, __LINE__);
        if (nconf < 16) configured[nconf++] = (const void*)kern;
    }
    dim3 grid((unsigned)((long)mt * nt * splitk));
    hipLaunchKernelGGL(kern, grid, dim3(NTW), shm, st, M, N, K, a, b, e, kper, mt, nt);
    BF_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Called by bf_gemm (gemm.hip) for bf16 problems this kernel covers; returns 1 if it declined.
int bf_gemm_wide_try(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, int splitk, hipStream_t st) {
    const bool ax = A->layout == BF_LAY_XC, bx = B->layout == BF_LAY_XC;
    if (ax && !bx) return 1;
    if (E->colsum) return 1;
    if (A->gw > 0 || B->gw > 0 || A->seglen > 0 || B->seglen > 0 || E->gw > 0 || E->seglen > 0) return 1;
    if (N % 8 || N < 192) return 1;                        // narrow outputs waste the 384-wide tile
    if ((long)(ax ? K : M) * A->ld >= (1L << 31) || (long)(bx ? K : N) * B->ld >= (1L << 31)) return 1;   // 32-bit tile offsets
    const bool ap = A->pro != BF_PRO_NONE, bp = B->pro != BF_PRO_NONE;
    if ((ap && bp) || (ax && ap) || (!ax && bp)) return 1;
    // measured on MI355X (tools/gemm_bench.py): the wide tile wins where the per-(frame, channel) prologue would otherwise be
    // redone by every 128-column tile (3E / 4E projections fed by an InstanceNorm); elsewhere the 4-wave kernel's two
    // co-resident workgroups per CU overlap better.  BF_GEMM_WIDE_ALL=1 lifts the restriction for experiments.
    static const bool all = []() { const char* v = getenv("BF_GEMM_WIDE_ALL"); return v && atoi(v) != 0; }();
    if (!all && !(!ax && !bx && ap && N >= 768)) return 1;
    OpDev a, b;
    auto cv = [](const bf_operand* o, OpDev& d) {
        d.p = o->p; d.ld = o->ld; d.layout = o->layout; d.seglen = 0; d.segstride = 0; d.gw = d.gh = d.gc = 0; d.pro = o->pro; d.sc = o->sc;
        d.sh = o->sh; d.rpf = o->rows_per_frame > 0 ? o->rows_per_frame : 1; d.nch = o->nch > 0 ? o->nch : 1;
    };
    cv(A, a); cv(B, b);
    EpiDev e;
    e.bias = E->bias; e.colscale = E->colscale; e.colshift = E->colshift; e.aux_mode = E->aux_mode; e.aux = E->aux; e.ld_aux = E->ld_aux;
    e.out_mode = E->out_mode; e.c = E->c; e.ldc = E->ldc; e.seglen = 0; e.segstride = 0; e.gw = e.gh = e.gc = 0; e.gelu_out = E->gelu_out; e.colsum = nullptr; e.rowscale = E->rowscale; e.rpg = E->rows_per_group > 0 ? E->rows_per_group : 1;
    if (splitk < 1) splitk = 1;
    const int ktiles = bf_cdiv(K, BKW);
    if (splitk > ktiles) splitk = ktiles;
    const int kper = bf_cdiv(ktiles, splitk) * BKW;
    splitk = bf_cdiv(K, kper);
    const int nt = bf_cdiv(N, BNW);
    const int tm = ((long)bf_cdiv(M, 128) * nt * splitk >= 224 || M <= 64) ? 4 : 2;   // keep ~one workgroup per CU
    const double es = 2.0;
    static thread_local char pname[96];
    snprintf(pname, sizeof(pname), "gemm_wide_kernel<bf16,%s,%s,pro%s,tm%d>", ax ? "xc" : "kc", bx ? "xc" : "kc", ap ? "A" : bp ? "B" : "0", tm);
    BfProfScope prof(st, pname, 2.0 * M * N * K,
                     (double)M * K * es + (double)N * K * es + (double)M * N * (E->out_mode == BF_OUT_STORE ? es : 4.0) +
                         (E->aux_mode != BF_AUX_NONE ? (double)M * N * es : 0.0));
#define GO(AX, BX, AP, BP)                                                                                                      \
    return tm == 4 ? launch_wide(gemm_wide_kernel<AX, BX, AP, BP, 4>, 4, M, N, K, a, b, e, kper, splitk, st)                   \
                   : launch_wide(gemm_wide_kernel<AX, BX, AP, BP, 2>, 2, M, N, K, a, b, e, kper, splitk, st)
    if (!ax && !bx) { if (ap) GO(false, false, true, false); else GO(false, false, false, false); }
    if (!ax && bx) { if (ap) GO(false, true, true, false); else GO(false, true, false, false); }
    if (bp) GO(true, true, false, true); else GO(true, true, false, false);
#undef GO
}
