// Tiled MFMA GEMM for gfx950 with fused operand prologues and epilogues.
//
//   C[M,N] (+)= epi( sum_k pro(A)[m,k] * pro(B)[n,k] )
//
// One kernel template covers every dense contraction of the FiLMAViT path:
//   forward  x @ W^T            A = activations  [M][K] (KC)   B = W  [N][K] (KC)
//   dA       dC @ W             A = dC           [M][N] (KC)   B = W  [N][K] read as [k=n][outer=k] (XC)
//   dW       dC^T @ x           A = dC           [m][N] (XC)   B = x  [m][K] (XC)      (split-K, fp32 atomics)
//   k2s2 conv / conv-transpose  patch gather on an operand, patch scatter on the store
// Operands are staged global -> registers (prologue: InstanceNorm affine and/or GELU, fp32) -> LDS;
// the next K-tile's loads are in flight while the current tile is multiplied (issue-early / write-late).
// 128x128 block tile; bf16: 8 waves (2x4), 64x32 per wave as 4x2 MFMA 16x16 tiles, two workgroups per CU, two LDS tile buffers in
// the prologue-free variants; f32 (and BF_GEMM_WAVES=4): 4 waves (2x2), 64x64 per wave.
//   bf16: v_mfma_f32_16x16x32_bf16, BK = 64;   f32: v_mfma_f32_16x16x4_f32 (exact fp32), BK = 32.
// The MFMA is issued "swapped" (B fragment as the first operand) so each lane ends up with 4
// consecutive output COLUMNS of one row: 8-/16-byte epilogue stores and vector bias loads.
// XC (outer-contiguous) bf16 tiles are read with the gfx950 transposing LDS read (ds_read_b64_tr_b16).
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int BN = 128;      // block tile columns; rows BM = 32 * TM; waves 2 x (NT / 128): NT = 256 -> 2 x 2 waves of TM x 4 MFMA tiles,
                             // NT = 512 -> 2 x 4 waves of TM x 2 tiles (half the accumulators per wave, twice the waves per CU)
#ifndef BF_GEMM_DEFAULT_WAVES
#define BF_GEMM_DEFAULT_WAVES 4
#endif
#ifndef BF_GEMM_DEFAULT_STAGES
#define BF_GEMM_DEFAULT_STAGES 1
#endif

template <typename T> struct GemmCfg;
// bf16 tiles are unpadded and XOR-swizzled (conflict-free ds_read_b128 / ds_read_b64_tr_b16, see lds_off); f32 tiles are padded
template <> struct GemmCfg<bf16> { static constexpr int BK = 64, KSTEP = 32, PADK = 0, PADX = 0; };
template <> struct GemmCfg<float> { static constexpr int BK = 32, KSTEP = 4, PADK = 4, PADX = 4; };

using namespace bfgemm;

// Stage one operand tile.  TR x TC elements, row-major in LDS with leading dim LDT.
// KC: rows = outer index (BM/BN, fixed for the block), cols = k (advance per K-tile).
// XC: rows = k (advance per K-tile), cols = outer index (fixed for the block).
// The fixed half of every chunk address is computed once (setup); issue() adds the moving half and puts the 16-byte
// loads in flight; commit() (later) applies the prologue and writes LDS.  32-bit index math throughout.
template <typename T, int TR, int TC, int LDT, bool XC, int NT>
struct Stager {
    static constexpr int CH = Chunk<T>::N;
    static constexpr int CPR = TC / CH;
    static constexpr int NCH = TR * TC / CH / NT;   // chunks per thread; chunk i sits at tile row r0 + (NT / CPR) * i, tile col cc
    static constexpr int RSTEP = NT / CPR;
    Chunk<T> data[NCH];
    unsigned valid;
    int k0;
};
template <typename T, int TR, int TC, int LDT, bool XC, int NT>
struct StagerFixed {
    static constexpr int CH = Chunk<T>::N;
    static constexpr int CPR = TC / CH;
    static constexpr int NCH = TR * TC / CH / NT;
    static constexpr int RSTEP = NT / CPR;
    // Everything that does not change from K-tile to K-tile is computed once here, so that the K loop spends its
    // vector-issue slots on MFMA and LDS traffic: per chunk one 64-bit add and one global load.
    long voff[NCH];        // plain rows: element offset of chunk i relative to the K-tile origin
    long gfix[XC ? 1 : NCH];  // gathered rows (k2s2 patches): KC row bases / XC column offset
    int fidx[XC ? 1 : NCH];   // KC: frame of each chunk row (prologue)
    int loff[NCH];         // LDS element offset of chunk i
    unsigned ok;           // KC: bit i = row in range; XC: bit 0 = column in range
    int r0, cc;
    bool gather;

    __device__ __forceinline__ void setup(const OpDev& op, int outer0, int nouter, int tid, bool need_frames) {
        r0 = tid / CPR; cc = (tid % CPR) * CH;
        gather = op.gw > 0 || op.seglen > 0;
        ok = 0u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) loff[i] = lds_off<T, XC, LDT>(r0 + RSTEP * i, cc);
        if constexpr (!XC) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int row = outer0 + r0 + RSTEP * i;
                const bool v = row < nouter;
                ok |= (v ? 1u : 0u) << i;
                gfix[i] = v ? row_base(row, op.ld, op.gw, op.gh, op.gc) : 0;
                voff[i] = gfix[i] + cc;
                fidx[i] = need_frames ? row / op.rpf : 0;
            }
        } else {
            const int col = outer0 + cc;
            ok = col < nouter ? 1u : 0u;
            gfix[0] = col_off(col, op.seglen, op.segstride);
            fidx[0] = 0;
#pragma unroll
            for (int i = 0; i < NCH; ++i) voff[i] = (long)(r0 + RSTEP * i) * op.ld + gfix[0];
        }
    }
    __device__ __forceinline__ void issue(Stager<T, TR, TC, LDT, XC, NT>& s, const OpDev& op, int k0, int kend) const {
        s.k0 = k0; s.valid = 0u;
        if constexpr (!XC) {
            const bool cv = k0 + cc < kend;
            if (!gather) {
                const T* tile = reinterpret_cast<const T*>(op.p) + k0;                  // wave-uniform
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    if (cv && ((ok >> i) & 1u)) { s.valid |= 1u << i; s.data[i].load(tile + voff[i]); }
                    else s.data[i].zero();
                }
            } else {
                const T* base = reinterpret_cast<const T*>(op.p) + col_off(k0 + cc, op.seglen, op.segstride);
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    if (cv && ((ok >> i) & 1u)) { s.valid |= 1u << i; s.data[i].load(base + gfix[i]); }
                    else s.data[i].zero();
                }
            }
        } else {
            const int left = kend - k0 - r0;                                             // rows of this thread still in range
            if (!gather) {
                const T* tile = reinterpret_cast<const T*>(op.p) + (long)k0 * op.ld;    // wave-uniform
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    if (ok && RSTEP * i < left) { s.valid |= 1u << i; s.data[i].load(tile + voff[i]); }
                    else s.data[i].zero();
                }
            } else {
                const T* base = reinterpret_cast<const T*>(op.p) + gfix[0];
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    if (ok && RSTEP * i < left) {
                        s.valid |= 1u << i;
                        s.data[i].load(base + row_base(k0 + r0 + RSTEP * i, op.ld, op.gw, op.gh, op.gc));
                    } else s.data[i].zero();
                }
            }
        }
    }
    template <bool PRO>
    __device__ __forceinline__ void commit(Stager<T, TR, TC, LDT, XC, NT>& s, const OpDev& op, const ProTab& tab, T* lds, int outer0) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (PRO && op.pro != BF_PRO_NONE && ((s.valid >> i) & 1u)) {
                float sc[CH], sh[CH];
                if (op.pro != BF_PRO_GELU) {
                    int f, ch, slot;          // slot: column inside the staged table (KC: the channel itself; XC: the tile column)
                    if constexpr (!XC) { f = fidx[i]; ch = s.k0 + cc; if (ch >= op.nch) ch %= op.nch; slot = ch - tab.c_lo; }
                    else { f = (int)((unsigned)(s.k0 + r0 + RSTEP * i) / (unsigned)op.rpf); ch = outer0 + cc; slot = ch - tab.c_lo; if (ch >= op.nch) ch %= op.nch; }
                    const long o = tab.ok ? (long)(f - tab.f_lo) * tab.cw + slot : (long)f * op.nch + ch;
#pragma unroll
                    for (int j = 0; j < CH; j += 4) {     // 16-byte aligned: channels come in whole chunks
                        const float4 a4 = *reinterpret_cast<const float4*>(tab.sc + o + j);
                        const float4 b4 = tab.sh ? *reinterpret_cast<const float4*>(tab.sh + o + j) : float4{0.f, 0.f, 0.f, 0.f};
                        sc[j] = a4.x; sc[j + 1] = a4.y; sc[j + 2] = a4.z; sc[j + 3] = a4.w;
                        sh[j] = b4.x; sh[j + 1] = b4.y; sh[j + 2] = b4.z; sh[j + 3] = b4.w;
                    }
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    float v = s.data[i].get(j);
                    if (op.pro != BF_PRO_GELU) v = v * sc[j] + sh[j];
                    if (op.pro != BF_PRO_AFFINE) v = gelu_t<T>(v);
                    s.data[i].set(j, v);
                }
            }
            s.data[i].store(lds + loff[i]);
        }
    }
};

template <typename T, bool AXC, bool BXC, bool APRO, bool BPRO, int NSTAGE, int TM, int NT>
__global__ void __launch_bounds__(NT) gemm_kernel(int M, int N, int K, OpDev A, OpDev B, EpiDev E, int kper, int mt, int nt, int dbg) {
    using Cfg = GemmCfg<T>;
    constexpr int BK = Cfg::BK;
    constexpr int BM = 32 * TM;            // 2 x WN waves, TM x TN MFMA tiles of 16 x 16 per wave
    constexpr int WN = NT / 128, TN = 8 / WN;
    constexpr int LDA = AXC ? (BM + Cfg::PADX) : (BK + Cfg::PADK);
    constexpr int LDB = BXC ? (BN + Cfg::PADX) : (BK + Cfg::PADK);
    constexpr int A_ELEMS = AXC ? BK * LDA : BM * LDA;
    constexpr int B_ELEMS = BXC ? BK * LDB : BN * LDB;
    constexpr bool ANYPRO = APRO || BPRO;
    constexpr size_t STG_BYTES = (size_t)64 * (BN + 4) * sizeof(float);              // epilogue staging (64 rows x 132 floats)
    // bf16, no prologue table, 8 waves: two LDS tile buffers -- tile k+1 is committed while tile k feeds the MFMAs (other waves of
    // the workgroup), one barrier per K-step instead of two.  2 x 32 KB per workgroup still lets two workgroups share a CU.
    constexpr bool DB = sizeof(T) == 2 && !ANYPRO && NT == 512 && NSTAGE == 1;      // with the 16 KB prologue table two buffers would exceed the 64 KB static LDS limit
    constexpr size_t TILE_BYTES = (size_t)(A_ELEMS + B_ELEMS) * sizeof(T) * (DB ? 2 : 1);
    __shared__ __attribute__((aligned(16))) T lds[(TILE_BYTES > STG_BYTES ? TILE_BYTES : STG_BYTES) / sizeof(T)];
    __shared__ __attribute__((aligned(16))) float ltab[ANYPRO ? 2 * TAB : 4];
    T* lA = lds;
    T* lB = lds + A_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int seq = xcd_remap(blockIdx.x, gridDim.x);
    const int zt = seq / (mt * nt), rem = seq - zt * (mt * nt);
    const int m0 = (rem / nt) * BM;
    const int n0 = (rem % nt) * BN;
    const int kbeg = zt * kper;
    const int kend = min(K, kbeg + kper);

    using StA = Stager<T, AXC ? BK : BM, AXC ? BM : BK, LDA, AXC, NT>;
    using StB = Stager<T, BXC ? BK : BN, BXC ? BN : BK, LDB, BXC, NT>;
    StagerFixed<T, AXC ? BK : BM, AXC ? BM : BK, LDA, AXC, NT> fa_;
    StagerFixed<T, BXC ? BK : BN, BXC ? BN : BK, LDB, BXC, NT> fb_;
    fa_.setup(A, m0, M, tid, APRO);
    fb_.setup(B, n0, N, tid, BPRO);
    StA sa[NSTAGE];
    StB sb[NSTAGE];

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // NSTAGE K-tiles of loads in flight before the first one is needed
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
        if (kbeg + s * BK < kend) { fa_.issue(sa[s], A, kbeg + s * BK, kend); fb_.issue(sb[s], B, kbeg + s * BK, kend); }

    ProTab ta{}, tb{};
    if constexpr (APRO) {
        if constexpr (AXC) ta = stage_table(A, kbeg, kend - 1, m0, BM, ltab, ltab + TAB, tid, NT);
        else ta = stage_table(A, m0, min(M, m0 + BM) - 1, 0, A.nch, ltab, ltab + TAB, tid, NT);
    }
    if constexpr (BPRO) {
        if constexpr (BXC) tb = stage_table(B, kbeg, kend - 1, n0, BN, ltab, ltab + TAB, tid, NT);
        else tb = stage_table(B, n0, min(N, n0 + BN) - 1, 0, B.nch, ltab, ltab + TAB, tid, NT);
    }

    float csum[Chunk<T>::N];
#pragma unroll
    for (int j = 0; j < Chunk<T>::N; ++j) csum[j] = 0.f;
    const bool do_colsum = AXC && E.colsum != nullptr && n0 == 0;

    int k0 = kbeg;
    if constexpr (DB) {
        constexpr int BUF = A_ELEMS + B_ELEMS;
        auto colsum_acc = [&]() {
            if constexpr (AXC) {
                if (do_colsum) {
#pragma unroll
                    for (int i = 0; i < StA::NCH; ++i)
#pragma unroll
                        for (int j = 0; j < Chunk<T>::N; ++j) csum[j] += sa[0].data[i].get(j);
                }
            }
        };
        if (k0 < kend) {      // tile 0 -> buffer 0, tile 1 in flight
            colsum_acc();
            fa_.template commit<APRO>(sa[0], A, ta, lA, m0);
            fb_.template commit<BPRO>(sb[0], B, tb, lB, n0);
            if (k0 + BK < kend) { fa_.issue(sa[0], A, k0 + BK, kend); fb_.issue(sb[0], B, k0 + BK, kend); }
        }
        __syncthreads();
        int cur = 0;
        for (; k0 < kend; k0 += BK) {
            const bf16* cA = (const bf16*)lA + cur * BUF;
            const bf16* cB = (const bf16*)lB + cur * BUF;
#pragma unroll
            for (int kk = 0; kk < BK; kk += Cfg::KSTEP) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = frag_bf16<AXC, LDA>(cA, wm * (16 * TM) + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = frag_bf16<BXC, LDB>(cB, wn * (16 * TN) + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            if (k0 + BK < kend) {      // next tile into the other buffer (nobody reads it until the barrier below)
                colsum_acc();
                fa_.template commit<APRO>(sa[0], A, ta, lA + (cur ^ 1) * BUF, m0);
                fb_.template commit<BPRO>(sb[0], B, tb, lB + (cur ^ 1) * BUF, n0);
                if (k0 + 2 * BK < kend) { fa_.issue(sa[0], A, k0 + 2 * BK, kend); fb_.issue(sb[0], B, k0 + 2 * BK, kend); }
            }
            __syncthreads();
            cur ^= 1;
        }
    } else
    while (k0 < kend) {
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s) {
            if (k0 < kend) {
                __syncthreads();
                if (!(dbg & 4) || k0 == kbeg) {
                    if constexpr (AXC) {            // fused bias gradient: column sums of the raw A operand (d output)
                    if (do_colsum) {
#pragma unroll
                        for (int i = 0; i < StA::NCH; ++i)
#pragma unroll
                            for (int j = 0; j < Chunk<T>::N; ++j) csum[j] += sa[s].data[i].get(j);
                    }
                }
                fa_.template commit<APRO>(sa[s], A, ta, lA, m0);
                    fb_.template commit<BPRO>(sb[s], B, tb, lB, n0);
                }
                __syncthreads();
                if (!(dbg & 2) && k0 + NSTAGE * BK < kend) { fa_.issue(sa[s], A, k0 + NSTAGE * BK, kend); fb_.issue(sb[s], B, k0 + NSTAGE * BK, kend); }
                if (!(dbg & 8))
#pragma unroll
                for (int kk = 0; kk < BK; kk += Cfg::KSTEP) {
                    if constexpr (sizeof(T) == 2) {
                        bf16x8 fa[TM], fb[TN];
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[i] = frag_bf16<AXC, LDA>((const bf16*)lA, wm * (16 * TM) + i * 16, kk, lane);
#pragma unroll
                        for (int j = 0; j < TN; ++j) fb[j] = frag_bf16<BXC, LDB>((const bf16*)lB, wn * (16 * TN) + j * 16, kk, lane);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    } else {
                        float fa[TM], fb[TN];
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[i] = frag_f32<AXC, LDA>((const float*)lA, wm * (16 * TM) + i * 16, kk, lane);
#pragma unroll
                        for (int j = 0; j < TN; ++j) fb[j] = frag_f32<BXC, LDB>((const float*)lB, wn * (16 * TN) + j * 16, kk, lane);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    }
                }
                k0 += BK;
            }
        }
    }

    if constexpr (AXC) {
        if (do_colsum) {            // reduce the per-thread partial column sums over the row groups, one atomic per column
            constexpr int CH_ = Chunk<T>::N, CPR_ = BM / CH_, RG_ = NT / CPR_;
            float* red = reinterpret_cast<float*>(lds);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < CH_; ++j) red[(tid / CPR_) * BM + (tid % CPR_) * CH_ + j] = csum[j];
            __syncthreads();
            if (tid < BM && m0 + tid < M) {
                float t = 0.f;
                for (int g = 0; g < RG_; ++g) t += red[g * BM + tid];
                atomicAdd(E.colsum + m0 + tid, t);
            }
        }
    }
    // ------------------------------------------------------------------ epilogue (row-major through LDS)
    if ((dbg & 1) && acc[0][0][0] != 12345.678f) return;     // timing experiment: skip the epilogue
    if (AXC && E.zstride) {      // split-K into slabs (bf_gemm_slabs): this K-slice's own [M][ldc] image -- one addend per element, summed in slice order later
        EpiDev Ez = E;
        Ez.c = reinterpret_cast<float*>(E.c) + (long)zt * E.zstride;
        epilogue_rows<T, TM, TN, 2, WN, AXC>(acc, Ez, M, N, m0, n0, reinterpret_cast<float*>(lds), tid);
        return;
    }
    epilogue_rows<T, TM, TN, 2, WN, AXC>(acc, E, M, N, m0, n0, reinterpret_cast<float*>(lds), tid);
}

OpDev to_dev(const bf_operand* o) {
    OpDev d;
    d.p = o->p; d.ld = o->ld; d.layout = o->layout; d.seglen = o->seglen; d.segstride = o->segstride;
    d.gw = o->gw; d.gh = o->gh; d.gc = o->gc; d.pro = o->pro; d.sc = o->sc; d.sh = o->sh;
    d.rpf = o->rows_per_frame > 0 ? o->rows_per_frame : 1; d.nch = o->nch > 0 ? o->nch : 1;
    return d;
}

template <typename T>
int launch(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, int splitk, hipStream_t st, long zstride = 0, int* splits_out = nullptr) {
    constexpr int BK = GemmCfg<T>::BK;
    OpDev a = to_dev(A), b = to_dev(B);
    EpiDev e;
    e.bias = E->bias; e.colscale = E->colscale; e.colshift = E->colshift; e.aux_mode = E->aux_mode; e.aux = E->aux;
    e.ld_aux = E->ld_aux; e.out_mode = E->out_mode; e.c = E->c; e.ldc = E->ldc; e.seglen = E->seglen;
    e.segstride = E->segstride; e.gw = E->gw; e.gh = E->gh; e.gc = E->gc; e.gelu_out = E->gelu_out; e.colsum = E->colsum; e.rowscale = E->rowscale; e.rpg = E->rows_per_group > 0 ? E->rows_per_group : 1;
    if (splitk < 1) splitk = 1;
    int ktiles = bf_cdiv(K, BK);
    if (splitk > ktiles) splitk = ktiles;
    const int kper = bf_cdiv(ktiles, splitk) * BK;
    splitk = bf_cdiv(K, kper);
    e.zstride = zstride;
    if (splits_out) *splits_out = splitk;
    const int nt = bf_cdiv(N, BN);
    static const int small_env = bf_knob("BF_GEMM_SMALL", -1);
    // measured: with 8-wave workgroups the 128 x 128 tile beats 64 x 128 even on grids of < 2 workgroups per CU
    // ... except where 128-row tiles leave most CUs without a workgroup (the patch stages at batch 1: 18-72 row tiles): 64-row tiles then
    static const int few_env = bf_knob("BF_GEMM_FEW_TILES", 100);
    const bool few = splitk <= 1 && (long)bf_cdiv(M, 128) * nt < few_env && M > 64;
    const bool small = small_env >= 0 ? (small_env != 0 && M > 64) : (M <= 64 || few);
    const int bm = small ? 64 : 128;
    const int mt = bf_cdiv(M, bm);
    dim3 grid((unsigned)((long)mt * nt * splitk));
    const bool ax = A->layout == BF_LAY_XC, bx = B->layout == BF_LAY_XC;
    const double es = sizeof(T);
    static const int dbg = bf_knob("BF_GEMM_DEBUG", 0);
    static const int w8env = bf_knob("BF_GEMM_WAVES", 0);
    // measured (tools/gemm_bench.py, MI355X): with one register stage, 8 waves of TM x 2 tiles (<= 128 VGPRs: two 8-wave
    // workgroups per CU) win on every shape of this model
    const bool w8 = sizeof(T) == 2 && (w8env == 8 || w8env == 0);
    // one profiler name per kernel instantiation, so bench.py's per-kernel averages line up 1:1 with rocprofv3's rows
    static thread_local char pname[96];
    snprintf(pname, sizeof(pname), "gemm_kernel<%s,%s,%s,pro%s,tm%d,w%d>", sizeof(T) == 2 ? "bf16" : "f32", ax ? "xc" : "kc", bx ? "xc" : "kc",
             A->pro != BF_PRO_NONE ? "A" : B->pro != BF_PRO_NONE ? "B" : "0", small ? 2 : 4, w8 ? 8 : 4);
    BfProfScope prof(st, pname, 2.0 * M * N * K,
                     (double)M * K * es + (double)N * K * es + (double)M * N * (E->out_mode == BF_OUT_STORE ? es : 4.0) +
                         (E->aux_mode != BF_AUX_NONE ? (double)M * N * es : 0.0));
    const bool ap = A->pro != BF_PRO_NONE, bp = B->pro != BF_PRO_NONE;
    static const int st_env = bf_knob("BF_GEMM_STAGES", 0);
    const bool s1 = st_env == 1;
#define BF_GEMM_GO2(AX, BX, AP, BP, NS)                                                                                       \
    do {                                                                                                                      \
        if (w8) {                                                                                                             \
            if (small) hipLaunchKernelGGL((gemm_kernel<T, AX, BX, AP, BP, NS, 2, 512>), grid, dim3(512), 0, st, M, N, K, a, b, e, kper, mt, nt, dbg); \
            else hipLaunchKernelGGL((gemm_kernel<T, AX, BX, AP, BP, NS, 4, 512>), grid, dim3(512), 0, st, M, N, K, a, b, e, kper, mt, nt, dbg);  \
        } else {                                                                                                              \
            if (small) hipLaunchKernelGGL((gemm_kernel<T, AX, BX, AP, BP, NS, 2, 256>), grid, dim3(256), 0, st, M, N, K, a, b, e, kper, mt, nt, dbg); \
            else hipLaunchKernelGGL((gemm_kernel<T, AX, BX, AP, BP, NS, 4, 256>), grid, dim3(256), 0, st, M, N, K, a, b, e, kper, mt, nt, dbg);  \
        }                                                                                                                     \
    } while (0)
    (void)s1;   // measured: one register stage (more workgroups per CU) beats three (deeper prefetch) on every shape of this model
#define BF_GEMM_GO(AX, BX, AP, BP) BF_GEMM_GO2(AX, BX, AP, BP, 1)
    if (!ax && !bx && !bp) { if (ap) BF_GEMM_GO(false, false, true, false); else BF_GEMM_GO(false, false, false, false); }
    else if (!ax && bx && !bp) { if (ap) BF_GEMM_GO(false, true, true, false); else BF_GEMM_GO(false, true, false, false); }
    else if (ax && bx && !ap) { if (bp) BF_GEMM_GO(true, true, false, true); else BF_GEMM_GO(true, true, false, false); }
    else return bf_fail_msg("bf_gemm: unsupported layout/prologue combination", __FILE__, __LINE__);
#undef BF_GEMM_GO
#undef BF_GEMM_GO2
    BF_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// gemm_stream.hip: weight-stationary persistent kernel for the short-K projections (0 = handled, 1 = not covered, < 0 = error)
int bf_gemm_stream_try(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, hipStream_t st);
int bf_gemm_pair_try(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, hipStream_t st);

extern "C" int bf_gemm(int dtype, int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E,
                       int splitk, bf_stream_t stream) {
    BF_REQUIRE(A && B && E && A->p && B->p && E->c, "bf_gemm: null operand");
    BF_REQUIRE(M > 0 && N > 0 && K > 0, "bf_gemm: empty problem");
    const int ch = dtype == BF_DTYPE_BF16 ? 8 : 4;
    // 16-byte chunk granularity along the contiguous direction of each operand
    BF_REQUIRE(A->layout == BF_LAY_XC ? (M % ch == 0) : (K % ch == 0), "bf_gemm: A contiguous extent must be a multiple of 16 bytes");
    BF_REQUIRE(B->layout == BF_LAY_XC ? (N % ch == 0) : (K % ch == 0), "bf_gemm: B contiguous extent must be a multiple of 16 bytes");
    BF_REQUIRE(A->ld % ch == 0 && B->ld % ch == 0, "bf_gemm: leading dims must be multiples of 16 bytes");
    BF_REQUIRE(A->seglen % ch == 0 && B->seglen % ch == 0 && A->segstride % ch == 0 && B->segstride % ch == 0 &&
               A->gc % ch == 0 && B->gc % ch == 0, "bf_gemm: gather geometry must keep 16-byte chunks whole");
    BF_REQUIRE(E->seglen % 4 == 0 && E->ldc % 4 == 0 && E->segstride % 4 == 0 && E->gc % 4 == 0,
               "bf_gemm: output geometry must keep 4-column groups whole");
    for (const bf_operand* o : {A, B}) {
        if (o->pro == BF_PRO_AFFINE || o->pro == BF_PRO_AFFINE_GELU) {
            BF_REQUIRE(o->sc && o->rows_per_frame > 0 && o->nch > 0 && o->nch % ch == 0,
                       "bf_gemm: affine prologue needs sc/sh, rows_per_frame and nch (multiple of the chunk)");
        }
    }
    BF_REQUIRE((A->layout == BF_LAY_XC) == (E->out_mode == BF_OUT_ATOMIC_F32),
               "bf_gemm: the token-reduction form (A outer-contiguous) accumulates with fp32 atomics, the other forms store");
    BF_REQUIRE((long)M < (1L << 31) && (long)N < (1L << 31) && (long)K < (1L << 31), "bf_gemm: extents must fit 31 bits");
    if (E->aux_mode != BF_AUX_NONE) BF_REQUIRE(E->aux != nullptr, "bf_gemm: aux pointer missing");
    BF_REQUIRE(!E->colsum || A->layout == BF_LAY_XC, "bf_gemm: colsum is defined for the token-reduction form only");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BF_DTYPE_BF16) {
        const int rc = bf_gemm_stream_try(M, N, K, A, B, E, st);
        if (rc <= 0) return rc;          // handled (0) or failed (< 0); 1 = not covered by the streaming kernel
        const int rp = bf_gemm_pair_try(M, N, K, A, B, E, st);      // long-K data gradients on 288-row tiles (gemm_frame.hip)
        if (rp <= 0) return rp;
    }
    if (dtype == BF_DTYPE_BF16) return launch<bf16>(M, N, K, A, B, E, splitk, st);
    if (dtype == BF_DTYPE_F32) return launch<float>(M, N, K, A, B, E, splitk, st);
    return bf_fail_msg("bf_gemm: unknown dtype", __FILE__, __LINE__);
}

// ---- split-K without float atomics on shared addresses (library-internal; bf_common.h).  The token-reduction form of bf_gemm with
// out = [M][ldc] fp32, C (+)= sum over K-slices: every slice adds its tile into ITS OWN zeroed image in `ws` (one addend per element: the
// atomic is a plain add there) and a second small kernel sums the images in slice order -- the same bits every run.  Returns 1 (nothing
// launched) when the workspace is too small; the caller then runs bf_gemm with its atomics.
namespace {
__global__ void __launch_bounds__(256) slab_sum_kernel(const float* __restrict__ ws, long n4, int splits, long zstride, float* __restrict__ out, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = accumulate ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < splits; s0 += 8) {          // eight slices in flight
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s0 + u < splits ? reinterpret_cast<const float4*>(ws + (long)(s0 + u) * zstride)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    reinterpret_cast<float4*>(out)[i] = a;
}
}  // namespace
int bf_gemm_slabs(int dtype, int M, int N, int K, const bf_operand* A, const bf_operand* B, float* out, long ldc, int accumulate, int splitk,
                  float* ws, long ws_floats, hipStream_t st) {
    BF_REQUIRE(A && B && out && ws && A->layout == BF_LAY_XC && ldc % 4 == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)ws & 15) == 0, "bf_gemm_slabs: bad arguments");
    constexpr int BK = 64;
    int s = splitk < 1 ? 1 : splitk;
    const int ktiles = bf_cdiv(K, BK);
    if (s > ktiles) s = ktiles;
    s = bf_cdiv(K, bf_cdiv(ktiles, s) * BK);          // the slice count launch() settles on
    const long zstride = (long)M * ldc;
    if ((long)s * zstride > ws_floats) return 1;
    if (hipMemsetAsync(ws, 0, (size_t)s * zstride * 4, st) != hipSuccess) return bf_fail_msg("bf_gemm_slabs: memset failed", __FILE__, __LINE__);
    bf_epilogue e = {};
    e.out_mode = BF_OUT_ATOMIC_F32; e.c = ws; e.ldc = ldc;
    int splits = 0;
    const int rc = dtype == BF_DTYPE_BF16 ? launch<bf16>(M, N, K, A, B, &e, s, st, zstride, &splits) : launch<float>(M, N, K, A, B, &e, s, st, zstride, &splits);
    if (rc) return rc;
    const long n4 = zstride / 4;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)bf_cdiv(n4, 256)), dim3(256), 0, st, ws, n4, splits, zstride, out, accumulate);
    BF_CHECK_LAUNCH();
    return 0;
}
