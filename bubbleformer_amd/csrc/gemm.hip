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
// 128x128 block tile, 4 waves (2x2), 64x64 per wave as 4x4 MFMA 16x16 tiles:
//   bf16: v_mfma_f32_16x16x32_bf16, BK = 64;   f32: v_mfma_f32_16x16x4_f32 (exact fp32), BK = 32.
// The MFMA is issued "swapped" (B fragment as the first operand) so each lane ends up with 4
// consecutive output COLUMNS of one row: 8-/16-byte epilogue stores and vector bias loads.
// XC (outer-contiguous) bf16 tiles are read with the gfx950 transposing LDS read (ds_read_b64_tr_b16).
#include "bf_common.h"
#include <stdlib.h>

namespace {

constexpr int BN = 128, NT = 256;
#ifndef BF_GEMM_DEFAULT_STAGES
#define BF_GEMM_DEFAULT_STAGES 1
#endif

template <typename T> struct GemmCfg;
template <> struct GemmCfg<bf16> { static constexpr int BK = 64, KSTEP = 32, PADK = 8, PADX = 8; };
template <> struct GemmCfg<float> { static constexpr int BK = 32, KSTEP = 4, PADK = 4, PADX = 4; };

struct OpDev {
    const void* p; long ld; int layout; int seglen; long segstride; int gw, gh, gc;
    int pro; const float* sc; const float* sh; int rpf; int nch;
};
struct EpiDev {
    const float* bias; const float* colscale; const float* colshift; int aux_mode; const void* aux; long ld_aux;
    int out_mode; void* c; long ldc; int seglen; long segstride; int gw, gh, gc; void* gelu_out;
};

__device__ __forceinline__ long row_base(int row, long ld, int gw, int gh, int gc) {
    if (gw <= 0) return (long)row * ld;
    const unsigned ur = (unsigned)row;
    const unsigned x = ur % (unsigned)gw, t = ur / (unsigned)gw;
    const unsigned y = t % (unsigned)gh, f = t / (unsigned)gh;
    return ((long)(f * 2u * gh + 2u * y) * (2L * gw) + 2 * x) * gc;
}
__device__ __forceinline__ long col_off(int col, int seglen, long segstride) {
    if (seglen <= 0) return col;
    const unsigned q = (unsigned)col / (unsigned)seglen;
    return (long)q * segstride + (col - (int)q * seglen);
}

// Prologue table: the per-(frame, channel) scale/shift an operand needs, staged once per block into LDS.
constexpr int TAB = 2048;   // floats per array (sc, sh): 16 KiB together
struct ProTab {
    const float* sc; const float* sh;   // LDS (ok) or global (fallback)
    int f_lo, c_lo, cw; bool ok;
};
// rows [r_lo, r_hi] of the operand (memory rows = tokens), channels [c_lo, c_lo + cw)
__device__ __forceinline__ ProTab stage_table(const OpDev& op, long r_lo, long r_hi, int c_lo, int cw, float* lds_sc, float* lds_sh, int tid) {
    ProTab t;
    t.f_lo = (int)(r_lo / op.rpf);
    const int nf = (int)(r_hi / op.rpf) - t.f_lo + 1;
    t.c_lo = c_lo; t.cw = cw;
    t.ok = (op.pro == BF_PRO_AFFINE || op.pro == BF_PRO_AFFINE_GELU) && (long)nf * cw <= TAB;
    if (t.ok) {
        for (int i = tid; i < nf * cw; i += NT) {
            const int fi = i / cw, c = c_lo + i % cw;
            const bool v = c < op.nch;
            lds_sc[i] = v ? op.sc[(long)(t.f_lo + fi) * op.nch + c] : 0.f;
            lds_sh[i] = v ? op.sh[(long)(t.f_lo + fi) * op.nch + c] : 0.f;
        }
        t.sc = lds_sc; t.sh = lds_sh;
    } else {
        t.sc = op.sc; t.sh = op.sh;
    }
    return t;
}

// Stage one operand tile.  TR x TC elements, row-major in LDS with leading dim LDT.
// KC: rows = outer index (BM/BN, fixed for the block), cols = k (advance per K-tile).
// XC: rows = k (advance per K-tile), cols = outer index (fixed for the block).
// The fixed half of every chunk address is computed once (setup); issue() adds the moving half and puts the 16-byte
// loads in flight; commit() (later) applies the prologue and writes LDS.  32-bit index math throughout.
template <typename T, int TR, int TC, int LDT, bool XC>
struct Stager {
    static constexpr int CH = Chunk<T>::N;
    static constexpr int CPR = TC / CH;
    static constexpr int NCH = TR * TC / CH / NT;   // chunks per thread; chunk i sits at tile row r0 + (NT / CPR) * i, tile col cc
    static constexpr int RSTEP = NT / CPR;
    Chunk<T> data[NCH];
    unsigned valid;
    int k0;
};
template <typename T, int TR, int TC, int LDT, bool XC>
struct StagerFixed {
    static constexpr int CH = Chunk<T>::N;
    static constexpr int CPR = TC / CH;
    static constexpr int NCH = TR * TC / CH / NT;
    static constexpr int RSTEP = NT / CPR;
    long fixed[XC ? 1 : NCH];      // KC: row_base per chunk row; XC: col_off of the thread's column
    int fidx[XC ? 1 : NCH];        // KC: frame of each chunk row (prologue)
    unsigned ok;                   // KC: bit i = row in range; XC: bit 0 = column in range
    int r0, cc;

    __device__ __forceinline__ void setup(const OpDev& op, int outer0, int nouter, int tid) {
        r0 = tid / CPR; cc = (tid % CPR) * CH;
        ok = 0u;
        if constexpr (!XC) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int row = outer0 + r0 + RSTEP * i;
                const bool v = row < nouter;
                ok |= (v ? 1u : 0u) << i;
                fixed[i] = v ? row_base(row, op.ld, op.gw, op.gh, op.gc) : 0;
                fidx[i] = row / op.rpf;
            }
        } else {
            const int col = outer0 + cc;
            ok = col < nouter ? 1u : 0u;
            fixed[0] = col_off(col, op.seglen, op.segstride);
            fidx[0] = 0;
        }
    }
    __device__ __forceinline__ void issue(Stager<T, TR, TC, LDT, XC>& s, const OpDev& op, int k0, int kend) const {
        const T* base = reinterpret_cast<const T*>(op.p);
        s.k0 = k0; s.valid = 0u;
        if constexpr (!XC) {
            const int col = k0 + cc;
            const bool cv = col < kend;
            const long co = col_off(col, op.seglen, op.segstride);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (cv && ((ok >> i) & 1u)) { s.valid |= 1u << i; s.data[i].load(base + fixed[i] + co); }
                else s.data[i].zero();
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int row = k0 + r0 + RSTEP * i;
                if (ok && row < kend) { s.valid |= 1u << i; s.data[i].load(base + row_base(row, op.ld, op.gw, op.gh, op.gc) + fixed[0]); }
                else s.data[i].zero();
            }
        }
    }
    template <bool PRO>
    __device__ __forceinline__ void commit(Stager<T, TR, TC, LDT, XC>& s, const OpDev& op, const ProTab& tab, T* lds, int outer0) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = r0 + RSTEP * i;
            if (PRO && op.pro != BF_PRO_NONE && ((s.valid >> i) & 1u)) {
                const float* sp = nullptr;
                const float* hp = nullptr;
                if (op.pro != BF_PRO_GELU) {
                    int f, ch;
                    if constexpr (!XC) { f = fidx[i]; ch = (s.k0 + cc) % op.nch; }
                    else { f = (s.k0 + r) / op.rpf; ch = (outer0 + cc) % op.nch; }
                    const long o = tab.ok ? (long)(f - tab.f_lo) * tab.cw + (ch - tab.c_lo) : (long)f * op.nch + ch;
                    sp = tab.sc + o; hp = tab.sh + o;
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    float v = s.data[i].get(j);
                    if (op.pro != BF_PRO_GELU) v = v * sp[j] + hp[j];
                    if (op.pro != BF_PRO_AFFINE) v = gelu_f(v);
                    s.data[i].set(j, v);
                }
            }
            s.data[i].store(lds + r * LDT + cc);
        }
    }
};

// ----------------------------------------------------------------------------- fragments
// bf16: 8 consecutive k for tile row (lane & 15), k-group lane >> 4.
template <bool XC, int LDT>
__device__ __forceinline__ bf16x8 frag_bf16(const bf16* t, int outer, int k0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if constexpr (!XC) {
        return *reinterpret_cast<const bf16x8*>(t + (outer + i) * LDT + k0 + 8 * g);
    } else {
        // tile is [k][outer]; transposing read: lane 4q+p of a 16-lane group supplies row q, cols 4p..4p+3,
        // lane i receives column i of the 4 rows.
        const int q = i >> 2, p = i & 3;
        const bf16* a0 = t + (k0 + 8 * g + q) * LDT + outer + 4 * p;
        typedef __attribute__((address_space(3))) s16x4* lds_ptr;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 4 * LDT));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    }
}
template <bool XC, int LDT>
__device__ __forceinline__ float frag_f32(const float* t, int outer, int k0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if constexpr (!XC) return t[(outer + i) * LDT + k0 + g];
    else return t[(k0 + g) * LDT + outer + i];
}

// Tile order.  The grid is 1-D; workgroups are dealt round-robin over the 8 XCDs (private L2 each), so each XCD is
// given a CONTIGUOUS run of the tile sequence (split slowest, then m, n fastest): workgroups that share an activation
// row panel (or, for split-K, a token slice) run back to back on one L2.  Bijective for any tile count; placement only
// affects speed.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, x = bid % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
}

template <typename T>
__device__ __forceinline__ void store4(T* c, const float (&u)[4], bool full, int nleft) {
    if (full) {
        if constexpr (sizeof(T) == 2) {
            const bf16x4 o = {(bf16)u[0], (bf16)u[1], (bf16)u[2], (bf16)u[3]};
            *reinterpret_cast<bf16x4*>(c) = o;
        } else {
            *reinterpret_cast<float4*>(c) = make_float4(u[0], u[1], u[2], u[3]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nleft) c[r] = from_f<T>(u[r]);
    }
}

template <typename T, bool AXC, bool BXC, bool APRO, bool BPRO, int NSTAGE, int TM>
__global__ void __launch_bounds__(NT) gemm_kernel(int M, int N, int K, OpDev A, OpDev B, EpiDev E, int kper, int mt, int nt) {
    using Cfg = GemmCfg<T>;
    constexpr int BK = Cfg::BK;
    constexpr int BM = 32 * TM;            // 2 x 2 waves, TM x 4 MFMA tiles of 16 x 16 per wave
    constexpr int LDA = AXC ? (BM + Cfg::PADX) : (BK + Cfg::PADK);
    constexpr int LDB = BXC ? (BN + Cfg::PADX) : (BK + Cfg::PADK);
    constexpr int A_ELEMS = AXC ? BK * LDA : BM * LDA;
    constexpr int B_ELEMS = BXC ? BK * LDB : BN * LDB;
    constexpr bool ANYPRO = APRO || BPRO;
    __shared__ __attribute__((aligned(16))) T lds[A_ELEMS + B_ELEMS];
    __shared__ __attribute__((aligned(16))) float ltab[ANYPRO ? 2 * TAB : 4];
    T* lA = lds;
    T* lB = lds + A_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int seq = xcd_remap(blockIdx.x, gridDim.x);
    const int zt = seq / (mt * nt), rem = seq - zt * (mt * nt);
    const int m0 = (rem / nt) * BM;
    const int n0 = (rem % nt) * BN;
    const int kbeg = zt * kper;
    const int kend = min(K, kbeg + kper);

    using StA = Stager<T, AXC ? BK : BM, AXC ? BM : BK, LDA, AXC>;
    using StB = Stager<T, BXC ? BK : BN, BXC ? BN : BK, LDB, BXC>;
    StagerFixed<T, AXC ? BK : BM, AXC ? BM : BK, LDA, AXC> fa_;
    StagerFixed<T, BXC ? BK : BN, BXC ? BN : BK, LDB, BXC> fb_;
    fa_.setup(A, m0, M, tid);
    fb_.setup(B, n0, N, tid);
    StA sa[NSTAGE];
    StB sb[NSTAGE];

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // NSTAGE K-tiles of loads in flight before the first one is needed
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
        if (kbeg + s * BK < kend) { fa_.issue(sa[s], A, kbeg + s * BK, kend); fb_.issue(sb[s], B, kbeg + s * BK, kend); }

    ProTab ta{}, tb{};
    if constexpr (APRO) {
        if constexpr (AXC) ta = stage_table(A, kbeg, kend - 1, m0, BM, ltab, ltab + TAB, tid);
        else ta = stage_table(A, m0, min(M, m0 + BM) - 1, 0, A.nch, ltab, ltab + TAB, tid);
    }
    if constexpr (BPRO) {
        if constexpr (BXC) tb = stage_table(B, kbeg, kend - 1, n0, BN, ltab, ltab + TAB, tid);
        else tb = stage_table(B, n0, min(N, n0 + BN) - 1, 0, B.nch, ltab, ltab + TAB, tid);
    }

    int k0 = kbeg;
    while (k0 < kend) {
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s) {
            if (k0 < kend) {
                __syncthreads();
                fa_.template commit<APRO>(sa[s], A, ta, lA, m0);
                fb_.template commit<BPRO>(sb[s], B, tb, lB, n0);
                __syncthreads();
                if (k0 + NSTAGE * BK < kend) { fa_.issue(sa[s], A, k0 + NSTAGE * BK, kend); fb_.issue(sb[s], B, k0 + NSTAGE * BK, kend); }
#pragma unroll
                for (int kk = 0; kk < BK; kk += Cfg::KSTEP) {
                    if constexpr (sizeof(T) == 2) {
                        bf16x8 fa[TM], fb[4];
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[i] = frag_bf16<AXC, LDA>((const bf16*)lA, wm * (16 * TM) + i * 16, kk, lane);
#pragma unroll
                        for (int j = 0; j < 4; ++j) fb[j] = frag_bf16<BXC, LDB>((const bf16*)lB, wn * 64 + j * 16, kk, lane);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    } else {
                        float fa[TM], fb[4];
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[i] = frag_f32<AXC, LDA>((const float*)lA, wm * (16 * TM) + i * 16, kk, lane);
#pragma unroll
                        for (int j = 0; j < 4; ++j) fb[j] = frag_f32<BXC, LDB>((const float*)lB, wn * 64 + j * 16, kk, lane);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    }
                }
                k0 += BK;
            }
        }
    }

    // ------------------------------------------------------------------ epilogue
    // lane holds rows m = .. + (lane & 15), columns n = .. + 4*(lane >> 4) + {0..3}.  Column vectors are fetched once per
    // column group; full 4-column groups take the vector path, the ragged right edge a scalar one.
    constexpr bool ATOMIC = AXC;            // the token-reduction (dW) form accumulates split-K partials
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + 4 * lg;
        if (n >= N) continue;
        const bool full = (n + 3 < N);
        float cb[4] = {0.f, 0.f, 0.f, 0.f}, cs[4] = {1.f, 1.f, 1.f, 1.f}, ch[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = min(n + r, N - 1);
            if (E.bias) cb[r] = E.bias[nn];
            if (E.colscale) { cs[r] = E.colscale[nn]; ch[r] = E.colshift[nn]; }
        }
        const long coff = col_off(n, E.seglen, E.segstride);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (16 * TM) + i * 16 + li;
            if (m >= M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (acc[i][j][r] + cb[r]) * cs[r] + ch[r];
            const long off = row_base(m, E.ldc, E.gw, E.gh, E.gc) + coff;
            if constexpr (ATOMIC) {
                float* c = reinterpret_cast<float*>(E.c) + off;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) atomicAdd(c + r, v[r]);
            } else {
                if (E.aux_mode != BF_AUX_NONE) {
                    const T* aux = reinterpret_cast<const T*>(E.aux) + (long)m * E.ld_aux + n;
                    float a[4];
                    if (full) {
                        if constexpr (sizeof(T) == 2) { const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(aux); a[0] = (float)t4[0]; a[1] = (float)t4[1]; a[2] = (float)t4[2]; a[3] = (float)t4[3]; }
                        else { const float4 t4 = *reinterpret_cast<const float4*>(aux); a[0] = t4.x; a[1] = t4.y; a[2] = t4.z; a[3] = t4.w; }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[r] = (n + r < N) ? to_f(aux[r]) : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (E.aux_mode == BF_AUX_ADD) ? (v[r] + a[r]) : (v[r] * dgelu_f(a[r]));
                }
                if (E.out_mode == BF_OUT_STORE_F32) {
                    float* c = reinterpret_cast<float*>(E.c) + off;
                    if (full) *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < N) c[r] = v[r];
                    }
                } else {
                    store4<T>(reinterpret_cast<T*>(E.c) + off, v, full, N - n);
                    if (E.gelu_out) {     // second output: gelu(v) (the MLP hidden activation next to its pre-activation)
                        const float u[4] = {gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])};
                        store4<T>(reinterpret_cast<T*>(E.gelu_out) + off, u, full, N - n);
                    }
                }
            }
        }
    }
}

OpDev to_dev(const bf_operand* o) {
    OpDev d;
    d.p = o->p; d.ld = o->ld; d.layout = o->layout; d.seglen = o->seglen; d.segstride = o->segstride;
    d.gw = o->gw; d.gh = o->gh; d.gc = o->gc; d.pro = o->pro; d.sc = o->sc; d.sh = o->sh;
    d.rpf = o->rows_per_frame > 0 ? o->rows_per_frame : 1; d.nch = o->nch > 0 ? o->nch : 1;
    return d;
}

template <typename T>
int launch(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, int splitk, hipStream_t st) {
    constexpr int BK = GemmCfg<T>::BK;
    OpDev a = to_dev(A), b = to_dev(B);
    EpiDev e;
    e.bias = E->bias; e.colscale = E->colscale; e.colshift = E->colshift; e.aux_mode = E->aux_mode; e.aux = E->aux;
    e.ld_aux = E->ld_aux; e.out_mode = E->out_mode; e.c = E->c; e.ldc = E->ldc; e.seglen = E->seglen;
    e.segstride = E->segstride; e.gw = E->gw; e.gh = E->gh; e.gc = E->gc; e.gelu_out = E->gelu_out;
    if (splitk < 1) splitk = 1;
    int ktiles = bf_cdiv(K, BK);
    if (splitk > ktiles) splitk = ktiles;
    const int kper = bf_cdiv(ktiles, splitk) * BK;
    splitk = bf_cdiv(K, kper);
    // 128 x 128 tiles unless that leaves the chip with fewer than two workgroups per CU: then 64 x 128
    const int nt = bf_cdiv(N, BN);
    const bool small = (long)bf_cdiv(M, 128) * nt * splitk < 512 && M > 64;
    const int bm = small ? 64 : 128;
    const int mt = bf_cdiv(M, bm);
    dim3 grid((unsigned)((long)mt * nt * splitk));
    const bool ax = A->layout == BF_LAY_XC, bx = B->layout == BF_LAY_XC;
    const double es = sizeof(T);
    const char* pname = sizeof(T) == 2 ? (ax ? "gemm_bf16_dW(xc,xc)" : bx ? "gemm_bf16_dA(kc,xc)" : "gemm_bf16_fwd(kc,kc)")
                                       : (ax ? "gemm_f32_dW(xc,xc)" : bx ? "gemm_f32_dA(kc,xc)" : "gemm_f32_fwd(kc,kc)");
    BfProfScope prof(st, pname, 2.0 * M * N * K,
                     (double)M * K * es + (double)N * K * es + (double)M * N * (E->out_mode == BF_OUT_STORE ? es : 4.0) +
                         (E->aux_mode != BF_AUX_NONE ? (double)M * N * es : 0.0));
    const bool ap = A->pro != BF_PRO_NONE, bp = B->pro != BF_PRO_NONE;
#define BF_GEMM_GO(AX, BX, AP, BP)                                                                                            \
    do {                                                                                                                      \
        if (small) hipLaunchKernelGGL((gemm_kernel<T, AX, BX, AP, BP, 3, 2>), grid, dim3(NT), 0, st, M, N, K, a, b, e, kper, mt, nt); \
        else hipLaunchKernelGGL((gemm_kernel<T, AX, BX, AP, BP, 3, 4>), grid, dim3(NT), 0, st, M, N, K, a, b, e, kper, mt, nt);  \
    } while (0)
    if (!ax && !bx && !bp) { if (ap) BF_GEMM_GO(false, false, true, false); else BF_GEMM_GO(false, false, false, false); }
    else if (!ax && bx && !bp) { if (ap) BF_GEMM_GO(false, true, true, false); else BF_GEMM_GO(false, true, false, false); }
    else if (ax && bx && !ap) { if (bp) BF_GEMM_GO(true, true, false, true); else BF_GEMM_GO(true, true, false, false); }
    else return bf_fail_msg("bf_gemm: unsupported layout/prologue combination", __FILE__, __LINE__);
#undef BF_GEMM_GO
    BF_CHECK_LAUNCH();
    return 0;
}

}  // namespace

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
            BF_REQUIRE(o->sc && o->sh && o->rows_per_frame > 0 && o->nch > 0 && o->nch % ch == 0,
                       "bf_gemm: affine prologue needs sc/sh, rows_per_frame and nch (multiple of the chunk)");
        }
    }
    BF_REQUIRE((A->layout == BF_LAY_XC) == (E->out_mode == BF_OUT_ATOMIC_F32),
               "bf_gemm: the token-reduction form (A outer-contiguous) accumulates with fp32 atomics, the other forms store");
    BF_REQUIRE((long)M < (1L << 31) && (long)N < (1L << 31) && (long)K < (1L << 31), "bf_gemm: extents must fit 31 bits");
    if (E->aux_mode != BF_AUX_NONE) BF_REQUIRE(E->aux != nullptr, "bf_gemm: aux pointer missing");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BF_DTYPE_BF16) return launch<bf16>(M, N, K, A, B, E, splitk, st);
    if (dtype == BF_DTYPE_F32) return launch<float>(M, N, K, A, B, E, splitk, st);
    return bf_fail_msg("bf_gemm: unknown dtype", __FILE__, __LINE__);
}
