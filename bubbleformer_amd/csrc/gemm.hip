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

namespace {

constexpr int BM = 128, BN = 128, NT = 256;

template <typename T> struct GemmCfg;
template <> struct GemmCfg<bf16> { static constexpr int BK = 64, KSTEP = 32, PADK = 8, PADX = 8; };
template <> struct GemmCfg<float> { static constexpr int BK = 32, KSTEP = 4, PADK = 4, PADX = 4; };

struct OpDev {
    const void* p; long ld; int layout; int seglen; long segstride; int gw, gh, gc;
    int pro; const float* sc; const float* sh; int rpf; int nch;
};
struct EpiDev {
    const float* bias; const float* colscale; const float* colshift; int aux_mode; const void* aux; long ld_aux;
    int out_mode; void* c; long ldc; int seglen; long segstride; int gw, gh, gc;
};

__device__ __forceinline__ long row_base(long row, long ld, int gw, int gh, int gc) {
    if (gw <= 0) return row * ld;
    const int x = (int)(row % gw);
    const long t = row / gw;
    const int y = (int)(t % gh);
    const long f = t / gh;
    return ((f * 2 * gh + 2 * y) * (2L * gw) + 2 * x) * gc;
}
__device__ __forceinline__ long col_off(int col, int seglen, long segstride) {
    if (seglen <= 0) return col;
    return (long)(col / seglen) * segstride + (col % seglen);
}

// Stage one operand tile.  TR x TC elements, row-major in LDS with leading dim LDT.
// KC: rows = outer index (BM/BN), cols = k.  XC: rows = k, cols = outer index.
template <typename T, int TR, int TC, int LDT, bool PRO>
struct Stager {
    static constexpr int CH = Chunk<T>::N;
    static constexpr int CPR = TC / CH;
    static constexpr int NCH = TR * TC / CH / NT;   // chunks per thread
    Chunk<T> data[NCH];
    float sc[PRO ? NCH : 1][CH], sh[PRO ? NCH : 1][CH];
    bool valid[NCH];

    // row0/col0: global row / col of the tile origin; nrows/ncols: global extents
    __device__ __forceinline__ void issue(const OpDev& op, long row0, long nrows, int col0, int ncols, int tid) {
        const T* base = reinterpret_cast<const T*>(op.p);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + NT * i;
            const int r = c / CPR, cc = (c % CPR) * CH;
            const long row = row0 + r;
            const int col = col0 + cc;
            valid[i] = (row < nrows) && (col < ncols);
            if (valid[i]) {
                const long a = row_base(row, op.ld, op.gw, op.gh, op.gc) + col_off(col, op.seglen, op.segstride);
                data[i].load(base + a);
                if (PRO && (op.pro == BF_PRO_AFFINE || op.pro == BF_PRO_AFFINE_GELU)) {
                    const long f = row / op.rpf;
                    const int ch = col % op.nch;
                    const float* s = op.sc + f * op.nch + ch;
                    const float* h = op.sh + f * op.nch + ch;
#pragma unroll
                    for (int j = 0; j < CH; j += 4) {
                        const float4 a4 = *reinterpret_cast<const float4*>(s + j);
                        const float4 b4 = *reinterpret_cast<const float4*>(h + j);
                        constexpr int ii_dummy = 0; (void)ii_dummy;
                        const int ii = PRO ? i : 0;
                        sc[ii][j] = a4.x; sc[ii][j + 1] = a4.y; sc[ii][j + 2] = a4.z; sc[ii][j + 3] = a4.w;
                        sh[ii][j] = b4.x; sh[ii][j + 1] = b4.y; sh[ii][j + 2] = b4.z; sh[ii][j + 3] = b4.w;
                    }
                }
            } else {
                data[i].zero();
            }
        }
    }
    __device__ __forceinline__ void commit(const OpDev& op, T* lds, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + NT * i;
            const int r = c / CPR, cc = (c % CPR) * CH;
            if (PRO && op.pro != BF_PRO_NONE && valid[i]) {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    float v = data[i].get(j);
                    if (op.pro != BF_PRO_GELU) v = v * sc[PRO ? i : 0][j] + sh[PRO ? i : 0][j];
                    if (op.pro != BF_PRO_AFFINE) v = gelu_f(v);
                    data[i].set(j, v);
                }
            }
            data[i].store(lds + r * LDT + cc);
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

template <typename T, bool AXC, bool BXC, bool APRO, bool BPRO>
__global__ void __launch_bounds__(NT) gemm_kernel(int M, int N, int K, OpDev A, OpDev B, EpiDev E, int kper) {
    using Cfg = GemmCfg<T>;
    constexpr int BK = Cfg::BK;
    constexpr int LDA = AXC ? (BM + Cfg::PADX) : (BK + Cfg::PADK);
    constexpr int LDB = BXC ? (BN + Cfg::PADX) : (BK + Cfg::PADK);
    constexpr int A_ELEMS = AXC ? BK * LDA : BM * LDA;
    constexpr int B_ELEMS = BXC ? BK * LDB : BN * LDB;
    __shared__ __attribute__((aligned(16))) T lds[A_ELEMS + B_ELEMS];
    T* lA = lds;
    T* lB = lds + A_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // blockIdx.x walks M fastest so neighbouring blocks share the weight panel in L2
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * kper;
    const int kend = min(K, kbeg + kper);

    using StA = Stager<T, AXC ? BK : BM, AXC ? BM : BK, LDA, APRO>;
    using StB = Stager<T, BXC ? BK : BN, BXC ? BN : BK, LDB, BPRO>;
    StA sa;
    StB sb;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int k0) {
        if constexpr (AXC) sa.issue(A, k0, kend, (int)m0, M, tid);
        else sa.issue(A, m0, M, k0, kend, tid);
        if constexpr (BXC) sb.issue(B, k0, kend, n0, N, tid);
        else sb.issue(B, n0, N, k0, kend, tid);
    };

    if (kbeg < kend) issue(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        sa.commit(A, lA, tid);
        sb.commit(B, lB, tid);
        __syncthreads();
        if (k0 + BK < kend) issue(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += Cfg::KSTEP) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = frag_bf16<AXC, LDA>((const bf16*)lA, wm * 64 + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = frag_bf16<BXC, LDB>((const bf16*)lB, wn * 64 + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            } else {
                float fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = frag_f32<AXC, LDA>((const float*)lA, wm * 64 + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = frag_f32<BXC, LDB>((const float*)lB, wn * 64 + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ------------------------------------------------------------------ epilogue
    // lane holds rows m = .. + (lane & 15), columns n = .. + 4*(lane >> 4) + {0..3}
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = m0 + wm * 64 + i * 16 + li;
        if (m >= M) continue;
        const long cbase = row_base(m, E.ldc, E.gw, E.gh, E.gc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * lg;
            if (n >= N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const bool full = (n + 3 < N);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < N) {
                    if (E.bias) v[r] += E.bias[n + r];
                    if (E.colscale) v[r] = v[r] * E.colscale[n + r] + E.colshift[n + r];
                }
            }
            if (E.aux_mode != BF_AUX_NONE) {
                const T* aux = reinterpret_cast<const T*>(E.aux) + m * E.ld_aux + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r < N) {
                        const float a = to_f(aux[r]);
                        v[r] = (E.aux_mode == BF_AUX_ADD) ? (v[r] + a) : (v[r] * dgelu_f(a));
                    }
                }
            }
            const long off = cbase + col_off(n, E.seglen, E.segstride);
            if (E.out_mode == BF_OUT_ATOMIC_F32) {
                float* c = reinterpret_cast<float*>(E.c) + off;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) atomicAdd(c + r, v[r]);
            } else if (E.out_mode == BF_OUT_STORE_F32) {
                float* c = reinterpret_cast<float*>(E.c) + off;
                if (full) *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int r = 0; r < 4; ++r)
                        if (n + r < N) c[r] = v[r];
            } else {
                T* c = reinterpret_cast<T*>(E.c) + off;
                if (full) {
                    if constexpr (sizeof(T) == 2) {
                        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(c) = o;
                    } else {
                        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < N) c[r] = from_f<T>(v[r]);
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
    e.segstride = E->segstride; e.gw = E->gw; e.gh = E->gh; e.gc = E->gc;
    if (splitk < 1) splitk = 1;
    int ktiles = bf_cdiv(K, BK);
    if (splitk > ktiles) splitk = ktiles;
    const int kper = bf_cdiv(ktiles, splitk) * BK;
    splitk = bf_cdiv(K, kper);
    dim3 grid(bf_cdiv(M, BM), bf_cdiv(N, BN), splitk);
    const bool ax = A->layout == BF_LAY_XC, bx = B->layout == BF_LAY_XC;
    const double es = sizeof(T);
    const char* pname = sizeof(T) == 2 ? (ax ? "gemm_bf16_dW(xc,xc)" : bx ? "gemm_bf16_dA(kc,xc)" : "gemm_bf16_fwd(kc,kc)")
                                       : (ax ? "gemm_f32_dW(xc,xc)" : bx ? "gemm_f32_dA(kc,xc)" : "gemm_f32_fwd(kc,kc)");
    BfProfScope prof(st, pname, 2.0 * M * N * K,
                     (double)M * K * es + (double)N * K * es + (double)M * N * (E->out_mode == BF_OUT_STORE ? es : 4.0) +
                         (E->aux_mode != BF_AUX_NONE ? (double)M * N * es : 0.0));
    const bool ap = A->pro != BF_PRO_NONE, bp = B->pro != BF_PRO_NONE;
#define BF_GEMM_GO(AX, BX, AP, BP) \
    hipLaunchKernelGGL((gemm_kernel<T, AX, BX, AP, BP>), grid, dim3(NT), 0, st, M, N, K, a, b, e, kper)
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
    BF_REQUIRE(splitk <= 1 || E->out_mode == BF_OUT_ATOMIC_F32, "bf_gemm: split-K needs the atomic fp32 output mode");
    if (E->aux_mode != BF_AUX_NONE) BF_REQUIRE(E->aux != nullptr, "bf_gemm: aux pointer missing");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BF_DTYPE_BF16) return launch<bf16>(M, N, K, A, B, E, splitk, st);
    if (dtype == BF_DTYPE_F32) return launch<float>(M, N, K, A, B, E, splitk, st);
    return bf_fail_msg("bf_gemm: unknown dtype", __FILE__, __LINE__);
}
