// Device helpers shared by the GEMM kernels (gemm.hip tile kernel, gemm_frame.hip whole-frame tiles, gemm_stream.hip / gemm_tokred.hip LDS-DMA kernels).
#pragma once
#include "bf_common.h"

namespace bfgemm {

#ifndef PP_SETPRIO
#define PP_SETPRIO 0      // s_setprio around the MFMA clusters of the ping-pong kernels (measured, see DESIGN)
#endif

constexpr int TAB = 2048;   // floats per prologue-table array (sc, sh): 16 KiB together

struct OpDev {
    const void* p; long ld; int layout; int seglen; long segstride; int gw, gh, gc;
    int pro; const float* sc; const float* sh; int rpf; int nch;
};
struct EpiDev {
    const float* bias; const float* colscale; const float* colshift; int aux_mode; const void* aux; long ld_aux;
    int out_mode; void* c; long ldc; int seglen; long segstride; int gw, gh, gc; void* gelu_out; float* colsum; const float* rowscale; int rpg;
    long zstride;      // split-K into slabs: K-slice z adds into c + z * zstride floats (0: every slice into c itself)
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

// Element offset of (row r, column c) inside an LDS operand tile.
//  bf16 KC tile [outer][64]: 128-byte rows; the 16-byte chunk index is XORed with (r >> 1) & 7 so the 16 lanes that one
//    ds_read_b128 group serves (rows i, i+.., chunk g) land on 16 distinct 16-byte slots of the 256-byte bank row.
//  bf16 XC tile [k][128]: 256-byte rows; the 8-byte unit index is XORed with 4 * ((r & 3) | ((r >> 3) & 1) << 2): a half-wave
//    of the transposing read takes rows {q, 8 + q} (lo) or {4 + q, 12 + q} (hi), q = 0..3, units 4t + p -- eight rows whose
//    keys are all different, so its 32 lanes cover the 64 banks exactly once (SQ_LDS_BANK_CONFLICT = 0 in the dW GEMMs; keying
//    on r & 7 alone left rows q and 8 + q on the same banks).  64-wide XC tiles stay linear.
//  f32 tiles: padded leading dimension, no swizzle.
template <typename T, bool XC, int LDT>
__device__ __forceinline__ int lds_off(int r, int c) {
    if constexpr (sizeof(T) == 2 && !XC) return r * LDT + ((((c >> 3) ^ ((r >> 1) & 7)) << 3) | (c & 7));
    else if constexpr (sizeof(T) == 2 && XC && (LDT % 128 == 0)) return r * LDT + ((((c >> 2) ^ (((r & 3) | ((r >> 1) & 4)) << 2)) << 2) | (c & 3));
    else return r * LDT + c;
}

// Prologue table: the per-(frame, channel) scale/shift an operand needs, staged once per block into LDS.
struct ProTab {
    const float* sc; const float* sh;   // LDS (ok) or global (fallback)
    int f_lo, c_lo, cw; bool ok;
};
// rows [r_lo, r_hi] of the operand (memory rows = tokens), channels [c_lo, c_lo + cw)
__device__ __forceinline__ ProTab stage_table(const OpDev& op, long r_lo, long r_hi, int c_lo, int cw, float* lds_sc, float* lds_sh, int tid, int nthreads) {
    ProTab t;
    t.f_lo = (int)(r_lo / op.rpf);
    const int nf = (int)(r_hi / op.rpf) - t.f_lo + 1;
    t.c_lo = c_lo; t.cw = cw;
    t.ok = (op.pro == BF_PRO_AFFINE || op.pro == BF_PRO_AFFINE_GELU) && (long)nf * cw <= TAB;
    if (t.ok) {
        for (int i = tid; i < nf * cw; i += nthreads) {
            // table slot = column of the tile; its channel wraps with nch (k2s2 patch rows hold 4 pixels x nch channels, so an
            // outer-contiguous tile at column >= nch still needs channel (column mod nch))
            const int fi = i / cw, c = (c_lo + i % cw) % op.nch;
            lds_sc[i] = op.sc[(long)(t.f_lo + fi) * op.nch + c];
            lds_sh[i] = op.sh ? op.sh[(long)(t.f_lo + fi) * op.nch + c] : 0.f;     // sh == NULL: pure scale
        }
        t.sc = lds_sc; t.sh = lds_sh;
    } else {
        t.sc = op.sc; t.sh = op.sh;
    }
    return t;
}

// ----------------------------------------------------------------------------- fragments
// bf16: 8 consecutive k for tile row (lane & 15), k-group lane >> 4.
template <bool XC, int LDT>
__device__ __forceinline__ bf16x8 frag_bf16(const bf16* t, int outer, int k0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if constexpr (!XC) {
        return *reinterpret_cast<const bf16x8*>(t + lds_off<bf16, false, LDT>(outer + i, k0 + 8 * g));
    } else {
        // tile is [k][outer]; transposing read: lane 4q+p of a 16-lane group supplies row q, cols 4p..4p+3,
        // lane i receives column i of the 4 rows.
        const int q = i >> 2, p = i & 3;
        typedef __attribute__((address_space(3))) s16x4* lds_ptr;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(t + lds_off<bf16, true, LDT>(k0 + 8 * g + q, outer + 4 * p)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(t + lds_off<bf16, true, LDT>(k0 + 8 * g + q + 4, outer + 4 * p)));
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

// ----------------------------------------------------------------------------- LDS-DMA (global -> LDS, no staging registers)
// One global_load_lds_dwordx4: lane l copies the 16 bytes at its OWN source address to LDS byte lds_dst + 16 * l (lds_dst wave-uniform).
// Written as inline asm on purpose: hipcc models the builtin's LDS write and then drains vmcnt(0) before the next ds_read of the same
// array, which serialises every K-step; hidden from it, the DMA is ordered for readers by the counted s_waitcnt vmcnt + s_barrier the
// kernels place themselves (cdna_hip_programming.md section 5.7 item 1).  M0 (the DMA's LDS base) is saved and restored.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The linear epilogue of the streaming and the frame-pair kernels, written once (explicit fma) so that they agree bit for bit:
//   v = acc + bias;  scaled: v = (v * cs + ch) * rs;  with a residual: v = (v * cs + ch) * rs + aux  (one fma: rs = 1 makes it a plain add)
__device__ __forceinline__ float epi_lin(float acc, float bias, float cs, float ch, float rs, bool scaled) {
    const float v = acc + bias;
    return scaled ? fmaf(v, cs, ch) * rs : v;
}
__device__ __forceinline__ float epi_lin_add(float acc, float bias, float cs, float ch, float rs, bool scaled, float aux) {
    const float v = acc + bias;
    return scaled ? fmaf(fmaf(v, cs, ch), rs, aux) : v + aux;
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


// ----------------------------------------------------------------------------- epilogue
// Accumulators (MFMA D layout: lane = one row, 4 consecutive columns) are first given their per-column terms, then staged
// through LDS so that the global side runs in whole rows: every wave-instruction reads / writes 16-byte chunks of
// consecutive columns (4 rows x 256 contiguous bytes), instead of 16 rows x 32-byte pieces.  Residual / gelu' operands,
// the optional second (gelu) output and the fp32 atomics of the split-K form all use the row-major phase; the atomics are
// issued as 64 consecutive floats per wave-instruction (the shape the memory-side atomic unit runs at full rate).
template <typename T, int TM, int TN, int WM, int WN, bool ATOMIC>
__device__ __forceinline__ void epilogue_rows(const f32x4 (&acc)[TM][TN], const EpiDev& E, int M, int N, int m0, int n0, float* stg, int tid) {
    constexpr int BN_ = WN * TN * 16, LDS_ = BN_ + 4, NTHR = 64 * WM * WN;
    constexpr int SUB = TM >= 2 ? 2 : 1;                 // row sub-tiles per wave per pass
    constexpr int RP = WM * SUB * 16;                    // staged rows per pass
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;
    float cb[TN][4], cs[TN][4], ch[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = min(n0 + wn * (16 * TN) + j * 16 + 4 * lg + r, N - 1);
            cb[j][r] = E.bias ? E.bias[nn] : 0.f;
            cs[j][r] = E.colscale ? E.colscale[nn] : 1.f;
            ch[j][r] = E.colscale ? E.colshift[nn] : 0.f;
        }
    __syncthreads();                                     // operand tiles are dead: LDS becomes the staging buffer
#pragma unroll
    for (int p = 0; p < TM / SUB; ++p) {
#pragma unroll
        for (int ii = 0; ii < SUB; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x4 a = acc[p * SUB + ii][j];
                const float4 v = make_float4(fmaf(a[0] + cb[j][0], cs[j][0], ch[j][0]), fmaf(a[1] + cb[j][1], cs[j][1], ch[j][1]),
                                             fmaf(a[2] + cb[j][2], cs[j][2], ch[j][2]), fmaf(a[3] + cb[j][3], cs[j][3], ch[j][3]));      // explicit fma: the kernels agree bit for bit
                *reinterpret_cast<float4*>(stg + ((wm * SUB + ii) * 16 + li) * LDS_ + wn * (16 * TN) + j * 16 + 4 * lg) = v;
            }
        __syncthreads();
        if constexpr (ATOMIC) {
            for (int idx = tid; idx < RP * BN_; idx += NTHR) {
                const int srow = idx / BN_, c = idx - srow * BN_;
                const int m = m0 + (srow / (SUB * 16)) * (16 * TM) + (p * SUB + (srow % (SUB * 16)) / 16) * 16 + (srow % 16);
                const int n = n0 + c;
                if (m < M && n < N) atomicAdd(reinterpret_cast<float*>(E.c) + row_base(m, E.ldc, E.gw, E.gh, E.gc) + col_off(n, E.seglen, E.segstride), stg[srow * LDS_ + c]);
            }
        } else {
            constexpr int G = 8;                          // columns per thread per step
            for (int idx = tid; idx < RP * (BN_ / G); idx += NTHR) {
                const int srow = idx / (BN_ / G), c = (idx - srow * (BN_ / G)) * G;
                const int m = m0 + (srow / (SUB * 16)) * (16 * TM) + (p * SUB + (srow % (SUB * 16)) / 16) * 16 + (srow % 16);
                const int n = n0 + c;
                if (m >= M || n >= N) continue;
                const float4 lo = *reinterpret_cast<const float4*>(stg + srow * LDS_ + c);
                const float4 hi = *reinterpret_cast<const float4*>(stg + srow * LDS_ + c + 4);
                float v[G] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const bool full = n + G <= N;
                if (E.rowscale) {                           // stochastic depth: the whole branch value is scaled per sample
                    const float rs = E.rowscale[m / E.rpg];
#pragma unroll
                    for (int q = 0; q < G; ++q) v[q] *= rs;
                }
                if (E.aux_mode != BF_AUX_NONE) {
                    const T* aux = reinterpret_cast<const T*>(E.aux) + (long)m * E.ld_aux + n;
                    float a[G];
                    if (full) {
                        if constexpr (sizeof(T) == 2) { const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(aux);
#pragma unroll
                            for (int q = 0; q < G; ++q) a[q] = (float)t8[q]; }
                        else { const float4 t0 = *reinterpret_cast<const float4*>(aux), t1 = *reinterpret_cast<const float4*>(aux + 4);
                            a[0] = t0.x; a[1] = t0.y; a[2] = t0.z; a[3] = t0.w; a[4] = t1.x; a[5] = t1.y; a[6] = t1.z; a[7] = t1.w; }
                    } else {
#pragma unroll
                        for (int q = 0; q < G; ++q) a[q] = (n + q < N) ? to_f(aux[q]) : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < G; ++q) v[q] = (E.aux_mode == BF_AUX_ADD) ? (v[q] + a[q]) : (v[q] * dgelu_t<T>(a[q]));
                }
                // scatter geometry keeps 4-column groups whole (host-checked), so address the two halves separately
                const long rb = row_base(m, E.ldc, E.gw, E.gh, E.gc);
                const long o0 = rb + col_off(n, E.seglen, E.segstride);
                const long o1 = rb + col_off(n + 4, E.seglen, E.segstride);
                const bool contig = (o1 == o0 + 4) && full;
                if (E.out_mode == BF_OUT_STORE_F32) {
                    float* c0 = reinterpret_cast<float*>(E.c);
                    if (contig) { *reinterpret_cast<float4*>(c0 + o0) = make_float4(v[0], v[1], v[2], v[3]); *reinterpret_cast<float4*>(c0 + o0 + 4) = make_float4(v[4], v[5], v[6], v[7]); }
                    else {
#pragma unroll
                        for (int q = 0; q < G; ++q)
                            if (n + q < N) c0[(q < 4 ? o0 : o1 - 4) + q] = v[q];
                    }
                } else {
                    auto put = [&](T* c0, const float (&u)[G]) {
                        if (contig) {
                            if constexpr (sizeof(T) == 2) { bf16x8 o8;
#pragma unroll
                                for (int q = 0; q < G; ++q) o8[q] = (bf16)u[q];
                                *reinterpret_cast<bf16x8*>(c0 + o0) = o8; }
                            else { *reinterpret_cast<float4*>(c0 + o0) = make_float4(u[0], u[1], u[2], u[3]); *reinterpret_cast<float4*>(c0 + o0 + 4) = make_float4(u[4], u[5], u[6], u[7]); }
                        } else {
#pragma unroll
                            for (int q = 0; q < G; ++q)
                                if (n + q < N) c0[(q < 4 ? o0 : o1 - 4) + q] = from_f<T>(u[q]);
                        }
                    };
                    put(reinterpret_cast<T*>(E.c), v);
                    if (E.gelu_out) {
                        float u[G];
#pragma unroll
                        for (int q = 0; q < G; ++q) u[q] = gelu_t<T>(v[q]);
                        put(reinterpret_cast<T*>(E.gelu_out), u);
                    }
                }
            }
        }
        if (p + 1 < TM / SUB) __syncthreads();
    }
}

}  // namespace bfgemm
