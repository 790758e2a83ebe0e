// Common device helpers for the gfx950 (MI355X) FiLMAViT kernels.
// Wavefront = 64 lanes everywhere in this tree; no 32-lane idioms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/bubbleformer_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define BF_WAVE 64
#define BF_IN_EPS 1e-5f

// last launch error is kept per process; every extern "C" entry returns 0 or a negative code.
#define BF_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return bf_fail(e__, __FILE__, __LINE__); \
    } while (0)
#define BF_REQUIRE(cond, msg)                                \
    do {                                                     \
        if (!(cond)) return bf_fail_msg(msg, __FILE__, __LINE__); \
    } while (0)

// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).  Off by default;
// when off a scope costs one load of a global flag.
int bf_prof_begin(hipStream_t st);
bool bf_prof_is_on();
void bf_prof_end(int idx, hipStream_t st, const char* name, double flops, double bytes);
struct BfProfScope {
    int idx; hipStream_t st; const char* name; double flops, bytes;
    BfProfScope(hipStream_t s, const char* n, double f, double b) : idx(bf_prof_begin(s)), st(s), name(n), flops(f), bytes(b) {}
    ~BfProfScope() { if (idx >= 0) bf_prof_end(idx, st, name, flops, bytes); }
};

// internal halves of bf_in_bwd / bf_attn_bwd: everything but the parameter-gradient reduction of the workspace (param_reduce.h)
int bf_in_bwd_partials(int dtype, const void* dy, const void* x, const void* add, void* dx, int frames, int S, int C, const float* mean,
                       const float* rstd, const float* w, const float* b, const float* g, int gdiv, int gelu, float* ws, hipStream_t stream);
int bf_attn_bwd_partials(int dtype, const void* qkv, const void* dout, void* dqkv, int64_t nseq, int L, int64_t inner, int64_t outer_stride,
                         int64_t inner_stride, int64_t tok_stride, int heads, int d, const float* qw, const float* qb, const float* kw,
                         const float* kb, const float* emb, const float* hscale, float* dqw, float* dqb, float* dkw, float* dkb, float* demb,
                         float* dhscale, float out_scale, int accumulate, float* ws, int64_t ws_floats, int* rows, hipStream_t stream);
// InstanceNorm statistics of x fused with out = resid + x * sc + sh (norm.hip; internal)
int bf_in_stats_apply(int dtype, const void* x, int frames, int S, int C, const float* w, const float* b, const float* g, int gdiv, const float* gb,
                      float* mean, float* rstd, float* sc, float* sh, float* ws, const void* resid, void* out, hipStream_t stream);
int bf_in_stats_apply_chain(int dtype, const void* x, int frames, int S, int C, const float* w, const float* b, const float* g, int gdiv, const float* gb,
                            float* mean, float* rstd, float* sc, float* sh, float* ws, const void* resid, void* out, const float* nw, const float* nb,
                            float* nmean, float* nrstd, float* nsc, float* nsh, void* nxn, bool* chained, hipStream_t stream);
// out = z * m[(row / S) / fdiv] (norm.hip; internal)
int bf_wprep_multi(int dtype, int n, const int* mode, const float* const* src, void* const* dst, const int* R, const int* K, const int* Kp, hipStream_t st);
int bf_frame_scale(int dtype, const void* z, const float* m, int fdiv, void* out, long nrows, int S, int C, hipStream_t st);

// narrow weight-gradient stream dW[C][16] = act(wide)^T narrow, or its transpose (gemm_tokred.hip; internal): 0 = handled, 1 = shape not covered
int bf_gemm_slabs(int dtype, int M, int N, int K, const bf_operand* A, const bf_operand* B, float* out, long ldc, int accumulate, int splitk,
                  float* ws, long ws_floats, hipStream_t st);      // gemm.hip: split-K into per-slice images + ordered sum (no float atomics)
int bf_tokred_narrow(int dtype, int C, int64_t P, const void* wide, const void* narrow, float* out, int ldo, int accumulate, int transposed,
                     const float* sc, const float* sh, int64_t rows_per_frame, float* ws, int64_t ws_floats, hipStream_t st);
// frame-pair data-gradient GEMM with a second, row-scaled output (gemm_frame.hip; internal): 0 = handled, 1 = shape not covered
int bf_gemm_pair_scaled(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, void* out2, const float* rowfac,
                        int rows_per_group, hipStream_t st);

// Experiment knobs (tile-shape sweeps, ablations, A/B switches) exist ONLY in builds made with `make EXTRA=-DBF_EXPERIMENTS`:
// a shipped library reads no environment variable on any launch path and runs the measured defaults.
#ifdef BF_EXPERIMENTS
#include <stdlib.h>
inline int bf_knob(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int bf_knob(const char*, int dflt) { return dflt; }
#endif

// "once per device" flag for hipFuncSetAttribute (the dynamic-LDS limit is a per-device property of a kernel: a process that drives a second
// device must set it there too)
struct BfPerDeviceOnce {
    bool done[64] = {};
    bool& flag() { int dev = 0; (void)hipGetDevice(&dev); return done[(dev >= 0 && dev < 64) ? dev : 0]; }
};
// host-side state that links one library call to the next (chain hints, pending work, alternating buffers): one instance per device, so that
// a process driving several devices (one stream of stage calls each) does not cross them
template <class T> struct BfPerDevice {
    T tab[64] = {};
    T& get() { int dev = 0; (void)hipGetDevice(&dev); return tab[(dev >= 0 && dev < 64) ? dev : 0]; }
};
int bf_fail(hipError_t e, const char* file, int line);
int bf_fail_msg(const char* msg, const char* file, int line);
int bf_decline(const char* msg);      // returns 1 (shape not covered) and records why

// ---------------------------------------------------------------- scalar conversions
__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }

// 16-byte chunks: 4 floats or 8 bf16
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int N = 4;
    float4 raw;
    __device__ __forceinline__ void load(const float* p) { raw = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = raw; }
    __device__ __forceinline__ void zero() { raw = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ float get(int i) const { return i == 0 ? raw.x : i == 1 ? raw.y : i == 2 ? raw.z : raw.w; }
    __device__ __forceinline__ void set(int i, float v) { if (i == 0) raw.x = v; else if (i == 1) raw.y = v; else if (i == 2) raw.z = v; else raw.w = v; }
};
template <> struct Chunk<bf16> {
    static constexpr int N = 8;
    bf16x8 raw;
    __device__ __forceinline__ void load(const bf16* p) { raw = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ void store(bf16* p) const { *reinterpret_cast<bf16x8*>(p) = raw; }
    __device__ __forceinline__ void zero() { for (int i = 0; i < 8; ++i) raw[i] = (bf16)0.f; }
    __device__ __forceinline__ float get(int i) const { return (float)raw[i]; }
    __device__ __forceinline__ void set(int i, float v) { raw[i] = (bf16)v; }
};

// ---------------------------------------------------------------- math
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level): one exp + one rcp instead of
// the library erff's branchy polynomial -- the exact-GELU of the reference (nn.GELU()) to fp32 accuracy.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
// d/dx gelu(x) = Phi(x) + x * phi(x).  The exponential inside erf_fast(x / sqrt 2) is exp(-x^2 / 2), i.e. phi's: evaluated once.
__device__ __forceinline__ float dgelu_f(float x) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = __expf(-0.5f * x * x);
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
    return cdf + x * (0.39894228040143267794f * e);
}

// bf16-storage kernels: Phi(x) and gelu'(x) as odd polynomials around 1/2 on |x| <= 4 (clamped beyond: Phi(4) = 1 - 3.2e-5), no
// transcendentals -- 11-13 plain VALU operations instead of ~20 with a reciprocal and an exponential.  |error| <= 2.5e-5 (Phi) and
// 8e-5 (gelu') in fp32 evaluation, i.e. a fiftieth of the bf16 rounding of an O(1) activation; least-squares Chebyshev fits
// (weight |x|) of the exact functions.  The exact-fp32 parity mode keeps gelu_f / dgelu_f.  The fc1 epilogue alone evaluates
// 57 M of these per launch: with the exact form its VALU time exceeded the GEMM's MFMA time.
__device__ __forceinline__ float phi_fast(float x) {
    const float xc = fminf(fmaxf(x, -4.0f), 4.0f), u = xc * xc;
    float r = -1.520480094e-09f;
    r = r * u + 1.180964698e-07f; r = r * u - 4.014221545e-06f; r = r * u + 7.960997465e-05f; r = r * u - 1.041295800e-03f;
    r = r * u + 9.641715296e-03f; r = r * u - 6.614117438e-02f; r = r * u + 3.988329119e-01f;
    return 0.5f + xc * r;
}
__device__ __forceinline__ float gelu_fast(float x) { return x * phi_fast(x); }
__device__ __forceinline__ float dgelu_fast(float x) {
    const float xc = fminf(fmaxf(x, -4.0f), 4.0f), u = xc * xc;
    float r = 9.387459194e-10f;
    r = r * u - 7.941240515e-08f; r = r * u + 2.950722870e-06f; r = r * u - 6.380590451e-05f; r = r * u + 8.975979855e-04f;
    r = r * u - 8.669717964e-03f; r = r * u + 5.833777581e-02f; r = r * u - 2.646917422e-01f; r = r * u + 7.975648121e-01f;
    return 0.5f + xc * r;
}

// storage-type dispatch: bf16 kernels take the polynomial forms, the exact-fp32 parity mode the exact ones
template <typename T> __device__ __forceinline__ float gelu_t(float x) { if constexpr (sizeof(T) == 2) return gelu_fast(x); else return gelu_f(x); }
template <typename T> __device__ __forceinline__ float dgelu_t(float x) { if constexpr (sizeof(T) == 2) return dgelu_fast(x); else return dgelu_f(x); }

// ---------------------------------------------------------------- wave / block reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int bf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t bf_esize(int dtype) { return dtype == BF_DTYPE_BF16 ? 2 : 4; }
