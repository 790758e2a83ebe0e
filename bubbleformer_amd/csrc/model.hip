// Stage-level orchestration: each FiLMAViT stage (patch embed + FiLM, temporal block, axial block,
// debed + loss) as one stream-ordered chain of the kernels in gemm/norm/attn/patch.hip.
// No allocation, no synchronisation: activations that the backward needs live in a caller-owned
// "saved" record per stage, transients in a caller-owned scratch arena.
//
// Algebra used to avoid extra passes (all exact in real arithmetic):
//  * InstanceNorm is applied in the consumer GEMM's operand prologue as a per-(frame, channel) affine;
//    only its statistics are a separate (two-pass, fp32) kernel.
//  * layer scale / residual / feature scaling are folded into the out-projection epilogue
//       out = x + alpha[n] * (on @ W^T)[m, n] + beta[n].
//    The spatial mean over (h, w) that feature scaling needs is data independent: InstanceNorm output has
//    per-channel mean exactly norm2.bias, so mean_hw(y)[n] = W[n, :] . norm2.bias + bias[n]  =: mc[n].
//  * parameter gradients of those folds come from G = dout^T @ on (one split-K GEMM) instead of a saved y:
//       dW = alpha * G (+ dmc x norm2.bias), dalpha[n] = <W[n, :], G[n, :]>, dbeta = colsum(dout).
#include <functional>
#include <vector>
#include "bf_common.h"
#include "param_reduce.h"
#include <stdlib.h>
#include <string.h>

bool bf_attn_raw_modes(int dtype, int d);
int bf_gemm_tokred_deferred(int dtype, int Nout, int Kin, int64_t M, const void* dy, int64_t ldy, const void* x, int64_t ldx, float* out,
                            int accumulate, float* colsum, float* ws, int64_t ws_floats, hipStream_t stream);
int bf_gemm_tokred_flush(hipStream_t st);
bool bf_gemm_tokred_pending();
const float* bf_gemm_tokred_pending_out();
int bf_gemm_inbwd_frames_scaled(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* x,
                                const void* add, void* out, int S, const float* mean, const float* rstd, const float* w, float* ws,
                                const float* fscale, int fdiv, void* out_s, const float* f_s, int fdiv_s, hipStream_t stream);

namespace {

struct D {
    int dtype, B, T, h, w, E, heads, attn_scale, feat_scale, patch, cin, cout, nfluid;
    long N, F, S;
    int d, nst;
    size_t es;
};
int get_dims(const bf_dims* s, D* o) {
    if (!s) return bf_fail_msg("dims: null", __FILE__, __LINE__);
    o->dtype = s->dtype; o->B = s->B; o->T = s->T; o->h = s->h; o->w = s->w; o->E = s->E; o->heads = s->heads;
    o->attn_scale = s->attn_scale; o->feat_scale = s->feat_scale; o->patch = s->patch; o->cin = s->cin; o->cout = s->cout;
    o->nfluid = s->nfluid;
    if (o->B < 1 || o->T < 1 || o->h < 1 || o->w < 1 || o->E < 8 || o->E > 1024 || o->heads < 1 || o->E % o->heads)
        return bf_fail_msg("dims: bad sizes", __FILE__, __LINE__);
    if (o->dtype != BF_DTYPE_F32 && o->dtype != BF_DTYPE_BF16) return bf_fail_msg("dims: bad dtype", __FILE__, __LINE__);
    o->F = (long)o->B * o->T; o->S = (long)o->h * o->w; o->N = o->F * o->S; o->d = o->E / o->heads;
    o->es = bf_esize(o->dtype);
    const int ch = o->dtype == BF_DTYPE_BF16 ? 8 : 4;
    if (o->E % ch || o->d % ch) return bf_fail_msg("dims: E and head dim must be multiples of the 16-byte chunk", __FILE__, __LINE__);
    if (o->T > 32 || o->h > 32 || o->w > 32) return bf_fail_msg("dims: attention axes are limited to 32 tokens", __FILE__, __LINE__);
    o->nst = 0;
    if (o->patch > 0) {
        int p = o->patch;
        while (p > 1) { if (p & 1) return bf_fail_msg("dims: patch must be a power of two", __FILE__, __LINE__); p >>= 1; o->nst++; }
        if (o->nst < 1 || o->nst > BF_MAX_STAGES) return bf_fail_msg("dims: patch size out of range", __FILE__, __LINE__);
        if (o->nst > 1 && (o->E / 4) % ch) return bf_fail_msg("dims: E/4 must be a multiple of the 16-byte chunk", __FILE__, __LINE__);
    }
    return 0;
}

// bump allocator over a caller-owned buffer, 256-byte aligned pieces
struct Arena {
    char* base; size_t off;
    explicit Arena(void* p) : base((char*)p), off(0) {}
    void* take(size_t bytes) { void* r = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return r; }
    float* f32(size_t n) { return (float*)take(n * 4); }
};

bf_operand op_plain(const void* p, long ld, int layout) {
    bf_operand o; memset(&o, 0, sizeof(o)); o.p = p; o.ld = ld; o.layout = layout; return o;
}
void op_affine(bf_operand& o, int pro, const float* sc, const float* sh, long rpf, int nch) {
    o.pro = pro; o.sc = sc; o.sh = sh; o.rows_per_frame = (int)rpf; o.nch = nch;
}
// rows are output-resolution pixels (gw x gh grid per frame) of a k2s2 patch over a [.., 2gh, 2gw, C] image
void op_gather(bf_operand& o, int gw, int gh, int C) { o.gw = gw; o.gh = gh; o.gc = C; o.seglen = 2 * C; o.segstride = 2L * gw * C; }
bf_epilogue epi_store(void* c, long ldc) { bf_epilogue e; memset(&e, 0, sizeof(e)); e.c = c; e.ldc = ldc; e.out_mode = BF_OUT_STORE; return e; }
bf_epilogue epi_atomic(float* c, long ldc) { bf_epilogue e = epi_store(c, ldc); e.out_mode = BF_OUT_ATOMIC_F32; return e; }
void epi_scatter(bf_epilogue& e, int gw, int gh, int C) { e.gw = gw; e.gh = gh; e.gc = C; e.seglen = 2 * C; e.segstride = 2L * gw * C; }

int splitk_for(int M, int N, long K) {
    // the split-K partials are added with fp32 atomics, so splits cost write traffic in proportion to the output size.  An isolated
    // sweep prefers ~64/sqrt(tiles) slices, but inside the full step that loses 5% (A/B on the bench: 351 vs
    // 369 samples/s) to the rule below.
    const long tiles = (long)bf_cdiv(M, 128) * bf_cdiv(N, 128);
    static const long target = bf_knob("BF_SPLITK_TARGET", 256);
    long s = (target + tiles / 2) / tiles;   // ~one wave of tiles over 256 CUs; more slices lose to atomic traffic in the full step
    const long kt = (K + 63) / 64;
    if (s > kt / 4) s = kt / 4;                         // at least 4 K-steps per slice
    if (s < 1) s = 1;
    return (int)s;
}

#define TRY(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)
#define ZERO(ptr, bytes) do { hipError_t e__ = hipMemsetAsync((ptr), 0, (bytes), st); if (e__ != hipSuccess) return bf_fail(e__, __FILE__, __LINE__); } while (0)
#define ZERO_ON(stream, ptr, bytes) do { hipError_t e__ = hipMemsetAsync((ptr), 0, (bytes), (stream)); if (e__ != hipSuccess) return bf_fail(e__, __FILE__, __LINE__); } while (0)
#define HIP_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return bf_fail(e__, __FILE__, __LINE__); } while (0)

// ------------------------------------------------------------------------------------------------ side stream
// The backward of every linear layer has two independent GEMMs over the same dy: the data gradient (on the critical path)
// and the weight gradient (needed only by the optimizer).  Alone, each runs ~one wave of tiles with its load / MFMA / epilogue
// phases in lock step across the chip; issued on two HIP streams they interleave and fill each other's bubbles.  The library
// owns one extra stream per device; a stage forks work onto it with an event and joins it before it returns, so the caller
// still sees plain stream-ordered semantics on ITS stream (and the fork/join pattern is hipGraph-capturable).
// BF_SIDE_STREAM=0 runs everything on the caller's stream.  The launch profiler times each kernel with events on the stream it was
// launched on, so its per-kernel durations are the contended ones of the real schedule (they agree with a rocprofv3 trace).
struct SideStream { hipStream_t st = nullptr; hipEvent_t fork = nullptr, join = nullptr, tail[2] = {nullptr, nullptr}; bool failed = false; bool pending[2] = {false, false}; };
// bf_side_defer(1): a trunk stage's backward does not join its weight-gradient work before it returns (every fork / join is a
// barrier packet that costs the caller's stream ~6 us, and the wait itself idles it when the side stream is behind).  Consecutive
// stages alternate between two scratch sets and a stage first waits for the side work of the stage before the previous one (the
// last user of its set), so nothing the side stream still reads is overwritten and the side stream may lag by a whole stage.
// Every other stage entry point joins everything at its start, as does bf_side_join().  Off by default: plain stream-ordered
// semantics (every stage joins before it returns).
bool g_side_defer = false;
SideStream* side_stream() {
    static SideStream tab[64];
    static const bool enabled = bf_knob("BF_SIDE_STREAM", 1) != 0;
    if (!enabled) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    SideStream& s = tab[dev];
    if (!s.st && !s.failed) {
        // lowest priority: the caller's stream is the one a consumer waits on (round 2: +0.3-0.5 % against normal priority; round 3: no
        // difference between lowest, normal and highest)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = least urgent (numerically greatest)
        // the two streams are on one device: the events need no system-scope fence (an L2 write-back + invalidate at every fork / join)
        const unsigned ef = hipEventDisableTiming | hipEventDisableSystemFence;
        if (hipStreamCreateWithPriority(&s.st, hipStreamNonBlocking, lo) != hipSuccess || hipEventCreateWithFlags(&s.fork, ef) != hipSuccess ||
            hipEventCreateWithFlags(&s.join, ef) != hipSuccess || hipEventCreateWithFlags(&s.tail[0], ef) != hipSuccess ||
            hipEventCreateWithFlags(&s.tail[1], ef) != hipSuccess) { s.failed = true; s.st = nullptr; }
    }
    return s.st ? &s : nullptr;
}
// the previous stage's deferred tail (if any) is ordered before what `main` is given next
int flush_pending_reduce(hipStream_t st);                    // (a spatial stage's parameter reductions waiting for the temporal stage behind it)
int side_join_pending(hipStream_t main, int set = -1) {      // set: 0 / 1 = the work that reads that scratch set, -1 = everything
    if (set < 0) { const int rc = flush_pending_reduce(main); if (rc) return rc; }
    SideStream* s = side_stream();
    if (set < 0 && bf_gemm_tokred_pending()) {               // the last weight-gradient GEMM's slab sum is still pending (bf_gemm_tokred_deferred): run it now
        hipStream_t ws_st = s ? s->st : main;
        const int rc = bf_gemm_tokred_flush(ws_st);
        if (rc) return rc;
        if (s) { HIP_TRY(hipEventRecord(s->join, s->st)); HIP_TRY(hipStreamWaitEvent(main, s->join, 0)); }
    }
    if (!s) return 0;
    for (int i = 0; i < 2; ++i)
        if ((set < 0 || set == i) && s->pending[i]) {
            HIP_TRY(hipStreamWaitEvent(main, s->tail[i], 0));
            s->pending[i] = false;
        }
    return 0;
}
struct Fork {
    hipStream_t main; SideStream* s; bool used = false; bool deferred; int set;
    std::vector<std::function<int(hipStream_t)>> jobs;          // deferred mode: the stage's side work, launched by flush()
    std::vector<std::function<int(hipStream_t)>> late;          // ... and what must follow the stage's LAST weight-gradient launch (see run_late)
    hipStream_t last_side = nullptr;
    explicit Fork(hipStream_t m, bool may_defer = false, int scratch_set = 0) : main(m), s(side_stream()), set(scratch_set) {
        deferred = may_defer && g_side_defer && s != nullptr;
    }
    // stream for work that depends only on what has been issued on `main` so far
    int begin(hipStream_t* out) {
        *out = main;
        if (!s) return 0;
        HIP_TRY(hipEventRecord(s->fork, main));
        HIP_TRY(hipStreamWaitEvent(s->st, s->fork, 0));
        used = true;
        *out = s->st;
        return 0;
    }
    // side work: job(stream) enqueues it.  Eager mode forks here; deferred mode keeps it for flush().
    template <class F> int run(F&& job) {
        hipStream_t ss;
        const int rc = begin(&ss);
        last_side = ss;
        return rc ? rc : job(ss);
    }
    // work that reads the result of a token-reduction GEMM whose slab sum rides in the NEXT such launch (bf_gemm_tokred_deferred): queued here,
    // it runs on the side stream after the stage's remaining weight-gradient launches (join() / flush())
    template <class F> void run_late(F&& job) { late.push_back(job); }
    int drain_late() {
        if (late.empty()) return 0;
        hipStream_t ss = last_side ? last_side : main;
        for (auto& j : late) { const int rc = j(ss); if (rc) return rc; }
        late.clear();
        return 0;
    }
    // deferred mode: one fork for everything collected so far
    int flush() {
        if (!deferred || jobs.empty()) return 0;
        int rc;
        hipStream_t ss;
        if ((rc = begin(&ss))) return rc;
        for (auto& j : jobs) if ((rc = j(ss))) return rc;
        jobs.clear();
        return 0;
    }
    // everything forked so far is ordered before what `main` is given next (deferred mode: before the next stage's fork point)
    int join() {
        if (deferred) {
            int rc = flush();
            if (rc) return rc;
            if ((rc = drain_late())) return rc;
            if (used) { HIP_TRY(hipEventRecord(s->tail[set], s->st)); s->pending[set] = true; used = false; }
            return 0;
        }
        {   // plain stream-ordered semantics: nothing of the stage may stay pending
            int rc = drain_late();
            if (rc) return rc;
            if (bf_gemm_tokred_pending() && (rc = bf_gemm_tokred_flush(last_side ? last_side : main))) return rc;
        }
        if (!s || !used) return 0;
        HIP_TRY(hipEventRecord(s->join, s->st));
        HIP_TRY(hipStreamWaitEvent(main, s->join, 0));
        used = false;
        return 0;
    }
};

// ------------------------------------------------------------------------------------------------ small param kernels
// out-projection fold.  mc[n] = <W[n,:], nb> + bias[n]; alpha = gamma*(1+hi); beta = gamma*(bias*(1+hi) + mc*(lo-hi))
struct PrepArgs { const float *W, *bias, *nb, *gamma, *lo, *hi; float *alpha, *beta, *mc; int E; void* wscaled; int dtype;
                  const float *tab_m, *tab_v; float* tab_out; int tab_F;
                  const float* tr_src; void* tr_dst; int tr_R, tr_C; };     // optional transposed bf16 copy tr_dst[c][r] = tr_src[r][c] (fc2 weight: the data gradient's K-contiguous operand)     // optional stochastic-depth table tab_out[f][c] = tab_m[f] * tab_v[c] (E columns)      // wscaled[n][k] = alpha[n] * W[n][k] (compute dtype): the data-gradient GEMM's weight
__device__ __forceinline__ void outproj_prep_row(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ nb,
                                                 const float* __restrict__ gamma, const float* __restrict__ lo, const float* __restrict__ hi,
                                                 float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ mc, int E, int n,
                                                 void* __restrict__ wscaled, int dtype) {
    __shared__ float red[4];
    __shared__ float s_alpha;
    float acc = 0.f;
    if (lo) for (int k = threadIdx.x; k < E; k += blockDim.x) acc += W[(long)n * E + k] * nb[k];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = bias[n];
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m += red[i];
        const float l = lo ? lo[n] : 0.f, h = hi ? hi[n] : 0.f, g = gamma[n];
        alpha[n] = g * (1.f + h);
        beta[n] = g * (bias[n] * (1.f + h) + (lo ? m * (l - h) : 0.f));
        mc[n] = m;
        s_alpha = g * (1.f + h);
    }
    if (wscaled) {
        __syncthreads();
        const float al = s_alpha;
        for (int k = threadIdx.x; k < E; k += blockDim.x) {
            const float v = al * W[(long)n * E + k];
            if (dtype == BF_DTYPE_BF16) reinterpret_cast<bf16*>(wscaled)[(long)n * E + k] = (bf16)v;
            else reinterpret_cast<float*>(wscaled)[(long)n * E + k] = v;
        }
    }
}
__global__ void __launch_bounds__(256) outproj_prep_kernel(PrepArgs a) {
    outproj_prep_row(a.W, a.bias, a.nb, a.gamma, a.lo, a.hi, a.alpha, a.beta, a.mc, a.E, blockIdx.x, a.wscaled, a.dtype);
}
// parameter gradients of the fold (see header comment).  grid = E rows.
__global__ void outproj_finalize_kernel(const float* __restrict__ G, const float* __restrict__ csum, const float* __restrict__ W,
                                        const float* __restrict__ bias, const float* __restrict__ nb, const float* __restrict__ gamma,
                                        const float* __restrict__ lo, const float* __restrict__ hi, const float* __restrict__ mc,
                                        float* __restrict__ dW, float* __restrict__ dbias, float* __restrict__ dnb, float* __restrict__ dgamma,
                                        float* __restrict__ dlo, float* __restrict__ dhi, int E) {
    __shared__ float red[4];
    __shared__ float s_dmc, s_alpha;
    if ((int)blockIdx.x >= E) {      // extra workgroups (feature scaling only): dnb[k] += sum_n dmc[n] * W[n][k], 16 columns each -- dmc[n] =
        // csum[n] * gamma[n] * (lo[n] - hi[n]) needs nothing the row workgroups compute, and one writer per column replaces E x E float
        // atomics on E addresses (the launch took 14 us with them)
        // 16 columns x 16 row groups per workgroup (row group q: rows q, q + 16, ...): with 64 columns x 4 row groups a thread walked 96 rows in
        // twelve dependent batches of eight loads, ~24 us on six workgroups while the rest of the launch took 5 -- the whole launch waited
        // for them (26 us, 24 times a step on the side queue).  Now three batches.
        __shared__ float part[16][16];
        __shared__ float coef[1024];           // dmc[n] (E <= 1024: host-checked)
        for (int n = threadIdx.x; n < E; n += blockDim.x) coef[n] = csum[n] * gamma[n] * (lo[n] - hi[n]);
        __syncthreads();
        const int c = threadIdx.x & 15, k = ((int)blockIdx.x - E) * 16 + c, q = threadIdx.x >> 4;
        float a = 0.f;
        if (k < E) {
            int n = q;
            for (; n + 112 < E; n += 128) {      // eight rows of W in flight per thread (a load-then-add loop pays a round trip per row)
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = W[(long)(n + 16 * u) * E + k];
#pragma unroll
                for (int u = 0; u < 8; ++u) a = fmaf(coef[n + 16 * u], wv[u], a);
            }
            for (; n < E; n += 16) a = fmaf(coef[n], W[(long)n * E + k], a);
        }
        part[q][c] = a;
        __syncthreads();
        // one atomic per column: the caller's stream adds norm2's own bias gradient to the same addresses at the same time (the stage's
        // InReduceJob) -- two addends on a zeroed slot give the same bits in either order, a plain += could lose one of them
        if (q == 0 && k < E) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += part[i][c];      // fixed order
            atomicAdd(dnb + k, t);
        }
        return;
    }
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int k = threadIdx.x; k < E; k += blockDim.x) acc += W[(long)n * E + k] * G[(long)n * E + k];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float dalpha = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) dalpha += red[i];
        const float dbeta = csum[n];
        const float l = lo ? lo[n] : 0.f, h = hi ? hi[n] : 0.f, g = gamma[n], b = bias[n], m = mc[n];
        const float mterm = lo ? m * (l - h) : 0.f;
        dgamma[n] += dalpha * (1.f + h) + dbeta * (b * (1.f + h) + mterm);
        float dmc = 0.f;
        if (lo) {
            dhi[n] += dalpha * g + dbeta * g * (b - m);
            dlo[n] += dbeta * g * m;
            dmc = dbeta * g * (l - h);
        }
        dbias[n] += dbeta * g * (1.f + h) + dmc;
        s_dmc = dmc;
        s_alpha = g * (1.f + h);
    }
    __syncthreads();
    const float dmc = s_dmc, alpha = s_alpha;
    for (int k = threadIdx.x; k < E; k += blockDim.x) {
        float v = alpha * G[(long)n * E + k];
        if (lo) v += dmc * nb[k];
        dW[(long)n * E + k] += v;
    }
}
// stochastic-depth helper: out[f][c] = m[f / fdiv] * (v ? v[c] : 1)
__global__ void frame_table_kernel(const float* __restrict__ m, int fdiv, const float* __restrict__ v, float* __restrict__ out, int F, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)F * C) return;
    const int f = (int)(i / C), c = (int)(i % C);
    out[i] = m[f / fdiv] * (v ? v[c] : 1.f);
}

// dst[c][r] = (bf16) src[r][c] (R x C fp32 -> C x R bf16) in 64 x 64 tiles through LDS: coalesced 16-byte reads along c, 16-byte stores along r
// (the first form read 8 rows per thread with a stride of C floats: every 4-byte read its own cache line -- 16x the bytes of the weight).
// `tile0`, `tstride`: the calling workgroup's first tile and the tile stride (256 threads; R, C multiples of 64).
__device__ __forceinline__ void transpose_cast(const float* __restrict__ src, bf16* __restrict__ dst, int R, int C, int tile0, int tstride) {
    __shared__ float tl[64][65];
    const int tr_ = (R + 63) / 64, tc_ = (C + 63) / 64;      // (R, C are multiples of 8: whole float4 reads, whole 8-row stores; edge tiles are partial)
    const int tid = threadIdx.x;
    for (int t = tile0; t < tr_ * tc_; t += tstride) {
        const int r0 = (t / tc_) * 64, c0 = (t % tc_) * 64;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {                  // 64 rows x 16 float4
            const int i = tid + 256 * k, r = i >> 4, c4 = (i & 15) * 4;
            if (r0 + r < R && c0 + c4 < C) {
                const float4 v = *reinterpret_cast<const float4*>(src + (long)(r0 + r) * C + c0 + c4);
                tl[r][c4] = v.x; tl[r][c4 + 1] = v.y; tl[r][c4 + 2] = v.z; tl[r][c4 + 3] = v.w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {                  // 64 columns x 8 groups of 8 rows
            const int i = tid + 256 * k, c = i >> 3, r8 = (i & 7) * 8;
            if (c0 + c < C && r0 + r8 < R) {
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)tl[r8 + q][c];
                *reinterpret_cast<bf16x8*>(dst + (long)(c0 + c) * R + r0 + r8) = o;
            }
        }
    }
}
// up to four plain fp32 -> bf16 weight casts in ONE launch (a stage's projection weights)
struct Cast4 { const float* src[4]; bf16* dst[4]; long n[4]; };
__global__ void __launch_bounds__(256) cast4_kernel(Cast4 j) {
    const int w = blockIdx.y;
    const long n4 = j.n[w] / 4;
    const float4* s4 = reinterpret_cast<const float4*>(j.src[w]);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = s4[i];
        const bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
        *reinterpret_cast<bf16x4*>(j.dst[w] + 4 * i) = o;
    }
}
// the same casts plus the out-projection fold (grid row cnt, one workgroup per output channel): a stage's parameter-only work in ONE launch
__global__ void __launch_bounds__(256) stage_prep_kernel(Cast4 j, int cnt, PrepArgs a) {
    if ((int)blockIdx.y == cnt + 2) {          // transposed copy
        if (a.tr_dst) transpose_cast(a.tr_src, (bf16*)a.tr_dst, a.tr_R, a.tr_C, (int)blockIdx.x, (int)gridDim.x);
        return;
    }
    if ((int)blockIdx.y == cnt + 1) {
        if (!a.tab_out) return;          // the stage's stochastic-depth table (frame_table_kernel's work, no launch of its own)
        const long n = (long)a.tab_F * a.E;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a.tab_out[i] = a.tab_m[i / a.E] * a.tab_v[i % a.E];
        return;
    }
    if ((int)blockIdx.y == cnt) {
        if ((int)blockIdx.x < a.E) outproj_prep_row(a.W, a.bias, a.nb, a.gamma, a.lo, a.hi, a.alpha, a.beta, a.mc, a.E, blockIdx.x, a.wscaled, a.dtype);
        return;
    }
    const int w = blockIdx.y;
    const long n4 = j.n[w] / 4;
    const float4* s4 = reinterpret_cast<const float4*>(j.src[w]);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = s4[i];
        const bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
        *reinterpret_cast<bf16x4*>(j.dst[w] + 4 * i) = o;
    }
}
// ... and the same for up to PREP_BATCH stages in one launch (blockIdx.z = stage): bf_prep_stages
constexpr int PREP_BATCH = 12;
struct PrepBatch { Cast4 j[PREP_BATCH]; PrepArgs a[PREP_BATCH]; int cnt[PREP_BATCH]; };
__global__ void __launch_bounds__(256) stage_prep_multi_kernel(PrepBatch b) {
    const int z = blockIdx.z, cnt = b.cnt[z];
    const Cast4& j = b.j[z];
    const PrepArgs& a = b.a[z];
    if ((int)blockIdx.y == cnt + 2) {
        if (a.tr_dst) transpose_cast(a.tr_src, (bf16*)a.tr_dst, a.tr_R, a.tr_C, (int)blockIdx.x, (int)gridDim.x);
        return;
    }
    if ((int)blockIdx.y == cnt + 1) {
        if (!a.tab_out) return;
        const long n = (long)a.tab_F * a.E;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a.tab_out[i] = a.tab_m[i / a.E] * a.tab_v[i % a.E];
        return;
    }
    if ((int)blockIdx.y > cnt + 2) return;
    if ((int)blockIdx.y == cnt) {
        if ((int)blockIdx.x < a.E) outproj_prep_row(a.W, a.bias, a.nb, a.gamma, a.lo, a.hi, a.alpha, a.beta, a.mc, a.E, blockIdx.x, a.wscaled, a.dtype);
        return;
    }
    const int w = blockIdx.y;
    const long n4 = j.n[w] / 4;
    const float4* s4 = reinterpret_cast<const float4*>(j.src[w]);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = s4[i];
        const bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
        *reinterpret_cast<bf16x4*>(j.dst[w] + 4 * i) = o;
    }
}
// set by bf_stage_prepared(1): the next trunk stage forward finds its weights prepared in `saved` (bf_prep_stages) and skips its own launch
bool g_stage_prepared = false;
// weights[i] (fp32, count n[i], multiples of 4) -> compute-dtype operands: one cast launch in bf16 mode, aliases in f32 mode;
// `prep` (optional): the out-projection fold of the stage, computed in the same launch
int wviews(const D& d, int cnt, const float* const* src, void* const* dst, const long* n, const void** out, hipStream_t st, const PrepArgs* prep = nullptr) {
    if (d.dtype == BF_DTYPE_F32) {
        for (int i = 0; i < cnt; ++i) out[i] = src[i];
        if (prep) { hipLaunchKernelGGL(outproj_prep_kernel, dim3(prep->E), dim3(256), 0, st, *prep); BF_CHECK_LAUNCH(); }
        return 0;
    }
    Cast4 j;
    for (int i = 0; i < 4; ++i) { j.src[i] = src[i < cnt ? i : 0]; j.dst[i] = (bf16*)dst[i < cnt ? i : 0]; j.n[i] = i < cnt ? n[i] : 0; out[i < cnt ? i : 0] = dst[i < cnt ? i : 0]; }
    for (int i = 0; i < cnt; ++i) out[i] = dst[i];
    if (prep) hipLaunchKernelGGL(stage_prep_kernel, dim3(std::max(64, prep->E), cnt + (prep->tr_dst ? 3 : prep->tab_out ? 2 : 1)), dim3(256), 0, st, j, cnt, *prep);
    else hipLaunchKernelGGL(cast4_kernel, dim3(64, cnt), dim3(256), 0, st, j);
    BF_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------ saved-record layouts
struct TemporalSaved {
    float *mean1, *rstd1, *sc1, *sh1, *mean2, *rstd2, *sc2, *sh2, *alpha, *beta, *mc;
    void *qkv, *o, *xn, *on, *win_c, *wout_c, *wout_s;      // wout_s = diag(alpha) W_out (data-gradient operand); xn / on: InstanceNorm'd block input / attention output (GEMM operands, fwd and dW)
    void* wout_t;                                           // W_out^T [in][out]: the frame-pair forward kernel's operand (bf_gemm_fwd_frames)
    size_t bytes;
    TemporalSaved(const D& d, void* base) {
        Arena a(base);
        const size_t fe = (size_t)d.F * d.E;
        mean1 = a.f32(fe); rstd1 = a.f32(fe); sc1 = a.f32(fe); sh1 = a.f32(fe);
        mean2 = a.f32(fe); rstd2 = a.f32(fe); sc2 = a.f32(fe); sh2 = a.f32(fe);
        alpha = a.f32(d.E); beta = a.f32(d.E); mc = a.f32(d.E);
        qkv = a.take((size_t)d.N * 3 * d.E * d.es);
        o = a.take((size_t)d.N * d.E * d.es);
        xn = a.take((size_t)d.N * d.E * d.es);
        on = a.take((size_t)d.N * d.E * d.es);
        win_c = a.take((size_t)3 * d.E * d.E * d.es);
        wout_c = a.take((size_t)d.E * d.E * d.es);
        wout_s = a.take((size_t)d.E * d.E * d.es);
        wout_t = a.take((size_t)d.E * d.E * d.es);
        bytes = a.off;
    }
};
struct SpatialSaved {
    float *mean1, *rstd1, *sc1, *sh1, *mean2, *rstd2, *sc2, *sh2, *mean3, *rstd3, *sc3, *sh3, *alpha, *beta, *mc, *gtab;
    void *qkv, *o, *xn, *on, *x1, *pre, *hid, *z, *win_c, *wout_c, *wout_s, *w1_c, *w2_c, *w2t_c;      // w2t_c = fc2.weight^T [4E][E]: K-contiguous operand of the fc2 data gradient
    size_t bytes;
    SpatialSaved(const D& d, void* base) {
        Arena a(base);
        const size_t fe = (size_t)d.F * d.E;
        mean1 = a.f32(fe); rstd1 = a.f32(fe); sc1 = a.f32(fe); sh1 = a.f32(fe);
        mean2 = a.f32(fe); rstd2 = a.f32(fe); sc2 = a.f32(fe); sh2 = a.f32(fe);
        mean3 = a.f32(fe); rstd3 = a.f32(fe); sc3 = a.f32(fe); sh3 = a.f32(fe);
        alpha = a.f32(d.E); beta = a.f32(d.E); mc = a.f32(d.E); gtab = a.f32(fe);
        qkv = a.take((size_t)d.N * 3 * d.E * d.es);
        o = a.take((size_t)d.N * d.E * d.es);
        xn = a.take((size_t)d.N * d.E * d.es);
        on = a.take((size_t)d.N * d.E * d.es);
        x1 = a.take((size_t)d.N * d.E * d.es);
        pre = a.take((size_t)d.N * 4 * d.E * d.es);
        hid = a.take((size_t)d.N * 4 * d.E * d.es);
        z = a.take((size_t)d.N * d.E * d.es);
        win_c = a.take((size_t)3 * d.E * d.E * d.es);
        wout_c = a.take((size_t)d.E * d.E * d.es);
        wout_s = a.take((size_t)d.E * d.E * d.es);
        w1_c = a.take((size_t)4 * d.E * d.E * d.es);
        w2_c = a.take((size_t)4 * d.E * d.E * d.es);
        w2t_c = a.take((size_t)4 * d.E * d.E * d.es);
        bytes = a.off;
    }
};

// transient scratch (backward is the larger user)
struct Scratch {
    float *G, *csum, *zeros, *ones, *wg, *attn_ws, *attn_ws2, *in_ws, *in_ws2, *in_ws3, *in_ws4, *in_ws5;   // wg: prepared-layout weight gradient scratch; in_ws4 / 5: the chained tails' partials (bf_stage_chain_tail), alternating
    float* tokred_ws; int64_t tokred_floats;      // slabs of the token-reduction (weight-gradient) GEMM
    static constexpr long ATTN_WS_FLOATS = 1024L * (4 * 128 + 32 * 16 + 16);
    void *t1, *t3, *t4, *t1b; int64_t t1b_floats;
    void *s1, *e5, *e6, *e7;     // [N][E] each: s1 feeds side-stream GEMMs only; e5..e7 keep side-stream inputs from being recycled within a stage
    size_t bytes;
    Scratch(const D& d, void* base) {
        Arena a(base);
        const int cm = d.nst > 1 ? d.E / 4 : d.E;
        size_t wgn = (size_t)d.E * d.E;
        wgn = std::max(wgn, (size_t)4 * cm * d.E);             // conv / convT prepared weights
        wgn = std::max(wgn, (size_t)d.E * 64);
        G = a.f32((size_t)d.E * d.E);
        csum = a.f32((size_t)4 * d.E);
        zeros = a.f32((size_t)4 * d.E);
        ones = a.f32((size_t)4 * d.E);
        wg = a.f32(wgn);
        tokred_floats = bf_gemm_tokred_ws_floats(4 * d.E, d.E, d.N);
        tokred_ws = a.f32((size_t)tokred_floats);
        attn_ws = a.f32(ATTN_WS_FLOATS);
        attn_ws2 = a.f32(ATTN_WS_FLOATS);       // second axial pass: both passes' rows are reduced together at the end of the stage
        {   // InstanceNorm workspace: the trunk (S tokens x E) and every embed / debed resolution (S * 4^i tokens x E/4)
            int64_t n = bf_in_ws_floats(d.dtype, (int)d.F, (int)d.S, d.E);
            long Si = d.S;
            for (int i = 1; i < d.nst; ++i) { Si *= 4; n = std::max(n, bf_in_ws_floats(d.dtype, (int)d.F, (int)Si, cm)); }
            in_ws = a.f32((size_t)n);
            // one partials region per InstanceNorm of a block: their reductions run together at the end of the stage
            const size_t nt = (size_t)bf_in_ws_floats(d.dtype, (int)d.F, (int)d.S, d.E);
            in_ws2 = a.f32(nt); in_ws3 = a.f32(nt); in_ws4 = a.f32(nt); in_ws5 = a.f32(nt);
        }
        // activation-sized transients; embed/debed stages work at up to (patch/2)^2 * N pixels of E/4 (or cin/cout) channels
        size_t tok = (size_t)d.N * d.E;
        size_t big = tok * 4;
        if (d.patch > 1) {
            const size_t P0 = (size_t)d.N * (d.patch / 2) * (d.patch / 2);
            const int kp = ((4 * std::max(d.cin, d.cout) + 7) / 8) * 8;
            big = std::max(big, P0 * (size_t)std::max(cm, kp) * 2);   // *2: fp32 patch-major prediction
        }
        t4 = a.take(big * d.es);
        t3 = a.take(std::max(tok * 3, big / 2) * d.es);
        t1 = a.take(std::max(tok, big / 2) * d.es);
        t1b = a.take(std::max(tok, big / 2) * d.es);
        t1b_floats = (int64_t)(std::max(tok, big / 2) * d.es / 4);
        s1 = a.take(tok * d.es); e5 = a.take(tok * d.es); e6 = a.take(tok * d.es); e7 = a.take(tok * d.es);
        bytes = a.off;
    }
};

// ------------------------------------------------------------------------------------------------ stage-end parameter reductions
// All InstanceNorm / attention parameter-gradient reductions of one stage backward in ONE launch (grid z = job): nothing on the
// critical path reads them, and eight dependent ~5 us launches per block are worth ~3 % of the step.
struct ReduceJobs { int n_in = 0, n_attn = 0; InReduceJob in[6]; AttnReduceJob at[4]; };      // room for a spatial + a temporal stage (see g_pending_reduce.get())
__global__ void __launch_bounds__(64 * BF_RED_FL) stage_param_reduce_kernel(ReduceJobs J) {
    __shared__ float red[5][BF_RED_FL][64];
    const int z = blockIdx.z;
    if (z < J.n_in) {
        const InReduceJob& j = J.in[z];
        if ((int)blockIdx.x < (j.C + 63) / 64 && (int)blockIdx.y < (j.frames + j.rdiv() - 1) / j.rdiv()) in_reduce_block(j, blockIdx.x, blockIdx.y, red);
    } else {
        const AttnReduceJob& j = J.at[z - J.n_in];
        const int nvals = 4 * j.D + 32 * j.heads + j.heads;
        if ((int)blockIdx.x < (nvals + 63) / 64 && blockIdx.y == 0) attn_reduce_block(j, blockIdx.x, 0, 1, red);
    }
}
int launch_reduce_jobs(ReduceJobs& J, hipStream_t st) {
    int k = 0;                                   // drop attention jobs without rows (fp32 / generic path accumulated directly)
    for (int i = 0; i < J.n_attn; ++i) if (J.at[i].rows > 0) J.at[k++] = J.at[i];
    J.n_attn = k;
    if (J.n_in + J.n_attn == 0) return 0;
    int gx = 1, gy = 1;
    for (int i = 0; i < J.n_in; ++i) { gx = std::max(gx, bf_cdiv(J.in[i].C, 64)); gy = std::max(gy, bf_cdiv(J.in[i].frames, J.in[i].rdiv())); }
    for (int i = 0; i < J.n_attn; ++i) { gx = std::max(gx, bf_cdiv(4 * J.at[i].D + 33 * J.at[i].heads, 64)); }
    hipLaunchKernelGGL(stage_param_reduce_kernel, dim3(gx, gy, J.n_in + J.n_attn), dim3(64 * BF_RED_FL), 0, st, J);
    BF_CHECK_LAUNCH();
    return 0;
}
// Deferred mode (bf_side_defer): the spatial stage's reductions wait for the temporal stage's backward that follows it and ride in ITS launch
// (one launch per block pair instead of two: 12 launches less on the caller's queue per step).  The two stages use different scratch sets, so
// the spatial stage's partial sums are intact until the next spatial stage, which flushes a leftover first -- as does every full join.
BfPerDevice<ReduceJobs> g_pending_reduce;
BfPerDevice<bool> g_pending_reduce_on;
int flush_pending_reduce(hipStream_t st) {
    if (!g_pending_reduce_on.get()) return 0;
    g_pending_reduce_on.get() = false;
    return launch_reduce_jobs(g_pending_reduce.get(), st);
}
int launch_with_pending(ReduceJobs& J, hipStream_t st) {
    if (g_pending_reduce_on.get()) {
        g_pending_reduce_on.get() = false;
        const ReduceJobs& P = g_pending_reduce.get();
        for (int i = 0; i < P.n_in && J.n_in < 6; ++i) J.in[J.n_in++] = P.in[i];
        for (int i = 0; i < P.n_attn && J.n_attn < 4; ++i) J.at[J.n_attn++] = P.at[i];
    }
    return launch_reduce_jobs(J, st);
}


// QKV projection + attention shared pieces -------------------------------------------------------
// The InstanceNorm'd operand is materialised by the statistics kernel itself (bf_in_stats_apply: the frame is in registers there),
// so the projection GEMMs run the prologue-free, double-buffered-LDS kernel and the weight-gradient GEMMs reuse the same tensor.
int qkv_gemm(const D& d, const void* xn, const void* w_c, const float* bias, void* qkv, hipStream_t st) {
    bf_operand A = op_plain(xn, d.E, BF_LAY_KC);
    bf_operand Bo = op_plain(w_c, d.E, BF_LAY_KC);
    bf_epilogue e = epi_store(qkv, 3L * d.E);
    e.bias = bias;
    return bf_gemm(d.dtype, (int)d.N, 3 * d.E, d.E, &A, &Bo, &e, 1, st);
}
// out = x + alpha * (affine(o) @ W^T) + beta
int outproj_gemm(const D& d, const void* on, const void* w_c, const float* alpha, const float* beta,
                 const void* resid, void* out, const float* drop, long rows_per_group, hipStream_t st) {
    bf_operand A = op_plain(on, d.E, BF_LAY_KC);
    bf_operand Bo = op_plain(w_c, d.E, BF_LAY_KC);
    bf_epilogue e = epi_store(out, d.E);
    e.colscale = alpha; e.colshift = beta; e.aux_mode = BF_AUX_ADD; e.aux = resid; e.ld_aux = d.E;
    e.rowscale = drop; e.rows_per_group = (int)rows_per_group;
    return bf_gemm(d.dtype, (int)d.N, d.E, d.E, &A, &Bo, &e, 1, st);
}
// scratch set of a trunk backward stage: deferred mode alternates, so that the side stream may still read the previous stage's set
BfPerDevice<int> g_scratch_parity;
void* bwd_scratch(const D& d, void* scratch) {
    if (!g_side_defer || !side_stream()) return scratch;
    g_scratch_parity.get() ^= 1;
    return (char*)scratch + (size_t)g_scratch_parity.get() * Scratch(d, nullptr).bytes;
}
// A data gradient dy @ W whose consumer is the backward of the InstanceNorm that fed the projection: when a frame is one
// 144-row GEMM tile the two run as ONE kernel (gemm_frame.hip); otherwise GEMM into `tmp`, then the InstanceNorm backward.
// the backward of the NEXT InstanceNorm in line, applied by the same launch where the frame-pair kernel covers it (bf_gemm_inbwd_frames_chain)
struct TailNorm { const void* z; void* dz; const float *mean, *rstd, *w, *g; int gdiv; float* ws; bool* done; };
struct InFuse { const void* x; const void* add; void* dx; const float* mean; const float* rstd; const float* w; const float* b; float* ws;
                const float* fscale = nullptr; int fdiv = 1;         // fscale: optional per-frame-group factor on dy (stochastic depth)
                const TailNorm* tail = nullptr;
                void* scaled_out = nullptr; const float* scaled_f = nullptr; int scaled_fdiv = 1; bool* scaled_done = nullptr; };      // optional second copy dx * scaled_f[frame / fdiv]
int dgrad_inbwd(const D& d, const void* dy, int Kdim, const void* w_xc, int Nout, void* tmp, const InFuse& f, hipStream_t st) {
    if (f.tail) {
        const TailNorm& t = *f.tail;
        const int crc = bf_gemm_inbwd_frames_chain(d.dtype, (int)d.N, Nout, Kdim, dy, Kdim, w_xc, Nout, f.x, f.add, f.dx, (int)d.S, f.mean, f.rstd, f.w, f.ws,
                                                   f.fscale, f.fdiv, t.z, t.dz, t.mean, t.rstd, t.w, t.g, t.gdiv, t.ws, st);
        if (crc < 0) return crc;
        if (crc == 0) { *t.done = true; return 0; }
    }
    if (f.scaled_out) {
        const int src = bf_gemm_inbwd_frames_scaled(d.dtype, (int)d.N, Nout, Kdim, dy, Kdim, w_xc, Nout, f.x, f.add, f.dx, (int)d.S, f.mean, f.rstd, f.w, f.ws,
                                                    f.fscale, f.fdiv, f.scaled_out, f.scaled_f, f.scaled_fdiv, st);
        if (src < 0) return src;
        if (src == 0) { *f.scaled_done = true; return 0; }
    }
    const int rc = bf_gemm_inbwd_frames(d.dtype, (int)d.N, Nout, Kdim, dy, Kdim, w_xc, Nout, f.x, f.add, f.dx, (int)d.S, f.mean, f.rstd, f.w, f.ws,
                                        f.fscale, f.fdiv, st);
    if (rc <= 0) return rc;
    bf_operand A = op_plain(dy, Kdim, BF_LAY_KC);
    bf_operand Bo = op_plain(w_xc, Nout, BF_LAY_XC);
    bf_epilogue e = epi_store(tmp, Nout);
    if (f.fscale) { e.rowscale = f.fscale; e.rows_per_group = (int)(d.S * f.fdiv); }
    TRY(bf_gemm(d.dtype, (int)d.N, Nout, Kdim, &A, &Bo, &e, 1, st));
    return bf_in_bwd_partials(d.dtype, tmp, f.x, f.add, f.dx, (int)d.F, (int)d.S, Nout, f.mean, f.rstd, f.w, f.b, nullptr, 1, 0, f.ws, st);
}
// backward of the folded out-projection: param grads + don = (dout * alpha) @ W
int outproj_bwd(const D& d, const Scratch& sc, const void* dout, const void* on, const void* w_s,
                const float* W, const float* bias, const float* nb, const float* gamma, const float* lo, const float* hi,
                const float* alpha, const float* mc, float* dW, float* dbias, float* dnb, float* dgamma, float* dlo, float* dhi,
                void* don, hipStream_t st, Fork& fk, const InFuse* fu = nullptr) {
    TRY(fk.run([=](hipStream_t ss) -> int {           // parameter-gradient side: G GEMM, finalize
        const void* dsrc = dout;
        // G[n][k] = sum_m dout[m][n] * on[m][k]; `on` is the normalised operand the forward saved; dbeta = colsum(dout) from the same pass
        const int trc = bf_gemm_tokred_deferred(d.dtype, d.E, d.E, d.N, dsrc, d.E, on, d.E, sc.G, 0, sc.csum, sc.tokred_ws, sc.tokred_floats, ss);
        if (trc < 0) return trc;
        if (trc == 1) {
            ZERO_ON(ss, sc.G, (size_t)((char*)sc.csum - (char*)sc.G) + (size_t)d.E * 4);     // G and csum are adjacent in the arena: one memset
            bf_operand A = op_plain(dsrc, d.E, BF_LAY_XC);
            bf_operand Bo = op_plain(on, d.E, BF_LAY_XC);
            bf_epilogue e = epi_atomic(sc.G, d.E);
            e.colsum = sc.csum;                  // dbeta = colsum(dout), fused into the same pass over dout
            TRY(bf_gemm(d.dtype, d.E, d.E, (int)d.N, &A, &Bo, &e, splitk_for(d.E, d.E, d.N), ss));
        }
        return 0;
    }));
    // G's slab sum rides in the stage's NEXT weight-gradient launch (bf_gemm_tokred_deferred): the fold's parameter gradients, which read G, follow
    // the stage's last such launch on the side stream (or flush the sum themselves when nothing followed)
    fk.run_late([=](hipStream_t ss) -> int {
        if (bf_gemm_tokred_pending_out() == sc.G) TRY(bf_gemm_tokred_flush(ss));
        hipLaunchKernelGGL(outproj_finalize_kernel, dim3(d.E + (lo ? bf_cdiv(d.E, 16) : 0)), dim3(256), 0, ss, sc.G, sc.csum, W, bias, nb, gamma, lo, hi, mc,
                           dW, dbias, dnb, dgamma, dlo, dhi, d.E);
        BF_CHECK_LAUNCH();
        return 0;
    });
    if (fu) return dgrad_inbwd(d, dout, d.E, w_s, d.E, don, *fu, st);      // ... followed by norm2's backward
    {   // don = (dout * alpha) @ W = dout @ (diag(alpha) W): the scaled weight was written by the forward's parameter prep
        bf_operand A = op_plain(dout, d.E, BF_LAY_KC);
        bf_operand Bo = op_plain(w_s, d.E, BF_LAY_XC);
        bf_epilogue e = epi_store(don, d.E);
        TRY(bf_gemm(d.dtype, (int)d.N, d.E, d.E, &A, &Bo, &e, 1, st));
    }
    return 0;
}
// backward of y = affine(x) @ W^T + b:  dW += dy^T affine(x), db += colsum(dy), dxn = dy @ W
int linear_bwd(const D& d, const Scratch& sc, const void* dy, int Nout, const void* x, int Kin, int xpro, const float* xsc, const float* xsh,
               const void* w_c, float* dW, float* db, void* dxn, const bf_epilogue* dx_epi, hipStream_t st, Fork& fk, const InFuse* fu = nullptr,
               bool last = false, const void* w_t = nullptr, void* scaled_out = nullptr, const float* rowfac = nullptr, int rpg = 1,
               bool* scaled_done = nullptr) {      // w_t: the weight transposed ([Kin][Nout], K-contiguous for the data gradient)
    TRY(fk.run([=](hipStream_t ss) -> int {      // weight gradient: side stream
        const void* xo = x;
        int pro = xpro;
        bf_operand A = op_plain(dy, Nout, BF_LAY_XC);
        if (pro == BF_PRO_AFFINE) {              // see outproj_bwd: materialise the normalised operand once
            TRY(bf_affine_apply(d.dtype, x, nullptr, xsc, xsh, sc.s1, d.N, (int)d.S, Kin, ss));
            xo = sc.s1;
            pro = BF_PRO_NONE;
        }
        bf_operand Bo = op_plain(xo, Kin, BF_LAY_XC);
        if (pro == BF_PRO_NONE) {
            const int trc = bf_gemm_tokred_deferred(d.dtype, Nout, Kin, d.N, dy, Nout, xo, Kin, dW, 1, db, sc.tokred_ws, sc.tokred_floats, ss);
            if (trc <= 0) return trc;
        }
        if (pro != BF_PRO_NONE) op_affine(Bo, pro, xsc, xsh, d.S, Kin);
        bf_epilogue e = epi_atomic(dW, Kin);
        e.colsum = db;                       // bias gradient = colsum(dy), fused into the same pass over dy
        return bf_gemm(d.dtype, Nout, Kin, (int)d.N, &A, &Bo, &e, splitk_for(Nout, Kin, d.N), ss);
    }));
    if (last) TRY(fk.flush());                   // deferred mode: the stage's one fork, ahead of its last kernel on the caller's stream
    if (fu) return dgrad_inbwd(d, dy, Nout, w_c, Kin, dxn, *fu, st);
    {
        bf_operand A = op_plain(dy, Nout, BF_LAY_KC);
        bf_operand Bo = w_t ? op_plain(w_t, Nout, BF_LAY_KC) : op_plain(w_c, Kin, BF_LAY_XC);      // K-contiguous: the streaming kernel's form
        bf_epilogue e = dx_epi ? *dx_epi : epi_store(dxn, Kin);
        e.c = dxn; e.ldc = Kin;
        if (scaled_out && rowfac && d.dtype == BF_DTYPE_BF16) {      // the row-scaled copy of dxn from the same kernel, where its shape is covered
            const int rc = bf_gemm_pair_scaled((int)d.N, Kin, Nout, &A, &Bo, &e, scaled_out, rowfac, rpg, st);
            if (rc < 0) return rc;
            if (rc == 0) { if (scaled_done) *scaled_done = true; return 0; }
        }
        TRY(bf_gemm(d.dtype, (int)d.N, Kin, Nout, &A, &Bo, &e, 1, st));
    }
    return 0;
}

}  // namespace

// ================================================================================================= sizes
extern "C" int64_t bf_temporal_saved_bytes(const bf_dims* s) { D d; if (get_dims(s, &d)) return -1; return (int64_t)TemporalSaved(d, nullptr).bytes; }
extern "C" int64_t bf_spatial_saved_bytes(const bf_dims* s) { D d; if (get_dims(s, &d)) return -1; return (int64_t)SpatialSaved(d, nullptr).bytes; }
// two scratch sets: consecutive trunk backward stages alternate between them in deferred mode (see SideStream)
// ... and, behind them, two [N][E] buffers for the pre-scaled gradient a spatial stage leaves for the temporal stage behind it (bf_stage_next_scale)
size_t dbr_bytes(const D& d) { return (((size_t)d.N * d.E * d.es) + 255) & ~(size_t)255; }
extern "C" int64_t bf_scratch_bytes(const bf_dims* s) { D d; if (get_dims(s, &d)) return -1; return 2 * (int64_t)Scratch(d, nullptr).bytes + 2 * (int64_t)dbr_bytes(d); }

// Deferred weight-gradient tails (see SideStream): opt-in for callers that join explicitly before they consume parameter gradients
extern "C" void bf_side_defer(int on) { g_side_defer = on != 0; }
extern "C" int bf_side_join(bf_stream_t s) { return side_join_pending((hipStream_t)s); }

// ================================================================================================= temporal block
// Chained stage heads: a stage that ends in `out = resid + InstanceNorm(z)` (the spatial stage's MLP branch) can leave the next stage's
// opening InstanceNorm(out) behind in the same launch (norm.hip, InChain).  bf_stage_chain_head arms it with the next temporal stage's
// parameters and saved record; the stage that consumed it remembers the record, and that stage's forward skips its own norm1.
namespace {
struct NextHead { bool armed = false; const float *w = nullptr, *b = nullptr; float *mean = nullptr, *rstd = nullptr, *sc = nullptr, *sh = nullptr; void* xn = nullptr; const void* saved = nullptr; };
BfPerDevice<NextHead> g_next_head;
BfPerDevice<const void*> g_head_done_for;
}  // namespace
// ... and the mirror image in the backward: the temporal stage's last kernel (QKV data gradient + norm1 backward) produces the output
// gradient of the spatial stage in front of it, whose backward opens with its MLP-branch InstanceNorm.  bf_stage_chain_tail arms that
// norm's backward (the spatial stage's parameters, saved record and whether its MLP branch carried stochastic depth) for the temporal
// backward called next; the spatial backward on the same record then finds dz and the partial sums in place.
namespace {
struct NextTail { bool armed = false; const bf_spatial_params* p = nullptr; const void* saved = nullptr; bool drop = false; };
BfPerDevice<NextTail> g_next_tail;
BfPerDevice<const void*> g_tail_done_for;
// where the chained tail left the norm's partial sums.  Two regions, alternating: the spatial stage's reduction of them waits for the temporal
// stage behind it (g_pending_reduce.get()), whose own chained tail -- for the NEXT spatial stage, same scratch set -- must not overwrite them
BfPerDevice<float*> g_tail_ws;
BfPerDevice<bool> g_tail_ws_flip;
BfPerDevice<const void*> g_tail_dx;     // the gradient tensor the chained tail was computed from: the spatial backward must be handed exactly that one
}  // namespace
extern "C" int bf_stage_chain_tail(const bf_spatial_params* prev_p, const void* prev_saved, int has_drop_mlp) {
    g_next_tail.get().armed = prev_p && prev_saved;
    g_next_tail.get().p = prev_p; g_next_tail.get().saved = prev_saved; g_next_tail.get().drop = has_drop_mlp != 0;
    return 0;
}
// Stochastic depth in the backward: a temporal stage multiplies its incoming gradient by its per-sample factors before anything else reads it.
// That gradient is produced by the last kernel of the spatial stage in front of it (QKV data gradient + norm1 backward): told the factors
// (bf_stage_next_scale, armed by the caller just before that spatial stage's backward), the kernel writes the scaled copy as well and the
// temporal backward finds it in place (one elementwise launch and one read of the gradient less per block).  Two buffers, alternating: the
// temporal stage's side-stream work may still read its copy while the next spatial stage writes the next one.
namespace {
struct NextScale { const float* f = nullptr; int fdiv = 1; };
BfPerDevice<NextScale> g_next_scale;
struct DbrReady { const void* dx = nullptr; const float* f = nullptr; void* buf = nullptr; };
BfPerDevice<DbrReady> g_dbr_ready;
BfPerDevice<bool> g_dbr_flip;
}  // namespace
// A forward pass starts at the embed and a backward pass at the debed: what an aborted pass left armed (a hint whose consumer never ran, a
// "done for" record whose address a later allocation may reuse) must not survive into the next one.
static void links_clear(bool forward_side) {
    if (forward_side) { g_next_head.get().armed = false; g_head_done_for.get() = nullptr; }
    g_next_tail.get().armed = false; g_tail_done_for.get() = nullptr; g_tail_dx.get() = nullptr;
    g_next_scale.get() = NextScale{}; g_dbr_ready.get() = DbrReady{};
}
extern "C" int bf_stage_next_scale(const float* factors, int fdiv) {
    g_next_scale.get().f = factors; g_next_scale.get().fdiv = fdiv > 0 ? fdiv : 1;
    return 0;
}
extern "C" int bf_stage_chain_next(const bf_dims* dims, int next_kind, const void* next_params, void* next_saved) {
    g_next_head.get().armed = false;
    if (!dims || !next_params || !next_saved) return 0;     // disarm
    BF_REQUIRE(next_kind == 0 || next_kind == 1, "bf_stage_chain_next: kind must be 0 (temporal) or 1 (spatial)");
    D d; TRY(get_dims(dims, &d));
    if (next_kind == 0) {
        const bf_temporal_params* np = (const bf_temporal_params*)next_params;
        TemporalSaved sv(d, next_saved);
        g_next_head.get().w = np->norm1_w; g_next_head.get().b = np->norm1_b;
        g_next_head.get().mean = sv.mean1; g_next_head.get().rstd = sv.rstd1; g_next_head.get().sc = sv.sc1; g_next_head.get().sh = sv.sh1; g_next_head.get().xn = sv.xn;
    } else {
        const bf_spatial_params* np = (const bf_spatial_params*)next_params;
        SpatialSaved sv(d, next_saved);
        g_next_head.get().w = np->norm1_w; g_next_head.get().b = np->norm1_b;
        g_next_head.get().mean = sv.mean1; g_next_head.get().rstd = sv.rstd1; g_next_head.get().sc = sv.sc1; g_next_head.get().sh = sv.sh1; g_next_head.get().xn = sv.xn;
    }
    g_next_head.get().saved = next_saved;
    g_next_head.get().armed = true;
    return 0;
}
extern "C" int bf_stage_chain_head(const bf_dims* dims, const bf_temporal_params* next_p, void* next_saved) {
    return bf_stage_chain_next(dims, 0, next_p, next_saved);
}

extern "C" int bf_temporal_fwd(const bf_dims* dims, const bf_temporal_params* p, const void* x, void* out, void* saved, void* scratch,
                               const float* drop, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(p && x && out && saved && scratch, "bf_temporal_fwd: null pointer");
    hipStream_t st = (hipStream_t)s;
    TRY(side_join_pending(st));
    TemporalSaved sv(d, saved);
    Scratch sc(d, scratch);
    const void *win_c, *wout_c;
    {
        const float* src[2] = {p->input_head_w, p->output_head_w};
        void* dst[2] = {sv.win_c, sv.wout_c};
        const long n[2] = {3L * d.E * d.E, (long)d.E * d.E};
        const void* out[4];
        const bool b16 = d.dtype == BF_DTYPE_BF16;
        const PrepArgs prep{p->output_head_w, p->output_head_b, p->norm2_b, p->gamma, nullptr, nullptr, sv.alpha, sv.beta, sv.mc, d.E, sv.wout_s, d.dtype,
                            nullptr, nullptr, nullptr, 0, b16 ? p->output_head_w : nullptr, b16 ? sv.wout_t : nullptr, d.E, d.E};
        const bool ready = g_stage_prepared && d.dtype == BF_DTYPE_BF16;
        g_stage_prepared = false;
        if (ready) { out[0] = sv.win_c; out[1] = sv.wout_c; }
        else TRY(wviews(d, 2, src, dst, n, out, st, &prep));
        win_c = out[0]; wout_c = out[1];
    }
    if (g_head_done_for.get() == saved) g_head_done_for.get() = nullptr;      // the stage in front left norm1's statistics and xn behind (bf_stage_chain_head)
    else
        TRY(bf_in_stats_apply(d.dtype, x, (int)d.F, (int)d.S, d.E, p->norm1_w, p->norm1_b, nullptr, 1, nullptr, sv.mean1, sv.rstd1, sv.sc1, sv.sh1, sc.in_ws,
                              nullptr, sv.xn, st));
    TRY(qkv_gemm(d, sv.xn, win_c, p->input_head_b, sv.qkv, st));
    // sequences along T for every (b, y, x): token = b*T*S + pos + t*S
    TRY(bf_attn_fwd(d.dtype, sv.qkv, sv.o, (long)d.B * d.S, d.T, d.S, (long)d.T * d.S, 1, d.S, d.heads, d.d, p->qnorm_w, p->qnorm_b,
                    p->knorm_w, p->knorm_b, p->rel_pos_emb, d.attn_scale ? p->attn_scale_factor : nullptr, 1.f, 0, st));
    TRY(bf_in_stats_apply(d.dtype, sv.o, (int)d.F, (int)d.S, d.E, p->norm2_w, p->norm2_b, nullptr, 1, nullptr, sv.mean2, sv.rstd2, sv.sc2, sv.sh2, sc.in_ws,
                          nullptr, sv.on, st));
    if (g_next_head.get().armed) {      // the stage behind opens with InstanceNorm(out): it rides in the out-projection's launch (bf_stage_chain_next)
        const NextHead h = g_next_head.get();
        g_next_head.get().armed = false;
        g_head_done_for.get() = nullptr;
        const bf_frame_norm n2{h.w, h.b, nullptr, 1, h.mean, h.rstd, h.sc, h.sh, nullptr, h.xn};
        const int rc = bf_gemm_fwd_frames(d.dtype, (int)d.N, d.E, d.E, sv.on, d.E, sv.wout_t, d.E, nullptr, sv.alpha, sv.beta, drop, d.T, x, out, (int)d.S,
                                          nullptr, &n2, s);
        if (rc < 0) return rc;
        if (rc == 0) { g_head_done_for.get() = h.saved; return 0; }
    }
    TRY(outproj_gemm(d, sv.on, wout_c, sv.alpha, sv.beta, x, out, drop, (long)d.T * d.S, st));   // mask per batch element
    return 0;
}

extern "C" int bf_temporal_bwd(const bf_dims* dims, const bf_temporal_params* p, const bf_temporal_params* g, const void* x, const void* dout,
                               void* dx, void* saved, void* scratch, const float* drop, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(p && g && x && dout && dx && saved && scratch, "bf_temporal_bwd: null pointer");
    hipStream_t st = (hipStream_t)s;
    TemporalSaved sv(d, saved);
    Scratch sc(d, bwd_scratch(d, scratch));
    const void* win_c = d.dtype == BF_DTYPE_F32 ? (const void*)p->input_head_w : sv.win_c;
    Fork fk(st, true, g_scratch_parity.get());      // weight-gradient GEMMs go to the side stream; nothing they read (dbr, t3, s1) is rewritten before the join
    if (fk.deferred) TRY(side_join_pending(st, fk.set));      // the stage before the previous one used this scratch set
    void* don = sc.t1;      // [N][E]
    void* dO = sc.t1b;      // [N][E]
    void* dqkv = sc.t3;     // [N][3E]
    // branch = drop[b] * (...): scale the incoming gradient once (2U pass), the rest is unchanged (making the scaled tensor on the side
    // stream instead was measured twice and lost: EXPERIMENTS.md)
    const void* dbr = dout;
    if (drop) {
        if (g_dbr_ready.get().dx == dout && g_dbr_ready.get().f == drop && g_dbr_ready.get().buf) dbr = g_dbr_ready.get().buf;      // the stage in front left the scaled copy behind
        else {
            TRY(bf_frame_scale(d.dtype, dout, drop, d.T, sc.t4, d.N, (int)d.S, d.E, st));
            dbr = sc.t4;
        }
    }
    g_dbr_ready.get() = DbrReady{};
    const InFuse fu2{sv.o, nullptr, dO, sv.mean2, sv.rstd2, p->norm2_w, p->norm2_b, sc.in_ws2};      // don @ ... then norm2's backward -> dO
    TRY(outproj_bwd(d, sc, dbr, sv.on, sv.wout_s, p->output_head_w, p->output_head_b, p->norm2_b, p->gamma, nullptr, nullptr,
                    sv.alpha, sv.mc, g->output_head_w, g->output_head_b, nullptr, g->gamma, nullptr, nullptr, don, st, fk, &fu2));
    ReduceJobs jobs;        // parameter-gradient reductions, all launched together at the end
    jobs.in[jobs.n_in++] = InReduceJob{sc.in_ws2, (int)d.F, d.E, p->norm2_w, p->norm2_b, nullptr, 1, g->norm2_w, g->norm2_b, nullptr, nullptr, nullptr, nullptr};
    {
        int rows = 0;
        TRY(bf_attn_bwd_partials(d.dtype, sv.qkv, dO, dqkv, (long)d.B * d.S, d.T, d.S, (long)d.T * d.S, 1, d.S, d.heads, d.d, p->qnorm_w, p->qnorm_b,
                                 p->knorm_w, p->knorm_b, p->rel_pos_emb, d.attn_scale ? p->attn_scale_factor : nullptr, g->qnorm_w, g->qnorm_b,
                                 g->knorm_w, g->knorm_b, g->rel_pos_emb, d.attn_scale ? g->attn_scale_factor : nullptr, 1.f, 0, sc.attn_ws,
                                 Scratch::ATTN_WS_FLOATS, &rows, st));
        jobs.at[jobs.n_attn++] = AttnReduceJob{sc.attn_ws, rows, d.d, d.heads, g->qnorm_w, g->qnorm_b, g->knorm_w, g->knorm_b, g->rel_pos_emb,
                                               d.attn_scale ? g->attn_scale_factor : nullptr};
    }
    void* dxn = sc.t1;      // don is dead
    InFuse fu1{x, dout, dx, sv.mean1, sv.rstd1, p->norm1_w, p->norm1_b, sc.in_ws};             // dqkv @ W_in, then norm1's backward + residual
    // Chained tail (bf_stage_chain_tail): dx is the output gradient of the spatial stage in front, whose backward opens with its MLP-branch
    // InstanceNorm -- applied here, into THAT stage's scratch set (the other one: its earlier user's side work is joined first, as the
    // spatial stage itself would at its start).
    TailNorm tn;
    bool tail_done = false;
    if (g_next_tail.get().armed) {
        const NextTail h = g_next_tail.get();
        g_next_tail.get().armed = false;
        static const bool chain_on = bf_knob("BF_BWD_CHAIN", 1) != 0;
        if (chain_on && fk.deferred && d.dtype == BF_DTYPE_BF16 && d.S == 144 && d.F % 2 == 0) {
            const int oset = g_scratch_parity.get() ^ 1;
            TRY(side_join_pending(st, oset));
            Scratch so(d, (char*)scratch + (size_t)oset * Scratch(d, nullptr).bytes);
            SpatialSaved ps(d, const_cast<void*>(h.saved));
            tn = TailNorm{ps.z, so.t1, ps.mean3, ps.rstd3, h.p->mlp_norm_w, h.drop ? ps.gtab : h.p->gamma_mlp, h.drop ? 1 : (int)d.F, g_tail_ws_flip.get() ? so.in_ws5 : so.in_ws4, &tail_done};
            g_tail_ws.get() = tn.ws;
            g_tail_ws_flip.get() = !g_tail_ws_flip.get();
            fu1.tail = &tn;
        }
        g_tail_done_for.get() = nullptr;
        TRY(linear_bwd(d, sc, dqkv, 3 * d.E, sv.xn, d.E, BF_PRO_NONE, nullptr, nullptr, win_c, g->input_head_w, g->input_head_b, dxn, nullptr, st, fk, &fu1, true));
        if (tail_done) { g_tail_done_for.get() = h.saved; g_tail_dx.get() = dx; }
    } else
    TRY(linear_bwd(d, sc, dqkv, 3 * d.E, sv.xn, d.E, BF_PRO_NONE, nullptr, nullptr, win_c, g->input_head_w, g->input_head_b, dxn, nullptr, st, fk, &fu1, true));
    jobs.in[jobs.n_in++] = InReduceJob{sc.in_ws, (int)d.F, d.E, p->norm1_w, p->norm1_b, nullptr, 1, g->norm1_w, g->norm1_b, nullptr, nullptr, nullptr, nullptr};
    TRY(launch_with_pending(jobs, st));      // ... with the reductions the spatial stage behind (in the forward) left pending
    return fk.join();
}

// Parameter preparation of many trunk stages at once (what each stage forward otherwise launches for itself): bf16 weight copies, the
// out-projection fold and the MLP branch's stochastic-depth table, written into each stage's `saved` record.  kinds[i]: 0 temporal
// (params[i] = bf_temporal_params*), 1 spatial (bf_spatial_params*); drop_mlp[i]: the spatial stage's MLP-branch factors or NULL.
// A stage forward consumes it when bf_stage_prepared(1) was called just before.  bf16 only (returns 1 in fp32 mode: nothing to cast).
extern "C" int bf_prep_stages(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, void* const* saved,
                              const float* const* drop_mlp, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(n >= 1 && kinds && params && saved, "bf_prep_stages: bad arguments");
    if (d.dtype != BF_DTYPE_BF16) return 1;
    hipStream_t st = (hipStream_t)s;
    TRY(side_join_pending(st));
    for (int i0 = 0; i0 < n; i0 += PREP_BATCH) {
        PrepBatch b;
        memset(&b, 0, sizeof(b));
        const int m = std::min(PREP_BATCH, n - i0);
        for (int i = 0; i < m; ++i) {
            BF_REQUIRE(params[i0 + i] && saved[i0 + i] && (kinds[i0 + i] == 0 || kinds[i0 + i] == 1), "bf_prep_stages: bad stage entry");
            Cast4& j = b.j[i];
            if (kinds[i0 + i] == 0) {
                const bf_temporal_params* p = (const bf_temporal_params*)params[i0 + i];
                TemporalSaved sv(d, saved[i0 + i]);
                const float* src[2] = {p->input_head_w, p->output_head_w};
                void* dst[2] = {sv.win_c, sv.wout_c};
                const long cn[2] = {3L * d.E * d.E, (long)d.E * d.E};
                for (int q = 0; q < 4; ++q) { j.src[q] = src[q < 2 ? q : 0]; j.dst[q] = (bf16*)dst[q < 2 ? q : 0]; j.n[q] = q < 2 ? cn[q] : 0; }
                b.cnt[i] = 2;
                b.a[i] = PrepArgs{p->output_head_w, p->output_head_b, p->norm2_b, p->gamma, nullptr, nullptr, sv.alpha, sv.beta, sv.mc, d.E, sv.wout_s, d.dtype,
                                  nullptr, nullptr, nullptr, 0, p->output_head_w, sv.wout_t, d.E, d.E};
            } else {
                const bf_spatial_params* p = (const bf_spatial_params*)params[i0 + i];
                SpatialSaved sv(d, saved[i0 + i]);
                const float* src[4] = {p->input_head_w, p->output_head_w, p->fc1_w, p->fc2_w};
                void* dst[4] = {sv.win_c, sv.wout_c, sv.w1_c, sv.w2_c};
                const long cn[4] = {3L * d.E * d.E, (long)d.E * d.E, 4L * d.E * d.E, 4L * d.E * d.E};
                for (int q = 0; q < 4; ++q) { j.src[q] = src[q]; j.dst[q] = (bf16*)dst[q]; j.n[q] = cn[q]; }
                b.cnt[i] = 4;
                const float* dm = drop_mlp ? drop_mlp[i0 + i] : nullptr;
                b.a[i] = PrepArgs{p->output_head_w, p->output_head_b, p->norm2_b, p->gamma_att, d.feat_scale ? p->low_freq_scalar : nullptr,
                                  d.feat_scale ? p->high_freq_scalar : nullptr, sv.alpha, sv.beta, sv.mc, d.E, sv.wout_s, d.dtype,
                                  dm, dm ? p->gamma_mlp : nullptr, dm ? sv.gtab : nullptr, (int)d.F,
                                  p->fc2_w, sv.w2t_c, d.E, 4 * d.E};
            }
        }
        hipLaunchKernelGGL(stage_prep_multi_kernel, dim3(std::max(64, d.E), 7, m), dim3(256), 0, st, b);
        BF_CHECK_LAUNCH();
    }
    return 0;
}
extern "C" void bf_stage_prepared(int on) { g_stage_prepared = on != 0; }

// ================================================================================================= axial (spatial) block
extern "C" int bf_spatial_fwd(const bf_dims* dims, const bf_spatial_params* p, const void* x, void* out, void* saved, void* scratch,
                              const float* drop_att, const float* drop_mlp, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(p && x && out && saved && scratch, "bf_spatial_fwd: null pointer");
    const bool head_done = g_head_done_for.get() == saved;      // the temporal stage in front left norm1's statistics and xn behind (bf_stage_chain_next)
    g_head_done_for.get() = nullptr;         // (a chained head is consumed by the stage called right after the stage that made it)
    hipStream_t st = (hipStream_t)s;
    TRY(side_join_pending(st));
    SpatialSaved sv(d, saved);
    Scratch sc(d, scratch);
    const void *win_c, *wout_c, *w1_c, *w2_c;
    {
        const float* src[4] = {p->input_head_w, p->output_head_w, p->fc1_w, p->fc2_w};
        void* dst[4] = {sv.win_c, sv.wout_c, sv.w1_c, sv.w2_c};
        const long n[4] = {3L * d.E * d.E, (long)d.E * d.E, 4L * d.E * d.E, 4L * d.E * d.E};
        const void* out[4];
        const bool tab = drop_mlp && d.dtype != BF_DTYPE_F32;      // gtab[f][c] = drop_mlp[f] * gamma_mlp[c], in the same launch
        const PrepArgs prep{p->output_head_w, p->output_head_b, p->norm2_b, p->gamma_att, d.feat_scale ? p->low_freq_scalar : nullptr,
                            d.feat_scale ? p->high_freq_scalar : nullptr, sv.alpha, sv.beta, sv.mc, d.E, sv.wout_s, d.dtype,
                            tab ? drop_mlp : nullptr, tab ? p->gamma_mlp : nullptr, tab ? sv.gtab : nullptr, (int)d.F,
                            d.dtype == BF_DTYPE_BF16 ? p->fc2_w : nullptr, d.dtype == BF_DTYPE_BF16 ? sv.w2t_c : nullptr, d.E, 4 * d.E};
        const bool ready = g_stage_prepared && d.dtype == BF_DTYPE_BF16;
        g_stage_prepared = false;
        if (ready) { out[0] = sv.win_c; out[1] = sv.wout_c; out[2] = sv.w1_c; out[3] = sv.w2_c; }
        else TRY(wviews(d, 4, src, dst, n, out, st, &prep));
        win_c = out[0]; wout_c = out[1]; w1_c = out[2]; w2_c = out[3];
    }
    if (!head_done)
        TRY(bf_in_stats_apply(d.dtype, x, (int)d.F, (int)d.S, d.E, p->norm1_w, p->norm1_b, nullptr, 1, nullptr, sv.mean1, sv.rstd1, sv.sc1, sv.sh1, sc.in_ws,
                              nullptr, sv.xn, st));
    TRY(qkv_gemm(d, sv.xn, win_c, p->input_head_b, sv.qkv, st));
    // along w (one sequence per (frame, row): contiguous tokens), then along h (per (frame, column): stride w), averaged
    {   // ... and norm2 in the same launch where the one-launch form applies
        const int rc = bf_attn_axial_norm_fwd(d.dtype, sv.qkv, sv.o, sv.on, d.F, (int)d.h, (int)d.w, d.heads, d.d, p->qnorm_w, p->qnorm_b, p->knorm_w,
                                              p->knorm_b, p->rel_pos_emb, d.attn_scale ? p->attn_scale_factor_x : nullptr,
                                              d.attn_scale ? p->attn_scale_factor_y : nullptr, p->norm2_w, p->norm2_b, sv.mean2, sv.rstd2, sv.sc2,
                                              sv.sh2, st);
        if (rc < 0) return rc;
        if (rc == 1) {
            TRY(bf_attn_axial_fwd(d.dtype, sv.qkv, sv.o, d.F, (int)d.h, (int)d.w, d.heads, d.d, p->qnorm_w, p->qnorm_b, p->knorm_w, p->knorm_b,
                                  p->rel_pos_emb, d.attn_scale ? p->attn_scale_factor_x : nullptr, d.attn_scale ? p->attn_scale_factor_y : nullptr, st));
            TRY(bf_in_stats_apply(d.dtype, sv.o, (int)d.F, (int)d.S, d.E, p->norm2_w, p->norm2_b, nullptr, 1, nullptr, sv.mean2, sv.rstd2, sv.sc2, sv.sh2,
                                  sc.in_ws, nullptr, sv.on, st));
        }
    }
    TRY(outproj_gemm(d, sv.on, wout_c, sv.alpha, sv.beta, x, sv.x1, drop_att, d.S, st));     // mask per frame
    {   // pre = x1 @ W1^T + b1 ; hid = gelu(pre) (both kept: pre for gelu', hid as the fc2 operand -- no erf in any prologue)
        bf_operand A = op_plain(sv.x1, d.E, BF_LAY_KC);
        bf_operand Bo = op_plain(w1_c, d.E, BF_LAY_KC);
        bf_epilogue e = epi_store(sv.pre, 4L * d.E);
        e.bias = p->fc1_b;
        e.gelu_out = sv.hid;
        TRY(bf_gemm(d.dtype, (int)d.N, 4 * d.E, d.E, &A, &Bo, &e, 1, st));
    }
    // out = x1 + drop_mlp[f] * gamma_mlp * InstanceNorm(z): the per-(frame, channel) factor rides in the InstanceNorm affine
    const float* g3 = p->gamma_mlp;
    int g3div = (int)d.F;
    if (drop_mlp) {
        if (d.dtype == BF_DTYPE_F32) {      // bf16: made by the stage's preparation launch above
            hipLaunchKernelGGL(frame_table_kernel, dim3(bf_cdiv(d.F * d.E, 256)), dim3(256), 0, st, drop_mlp, 1, (const float*)p->gamma_mlp, sv.gtab, (int)d.F, d.E);
            BF_CHECK_LAUNCH();
        }
        g3 = sv.gtab; g3div = 1;
    }
    if (d.dtype == BF_DTYPE_BF16) {      // z = hid @ W2^T + b2, the MLP-branch norm + residual and (armed) the next stage's opening norm in ONE launch
        const bf_frame_norm n1{p->mlp_norm_w, p->mlp_norm_b, g3, g3div, sv.mean3, sv.rstd3, sv.sc3, sv.sh3, sv.x1, out};
        const NextHead h = g_next_head.get();
        const bf_frame_norm n2{h.w, h.b, nullptr, 1, h.mean, h.rstd, h.sc, h.sh, nullptr, h.xn};
        const int rc = bf_gemm_fwd_frames(d.dtype, (int)d.N, d.E, 4 * d.E, sv.hid, 4L * d.E, sv.w2t_c, d.E, p->fc2_b, nullptr, nullptr, nullptr, 1, nullptr, sv.z,
                                          (int)d.S, &n1, h.armed ? &n2 : nullptr, s);
        if (rc < 0) return rc;
        if (rc == 0) {
            g_next_head.get().armed = false;
            g_head_done_for.get() = h.armed ? h.saved : nullptr;
            return 0;
        }
    }
    {   // z = hid @ W2^T + b2
        bf_operand A = op_plain(sv.hid, 4L * d.E, BF_LAY_KC);
        bf_operand Bo = op_plain(w2_c, 4L * d.E, BF_LAY_KC);
        bf_epilogue e = epi_store(sv.z, d.E);
        e.bias = p->fc2_b;
        TRY(bf_gemm(d.dtype, (int)d.N, d.E, 4 * d.E, &A, &Bo, &e, 1, st));
    }
    if (g_next_head.get().armed) {
        const NextHead h = g_next_head.get();
        g_next_head.get().armed = false;
        bool chained = false;
        TRY(bf_in_stats_apply_chain(d.dtype, sv.z, (int)d.F, (int)d.S, d.E, p->mlp_norm_w, p->mlp_norm_b, g3, g3div, nullptr, sv.mean3, sv.rstd3, sv.sc3, sv.sh3,
                                    sc.in_ws, sv.x1, out, h.w, h.b, h.mean, h.rstd, h.sc, h.sh, h.xn, &chained, st));
        g_head_done_for.get() = chained ? h.saved : nullptr;
        return 0;
    }
    TRY(bf_in_stats_apply(d.dtype, sv.z, (int)d.F, (int)d.S, d.E, p->mlp_norm_w, p->mlp_norm_b, g3, g3div, nullptr, sv.mean3, sv.rstd3,
                          sv.sc3, sv.sh3, sc.in_ws, sv.x1, out, st));
    return 0;
}

extern "C" int bf_spatial_bwd(const bf_dims* dims, const bf_spatial_params* p, const bf_spatial_params* g, const void* x, const void* dout,
                              void* dx, void* saved, void* scratch, const float* drop_att, const float* drop_mlp, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(p && g && x && dout && dx && saved && scratch, "bf_spatial_bwd: null pointer");
    hipStream_t st = (hipStream_t)s;
    SpatialSaved sv(d, saved);
    Scratch sc(d, bwd_scratch(d, scratch));
    const bool f32 = d.dtype == BF_DTYPE_F32;
    const void* win_c = f32 ? (const void*)p->input_head_w : sv.win_c;
    const void* w1_c = f32 ? (const void*)p->fc1_w : sv.w1_c;
    const void* w2_c = f32 ? (const void*)p->fc2_w : sv.w2_c;
    // weight-gradient GEMMs go to the side stream and are joined at the end; every buffer they read (dz = t1, dpre = t4,
    // dx1 = t1b, dbr = e5, dqkv = t3, s1) is written once per call, so the critical path below never recycles one under them.
    TRY(flush_pending_reduce(st));      // (a spatial stage that no temporal stage followed: its partial sums live in the set this stage is about to use)
    Fork fk(st, true, g_scratch_parity.get());
    if (fk.deferred) TRY(side_join_pending(st, fk.set));      // the stage before the previous one used this scratch set
    // out = x1 + gamma_mlp * IN(z)
    void* dz = sc.t1;
    ReduceJobs jobs;        // parameter-gradient reductions, all launched together at the end
    // the temporal stage behind left dz and the partial sums in place (bf_stage_chain_tail) -- usable only if this call's dout IS the dx that stage
    // wrote (another consumer of the stage's output, or a gradient hook, makes autograd hand over a different, summed tensor: recompute then)
    const bool tail_done = g_tail_done_for.get() == saved && g_tail_dx.get() == dout;
    g_tail_done_for.get() = nullptr; g_tail_dx.get() = nullptr;
    if (drop_mlp) {   // gtab[f][c] = drop_mlp[f] * gamma_mlp[c] was the scale: d gamma_mlp = sum_f drop_mlp[f] * (w s2 + b s1), folded in the reduction
        if (!tail_done)
            TRY(bf_in_bwd_partials(d.dtype, dout, sv.z, nullptr, dz, (int)d.F, (int)d.S, d.E, sv.mean3, sv.rstd3, p->mlp_norm_w, p->mlp_norm_b, sv.gtab, 1, 0, sc.in_ws3, st));
        jobs.in[jobs.n_in++] = InReduceJob{tail_done ? g_tail_ws.get() : sc.in_ws3, (int)d.F, d.E, p->mlp_norm_w, p->mlp_norm_b, sv.gtab, 1, g->mlp_norm_w, g->mlp_norm_b, nullptr, nullptr,
                                           drop_mlp, g->gamma_mlp};
    } else {
        if (!tail_done)
            TRY(bf_in_bwd_partials(d.dtype, dout, sv.z, nullptr, dz, (int)d.F, (int)d.S, d.E, sv.mean3, sv.rstd3, p->mlp_norm_w, p->mlp_norm_b, p->gamma_mlp,
                                   (int)d.F, 0, sc.in_ws3, st));
        jobs.in[jobs.n_in++] = InReduceJob{tail_done ? g_tail_ws.get() : sc.in_ws3, (int)d.F, d.E, p->mlp_norm_w, p->mlp_norm_b, p->gamma_mlp, (int)d.F, g->mlp_norm_w, g->mlp_norm_b,
                                           g->gamma_mlp, nullptr, nullptr, nullptr};
    }
    // fc2: z = gelu(pre) @ W2^T + b2 ; dpre = (dz @ W2) * gelu'(pre)
    void* dpre = sc.t4;
    {
        bf_epilogue e; memset(&e, 0, sizeof(e));
        e.aux_mode = BF_AUX_DGELU; e.aux = sv.pre; e.ld_aux = 4L * d.E; e.out_mode = BF_OUT_STORE;
        static const bool use_t = bf_knob("BF_FC2_DGRAD_T", 1) != 0;      // K-contiguous transposed weight: the weight-stationary ring kernel's form
        TRY(linear_bwd(d, sc, dz, d.E, sv.hid, 4 * d.E, BF_PRO_NONE, nullptr, nullptr, w2_c, g->fc2_w, g->fc2_b, dpre, &e, st, fk, nullptr, false,
                       (!f32 && use_t) ? sv.w2t_c : nullptr));
    }
    // fc1: pre = x1 @ W1^T + b1 ; dx1 = dout + dpre @ W1
    void* dx1 = sc.t1b;
    bool dbr_done = false;
    {
        bf_epilogue e; memset(&e, 0, sizeof(e));
        e.aux_mode = BF_AUX_ADD; e.aux = dout; e.ld_aux = d.E; e.out_mode = BF_OUT_STORE;
        // ... and, under stochastic depth, the gradient entering the attention branch (drop_att[f] * dx1) as the kernel's second output
        TRY(linear_bwd(d, sc, dpre, 4 * d.E, sv.x1, d.E, BF_PRO_NONE, nullptr, nullptr, w1_c, g->fc1_w, g->fc1_b, dx1, &e, st, fk, nullptr, false, nullptr,
                       drop_att ? sc.e5 : nullptr, drop_att, (int)d.S, &dbr_done));
    }
    // folded out-projection
    void* don = sc.e6;
    const void* dbr = dx1;  // gradient entering the attention branch (dx1 itself continues down the residual)
    if (dbr_done) dbr = sc.e5;
    else if (drop_att) {
        TRY(bf_frame_scale(d.dtype, dx1, drop_att, 1, sc.e5, d.N, (int)d.S, d.E, st));
        dbr = sc.e5;
    }
    void* dO = sc.e7;       // [N][E]
    const InFuse fu2{sv.o, nullptr, dO, sv.mean2, sv.rstd2, p->norm2_w, p->norm2_b, sc.in_ws2};      // don @ ... then norm2's backward -> dO
    TRY(outproj_bwd(d, sc, dbr, sv.on, sv.wout_s, p->output_head_w, p->output_head_b, p->norm2_b, p->gamma_att,
                    d.feat_scale ? p->low_freq_scalar : nullptr, d.feat_scale ? p->high_freq_scalar : nullptr, sv.alpha, sv.mc,
                    g->output_head_w, g->output_head_b, g->norm2_b, g->gamma_att, d.feat_scale ? g->low_freq_scalar : nullptr,
                    d.feat_scale ? g->high_freq_scalar : nullptr, don, st, fk, &fu2));
    jobs.in[jobs.n_in++] = InReduceJob{sc.in_ws2, (int)d.F, d.E, p->norm2_w, p->norm2_b, nullptr, 1, g->norm2_w, g->norm2_b, nullptr, nullptr, nullptr, nullptr};
    void* dqkv = sc.t3;
    {
        int rows = 0;
        // the two passes share their tokens' q / k LayerNorms, whose backward is linear in the incoming gradient: the W pass leaves its raw
        // gradients, the H pass adds its own and runs that backward (and the LayerNorm parameter sums) once (bf16 MFMA path)
        static const bool raw_on = bf_knob("BF_ATTN_RAW", 1) != 0;
        const bool rawm = raw_on && bf_attn_raw_modes(d.dtype, d.d);
        TRY(bf_attn_bwd_partials(d.dtype, sv.qkv, dO, dqkv, d.F * d.h, d.w, 1, d.w, 0, 1, d.heads, d.d, p->qnorm_w, p->qnorm_b, p->knorm_w, p->knorm_b,
                                 p->rel_pos_emb, d.attn_scale ? p->attn_scale_factor_x : nullptr, g->qnorm_w, g->qnorm_b, g->knorm_w, g->knorm_b,
                                 g->rel_pos_emb, d.attn_scale ? g->attn_scale_factor_x : nullptr, 0.5f, rawm ? 2 : 0, sc.attn_ws, Scratch::ATTN_WS_FLOATS, &rows, st));
        jobs.at[jobs.n_attn++] = AttnReduceJob{sc.attn_ws, rows, d.d, d.heads, g->qnorm_w, g->qnorm_b, g->knorm_w, g->knorm_b, g->rel_pos_emb,
                                               d.attn_scale ? g->attn_scale_factor_x : nullptr};
        TRY(bf_attn_bwd_partials(d.dtype, sv.qkv, dO, dqkv, d.F * d.w, d.h, d.w, d.S, 1, d.w, d.heads, d.d, p->qnorm_w, p->qnorm_b, p->knorm_w, p->knorm_b,
                                 p->rel_pos_emb, d.attn_scale ? p->attn_scale_factor_y : nullptr, g->qnorm_w, g->qnorm_b, g->knorm_w, g->knorm_b,
                                 g->rel_pos_emb, d.attn_scale ? g->attn_scale_factor_y : nullptr, 0.5f, rawm ? 5 : 1, sc.attn_ws2, Scratch::ATTN_WS_FLOATS, &rows, st));
        jobs.at[jobs.n_attn++] = AttnReduceJob{sc.attn_ws2, rows, d.d, d.heads, g->qnorm_w, g->qnorm_b, g->knorm_w, g->knorm_b, g->rel_pos_emb,
                                               d.attn_scale ? g->attn_scale_factor_y : nullptr};
    }
    void* dxn = sc.e6;      // don is dead (it was only read on this stream)
    InFuse fu1{x, dx1, dx, sv.mean1, sv.rstd1, p->norm1_w, p->norm1_b, sc.in_ws};              // dqkv @ W_in, then norm1's backward + residual
    bool scaled_done = false;
    const NextScale ns = g_next_scale.get();
    g_next_scale.get() = NextScale{};
    g_dbr_ready.get() = DbrReady{};
    static const bool scaled_on = bf_knob("BF_BWD_SCALED_COPY", 1) != 0;
    if (ns.f && scaled_on && fk.deferred && d.dtype == BF_DTYPE_BF16) {
        fu1.scaled_out = (char*)scratch + 2 * Scratch(d, nullptr).bytes + (g_dbr_flip.get() ? dbr_bytes(d) : 0);
        fu1.scaled_f = ns.f; fu1.scaled_fdiv = ns.fdiv; fu1.scaled_done = &scaled_done;
    }
    TRY(linear_bwd(d, sc, dqkv, 3 * d.E, sv.xn, d.E, BF_PRO_NONE, nullptr, nullptr, win_c, g->input_head_w, g->input_head_b, dxn, nullptr, st, fk, &fu1, true));
    if (scaled_done) { g_dbr_ready.get() = DbrReady{dx, ns.f, fu1.scaled_out}; g_dbr_flip.get() = !g_dbr_flip.get(); }
    jobs.in[jobs.n_in++] = InReduceJob{sc.in_ws, (int)d.F, d.E, p->norm1_w, p->norm1_b, nullptr, 1, g->norm1_w, g->norm1_b, nullptr, nullptr, nullptr, nullptr};
    static const bool merge_on = bf_knob("BF_REDUCE_MERGE", 1) != 0;
    if (fk.deferred && merge_on) { g_pending_reduce.get() = jobs; g_pending_reduce_on.get() = true; }      // rides in the next temporal stage's launch (or the next join)
    else TRY(launch_reduce_jobs(jobs, st));
    return fk.join();
}

// ================================================================================================= the training trunk in one call per direction
// The n trunk stages of a training step (SpaceTimeBlock x 12: temporal, spatial, temporal, ...; models/axial_vit.py:58-63, 234-235) enqueued
// by ONE native call each way instead of one Python -> ctypes round trip per stage: the stage forwards / backwards above, called in a
// loop with the chain hints between them (bf_stage_chain_next, bf_stage_chain_tail, bf_stage_next_scale) set here, where the whole sequence
// is known.  Everything is caller-owned: saved[i] = stage i's record (bf_temporal_saved_bytes / bf_spatial_saved_bytes), acts[i] = stage
// i's output [N][E] (the last one is the trunk's output), three [N][E] gradient buffers that rotate between consecutive backward stages
// (a stage's side-stream work may read its incoming gradient until the stage after the next one starts).
extern "C" int bf_trunk_train_fwd(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, void* const* saved,
                                  const float* const* drop_a, const float* const* drop_b, const void* x, void* const* acts, void* scratch,
                                  bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(n >= 1 && kinds && params && saved && x && acts && scratch, "bf_trunk_train_fwd: bad arguments");
    for (int i = 0; i < n; ++i)
        BF_REQUIRE(params[i] && saved[i] && acts[i] && (kinds[i] == 0 || kinds[i] == 1), "bf_trunk_train_fwd: bad stage entry");
    const bool b16 = d.dtype == BF_DTYPE_BF16;
    if (b16) {
        const int rc = bf_prep_stages(dims, n, kinds, params, saved, drop_b, s);
        if (rc < 0) return rc;
    }
    for (int i = 0; i < n; ++i) {
        const void* in = i ? acts[i - 1] : x;
        // the next stage's opening InstanceNorm rides in this stage's last GEMM launch where the kinds alternate (bf16)
        if (b16 && i + 1 < n && kinds[i + 1] != kinds[i]) TRY(bf_stage_chain_next(dims, kinds[i + 1], params[i + 1], saved[i + 1]));
        else TRY(bf_stage_chain_next(nullptr, 0, nullptr, nullptr));
        bf_stage_prepared(b16 ? 1 : 0);
        const int rc = kinds[i] == 0
            ? bf_temporal_fwd(dims, (const bf_temporal_params*)params[i], in, acts[i], saved[i], scratch, drop_a ? drop_a[i] : nullptr, s)
            : bf_spatial_fwd(dims, (const bf_spatial_params*)params[i], in, acts[i], saved[i], scratch, drop_a ? drop_a[i] : nullptr,
                             drop_b ? drop_b[i] : nullptr, s);
        if (rc) { (void)bf_stage_chain_next(nullptr, 0, nullptr, nullptr); return rc; }
    }
    return 0;
}

// grads[i]: the gradient struct of stage i (same type as params[i]); gradients ACCUMULATE.  dout: gradient of acts[n - 1]; dx: gradient of x.
// stage_done (optional) is called on the host right after stage i's backward has been enqueued (its last weight-gradient GEMM possibly still
// deferred, see bf_side_defer): the data-parallel bucket reducer's hook.
extern "C" int bf_trunk_train_bwd(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, const void* const* grads,
                                  void* const* saved, const float* const* drop_a, const float* const* drop_b, const void* x, void* const* acts,
                                  const void* dout, void* const* gbuf3, void* dx, void* scratch, bf_stage_done_fn stage_done, void* user,
                                  bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(n >= 1 && kinds && params && grads && saved && x && acts && dout && gbuf3 && dx && scratch, "bf_trunk_train_bwd: bad arguments");
    BF_REQUIRE(gbuf3[0] && gbuf3[1] && gbuf3[2], "bf_trunk_train_bwd: three gradient buffers are needed");
    const void* cur = dout;
    int rot = 0;
    for (int i = n - 1; i >= 0; --i) {
        BF_REQUIRE(params[i] && grads[i] && saved[i] && (kinds[i] == 0 || kinds[i] == 1), "bf_trunk_train_bwd: bad stage entry");
        const void* in = i ? acts[i - 1] : x;
        void* gout = i ? gbuf3[rot] : dx;
        rot = (rot + 1) % 3;
        int rc;
        if (kinds[i] == 0) {
            // the spatial stage in front (in the forward) opens its backward with its MLP-branch norm: applied by this stage's last kernel
            if (i > 0 && kinds[i - 1] == 1)
                TRY(bf_stage_chain_tail((const bf_spatial_params*)params[i - 1], saved[i - 1], drop_b && drop_b[i - 1] ? 1 : 0));
            rc = bf_temporal_bwd(dims, (const bf_temporal_params*)params[i], (const bf_temporal_params*)grads[i], in, cur, gout, saved[i], scratch,
                                 drop_a ? drop_a[i] : nullptr, s);
            (void)bf_stage_chain_tail(nullptr, nullptr, 0);
        } else {
            // the temporal stage in front scales this stage's input gradient by its stochastic-depth factors: written here as a second copy
            if (i > 0 && kinds[i - 1] == 0 && drop_a && drop_a[i - 1]) TRY(bf_stage_next_scale(drop_a[i - 1], d.T));
            rc = bf_spatial_bwd(dims, (const bf_spatial_params*)params[i], (const bf_spatial_params*)grads[i], in, cur, gout, saved[i], scratch,
                                drop_a ? drop_a[i] : nullptr, drop_b ? drop_b[i] : nullptr, s);
            (void)bf_stage_next_scale(nullptr, 0);
        }
        if (rc) return rc;
        if (stage_done) stage_done(i, user);
        cur = gout;
    }
    return 0;
}

// ================================================================================================= inference forward of the trunk
// Eval forward of n trunk stages in one call (scripts/inference.py:239-252: FiLMConditionedAViT.forward under torch.no_grad, one clip at a
// time): nothing is saved for a backward, the InstanceNorms ride inside the whole-frame projection kernels (frame_fwd.hip) and the bf16
// weight copies / out-projection folds live in a caller-owned arena that is prepared ONCE per set of weights, not once per forward.
namespace {
struct EvalStage {       // one stage's slice of the arena
    void *win_c, *wout_c, *w1_c, *w2_c; float *alpha, *beta, *mc;
    size_t bytes;
    EvalStage(const D& d, int kind, void* base) {
        Arena a(base);
        alpha = a.f32(d.E); beta = a.f32(d.E); mc = a.f32(d.E);
        win_c = a.take((size_t)3 * d.E * d.E * 2);
        wout_c = a.take((size_t)d.E * d.E * 2);
        w1_c = kind == 1 ? a.take((size_t)4 * d.E * d.E * 2) : nullptr;
        w2_c = kind == 1 ? a.take((size_t)4 * d.E * d.E * 2) : nullptr;
        bytes = a.off;
    }
};
bool trunk_eval_covers(const D& d) { return d.dtype == BF_DTYPE_BF16 && d.S == 144 && d.E == 384 && d.h <= 16 && d.w <= 16 && d.T <= 32; }
}  // namespace

extern "C" int64_t bf_trunk_eval_weights_bytes(const bf_dims* dims, int n, const int32_t* kinds) {
    D d; if (get_dims(dims, &d) || n < 1 || !kinds) return -1;
    int64_t t = 0;
    for (int i = 0; i < n; ++i) t += (int64_t)EvalStage(d, kinds[i], nullptr).bytes;
    return t;
}

extern "C" int bf_trunk_eval_prepare(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, void* weights, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(n >= 1 && kinds && params && weights, "bf_trunk_eval_prepare: bad arguments");
    if (!trunk_eval_covers(d)) return 1;
    hipStream_t st = (hipStream_t)s;
    char* base = (char*)weights;
    for (int i0 = 0; i0 < n; i0 += PREP_BATCH) {
        PrepBatch b;
        memset(&b, 0, sizeof(b));
        const int m = std::min(PREP_BATCH, n - i0);
        for (int i = 0; i < m; ++i) {
            BF_REQUIRE(params[i0 + i] && (kinds[i0 + i] == 0 || kinds[i0 + i] == 1), "bf_trunk_eval_prepare: bad stage entry");
            EvalStage ev(d, kinds[i0 + i], base);
            base += ev.bytes;
            Cast4& j = b.j[i];
            if (kinds[i0 + i] == 0) {
                const bf_temporal_params* p = (const bf_temporal_params*)params[i0 + i];
                const float* src[2] = {p->input_head_w, p->output_head_w};
                void* dst[2] = {ev.win_c, ev.wout_c};
                const long cn[2] = {3L * d.E * d.E, (long)d.E * d.E};
                for (int q = 0; q < 4; ++q) { j.src[q] = src[q < 2 ? q : 0]; j.dst[q] = (bf16*)dst[q < 2 ? q : 0]; j.n[q] = q < 2 ? cn[q] : 0; }
                b.cnt[i] = 2;
                b.a[i] = PrepArgs{p->output_head_w, p->output_head_b, p->norm2_b, p->gamma, nullptr, nullptr, ev.alpha, ev.beta, ev.mc, d.E, nullptr, d.dtype,
                                  nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0};
            } else {
                const bf_spatial_params* p = (const bf_spatial_params*)params[i0 + i];
                const float* src[4] = {p->input_head_w, p->output_head_w, p->fc1_w, p->fc2_w};
                void* dst[4] = {ev.win_c, ev.wout_c, ev.w1_c, ev.w2_c};
                const long cn[4] = {3L * d.E * d.E, (long)d.E * d.E, 4L * d.E * d.E, 4L * d.E * d.E};
                for (int q = 0; q < 4; ++q) { j.src[q] = src[q]; j.dst[q] = (bf16*)dst[q]; j.n[q] = cn[q]; }
                b.cnt[i] = 4;
                b.a[i] = PrepArgs{p->output_head_w, p->output_head_b, p->norm2_b, p->gamma_att, d.feat_scale ? p->low_freq_scalar : nullptr,
                                  d.feat_scale ? p->high_freq_scalar : nullptr, ev.alpha, ev.beta, ev.mc, d.E, nullptr, d.dtype,
                                  nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0};
            }
        }
        hipLaunchKernelGGL(stage_prep_multi_kernel, dim3(std::max(64, d.E), 5, m), dim3(256), 0, st, b);
        BF_CHECK_LAUNCH();
    }
    return 0;
}

// x, out: [N][E] tokens.  `weights`: the arena bf_trunk_eval_prepare filled for exactly these stages; `scratch`: bf_scratch_bytes.
// Returns 0 when done, 1 when the shape is not covered (bf16, 144-token frames, E = 384): the caller then runs the stage forwards.
extern "C" int bf_trunk_eval_fwd(const bf_dims* dims, int n, const int32_t* kinds, const void* const* params, const void* weights,
                                 const void* x, void* out, void* scratch, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(n >= 1 && kinds && params && weights && x && out && scratch, "bf_trunk_eval_fwd: bad arguments");
    if (!trunk_eval_covers(d)) return 1;
    hipStream_t st = (hipStream_t)s;
    TRY(side_join_pending(st));
    Scratch sc(d, scratch);
    const int F = (int)d.F, E = d.E;
    void *qkv = sc.t3, *o = sc.t1, *on = sc.t1b, *x1 = sc.e5, *hid = sc.t4;
    float* stat = sc.tokred_ws;                           // mean | rstd | sc | sh of the axial block's norm2, [F][E] each
    BF_REQUIRE(sc.tokred_floats >= (int64_t)4 * F * E, "bf_trunk_eval_fwd: scratch too small");
    const char* base = (const char*)weights;
    const void* cur = x;
    const void* xn = nullptr;                             // norm1(cur) of the stage about to run, when the previous stage's last kernel made it
    void* xn_buf = sc.s1;
    auto norm1_of = [&](int i, const float** w, const float** b) {
        if (kinds[i] == 0) { const bf_temporal_params* q = (const bf_temporal_params*)params[i]; *w = q->norm1_w; *b = q->norm1_b; }
        else { const bf_spatial_params* q = (const bf_spatial_params*)params[i]; *w = q->norm1_w; *b = q->norm1_b; }
    };
    for (int i = 0; i < n; ++i) {
        BF_REQUIRE(params[i] && (kinds[i] == 0 || kinds[i] == 1), "bf_trunk_eval_fwd: bad stage entry");
        EvalStage ev(d, kinds[i], (void*)base);
        base += ev.bytes;
        void* nxt = i == n - 1 ? out : ((i & 1) ? sc.e7 : sc.e6);
        // the stage's last kernel holds whole-frame columns of its output: it also writes the NEXT stage's norm1 of it
        const float *nw = nullptr, *nb = nullptr;
        if (i + 1 < n) { BF_REQUIRE(params[i + 1] && (kinds[i + 1] == 0 || kinds[i + 1] == 1), "bf_trunk_eval_fwd: bad stage entry"); norm1_of(i + 1, &nw, &nb); }
        void* xn_next = nw ? xn_buf : nullptr;
#define FRL(...) do { const int rc_ = bf_frame_linear(__VA_ARGS__); if (rc_ != 0) return rc_ < 0 ? rc_ : bf_fail_msg("bf_trunk_eval_fwd: frame kernel refused a covered shape", __FILE__, __LINE__); } while (0)
        const float *w1n, *b1n;
        norm1_of(i, &w1n, &b1n);
        const float* qkv_bias = kinds[i] == 0 ? ((const bf_temporal_params*)params[i])->input_head_b : ((const bf_spatial_params*)params[i])->input_head_b;
        if (xn) FRL(d.dtype, F, 144, E, 3 * E, xn, E, ev.win_c, E, nullptr, nullptr, qkv_bias, nullptr, nullptr, nullptr, 0, 0,
                    nullptr, nullptr, nullptr, qkv, 3L * E, nullptr, nullptr, nullptr, 0, s);
        else FRL(d.dtype, F, 144, E, 3 * E, cur, E, ev.win_c, E, w1n, b1n, qkv_bias, nullptr, nullptr, nullptr, 0, 0,
                 nullptr, nullptr, nullptr, qkv, 3L * E, nullptr, nullptr, nullptr, 0, s);
        if (kinds[i] == 0) {
            const bf_temporal_params* p = (const bf_temporal_params*)params[i];
            TRY(bf_attn_fwd(d.dtype, qkv, o, (long)d.B * d.S, d.T, d.S, (long)d.T * d.S, 1, d.S, d.heads, d.d, p->qnorm_w, p->qnorm_b,
                            p->knorm_w, p->knorm_b, p->rel_pos_emb, d.attn_scale ? p->attn_scale_factor : nullptr, 1.f, 0, st));
            FRL(d.dtype, F, 144, E, E, o, E, ev.wout_c, E, p->norm2_w, p->norm2_b, nullptr, ev.alpha, ev.beta, cur, E, 0,
                nullptr, nullptr, nullptr, nxt, E, nw, nb, xn_next, E, s);
        } else {
            const bf_spatial_params* p = (const bf_spatial_params*)params[i];
            const int rc = bf_attn_axial_norm_fwd(d.dtype, qkv, o, on, d.F, (int)d.h, (int)d.w, d.heads, d.d, p->qnorm_w, p->qnorm_b, p->knorm_w,
                                                  p->knorm_b, p->rel_pos_emb, d.attn_scale ? p->attn_scale_factor_x : nullptr,
                                                  d.attn_scale ? p->attn_scale_factor_y : nullptr, p->norm2_w, p->norm2_b, stat, stat + (size_t)F * E,
                                                  stat + (size_t)2 * F * E, stat + (size_t)3 * F * E, st);
            if (rc < 0) return rc;
            if (rc == 1) {      // the one-launch attention + norm2 form refused: attention, then norm2 inside the out-projection
                TRY(bf_attn_axial_fwd(d.dtype, qkv, o, d.F, (int)d.h, (int)d.w, d.heads, d.d, p->qnorm_w, p->qnorm_b, p->knorm_w, p->knorm_b,
                                      p->rel_pos_emb, d.attn_scale ? p->attn_scale_factor_x : nullptr, d.attn_scale ? p->attn_scale_factor_y : nullptr, st));
                FRL(d.dtype, F, 144, E, E, o, E, ev.wout_c, E, p->norm2_w, p->norm2_b, nullptr, ev.alpha, ev.beta, cur, E, 0,
                    nullptr, nullptr, nullptr, x1, E, nullptr, nullptr, nullptr, 0, s);
            } else {
                FRL(d.dtype, F, 144, E, E, on, E, ev.wout_c, E, nullptr, nullptr, nullptr, ev.alpha, ev.beta, cur, E, 0,
                    nullptr, nullptr, nullptr, x1, E, nullptr, nullptr, nullptr, 0, s);
            }
            FRL(d.dtype, F, 144, E, 4 * E, x1, E, ev.w1_c, E, nullptr, nullptr, p->fc1_b, nullptr, nullptr, nullptr, 0, 1,
                nullptr, nullptr, nullptr, hid, 4L * E, nullptr, nullptr, nullptr, 0, s);
            FRL(d.dtype, F, 144, 4 * E, E, hid, 4L * E, ev.w2_c, 4L * E, nullptr, nullptr, p->fc2_b, nullptr, nullptr, x1, E, 0,
                p->mlp_norm_w, p->mlp_norm_b, p->gamma_mlp, nxt, E, nw, nb, xn_next, E, s);
        }
        xn = xn_next;
#undef FRL
        cur = nxt;
    }
    return 0;
}

// ================================================================================================= patch embed (+ FiLM)
namespace {
inline int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct EmbedSaved {
    float *gb, *dgb, *chat, *crstd;
    void* patches; int Kp;
    void* y[BF_MAX_STAGES]; void* wc[BF_MAX_STAGES];
    float *mean[BF_MAX_STAGES], *rstd[BF_MAX_STAGES], *sc[BF_MAX_STAGES], *sh[BF_MAX_STAGES];
    int C[BF_MAX_STAGES], gh[BF_MAX_STAGES], gw[BF_MAX_STAGES]; long P[BF_MAX_STAGES];
    size_t bytes;
    EmbedSaved(const D& d, void* base) {
        Arena a(base);
        const int np = d.nfluid > 0 ? d.nfluid : 1;
        gb = a.f32((size_t)2 * d.B * d.E); dgb = a.f32((size_t)2 * d.B * d.E); chat = a.f32((size_t)d.B * np); crstd = a.f32(d.B);
        Kp = roundup(4 * d.cin, 8);
        const int H = d.h * d.patch, W = d.w * d.patch;
        for (int i = 0; i < d.nst; ++i) {
            C[i] = (i == d.nst - 1) ? d.E : d.E / 4;
            gh[i] = H >> (i + 1); gw[i] = W >> (i + 1);
            P[i] = d.F * gh[i] * gw[i];
        }
        patches = a.take((size_t)P[0] * Kp * d.es);
        for (int i = 0; i < d.nst; ++i) {
            const int kin = i == 0 ? Kp : 4 * C[i - 1];
            y[i] = a.take((size_t)P[i] * C[i] * d.es);
            wc[i] = a.take((size_t)C[i] * kin * d.es);
            const size_t fc = (size_t)d.F * C[i];
            mean[i] = a.f32(fc); rstd[i] = a.f32(fc); sc[i] = a.f32(fc); sh[i] = a.f32(fc);
        }
        bytes = a.off;
    }
};
struct DebedSaved {
    float *lossbuf, *coef;
    void* y[BF_MAX_STAGES]; void* wc[BF_MAX_STAGES];
    float *mean[BF_MAX_STAGES], *rstd[BF_MAX_STAGES], *sc[BF_MAX_STAGES], *sh[BF_MAX_STAGES];
    int Cin[BF_MAX_STAGES], Co[BF_MAX_STAGES], gh[BF_MAX_STAGES], gw[BF_MAX_STAGES]; long Pin[BF_MAX_STAGES];
    int Np;
    size_t bytes;
    DebedSaved(const D& d, void* base) {
        Arena a(base);
        lossbuf = a.f32((size_t)d.F * d.cout * 2 * 2 * BF_LOSS_LIMBS); coef = a.f32((size_t)d.F * d.cout);      // [F][Co][2][limbs] int64
        Np = roundup(4 * d.cout, 8);
        for (int i = 0; i < d.nst; ++i) {
            Cin[i] = i == 0 ? d.E : d.E / 4;
            Co[i] = (i == d.nst - 1) ? d.cout : d.E / 4;
            gh[i] = d.h << i; gw[i] = d.w << i;
            Pin[i] = d.F * gh[i] * gw[i];
            const bool last = i == d.nst - 1;
            wc[i] = a.take((size_t)Cin[i] * (last ? Np : 4 * Co[i]) * d.es);
            if (!last) {
                y[i] = a.take((size_t)Pin[i] * 4 * Co[i] * d.es);
                const size_t fc = (size_t)d.F * Co[i];
                mean[i] = a.f32(fc); rstd[i] = a.f32(fc); sc[i] = a.f32(fc); sh[i] = a.f32(fc);
            } else { y[i] = nullptr; mean[i] = rstd[i] = sc[i] = sh[i] = nullptr; }
        }
        bytes = a.off;
    }
};
}  // namespace

extern "C" int64_t bf_embed_saved_bytes(const bf_dims* s) { D d; if (get_dims(s, &d) || d.nst < 1) return -1; return (int64_t)EmbedSaved(d, nullptr).bytes; }
extern "C" int64_t bf_debed_saved_bytes(const bf_dims* s) { D d; if (get_dims(s, &d) || d.nst < 1) return -1; return (int64_t)DebedSaved(d, nullptr).bytes; }

// saved records whose stage-0 map was NOT stored by the forward (host-side memory of a per-call decision; the record itself is device memory)
namespace {
BfPerDevice<std::vector<const void*>> g_embed_lean;
void embed_lean_set(const void* saved, bool lean) {
    for (size_t i = 0; i < g_embed_lean.get().size(); ++i)
        if (g_embed_lean.get()[i] == saved) { if (!lean) { g_embed_lean.get()[i] = g_embed_lean.get().back(); g_embed_lean.get().pop_back(); } return; }
    if (lean) g_embed_lean.get().push_back(saved);
}
bool embed_lean_get(const void* saved) {
    for (const void* p : g_embed_lean.get()) if (p == saved) return true;
    return false;
}
}  // namespace
extern "C" int bf_embed_fwd(const bf_dims* dims, const bf_embed_params* p, const float* x, const float* fluid, void* out, void* saved,
                            void* scratch, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(p && x && out && saved && scratch && d.nst >= 1 && d.cin >= 1, "bf_embed_fwd: bad arguments");
    BF_REQUIRE((d.nfluid > 0) == (fluid != nullptr), "bf_embed_fwd: fluid parameters must be given exactly when nfluid > 0");
    hipStream_t st = (hipStream_t)s;
    links_clear(true);
    TRY(side_join_pending(st));
    EmbedSaved sv(d, saved);
    Scratch sc(d, scratch);
    const int n = d.nst, H = d.h * d.patch, W = d.w * d.patch;
    if (d.nfluid > 0)
        TRY(bf_film_net_fwd(fluid, p->film_ln_w, p->film_ln_b, p->film_w, p->film_b, sv.gb, sv.chat, sv.crstd, d.B, d.nfluid, 2 * d.E, st));
    bool stats_done = false;
    {   // every stage's convolution weight in GEMM layout / compute dtype: one launch
        int mode[BF_MAX_STAGES], R[BF_MAX_STAGES], K[BF_MAX_STAGES], Kp[BF_MAX_STAGES];
        const float* src[BF_MAX_STAGES]; void* dst[BF_MAX_STAGES];
        for (int i = 0; i < n; ++i) {
            src[i] = p->conv_w[i]; dst[i] = sv.wc[i]; mode[i] = i == 0 ? 0 : 1; R[i] = sv.C[i];
            K[i] = i == 0 ? 4 * d.cin : 4 * sv.C[i - 1]; Kp[i] = i == 0 ? sv.Kp : K[i];
        }
        TRY(bf_wprep_multi(d.dtype, n, mode, src, dst, R, K, Kp, st));
    }
    for (int i = 0; i < n; ++i) {
        const void* wc;
        if (i == 0) {
            wc = sv.wc[0];
            // patch rows and the K = 16 contraction in one streaming pass where it applies, else im2col + GEMM
            // ... which also leaves the InstanceNorm slice partials of its output (no second read of the 226 MB map for the statistics)
            const int S0 = sv.gh[0] * sv.gw[0];
            static const bool part_on = bf_knob("BF_EMBED_STATS", 1) != 0;
            const bool part_ok = part_on && n > 1 && bf_in_ws_floats(d.dtype, (int)d.F, S0, sv.C[0]) >= (int64_t)2 * d.F * sv.C[0] * (1 + (S0 + 255) / 256);      // the sliced workspace holds 256-row slices
            // lean: the stage-0 map is W0 . patch -- when every consumer of this call's saved record can rebuild its rows (the streaming
            // stage-1 kernels, the one-pass backward tail) it is not stored at all; g_embed_lean.get() remembers the decision for the backward
            static const bool lean_on = bf_knob("BF_EMBED_LEAN", 1) != 0;
            // ... and only when the BACKWARD kernels that rebuild the rows will take this frame count with the workspaces this call's scratch holds
            // (the one-pass tail's partials live in the token-reduction workspace, the rebuilt-rows weight gradient's slabs in t1b: a batch of
            // 23+ clips of 16 frames at 192 x 192 exceeds the first): otherwise the map is stored and the generic chain runs, as before
            const int64_t tail_need = n > 1 ? bf_embed_tail_ws_floats((int)d.F, sv.gh[1], sv.gw[1], sv.C[0], sv.Kp) : 0;
            const bool bwd_fits = tail_need > 0 && tail_need + (int64_t)sv.C[0] * sv.Kp <= sc.tokred_floats && d.F <= 512 &&
                                  (int64_t)d.F * (4 * 96 * 96) <= sc.t1b_floats;
            const bool lean = lean_on && part_ok && bwd_fits && d.dtype == BF_DTYPE_BF16 && sv.Kp == 16 && d.cin <= 4 && sv.C[0] == 96 && sv.C[1] == 96 && (W / 2) % 16 == 0 &&
                              sv.gw[1] % 16 == 0 && ((long)sv.gh[1] * sv.gw[1]) % 128 == 0 && S0 >= 1024;
            const int rc = bf_embed_first(d.dtype, x, wc, sv.patches, lean ? nullptr : sv.y[0], (int)d.F, sv.C[0], d.cin, H / 2, W / 2, sv.Kp,
                                          part_ok ? sc.in_ws + (size_t)2 * d.F * sv.C[0] : nullptr, st);
            if (rc < 0) return rc;
            if (rc == 1 && lean) return bf_fail_msg("bf_embed_fwd: the first-stage kernel declined a shape the lean path was chosen for", __FILE__, __LINE__);
            embed_lean_set(saved, lean);
            stats_done = rc == 0 && part_ok;
            if (rc == 1) {
                TRY(bf_im2col_nchw(d.dtype, x, sv.patches, (int)d.F, d.cin, H, W, sv.Kp, st));
                bf_operand A = op_plain(sv.patches, sv.Kp, BF_LAY_KC);
                bf_operand Bo = op_plain(wc, sv.Kp, BF_LAY_KC);
                bf_epilogue e = epi_store(sv.y[0], sv.C[0]);
                TRY(bf_gemm(d.dtype, (int)sv.P[0], sv.C[0], sv.Kp, &A, &Bo, &e, 1, st));
            }
        } else {
            const int cp = sv.C[i - 1];
            // the 96 -> 96 channel stages stream their map once through a weight-stationary kernel (gather_gemm.hip)
            const bool reb = i == 1 && embed_lean_get(saved);
            const int grc = reb ? bf_gather_gemm_rebuilt(d.dtype, sv.patches, sv.wc[0], sv.wc[i], 0, sv.sc[0], sv.sh[0], sv.y[i], (int)d.F, sv.gh[i], sv.gw[i], cp, sv.C[i], st)
                                : bf_gather_gemm(d.dtype, sv.y[i - 1], sv.wc[i], 0, sv.sc[i - 1], sv.sh[i - 1], sv.y[i], (int)d.F, sv.gh[i], sv.gw[i], cp, sv.C[i], st);
            if (grc < 0) return grc;
            if (grc == 1 && reb) return bf_fail_msg("bf_embed_fwd: the rebuilt-rows stage kernel declined a shape the lean path was chosen for", __FILE__, __LINE__);
            if (grc == 1) {
                bf_operand A = op_plain(sv.y[i - 1], cp, BF_LAY_KC);
                op_gather(A, sv.gw[i], sv.gh[i], cp);
                op_affine(A, BF_PRO_AFFINE_GELU, sv.sc[i - 1], sv.sh[i - 1], (long)sv.gh[i] * sv.gw[i], cp);
                bf_operand Bo = op_plain(sv.wc[i], 4L * cp, BF_LAY_KC);
                bf_epilogue e = epi_store(sv.y[i], sv.C[i]);
                TRY(bf_gemm(d.dtype, (int)sv.P[i], sv.C[i], 4 * cp, &A, &Bo, &e, 1, st));
            }
        }
        const bool last = i == n - 1;
        const bool film = last && d.nfluid > 0;
        if (i == 0 && stats_done) {
            const int mrc = bf_in_stats_merge_slices(d.dtype, (int)d.F, sv.gh[0] * sv.gw[0], sv.C[0], 256, p->in_w[0], p->in_b[0], nullptr, 1, nullptr,
                                                     sv.mean[0], sv.rstd[0], sv.sc[0], sv.sh[0], sc.in_ws, st);
            if (mrc < 0) return mrc;
            if (mrc == 0) continue;
            if (embed_lean_get(saved)) return bf_fail_msg("bf_embed_fwd: slice statistics declined on the lean path", __FILE__, __LINE__);
        }
        if (last) {       // the tokens (InstanceNorm affine, FiLM folded in) leave the statistics kernel itself where a frame fits its registers
            TRY(bf_in_stats_apply(d.dtype, sv.y[i], (int)d.F, sv.gh[i] * sv.gw[i], sv.C[i], p->in_w[i], p->in_b[i], film ? sv.gb : nullptr, d.T,
                                  film ? sv.gb + (size_t)d.B * d.E : nullptr, sv.mean[i], sv.rstd[i], sv.sc[i], sv.sh[i], sc.in_ws, nullptr, out, st));
            break;
        }
        TRY(bf_in_stats(d.dtype, sv.y[i], (int)d.F, sv.gh[i] * sv.gw[i], sv.C[i], p->in_w[i], p->in_b[i], film ? sv.gb : nullptr, d.T,
                        film ? sv.gb + (size_t)d.B * d.E : nullptr, sv.mean[i], sv.rstd[i], sv.sc[i], sv.sh[i], sc.in_ws, st));
    }
    return 0;
}

extern "C" int bf_embed_bwd(const bf_dims* dims, const bf_embed_params* p, const bf_embed_params* g, const void* dout, float* dx_in,
                            void* saved, void* scratch, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(p && g && dout && saved && scratch && d.nst >= 1, "bf_embed_bwd: bad arguments");
    hipStream_t st = (hipStream_t)s;
    TRY(side_join_pending(st));
    EmbedSaved sv(d, saved);
    Scratch sc(d, scratch);
    const int n = d.nst, H = d.h * d.patch, W = d.w * d.patch;
    auto buf = [&](int stage) { return (stage & 1) ? sc.t3 : sc.t4; };
    const bool film = d.nfluid > 0;
    if (film) ZERO(sv.dgb, (size_t)2 * d.B * d.E * 4);
    // last stage: out = (xhat*w + b) * gamma_b + beta_b
    void* dy = buf(n - 1);
    TRY(bf_in_bwd(d.dtype, dout, sv.y[n - 1], nullptr, dy, (int)d.F, sv.gh[n - 1] * sv.gw[n - 1], sv.C[n - 1], sv.mean[n - 1], sv.rstd[n - 1],
                  p->in_w[n - 1], p->in_b[n - 1], film ? sv.gb : nullptr, d.T, 0, g->in_w[n - 1], g->in_b[n - 1], film ? sv.dgb : nullptr,
                  film ? sv.dgb + (size_t)d.B * d.E : nullptr, sc.in_ws, st));
    if (film)
        TRY(bf_film_net_bwd(sv.dgb, sv.chat, p->film_ln_w, p->film_ln_b, p->film_w, g->film_w, g->film_b, g->film_ln_w, g->film_ln_b, d.B,
                            d.nfluid, 2 * d.E, st));
    // Weight gradients (memset, split-K GEMM into the prepared-layout scratch, un-prepare into the gradient) run on the side
    // stream while this stream continues with the data gradient and the InstanceNorm backward of the same stage.  The side work
    // of stage i is joined before stage i-1 forks: the two ping-pong gradient buffers and sc.wg are then never recycled under it.
    Fork fk(st);
    hipStream_t ss;
    for (int i = n - 1; i >= 1; --i) {
        const int cp = sv.C[i - 1], K4 = 4 * cp;
        const long rpf = (long)sv.gh[i] * sv.gw[i];
        TRY(fk.join());
        TRY(fk.begin(&ss));
        // dWprep[co][k] = sum_p dy[p][co] * act(patch)[p][k]: the 96-channel stages as one stream over the map with slabs summed in a fixed
        // order (gather_gemm.hip; its slabs live in t1b, which nothing else of this call touches), else split-K with fp32 atomics
        const bool lean = embed_lean_get(saved);
        if (i == 1 && lean && dx_in) {      // the input wants a gradient after all: the generic chain below reads the map, so store it now (y0 = patches @ W0^T)
            bf_operand A0 = op_plain(sv.patches, sv.Kp, BF_LAY_KC);
            bf_operand B0 = op_plain(sv.wc[0], sv.Kp, BF_LAY_KC);
            bf_epilogue e0 = epi_store(sv.y[0], sv.C[0]);
            TRY(bf_gemm(d.dtype, (int)sv.P[0], sv.C[0], sv.Kp, &A0, &B0, &e0, 1, st));
            embed_lean_set(saved, false);
            TRY(fk.join());                 // the side stream forked before the map existed
            TRY(fk.begin(&ss));
        }
        const bool reb = i == 1 && embed_lean_get(saved);
        const int wrc = reb ? bf_gather_wgrad_rebuilt(d.dtype, sv.patches, sv.wc[0], dy, sv.sc[0], sv.sh[0], sc.wg, 1, (int)d.F, sv.gh[i], sv.gw[i], cp, sv.C[i],
                                                      (float*)sc.t1b, sc.t1b_floats, ss)
                            : bf_gather_wgrad(d.dtype, sv.y[i - 1], dy, sv.sc[i - 1], sv.sh[i - 1], nullptr, nullptr, sc.wg, 1, (int)d.F, sv.gh[i], sv.gw[i], cp,
                                              sv.C[i], (float*)sc.t1b, sc.t1b_floats, ss);
        if (wrc < 0) return wrc;
        if (wrc == 1 && reb) return bf_fail_msg("bf_embed_bwd: the rebuilt-rows weight gradient declined a shape the lean path was chosen for", __FILE__, __LINE__);
        if (wrc == 1) {
            bf_operand A = op_plain(dy, sv.C[i], BF_LAY_XC);
            bf_operand Bo = op_plain(sv.y[i - 1], cp, BF_LAY_XC);
            op_gather(Bo, sv.gw[i], sv.gh[i], cp);
            op_affine(Bo, BF_PRO_AFFINE_GELU, sv.sc[i - 1], sv.sh[i - 1], rpf, cp);
            // split-K into per-slice images summed in order (no float atomics on shared addresses: the same bits every run); atomics only
            // where the workspace cannot hold the images
            const int src = bf_gemm_slabs(d.dtype, sv.C[i], K4, (int)sv.P[i], &A, &Bo, sc.wg, K4, 0, splitk_for(sv.C[i], K4, sv.P[i]), (float*)sc.t1b, sc.t1b_floats, ss);
            if (src < 0) return src;
            if (src == 1) {
                ZERO_ON(ss, sc.wg, (size_t)sv.C[i] * K4 * 4);
                bf_epilogue e = epi_atomic(sc.wg, K4);
                TRY(bf_gemm(d.dtype, sv.C[i], K4, (int)sv.P[i], &A, &Bo, &e, splitk_for(sv.C[i], K4, sv.P[i]), ss));
            }
        }
        TRY(bf_wgrad_unprep(1, sc.wg, g->conv_w[i], sv.C[i], K4, K4, 0, ss));
        if (i == 1 && !dx_in) {
            // Nothing but sums over pixels is wanted behind this stage's data gradient (GELU', the stage-0 InstanceNorm backward, the
            // stage-0 weight gradient): one pass that keeps the gradient map in registers (embed_tail.hip).  Its partials and the
            // prepared-layout gradient live in the token-reduction workspace, which no side-stream kernel of this call touches.
            static const bool tail_on = bf_knob("BF_EMBED_TAIL", 1) != 0;
            const int64_t need = bf_embed_tail_ws_floats((int)d.F, sv.gh[1], sv.gw[1], cp, sv.Kp);
            if (tail_on && need > 0 && need + (int64_t)cp * sv.Kp <= sc.tokred_floats) {
                float* dwprep = sc.tokred_ws + need;
                static const bool tail_map = bf_knob("BF_EMBED_TAIL_MAP", 0) != 0;      // 1: read the stored stage-0 map instead of rebuilding its rows
                const int trc = bf_embed_tail_bwd(d.dtype, dy, sv.wc[1], (tail_map && !embed_lean_get(saved)) ? sv.y[0] : nullptr, sv.patches, sv.wc[0], sv.sc[0], sv.sh[0], sv.mean[0], sv.rstd[0],
                                                  p->in_w[0], dwprep, g->in_w[0], g->in_b[0], (int)d.F, sv.gh[1], sv.gw[1], sv.C[1], cp, sv.Kp,
                                                  sc.tokred_ws, need, s);
                if (trc < 0) return trc;
                if (trc == 0) {
                    TRY(bf_wgrad_unprep(0, dwprep, g->conv_w[0], sv.C[0], 4 * d.cin, sv.Kp, 0, st));
                    return fk.join();
                }
            }
            if (embed_lean_get(saved)) return bf_fail_msg("bf_embed_bwd: the one-pass tail declined on the lean path (no stored stage-0 map)", __FILE__, __LINE__);
        }
        void* dact = buf(i - 1);
        {   // d(act patch)[p][k] = sum_co dy[p][co] * Wprep[co][k], scattered back to the input grid
            const int src = bf_scatter_gemm(d.dtype, dy, sv.wc[i], 1, nullptr, nullptr, dact, nullptr, (int)d.F, sv.gh[i], sv.gw[i], sv.C[i], cp, st);
            if (src < 0) return src;
            if (src == 1) {
                bf_operand A = op_plain(dy, sv.C[i], BF_LAY_KC);
                bf_operand Bo = op_plain(sv.wc[i], K4, BF_LAY_XC);
                bf_epilogue e = epi_store(dact, cp);
                epi_scatter(e, sv.gw[i], sv.gh[i], cp);
                TRY(bf_gemm(d.dtype, (int)sv.P[i], K4, sv.C[i], &A, &Bo, &e, 1, st));
            }
        }
        TRY(bf_in_bwd(d.dtype, dact, sv.y[i - 1], nullptr, dact, (int)d.F, sv.gh[i - 1] * sv.gw[i - 1], cp, sv.mean[i - 1], sv.rstd[i - 1],
                      p->in_w[i - 1], p->in_b[i - 1], nullptr, 1, 1, g->in_w[i - 1], g->in_b[i - 1], nullptr, nullptr, sc.in_ws, st));
        dy = dact;
    }
    {   // stage 0
        TRY(fk.join());
        if (dx_in) TRY(fk.begin(&ss)); else ss = st;      // nothing left to overlap with when the input needs no gradient
        // dWprep[co][k] = sum_p dy[p][co] * patch[p][k]: a 16-wide stream where it applies (the LAST kernel of the step: nothing to hide behind)
        const int nrc = sv.Kp == 16 ? bf_tokred_narrow(d.dtype, sv.C[0], sv.P[0], dy, sv.patches, sc.wg, sv.Kp, 0, 0, nullptr, nullptr, 0, sc.tokred_ws, sc.tokred_floats, ss) : 1;
        if (nrc < 0) return nrc;
        if (nrc == 1) {
            ZERO_ON(ss, sc.wg, (size_t)sv.C[0] * sv.Kp * 4);
            bf_operand A = op_plain(dy, sv.C[0], BF_LAY_XC);
            bf_operand Bo = op_plain(sv.patches, sv.Kp, BF_LAY_XC);
            bf_epilogue e = epi_atomic(sc.wg, sv.Kp);
            TRY(bf_gemm(d.dtype, sv.C[0], sv.Kp, (int)sv.P[0], &A, &Bo, &e, splitk_for(sv.C[0], sv.Kp, sv.P[0]), ss));
        }
        TRY(bf_wgrad_unprep(0, sc.wg, g->conv_w[0], sv.C[0], 4 * d.cin, sv.Kp, 0, ss));
        if (dx_in) {
            void* dpatch = sc.t1;
            bf_operand A2 = op_plain(dy, sv.C[0], BF_LAY_KC);
            bf_operand B2 = op_plain(sv.wc[0], sv.Kp, BF_LAY_XC);
            bf_epilogue e2 = epi_store(dpatch, sv.Kp);
            TRY(bf_gemm(d.dtype, (int)sv.P[0], sv.Kp, sv.C[0], &A2, &B2, &e2, 1, st));
            TRY(bf_col2im_nchw(d.dtype, dpatch, dx_in, (int)d.F, d.cin, H, W, sv.Kp, st));
        }
    }
    return fk.join();
}

// ================================================================================================= debed (+ relative-L2 loss)
extern "C" int bf_debed_fwd(const bf_dims* dims, const bf_debed_params* p, const void* x, float* pred, const float* target, float* loss,
                            void* saved, void* scratch, bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(p && x && pred && saved && scratch && d.nst >= 1 && d.cout >= 1, "bf_debed_fwd: bad arguments");
    BF_REQUIRE(!target || loss, "bf_debed_fwd: loss output missing");
    hipStream_t st = (hipStream_t)s;
    TRY(side_join_pending(st));
    DebedSaved sv(d, saved);
    Scratch sc(d, scratch);
    const int n = d.nst;
    {   // every stage's transposed-convolution weight in GEMM layout / compute dtype: one launch
        int mode[BF_MAX_STAGES], R[BF_MAX_STAGES], K[BF_MAX_STAGES], Kp[BF_MAX_STAGES];
        const float* src[BF_MAX_STAGES]; void* dst[BF_MAX_STAGES];
        for (int i = 0; i < n; ++i) {
            const bool last = i == n - 1;
            src[i] = p->conv_w[i]; dst[i] = sv.wc[i]; mode[i] = last ? 0 : 2;
            R[i] = last ? sv.Cin[i] : 4 * sv.Co[i]; K[i] = last ? 4 * sv.Co[i] : sv.Cin[i]; Kp[i] = last ? sv.Np : sv.Cin[i];
        }
        TRY(bf_wprep_multi(d.dtype, n, mode, src, dst, R, K, Kp, st));
    }
    for (int i = 0; i < n; ++i) {
        const bool last = i == n - 1;
        const int cin = sv.Cin[i], co = sv.Co[i];
        bf_operand A = op_plain(i == 0 ? x : sv.y[i - 1], cin, BF_LAY_KC);
        if (i > 0) op_affine(A, BF_PRO_AFFINE_GELU, sv.sc[i - 1], sv.sh[i - 1], (long)sv.gh[i] * sv.gw[i], cin);
        if (!last) {
            // the 96 -> 4 x 96 channel stages: one streaming kernel that also leaves the InstanceNorm slice partials of the map it writes
            // (gather_gemm.hip); the statistics then need no second pass over the map
            const int S4 = 4 * sv.gh[i] * sv.gw[i];
            const bool part_ok = i > 0 && S4 % 128 == 0 &&
                                 bf_in_ws_floats(d.dtype, (int)d.F, S4, co) >= (int64_t)2 * d.F * co * (1 + S4 / 128);
            const int src = i > 0 ? bf_scatter_gemm(d.dtype, sv.y[i - 1], sv.wc[i], 0, sv.sc[i - 1], sv.sh[i - 1], sv.y[i],
                                                    part_ok ? sc.in_ws + (size_t)2 * d.F * co : nullptr, (int)d.F, sv.gh[i], sv.gw[i], cin, co, st) : 1;
            if (src < 0) return src;
            if (src == 0 && part_ok) {
                const int mrc = bf_in_stats_merge_slices(d.dtype, (int)d.F, S4, co, 128, p->in_w[i], p->in_b[i], nullptr, 1, nullptr, sv.mean[i], sv.rstd[i],
                                                         sv.sc[i], sv.sh[i], sc.in_ws, st);
                if (mrc < 0) return mrc;
                if (mrc == 0) continue;
            }
            if (src == 1) {
                bf_operand Bo = op_plain(sv.wc[i], cin, BF_LAY_KC);
                bf_epilogue e = epi_store(sv.y[i], co);
                epi_scatter(e, sv.gw[i], sv.gh[i], co);
                TRY(bf_gemm(d.dtype, (int)sv.Pin[i], 4 * co, cin, &A, &Bo, &e, 1, st));
            }
            TRY(bf_in_stats(d.dtype, sv.y[i], (int)d.F, S4, co, p->in_w[i], p->in_b[i], nullptr, 1, nullptr, sv.mean[i],
                            sv.rstd[i], sv.sc[i], sv.sh[i], sc.in_ws, st));
        } else {
            if (target) ZERO(sv.lossbuf, (size_t)d.F * d.cout * 2 * BF_LOSS_LIMBS * 8);
            // InstanceNorm affine + GELU + the 2x2 transposed convolution + NCHW store + loss partials in one streaming pass where it applies
            const int rc = i > 0 ? bf_debed_last(d.dtype, sv.y[i - 1], sv.sc[i - 1], sv.sh[i - 1], sv.wc[i], pred, target, sv.lossbuf, (int)d.F, cin, co,
                                                 sv.gh[i], sv.gw[i], sv.Np, st) : 1;
            if (rc < 0) return rc;
            if (rc == 1) {
                bf_operand Bo = op_plain(sv.wc[i], sv.Np, BF_LAY_XC);
                float* pm = (float*)sc.t4;
                bf_epilogue e = epi_store(pm, sv.Np);
                e.out_mode = BF_OUT_STORE_F32;
                TRY(bf_gemm(d.dtype, (int)sv.Pin[i], sv.Np, cin, &A, &Bo, &e, 1, st));
                TRY(bf_pm2nchw(pm, pred, target, sv.lossbuf, (int)d.F, co, sv.gh[i], sv.gw[i], sv.Np, st));
            }
            if (target) TRY(bf_lploss_finalize(sv.lossbuf, (int)d.F, co, loss, sv.coef, st));
        }
    }
    return 0;
}

extern "C" int bf_debed_bwd(const bf_dims* dims, const bf_debed_params* p, const bf_debed_params* g, const void* x, const float* dpred,
                            const float* pred, const float* target, const float* loss_scale, void* dx, void* saved, void* scratch,
                            bf_stream_t s) {
    D d; TRY(get_dims(dims, &d));
    BF_REQUIRE(p && g && x && dx && saved && scratch && d.nst >= 1, "bf_debed_bwd: bad arguments");
    BF_REQUIRE(dpred || (pred && target), "bf_debed_bwd: need dpred or (pred, target) of the fused loss");
    hipStream_t st = (hipStream_t)s;
    links_clear(false);
    TRY(side_join_pending(st));
    DebedSaved sv(d, saved);
    Scratch sc(d, scratch);
    const int n = d.nst;
    auto buf = [&](int stage) { return (stage & 1) ? sc.t3 : sc.t4; };   // gradient w.r.t. the INPUT of `stage`
    void* dy = nullptr;   // gradient w.r.t. the raw output of stage i-1 == (after IN/GELU backward) input of stage i
    Fork fk(st);          // weight gradients on the side stream, joined before the next stage forks (see bf_embed_bwd)
    hipStream_t ss;
    for (int i = n - 1; i >= 0; --i) {
        const bool last = i == n - 1;
        const int cin = sv.Cin[i], co = sv.Co[i];
        const long rpf = (long)sv.gh[i] * sv.gw[i];
        const void* ain = i == 0 ? x : sv.y[i - 1];
        void* dact = i == 0 ? dx : buf(i);
        bool normed = false;          // dact already holds the gradient of the raw map in front of stage i-1's InstanceNorm
        if (last) {
            void* dpm = sc.t1;
            // ... together with the InstanceNorm + GELU backward of the stage in front where that applies: the full-resolution gradient map
            // has rank 16 and is never stored (patch.hip, debed_last_inbwd_kernel)
            if (i > 0) {
                const int nrc = bf_debed_last_bwd_norm(d.dtype, dpred, pred, target, sv.coef, loss_scale, sv.wc[i], dpm, ain, sv.mean[i - 1], sv.rstd[i - 1],
                                                       p->in_w[i - 1], p->in_b[i - 1], dact, g->in_w[i - 1], g->in_b[i - 1], (int)d.F, cin, co, sv.gh[i],
                                                       sv.gw[i], sv.Np, sc.in_ws, bf_in_ws_floats(d.dtype, (int)d.F, (int)rpf, cin), st);
                if (nrc < 0) return nrc;
                normed = nrc == 0;
            }
            // the loss gradient in patch-major rows and the data gradient of the transposed convolution in one pass where it applies
            const int rc = normed ? 0 : bf_debed_last_bwd(d.dtype, dpred, pred, target, sv.coef, loss_scale, sv.wc[i], dpm, dact, (int)d.F, cin, co, sv.gh[i], sv.gw[i], sv.Np, st);
            if (rc < 0) return rc;
            if (rc == 1) TRY(bf_nchw2pm(d.dtype, dpred, pred, target, sv.coef, loss_scale, dpm, (int)d.F, co, sv.gh[i], sv.gw[i], sv.Np, st));
            TRY(fk.begin(&ss));
            // wg[n][ci] = sum_p dpm[p][n] * act[p][ci]: the 16-wide stream (transposed output, InstanceNorm + GELU applied to the map's
            // fragments in registers) where it applies
            const int nrc = (sv.Np == 16 && i > 0) ? bf_tokred_narrow(d.dtype, cin, sv.Pin[i], ain, dpm, sc.wg, cin, 0, 1, sv.sc[i - 1], sv.sh[i - 1], rpf,
                                                                      sc.tokred_ws, sc.tokred_floats, ss) : 1;
            if (nrc < 0) return nrc;
            if (nrc == 1) {
                ZERO_ON(ss, sc.wg, (size_t)sv.Np * cin * 4);
                bf_operand A = op_plain(dpm, sv.Np, BF_LAY_XC);
                bf_operand Bo = op_plain(ain, cin, BF_LAY_XC);
                if (i > 0) op_affine(Bo, BF_PRO_AFFINE_GELU, sv.sc[i - 1], sv.sh[i - 1], rpf, cin);
                bf_epilogue e = epi_atomic(sc.wg, cin);
                TRY(bf_gemm(d.dtype, sv.Np, cin, (int)sv.Pin[i], &A, &Bo, &e, splitk_for(sv.Np, cin, sv.Pin[i]), ss));
            }
            TRY(bf_wgrad_unprep(0, sc.wg, g->conv_w[i], cin, 4 * co, sv.Np, 1, ss));
            if (rc == 1) {   // dact[p][ci] = sum_n dpm[p][n] * wt[ci][n]
                bf_operand A = op_plain(dpm, sv.Np, BF_LAY_KC);
                bf_operand Bo = op_plain(sv.wc[i], sv.Np, BF_LAY_KC);
                bf_epilogue e = epi_store(dact, cin);
                TRY(bf_gemm(d.dtype, (int)sv.Pin[i], cin, sv.Np, &A, &Bo, &e, 1, st));
            }
        } else {
            const int N4 = 4 * co;
            TRY(fk.join());
            TRY(fk.begin(&ss));
            // wg[(q,co)][ci] = sum_p dy_gathered[p][(q,co)] * act[p][ci]: as in bf_embed_bwd, the transformed side here being the coarse rows
            const int wrc = i > 0 ? bf_gather_wgrad(d.dtype, dy, ain, nullptr, nullptr, sv.sc[i - 1], sv.sh[i - 1], sc.wg, 0, (int)d.F, sv.gh[i], sv.gw[i], co,
                                                    cin, (float*)sc.t1b, sc.t1b_floats, ss) : 1;
            if (wrc < 0) return wrc;
            if (wrc == 1) {
                bf_operand A = op_plain(dy, co, BF_LAY_XC);
                op_gather(A, sv.gw[i], sv.gh[i], co);
                bf_operand Bo = op_plain(ain, cin, BF_LAY_XC);
                if (i > 0) op_affine(Bo, BF_PRO_AFFINE_GELU, sv.sc[i - 1], sv.sh[i - 1], rpf, cin);
                const int src = bf_gemm_slabs(d.dtype, N4, cin, (int)sv.Pin[i], &A, &Bo, sc.wg, cin, 0, splitk_for(N4, cin, sv.Pin[i]), (float*)sc.t1b, sc.t1b_floats, ss);      // (see bf_embed_bwd)
                if (src < 0) return src;
                if (src == 1) {
                    ZERO_ON(ss, sc.wg, (size_t)N4 * cin * 4);
                    bf_epilogue e = epi_atomic(sc.wg, cin);
                    TRY(bf_gemm(d.dtype, N4, cin, (int)sv.Pin[i], &A, &Bo, &e, splitk_for(N4, cin, sv.Pin[i]), ss));
                }
            }
            TRY(bf_wgrad_unprep(2, sc.wg, g->conv_w[i], N4, cin, cin, 0, ss));
            {
                const int grc = bf_gather_gemm(d.dtype, dy, sv.wc[i], 1, nullptr, nullptr, dact, (int)d.F, sv.gh[i], sv.gw[i], co, cin, st);
                if (grc < 0) return grc;
                if (grc == 1) {
                    bf_operand A = op_plain(dy, co, BF_LAY_KC);
                    op_gather(A, sv.gw[i], sv.gh[i], co);
                    bf_operand Bo = op_plain(sv.wc[i], cin, BF_LAY_XC);
                    bf_epilogue e = epi_store(dact, cin);
                    TRY(bf_gemm(d.dtype, (int)sv.Pin[i], cin, N4, &A, &Bo, &e, 1, st));
                }
            }
        }
        if (i > 0) {
            if (!normed)
                TRY(bf_in_bwd(d.dtype, dact, sv.y[i - 1], nullptr, dact, (int)d.F, (int)rpf, cin, sv.mean[i - 1], sv.rstd[i - 1], p->in_w[i - 1],
                              p->in_b[i - 1], nullptr, 1, 1, g->in_w[i - 1], g->in_b[i - 1], nullptr, nullptr, sc.in_ws, st));
            dy = dact;
        }
    }
    return fk.join();
}
