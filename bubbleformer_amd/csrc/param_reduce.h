// Parameter-gradient reductions of a stage backward, as device functions that either run as their own small kernels (the C ABI
// entry points bf_in_bwd / bf_attn_bwd) or all together in ONE launch at the end of a stage (model.hip: stage_param_reduce_kernel).
// Every small dependent launch costs ~5 us of critical path; a FiLMAViT block has eight of these reductions.
#pragma once
#include "bf_common.h"

// InstanceNorm: ws[f][c] = {s1, s2} per frame (in_bwd_kernel).  dw += sum_f g*s2, db += sum_f g*s1 (g = 1 or g[f / gdiv][c]);
// optional per-group outputs dg[f / gdiv][c] += w*S2 + b*S1, dgb += S1 (then a block covers exactly one group);
// optional masked fold dmask_v[c] += sum_f mask[f] * (w*s2 + b*s1)   (stochastic depth: g[f][c] = mask[f] * v[c], this is dv).
struct InReduceJob {
    const float* ws; int frames, C; const float *w, *b, *g; int gdiv; float *dw, *db, *dg, *dgb; const float* mask; float* dmask_v;
    // frames per workgroup: one group where per-group outputs are asked for, otherwise ALL frames -- a single workgroup per 64 channels sums
    // them in a fixed order (with 16-frame slices meeting in float atomics the small parameter gradients differed from run to run)
    __host__ __device__ int rdiv() const { return (dg || dgb) ? (gdiv > 0 ? gdiv : 1) : (frames > 0 ? frames : 1); }
};
// attention: ws[row][nvals] per workgroup of attn_bwd_mfma, nvals = 4*D + 32*heads + heads
struct AttnReduceJob { const float* ws; int rows, D, heads; float *dqw, *dqb, *dkw, *dkb, *demb, *dhscale; };

// grid (ceil(C/64), ceil(frames / rdiv)), 64 * FL threads = 64 channels x FL frame lanes (FL = 16: with one workgroup per column block
// walking all frames a thread takes 8 of 128 frames in one batch of loads)
constexpr int BF_RED_FL = 16;
template <int FL>
__device__ __forceinline__ void in_reduce_block(const InReduceJob& j, int bx, int by, float (*red)[FL][64]) {
    const int l = threadIdx.x & 63, fl = threadIdx.x >> 6;
    const int c = bx * 64 + l;
    const bool cv = c < j.C;
    const int rd = j.rdiv(), gdiv = j.gdiv > 0 ? j.gdiv : 1;
    const int f0 = by * rd, f1 = min(j.frames, f0 + rd);
    float t1 = 0.f, t2 = 0.f, g1 = 0.f, g2 = 0.f, sm = 0.f;
    const float wc = cv ? j.w[c] : 0.f, bc = cv ? j.b[c] : 0.f;
    if (cv) {
        // eight frames' loads in flight per thread (a load-then-add loop pays a memory round trip per frame: with one workgroup per column
        // block walking ALL frames that was 30 round trips on the caller's queue); the adds stay in frame order
        int f = f0 + fl;
        for (; f + 7 * FL < f1; f += 8 * FL) {
            float2 p[8]; float gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p[u] = *reinterpret_cast<const float2*>(j.ws + ((long)(f + FL * u) * j.C + c) * 2);
                gv[u] = j.g ? j.g[(long)((f + FL * u) / gdiv) * j.C + c] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { t1 += p[u].x; t2 += p[u].y; g1 += gv[u] * p[u].x; g2 += gv[u] * p[u].y; }
        }
        for (; f < f1; f += FL) {
            const float a = j.ws[((long)f * j.C + c) * 2], b2 = j.ws[((long)f * j.C + c) * 2 + 1];
            const float gv = j.g ? j.g[(long)(f / gdiv) * j.C + c] : 1.f;
            t1 += a; t2 += b2; g1 += gv * a; g2 += gv * b2;
        }
    }
    // the masked fold is ONE sum over all frames per channel: the first workgroup of the column takes it whole (one writer, fixed order) --
    // and likewise the affine gradients of a job that is cut into groups for its per-group outputs
    const bool grouped = j.dg || j.dgb;
    if (cv && by == 0 && ((j.mask && j.dmask_v) || (grouped && (j.dw || j.db)))) {
        const bool redo = grouped && (j.dw || j.db), msk = j.mask && j.dmask_v;
        if (redo) g1 = g2 = 0.f;
        int f = fl;
        for (; f + 7 * FL < j.frames; f += 8 * FL) {
            float2 p[8]; float gv[8], mk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p[u] = *reinterpret_cast<const float2*>(j.ws + ((long)(f + FL * u) * j.C + c) * 2);
                gv[u] = j.g ? j.g[(long)((f + FL * u) / gdiv) * j.C + c] : 1.f;
                mk[u] = msk ? j.mask[f + FL * u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (msk) sm += mk[u] * (wc * p[u].y + bc * p[u].x);
                if (redo) { g1 += gv[u] * p[u].x; g2 += gv[u] * p[u].y; }
            }
        }
        for (; f < j.frames; f += FL) {
            const float a = j.ws[((long)f * j.C + c) * 2], b2 = j.ws[((long)f * j.C + c) * 2 + 1];
            if (msk) sm += j.mask[f] * (wc * b2 + bc * a);
            if (redo) { const float gv = j.g ? j.g[(long)(f / gdiv) * j.C + c] : 1.f; g1 += gv * a; g2 += gv * b2; }
        }
    }
    red[0][fl][l] = t1; red[1][fl][l] = t2; red[2][fl][l] = g1; red[3][fl][l] = g2; red[4][fl][l] = sm;
    __syncthreads();
    if (fl != 0 || !cv || f0 >= f1) return;
    float v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        v[q] = red[q][0][l];
#pragma unroll
        for (int k = 1; k < FL; ++k) v[q] += red[q][k][l];      // fixed order
    }
    if (j.dw && (!grouped || by == 0)) atomicAdd(j.dw + c, v[3]);
    if (j.db && (!grouped || by == 0)) atomicAdd(j.db + c, v[2]);
    const long gi = (long)(f0 / gdiv) * j.C + c;
    if (j.dg) j.dg[gi] += wc * v[1] + bc * v[0];
    if (j.dgb) j.dgb[gi] += v[0];
    if (j.dmask_v && by == 0) atomicAdd(j.dmask_v + c, v[4]);
}

// grid (ceil(nvals / 64), ny row slices), 256 threads = 64 values x 4 row lanes.  ny = 1 (what the library launches): one writer per value,
// fixed order; the atomic only keeps the add safe beside another launch's single addend on the same slot
template <int FL>
__device__ __forceinline__ void attn_reduce_block(const AttnReduceJob& j, int bx, int by, int ny, float (*red)[FL][64]) {
    const int nvals = 4 * j.D + 32 * j.heads + j.heads;
    const int l = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int i = bx * 64 + l;
    const int per = (j.rows + ny - 1) / ny;
    const int r0 = by * per, r1 = min(j.rows, r0 + per);
    float acc = 0.f;
    if (i < nvals) {
        int r = r0 + rg;
        for (; r + 15 * FL < r1; r += 16 * FL) {          // sixteen rows in flight per thread, added in row order
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = j.ws[(long)(r + FL * u) * nvals + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; r < r1; r += FL) acc += j.ws[(long)r * nvals + i];
    }
    red[0][rg][l] = acc;
    __syncthreads();
    if (rg != 0 || i >= nvals) return;
    acc = red[0][0][l];
#pragma unroll
    for (int k = 1; k < FL; ++k) acc += red[0][k][l];      // fixed order
    float* dst;
    if (i < 4 * j.D) { const int q = i / j.D, e = i % j.D; dst = (q == 0 ? j.dqw : q == 1 ? j.dqb : q == 2 ? j.dkw : j.dkb); if (dst) dst += e; }
    else if (i < 4 * j.D + 32 * j.heads) dst = j.demb ? j.demb + (i - 4 * j.D) : nullptr;
    else dst = j.dhscale ? j.dhscale + (i - 4 * j.D - 32 * j.heads) : nullptr;
    if (dst && acc != 0.f) atomicAdd(dst, acc);
}
