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

// grid (ceil(C/64), ceil(frames / rdiv)), 256 threads = 64 channels x 4 frame lanes
__device__ __forceinline__ void in_reduce_block(const InReduceJob& j, int bx, int by, float (*red)[4][64]) {
    const int l = threadIdx.x & 63, fl = threadIdx.x >> 6;
    const int c = bx * 64 + l;
    const bool cv = c < j.C;
    const int rd = j.rdiv(), gdiv = j.gdiv > 0 ? j.gdiv : 1;
    const int f0 = by * rd, f1 = min(j.frames, f0 + rd);
    float t1 = 0.f, t2 = 0.f, g1 = 0.f, g2 = 0.f, sm = 0.f;
    const float wc = cv ? j.w[c] : 0.f, bc = cv ? j.b[c] : 0.f;
    if (cv)
        for (int f = f0 + fl; f < f1; f += 4) {
            const float a = j.ws[((long)f * j.C + c) * 2], b2 = j.ws[((long)f * j.C + c) * 2 + 1];
            const float gv = j.g ? j.g[(long)(f / gdiv) * j.C + c] : 1.f;
            t1 += a; t2 += b2; g1 += gv * a; g2 += gv * b2;
        }
    // the masked fold is ONE sum over all frames per channel: the first workgroup of the column takes it whole (one writer, fixed order) --
    // and likewise the affine gradients of a job that is cut into groups for its per-group outputs
    const bool grouped = j.dg || j.dgb;
    if (cv && j.mask && j.dmask_v && by == 0)
        for (int f = fl; f < j.frames; f += 4)
            sm += j.mask[f] * (wc * j.ws[((long)f * j.C + c) * 2 + 1] + bc * j.ws[((long)f * j.C + c) * 2]);
    if (cv && grouped && by == 0 && (j.dw || j.db)) {
        g1 = g2 = 0.f;
        for (int f = fl; f < j.frames; f += 4) {
            const float gv = j.g ? j.g[(long)(f / gdiv) * j.C + c] : 1.f;
            g1 += gv * j.ws[((long)f * j.C + c) * 2]; g2 += gv * j.ws[((long)f * j.C + c) * 2 + 1];
        }
    }
    red[0][fl][l] = t1; red[1][fl][l] = t2; red[2][fl][l] = g1; red[3][fl][l] = g2; red[4][fl][l] = sm;
    __syncthreads();
    if (fl != 0 || !cv || f0 >= f1) return;
    float v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = red[q][0][l] + red[q][1][l] + red[q][2][l] + red[q][3][l];
    if (j.dw && (!grouped || by == 0)) atomicAdd(j.dw + c, v[3]);
    if (j.db && (!grouped || by == 0)) atomicAdd(j.db + c, v[2]);
    const long gi = (long)(f0 / gdiv) * j.C + c;
    if (j.dg) j.dg[gi] += wc * v[1] + bc * v[0];
    if (j.dgb) j.dgb[gi] += v[0];
    if (j.dmask_v && by == 0) atomicAdd(j.dmask_v + c, v[4]);
}

// grid (ceil(nvals / 64), ny row slices), 256 threads = 64 values x 4 row lanes.  ny = 1 (what the library launches): one writer per value,
// fixed order; the atomic only keeps the add safe beside another launch's single addend on the same slot
__device__ __forceinline__ void attn_reduce_block(const AttnReduceJob& j, int bx, int by, int ny, float (*red)[4][64]) {
    const int nvals = 4 * j.D + 32 * j.heads + j.heads;
    const int l = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int i = bx * 64 + l;
    const int per = (j.rows + ny - 1) / ny;
    const int r0 = by * per, r1 = min(j.rows, r0 + per);
    float acc = 0.f;
    if (i < nvals)
        for (int r = r0 + rg; r < r1; r += 4) acc += j.ws[(long)r * nvals + i];
    red[0][rg][l] = acc;
    __syncthreads();
    if (rg != 0 || i >= nvals) return;
    acc = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
    float* dst;
    if (i < 4 * j.D) { const int q = i / j.D, e = i % j.D; dst = (q == 0 ? j.dqw : q == 1 ? j.dqb : q == 2 ? j.dkw : j.dkb); if (dst) dst += e; }
    else if (i < 4 * j.D + 32 * j.heads) dst = j.demb ? j.demb + (i - 4 * j.D) : nullptr;
    else dst = j.dhscale ? j.dhscale + (i - 4 * j.D - 32 * j.heads) : nullptr;
    if (dst && acc != 0.f) atomicAdd(dst, acc);
}
