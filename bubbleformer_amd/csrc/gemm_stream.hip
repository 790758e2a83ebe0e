// Weight-stationary streaming GEMM for gfx950: C[M][N] = epi(A[M][K] @ W[N][K]^T) for the trunk's SHORT-K projections
// (K = E = 256..384: QKV, out-projection, fc1 forward and the fc2 data gradient; layers/attention.py:78,121,210,299, linear_layers.py:18).
//
// These products have tens of thousands of token rows and a few hundred weight rows / reduction steps, so a tile-per-workgroup kernel
// spends its life in prologues and epilogues (six K-steps between a cold start and an HBM-rate store burst).  Here instead:
//  * ONE persistent workgroup per CU owns a 128-column block of W for a run of row tiles; that block (128 x K bf16, <= 96 KB) is
//    loaded into LDS once and stays there -- the MI355X CU has 160 KB of LDS, which is what makes the block fit beside a ring;
//  * the token rows stream through a 2-slot ring of 256 x 64 chunks, global -> LDS by DMA (global_load_lds_dwordx4, no staging
//    registers), one chunk in flight while one is multiplied (a ~1 us step), across tile boundaries: the next tile's first chunk lands
//    while this tile finishes.  256-row tiles with 64 x 64 per wave halve the LDS read bytes and the barriers per FLOP of a 128-row tile;
//  * one raw s_barrier per K-step and a counted s_waitcnt vmcnt that never drains the DMA queue (stores and epilogue operand loads are
//    counted in issue order, cdna_hip_programming.md section 5 "Pipelining across barriers");
//  * the epilogue needs no LDS: two 16 x 16 accumulator tiles are exchanged between lane rows with v_permlane16_swap so that a lane
//    owns 8 consecutive columns -- 16-byte residual / gelu' loads and 16-byte stores straight from registers;
//  * only the token chunks are fetched per step: 32 KB per 4.2 MFLOP = 128 FLOP per DMA byte, the same as a 256 x 256 tile (a CU
//    takes in ~70 GB/s from L2, MI355X_MICROARCH.md "Indexed rows", which caps a 128 x 128 two-operand tile at 46 % of the MFMA rate).
// Workgroups form TEAMS of N/128 (one per column block) that sit on one XCD and sweep the same row tiles in step: a token chunk comes
// from HBM once and the rest of the team reads it from that XCD's L2 (with runs cut from a column-block-major sequence instead, every
// workgroup re-fetched its rows through the fabric: 36 us for the QKV projection, HBM-bound on re-reads).
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

namespace {
using namespace bfgemm;

constexpr int BM = 256, BNB = 128, BK = 64;
constexpr int CHUNK = BNB * BK;         // elements of one [128][64] weight chunk (16 KB)
constexpr int ACHUNK = BM * BK;         // elements of one [256][64] token chunk (32 KB)
constexpr int NSLOT = 2;                // token ring: one chunk multiplied, one in flight

struct StreamArgs {
    const bf16* A; long lda; const bf16* W; long ldw; bf16* C; long ldc;
    const float* bias; const float* colscale; const float* colshift; const float* rowscale; int rpg;
    int aux_mode; const bf16* aux; long ld_aux; bf16* gelu_out;
    int KS, mt, nb, ng;       // K chunks, 256-row tiles, column blocks per team, column groups (nb * ng blocks in all)
    int mh, hunit;            // 128-row half tiles; the weight-stationary kernel deals rows in units of `hunit` of them (1; 2 = whole tiles only)
    int stagger;
    int dbg;      // timing experiments (BF_STREAM_DEBUG): 1 no DMA waits, 2 every tile reads the rows of tile 0, 4 no LDS reads / MFMA, 8 no stores
};

__device__ __forceinline__ void wait_vm_n(int n) {      // n is wave-uniform; a smaller count than asked for is always safe (in-order retirement)
    if (n >= 16) { wait_vm<16>(); return; }
    switch (n) {
        case 0: wait_vm<0>(); break;   case 1: wait_vm<1>(); break;   case 2: wait_vm<2>(); break;   case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break;   case 5: wait_vm<5>(); break;   case 6: wait_vm<6>(); break;   case 7: wait_vm<7>(); break;
        case 8: wait_vm<8>(); break;   case 9: wait_vm<9>(); break;   case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
        case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; default: wait_vm<15>(); break;
    }
}
// 16-byte load the compiler does not count (it would drain the DMA queue at the first use): completion by the caller's wait_vm_n
__device__ __forceinline__ void gload16(uint4& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }

__device__ __forceinline__ void gload4(float& dst, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }

template <int AUX, bool GELU2, bool CS>        // AUX: BF_AUX_*; GELU2: second output gelu(value); CS: column scale / shift and per-row-group factor
__global__ void __launch_bounds__(512) stream_gemm_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Bres = reinterpret_cast<bf16*>(smem);                  // KS chunks [128 n][64 k], swizzled like every KC tile (lds_off)
    bf16* ring = Bres + (size_t)a.KS * CHUNK;                    // NSLOT chunks [256 m][64 k]
    const int KS = a.KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                     // 4 x 2 waves, 64 x 64 outputs each

    // ---- teams: the nb workgroups of a team hold the nb column blocks and sweep the SAME row tiles at the same time, all on one XCD
    // (label = blockIdx % 8 under round-robin placement; speed only) -- a chunk of A is fetched from HBM once and the rest of the team
    // finds it in that XCD's L2.  Row tiles are dealt evenly to the 8 * (workgroups per XCD / nb) teams.
    const int xl = blockIdx.x & 7, j = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int tpx = wpx / a.nb;                       // teams per XCD (host guarantees >= 1)
    if (j >= tpx * a.nb) return;
    const int team = xl * tpx + j / a.nb, nteams = 8 * tpx;
    // column groups (team_split): team t holds group t % ng and is the (t / ng)-th of that group's teams, which share the row tiles
    const int grp = team % a.ng, rank = team / a.ng, gteams = (nteams - grp + a.ng - 1) / a.ng;
    const int c_nb = grp * a.nb + j % a.nb;
    // Rows are dealt in HALF tiles (128 rows): a run of an odd number of them ends in a tile whose upper half belongs to the next team.
    // Waves 4-7 own exactly that half (DMA pieces 16..31 = rows 128..255, fragments of rows 128..255, the stores of those rows): in that
    // tile they issue nothing, multiply nothing and store nothing -- they only keep the barriers.  72 full tiles over 21 teams are 3 or
    // 4 each (the launch lasts 4); 144 halves are 6 or 7 (3.5).
    const int h_beg = a.hunit * (int)((long)(a.mh / a.hunit) * rank / gteams), h_end = a.hunit * (int)((long)(a.mh / a.hunit) * (rank + 1) / gteams);
    if (h_beg >= h_end) return;
    const int ntile = (h_end - h_beg + 1) >> 1;
    const bool odd_run = (h_end - h_beg) & 1;
    const int total_steps = ntile * KS;

    // ---- DMA geometry: piece p = rows 8p .. 8p+7 of a chunk; lane -> row 8p + (lane >> 3), LDS chunk (lane & 7) = global chunk
    // (lane & 7) ^ ((row >> 1) & 7) of that row (lds_off<bf16, false, 64>).  A chunk: 32 pieces, 4 per wave; W chunk: 16, 2 per wave.
    long roffA[4], roffB[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = 8 * (wave * 4 + t) + (lane >> 3);
        roffA[t] = (long)r * a.lda + 8 * ((lane & 7) ^ ((r >> 1) & 7));
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = 8 * (wave * 2 + t) + (lane >> 3);
        roffB[t] = (long)r * a.ldw + 8 * ((lane & 7) ^ ((r >> 1) & 7));
    }
    const unsigned lds0 = lds_addr(smem);
    const unsigned ldsB = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 2048u);                                   // this wave's two pieces inside a W chunk
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)KS * (unsigned)(CHUNK * 2) + (unsigned)wave * 4096u);   // ... four pieces inside an A chunk

    int issued = 0;                                  // vector-memory operations this wave has issued so far (DMA, loads, stores)
    // ---- DMA issue cursor (one chunk in front of the compute loop, across tile boundaries)
    int i_ks = 0, i_left = total_steps, i_slot = 0;
    const bf16* i_row = a.A + (long)((a.dbg & 2) ? 0 : h_beg) * (BM / 2) * a.lda;
    auto issue_A = [&]() {
        const unsigned dst = ldsA + (unsigned)i_slot * (unsigned)(ACHUNK * 2);
        if (!(odd_run && i_left <= KS && wave >= 4)) {      // (a chunk of the run's last, half tile: rows 128..255 are not this team's)
#pragma unroll
            for (int t = 0; t < 4; ++t) glds16(i_row + roffA[t] + i_ks * BK, dst + t * 1024u);
            issued += 4;
        }
        i_slot ^= 1;
        --i_left;
        if (++i_ks == KS) { i_ks = 0; if (!(a.dbg & 2)) i_row += (long)BM * a.lda; }
    };

    // ---- epilogue constants: after the lane-row exchange a lane owns 8 consecutive columns of each of its two 32-column halves
    const int li = lane & 15, lg = lane >> 4;
    const int c8 = wn * 64 + (lg & 1) * 16 + (lg >> 1) * 8;       // + 32 * pp
    float e_bias[2][8], e_cs[2][CS ? 8 : 1], e_ch[2][CS ? 8 : 1];
    {   // plain loads, forced complete here: inside the tile loop a pending compiler-visible load would make hipcc drain the DMA queue at its use
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int n = c_nb * BNB + c8 + 32 * pp + q;
                e_bias[pp][q] = a.bias ? a.bias[n] : 0.f;
                if constexpr (CS) { e_cs[pp][q] = a.colscale ? a.colscale[n] : 1.f; e_ch[pp][q] = a.colscale ? a.colshift[n] : 0.f; }
            }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                asm volatile("" : "+v"(e_bias[pp][q]));
                if constexpr (CS) { asm volatile("" : "+v"(e_cs[pp][q])); asm volatile("" : "+v"(e_ch[pp][q])); }
            }
    }

    // ---- prologue: the resident column block and the first two token chunks, interleaved so that step 0 needs only {W chunk 0, A
    // chunk 0} and step 1 only {W chunk 1, A chunk 1}: the rest of the block (80 KB) lands under the first two steps instead of in
    // front of them.  bmark[kb] / mark: `issued` right after W chunk kb / the A chunk of the step being waited for went out.
    int bmark[6];
    const bf16* w0 = a.W + (long)c_nb * BNB * a.ldw;
    auto load_W = [&](int kb) {
#pragma unroll
        for (int t = 0; t < 2; ++t) glds16(w0 + roffB[t] + kb * BK, ldsB + (unsigned)kb * (unsigned)(CHUNK * 2) + t * 1024u);
        issued += 2;
        return issued;
    };
    bmark[0] = load_W(0);
    issue_A();
    int mark = issued, mark_next = issued;
    bmark[1] = load_W(1);
    const bool pre1 = i_left > 0;                    // the second chunk goes out here too (its slot is free): step 0 then issues nothing
    if (pre1) { issue_A(); mark_next = issued; }
#pragma unroll
    for (int kb = 2; kb < 6; ++kb) bmark[kb] = kb < KS ? load_W(kb) : issued;
    bool first_tile = true;

    f32x4 acc[4][4];
    uint4 auxr[4][2];
    float rsr[4] = {1.f, 1.f, 1.f, 1.f};
    const bool has_rs = CS && a.rowscale != nullptr;
    int aux_mark = 0;
    int slot = 0;
    auto run_tile = [&](int tt) __attribute__((always_inline)) {
        const int m0 = h_beg * (BM / 2) + tt * BM;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks) {
            int need = mark;                         // this step's A chunk and, during the first tile, its W chunk (everything older included)
            if (first_tile) { const int bm = ks == 0 ? bmark[0] : ks == 1 ? bmark[1] : ks == 2 ? bmark[2] : ks == 3 ? bmark[3] : ks == 4 ? bmark[4] : bmark[5]; need = max(need, bm); }
            if (!(a.dbg & 1)) wait_vm_n(issued - need);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // ... for every wave; and nobody reads the previous step's slot any more
            if (ks == 0) {                           // this tile's residual / gelu' operand and row factors: issued BEFORE the younger DMAs, in
                if constexpr (AUX != BF_AUX_NONE) {  // flight for the whole tile (a load issued in the epilogue would wait for every older DMA)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
                            gload16(auxr[i][pp], a.aux + (long)(m0 + wm * 64 + i * 16 + li) * a.ld_aux + c_nb * BNB + c8 + 32 * pp);
                    issued += 8;
                }
                if (has_rs) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gload4(rsr[i], a.rowscale + (m0 + wm * 64 + i * 16 + li) / a.rpg);
                    issued += 4;
                }
                aux_mark = issued;
            }
            // the next chunk's DMA goes into the slot the previous step read.  Waves 0-3 issue it here, waves 4-7 after their first
            // K-half (BF_STREAM_STAGGER, default on): the two waves of a SIMD then do not both spend the same ~400 cycles issuing DMA
            // while the matrix pipe idles (MI355X_MICROARCH.md "Two waves per SIMD", item 9: split roles by wave >= 4)
            const bool late = a.stagger && wave >= 4;
            const bool skip = first_tile && ks == 0 && pre1;      // the prologue already sent this step's look-ahead chunk
            if (!late && !skip) { if (i_left > 0 && !(a.dbg & 32)) issue_A(); mark_next = issued; }
            const bf16* cA = ring + (size_t)slot * ACHUNK;
            const bf16* cB = Bres + (size_t)ks * CHUNK;
            if (!(a.dbg & 4))
#pragma unroll
            for (int kk = 0; kk < BK; kk += 32) {
                if (kk == 32 && late && !skip) { if (i_left > 0 && !(a.dbg & 32)) issue_A(); mark_next = issued; }
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = frag_bf16<false, BK>(cA, wm * 64 + i * 16, kk, lane);
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) fb[jn] = frag_bf16<false, BK>(cB, wn * 64 + jn * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jn = 0; jn < 4; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[jn], fa[i], acc[i][jn], 0, 0, 0);
            }
            slot ^= 1;
            mark = mark_next;
        }
        first_tile = false;
        // ---- epilogue of the tile, from registers.  acc[i][jn]: row 16i + li, columns 16jn + 4lg .. +3.  After exchanging the odd lane
        // rows of tile 2pp with the even lane rows of tile 2pp + 1 a lane holds 8 consecutive columns at c8 + 32pp.
        if (AUX != BF_AUX_NONE || has_rs) {
            wait_vm_n(issued - aux_mark);
            if constexpr (AUX != BF_AUX_NONE) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) asm volatile("" : "+v"(auxr[i][pp].x), "+v"(auxr[i][pp].y), "+v"(auxr[i][pp].z), "+v"(auxr[i][pp].w));
            }
            if constexpr (CS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(rsr[i]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + li;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][2 * pp][r]), __float_as_uint(acc[i][2 * pp + 1][r]), false, false);
                    v[r] = __uint_as_float(sw[0]); v[4 + r] = __uint_as_float(sw[1]);
                }
                if constexpr (AUX == BF_AUX_ADD) {      // (gemm_common.h: epi_lin* -- the frame-pair forward kernel applies the same expressions)
                    const bf16x8 ax = __builtin_bit_cast(bf16x8, auxr[i][pp]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float cs_ = 1.f, ch_ = 0.f;
                        if constexpr (CS) { cs_ = e_cs[pp][q]; ch_ = e_ch[pp][q]; }
                        v[q] = epi_lin_add(v[q], e_bias[pp][q], cs_, ch_, has_rs ? rsr[i] : 1.f, CS, (float)ax[q]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float cs_ = 1.f, ch_ = 0.f;
                        if constexpr (CS) { cs_ = e_cs[pp][q]; ch_ = e_ch[pp][q]; }
                        v[q] = epi_lin(v[q], e_bias[pp][q], cs_, ch_, has_rs ? rsr[i] : 1.f, CS);
                    }
                    if constexpr (AUX == BF_AUX_DGELU) {
                        const bf16x8 ax = __builtin_bit_cast(bf16x8, auxr[i][pp]);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = v[q] * dgelu_fast((float)ax[q]);
                    }
                }
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)v[q];
                const long off = (long)m * a.ldc + c_nb * BNB + c8 + 32 * pp;
                if (a.dbg & 16) asm volatile("" :: "v"(o)); else
                if (!(a.dbg & 8)) *reinterpret_cast<bf16x8*>(a.C + off) = o;
                if constexpr (GELU2) {
                    bf16x8 g8;
#pragma unroll
                    for (int q = 0; q < 8; ++q) g8[q] = (bf16)gelu_fast(v[q]);
                    *reinterpret_cast<bf16x8*>(a.gelu_out + off) = g8;
                }
            }
        }
        issued += GELU2 ? 16 : 8;
    };
    // the full tiles by all eight waves; the closing half tile of an odd run by waves 0-3 alone -- waves 4-7 (whose rows, DMA pieces and
    // stores it does not contain) only keep its KS barriers, after their own pieces of the resident weight have landed
    const int nfull = odd_run ? ntile - 1 : ntile;
    for (int tt = 0; tt < nfull; ++tt) run_tile(tt);
    if (odd_run) {
        if (wave < 4) run_tile(ntile - 1);
        else
            for (int ks = 0; ks < KS; ++ks) {
                wait_vm<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
    }
}


// ================================================================================================ ping-pong form (both operands streamed)
// The same product, tiles, teams and epilogue, with a different main loop (cdna_hip_programming.md section 5, the 8-phase template's
// wave stagger; MI355X_MICROARCH.md "Two waves per SIMD"):
//  * nothing is resident: each 64-deep K-step brings a [256][64] token chunk AND the [128][64] weight chunk (from L2: the weight is a
//    few hundred KB that every workgroup re-reads) into one of THREE 48 KB slots; two K-steps are always in flight ahead of the one being
//    multiplied -- across tile boundaries -- so a DMA has more than a whole K-step to land and any K (a multiple of 64) is covered;
//  * a K-step is two halves (k 0..31, 32..63); per half a wave runs a LOAD segment (8 ds_read_b128 for 4 + 4 fragments, 3 of its 6 DMA
//    pieces of step + 2, the counted vmcnt wait) and a COMPUTE segment (16 MFMA = 256 matrix-pipe cycles), each closed by a raw
//    s_barrier.  Waves 4-7 run one barrier behind waves 0-3: on every SIMD one wave computes while its partner loads, so the matrix pipe
//    alternates between the two instead of idling through LDS reads, DMA issue and barrier skew (the lock-step loop above spends a step as
//    the SUM of those parts);
//  * at the end of a tile the two halves re-align (one barrier), run the register epilogue at the same time (its stores drain under the
//    next tile's first steps: the DMAs those steps wait for are OLDER than the stores in the in-order vmcnt queue) and stagger again.
constexpr int PSLOT = (ACHUNK + CHUNK) * 2;   // bytes of one ring slot: token chunk + weight chunk
constexpr int PNSLOT = 3;

__device__ __forceinline__ void wait_vm_wide(int n) {      // as wait_vm_n, up to 40 outstanding operations (epilogue stores + look-ahead DMAs)
    if (n < 16) { wait_vm_n(n); return; }
    if (n >= 40) { wait_vm<40>(); return; }
    switch ((n - 16) >> 2) {          // steps of 4: a smaller count than asked for is always safe
        case 0: wait_vm<16>(); break; case 1: wait_vm<20>(); break; case 2: wait_vm<24>(); break;
        case 3: wait_vm<28>(); break; case 4: wait_vm<32>(); break; default: wait_vm<36>(); break;
    }
}

template <int AUX, bool GELU2, bool CS>
__global__ void __launch_bounds__(512) stream_pp_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KS = a.KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                     // 4 x 2 waves, 64 x 64 outputs each
    const bool late = wave >= 4;                                 // the half that runs one barrier behind

    // ---- teams (as above): nb workgroups on one XCD hold the nb column blocks and sweep the same row tiles
    const int xl = blockIdx.x & 7, j = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int tpx = wpx / a.nb;
    if (j >= tpx * a.nb) return;
    const int team = xl * tpx + j / a.nb, nteams = 8 * tpx;
    const int grp = team % a.ng, rank = team / a.ng, gteams = (nteams - grp + a.ng - 1) / a.ng;
    const int c_nb = grp * a.nb + j % a.nb;
    const int t_beg = (int)((long)a.mt * rank / gteams), t_end = (int)((long)a.mt * (rank + 1) / gteams);
    if (t_beg >= t_end) return;
    const int total_steps = (t_end - t_beg) * KS;

    // ---- DMA geometry (as above): piece = 8 rows x 128 bytes; token chunk 32 pieces (4 per wave), weight chunk 16 (2 per wave)
    long roffA[4], roffB[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = 8 * (wave * 4 + t) + (lane >> 3);
        roffA[t] = (long)r * a.lda + 8 * ((lane & 7) ^ ((r >> 1) & 7));
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = 8 * (wave * 2 + t) + (lane >> 3);
        roffB[t] = (long)r * a.ldw + 8 * ((lane & 7) ^ ((r >> 1) & 7));
    }
    const unsigned lds0 = lds_addr(smem);
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 4096u);                              // this wave's four pieces of a token chunk
    const unsigned ldsB = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(ACHUNK * 2) + (unsigned)wave * 2048u);     // ... two pieces of a weight chunk
    const bf16* w0 = a.W + (long)c_nb * BNB * a.ldw;

    int issued = 0;                                  // vector-memory operations this wave has issued so far (DMA, loads, stores)
    // ---- DMA issue cursor: two K-steps in front of the compute loop, across tile boundaries
    int i_ks = 0, i_left = total_steps, i_slot = 0;
    const bf16* i_row = a.A + (long)t_beg * BM * a.lda;
    auto issue_lo = [&]() {                          // first half of a step's pieces: token pieces 0 .. 2
        const unsigned dst = ldsA + (unsigned)i_slot * (unsigned)PSLOT;
#pragma unroll
        for (int t = 0; t < 3; ++t) glds16(i_row + roffA[t] + i_ks * BK, dst + t * 1024u);
        issued += 3;
    };
    auto issue_hi = [&]() {                          // second half: token piece 3 and the two weight pieces; the cursor moves on
        glds16(i_row + roffA[3] + i_ks * BK, ldsA + (unsigned)i_slot * (unsigned)PSLOT + 3 * 1024u);
        const unsigned dst = ldsB + (unsigned)i_slot * (unsigned)PSLOT;
#pragma unroll
        for (int t = 0; t < 2; ++t) glds16(w0 + roffB[t] + i_ks * BK, dst + t * 1024u);
        issued += 3;
        i_slot = i_slot == PNSLOT - 1 ? 0 : i_slot + 1;
        --i_left;
        if (++i_ks == KS) { i_ks = 0; i_row += (long)BM * a.lda; }
    };

    // ---- epilogue constants (as above)
    const int li = lane & 15, lg = lane >> 4;
    const int c8 = wn * 64 + (lg & 1) * 16 + (lg >> 1) * 8;       // + 32 * pp
    float e_bias[2][8], e_cs[2][CS ? 8 : 1], e_ch[2][CS ? 8 : 1];
    {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int n = c_nb * BNB + c8 + 32 * pp + q;
                e_bias[pp][q] = a.bias ? a.bias[n] : 0.f;
                if constexpr (CS) { e_cs[pp][q] = a.colscale ? a.colscale[n] : 1.f; e_ch[pp][q] = a.colscale ? a.colshift[n] : 0.f; }
            }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                asm volatile("" : "+v"(e_bias[pp][q]));
                if constexpr (CS) { asm volatile("" : "+v"(e_cs[pp][q])); asm volatile("" : "+v"(e_ch[pp][q])); }
            }
    }

    // ---- prologue: steps 0 and 1 in flight, step 0 landed for every wave
    issue_lo(); issue_hi();
    const int mark0 = issued;
    if (i_left > 0) { issue_lo(); issue_hi(); }
    int mark_next = issued;                          // `issued` right after the last piece of the step FOLLOWING the one being multiplied
    wait_vm_wide(issued - mark0);
    __builtin_amdgcn_s_barrier();

    f32x4 acc[4][4];
    uint4 auxr[4][2];
    float rsr[4] = {1.f, 1.f, 1.f, 1.f};
    const bool has_rs = CS && a.rowscale != nullptr;
    int aux_mark = 0;
    int slot = 0, done = 0;                          // done: K-steps multiplied so far
    for (int t = t_beg; t < t_end; ++t) {
        const int m0 = t * BM;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (late) __builtin_amdgcn_s_barrier();      // stagger: waves 4-7 one barrier behind
        for (int ks = 0; ks < KS; ++ks, ++done) {
            const bf16* cA = reinterpret_cast<const bf16*>(smem + (size_t)slot * PSLOT);
            const bf16* cB = cA + ACHUNK;
            bf16x8 fa[4], fb[4];
            int mark_nn = mark_next;
            // ======== load segment, k 0..31
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = frag_bf16<false, BK>(cA, wm * 64 + i * 16, 0, lane);
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) fb[jn] = frag_bf16<false, BK>(cB, wn * 64 + jn * 16, 0, lane);
            if (ks == 0) {                           // this tile's residual / gelu' operand and row factors: in flight for the whole tile
                if constexpr (AUX != BF_AUX_NONE) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
                            gload16(auxr[i][pp], a.aux + (long)(m0 + wm * 64 + i * 16 + li) * a.ld_aux + c_nb * BNB + c8 + 32 * pp);
                    issued += 8;
                }
                if (has_rs) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gload4(rsr[i], a.rowscale + (m0 + wm * 64 + i * 16 + li) / a.rpg);
                    issued += 4;
                }
                aux_mark = issued;
            }
            // step done + 2 goes into the slot that held step done - 1: every wave finished reading it before the barrier that ended
            // ITS load segment of that step's second half
            const bool more = i_left > 0 && !(a.dbg & 32);
            if (more) issue_lo();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ======== compute segment
            if (PP_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[jn], fa[i], acc[i][jn], 0, 0, 0);
            if (PP_SETPRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ======== load segment, k 32..63
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = frag_bf16<false, BK>(cA, wm * 64 + i * 16, 32, lane);
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) fb[jn] = frag_bf16<false, BK>(cB, wn * 64 + jn * 16, 32, lane);
            if (more) { issue_hi(); mark_nn = issued; }
            if (done + 1 < total_steps && !(a.dbg & 1)) wait_vm_wide(issued - mark_next);      // this wave's pieces of the next step have landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ======== compute segment
            if (PP_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[jn], fa[i], acc[i][jn], 0, 0, 0);
            if (PP_SETPRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            slot = slot == PNSLOT - 1 ? 0 : slot + 1;
            mark_next = mark_nn;
        }
        if (!late) __builtin_amdgcn_s_barrier();     // re-align: pairs with the extra barrier of waves 4-7
        // ---- epilogue of the tile, from registers, both halves at the same time
        if (AUX != BF_AUX_NONE || has_rs) {
            wait_vm_wide(issued - aux_mark);
            if constexpr (AUX != BF_AUX_NONE) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) asm volatile("" : "+v"(auxr[i][pp].x), "+v"(auxr[i][pp].y), "+v"(auxr[i][pp].z), "+v"(auxr[i][pp].w));
            }
            if constexpr (CS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(rsr[i]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + li;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][2 * pp][r]), __float_as_uint(acc[i][2 * pp + 1][r]), false, false);
                    v[r] = __uint_as_float(sw[0]); v[4 + r] = __uint_as_float(sw[1]);
                }
                if constexpr (AUX == BF_AUX_ADD) {      // (gemm_common.h: epi_lin* -- the frame-pair forward kernel applies the same expressions)
                    const bf16x8 ax = __builtin_bit_cast(bf16x8, auxr[i][pp]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float cs_ = 1.f, ch_ = 0.f;
                        if constexpr (CS) { cs_ = e_cs[pp][q]; ch_ = e_ch[pp][q]; }
                        v[q] = epi_lin_add(v[q], e_bias[pp][q], cs_, ch_, has_rs ? rsr[i] : 1.f, CS, (float)ax[q]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float cs_ = 1.f, ch_ = 0.f;
                        if constexpr (CS) { cs_ = e_cs[pp][q]; ch_ = e_ch[pp][q]; }
                        v[q] = epi_lin(v[q], e_bias[pp][q], cs_, ch_, has_rs ? rsr[i] : 1.f, CS);
                    }
                    if constexpr (AUX == BF_AUX_DGELU) {
                        const bf16x8 ax = __builtin_bit_cast(bf16x8, auxr[i][pp]);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = v[q] * dgelu_fast((float)ax[q]);
                    }
                }
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)v[q];
                const long off = (long)m * a.ldc + c_nb * BNB + c8 + 32 * pp;
                *reinterpret_cast<bf16x8*>(a.C + off) = o;
                if constexpr (GELU2) {
                    bf16x8 g8;
#pragma unroll
                    for (int q = 0; q < 8; ++q) g8[q] = (bf16)gelu_fast(v[q]);
                    *reinterpret_cast<bf16x8*>(a.gelu_out + off) = g8;
                }
            }
        }
        issued += GELU2 ? 16 : 8;
    }
}


int num_cus() {
    static const int n = []() {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount < 8) return 256;
        return p.multiProcessorCount;
    }();
    return n;
}

// A team is one workgroup per column block of ONE column group, on one XCD (wpx workgroups each).  All NB blocks in one team is the
// least re-reading of the token rows, but 12 blocks (N = 1536: fc1, the fc2 data gradient) leave room for 2 teams of 12 on an XCD's 32
// workgroups: 16 teams for 72 row tiles = 5 tiles for some, with a quarter of the CUs unused.  Two groups of 6 blocks: 5 teams per XCD,
// 20 per group, 4 tiles at most.  With rows dealt in half tiles (stream_gemm_kernel) three groups of 4 blocks do better still: 21-22 teams
// per group, 7 half tiles at most.  Pick the split with the shortest longest run; ties go to fewer groups (fewer reads of the rows).
bool team_split(int NB, int mt, int wpx, int* nb, int* ng) {
    static const int force = bf_knob("BF_STREAM_GROUPS", 0);
    int best = 0, best_cost = 0;
    for (int g = 1; g <= NB; ++g) {
        if (NB % g || NB / g > wpx || (force > 0 && g != force)) continue;
        const int nteams = 8 * (wpx / (NB / g));
        if (nteams < g) continue;                              // every group needs a team
        const int fewest = nteams / g;                         // teams of the last group
        const int cost = (mt + fewest - 1) / fewest;
        if (!best || cost < best_cost) { best = g; best_cost = cost; }
    }
    if (!best) return false;
    *ng = best; *nb = NB / best;
    return true;
}

}  // namespace

// 0 = handled, 1 = shape / feature not covered (the caller's tile kernel runs), < 0 = error
int bf_gemm_stream_try(int M, int N, int K, const bf_operand* A, const bf_operand* B, const bf_epilogue* E, hipStream_t st) {
    static const int enabled = bf_knob("BF_GEMM_STREAM", 1);
    if (!enabled) return 1;
    if (A->layout != BF_LAY_KC || B->layout != BF_LAY_KC || A->pro != BF_PRO_NONE || B->pro != BF_PRO_NONE) return 1;
    if (A->gw > 0 || A->seglen > 0 || B->gw > 0 || B->seglen > 0 || E->gw > 0 || E->seglen > 0) return 1;
    if (E->out_mode != BF_OUT_STORE || E->colsum) return 1;
    // Two main loops share the tiles, teams and epilogue.  K = 256..384 (the weight block fits the LDS beside a 2-slot token ring): the
    // weight-stationary loop -- measured 31.0 vs 34.5 us (QKV), 56.7 vs 60.9 us (fc1) against the ping-pong form, whose per-step gain is
    // eaten by streaming the weight chunk too (6 instead of 4 DMA pieces per wave and step, ~65 cycles of wave time each).  Longer K
    // (fc2 forward, K = 1536): only the ping-pong form applies, 36 us against 40 us for the 128 x 128 tile kernel.  BF_STREAM_PP=1 / 0
    // forces one form where both apply.
    static const int pp_env = bf_knob("BF_STREAM_PP", -1);
    if (M % BM || N % BNB || K % BK || K < 2 * BK) return 1;
    const bool ws_ok = K >= 4 * BK && K <= 6 * BK;
    const bool use_pp = pp_env == 1 || !ws_ok;
    if (pp_env == 0 && !ws_ok) return 1;
    if (A->ld % 8 || B->ld % 8 || E->ldc % 8 || (E->aux_mode != BF_AUX_NONE && E->ld_aux % 8)) return 1;
    if (((uintptr_t)A->p | (uintptr_t)B->p | (uintptr_t)E->c | (uintptr_t)E->aux | (uintptr_t)E->gelu_out) & 15) return 1;
    if (E->gelu_out && E->aux_mode != BF_AUX_NONE) return 1;
    if (E->colscale && !E->colshift) return 1;
    StreamArgs a;
    a.A = (const bf16*)A->p; a.lda = A->ld; a.W = (const bf16*)B->p; a.ldw = B->ld; a.C = (bf16*)E->c; a.ldc = E->ldc;
    a.bias = E->bias; a.colscale = E->colscale; a.colshift = E->colshift; a.rowscale = E->rowscale; a.rpg = E->rows_per_group > 0 ? E->rows_per_group : 1;
    a.aux_mode = E->aux_mode; a.aux = (const bf16*)E->aux; a.ld_aux = E->ld_aux; a.gelu_out = (bf16*)E->gelu_out;
    a.KS = K / BK; a.mt = M / BM; a.mh = M / (BM / 2);
    static const int half_on = bf_knob("BF_STREAM_HALF", 1);
    a.hunit = half_on ? 1 : 2;
    const int grid = (num_cus() / 8) * 8;
    if (!team_split(N / BNB, (use_pp || !half_on) ? a.mt : a.mh, grid / 8, &a.nb, &a.ng)) return 1;      // rows are dealt in half tiles by the weight-stationary kernel
    static const int dbg = bf_knob("BF_STREAM_DEBUG", 0);
    a.dbg = dbg;
    static const int stagger = bf_knob("BF_STREAM_STAGGER", 1);
    a.stagger = stagger;
    const int lds_bytes = a.KS * CHUNK * 2 + NSLOT * ACHUNK * 2;
    static thread_local char pname[64];
    snprintf(pname, sizeof(pname), use_pp ? "stream_pp<%s>" : "stream_gemm<%s>", E->gelu_out ? "gelu2" : E->aux_mode == BF_AUX_ADD ? "add" : E->aux_mode == BF_AUX_DGELU ? "dgelu" : "plain");
    BfProfScope prof(st, pname, 2.0 * M * N * K, ((double)M * K + (double)N * K + (double)M * N * (E->gelu_out ? 2 : 1) + (E->aux_mode != BF_AUX_NONE ? (double)M * N : 0.0)) * 2.0);
#define BF_STREAM_GO(AUXM, G2, CSF)                                                                                                       \
    do {                                                                                                                                \
        static BfPerDeviceOnce attr_once, attr_pp_once; bool& attr_done = attr_once.flag(); bool& attr_pp_done = attr_pp_once.flag();                                                                            \
        if (use_pp) {                                                                                                                   \
            if (!attr_pp_done) {                                                                                                        \
                hipError_t e_ = hipFuncSetAttribute((const void*)stream_pp_kernel<AUXM, G2, CSF>, hipFuncAttributeMaxDynamicSharedMemorySize, PNSLOT * PSLOT); \
                if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);                                                           \
                attr_pp_done = true;                                                                                                    \
            }                                                                                                                           \
            hipLaunchKernelGGL((stream_pp_kernel<AUXM, G2, CSF>), dim3(grid), dim3(512), PNSLOT * PSLOT, st, a);                        \
            break;                                                                                                                      \
        }                                                                                                                               \
        if (!attr_done) {                                                                                                               \
            hipError_t e_ = hipFuncSetAttribute((const void*)stream_gemm_kernel<AUXM, G2, CSF>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * CHUNK * 2 + NSLOT * ACHUNK * 2); \
            if (e_ != hipSuccess) return bf_fail(e_, __FILE__, __LINE__);                                                               \
            attr_done = true;                                                                                                           \
        }                                                                                                                               \
        hipLaunchKernelGGL((stream_gemm_kernel<AUXM, G2, CSF>), dim3(grid), dim3(512), lds_bytes, st, a);                                    \
    } while (0)
    const bool cs = E->colscale != nullptr || E->rowscale != nullptr;
    if (E->gelu_out && !cs) BF_STREAM_GO(BF_AUX_NONE, true, false);
    else if (E->gelu_out) return 1;
    else if (E->aux_mode == BF_AUX_ADD && cs) BF_STREAM_GO(BF_AUX_ADD, false, true);
    else if (E->aux_mode == BF_AUX_ADD) BF_STREAM_GO(BF_AUX_ADD, false, false);
    else if (E->aux_mode == BF_AUX_DGELU && !cs) BF_STREAM_GO(BF_AUX_DGELU, false, false);
    else if (E->aux_mode == BF_AUX_DGELU) return 1;
    else if (cs) return 1;
    else BF_STREAM_GO(BF_AUX_NONE, false, false);
#undef BF_STREAM_GO
    BF_CHECK_LAUNCH();
    return 0;
}
