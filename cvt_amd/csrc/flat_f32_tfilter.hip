// flat_f32_tfilter.hip -- exhaustive fp32 search (BruteforceSearch<float>::searchKnn, brutoforce.hpp:73-93, with InnerProductSpace,
// space_ip.hpp:211-239, or L2Space, space_l2.h:153-184) for BATCHES (16 queries and more, 262 144 rows and more) of any width that is a
// multiple of 4 up to 2048-d, as a threshold filter (round 6).
//
// The answer -- the k smallest (distance, row) per query, distances in the reference's own summation order -- comes out bit for
// bit because every reported distance is evaluated by the exact code (fs_exact, dist_f32.h order).  The matrix cores only decide
// WHICH rows need one.  The stream kernels of flat_f32_stream.hip do that with the queries in registers and a (best, second) pair
// per lane, query and group: 0.32-0.36 ms per pass of 384 queries over 1 M x 128-d rows whatever the pass carries, 0.27 of the bf16
// matrix peak at 1000 queries.  Here the operands swap roles and nothing is shared between waves:
//   * a workgroup (8 waves, two per SIMD) keeps up to 512 QUERIES in LDS for its whole life, as bf16 matrix operands
//     ([block of 32][K step][term][64 lanes] x 16 B: one conflict-free ds_read_b128 per operand);
//   * a wave takes RT 32-row tiles of the bf16 operand copy of the rows (flat_pack_kernel: [tile][K step][hi | lo][64 lanes] x 16 B,
//     1 KB per load instruction) straight into REGISTERS and walks the query blocks: T = x.q + b_x (b_x = -|x|^2/2 for L2, 0 for
//     the inner product: the start value of the accumulators, SrcC of a block's first matrix instruction).  With A = rows and
//     B = queries a lane ends up with the scores of ITS QUERY against 16 rows per tile, so the query's threshold is a per-lane
//     constant: a v_max3 tree and one compare per tile and block, no ring, no barrier, no (best, second);
//   * NPROD says how much of the two-term bf16 split is multiplied: 3 = x1.q1 + x2.q1 + x1.q2 (error 2^-13 Q as in the stream
//     kernels), 2 = (x1 + x2).q1, 1 = x1.q1 alone.  Fewer products = a wider margin = more candidates (1 M SIFT-like rows, top-100,
//     sample of an eighth: 850 / 1330 / 2040 per query) for a third / two thirds less matrix work; the rows that get exact
//     distances in the end stay k and a few (101 / 119 / 136);
//   * a lane whose 16 scores reach its threshold writes them as ONE record (16 scores, query, first row: 80 bytes) into its
//     wave's own region of a record area -- no atomic, no wait; `wcnt` counts a wave's records.
// Pipeline: (1) MAX mode over a sample of the rows (a fifth of the tile groups, spread evenly): 1024 maxima of disjoint row sets per query (global atomic max, no
// return); the k-th largest of them is reached by k distinct rows, so it bounds the k-th best score from below, and theta - margin
// admits every row that can be among the k best (ft_theta_kernel).  (2) FILTER mode over all rows.  (3) ft_bucket_kernel: the
// records' scores at or above the threshold go to per-query candidate lists.  (4) ft_finish_kernel, one workgroup per query: the
// k-th largest candidate score by a radix select, the candidates at or above it minus the margin get exact distances, ranked by
// (distance, row).  A query whose list runs over, whose sample has fewer than k slots filled, whose bound does not hold (non-finite
// values, magnitudes outside 2^-60 .. 2^60) or whose wave region ran full is flagged in redo[]: the exact kernels answer it.
// Workgroups that hold DIFFERENT query chunks walk the SAME tiles on the same XCD: the rows cross the fabric once per XCD.
//
// Bound (u = 2^-24, Q = |q|^2 + max |x|^2 >= 2 |x||q|, |T| <= Q): bf16 keeps 8 significant bits and rounds to nearest even,
// |v - v1| <= e |v| with e = 2^-8 (half a unit in the last place, relative to the smallest v of a binade), |v - v1 - v2| <= e^2 |v|.
// Accumulation of m terms whose partial sums stay below Q / 2 in magnitude: 2u per term, m u Q.  Reference sum ~200 uQ in T units.
//   NPROD 3: omitted x2.q2 <= e^2 |x||q| = 128 uQ, the two split residuals 256 uQ, 3 D + 1 terms 385 uQ, b_x 8 uQ: 777 uQ;
//            margin 2^-13 Q = 2048 uQ >= 2 (777 + 200).
//   NPROD 2: x.q - (x1 + x2).q1 = (x - x1 - x2).q + (x1 + x2).(q - q1) <= 128 uQ + (1 + e^2) max |x| |q - q1| =: 128 uQ + W_q, 2 D + 1 terms
//            257 uQ, b_x 8 uQ.  W_q is evaluated per query (ft_qlow: the residues of the query's own rounding, Cauchy-Schwarz against
//            the longest row) instead of its worst case e |x||q| = 32 768 uQ -- a query's residues add up to a third of that:
//            margin 2^-13 Q + 2 W_q >= 2 (393 uQ + W_q + 200 uQ).
//   NPROD 1: x.q - x1.q1 = (x - x1).q + x1.(q - q1) <= e (2 + e) |x||q| <= 65 664 uQ, D + 1 terms 129 uQ, b_x 8 uQ: 65 801 uQ;
//            margin 2^-7 Q + 2^-11 Q = 139 264 uQ >= 2 (65 801 + 200).
#include <algorithm>
#include <cmath>
#include <atomic>
#include <vector>

#include "common.h"
#include "flat_f32_common.h"
#include "kernels.h"

namespace cvtmi {

namespace {

constexpr int FT_WAVES = 8;          // per workgroup: two per SIMD
constexpr int FT_GRID = 256;         // workgroups: one per CU
constexpr int FT_SLOTS = 1024;       // sample maxima per query (k <= 128)
constexpr int FT_SLOTS_BIG = 4096;   // k = 129 .. 2048
constexpr int FT_CAP_BIG = 32768;    // candidates per query of ft_finish_big_kernel (its keys fill 128 KB of LDS)
constexpr int FT_KEEP_BIG = 4096;    // rows per query it gives exact distances (k and the rows inside the margin band)
constexpr int FT_PASS_BIG = 256;     // queries per pass for k > 128 (the record area: ~k ln-ish times more hits per query)
constexpr int FT_CAP = 8192;         // candidates per query the finish takes (one product on 1 M SIFT-like rows: 2 700 on average, 4 900 at most)
constexpr int FT_NBMAX = 16;         // query blocks a workgroup holds
constexpr int FT_PASS = 1024;        // queries per pass of the pipeline (sizes the record area)
constexpr int FT_KEEP = 1024;        // rows per query the finish gives exact distances
constexpr int FT_DMAX = 2048;        // widest row
constexpr int FT_SLACK = 4096;       // bytes of LDS behind the queries' operands that the one-wave-per-SIMD form's operand requests may read

struct FtArgs {
    const uint4 *pack;       // bf16 operand copy of the rows
    const float *bias;       // b_x per row (padding rows: FS_PAD_BIAS)
    int64_t t1, n_tiles;     // tiles [0, t1) of n_tiles
    int64_t n_sample;        // MAX mode: the sample is n_sample groups of RT tiles spread evenly over all groups (rows arrive video by video: a leading
                             // block would be the first videos only) -- a whole number of groups per wave, or the pass waits for the waves with one more
    const float *Q;
    int D;                   // floats per query (<= 16 NCH: the operands are zero beyond it)
    int nq, chunks, qper;    // queries of chunk c: [c qper, min(nq, (c + 1) qper)), qper a multiple of 32
    uint32_t *smax;          // MAX mode: [nq][nslots] ordered keys (zeroed by the caller)
    int nslots;              // sample maxima per query (a power of two)
    const float *thr;        // FILTER mode: [nq] (NaN: nothing passes)
    uint4 *rec;              // FILTER mode: [FT_GRID waves][cap] records of 5 x 16 bytes
    uint32_t *wcnt;          // FILTER mode: [FT_GRID waves] records a wave had (beyond cap: not stored)
    uint32_t cap;
    const uint32_t *qlist;   // second attempt (FILTER mode, chunks == 1): the pass's queries are qlist[0 .. min(*qcount, qper)) -- both on the device
    const uint32_t *qcount;
    int dbg;                 // "flat_f32_dbg": 8 = the filter passes alone (timing; results stale), 32 = thresholds loosened by 0.02 Q (tests: lists run over)
};

__device__ __forceinline__ float ft_max3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float ft_max16(const f32x16 &v)
{
    // two chains of four (a chain of eight dependent instructions left the vector pipe idle between them), no canonicalising v_max at the end
    float m0 = ft_max3(v[0], v[1], v[2]), m1 = ft_max3(v[3], v[4], v[5]);
    m0 = ft_max3(m0, v[6], v[7]); m1 = ft_max3(m1, v[8], v[9]);
    m0 = ft_max3(m0, v[10], v[11]); m1 = ft_max3(m1, v[12], v[13]);
    m0 = ft_max3(m0, v[14], v[15]);
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(m0), "v"(m1));
    return r;
}

// KH = 2 (1536 / 2048-d): a row tile's K steps do not fit 512 registers -- a wave holds HALF of them at a time (NCH K steps), the
// accumulators of its ONE query block (32 queries: 128 KB of operands at 2048-d) run through both halves.
template <int NCH, int NPROD, int RT, bool MAXMODE, int NW = FT_WAVES, int KH = 1>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void flat_f32_tfilter_kernel(const FtArgs a)
{
    constexpr int NCHT = NCH * KH;            // K steps of a row
    static_assert(KH == 1 || (KH == 2 && NW == 4 && NPROD == 1 && RT == 1), "two K halves: one wave per SIMD, one tile, one product");
    constexpr int NT = NPROD == 3 ? 2 : 1;    // terms of a query in LDS
    constexpr int NA = NPROD >= 2 ? 2 : 1;    // terms of a row in registers
    extern __shared__ __attribute__((aligned(16))) uint8_t ft_q[];   // [block][K step][term] x 1 KB, then the blocks' thresholds
    const int tid = threadIdx.x, lane = tid & 63, lj = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (chunk, slice): the workgroups of one XCD (blockIdx & 7) that hold different chunks share their slices
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int chunk = idx % a.chunks, slices = 8 * ((int)(gridDim.x >> 3) / a.chunks), slice = xcd + 8 * (idx / a.chunks);
    const int q0 = chunk * a.qper;
    int nq_all = a.nq;
    if (a.qlist) {   // (wave-uniform: a scalar load)
        const int listed = (int)*a.qcount;
        nq_all = listed < a.qper ? listed : a.qper;
    }
    const int nqc = nq_all - q0 < a.qper ? nq_all - q0 : a.qper;
    const int wave_g = blockIdx.x * NW + wave;
    if (nqc <= 0) {
        if (!MAXMODE && lane == 0) a.wcnt[wave_g] = 0u;
        return;
    }
    const int nb = (nqc + 31) >> 5;
    float *thr_s = reinterpret_cast<float *>(ft_q + (size_t)nb * NCHT * NT * 1024 + FT_SLACK);
    for (int i = tid; i < nb * 32 * NCHT * 2; i += 64 * NW) {   // (query, K step, half) -> its 16-byte slots of the terms
        const int hl = i & 1, ss = (i >> 1) % NCHT, qq = i / (2 * NCHT);
        int qi = q0 + (qq < nqc ? qq : nqc - 1);
        if (a.qlist) qi = (int)a.qlist[qi];
        const int e0 = 16 * ss + 8 * hl;
        const float *qp = a.Q + (int64_t)qi * a.D + e0;
        float v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // (widths between two kernels' K steps: zeros beyond D, as in the rows' operand copy)
        if (e0 < a.D) *reinterpret_cast<float4 *>(&v[0]) = *reinterpret_cast<const float4 *>(qp);
        if (e0 + 4 < a.D) *reinterpret_cast<float4 *>(&v[4]) = *reinterpret_cast<const float4 *>(qp + 4);
        bf16x8 h, l;
        fs_split(v, h, l);
        uint8_t *dst = ft_q + ((size_t)(((qq >> 5) * NCHT + ss) * NT) * 1024) + (size_t)(hl * 32 + (qq & 31)) * 16;
        *reinterpret_cast<bf16x8 *>(dst) = h;
        if constexpr (NT == 2) *reinterpret_cast<bf16x8 *>(dst + 1024) = l;
    }
    uint32_t *id_s = reinterpret_cast<uint32_t *>(thr_s + nb * 32);   // the blocks' query numbers
    if constexpr (!MAXMODE)
        for (int i = tid; i < nb * 32; i += 64 * NW) {
            const uint32_t id = i < nqc ? (a.qlist ? a.qlist[q0 + i] : (uint32_t)(q0 + i)) : 0u;
            id_s[i] = id;
            thr_s[i] = i < nqc ? a.thr[id] : __uint_as_float(0x7fc00000u);   // NaN: no comparison succeeds
        }
    __syncthreads();
    // groups of RT tiles: wave w of slice s takes groups s + slices (w + FT_WAVES i)
    const int64_t all_groups = (a.t1 + RT - 1) / RT, n_groups = MAXMODE ? a.n_sample : all_groups, stride = (int64_t)slices * NW;
    // the second-dispatched half of the workgroup loses every arbitration by age: static priority for it (MI355X_MICROARCH.md "two waves
    // per SIMD"; measured 0.65 -> 0.59 ms on the filter passes of 1000 queries x 1 M rows)
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    // sample maxima: FT_SLOTS disjoint row sets per query -- (wave, half) of the chunk's workgroups, and when those are fewer than
    // FT_SLOTS (many chunks: few slices each) every wave deals its tile groups round-robin to sub_slots sets of its own (a threshold
    // from 128 maxima for k = 100 let 8 000 - 22 000 rows per query through)
    const int wave_slots = slices * NW * 2, sub_slots = wave_slots >= a.nslots ? 1 : a.nslots / wave_slots;
    const int slot0 = ((slice * NW + wave) * 2 + lk) * sub_slots;
    int it = 0;
    uint32_t wcnt = 0;
    float mreg[MAXMODE && KH == 1 ? FT_NBMAX : 1];   // MAX mode: the wave's maxima per query block
#pragma unroll
    for (int b = 0; b < (MAXMODE && KH == 1 ? FT_NBMAX : 1); ++b) mreg[b] = FS_EMPTY;
    uint8_t *rec_w = reinterpret_cast<uint8_t *>(a.rec) + (size_t)wave_g * a.cap * 80;   // (wave-uniform)
    bf16x8 xa[RT][NCH][NA];
    f32x16 bias[RT];
    auto fetch = [&](int64_t g, int half) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int64_t t = g * RT + r;
            const int64_t tc = t < a.t1 ? t : a.t1 - 1;
            const uint4 *tp = a.pack + ((tc * NCHT + half * NCH) * 2) * 64 + lane;
            if constexpr (NCH >= 24 && NA == 1) {
                // a running pointer the compiler cannot see through: left alone it keeps one 64-bit address per K step in registers (the
                // 2 KB stride is beyond the loads' immediate offsets) -- 128 of them at 64 K steps, more than the operands themselves
#pragma unroll
                for (int s_ = 0; s_ < NCH; s_ += 2) {
                    asm volatile("" : "+v"(tp));
                    xa[r][s_][0] = __builtin_bit_cast(bf16x8, tp[0]);
                    if (s_ + 1 < NCH) xa[r][s_ + 1][0] = __builtin_bit_cast(bf16x8, tp[2 * 64]);
                    tp += 4 * 64;
                }
            } else {
#pragma unroll
                for (int s_ = 0; s_ < NCH; ++s_)
#pragma unroll
                    for (int x = 0; x < NA; ++x) {
                        const uint4 w = tp[(s_ * 2 + x) * 64];
                        xa[r][s_][x] = __builtin_bit_cast(bf16x8, w);
                    }
            }
            if (half != 0) continue;
            // the biases of the 16 rows a lane's accumulators stand for: row (e & 3) + 8 (e >> 2) + 4 lk of the tile
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 bb = *reinterpret_cast<const float4 *>(a.bias + tc * 32 + 8 * g4 + 4 * lk);
                if (t >= a.t1) bb = make_float4(FS_PAD_BIAS, FS_PAD_BIAS, FS_PAD_BIAS, FS_PAD_BIAS);
                bias[r][4 * g4] = bb.x; bias[r][4 * g4 + 1] = bb.y; bias[r][4 * g4 + 2] = bb.z; bias[r][4 * g4 + 3] = bb.w;
            }
        }
    };
    for (int64_t g = slice + (int64_t)slices * wave; g < n_groups; g += stride, ++it) {
        const int64_t g_rows = MAXMODE ? g * all_groups / a.n_sample : g;
        fetch(g_rows, 0);
        const uint32_t tile_row = (uint32_t)(g * RT * 32 + 4 * lk);   // first of a lane's rows in the group's first tile
        // (requesting the queries' operands two or three K steps ahead of their matrix instructions -- pinned with sched_barrier, across
        //  the block boundary -- or a block's operands at its start measured 0 .. 8 % SLOWER than the compiler's own order: read a K
        //  step's operands, wait, multiply; the partner wave of the SIMD covers the round trip)
        //  ONE wave per SIMD (NW == 4: 768 / 1024-d) has no partner: there the operands ARE requested PD K steps ahead, across the block
        //  boundary, through a ring of four register sets.
        constexpr bool RING = NW == 4 && NT == 1;
        constexpr int PD = 3;
        bf16x8 bq[RING ? 4 : 1];
        auto request = [&](int j_, int set) __attribute__((always_inline)) {
            bq[set] = *reinterpret_cast<const bf16x8 *>(ft_q + (size_t)j_ * 1024 + lane * 16);   // (the last PD of a group read the slack behind the operands: unused)
        };
        if constexpr (RING) {
#pragma unroll
            for (int j_ = 0; j_ < PD; ++j_) request(j_, j_);
        }
        auto block = [&](const int b, float &mx) __attribute__((always_inline)) {
            const uint8_t *qb = ft_q + (size_t)b * (NCHT * NT * 1024) + lane * 16;
            f32x16 acc[RT];
#pragma unroll
          for (int half = 0; half < KH; ++half) {
            if (half != 0) {   // (KH == 2: one query block per workgroup, so a group's halves are fetched once each)
                __builtin_amdgcn_sched_barrier(0);   // (not above the first half's matrix instructions: both halves do not fit the registers)
                asm volatile("" ::: "memory");
                fetch(g_rows, half);
            }
#pragma unroll
            for (int s_ = 0; s_ < NCH; ++s_) {
                bf16x8 qh;
                if constexpr (RING) {
                    request(b * NCHT + half * NCH + s_ + PD, (half * NCH + s_ + PD) & 3);
                    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the request down to its first use)
                    qh = bq[(half * NCH + s_) & 3];
                } else {
                    qh = *reinterpret_cast<const bf16x8 *>(qb + (s_ * NT) * 1024);
                }
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[r][s_][0], qh, (s_ == 0 && half == 0) ? bias[r] : acc[r], 0, 0, 0);   // b_x rides in as SrcC
                if constexpr (NPROD >= 2) {   // (the low terms on accumulators of their own, four independent chains: no faster)
#pragma unroll
                    for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[r][s_][1], qh, acc[r], 0, 0, 0);
                }
                if constexpr (NPROD == 3) {
                    const bf16x8 ql = *reinterpret_cast<const bf16x8 *>(qb + (s_ * NT + 1) * 1024);
#pragma unroll
                    for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[r][s_][0], ql, acc[r], 0, 0, 0);
                }
            }
          }
            if (KH > 1 && b + 1 < nb) {   // (more than one block with two halves: the first half again)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" ::: "memory");
                fetch(g_rows, 0);
            }
            // the maxima below are inline assembly: the compiler's hazard recogniser does not put the wait states between a matrix
            // instruction's result and a vector instruction that reads it in front of those (measured: rows lost), so they are spelled out
            static_assert(RT >= 1 && RT <= 4, "operand lists below");
            if constexpr (RT == 1) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]));
            if constexpr (RT == 2) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
            if constexpr (RT == 3) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
            if constexpr (RT == 4) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            const int qq = 32 * b + lj;
            if constexpr (MAXMODE) {
                float m = ft_max16(acc[0]);
#pragma unroll
                for (int r = 1; r < RT; ++r) m = fmaxf(m, ft_max16(acc[r]));
                // (32-bit offset from the scalar base: the 64-bit address arithmetic of a 64-lane scatter was a third of this pass)
                if (KH == 1 && sub_slots == 1) {   // one slot for all of the wave's groups: the maximum stays in a register until the rows are through (a lane's atomic
                    mx = fmaxf(mx, m);  // touches a cache line of its own -- 64 per instruction, once per block and GROUP they were half of this pass)
                } else {
                    const uint32_t off = (uint32_t)(q0 + qq) * (uint32_t)a.nslots + (uint32_t)((slot0 + (it & (sub_slots - 1))) & (a.nslots - 1));
                    if (qq < nqc && m > FS_EMPTY) atomicMax(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(a.smax) + off * 4u), f32_key(m));
                }
            } else {
                const float tb = thr_s[qq];
                const uint32_t qid = id_s[qq];   // (read beside the threshold: the hit path waits for nothing)
                float tmax[RT];   // (all of the block's tiles first: their chains interleave; the tests and the rare hit path after)
#pragma unroll
                for (int r = 0; r < RT; ++r) tmax[r] = ft_max16(acc[r]);
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const bool hit = tmax[r] >= tb;
                    const unsigned long long hm = __ballot(hit);
                    if (hm) {   // some row of the tile reaches some query's threshold: the lanes concerned write their 16 scores
                        const uint32_t pos = wcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                        wcnt += (uint32_t)__popcll(hm);
                        if (hit && pos < a.cap) {
                            // the wave's region as a scalar base + a 32-bit byte offset (one multiply per record instead of two 64-bit ones)
                            uint8_t *dst = rec_w + pos * 80u;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                *reinterpret_cast<uint4 *>(dst + 16 * j) = make_uint4(__float_as_uint(acc[r][4 * j]), __float_as_uint(acc[r][4 * j + 1]),
                                                                                      __float_as_uint(acc[r][4 * j + 2]), __float_as_uint(acc[r][4 * j + 3]));
                            *reinterpret_cast<uint4 *>(dst + 64) = make_uint4(qid, tile_row + (uint32_t)(32 * r), (uint32_t)qq, 0u);   // (query, first row, query within the chunk)
                        }
                    }
                }
            }
        };
        if constexpr (MAXMODE && KH == 1) {
#pragma unroll
            for (int b = 0; b < FT_NBMAX; ++b)
                if (b < nb) block(b, mreg[b]);
        } else {
#pragma unroll 1
            for (int b = 0; b < nb; ++b) block(b, mreg[0]);
        }
    }
    if constexpr (MAXMODE) {
        if (sub_slots == 1) {
            if constexpr (KH == 1) {
#pragma unroll
                for (int b = 0; b < FT_NBMAX; ++b) {
                    const int qq = 32 * b + lj;
                    const uint32_t off = (uint32_t)(q0 + qq) * (uint32_t)a.nslots + (uint32_t)(slot0 & (a.nslots - 1));
                    if (b < nb && qq < nqc && mreg[b] > FS_EMPTY) atomicMax(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(a.smax) + off * 4u), f32_key(mreg[b]));
                }
            }   // (two K halves: the maxima went out block by block)
        }
    } else {
        if (lane == 0) a.wcnt[wave_g] = wcnt;
    }
}

// margin of a query: Qb = (|q|^2 + max |x|^2) 1.001; w = what the omitted low terms can add up to for THIS query, by Cauchy-Schwarz
// against the longest row / the row with the largest rounding residue (ft_theta_kernel): the worst case e |x||q| per omitted term
// is two to three times larger than a query's own residues
template <bool IP>
__device__ __forceinline__ float ft_margin(float Qb, float w, float theta)
{
    float m = Qb * 0x1p-13f + 2.0f * w;
    if (IP) m += (2.0f + fabsf(theta)) * 0x1p-20f;   // the rounding of 1 - sum in the reference
    return m;
}
// one wave per query: theta = the k-th largest of its sample maxima, thr = theta - margin (NaN + redo when the bound does not hold)
template <bool IP, int NK>
__global__ __launch_bounds__(256) void ft_theta_kernel(const uint32_t *__restrict__ smax, const float *__restrict__ Q, int nq, int D, int k, int nprod,
                                                       const uint32_t *__restrict__ stats, const uint32_t *__restrict__ pstats, float loosen, float *__restrict__ thr, float *__restrict__ qbnd,
                                                       uint32_t *__restrict__ redo, uint32_t *__restrict__ cnt)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    uint32_t key[NK];   // NK x 64 sample maxima
#pragma unroll
    for (int j = 0; j < NK; ++j) key[j] = smax[(size_t)q * (NK * 64) + j * 64 + lane];
    const float qq = fs_qnorm(Q + (int64_t)q * D, D);
    const float Qb = (qq + __uint_as_float(stats[0])) * 1.001f;
    // nprod 2: (x1 + x2).(q - q1) <= (1 + e^2) max |x| |q - q1|; nprod 1: that and (x - x1).q <= max |x - x1| |q|
    float xq2 = 0.0f;
    if (nprod <= 2) xq2 = sqrtf(__uint_as_float(stats[0]) * fs_qlow(Q + (int64_t)q * D, D)) * 1.01f;
    if (nprod == 1) xq2 += sqrtf(__uint_as_float(pstats[2]) * qq) * 1.01f;
    float cut = __uint_as_float(0x7fc00000u);
    if (Qb > 0x1p-60f && Qb < 0x1p60f) {
        const uint32_t sel = fs_wave_select(key, k, k + k / 4 + 8);
        if (sel != 0u) {   // at least k slots hold a row
            const float theta = key_f32(sel);
            cut = theta - ft_margin<IP>(Qb, xq2, theta) - loosen * Qb;   // (loosen: test hook, "flat_f32_dbg" 32 -- lists run over, second attempts happen)
        }
    }
    if (lane == 0) {
        thr[q] = cut;
        qbnd[q] = Qb;
        qbnd[nq + q] = xq2;
        redo[q] = (cut == cut) ? 0u : 1u;   // (the query's flag and list counter start here, the pass's list lengths with query 0: two memsets less per pass)
        cnt[q] = 0u;
        if (q == 0) { cnt[2 * nq] = 0u; cnt[2 * nq + 1] = 0u; }
    }
}

// Records -> per-query candidate lists.  One workgroup per workgroup of the filter (its eight wave regions: one query chunk).  The
// scores at or above their query's threshold are counted per query in LDS AND kept there (score, row, query within the chunk: 10 bytes
// a hit, FT_STAGE of them); ONE global atomic per query with hits then reserves that many places of the query's list (an atomic per
// candidate measured 0.29 ms per million: returning device-scope atomics run at ~3.5 G/s whatever their addresses) and the staged hits
// are written behind the reserved base -- the 80-byte records are read once (2.7 M records of 1000 queries: 0.137 -> ~0.08 ms).  A
// workgroup with more hits than the stage holds reads its records a second time instead.
constexpr int FT_BUCKET_T = 1024;
constexpr int FT_STAGE = 12288;
__global__ __launch_bounds__(FT_BUCKET_T) void ft_bucket_kernel(const uint4 *__restrict__ rec, const uint32_t *__restrict__ wcnt, uint32_t cap,
                                                                const float *__restrict__ thr, uint32_t *__restrict__ cnt, uint2 *__restrict__ cand,
                                                                int chunks, int qper, int nq, uint32_t *__restrict__ redo, int nw,
                                                                const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ qcount, uint32_t fcap)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t ft_stage[];   // uint2 [FT_STAGE] (score, row), then uint16 [FT_STAGE] query within the chunk
    __shared__ uint32_t hist[32 * FT_NBMAX], base_s[32 * FT_NBMAX];
    __shared__ uint32_t nstage_s;
    uint2 *st_sr = reinterpret_cast<uint2 *>(ft_stage);
    uint16_t *st_q = reinterpret_cast<uint16_t *>(ft_stage + (size_t)FT_STAGE * sizeof(uint2));
    const int j = blockIdx.x, tid = threadIdx.x;
    const int chunk = (j >> 3) % chunks, q0 = chunk * qper;
    if (qlist) nq = (int)*qcount < qper ? (int)*qcount : qper;   // second attempt: one chunk of listed queries
    const int nqc = nq - q0 < qper ? nq - q0 : qper;
    if (nqc <= 0) return;
    for (int i = tid; i < nqc; i += FT_BUCKET_T) hist[i] = 0u;
    if (tid == 0) nstage_s = 0u;
    const int per = FT_BUCKET_T / nw, r = tid / per, t = tid % per;   // 128 (256) threads per wave region
    uint32_t n = wcnt[j * nw + r];
    if (n > cap) {   // the region ran full: the exact kernels answer the queries its workgroup held
        for (int i = t; i < nqc; i += per) redo[qlist ? qlist[q0 + i] : (uint32_t)(q0 + i)] = 1u;
        n = cap;
    }
    const uint4 *rp = rec + (size_t)(j * nw + r) * cap * 5;
    __syncthreads();
    for (uint32_t i = t; i < n; i += per) {
        const uint4 h = rp[(size_t)i * 5 + 4];
        const float tb = thr[h.x];
        uint32_t c = 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const uint4 s4 = rp[(size_t)i * 5 + jj];
            const uint32_t sv[4] = { s4.x, s4.y, s4.z, s4.w };
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (__uint_as_float(sv[e]) >= tb) {
                    ++c;
                    const uint32_t pos = atomicAdd(&nstage_s, 1u);
                    if (pos < (uint32_t)FT_STAGE) {
                        st_sr[pos] = make_uint2(sv[e], h.y + (uint32_t)(e + 8 * jj));
                        st_q[pos] = (uint16_t)h.z;
                    }
                }
            }
        }
        if (c) atomicAdd(&hist[h.z], c);
    }
    __syncthreads();
    const uint32_t nstage = nstage_s;
    for (int i = tid; i < nqc; i += FT_BUCKET_T) {
        const uint32_t c = hist[i];
        base_s[i] = c ? atomicAdd(&cnt[qlist ? qlist[q0 + i] : (uint32_t)(q0 + i)], c) : 0u;
        hist[i] = 0u;
    }
    __syncthreads();
    if (nstage <= (uint32_t)FT_STAGE) {   // (workgroup-uniform)
        for (uint32_t i = tid; i < nstage; i += FT_BUCKET_T) {
            const uint32_t ql = st_q[i];
            const uint32_t q = qlist ? qlist[q0 + ql] : (uint32_t)(q0 + ql);
            const uint32_t pos = base_s[ql] + atomicAdd(&hist[ql], 1u);
            if (pos < fcap) cand[(size_t)q * fcap + pos] = st_sr[i];
        }
        return;
    }
    for (uint32_t i = t; i < n; i += per) {   // more hits than the stage holds: the records once more
        const uint4 h = rp[(size_t)i * 5 + 4];
        const float tb = thr[h.x];
        const uint32_t ql = h.z;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const uint4 s4 = rp[(size_t)i * 5 + jj];
            const uint32_t sv[4] = { s4.x, s4.y, s4.z, s4.w };
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (__uint_as_float(sv[e]) >= tb) {
                    const uint32_t pos = base_s[ql] + atomicAdd(&hist[ql], 1u);
                    if (pos < fcap) cand[(size_t)h.x * fcap + pos] = make_uint2(sv[e], h.y + (uint32_t)(e + 8 * jj));
                }
            }
        }
    }
}

__device__ __forceinline__ uint32_t ft_dist_key(float d)
{
    const uint32_t dk = f32_key(d);
    return dk >= 0xfffffff0u ? 0xffffffefu : dk;   // (NaN patterns: keep the absent-row codes free)
}

// one workgroup per query.  The candidates carry their matrix-core scores: the k-th largest of them (k distinct rows reach it) is
// theta, found by a radix select over the keys (8 bits a round, counted in LDS); only the candidates at or above theta - margin --
// k and a few, not the 1-2 k the sample's threshold let through -- get exact reference-order distances (32 pieces 1 KB apart per
// row in the blocked layout) and are ranked by (distance, row) by counting.
template <bool IP, int LANES>
__global__ __launch_bounds__(256) void ft_finish_kernel(const float *__restrict__ X, int64_t n, int D, const float *__restrict__ Q, int k,
                                                        float *__restrict__ thr, const float *__restrict__ qbnd, uint32_t *__restrict__ cnt,
                                                        const uint2 *__restrict__ cand, float *__restrict__ out_d, int64_t *__restrict__ out_i,
                                                        uint32_t *__restrict__ redo, uint32_t *__restrict__ rcount, uint32_t *__restrict__ rlist, int rcap, int second, int nqb,
                                                        uint32_t *__restrict__ bcount, uint32_t *__restrict__ blist, int rowmajor)
{
    __shared__ unsigned long long sel[FT_KEEP];
    __shared__ __attribute__((aligned(16))) float q_s[FT_DMAX];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t pick_s[2];
    __shared__ int m2_s;
    const int tid = threadIdx.x, lane = tid & 63;
    int q = blockIdx.x;
    if (second) {   // the queries the first attempt listed
        if ((uint32_t)q >= *rcount || q >= rcap) return;
        q = (int)rlist[q];
    }
    const float cut = thr[q];
    uint32_t nc = cnt[q];
    const int64_t want = k < n ? k : n;
    if (redo[q] != 0u) return;
    // A list that ran over (a sharp shell of neighbours the sample's threshold does not see: 46 000 rows above it for one query of
    // 1000 on clustered 300-d rows) still holds FT_CAP rows that reach the threshold: their k-th best is a lower bound of the k-th
    // best score too, and a much tighter one.  The first attempt lists such a query for ONE second filter pass under the new
    // threshold (rcap queries at most); the exact kernels -- a millisecond and more per query -- answer only what runs over twice.
    const bool over = nc > (uint32_t)FT_CAP;
    if (!(cut == cut) || (int64_t)nc < want || (over && (second || !rlist))) {   // workgroup-uniform: the exact kernels answer this query
        if (tid == 0) redo[q] = 1u;
        return;
    }
    if (over) nc = (uint32_t)FT_CAP;
    for (int i = tid; i < D; i += 256) q_s[i] = Q[(int64_t)q * D + i];
    if (tid == 0) m2_s = 0;
    constexpr int PER = FT_CAP / 256;
    uint32_t key[PER], row[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t i = (uint32_t)(tid + j * 256);
        key[j] = 0u; row[j] = 0u;
        if (i < nc) {
            const uint2 c = cand[(size_t)q * FT_CAP + i];
            key[j] = f32_key(__uint_as_float(c.x));
            key[j] = key[j] == 0u ? 1u : key[j];
            row[j] = c.y;
        }
    }
    // radix select of the want-th largest key: prefix / mask grow by 8 bits a round.  The candidates' scores lie in a narrow band: the
    // rounds start at the first byte in which two keys differ (with all of them in ONE bin of the top bytes a round was 2 700 LDS
    // atomics on one address)
    uint32_t kmx = 0u, kmn = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (key[j] != 0u) { kmx = key[j] > kmx ? key[j] : kmx; kmn = key[j] < kmn ? key[j] : kmn; }
    kmx = fs_wave_max_u32(kmx); kmn = fs_wave_min_u32(kmn);
    if (lane == 0) { hist[tid >> 6] = kmx; hist[4 + (tid >> 6)] = kmn; }
    __syncthreads();
    kmx = max(max(hist[0], hist[1]), max(hist[2], hist[3]));
    kmn = min(min(hist[4], hist[5]), min(hist[6], hist[7]));
    __syncthreads();
    const int top_byte = kmx == kmn ? -1 : (31 - __builtin_clz(kmx ^ kmn)) >> 3;   // -1: all keys equal
    uint32_t mask = top_byte >= 3 ? 0u : (0xffffffffu << (8 * (top_byte + 1)));
    uint32_t prefix = kmx & mask, remaining = (uint32_t)want;
    for (int shift = 8 * top_byte; shift >= 0; shift -= 8) {
        hist[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (key[j] != 0u && (key[j] & mask) == prefix) atomicAdd(&hist[(key[j] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid < 64) {   // bins 4 lane .. 4 lane + 3; suffix sums from the top bin down
            const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            uint32_t above = mine;   // inclusive suffix sum over lanes >= this one
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = (uint32_t)__shfl_down((int)above, o, 64);
                if (lane + o < 64) above += v;
            }
            const uint32_t excl = above - mine;   // keys in bins above this lane's four
            if (excl < remaining && remaining <= above) {   // exactly one lane
                uint32_t acc_ = excl, bin = 0u;
                const uint32_t hh[4] = { h0, h1, h2, h3 };
#pragma unroll
                for (int b = 3; b >= 0; --b) {
                    if (acc_ < remaining && remaining <= acc_ + hh[b]) { bin = (uint32_t)(4 * lane + b); pick_s[1] = remaining - acc_; }
                    acc_ += hh[b];
                }
                pick_s[0] = bin;
            }
        }
        __syncthreads();
        prefix |= pick_s[0] << shift;
        mask |= 255u << shift;
        remaining = pick_s[1];
        __syncthreads();
    }
    const float theta = key_f32(prefix);
    const float cut2 = theta - ft_margin<IP>(qbnd[q], qbnd[nqb + q], theta);
    if (over) {   // (workgroup-uniform)
        if (tid == 0) {
            const uint32_t pos = cut2 > cut ? atomicAdd(rcount, 1u) : 0xffffffffu;
            if (pos < (uint32_t)rcap) {
                thr[q] = cut2;
                cnt[q] = 0u;
                rlist[pos] = (uint32_t)q;
            } else {
                redo[q] = 1u;
            }
        }
        return;
    }
    const uint32_t cut2_key = f32_key(cut2);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (key[j] != 0u && key[j] >= cut2_key) {
            const int pos = atomicAdd(&m2_s, 1);
            if (pos < FT_KEEP) sel[pos] = row[j];
        }
    }
    __syncthreads();
    const int m2 = m2_s;
    if (m2 > FT_KEEP) {   // more rows inside the margin band than this kernel ranks (tight scores on wide rows; masses of near ties): 2 = ft_finish_big_kernel tries (up to FT_KEEP_BIG)
        if (tid == 0) {
            redo[q] = 2u;
            if (bcount) blist[atomicAdd(bcount, 1u)] = (uint32_t)q;   // (at most one entry per query of the pass)
            else redo[q] = 1u;
        }
        return;
    }
    unsigned long long mine[FT_KEEP / 256];
#pragma unroll
    for (int j = 0; j < FT_KEEP / 256; ++j) {
        const int i = tid + j * 256;
        mine[j] = ~0ull;
        if (i < m2) {
            const uint32_t r = (uint32_t)sel[i];
            if ((int64_t)r < n) mine[j] = ((unsigned long long)ft_dist_key(fs_exact<IP, LANES>(X, D, r, reinterpret_cast<const float4 *>(q_s), rowmajor != 0)) << 32) | r;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FT_KEEP / 256; ++j)
        if (tid + j * 256 < m2) sel[tid + j * 256] = mine[j];
    __syncthreads();
    // rank by counting: the keys are distinct (the row is part of them)
#pragma unroll
    for (int j = 0; j < FT_KEEP / 256; ++j) {
        if (tid + j * 256 < m2 && mine[j] != ~0ull) {
            int rank = 0;
            for (int i = 0; i < m2; ++i) rank += sel[i] < mine[j] ? 1 : 0;
            if (rank < k) {
                out_d[(int64_t)q * k + rank] = key_f32((uint32_t)(mine[j] >> 32));
                out_i[(int64_t)q * k + rank] = (int64_t)(uint32_t)mine[j];
            }
        }
    }
    for (int i = (int)want + tid; i < k; i += 256) {   // fewer rows than k
        out_d[(int64_t)q * k + i] = __uint_as_float(0x7f800000u);
        out_i[(int64_t)q * k + i] = -1;
    }
}


// k = 129 .. 2048: one workgroup of 1024 threads per query, the candidates' keys in LDS (FT_CAP_BIG of them), the same steps as above --
// radix select of the k-th largest score, the rows at or above it minus the margin (at most FT_KEEP_BIG) get exact distances -- and a
// bitonic sort of the (distance, row) pairs instead of ranking by counting.
__device__ __forceinline__ void ft_bitonic_u64(unsigned long long *e, int np, int nthreads)
{
    for (int size = 2; size <= np; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < np / 2; i += nthreads) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a0 = e[lo], a1 = e[hi];
                if ((a0 > a1) == up) { e[lo] = a1; e[hi] = a0; }
            }
        }
    }
    __syncthreads();
}
template <bool IP, int LANES>
__global__ __launch_bounds__(1024) void ft_finish_big_kernel(const float *__restrict__ X, int64_t n, int D, const float *__restrict__ Q, int k,
                                                             const float *__restrict__ thr, const float *__restrict__ qbnd, const uint32_t *__restrict__ cnt,
                                                             const uint2 *__restrict__ cand, float *__restrict__ out_d, int64_t *__restrict__ out_i,
                                                             uint32_t *__restrict__ redo, int nqb, int cstride, int only2, const uint32_t *__restrict__ bcount, const uint32_t *__restrict__ blist, int rowmajor)
{
    // only2: the second chance of the queries ft_finish_kernel marked 2 (their lists are whole, their margin band holds more than FT_KEEP rows):
    // cstride = that kernel's list stride; the mark becomes 0 (answered here) or 1 (the exact kernels)
    extern __shared__ __attribute__((aligned(16))) uint32_t fb_keys[];   // [FT_CAP_BIG] keys; later [FT_KEEP_BIG] (distance, row) pairs
    __shared__ __attribute__((aligned(16))) float q_s[FT_DMAX];
    __shared__ uint32_t rows_s[FT_KEEP_BIG];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t pick_s[2];
    __shared__ int m2_s;
    constexpr int NTH = 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t n_list = only2 ? *bcount : 0u;
    for (uint32_t it = blockIdx.x; it < (only2 ? n_list : gridDim.x); it += gridDim.x) {   // (only2: a fixed grid walks the list -- an empty list costs one round of empty workgroups)
    __syncthreads();
    const int q = only2 ? (int)blist[it] : (int)it;
    [&]() {
    const float cut = thr[q];
    const uint32_t nc = cnt[q];
    const int64_t want = k < n ? k : n;
    if (only2 ? redo[q] != 2u : redo[q] != 0u) return;
    if (!(cut == cut) || nc > (uint32_t)FT_CAP_BIG || (int64_t)nc < want) {   // workgroup-uniform: the exact kernels answer this query
        if (tid == 0) redo[q] = 1u;
        return;
    }
    for (int i = tid; i < D; i += NTH) q_s[i] = Q[(int64_t)q * D + i];
    if (tid == 0) m2_s = 0;
    const uint2 *cq = cand + (size_t)q * (size_t)cstride;
    uint32_t kmx = 0u, kmn = 0xffffffffu;
    for (uint32_t i = tid; i < nc; i += NTH) {
        uint32_t key = f32_key(__uint_as_float(cq[i].x));
        key = key == 0u ? 1u : key;
        fb_keys[i] = key;
        kmx = key > kmx ? key : kmx; kmn = key < kmn ? key : kmn;
    }
    kmx = fs_wave_max_u32(kmx); kmn = fs_wave_min_u32(kmn);
    if (lane == 0) { hist[tid >> 6] = kmx; hist[16 + (tid >> 6)] = kmn; }
    __syncthreads();
    kmx = 0u; kmn = 0xffffffffu;
    for (int w = 0; w < 16; ++w) { kmx = hist[w] > kmx ? hist[w] : kmx; kmn = hist[16 + w] < kmn ? hist[16 + w] : kmn; }
    __syncthreads();
    const int top_byte = kmx == kmn ? -1 : (31 - __builtin_clz(kmx ^ kmn)) >> 3;
    uint32_t mask = top_byte >= 3 ? 0u : (0xffffffffu << (8 * (top_byte + 1)));
    uint32_t prefix = kmx & mask, remaining = (uint32_t)want;
    for (int shift = 8 * top_byte; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        for (uint32_t i = tid; i < nc; i += NTH) {
            const uint32_t key = fb_keys[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {   // bins 4 lane .. 4 lane + 3; suffix sums from the top bin down
            const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            uint32_t above = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = (uint32_t)__shfl_down((int)above, o, 64);
                if (lane + o < 64) above += v;
            }
            const uint32_t excl = above - mine;
            if (excl < remaining && remaining <= above) {
                uint32_t acc_ = excl, bin = 0u;
                const uint32_t hh[4] = { h0, h1, h2, h3 };
#pragma unroll
                for (int b = 3; b >= 0; --b) {
                    if (acc_ < remaining && remaining <= acc_ + hh[b]) { bin = (uint32_t)(4 * lane + b); pick_s[1] = remaining - acc_; }
                    acc_ += hh[b];
                }
                pick_s[0] = bin;
            }
        }
        __syncthreads();
        prefix |= pick_s[0] << shift;
        mask |= 255u << shift;
        remaining = pick_s[1];
        __syncthreads();
    }
    const float theta = key_f32(prefix);
    const uint32_t cut2_key = f32_key(theta - ft_margin<IP>(qbnd[q], qbnd[nqb + q], theta));
    for (uint32_t i = tid; i < nc; i += NTH) {
        if (fb_keys[i] >= cut2_key) {
            const int pos = atomicAdd(&m2_s, 1);
            if (pos < FT_KEEP_BIG) rows_s[pos] = cq[i].y;
        }
    }
    __syncthreads();
    const int m2 = m2_s;
    if (m2 > FT_KEEP_BIG) {   // masses of near ties
        if (tid == 0) redo[q] = 1u;
        return;
    }
    int np2 = 256;
    while (np2 < m2) np2 <<= 1;
    unsigned long long *sel = reinterpret_cast<unsigned long long *>(fb_keys);   // (every thread has passed the barrier behind the last read of the keys)
    for (int i = tid; i < np2; i += NTH) {
        unsigned long long e = ~0ull;
        if (i < m2) {
            const uint32_t r = rows_s[i];
            if ((int64_t)r < n) e = ((unsigned long long)ft_dist_key(fs_exact<IP, LANES, 16>(X, D, r, reinterpret_cast<const float4 *>(q_s), rowmajor != 0)) << 32) | r;
        }
        sel[i] = e;
    }
    ft_bitonic_u64(sel, np2, NTH);
    if (only2 && tid == 0) redo[q] = 0u;
    for (int i = tid; i < k; i += NTH) {
        const unsigned long long e = i < np2 ? sel[i] : ~0ull;
        if ((uint32_t)(e >> 32) < 0xfffffff0u && i < want) {
            out_d[(int64_t)q * k + i] = key_f32((uint32_t)(e >> 32));
            out_i[(int64_t)q * k + i] = (int64_t)(uint32_t)e;
        } else {
            out_d[(int64_t)q * k + i] = __uint_as_float(0x7f800000u);
            out_i[(int64_t)q * k + i] = -1;
        }
    }
    }();
    }
}

}  // namespace

// ---- host side ----
static std::atomic<int> g_ft_on{4};        // cvtmi_set_tuning("flat_f32_tfilter"): 0 = the stream kernels for every batch, 1 .. 3 = products, 4 = choose
static std::atomic<int> g_ft_min_rows{262144};   // "flat_f32_tfilter_min_rows": smallest table that takes the pipeline
// "flat_f32_tfilter_sample": the sample is about 1 / this of the rows; 0 (default) = 3 .. 32 by the table's bytes and k -- the sample pass costs bytes / div, the
// candidates k x div: sqrt(1280 x GB of rows / k) within 3 .. 32 (tools/f32_sample_sweep.py, profiles/r06_f32_sample_sweep.txt: 1 M x 128-d k = 10 0.41 -> 0.36 ms per
// 1000 queries, 4 M x 128-d 1.31 -> 1.04, 524 288 x 512-d 0.82 -> 0.71; k = 100 keeps 5: above 8 its lists run over on tight data)
static std::atomic<int> g_ft_sample_div{0};
static int ft_sample_div(int k, int64_t n, int D)
{
    const int v = g_ft_sample_div.load();
    if (v) return v;
    const double want = std::sqrt(1280.0 * ((double)n * D * 4e-9) / (double)k);
    return std::max(3, std::min(32, (int)(want + 0.5)));   // (3, not 5, since the sample pass keeps its maxima in registers: 1 M x 128-d, k = 100 0.419 -> 0.403 ms)
}
static std::atomic<int> g_ft_bigk{1};      // "flat_f32_tfilter_bigk": 1 = k = 129 .. 2048 through the pipeline (4096 sample maxima, lists of 32 768, ft_finish_big_kernel), 0 = exact kernels
static std::atomic<int> g_ft_wide_band{1};   // "flat_f32_tfilter_wide_band": 1 = queries with more than FT_KEEP rows inside the margin band get a second finish (ft_finish_big_kernel), 0 = the exact kernels
static std::atomic<int> g_ft_retry{0};     // "flat_f32_tfilter_retry": 1 = a second filter pass for the queries whose candidate lists ran over, 0 (default) = the exact kernels at once
static std::atomic<int> g_ft_one_max{1 << 30}; // "flat_f32_tfilter_one": largest batch that multiplies one product (with the staged bucket pass one product wins at every
                                           // batch size measured: 1000 queries 0.74 -> 0.64 ms, 4096 2.9 -> 2.5, 10 000 7.0 -> 6.5; two products beyond this many queries)
static std::atomic<int> g_ft_min_nq{0};    // cvtmi_set_tuning("flat_f32_tfilter_min"): smallest batch that takes the pipeline; 0 = choose (ft_auto_min): 65 - 97 at the widths the
                                           // stream kernels take (up to 64 queries they stream the operand copy's first terms: 1 M x 128-d, 16 / 64 queries
                                           // 0.079 / 0.096 ms against 0.112 / 0.125 here, level at 80-96), 16 elsewhere (against the exact kernels)
void set_flat_f32_tfilter(int v) { g_ft_on = v < 0 ? 0 : (v > 4 ? 4 : v); }
void set_flat_f32_tfilter_one(int v) { g_ft_one_max = v < 0 ? 0 : v; }
void set_flat_f32_tfilter_wide_band(int v) { g_ft_wide_band = v != 0; }
void set_flat_f32_tfilter_retry(int v) { g_ft_retry = v != 0; }
void set_flat_f32_tfilter_bigk(int v) { g_ft_bigk = v != 0; }
void set_flat_f32_tfilter_min_rows(int v) { g_ft_min_rows = v < 32768 ? 32768 : v; }
int64_t flat_f32_tfilter_min_rows() { return g_ft_min_rows.load(); }
void set_flat_f32_tfilter_sample(int v) { g_ft_sample_div = v < 0 ? 0 : (v > 64 ? 64 : v); }
void set_flat_f32_tfilter_min(int v) { g_ft_min_nq = v < 0 ? 0 : v; }
// widths: the K steps (16 dimensions each) of a row tile stay in a wave's registers (RT tiles of 32 rows: RT x K steps x terms x 4 registers
// <= 128, 256 with one wave per SIMD; 96 / 128 K steps in two halves of 48 / 64).  A kernel exists for 2 / 4 / 6 / 8 / 10 / 12 / 16 / 24 / 32 /
// 48 / 64 / 96 / 128 K steps; a width in between (any multiple of 4 up to 2048: 100-d, 200-d, 300-d ...) runs on the next one over
// zero-padded operands.  0: no kernel
int flat_f32_tfilter_nch(int D)
{
    static const int steps[] = { 2, 4, 6, 8, 10, 12, 16, 24, 32, 48, 64, 96, 128 };
    if (D < 4 || D > 2048 || D % 4 != 0) return 0;
    for (int s_ : steps)
        if (16 * s_ >= D) return s_;
    return 0;
}
bool flat_f32_tfilter_width(int D) { return flat_f32_tfilter_nch(D) != 0; }
// smallest batch under "flat_f32_tfilter_min" 0: beyond what one pass of the private-ring stream takes over the operand copy where that is
// ahead (128-d: 96 queries 0.125 against 0.143 ms here; 64-d: 80 queries 0.137 against 0.129), 16 at widths the stream does not take
static int ft_auto_min(int D, int k)
{
    // (no stream kernel: the exact kernels are the alternative -- tools/f32_tiny_batches.py, profiles/r06_f32_tiny_batches.txt: one query over 524 288 x 512-d
    //  0.35 -> 0.17 ms, over 1 M x 100-d 0.16 -> 0.10; rows wider than 512-d with more than 32 neighbours stay at 16: on tight rows a query's exact finish
    //  costs more there than its share of an exact scan)
    if (flat_f32_stream_qmax(D) == 0) return (D <= 512 || k <= 32) ? 1 : 16;
    return D >= 96 ? std::max(65, flat_f32_stream_private_max(D) + 1) : 65;
}
bool flat_f32_tfilter_applies(int metric, int D, int64_t n, int64_t nq, int k)
{
    return g_ft_on.load() && (metric == CVTMI_METRIC_IP || metric == CVTMI_METRIC_L2F) && flat_f32_tfilter_width(D) &&
           // (k > 128: nothing else is fast on a small table either -- from 65 536 rows and 48 k, where the sample still fills its slots; measured in
           //  tools/flat_bigk_small_tables.py, profiles/r06_flat_bigk_small_tables.txt: 100 000 x 128-d, 1000 queries, k = 129 2.3 -> 0.58 ms)
           // (k <= 128 at the widths the stream kernels do not take: the exact kernels are all there is under 262 144 rows -- from 65 536 rows on,
           //  rows wider than 512-d with more than 32 neighbours from 129 queries; tools/f32_small_tables.py, profiles/r06_f32_small_tables.txt:
           //  100 000 x 512-d, 1000 queries, k = 10 2.97 -> 0.30 ms)
           (n >= g_ft_min_rows.load() || (k > 128 && n >= std::max<int64_t>(65536, 48 * (int64_t)k) && (D <= 512 || nq >= 16)) ||
            (k <= 128 && n >= std::min<int64_t>(65536, g_ft_min_rows.load()) && flat_f32_stream_qmax(D) == 0 && (D <= 512 || k <= 32 || nq >= 129))) && n < 0xffffffe0LL &&
           k >= 1 && k <= CVTMI_K_MAX && g_ft_bigk.load() + (k <= 128) > 0 &&   // (k > 128: the stream kernels do not take it -- every batch size comes here)
           (k > 128 || nq >= (g_ft_min_nq.load() > 0 ? g_ft_min_nq.load() : ft_auto_min(D, k)));
}
// records a wave region holds: three times what 1 M SIFT-like rows gave per wave at k = 100, scaled with k beyond 128
static uint32_t ft_rec_cap(int64_t m, int k)
{
    const int64_t f = k <= 128 ? 1 : 1 + k / 256;
    return (uint32_t)std::min<int64_t>(8192, std::max<int64_t>(256, 3 * m * f));
}
size_t flat_f32_tfilter_scratch(int64_t nq, int k)
{
    const bool big = k > 128;
    const int64_t m = std::min<int64_t>(nq, big ? FT_PASS_BIG : FT_PASS);
    return (size_t)m * ((big ? FT_SLOTS_BIG : FT_SLOTS) + 8) * sizeof(uint32_t) + (size_t)m * (big ? FT_CAP_BIG : FT_CAP) * sizeof(uint2) +
           (size_t)FT_GRID * FT_WAVES * (sizeof(uint32_t) + (size_t)ft_rec_cap(m, k) * 80) + 1024;
}

template <int NCH, int NPROD, int RT, int NW = FT_WAVES, int KH = 1>
static int ft_launch(bool maxmode, const FtArgs &a, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attr_a[16] = {}, attr_b[16] = {};
    if (maxmode) {
        CVTMI_TRY(fs_set_lds((const void *)flat_f32_tfilter_kernel<NCH, NPROD, RT, true, NW, KH>, 163840, attr_a));
        hipLaunchKernelGGL((flat_f32_tfilter_kernel<NCH, NPROD, RT, true, NW, KH>), dim3(FT_GRID), dim3(64 * NW), lds, st, a);
    } else {
        CVTMI_TRY(fs_set_lds((const void *)flat_f32_tfilter_kernel<NCH, NPROD, RT, false, NW, KH>, 163840, attr_b));
        hipLaunchKernelGGL((flat_f32_tfilter_kernel<NCH, NPROD, RT, false, NW, KH>), dim3(FT_GRID), dim3(64 * NW), lds, st, a);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}
// products a width can multiply (two / three need both terms of a row tile in registers)
static int ft_products(int D, int want)
{
    const int nch = flat_f32_tfilter_nch(D), most = nch <= 8 ? 3 : (nch <= 16 ? 2 : 1);
    return want < most ? want : most;
}
// row tiles per group of the kernel ft_launch_any picks (the table below)
static int ft_rt(int nch, int nprod)
{
    if (nprod == 1) return nch <= 4 ? 4 : (nch <= 10 ? 3 : (nch <= 16 ? 2 : 1));
    if (nch == 2) return 4;
    if (nch == 4) return 3;
    if (nch <= 8) return 2;
    return 1;
}
static int ft_launch_any(int D, int nprod, bool maxmode, const FtArgs &a, size_t lds, hipStream_t st)
{
    switch (flat_f32_tfilter_nch(D) * 4 + nprod) {
    case 2 * 4 + 1: return ft_launch<2, 1, 4>(maxmode, a, lds, st);
    case 2 * 4 + 2: return ft_launch<2, 2, 4>(maxmode, a, lds, st);
    case 2 * 4 + 3: return ft_launch<2, 3, 4>(maxmode, a, lds, st);
    case 4 * 4 + 1: return ft_launch<4, 1, 4>(maxmode, a, lds, st);
    case 4 * 4 + 2: return ft_launch<4, 2, 3>(maxmode, a, lds, st);
    case 4 * 4 + 3: return ft_launch<4, 3, 3>(maxmode, a, lds, st);
    case 6 * 4 + 1: return ft_launch<6, 1, 3>(maxmode, a, lds, st);
    case 6 * 4 + 2: return ft_launch<6, 2, 2>(maxmode, a, lds, st);
    case 6 * 4 + 3: return ft_launch<6, 3, 2>(maxmode, a, lds, st);
    case 8 * 4 + 1: return ft_launch<8, 1, 3>(maxmode, a, lds, st);
    case 8 * 4 + 2: return ft_launch<8, 2, 2>(maxmode, a, lds, st);
    case 8 * 4 + 3: return ft_launch<8, 3, 2>(maxmode, a, lds, st);
    case 10 * 4 + 1: return ft_launch<10, 1, 3>(maxmode, a, lds, st);
    case 10 * 4 + 2: return ft_launch<10, 2, 1>(maxmode, a, lds, st);
    case 12 * 4 + 1: return ft_launch<12, 1, 2>(maxmode, a, lds, st);
    case 12 * 4 + 2: return ft_launch<12, 2, 1>(maxmode, a, lds, st);
    case 16 * 4 + 1: return ft_launch<16, 1, 2>(maxmode, a, lds, st);
    case 16 * 4 + 2: return ft_launch<16, 2, 1>(maxmode, a, lds, st);
    case 24 * 4 + 1: return ft_launch<24, 1, 1>(maxmode, a, lds, st);
    case 32 * 4 + 1: return ft_launch<32, 1, 1>(maxmode, a, lds, st);
    case 48 * 4 + 1: return ft_launch<48, 1, 1, 4>(maxmode, a, lds, st);   // 768-d, 1024-d: one wave per SIMD (512 registers)
    case 64 * 4 + 1: return ft_launch<64, 1, 1, 4>(maxmode, a, lds, st);   // 1024-d
    case 96 * 4 + 1: return ft_launch<48, 1, 1, 4, 2>(maxmode, a, lds, st);   // 1536-d, 2048-d: a row tile's K steps in two halves
    case 128 * 4 + 1: return ft_launch<64, 1, 1, 4, 2>(maxmode, a, lds, st);
    }
    return fail(CVTMI_EINVAL, "flat_f32_tfilter: no kernel for D=%d with %d products", D, nprod);
}

// nq queries against rows [0, n); results for the queries whose redo flag stays 0 (redo[nq] is zeroed here)
int launch_flat_f32_tfilter(int metric, int D, const float *X, const float *Xrows, const void *pack, const uint32_t *pstats, const float *bias, const uint32_t *stats, int64_t n,
                            const float *q, int64_t nq, int k, void *scratch, float *out_d, int64_t *out_i, uint32_t *redo, hipStream_t st)
{
    if (!flat_f32_tfilter_applies(metric, D, n, nq, k)) return fail(CVTMI_EINVAL, "flat_f32_tfilter: D=%d nq=%lld", D, (long long)nq);
    const float *Xe = Xrows ? Xrows : X;   // what the finish kernels gather exact distances from: the row-major copy when the handle keeps one
    const int rm = Xrows ? 1 : 0;
    const int mode = g_ft_on.load();
    // (k > 384: as many products as the width has -- the narrower margin keeps the rows that need exact distances near k: at k = 2048 one
    //  product put more than FT_KEEP_BIG rows inside the band and every query went to the exact kernels)
    //  (1000 queries, one / three products: k = 129 1.28 / 1.73 ms, 256 1.53 / 2.05, 1000 3.17 / 2.42, 2048 42 / 3.25)
    //  End of round 6 (the exact finish reads a row-major copy now, so rows inside a wide band cost less; tools/f32_products_by_k.py,
    //  profiles/r06_f32_products_by_k.txt, one / three products, ms per 1000 queries): 1 M x 128-d k = 512 1.13 / 1.35, 768 1.38 / 1.42, 1024 1.62 / 1.52,
    //  1536 26.9 / 1.68; 4 M: 768 2.16 / 3.67, 1024 2.47 / 3.87, 1536 6.4 / 4.1; 10 M: 768 4.1 / 8.5, 1536 5.5 / 9.0, 2048 31.9 / 9.5; 2 M x 256-d: 768 2.7 / 3.3,
    //  1024 3.50 / 3.56, 1536 258 / 4.1 -- three products cost in proportion to the rows, one product's band overflows somewhere past k = 1024: one up to 768
    const int nprod = ft_products(D, mode == 4 ? (k > 768 ? 3 : (nq <= g_ft_one_max.load() ? 1 : 2)) : mode);
    const int nt = nprod == 3 ? 2 : 1;
    const int nch = flat_f32_tfilter_nch(D);
    const int qcap = std::min(32 * FT_NBMAX, (int)((size_t)(160 * 1024 - 32 * FT_NBMAX * 8 - FT_SLACK) / ((size_t)nch * nt * 1024)) * 32);   // queries a workgroup holds
    const int64_t n_tiles = (n + 31) / 32;
    const bool big = k > 128;
    const int pass = big ? FT_PASS_BIG : FT_PASS, nslots = big ? FT_SLOTS_BIG : FT_SLOTS;
    const uint32_t fcap = big ? FT_CAP_BIG : FT_CAP;
    for (int64_t a0 = 0; a0 < nq; a0 += pass) {
        const int64_t m = std::min<int64_t>(nq - a0, pass);
        int chunks = 1;
        while (chunks < 32 && (m + chunks - 1) / chunks > qcap) chunks *= 2;
        const int qper = (int)(((m + chunks - 1) / chunks + 31) / 32 * 32);
        const int nw = nch > 32 ? 4 : FT_WAVES;   // waves per workgroup of the filter kernel
        const uint32_t cap = ft_rec_cap(m, k) * (uint32_t)(FT_WAVES / nw);   // (the record area is the same: fewer, larger regions)
        uint32_t *smax = reinterpret_cast<uint32_t *>(scratch);
        float *thr = reinterpret_cast<float *>(smax + (size_t)m * nslots);
        float *qbnd = thr + m;
        uint32_t *cnt = reinterpret_cast<uint32_t *>(qbnd + 2 * m);
        uint32_t *rlist = cnt + m;                 // queries of the second attempt, their number in front of the wave counters
        uint32_t *rcount = rlist + m;
        uint32_t *bcount = rcount + 1;             // queries whose margin band is wider than ft_finish_kernel ranks: their number, then the list
        uint32_t *blist = bcount + 1;
        uint32_t *wcnt = blist + m + (((uintptr_t)(blist + m) & 7) ? 1 : 0);   // (the candidate lists behind the wave counters start on 8 bytes)
        uint2 *cand = reinterpret_cast<uint2 *>(wcnt + FT_GRID * FT_WAVES);
        uint4 *rec = reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(cand + (size_t)m * fcap) + 256 - (((uintptr_t)(cand + (size_t)m * fcap)) & 15));
        CVTMI_HIP(hipMemsetAsync(smax, 0, (size_t)m * nslots * sizeof(uint32_t), st));
        FtArgs a;
        a.pack = reinterpret_cast<const uint4 *>(pack); a.bias = bias; a.n_tiles = n_tiles; a.Q = q + a0 * D; a.D = D; a.nq = (int)m; a.chunks = chunks; a.qper = qper;
        a.smax = smax; a.nslots = nslots; a.thr = thr; a.rec = rec; a.wcnt = wcnt; a.cap = cap; a.qlist = nullptr; a.qcount = nullptr; a.dbg = get_flat_f32_dbg();
        const size_t lds = (size_t)(qper / 32) * nch * nt * 1024 + FT_SLACK + (size_t)qper * 2 * sizeof(float);
        a.t1 = n_tiles;
        {   // the sample: about an eighth of the groups (at least ~65 536 rows), a whole number per wave of a chunk
            const int rt = ft_rt(nch, nprod);
            const int64_t all_groups = (n_tiles + rt - 1) / rt, streams = (int64_t)(FT_GRID / chunks) * nw;
            // (k > 128: a third of the rows -- the k-th largest of 4096 maxima needs many more rows than k behind it)
            const int64_t want = std::max<int64_t>(all_groups / (big ? std::min(3, ft_sample_div(k, n, D)) : ft_sample_div(k, n, D)), std::min<int64_t>(all_groups, (2048 + rt - 1) / rt));
            a.n_sample = std::min<int64_t>(all_groups, std::max<int64_t>(1, (want + streams / 2) / streams) * streams);
        }
        CVTMI_TRY(ft_launch_any(D, nprod, true, a, lds, st));
        const unsigned tg = (unsigned)((m + 3) / 4);
        const float loosen = (a.dbg & 32) ? 0.02f : 0.0f;
        if (metric == CVTMI_METRIC_IP) {
            if (big) hipLaunchKernelGGL((ft_theta_kernel<true, FT_SLOTS_BIG / 64>), dim3(tg), dim3(256), 0, st, smax, a.Q, (int)m, D, k, nprod, stats, pstats, loosen, thr, qbnd, redo + a0, cnt);
            else hipLaunchKernelGGL((ft_theta_kernel<true, FT_SLOTS / 64>), dim3(tg), dim3(256), 0, st, smax, a.Q, (int)m, D, k, nprod, stats, pstats, loosen, thr, qbnd, redo + a0, cnt);
        } else {
            if (big) hipLaunchKernelGGL((ft_theta_kernel<false, FT_SLOTS_BIG / 64>), dim3(tg), dim3(256), 0, st, smax, a.Q, (int)m, D, k, nprod, stats, pstats, loosen, thr, qbnd, redo + a0, cnt);
            else hipLaunchKernelGGL((ft_theta_kernel<false, FT_SLOTS / 64>), dim3(tg), dim3(256), 0, st, smax, a.Q, (int)m, D, k, nprod, stats, pstats, loosen, thr, qbnd, redo + a0, cnt);
        }
        a.t1 = n_tiles; a.n_sample = 0;
        CVTMI_TRY(ft_launch_any(D, nprod, false, a, lds, st));
        if (a.dbg & 8) continue;   // timing experiments: the filter passes alone (results stale)
        const bool retry = g_ft_retry.load() != 0 && !big;
        const size_t bucket_lds = (size_t)FT_STAGE * (sizeof(uint2) + sizeof(uint16_t));
        {
            static std::atomic<bool> attr_k[16] = {};
            CVTMI_TRY(fs_set_lds((const void *)ft_bucket_kernel, bucket_lds, attr_k));
        }
        const int rcap = std::min<int>(qcap, (int)((m + 31) / 32 * 32));   // queries one second attempt takes (a single chunk)
        const bool wide = !big && D >= 256 && g_ft_wide_band.load() != 0;   // (narrow rows: the exact kernels' turn costs less than this launch on every call)
        auto finish = [&](int second) {
            if (big || second == 2) {   // k = 129 .. 2048: the candidates' keys in LDS, a bitonic sort of the exact distances (2: the second chance of the other form's wide bands)
                const int cstride = big ? FT_CAP_BIG : FT_CAP, only2 = big ? 0 : 1;
                const unsigned bgrid = big ? (unsigned)m : (unsigned)std::min<int64_t>(m, 256);
                const size_t lds_b = (size_t)FT_CAP_BIG * sizeof(uint32_t);
                static std::atomic<bool> attr_f[3][16] = {};
                if (metric == CVTMI_METRIC_IP) {
                    (void)fs_set_lds((const void *)ft_finish_big_kernel<true, 4>, lds_b, attr_f[0]);
                    hipLaunchKernelGGL((ft_finish_big_kernel<true, 4>), dim3(bgrid), dim3(1024), lds_b, st, Xe, n, D, a.Q, k, thr, qbnd, cnt, cand, out_d + a0 * k, out_i + a0 * k, redo + a0, (int)m, cstride, only2, bcount, blist, rm);
                } else if (D % 16 == 0) {
                    (void)fs_set_lds((const void *)ft_finish_big_kernel<false, 8>, lds_b, attr_f[1]);
                    hipLaunchKernelGGL((ft_finish_big_kernel<false, 8>), dim3(bgrid), dim3(1024), lds_b, st, Xe, n, D, a.Q, k, thr, qbnd, cnt, cand, out_d + a0 * k, out_i + a0 * k, redo + a0, (int)m, cstride, only2, bcount, blist, rm);
                } else {
                    (void)fs_set_lds((const void *)ft_finish_big_kernel<false, 4>, lds_b, attr_f[2]);
                    hipLaunchKernelGGL((ft_finish_big_kernel<false, 4>), dim3(bgrid), dim3(1024), lds_b, st, Xe, n, D, a.Q, k, thr, qbnd, cnt, cand, out_d + a0 * k, out_i + a0 * k, redo + a0, (int)m, cstride, only2, bcount, blist, rm);
                }
                return;
            }
            const unsigned grid = second ? (unsigned)rcap : (unsigned)m;
            uint32_t *rl = retry ? rlist : nullptr;
            if (metric == CVTMI_METRIC_IP)
                hipLaunchKernelGGL((ft_finish_kernel<true, 4>), dim3(grid), dim3(256), 0, st, Xe, n, D, a.Q, k, thr, qbnd, cnt, cand, out_d + a0 * k, out_i + a0 * k, redo + a0, rcount, rl, rcap, second, (int)m, wide ? bcount : nullptr, blist, rm);
            else if (D % 16 == 0)   // (the reference's L2 sums in 8 lanes when D % 16 == 0, in 4 lanes otherwise: space_l2.h:40-151)
                hipLaunchKernelGGL((ft_finish_kernel<false, 8>), dim3(grid), dim3(256), 0, st, Xe, n, D, a.Q, k, thr, qbnd, cnt, cand, out_d + a0 * k, out_i + a0 * k, redo + a0, rcount, rl, rcap, second, (int)m, wide ? bcount : nullptr, blist, rm);
            else
                hipLaunchKernelGGL((ft_finish_kernel<false, 4>), dim3(grid), dim3(256), 0, st, Xe, n, D, a.Q, k, thr, qbnd, cnt, cand, out_d + a0 * k, out_i + a0 * k, redo + a0, rcount, rl, rcap, second, (int)m, wide ? bcount : nullptr, blist, rm);
        };
        hipLaunchKernelGGL(ft_bucket_kernel, dim3(FT_GRID), dim3(FT_BUCKET_T), bucket_lds, st, rec, wcnt, cap, thr, cnt, cand, chunks, qper, (int)m, redo + a0, nw, nullptr, nullptr, fcap);
        finish(0);
        if (retry) {   // the queries whose lists ran over, under the thresholds their own candidates give (nothing listed: three empty launches)
            FtArgs b = a;
            b.qlist = rlist; b.qcount = rcount; b.chunks = 1; b.qper = rcap;
            const size_t lds2 = (size_t)(rcap / 32) * nch * nt * 1024 + FT_SLACK + (size_t)rcap * 2 * sizeof(float);
            CVTMI_TRY(ft_launch_any(D, nprod, false, b, lds2, st));
            hipLaunchKernelGGL(ft_bucket_kernel, dim3(FT_GRID), dim3(FT_BUCKET_T), bucket_lds, st, rec, wcnt, cap, thr, cnt, cand, 1, rcap, (int)m, redo + a0, nw, rlist, rcount, fcap);
            finish(1);
        }
        if (wide) finish(2);
        CVTMI_HIP(hipGetLastError());
        if (getenv("CVTMI_FT_DEBUG")) {   // counts of the pass (synchronises)
            CVTMI_HIP(hipStreamSynchronize(st));
            std::vector<uint32_t> hc(m), hw(FT_GRID * FT_WAVES), hr(m);
            std::vector<float> ht(m);
            CVTMI_HIP(hipMemcpy(hc.data(), cnt, m * 4, hipMemcpyDeviceToHost));
            CVTMI_HIP(hipMemcpy(hw.data(), wcnt, hw.size() * 4, hipMemcpyDeviceToHost));
            CVTMI_HIP(hipMemcpy(hr.data(), redo + a0, m * 4, hipMemcpyDeviceToHost));
            CVTMI_HIP(hipMemcpy(ht.data(), thr, m * 4, hipMemcpyDeviceToHost));
            double sc = 0, sw = 0; uint32_t mc = 0, mw = 0, nr = 0;
            for (auto v : hc) { sc += v; mc = std::max(mc, v); }
            for (auto v : hw) { sw += v; mw = std::max(mw, v); }
            for (auto v : hr) nr += v != 0;
            uint32_t hrc = 0;
            CVTMI_HIP(hipMemcpy(&hrc, rcount, 4, hipMemcpyDeviceToHost));
            fprintf(stderr, "ft debug: nprod %d m %lld chunks %d qper %d cap %u: candidates mean %.0f max %u; records total %.0f per wave max %u; second attempts %u; redo %u; thr[0] %g\n",
                    nprod, (long long)m, chunks, qper, cap, sc / m, mc, sw, mw, hrc, nr, ht[0]);
        }
    }
    return CVTMI_OK;
}

}  // namespace cvtmi
