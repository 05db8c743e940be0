// flat_u8_tfilter.hip -- exhaustive uint8 L2 search (BruteforceSearch + L2SpaceI, brutoforce.hpp:73-93, space_l2.h:186-245) as a threshold
// filter (round 6): every batch with k = 129 .. 2048 and, for k <= 128, batches from 129 queries on.  The stream and filter
// pipelines of flat_mfma.hip stop at k = 128 / 64; behind them the exact kernel took one query per workgroup (2 M x 512-d, 1000 queries:
// k = 128 3.9 ms, k = 129 139 ms).  This is the fp32 threshold filter of flat_f32_tfilter.hip on v_mfma_i32_32x32x32_i8 over the packed
// rows (flat_u8_pack_kernel: x - 128 as int8, [tile][K step of 32][64 lanes] x 16 B):
//   * the scores are EXACT integers -- with x' = x - 128, q' = q - 128: d = |x'|^2 + |q'|^2 - 2 x'.q' -- so there is no margin and no
//     second evaluation: a row's accumulator starts at -(|x'|^2 >> 1) and ends as a = x'.q' - (|x'|^2 >> 1), d = |q'|^2 - 2 a + (|x'|^2 & 1);
//   * a workgroup keeps up to 256 queries in LDS (128 KB at 512-d), a wave RT row tiles in registers; a lane ends up with 16 values of
//     ITS query per tile: v_max3_i32 tree, one compare, 80-byte records into the wave's own region (no atomics);
//   * up to four query chunks per pass: workgroups with different queries walk the same rows in the same order on the same XCD, so the
//     rows leave HBM once per pass of 1024 queries;
//   * sample pass: up to 4096 maxima of a over disjoint row sets per query (a half wave's own slot, kept in registers until its rows are
//     through); the k-th largest, A, is reached by k distinct rows, whose distances are <= |q'|^2 - 2 A + 1 =: tau.  A row with d <= tau
//     has a >= A - (1 - parity) / 2, i.e. a >= A: the threshold is A itself;
//   * bucket pass: the records' values at or above the threshold become (distance, row) candidates, per query; finish: one workgroup
//     of 1024 threads per query, the distances in LDS, radix select of the k-th smallest, everything at or below it sorted by
//     (distance, row).
// A query whose sample has fewer than k filled slots, whose list runs over or whose k-th distance ties with more rows than the finish
// keeps raises ITS flag; a wave region that runs over raises the call's.  The caller reads the flags (the exact kernels take no predicate
// on this metric) and lets the round-5 paths answer the flagged queries, or the whole call.  Distances come out as int32 bits in the
// float array, as from every uint8 path.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "flat_f32_common.h"
#include "kernels.h"

namespace cvtmi {

namespace {

using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int UT_WAVES = 8, UT_GRID = 256, UT_SLOTS = 4096, UT_CAP = 32768, UT_KEEP = 4096, UT_QPER = 256, UT_NBMAX = 16, UT_STAGE = 12288;
constexpr int UT_NBQ = UT_QPER / 32;   // query blocks of a workgroup
constexpr int UT_NEG = -(1 << 30);   // start value of a padding row's accumulator: never reaches a threshold

struct UtArgs {
    const uint4 *pack; const int32_t *norms; int64_t n, n_tiles, n_sample;
    const uint8_t *Q; int D, nq, chunks, qper;
    uint32_t *smax;          // MAX mode: [nq][UT_SLOTS] (a ^ 0x80000000), zeroed by the caller
    const int32_t *thr;      // FILTER mode: [nq] (INT_MAX: nothing passes)
    uint4 *rec; uint32_t *wcnt; uint32_t cap;
    int dbg;   // CVTMI_UT_DBG: timing experiments (results wrong) 1 = no epilogue, 2 = rows loaded once; test hook (results right) 4 = record regions of two records
};

__device__ __forceinline__ int ut_max3(int a, int b, int c)
{
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int ut_max16(const i32x16 &v)
{
    // two chains of four (a chain of eight dependent instructions leaves the vector pipe idle between them)
    int m0 = ut_max3(v[0], v[1], v[2]), m1 = ut_max3(v[3], v[4], v[5]);
    m0 = ut_max3(m0, v[6], v[7]); m1 = ut_max3(m1, v[8], v[9]);
    m0 = ut_max3(m0, v[10], v[11]); m1 = ut_max3(m1, v[12], v[13]);
    m0 = ut_max3(m0, v[14], v[15]);
    return m0 > m1 ? m0 : m1;
}

// KS K steps of 32 dimensions, RT row tiles per wave
template <int KS, int RT, bool MAXMODE>
__global__ __launch_bounds__(64 * UT_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void flat_u8_tfilter_kernel(const UtArgs a)
{
    constexpr int NW = UT_WAVES;
    extern __shared__ __attribute__((aligned(16))) uint8_t ut_q[];   // [block][K step] x 1 KB, the blocks' thresholds, the waves' row terms, 3 KB of padding
    const int tid = threadIdx.x, lane = tid & 63, lj = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int chunk = idx % a.chunks, slices = 8 * ((int)(gridDim.x >> 3) / a.chunks), slice = xcd + 8 * (idx / a.chunks);
    const int q0 = chunk * a.qper;
    const int nqc = a.nq - q0 < a.qper ? a.nq - q0 : a.qper;
    const int wave_g = blockIdx.x * NW + wave;
    if (nqc <= 0) {
        if (!MAXMODE && lane == 0) a.wcnt[wave_g] = 0u;
        return;
    }
    const int nb = (nqc + 31) >> 5;
    int *thr_s = reinterpret_cast<int *>(ut_q + (size_t)nb * KS * 1024);
    for (int i = tid; i < nb * 32 * KS * 2; i += 64 * NW) {   // (query, K step, half) -> its 16-byte slot
        const int hl = i & 1, ss = (i >> 1) % KS, qq = i / (2 * KS);
        const int qi = q0 + (qq < nqc ? qq : nqc - 1);
        uint4 v = *reinterpret_cast<const uint4 *>(a.Q + (int64_t)qi * a.D + 32 * ss + 16 * hl);
        v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
        *reinterpret_cast<uint4 *>(ut_q + ((size_t)((qq >> 5) * KS + ss) * 1024) + (size_t)(hl * 32 + (qq & 31)) * 16) = v;
    }
    if constexpr (!MAXMODE)
        for (int i = tid; i < nb * 32; i += 64 * NW) thr_s[i] = i < nqc ? a.thr[q0 + i] : 0x7fffffff;
    __syncthreads();
    const int64_t all_groups = (a.n_tiles + RT - 1) / RT, n_groups = MAXMODE ? a.n_sample : all_groups, stride = (int64_t)slices * NW;
    const int slot = (slice * NW + wave) * 2 + lk;   // MAX mode: this half wave's own slot (UT_GRID x UT_WAVES x 2 = UT_SLOTS; one chunk, see the launcher)
    int mreg[UT_NBQ];                                // ... its maxima per query block, in registers until the rows are through
#pragma unroll
    for (int b = 0; b < UT_NBQ; ++b) mreg[b] = UT_NEG;
    uint32_t wcnt = 0;
    int none = 0;   // (the timing experiments' sink)
    uint8_t *rec_w = reinterpret_cast<uint8_t *>(a.rec) + (size_t)wave_g * a.cap * 80;
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    i32x4 xa[RT][KS];
    // the rows' starting values -(|x'|^2 >> 1) reach the accumulators through the wave's own LDS words (no registers held for them);
    // the queries' operands through a ring of R registers, R - 1 K steps ahead across the blocks of a group
    int *bias_w = thr_s + nb * 32 + wave * (RT * 32);
    constexpr int R = KS % 4 == 0 ? 4 : (KS % 3 == 0 ? 3 : (KS % 2 == 0 ? 2 : 1));   // (the ring runs on across the blocks: R divides KS; 1 = read, then use)
    const uint8_t *qp = ut_q + lane * 16;
    for (int64_t g = slice + (int64_t)slices * wave; g < n_groups; g += stride) {
        const int64_t g_rows = MAXMODE ? g * all_groups / a.n_sample : g;
        if (!(a.dbg & 2) || g < stride) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int64_t t = g_rows * RT + r;
            const int64_t tc = t < a.n_tiles ? t : a.n_tiles - 1;
            const uint4 *tp = a.pack + (tc * KS) * 64 + lane;
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) xa[r][s_] = __builtin_bit_cast(i32x4, tp[s_ * 64]);
        }
        }
#pragma unroll
        for (int i = lane; i < RT * 32; i += 64) {
            const int64_t row = g_rows * RT * 32 + i;
            bias_w[i] = row < a.n ? -(a.norms[row] >> 1) : UT_NEG;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t tile_row = (uint32_t)(g_rows * RT * 32 + 4 * lk);
        i32x4 qr[R];
#pragma unroll
        for (int j = 0; j < R - 1; ++j) qr[j] = *reinterpret_cast<const i32x4 *>(qp + j * 1024);
        auto block = [&](const int b, int &mx) {
            i32x16 acc[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // element 4 j + i: row i + 8 j + 4 lk of the tile
                    const int4 v = *reinterpret_cast<const int4 *>(bias_w + r * 32 + 8 * j + 4 * lk);
                    acc[r][4 * j] = v.x; acc[r][4 * j + 1] = v.y; acc[r][4 * j + 2] = v.z; acc[r][4 * j + 3] = v.w;
                }
            const uint8_t *qb = qp + (size_t)b * (KS * 1024);
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                // (the last R - 1 steps of the last block read past the queries, into the thresholds and the padding behind them: never used)
                qr[(s_ + R - 1) % R] = *reinterpret_cast<const i32x4 *>(qb + (s_ + R - 1) * 1024);
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[r][s_], qr[s_ % R], acc[r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // (inline-assembly readers of matrix results: the wait states are spelled out, see flat_f32_tfilter.hip)
            static_assert(RT >= 2 && RT <= 4, "operand lists below");
            if constexpr (RT == 2) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
            if constexpr (RT == 3) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
            if constexpr (RT == 4) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            const int qq = 32 * b + lj;
            if (a.dbg & 1) { mx = acc[0][0] + acc[RT - 1][3] > mx ? acc[0][0] : mx; return; }
            if constexpr (MAXMODE) {
                int m = ut_max16(acc[0]);
#pragma unroll
                for (int r = 1; r < RT; ++r) { const int m2 = ut_max16(acc[r]); m = m2 > m ? m2 : m; }
                mx = m > mx ? m : mx;
            } else {
                const int tb = thr_s[qq];
                int tmax[RT];   // (all of the block's tiles first: their chains interleave; the tests and the rare hit path after)
#pragma unroll
                for (int r = 0; r < RT; ++r) tmax[r] = ut_max16(acc[r]);
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const bool hit = tmax[r] >= tb;
                    const unsigned long long hm = __ballot(hit);
                    if (hm) {
                        const uint32_t pos = wcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                        wcnt += (uint32_t)__popcll(hm);
                        if (hit && pos < a.cap) {
                            uint8_t *dst = rec_w + pos * 80u;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                *reinterpret_cast<uint4 *>(dst + 16 * j) = make_uint4((uint32_t)acc[r][4 * j], (uint32_t)acc[r][4 * j + 1], (uint32_t)acc[r][4 * j + 2], (uint32_t)acc[r][4 * j + 3]);
                            *reinterpret_cast<uint4 *>(dst + 64) = make_uint4((uint32_t)(q0 + qq), tile_row + (uint32_t)(32 * r), (uint32_t)qq, 0u);
                        }
                    }
                }
            }
        };
        if constexpr (MAXMODE) {
#pragma unroll
            for (int b = 0; b < UT_NBQ; ++b)
                if (b < nb) block(b, mreg[b]);
        } else {
#pragma unroll 1
            for (int b = 0; b < nb; ++b) block(b, none);
        }
    }
    if constexpr (MAXMODE) {
#pragma unroll
        for (int b = 0; b < UT_NBQ; ++b) {   // (a slot nobody's rows reached stays 0 = empty: the caller zeroed the array)
            const int qq = 32 * b + lj;
            if (b < nb && qq < nqc && mreg[b] > UT_NEG / 2) a.smax[(size_t)(q0 + qq) * UT_SLOTS + slot] = (uint32_t)mreg[b] ^ 0x80000000u;
        }
    } else {
        if (lane == 0) a.wcnt[wave_g] = wcnt + (none == 0x12345678 ? 1u : 0u);
    }
}

// one wave per query: |q'|^2 and the threshold A = the k-th largest sample maximum (INT_MAX + the call's flag when fewer than k slots hold a row)
template <int NK>   // NK x 64 maxima per query: the 4096 slots, or (NK = 16) the maxima of four neighbours each -- row sets stay disjoint
__global__ __launch_bounds__(256) void ut_theta_kernel(const uint32_t *__restrict__ smax, const uint8_t *__restrict__ Q, int nq, int D, int k,
                                                       int32_t *__restrict__ thr, int32_t *__restrict__ qq_out, uint32_t *__restrict__ flag, uint32_t *__restrict__ cnt,
                                                       uint32_t *__restrict__ call_flag)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    uint32_t key[NK];
    if constexpr (NK == UT_SLOTS / 64) {
#pragma unroll
        for (int j = 0; j < NK; ++j) key[j] = smax[(size_t)q * UT_SLOTS + j * 64 + lane];
    } else {
        static_assert(NK * 256 == UT_SLOTS, "four slots per key");
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const uint4 v = *reinterpret_cast<const uint4 *>(smax + (size_t)q * UT_SLOTS + (size_t)(j * 64 + lane) * 4);
            const uint32_t m0 = v.x > v.y ? v.x : v.y, m1 = v.z > v.w ? v.z : v.w;
            key[j] = m0 > m1 ? m0 : m1;
        }
    }
    int s_ = 0;
    for (int e = lane; e < D; e += 64) { const int v = (int)Q[(int64_t)q * D + e] - 128; s_ += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s_ += __shfl_xor(s_, o, 64);
    const uint32_t sel = fs_wave_select(key, k, k + k / 4 + 8);
    if (lane == 0) {
        thr[q] = sel != 0u ? (int32_t)(sel ^ 0x80000000u) : 0x7fffffff;
        qq_out[q] = s_;
        flag[q] = sel == 0u ? 1u : 0u;   // (this query's own word; its list counter starts here too: two memsets less per pass)
        cnt[q] = 0u;
        if (q == 0) *call_flag = 0u;   // (the word this pass's bucket kernel raises when a wave's region ran over: the pass's finish flags every query then)
    }
}

// records -> per-query (distance, row) lists: as ft_bucket_kernel (flat_f32_tfilter.hip), the distance made exact on the way
__global__ __launch_bounds__(1024) void ut_bucket_kernel(const uint4 *__restrict__ rec, const uint32_t *__restrict__ wcnt, uint32_t cap,
                                                         const int32_t *__restrict__ thr, const int32_t *__restrict__ qqv, const int32_t *__restrict__ norms,
                                                         uint32_t *__restrict__ cnt, uint2 *__restrict__ cand, int chunks, int qper, int nq, uint32_t *__restrict__ flag)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t ut_stage[];
    __shared__ uint32_t hist[32 * UT_NBMAX], base_s[32 * UT_NBMAX];
    __shared__ uint32_t nstage_s;
    uint2 *st_dr = reinterpret_cast<uint2 *>(ut_stage);
    uint16_t *st_q = reinterpret_cast<uint16_t *>(ut_stage + (size_t)UT_STAGE * sizeof(uint2));
    const int j = blockIdx.x, tid = threadIdx.x;
    const int chunk = (j >> 3) % chunks, q0 = chunk * qper;
    const int nqc = nq - q0 < qper ? nq - q0 : qper;
    if (nqc <= 0) return;
    for (int i = tid; i < nqc; i += 1024) hist[i] = 0u;
    if (tid == 0) nstage_s = 0u;
    const int per = 1024 / UT_WAVES, r = tid / per, t = tid % per;
    uint32_t n = wcnt[j * UT_WAVES + r];
    if (n > cap) { if (t == 0) atomicOr(flag, 2u); n = cap; }   // (the call's word: whose records were lost is not known)
    const uint4 *rp = rec + (size_t)(j * UT_WAVES + r) * cap * 5;
    __syncthreads();
    auto each_hit = [&](auto &&f) {
        for (uint32_t i = t; i < n; i += per) {
            const uint4 h = rp[(size_t)i * 5 + 4];
            const int tb = thr[h.x], qq = qqv[h.x];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint4 s4 = rp[(size_t)i * 5 + jj];
                const int sv[4] = { (int)s4.x, (int)s4.y, (int)s4.z, (int)s4.w };
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (sv[e] >= tb) {
                        const uint32_t row = h.y + (uint32_t)(e + 8 * jj);
                        f(h.x, h.z, (uint32_t)(qq - 2 * sv[e] + (norms[row] & 1)), row);
                    }
            }
        }
    };
    each_hit([&](uint32_t, uint32_t ql, uint32_t d, uint32_t row) {
        atomicAdd(&hist[ql], 1u);
        const uint32_t pos = atomicAdd(&nstage_s, 1u);
        if (pos < (uint32_t)UT_STAGE) { st_dr[pos] = make_uint2(d, row); st_q[pos] = (uint16_t)ql; }
    });
    __syncthreads();
    const uint32_t nstage = nstage_s;
    for (int i = tid; i < nqc; i += 1024) {
        const uint32_t c = hist[i];
        base_s[i] = c ? atomicAdd(&cnt[q0 + i], c) : 0u;
        hist[i] = 0u;
    }
    __syncthreads();
    if (nstage <= (uint32_t)UT_STAGE) {
        for (uint32_t i = tid; i < nstage; i += 1024) {
            const uint32_t ql = st_q[i];
            const uint32_t pos = base_s[ql] + atomicAdd(&hist[ql], 1u);
            if (pos < (uint32_t)UT_CAP) cand[(size_t)(q0 + ql) * UT_CAP + pos] = st_dr[i];
        }
        return;
    }
    each_hit([&](uint32_t q, uint32_t ql, uint32_t d, uint32_t row) {   // more hits than the stage holds: the records once more
        const uint32_t pos = base_s[ql] + atomicAdd(&hist[ql], 1u);
        if (pos < (uint32_t)UT_CAP) cand[(size_t)q * UT_CAP + pos] = make_uint2(d, row);
    });
}

__device__ __forceinline__ void ut_bitonic_u64(unsigned long long *e, int np)
{
    for (int size = 2; size <= np; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < np / 2; i += 1024) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a0 = e[lo], a1 = e[hi];
                if ((a0 > a1) == up) { e[lo] = a1; e[hi] = a0; }
            }
        }
    }
    __syncthreads();
}
// one workgroup of 1024 threads per query: the k-th smallest distance by a radix select over the list in LDS, everything at or below it
// sorted by (distance, row)
__global__ __launch_bounds__(1024) void ut_finish_kernel(int64_t n, int k, const uint32_t *__restrict__ cnt, const uint2 *__restrict__ cand,
                                                         float *__restrict__ out_d, int64_t *__restrict__ out_i, uint32_t *__restrict__ flag, const uint32_t *__restrict__ pass_flag)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ub_keys[];   // [UT_CAP] keys = ~distance; later [UT_KEEP] (distance, row) pairs
    __shared__ uint32_t hist[256];
    __shared__ uint32_t pick_s[2];
    __shared__ int m2_s;
    constexpr int NTH = 1024;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const uint32_t nc = cnt[q];
    const int64_t want = k < n ? k : n;
    if (*pass_flag != 0u || nc > (uint32_t)UT_CAP || (int64_t)nc < want) {   // (pass_flag: a wave's record region ran over in this pass -- whose hits were lost is not known)
        if (tid == 0) flag[q] = 1u;
        return;
    }
    if (tid == 0) m2_s = 0;
    const uint2 *cq = cand + (size_t)q * UT_CAP;
    uint32_t kmx = 0u, kmn = 0xffffffffu;
    for (uint32_t i = tid; i < nc; i += NTH) {
        const uint32_t key = ~cq[i].x;   // (distances stay far below 2^32 - 1: keys are never 0)
        ub_keys[i] = key;
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
            const uint32_t key = ub_keys[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
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
    // prefix = ~(the k-th smallest distance): everything at or below that distance is kept
    __shared__ uint32_t keep_i[UT_KEEP];
    for (uint32_t i = tid; i < nc; i += NTH) {
        if (ub_keys[i] >= prefix) {
            const int pos = atomicAdd(&m2_s, 1);
            if (pos < UT_KEEP) keep_i[pos] = i;
        }
    }
    __syncthreads();
    const int m2 = m2_s;
    if (m2 > UT_KEEP) {   // masses of rows at the k-th distance
        if (tid == 0) flag[q] = 1u;
        return;
    }
    int np2 = 256;
    while (np2 < m2) np2 <<= 1;
    unsigned long long *sel = reinterpret_cast<unsigned long long *>(ub_keys);
    unsigned long long mine[UT_KEEP / NTH];
#pragma unroll
    for (int j = 0; j < UT_KEEP / NTH; ++j) {
        const int i = tid + j * NTH;
        mine[j] = ~0ull;
        if (i < m2) { const uint2 c = cq[keep_i[i]]; mine[j] = ((unsigned long long)c.x << 32) | c.y; }
    }
    __syncthreads();   // (the keys have been read by everybody: their space takes the pairs)
#pragma unroll
    for (int j = 0; j < UT_KEEP / NTH; ++j)
        if (tid + j * NTH < np2) sel[tid + j * NTH] = mine[j];
    ut_bitonic_u64(sel, np2);
    for (int i = tid; i < k; i += NTH) {
        const unsigned long long e = i < np2 ? sel[i] : ~0ull;
        if (e != ~0ull && i < want) {
            out_d[(int64_t)q * k + i] = __uint_as_float((uint32_t)(e >> 32));   // int32 distance bits
            out_i[(int64_t)q * k + i] = (int64_t)(uint32_t)e;
        } else {
            out_d[(int64_t)q * k + i] = __uint_as_float(0x7f800000u);
            out_i[(int64_t)q * k + i] = -1;
        }
    }
}

template <int KS, int RT>
static int ut_launch(bool maxmode, const UtArgs &a, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attr_a[16] = {}, attr_b[16] = {};
    if (maxmode) {
        CVTMI_TRY(fs_set_lds((const void *)flat_u8_tfilter_kernel<KS, RT, true>, 163840, attr_a));
        hipLaunchKernelGGL((flat_u8_tfilter_kernel<KS, RT, true>), dim3(UT_GRID), dim3(64 * UT_WAVES), lds, st, a);
    } else {
        CVTMI_TRY(fs_set_lds((const void *)flat_u8_tfilter_kernel<KS, RT, false>, 163840, attr_b));
        hipLaunchKernelGGL((flat_u8_tfilter_kernel<KS, RT, false>), dim3(UT_GRID), dim3(64 * UT_WAVES), lds, st, a);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace

static std::atomic<int> g_ut_on{1};   // cvtmi_set_tuning("flat_u8_tfilter"): 1 = uint8 searches with k = 129 .. 2048 take this pipeline, 0 = the exact kernels
// "flat_u8_tfilter_min_k" / "_min_nq": k <= 128 from this k and this batch on.  Measured (tools/flat_u8_bigk.py, profiles/r06_flat_u8_tfilter.txt; 1 M x 256-d,
// 2 M x 512-d, 10 M x 128-d, 10 M x 512-d; k = 10 / 64 / 100 / 128): from 128 queries on this pipeline is level with or ahead of both the
// streaming passes and the sample + filter pipeline of flat_mfma.hip at every point (k = 10: 0.9-1.0 x at 128 queries, 0.75-0.95 x from 256;
// k = 100: 0.7 x at 128 queries, 0.42-0.55 x from 256); below, one streaming pass over the raw rows wins
static std::atomic<int> g_ut_min_k{1}, g_ut_min_nq{129}, g_ut_min_nq_k65{97};   // ("_min_nq_k65": the batch bound for k = 65 .. 128, where a streaming pass costs more)
static std::atomic<int64_t> g_ut_min_rows{262144};   // "flat_u8_tfilter_min_rows": tables under it (from 65 536 rows) come here from "flat_u8_tfilter_small_min_nq" queries on
static std::atomic<int> g_ut_small_min_nq{129};   // (tools/flat_u8_small_tables.py, profiles/r06_flat_u8_small_tables.txt: ahead of the streaming passes from 129 queries on at every size, 2-6 x at 1000)
static std::atomic<int> g_ut_sample{0};   // "flat_u8_tfilter_sample": the sample pass takes one tile group in this many (0: by k -- 8 up to k = 512, 5 up to 1024, 3 beyond)
void set_flat_u8_tfilter(int v) { g_ut_on = v != 0; }
void set_flat_u8_tfilter_min_rows(int64_t v) { g_ut_min_rows = v < 65536 ? 65536 : v; }
void set_flat_u8_tfilter_small_min_nq(int v) { g_ut_small_min_nq = v < 1 ? 1 : v; }
void set_flat_u8_tfilter_min_k(int v) { g_ut_min_k = v < 1 ? 1 : v; }
void set_flat_u8_tfilter_min_nq(int v) { g_ut_min_nq = v < 1 ? 1 : v; }
void set_flat_u8_tfilter_min_nq_k65(int v) { g_ut_min_nq_k65 = v < 1 ? 1 : v; }
void set_flat_u8_tfilter_sample(int v) { g_ut_sample = v < 0 ? 0 : v > 64 ? 64 : v; }
// Workgroups that hold different queries walk the same rows in the same order on the same XCD ("chunks" of a pass): the rows come from
// HBM once and from the caches behind it for the others -- as many chunks as the sample's slots allow (4096 / chunks of them are filled;
// twice k wanted), "flat_u8_tfilter_chunks" caps it
static std::atomic<int> g_ut_chunks{4};
void set_flat_u8_tfilter_chunks(int v) { g_ut_chunks = v >= 4 ? 4 : (v >= 2 ? 2 : 1); }
static int ut_chunks_max(int k) { return std::min(g_ut_chunks.load(), k <= 512 ? 4 : (k <= 1024 ? 2 : 1)); }
static int ut_rt(int ks) { return ks >= 12 ? 2 : (ks >= 4 ? 3 : 4); }   // row tiles per wave: RT x 4 KS + 16 RT registers of 256 (four at 128-d: 5 % faster on large
                                                                          // tables, but a third fewer tile groups -- sample slots -- on small ones: three)
bool flat_u8_tfilter_width(int D) { return D % 32 == 0 && D >= 32 && D <= 512; }   // (a kernel per K-step count: 32 .. 512 bytes per row in steps of 32)
bool flat_u8_tfilter_applies(int D, int64_t n, int64_t nq, int k)
{
    if (!g_ut_on.load() || !flat_u8_tfilter_width(D) || n >= 0x7fffffe0LL || nq < 1 || k > CVTMI_K_MAX) return false;
    if (k > 128) {   // smaller tables too, while the sample can fill 1.25 k slots (two per wave that gets a tile group): nothing else is fast there
        const int ks = D / 32, rt = ut_rt(ks);
        const int64_t groups = (n + 32 * rt - 1) / (32 * rt);
        return n >= 65536 && 8 * std::min<int64_t>(groups, UT_GRID * UT_WAVES / ut_chunks_max(k)) >= 5 * (int64_t)k;
    }
    // widths the streaming kernel does not take (it exists at 128 / 256 / 512-d): the row-tile kernels behind cost 0.1 .. 4 ms whatever the batch -- every
    // batch of two queries or more comes here (1 M x 96-d, 8 queries: 0.43 -> 0.08 ms)
    const bool has_stream = D == 128 || D == 256 || D == 512;
    if (!has_stream && n >= 65536 && nq >= 2 && k >= g_ut_min_k.load()) {
        const int ks = D / 32, rt = ut_rt(ks);
        const int64_t groups = (n + 32 * rt - 1) / (32 * rt);
        if (8 * std::min<int64_t>(groups, UT_GRID * UT_WAVES / 4) >= 5 * (int64_t)k) return true;
    }
    if (n < g_ut_min_rows.load()) {   // small tables, k <= 128: large batches only (the stream's passes are short there), and while the sample fills its slots
        const int ks = D / 32, rt = ut_rt(ks);
        const int64_t groups = (n + 32 * rt - 1) / (32 * rt);
        if (n < 65536 || nq < g_ut_small_min_nq.load() || k < g_ut_min_k.load() || 8 * std::min<int64_t>(groups, UT_GRID * UT_WAVES / 4) < 5 * (int64_t)k) return false;
        return true;
    }
    // (the lower batch bound also for tables of a GB and more at any k: 10 M x 128-d, 96 queries, k = 10 0.43 -> 0.31 ms, 10 M x 512-d 1.15 -> 0.97; at 1 M x 256-d
    //  the streaming pass is still ahead there, 0.113 against 0.123)
    const bool low = k > 64 || (double)n * D >= 1e9;
    return (k >= g_ut_min_k.load() && nq >= (low ? std::min(g_ut_min_nq.load(), g_ut_min_nq_k65.load()) : g_ut_min_nq.load()));
}
// the sample pass takes one tile group in so many: about k x div rows pass the threshold (~0.7 x 4096 x div at k = 2048)
// Measured (tools/flat_u8_sample_sweep.py, profiles/r06_flat_u8_sample_sweep.txt: 0.26 .. 5 GB of rows, k = 10 .. 1024): the sample pass costs
// bytes / div, the candidates' way through the bucket and finish kernels k x div -- the best div follows sqrt(8000 x GB / k) within a few per
// cent everywhere; beyond ~4096 / k the lists and the waves' record regions run over (the call falls back)
static int ut_sample_div(int k, int64_t n, int D)
{
    if (g_ut_sample.load()) return g_ut_sample.load();
    const double want = std::sqrt(8000.0 * ((double)n * D * 1e-9) / (double)k);
    const int hi = std::min(32, std::max(3, 4096 / k));
    return std::max(2, std::min(hi, (int)(want + 0.5)));
}
// records per wave: three times the expected count (rows reach the waves tile group by tile group, evenly)
static uint32_t ut_rec_cap(int64_t m, int k, int div)
{
    const double pass = std::min<double>((double)k, 0.7 * UT_SLOTS) * div * 1.5;
    return (uint32_t)std::min<double>(8192.0, std::max<double>(512.0, 3.0 * (double)m * pass / (UT_GRID * UT_WAVES)));
}
size_t flat_u8_tfilter_scratch(int D, int64_t n, int64_t nq, int k)
{
    const int64_t m = std::min<int64_t>(nq, (int64_t)UT_QPER * ut_chunks_max(k));
    return (size_t)m * (UT_SLOTS + 4) * sizeof(uint32_t) + (size_t)m * UT_CAP * sizeof(uint2) + (size_t)UT_GRID * UT_WAVES * (sizeof(uint32_t) + (size_t)ut_rec_cap(m, k, ut_sample_div(k, n, D)) * 80) + 1024;
}

// nq queries against rows [0, n); flags[nq + 1] (device, zeroed here): flags[1 + q] != 0 afterwards = query q could not be answered (flags[0]: a wave's record
// region ran over in the last pass -- every query of such a pass is flagged) -- the caller re-runs those under flags + 1 as a predicate
int launch_flat_u8_tfilter(int D, const void *pack, const int32_t *norms, int64_t n, const uint8_t *q, int64_t nq, int k, void *scratch, float *out_d,
                           int64_t *out_i, uint32_t *flags, hipStream_t st)
{
    if (!flat_u8_tfilter_applies(D, n, nq, k)) return fail(CVTMI_EINVAL, "flat_u8_tfilter: D=%d nq=%lld k=%d", D, (long long)nq, k);
    const int ks = D / 32, rt = ut_rt(ks);   // (tiles per wave: RT x (4 KS + 32) registers of 256)
    const int qcap = std::min(32 * UT_NBMAX, (int)((size_t)(160 * 1024 - 32 * UT_NBMAX * 4 - UT_WAVES * 4 * 32 * 4 - 3072) / ((size_t)ks * 1024)) * 32);
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t pass = (int64_t)UT_QPER * ut_chunks_max(k);
    if (qcap < UT_QPER) return fail(CVTMI_EINVAL, "flat_u8_tfilter: %d queries do not fit the LDS at D=%d", UT_QPER, D);
    for (int64_t a0 = 0; a0 < nq; a0 += pass) {
        const int64_t m = std::min<int64_t>(nq - a0, pass);
        const int chunks = m <= UT_QPER ? 1 : (m <= 2 * UT_QPER ? 2 : 4);
        const int qper = (int)(((m + chunks - 1) / chunks + 31) / 32 * 32);
        const int div = ut_sample_div(k, n, D);
        // (CVTMI_UT_DBG 4, test hook: regions of two records -- every wave's runs over, the pass flags all of its queries, the predicated kernels answer)
        const char *dbg_env = getenv("CVTMI_UT_DBG");
        const uint32_t cap = (dbg_env && (atoi(dbg_env) & 4)) ? 2u : ut_rec_cap(m, k, div);
        uint32_t *smax = reinterpret_cast<uint32_t *>(scratch);
        int32_t *thr = reinterpret_cast<int32_t *>(smax + (size_t)m * UT_SLOTS);
        int32_t *qqv = thr + m;
        uint32_t *cnt = reinterpret_cast<uint32_t *>(qqv + m);
        uint32_t *wcnt = cnt + m + (m & 1);
        uint2 *cand = reinterpret_cast<uint2 *>(wcnt + UT_GRID * UT_WAVES);
        uint4 *rec = reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(cand + (size_t)m * UT_CAP) + 256 - (((uintptr_t)(cand + (size_t)m * UT_CAP)) & 15));
        CVTMI_HIP(hipMemsetAsync(smax, 0, (size_t)m * UT_SLOTS * sizeof(uint32_t), st));
        UtArgs a;
        a.pack = reinterpret_cast<const uint4 *>(pack); a.norms = norms; a.n = n; a.n_tiles = n_tiles; a.Q = q + a0 * D; a.D = D; a.nq = (int)m;
        { const char *e = getenv("CVTMI_UT_DBG"); a.dbg = e ? atoi(e) : 0; }
        a.chunks = chunks; a.qper = qper; a.smax = smax; a.thr = thr; a.rec = rec; a.wcnt = wcnt; a.cap = cap;
        {   // the sample: a whole number of tile groups per wave of a chunk
            const int64_t all_groups = (n_tiles + rt - 1) / rt, streams = (int64_t)(UT_GRID / chunks) * UT_WAVES;
            const int64_t want = std::max<int64_t>(all_groups / div, std::min<int64_t>(all_groups, 2048 / rt));
            a.n_sample = std::min<int64_t>(all_groups, std::max<int64_t>(1, (want + streams - 1) / streams) * streams);   // (never a smaller sample than asked for)
        }
        const size_t lds = (size_t)(qper / 32) * ks * 1024 + (size_t)qper * sizeof(int) + (size_t)UT_WAVES * rt * 32 * sizeof(int) + 3072;
#define CVTMI_UT(MAXM) \
        (ks == 16 ? ut_launch<16, 2>(MAXM, a, lds, st) : ks == 15 ? ut_launch<15, 2>(MAXM, a, lds, st) : ks == 14 ? ut_launch<14, 2>(MAXM, a, lds, st) : \
         ks == 13 ? ut_launch<13, 2>(MAXM, a, lds, st) : ks == 12 ? ut_launch<12, 2>(MAXM, a, lds, st) : ks == 11 ? ut_launch<11, 3>(MAXM, a, lds, st) : \
         ks == 10 ? ut_launch<10, 3>(MAXM, a, lds, st) : ks == 9 ? ut_launch<9, 3>(MAXM, a, lds, st) : ks == 8 ? ut_launch<8, 3>(MAXM, a, lds, st) : \
         ks == 7 ? ut_launch<7, 3>(MAXM, a, lds, st) : ks == 6 ? ut_launch<6, 3>(MAXM, a, lds, st) : ks == 5 ? ut_launch<5, 3>(MAXM, a, lds, st) : \
         ks == 4 ? ut_launch<4, 3>(MAXM, a, lds, st) : ks == 3 ? ut_launch<3, 4>(MAXM, a, lds, st) : ks == 2 ? ut_launch<2, 4>(MAXM, a, lds, st) : ut_launch<1, 4>(MAXM, a, lds, st))
        CVTMI_TRY(CVTMI_UT(true));
        // (merged keys while that leaves eight per neighbour wanted)
        if (k * 8 <= UT_SLOTS / 4 / chunks) hipLaunchKernelGGL(ut_theta_kernel<UT_SLOTS / 256>, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, smax, a.Q, (int)m, D, k, thr, qqv, flags + 1 + a0, cnt, flags);
        else hipLaunchKernelGGL(ut_theta_kernel<UT_SLOTS / 64>, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, smax, a.Q, (int)m, D, k, thr, qqv, flags + 1 + a0, cnt, flags);
        CVTMI_TRY(CVTMI_UT(false));
#undef CVTMI_UT
        const size_t bucket_lds = (size_t)UT_STAGE * (sizeof(uint2) + sizeof(uint16_t)), fin_lds = (size_t)UT_CAP * sizeof(uint32_t);
        static std::atomic<bool> attr_k[16] = {}, attr_f[16] = {};
        CVTMI_TRY(fs_set_lds((const void *)ut_bucket_kernel, bucket_lds, attr_k));
        CVTMI_TRY(fs_set_lds((const void *)ut_finish_kernel, fin_lds, attr_f));
        hipLaunchKernelGGL(ut_bucket_kernel, dim3(UT_GRID), dim3(1024), bucket_lds, st, rec, wcnt, cap, thr, qqv, norms, cnt, cand, chunks, qper, (int)m, flags);
        hipLaunchKernelGGL(ut_finish_kernel, dim3((unsigned)m), dim3(1024), fin_lds, st, n, k, cnt, cand, out_d + a0 * k, out_i + a0 * k, flags + 1 + a0, flags);
        CVTMI_HIP(hipGetLastError());
    }
    return CVTMI_OK;
}

}  // namespace cvtmi
