// adc_scan16.h -- pieces shared by the M = 16 skewed scan kernels (adc_scan.hip: adc_scan16q / adc_scan16a,
// adc_scan_h.hip: adc_scan16h): kernel arguments, the 15-bit lower-bound tables' parameters, the conflict-free look-up
// loop of one row, the exact re-sum of candidates and the lazy (integer-key) selection.  See adc_scan.hip for the
// reference arithmetic (opq/src/IVFOPQ.cpp:273-306) and the derivations.
#pragma once
#include "block_topk.h"
#include "kernels.h"

namespace cvtmi {

struct ScanArgs {
    const uint8_t *codes;
    int64_t n_rows;
    int64_t id_base;
    const float *q_rot;
    int nq;
    const float *books;
    const float *centroid;  // coarse[0]
    int D, step, K, k;
    int splits;
    int64_t rows_per_split;
    int groups;
    int groups_a, splits_b, stride;  // two-region plan (kernels.h), adc_scan16q only; stride = partial slots per query
    int64_t rows_per_split_b;
    float *part_d;
    int64_t *part_id;
    float *out_d;      // adc_scan16q, two-region plan whose first region has ONE row split: that region's groups write their (final) lists
    int64_t *out_id;   // straight to [nq][k] here and the merge only visits the other queries (scan_in_place_queries); null: everything to part_*
    const float *lut_g;  // [nq][M][256] fp32 tables in HBM, +inf past K (adc_scan16q only)
    const uint8_t *codes_rot;  // adc_scan16q: copy of the code rows with row r rotated left by r & 15 bytes (or null)
    uint32_t *gthr;            // adc_scan16q: [nq] filter thresholds (table units) shared by the row splits of a query, or null
    int lazy;                  // adc_scan16q: 1 = intermediate compactions select on the integer lower bounds (exact sums only at the end)
    int seed;                  // adc_scan16q / 16a: 1 = first thresholds from a histogram of the split's first rows (scan16q_seed)
    const uint32_t *only = nullptr;  // adc_scan_kernel with one query per workgroup (k > 128): answer query g only when only[g] != 0 (the
                                     // big-k filter pipeline's fall-back, adc_scan_h.hip); null: every query
};

#ifdef CVTMI_SCAN_TIMING
static __device__ unsigned long long g_scan_dbg[8];   // (one copy per translation unit: each has its own debug accessor)
static __device__ unsigned long long g_scan_dbg2[8];  // adc_scan16a slow path, lane 0 of every wave: calls, cycles, compactions, compaction cycles, cycles waiting for stores, retry rounds
static __device__ unsigned long long g_scan_trace[4 * 16384];  // per workgroup: wall start, wall end (100 MHz), shader clocks, HW_ID | XCC_ID << 32
#define SQ_T(i) do { if (threadIdx.x == 0) { const unsigned long long now__ = clock64(); t_acc__[i] += now__ - t_last__; t_last__ = now__; } } while (0)
#define SQ_T0() unsigned long long t_last__ = clock64(); unsigned long long t_acc__[5] = { 0, 0, 0, 0, 0 }; \
    const unsigned long long t_first__ = t_last__, w_first__ = wall_clock64()
#define SQ_TEND() do { if (threadIdx.x == 0) { for (int i__ = 0; i__ < 5; ++i__) atomicAdd(&g_scan_dbg[i__], t_acc__[i__]); \
    if (blockIdx.x < 16384) { unsigned long long *tr__ = g_scan_trace + 4 * blockIdx.x; tr__[0] = w_first__; tr__[1] = wall_clock64(); \
        tr__[2] = clock64() - t_first__; \
        tr__[3] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); } } } while (0)
#define SQA_ADD(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&ck.dbg[i], (unsigned long long)(v)); } while (0)  // LDS; flushed once per workgroup
#define SQA_SET(i, v) do { if ((threadIdx.x & 63) == 0) ck.dbg[i] += (unsigned long long)(v); } while (0)
#define SQA_NOW() clock64()
#else
#define SQA_SET(i, v) do { } while (0)
#define SQA_ADD(i, v) do { } while (0)
#define SQA_NOW() 0ll
#define SQ_T(i) do { } while (0)
#define SQ_T0() do { } while (0)
#define SQ_TEND() do { } while (0)
#endif
constexpr int SQ_QT = 8;
constexpr int SQ_CAP = 238;   // two workgroups per CU: (81920 - 65536 table - ~1 KB control) / 8 queries / 8 bytes
constexpr int SQ_TRIG = 192;
constexpr int SQ_MAXSUM = 32766;

struct QuantParams {
    union {
        float mn[SQ_QT][16];          // per (query, sub-quantiser) minimum finite table entry
        uint32_t mn_bits[SQ_QT][16];  // same words while the minimum is being reduced (non-negative floats)
    };
    float inv_scale[SQ_QT];
    double scale_eff[SQ_QT];  // 1 / inv_scale, the scale the integers are really in
    double bias[SQ_QT];       // sum_m mn[m]
    uint32_t slack[SQ_QT];    // lazy selection: a row whose integer sum is >= (k-th smallest integer sum) + slack is out (0 = not usable)
};

struct QuantThr {
    const QuantParams *qp;
    __device__ __forceinline__ uint32_t operator()(int q, uint32_t t) const
    {
        if (t >= 0x7f800000u) return 32767u;  // no finite threshold yet: every sum (<= 32766) passes
        const double x = ((double)__uint_as_float(t) * (1.0 + 1e-6) - qp->bias[q]) / qp->scale_eff[q];
        if (!(x > 0.0)) return 2u;
        const double f = floor(x) + 2.0;
        return f > 32767.0 ? 32767u : (uint32_t)f;
    }
};

// exact reference-order distance of (row, query): sum over m ascending of the fp32 table entries
// (IVFOPQ.cpp:302-306) gathered from the per-query tables in HBM -- 16 independent loads.
struct ExactFromLut {
    static constexpr bool enabled = true;
    const uint4 *rows;
    const float *lut_g;
    int K, nq, group;
    __device__ __forceinline__ unsigned long long operator()(int q, unsigned long long e) const
    {
        const uint32_t row = (uint32_t)e;
        const uint4 c = rows[row];
        const uint32_t w[4] = { c.x, c.y, c.z, c.w };
        int qi = group * SQ_QT + q;
        qi = qi < nq ? qi : nq - 1;
        const float *t = lut_g + (int64_t)qi * 16 * 256;
        float v[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int j = (int)((w[m >> 2] >> (8 * (m & 3))) & 0xffu);
            v[m] = t[m * 256 + j];  // padded with +inf past K
        }
        float s = 0.0f;
#pragma unroll
        for (int m = 0; m < 16; ++m) s = __fadd_rn(s, v[m]);
        return ((unsigned long long)__float_as_uint(s) << 32) | row;
    }
};

// batched form for the register compaction: up to 4 entries per lane, all row loads first, then all
// 64 table gathers, then the in-order sums -- two dependent memory round trips per batch instead of 2 x 4
struct ExactFromLutBatch {
    const uint4 *rows;
    const float *lut_g;
    int K, nq, group;
    __device__ __forceinline__ void operator()(int q, unsigned long long (&e)[4], const bool (&need)[4]) const
    {
        int qi = group * SQ_QT + q;
        qi = qi < nq ? qi : nq - 1;
        const float *t = lut_g + (int64_t)qi * 16 * 256;
        uint4 c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = rows[need[r] ? (uint32_t)e[r] : 0u];
#pragma unroll
        for (int h = 0; h < 4; h += 2) {  // two entries at a time: 32 gathers in flight, 32 registers
            if (__ballot(need[h] || need[h + 1]) == 0) continue;  // wave-uniform: nothing to fix in this pair
            float v[2][16];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t w[4] = { c[h + r].x, c[h + r].y, c[h + r].z, c[h + r].w };
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const int j = (int)((w[m >> 2] >> (8 * (m & 3))) & 0xffu);
                    // (a predicated gather made hipcc wait vmcnt(0) after every one of the 64 loads:
                    //  64 serialized memory round trips, 14 us per compaction)
                    v[r][m] = t[m * 256 + j];  // scratch tables are padded to 256 entries (+inf past K): no predicate
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int m = 0; m < 16; ++m) s = __fadd_rn(s, v[r][m]);
                if (need[h + r]) e[h + r] = ((unsigned long long)__float_as_uint(s) << 32) | (uint32_t)e[h + r];
            }
        }
    }
};

typedef unsigned short cvt_us2 __attribute__((ext_vector_type(2)));
typedef short cvt_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (cvt_us2)(__builtin_bit_cast(cvt_us2, a) + __builtin_bit_cast(cvt_us2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (cvt_s2)(__builtin_bit_cast(cvt_s2, a) - __builtin_bit_cast(cvt_s2, b)));
}

// ---- lazy selection (intermediate compactions of adc_scan16q) -----------------------------------------------------------
// The integer sum S of a row is a two-sided bound of its real table sum D in table units u = (D - bias) / scale_eff:
//     S <= u < S + 32.01        (each entry: qv = max(0, floor(f) - 1), f = fl((v - mn) * inv) within 2^-22 of (v - mn) / scale_eff)
// and the reference's fp32 sum d_ref is within 15 rounding steps (9e-7 relative) of D.  Let S_k be the k-th smallest S seen
// so far.  A row with S >= S_k + slack, slack = 34 + ceil(4e-6 * (32767 + bias / scale_eff)), has
//     D >= bias + scale (S_k + slack)  >  (bias + scale (S_k + 32.01)) (1 + 2e-6)  >  D_i (1 + 2e-6)   for each of the k rows i with S_i <= S_k,
// hence d_ref > d_ref,i for k rows: it is not among the k smallest (distance, id) pairs, ties included.  So between checkpoints
// the buffer only needs the integer keys: keep every entry with S < T = S_k + slack (k plus the few rows within `slack` units
// of the k-th), push under the same T, and compute exact reference-order sums ONCE, in the final compaction (entries keep
// exact_n = 0, so the final pass re-sums all of them).  The two dependent memory round trips of the exact re-sum leave every
// intermediate compaction.  If the band [S_k, S_k + slack) is crowded (more than `keep_max` entries stay), the query switches to
// the exact-key protocol for the rest of the scan: its entries are re-summed now and from then on it is compacted by
// topk_compact_wave_q.  Queries whose tables hold non-finite entries never start lazy (slack = 0): S is no upper bound there.
template <int QT, int CAP, class FixB, class ThrX>
__device__ __attribute__((noinline)) int scan_compact_lazy_q(TopKShared<QT, CAP> &s, int q, int k, const FixB &fixb, const ThrX &thrx,
                                                            uint32_t slack, int keep_max, int *lazy_flag, int n_in = -1)
{
    static_assert(CAP <= 256, "register selection holds 256 entries per wave");
    constexpr int NR = CAP <= 64 ? 1 : (CAP <= 128 ? 2 : 4);
    const int lane = threadIdx.x & 63;
    unsigned long long *b = s.buf[q];
    int n = n_in >= 0 ? n_in : s.cnt[q];
    n = n < CAP ? n : CAP;
    unsigned long long e[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int p = r * 64 + lane;
        e[r] = p < n ? b[p] : ~0ull;
    }
    if (n < k) {  // fewer than k rows seen: everything stays where it is, every sum passes
        if (lane == 0) { s.exact_n[q] = 0; s.thr[q] = KEY_MAX; s.thr_x[q] = 32767u; }
        return n;
    }
    const uint32_t T = wave_select_field<NR, 15>(e, k) + slack;  // the k-th smallest integer sum (sums are below 2^15) + the band
    int total = 0;
    unsigned long long in_m[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        in_m[r] = __ballot((uint32_t)(e[r] >> 32) < T);  // empty slots carry 0xffffffff: never in
        total += __popcll(in_m[r]);
    }
    if (total <= keep_max) {  // wave-uniform
        int base = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool in = (in_m[r] >> lane) & 1ull;
            const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(in_m[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)in_m[r], 0u));
            if (in) b[pos] = e[r];
            base += __popcll(in_m[r]);
        }
        if (lane == 0) { s.exact_n[q] = 0; s.thr[q] = KEY_MAX; s.thr_x[q] = T < 32767u ? T : 32767u; }
        return total;
    }
    // crowded band: this query leaves the lazy protocol -- exact keys for everything it holds, then the ordinary selection
    bool need[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) need[r] = r * 64 + lane < n;
    fixb(q, e, need);
    const unsigned long long kx = wave_select<NR>(e, k);
    int base = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool in = e[r] <= kx;
        const unsigned long long m = __ballot(in);
        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (in && pos < k) b[pos] = e[r];
        base += __popcll(m);
    }
    if (lane == 0) {
        *lazy_flag = 0;
        s.exact_n[q] = k;
        s.thr[q] = (uint32_t)(kx >> 32);
        s.thr_x[q] = thrx(q, (uint32_t)(kx >> 32));
    }
    return k;
}

// the 16 look-ups of one row: packed 15-bit sums of the SQ_QT queries, two per word -- they can never carry across the 16-bit
// halves, so they are accumulated with plain 32-bit adds, two look-ups per v_add3_u32 (half the VALU of v_pk_add_u16).  Integer
// sums, any order.  NG look-ups are in flight at a time: 16 in adc_scan16q; adc_scan16a takes groups of four -- there the scheduler
// would otherwise form all 16 addresses first, more registers than its waves have left.
// BIASED: the sums start from s0 .. s3 as passed in instead of zero (adc_scan16q: 0x8000 - T per 16-bit field, so that "sum < T" is a
// CLEAR bit 15 and the test of a row is an AND over the four words instead of four packed subtractions; sum + 0x8000 - T < 2^16: still no carry)
template <bool PREROT, int NG = 4, bool BIASED = false>
__device__ __forceinline__ void scan16q_row_sums(const uint4 &row, const uint32_t (&moffp)[4], uint32_t cr8, uint32_t cq, const char *lut_b,
                                                 uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3)
{
    uint32_t d0 = row.x, d1 = row.y, d2 = row.z, d3 = row.w;
    if constexpr (!PREROT) {
        d0 = __builtin_amdgcn_alignbit(row.y, row.x, cr8);
        d1 = __builtin_amdgcn_alignbit(row.z, row.y, cr8);
        d2 = __builtin_amdgcn_alignbit(row.w, row.z, cr8);
        d3 = __builtin_amdgcn_alignbit(row.x, row.w, cr8);
        const bool b0 = cq & 1;
        const uint32_t e0 = b0 ? d1 : d0, e1 = b0 ? d2 : d1, e2 = b0 ? d3 : d2, e3 = b0 ? d0 : d3;
        const bool b1 = cq & 2;
        d0 = b1 ? e2 : e0; d1 = b1 ? e3 : e1; d2 = b1 ? e0 : e2; d3 = b1 ? e1 : e3;
    }
    const uint32_t rot[4] = { d0, d1, d2, d3 };
    if constexpr (!BIASED) { s0 = 0; s1 = 0; s2 = 0; s3 = 0; }
#pragma unroll
    for (int h = 0; h < 16 / NG; ++h) {
        uint4 v[NG];
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int tt = NG * h + t;
            const uint32_t sel = 0x0c0c0000u | ((4u + (tt & 3)) << 8) | (uint32_t)(tt & 3);
            const uint32_t addr = __builtin_amdgcn_perm(rot[tt >> 2], moffp[tt >> 2], sel);  // code*256 + m*16
            v[t] = *reinterpret_cast<const uint4 *>(lut_b + addr);
        }
#pragma unroll
        for (int t = 0; t < NG; t += 2) {
            s0 = s0 + v[t].x + v[t + 1].x; s1 = s1 + v[t].y + v[t + 1].y;
            s2 = s2 + v[t].z + v[t + 1].z; s3 = s3 + v[t].w + v[t + 1].w;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Seed: a first filter threshold per query from a histogram of the split's first SQ_SEED_CHUNKS x 64 rows.
// Without it a scan starts with "every row passes": a thousand rows in flight against SQ_CAP slots, and four to five rounds of
// the whole workgroup waiting for compactions before the pass rate has fallen.  bin = sum >> 7; the first bin b at which the
// cumulative count reaches k proves k rows with sum < (b + 1) << 7 =: S, hence S_k <= S, and T = S + slack is a valid (looser)
// lazy-selection threshold (scan_compact_lazy_q).  Queries that cannot select lazily keep "pass all".  The seed rows are
// scanned again by the main loop.  hist = SQ_QT x 256 words (the still unused selection buffers).  Whole workgroup; ends with a barrier.
constexpr uint32_t SQ_SEED_CHUNKS = 32;
template <int NT, bool PREROT, class LoadRow>
__device__ __forceinline__ void scan16q_seed(int k, const LoadRow &load64, const uint32_t (&moffp)[4], uint32_t cr8, uint32_t cq, const char *lut_b,
                                             uint32_t *hist, const QuantParams &qp, const int *lazy, uint32_t *thr_x, uint32_t *thr_pk)
{
    constexpr int QT = SQ_QT, NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < QT * 256; i += NT) hist[i] = 0;
    __syncthreads();
    for (uint32_t ch = wave; ch < SQ_SEED_CHUNKS; ch += NW) {
        const uint4 row = load64(ch);
        uint32_t sm[4];
        scan16q_row_sums<PREROT>(row, moffp, cr8, cq, lut_b, sm[0], sm[1], sm[2], sm[3]);
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const uint32_t sq = (sm[q >> 1] >> (16 * (q & 1))) & 0xffffu;
            atomicAdd(&hist[q * 256 + (sq >> 7)], 1u);
        }
    }
    __syncthreads();
    if (wave < QT && lazy[wave]) {
        const int q = wave;
        uint32_t c4[4], mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c4[j] = hist[q * 256 + lane * 4 + j]; mine += c4[j]; }
        uint32_t incl = mine;  // inclusive prefix over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const unsigned long long reach = __ballot(incl >= (uint32_t)k);
        if (reach) {  // wave-uniform
            const int l0 = __ffsll((long long)reach) - 1;
            if (lane == l0) {
                uint32_t cum = incl - mine;
                int b = lane * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cum += c4[j];
                    if (cum >= (uint32_t)k) { b = lane * 4 + j; break; }
                }
                uint32_t t = ((uint32_t)(b + 1) << 7) + qp.slack[q];
                t = t < 32767u ? t : 32767u;
                const uint32_t have = thr_x[q];  // what the other row splits have established already
                t = have < t ? have : t;
                thr_x[q] = t;
                reinterpret_cast<uint16_t *>(thr_pk)[q] = (uint16_t)t;
            }
        }
    }
    __syncthreads();
}

}  // namespace cvtmi
