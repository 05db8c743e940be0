// adc_scan_h.hip -- adc_scan16h: the M = 16 scan (15-bit lower-bound tables, 8 queries per pass, conflict-free skewed
// look-ups: adc_scan.hip) with the per-workgroup fixed costs taken out.
//
// Reference arithmetic (opq/src/IVFOPQ.cpp): tables :273-291, per-row sum :300-306 (fp32, m ascending), k smallest
// (score, id) common.h:25-37.  Results are bit-identical to it: the loop is a filter, survivors are re-summed in the
// reference's order from the fp32 tables (ExactFromLutBatch) before they are ranked.
//
// What adc_scan16q pays per workgroup besides its look-ups (profiles/r02_scan_phase_timing.txt: 250 of 1300 us at
// SIFT-1M, fitted at 0.38 M row-equivalents by its planner), and what happens to it here:
//   * checkpoint compactions (every buffer fill parks the 16 waves: 9-13 stops of ~13 us)
//       -> candidates go to a per-workgroup area in HBM (8-byte entries, 4096 per query) and are selected ONCE, when the
//          workgroup's rows are exhausted.  The filter threshold tightens without any selection: every candidate also counts
//          into a 256-bin histogram of its integer sum (bin = sum >> 7, one LDS atomic); the first bin at which the
//          cumulative count reaches k proves k rows with sum < edge, so T = edge + slack is a valid filter bound (the
//          lazy-selection argument of adc_scan16.h: a row with S >= S_k + slack is beaten by k rows in the reference's fp32
//          arithmetic, and S_k < edge).  Whoever stores a query's 32nd, 64th, ... candidate recomputes that query's bound
//          (one wave, ~1 us, nobody else waits) and bumps an epoch word the other waves look at once per 64 rows.
//   * table build (two passes over 8 x 16 KB of fp32 tables per workgroup)
//       -> scan16h_prep_kernel builds each query group's quantised tables once, in the LDS image the loop reads
//          ([code][m][8 x u16], 64 KB) next to the fp32 tables; a workgroup copies the image in.
//   * round packing (1250 workgroups on 512 slots: the last round runs at 44 % occupancy)
//       -> a persistent grid of (at most) two workgroups per CU walks a host-built item table.  When the code matrix is
//          cache-resident the flat (query group x row) space is cut into equal shares, one per workgroup, so every CU is
//          busy until the end (a query group cut by a share boundary yields two partial lists, merged afterwards); large
//          shards keep adc_scan16q's (group, row split) blocks and their XCD mapping -- there the row splits of one XCD
//          share each row chunk through its L2 -- and the workgroups take them round-robin.
// A spill area that fills up (masses of equal rows; tables with non-finite entries, whose sums bound nothing) stops the
// workgroup once: every query's candidates are reduced to its k best exact entries, the tables are copied in again and
// the scan resumes -- slow, and only met by such inputs (tests/test_gpu_opq.py::test_search_edge_cases).
#include <algorithm>
#include <vector>

#include "adc_scan16.h"

namespace cvtmi {

constexpr int SH_CAPG = 4096;   // spill entries per (workgroup, query)
constexpr int SH_BINS = 256;    // histogram bins of 128 table units (sums are below 2^15)
constexpr uint32_t SH_UPD = 32; // a query's bound is recomputed every SH_UPD stored candidates
constexpr uint32_t SH_STOP = 0x80000000u;

struct ScanHArgs {
    const uint8_t *codes, *codes_rot;
    int64_t id_base;
    int nq, k, K;
    const ScanItem *items;  // item of workgroup w in round i: items[i * grid + w]; nseg == 0: none
    int rounds;
    const uint4 *qlut;          // [groups][4096]: quantised tables, LDS image
    const QuantParams *qp_g;    // [groups]
    const float *lut_g;         // [nq][16][256] fp32 tables (+inf past K)
    unsigned long long *spill;  // [grid][8][SH_CAPG]
    uint32_t *gthr;             // [nq] bounds shared by the row segments of a query group, or null
    int stride;                 // partial lists per query
    float *part_d;
    int64_t *part_id;
    int seed;
};

// ---- tables of one query group, once: fp32 (IVFOPQ.cpp:279-291 arithmetic) + the quantised LDS image + its parameters ----
__global__ __launch_bounds__(1024) void scan16h_prep_kernel(const float *__restrict__ q_rot, int nq, int D, int step, int K,
                                                            const float *__restrict__ books, const float *__restrict__ centroid,
                                                            float *__restrict__ lut_g, uint4 *__restrict__ qlut,
                                                            QuantParams *__restrict__ qp_g, int lazy_on)
{
    constexpr int NT = 1024, M = 16, QT = SQ_QT;
    __shared__ __attribute__((aligned(16))) uint4 stage[256 * 16];
    __shared__ float res[QT * 256];
    __shared__ QuantParams qp;
    __shared__ uint32_t mx_bits[QT][16];
    __shared__ int nonfinite[QT];
    const int tid = threadIdx.x, lane = tid & 63, group = blockIdx.x;
    for (int i = tid; i < QT * D; i += NT) {
        const int q = i / D, d = i - q * D;
        int qi = group * QT + q;
        qi = qi < nq ? qi : nq - 1;  // ragged last group: the last query again (its slots are never reported)
        res[q * 256 + d] = __fsub_rn(q_rot[(int64_t)qi * D + d], centroid[d]);
    }
    if (tid < QT * 16) {
        qp.mn_bits[tid >> 4][tid & 15] = 0x7f7fffffu;
        mx_bits[tid >> 4][tid & 15] = 0u;
    }
    if (tid < QT) nonfinite[tid] = 0;
    __syncthreads();
    float acc[4][QT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * NT, m = e >> 8, j = e & 255;  // a wave covers 64 consecutive j of one m
        if (j < K) {
            const float *cb = books + ((int64_t)m * K + j) * step;
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[i][q] = 0.0f;
            for (int kk = 0; kk < step; ++kk) {
                const float c = cb[kk];
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const float t = __fsub_rn(res[q * 256 + m * step + kk], c);
                    acc[i][q] = __fadd_rn(acc[i][q], __fmul_rn(t, t));
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[i][q] = __uint_as_float(0x7f800000u);  // code >= K: never a match
        }
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const int qi = group * QT + q;
            if (qi < nq) lut_g[((int64_t)qi * M + m) * 256 + j] = acc[i][q];
            const uint32_t bits = __float_as_uint(acc[i][q]);
            uint32_t lo = bits < 0x7f800000u ? bits : 0x7f7fffffu;  // non-finite: ignored
            uint32_t hi = bits < 0x7f800000u ? bits : 0u;
            if (__ballot(bits >= 0x7f800000u) != 0 && lane == 0) nonfinite[q] = 1;  // (the +inf padding past K counts: a code >= K reaches it)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const uint32_t l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
                lo = l2 < lo ? l2 : lo;
                hi = h2 > hi ? h2 : hi;
            }
            if (lane == 0) {
                atomicMin(&qp.mn_bits[q][m], lo);
                atomicMax(&mx_bits[q][m], hi);
            }
        }
    }
    __syncthreads();
    if (tid < QT) {  // scale, bias, lazy-selection band: as scan16q_build_tables (adc_scan.hip)
        const int q = tid;
        float range = 0.0f;
        double bias = 0.0;
        for (int m = 0; m < M; ++m) {
            uint32_t lo = qp.mn_bits[q][m], hi = mx_bits[q][m];
            if (lo > hi) { lo = 0u; hi = 0u; }  // no finite entry at all
            const float fl = __uint_as_float(lo), fh = __uint_as_float(hi);
            qp.mn[q][m] = fl;
            range += fh - fl;
            bias += (double)fl;
        }
        float scale = range > 0.0f ? range / (float)SQ_MAXSUM * 1.001f : 1.0f;
        if (!(scale > 1e-37f)) scale = 1e-37f;
        const float inv = 1.0f / scale;
        qp.inv_scale[q] = inv;
        qp.scale_eff[q] = 1.0 / (double)inv;
        qp.bias[q] = bias;
        const double sl = 34.0 + ceil(4e-6 * (32767.0 + bias * (double)inv));
        const bool lazy_ok = lazy_on && !nonfinite[q] && sl < 1024.0 && bias >= 0.0;
        qp.slack[q] = lazy_ok ? (uint32_t)sl : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * NT, m = e >> 8, j = e & 255;
        uint32_t qv[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const float v = acc[i][q];
            int iv = 0;
            if (__float_as_uint(v) < 0x7f800000u) {
                const float f = __fmul_rn(__fsub_rn(v, qp.mn[q][m]), qp.inv_scale[q]);
                iv = (int)floorf(f) - 1;
                iv = iv < 0 ? 0 : (iv > SQ_MAXSUM ? SQ_MAXSUM : iv);
            }
            qv[q] = (uint32_t)iv;
        }
        stage[j * 16 + m] = make_uint4(qv[0] | (qv[1] << 16), qv[2] | (qv[3] << 16), qv[4] | (qv[5] << 16), qv[6] | (qv[7] << 16));
    }
    __syncthreads();
    uint4 *dst = qlut + (size_t)group * 4096;
    for (int i = tid; i < 4096; i += NT) dst[i] = stage[i];
    static_assert(sizeof(QuantParams) % 4 == 0, "QuantParams is copied word by word");
    if (tid < (int)(sizeof(QuantParams) / 4))
        reinterpret_cast<uint32_t *>(qp_g + group)[tid] = reinterpret_cast<const uint32_t *>(&qp)[tid];
}

// control words of one workgroup
struct ScanHCtl {
    __attribute__((aligned(16))) uint32_t thr_pk[SQ_QT / 2];  // 15-bit bounds, two per word (what the loop compares with)
    uint32_t ctl;         // epoch of thr_pk (low bits) | SH_STOP
    uint32_t next_chunk;
    int done_waves;
    int cnt[SQ_QT];       // candidates stored (or refused, past SH_CAPG) per query
    uint32_t thr_x[SQ_QT];
    int exact_n[SQ_QT];   // leading spill entries that carry exact keys (after a mid-scan reduction)
    int lazy[SQ_QT];      // the integer sums bound the real sums from both sides (slack != 0)
    // what the out-of-line candidate path needs of the kernel's arguments (uniform; kept here so that its call passes one pointer)
    unsigned long long *spill;
    uint32_t *gthr;
    int k, group, nq;
};
struct ScanHShared {
    __attribute__((aligned(16))) uint32_t hist[SQ_QT][SH_BINS];
    QuantParams qp;
    ScanHCtl ck;
};

// bound of one query from its histogram: (upper edge of the first bin at which the cumulative count reaches k) + slack,
// 0xffffffff while fewer than k rows are counted.  Whole wave, wave-uniform result; counts only grow, so any snapshot of
// the bins is a valid "at least".
__device__ __forceinline__ uint32_t scanh_hist_bound(const uint32_t *hq, int k, uint32_t slack)
{
    const int lane = threadIdx.x & 63;
    const uint4 c = *reinterpret_cast<const uint4 *>(hq + lane * 4);
    const uint32_t mine = c.x + c.y + c.z + c.w;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const unsigned long long reach = __ballot(incl >= (uint32_t)k);
    if (!reach) return 0xffffffffu;
    const int l0 = __ffsll((long long)reach) - 1;
    uint32_t cum = (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), l0);
    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c.x, l0), c1 = (uint32_t)__builtin_amdgcn_readlane((int)c.y, l0),
                   c2 = (uint32_t)__builtin_amdgcn_readlane((int)c.z, l0);
    uint32_t b = (uint32_t)l0 * 4u;
    cum += c0;
    if (cum < (uint32_t)k) { ++b; cum += c1; if (cum < (uint32_t)k) { ++b; cum += c2; if (cum < (uint32_t)k) ++b; } }
    const uint32_t t = ((b + 1u) << 7) + slack;
    return t < 32767u ? t : 32767u;
}

// publish a (possibly) tighter bound of query q: thr_x, its half of thr_pk, the epoch; the segments of the query group share
// it through gthr.  Whole wave (no one-lane regions in the callers' loops: adc_scan.hip "coding rule"); t wave-uniform.
__device__ __forceinline__ void scanh_publish(ScanHCtl &ck, int q, uint32_t t, uint32_t *gthr, int qi, int nq)
{
    const bool l0 = (threadIdx.x & 63) == 0;
    const uint32_t old = (uint32_t)__builtin_amdgcn_readfirstlane((int)atomicMin(&ck.thr_x[q], l0 ? t : 0xffffffffu));
    if (t < old) {  // wave-uniform
        __hip_atomic_store(reinterpret_cast<uint16_t *>(ck.thr_pk) + q, (uint16_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        atomicAdd(&ck.ctl, l0 ? 1u : 0u);
        if (gthr && qi < nq && l0) __hip_atomic_fetch_min(&gthr[qi], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Selection of one query's spilled candidates by ONE wave through the register selection of block_topk.h: entries
// [0, ex) carry exact keys (fp32 bits), the rest integer sums of which only those below T matter.  Leaves the k smallest
// exact (distance, row) entries in tk.buf[q] (sorted when SORTED) and returns their number.  The buffer takes SQ_CAP entries:
// when more survive T, what is there is reduced to its k best exact entries first, whose k-th distance tightens T.
template <bool SORTED>
__device__ __attribute__((noinline)) int scanh_select_q(TopKShared<SQ_QT, SQ_CAP> &tk, int q, int k, const unsigned long long *sp, int n,
                                                       int ex, uint32_t T, const ExactFromLutBatch &fixb, const QuantThr &thrx)
{
    const int lane = threadIdx.x & 63;
    unsigned long long *b = tk.buf[q];
    for (int i = lane; i < ex; i += 64) b[i] = sp[i];  // ex <= k <= 128 < SQ_CAP
    if (lane == 0) { tk.exact_n[q] = ex; tk.thr[q] = KEY_MAX; tk.thr_x[q] = T; }
    int have = ex;
    for (int base = ex; base < n; base += 64) {  // wave-uniform trip count
        const int i = base + lane;
        const unsigned long long e = i < n ? sp[i] : ~0ull;
        bool in = i < n && (uint32_t)(e >> 32) < T;
        unsigned long long m = __ballot(in);
        if (have + __popcll(m) > SQ_CAP) {  // wave-uniform
            have = topk_compact_wave_q<SQ_QT, SQ_CAP, false>(tk, q, k, fixb, thrx, have);
            const uint32_t t2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk.thr_x[q]);
            T = t2 < T ? t2 : T;
            in = in && (uint32_t)(e >> 32) < T;
            m = __ballot(in);
        }
        const int pos = have + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (in) b[pos] = e;
        have += __popcll(m);
    }
    return topk_compact_wave_q<SQ_QT, SQ_CAP, SORTED>(tk, q, k, fixb, thrx, have);
}

template <bool PREROT>
__global__ __launch_bounds__(1024, 8) void adc_scan16h_kernel(const ScanHArgs a)
{
    constexpr int NT = 1024, QT = SQ_QT, NW = NT / 64;
    using TopK = TopKShared<QT, SQ_CAP>;
    static_assert(sizeof(TopK) <= 256 * 16 * QT * 2, "the selection buffers alias the tables");
    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * 16 * QT / 2];  // u16 [code j][m][q]; between scans: the selection buffers
    __shared__ ScanHShared sh;
    uint32_t (&hist)[QT][SH_BINS] = sh.hist;
    QuantParams &qp = sh.qp;
    ScanHCtl &ck = sh.ck;

    SQ_T0();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TopK &tk = *reinterpret_cast<TopK *>(lut);
    const uint4 *rows = reinterpret_cast<const uint4 *>(a.codes);
    unsigned long long *spill = a.spill + (size_t)blockIdx.x * QT * SH_CAPG;
    const QuantThr thrx{ &qp };

    // lane constants of the skewed look-ups (adc_scan.hip, adc_scan16): rotation c = lane & 15 and, one byte per step, the offset
    // m * 16 of sub-quantiser m = (t + c) & 15 inside a 256-byte table row.  Recomputed where they are needed (four registers that
    // would otherwise have to survive every call of the selection code: the allocator then spilled the scan loop's rows instead).
    auto lane_consts = [&](uint32_t (&moffp)[4], uint32_t &cr8, uint32_t &cq) {
        uint32_t c = tid & 15;
        asm volatile("" : "+v"(c));  // (not loop-invariant as far as hipcc can tell: stays where it is written)
        cr8 = (c & 3) * 8; cq = c >> 2;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            moffp[w] = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
        }
    };
    const char *lut_b = reinterpret_cast<const char *>(lut);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t next_chunk_addr = (uint32_t)(uintptr_t)&ck.next_chunk;

    for (int round = 0; round < a.rounds; ++round) {
        ScanItem item = a.items[(size_t)round * gridDim.x + blockIdx.x];
        {  // (workgroup-uniform: keep everything derived from it in scalar registers)
            item.group = __builtin_amdgcn_readfirstlane(item.group);
            item.row0_64 = (uint32_t)__builtin_amdgcn_readfirstlane((int)item.row0_64);
            item.rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)item.rows);
            const int sn = __builtin_amdgcn_readfirstlane((int)item.sidx | ((int)item.nseg << 16));
            item.sidx = (uint16_t)sn; item.nseg = (uint16_t)(sn >> 16);
        }
        if (item.nseg == 0) continue;  // no item in this round (workgroup-uniform); an EMPTY segment still writes its empty lists
        const int group = item.group;
        const int64_t row_begin = (int64_t)item.row0_64 * 64;
        const uint32_t n_local = item.rows;
        const ExactFromLutBatch fixb{ rows, a.lut_g, a.K, a.nq, group };

        auto load_tables = [&]() {
            const uint4 *src = a.qlut + (size_t)group * 4096;
            uint4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = src[tid + i * NT];
#pragma unroll
            for (int i = 0; i < 4; ++i) reinterpret_cast<uint4 *>(lut)[tid + i * NT] = t[i];
        };
        load_tables();
        if (tid < (int)(sizeof(QuantParams) / 4))
            reinterpret_cast<uint32_t *>(&qp)[tid] = reinterpret_cast<const uint32_t *>(a.qp_g + group)[tid];
        for (int i = tid; i < QT * SH_BINS; i += NT) (&hist[0][0])[i] = 0;
        if (tid < QT / 2) {  // pass-all until k rows are counted -- or what the other segments of these queries have established
            uint32_t t2[2] = { 32767u, 32767u };
            if (a.gthr) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int qi = group * QT + 2 * tid + h;
                    if (qi < a.nq) {  // a stale value is an older, looser, still valid bound
                        const uint32_t g = __hip_atomic_load(&a.gthr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        t2[h] = g < t2[h] ? g : t2[h];
                    }
                }
            }
            ck.thr_x[2 * tid] = t2[0]; ck.thr_x[2 * tid + 1] = t2[1];
            ck.thr_pk[tid] = t2[0] | (t2[1] << 16);
        }
        if (tid < QT) { ck.cnt[tid] = 0; ck.exact_n[tid] = 0; }
        if (tid == 0) {
            ck.ctl = 0; ck.next_chunk = 0; ck.done_waves = 0;
            ck.spill = spill; ck.gthr = a.gthr; ck.k = a.k; ck.group = group; ck.nq = a.nq;
        }
        __syncthreads();
        if (tid < QT) ck.lazy[tid] = qp.slack[tid] != 0;
        SQ_T(0);  // tables in

        const char *rows_b = PREROT ? reinterpret_cast<const char *>(reinterpret_cast<const uint4 *>(a.codes_rot) + row_begin)
                                    : reinterpret_cast<const char *>(rows + row_begin);
        const uint32_t n_chunks = (n_local + 63) / 64;
        const uint32_t last_chunk = n_chunks ? n_chunks - 1 : 0;
        // 64 rows of chunk c: one 16-byte load per lane off a scalar base; chunks past the end read the last one, whose lanes past
        // n_local read the next segment's rows or the slack every device buffer carries (never used: adc_scan16q_kernel)
        auto load_rows = [&](uint32_t chunk) -> uint4 {
            const uint32_t cc = chunk < last_chunk ? chunk : last_chunk;  // wave-uniform
            // scalar base + 32-bit lane offset (spelled out: hipcc otherwise keeps rows_b + lane16 as a vector register pair across the loop)
            const uint64_t sb = (uint64_t)(uintptr_t)rows_b + (uint64_t)cc * 1024u;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb >> 32));
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef const u32x4 __attribute__((address_space(1))) *GlobalRows;
            const u32x4 v = reinterpret_cast<GlobalRows>(((uint64_t)hi << 32) | lo)[lane];
            return make_uint4(v.x, v.y, v.z, v.w);
        };
        if (a.seed && n_chunks >= 4 * SQ_SEED_CHUNKS) {  // workgroup-uniform: first bounds from the segment's first 2048 rows
            __syncthreads();  // ck.lazy
            uint32_t moffp[4], cr8, cq;
            lane_consts(moffp, cr8, cq);
            scan16q_seed<NT, PREROT>(a.k, load_rows, moffp, cr8, cq, lut_b, &hist[0][0], qp, ck.lazy, ck.thr_x, ck.thr_pk);
            for (int i = tid; i < QT * SH_BINS; i += NT) (&hist[0][0])[i] = 0;  // those rows are scanned (and counted) again
        }
        __syncthreads();
        SQ_T(1);  // seed

        auto grab_pair = [&]() -> uint32_t {  // chunks go out in pairs: one LDS atomic per 128 rows (adc_scan16q_kernel)
            uint32_t v;
            asm volatile("" : "=v"(v));
            if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(next_chunk_addr), "v"(1u) : "memory");
            return 2u * (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        };
        auto after = [&](uint32_t chunk) -> uint32_t { return (chunk & 1u) ? grab_pair() : chunk + 1u; };
        uint32_t it = grab_pair(), it_next = it + 1u, done = 0, done_for = 0xffffffffu;
        bool counted = false;
        for (;;) {
            // (after a stop the wave comes back to chunk `it`: its rows are loaded again rather than kept across the reduction's calls)
            uint32_t moffp[4], cr8, cq;
            lane_consts(moffp, cr8, cq);
            uint4 cur = make_uint4(0, 0, 0, 0), nxt;
            if (it < n_chunks) cur = load_rows(it);
            bool raised = false;  // this wave asked for a stop
            for (;;) {  // one round per set of bounds: inside the chunk loop they are constants (re-read here whenever the epoch moved)
                uint32_t tpk[QT / 2];
#pragma unroll
                for (int i = 0; i < QT / 2; ++i) tpk[i] = __hip_atomic_load(&ck.thr_pk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t ctl_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ck.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (it >= n_chunks || (ctl_seen & SH_STOP)) break;
                while (it < n_chunks) {
                    const uint32_t ctl_now = __hip_atomic_load(&ck.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // read early, used after the chunk
                    const uint32_t base = it * 64u;
                    nxt = load_rows(it_next);
                    uint32_t s0, s1, s2, s3;
                    scan16q_row_sums<PREROT, 16>(cur, moffp, cr8, cq, lut_b, s0, s1, s2, s3);
                    // sum < T for any of the 8 queries  <=>  a sign bit in the packed (sum - T)  (both < 2^15)
                    const uint32_t sg = (pk_sub_i16(s0, tpk[0]) | pk_sub_i16(s1, tpk[1]) | pk_sub_i16(s2, tpk[2]) | pk_sub_i16(s3, tpk[3])) & 0x80008000u;
                    bool failed_any = false, moved = false;  // wave-uniform
                    if (__builtin_expect(__ballot(sg != 0) != 0, 0)) {  // rare once the bounds have tightened (and the register allocator must know: the loops in here are not the hot ones)
                        if (done_for != it) { done = 0; done_for = it; }  // (the bits belong to one chunk; only this path sets or reads them)
                        uint32_t lrow = base + lane;
                        asm volatile("" : "+v"(lrow));  // (keeps the tail test in here)
                        bool failed = false;
                        uint32_t upd = 0;
                        if (sg != 0 && lrow < n_local) {
                            // ONE copy of the store code, walked by a rolled loop over the query pairs (the sums and bounds rotate through
                            // element 0): unrolled eight times, hipcc kept the eight spill bases and counter / histogram addresses in vector
                            // registers across the whole scan loop and spilled the current rows instead
                            uint32_t sr[4] = { s0, s1, s2, s3 }, tr[4] = { tpk[0], tpk[1], tpk[2], tpk[3] };
                            const uint32_t row = (uint32_t)(row_begin + lrow);
#pragma unroll 1
                            for (int q2 = 0; q2 < QT / 2; ++q2) {
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const int q = 2 * q2 + h;
                                    const uint32_t bit = 1u << q;
                                    const uint32_t sq = h ? sr[0] >> 16 : sr[0] & 0xffffu;
                                    const uint32_t tq = h ? tr[0] >> 16 : tr[0] & 0xffffu;
                                    if (sq < tq && !(done & bit)) {
                                        const int pos = atomicAdd(&ck.cnt[q], 1);
                                        if (pos < SH_CAPG) {
                                            spill[(size_t)q * SH_CAPG + pos] = ((unsigned long long)sq << 32) | row;
                                            atomicAdd(&hist[q][sq >> 7], 1u);
                                            done |= bit;
                                            if ((((uint32_t)pos + 1u) & (SH_UPD - 1u)) == 0) upd |= bit;
                                        } else {
                                            failed = true;
                                        }
                                    }
                                }
                                const uint32_t s_ = sr[0], t_ = tr[0];
                                sr[0] = sr[1]; sr[1] = sr[2]; sr[2] = sr[3]; sr[3] = s_;
                                tr[0] = tr[1]; tr[1] = tr[2]; tr[2] = tr[3]; tr[3] = t_;
                            }
                        }
                        failed_any = __ballot(failed) != 0;
                        if (__builtin_expect(__ballot(upd != 0) != 0, 0)) {  // some query of this wave reached its next multiple of SH_UPD candidates: recompute its bound
#pragma unroll 1
                            for (int q = 0; q < QT; ++q) {
                                if (!__ballot((upd >> q) & 1u)) continue;  // wave-uniform
                                if (!__builtin_amdgcn_readfirstlane(ck.lazy[q])) continue;
                                const uint32_t t = scanh_hist_bound(hist[q], a.k, qp.slack[q]);
                                if (t != 0xffffffffu) { scanh_publish(ck, q, t, a.gthr, group * QT + q, a.nq); moved = true; }
                            }
                        }
                    }
                    if (failed_any) {  // a spill area is full: stop everyone, come back to this chunk after the reduction
                        atomicOr(&ck.ctl, lane == 0 ? SH_STOP : 0u);
                        raised = true;
                        break;
                    }
                    cur = nxt;
                    it = it_next;
                    it_next = after(it);
                    // scalar compare: somebody (this wave included) published a bound or asked for a stop -> next round
                    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)ctl_now) != ctl_seen || moved) break;
                }
                if (raised) break;
            }
            if (it >= n_chunks && !counted) {
                counted = true;
                atomicAdd(&ck.done_waves, lane == 0 ? 1 : 0);
            }
            __syncthreads();  // (A) every wave is out of rows, or a stop is up
            const bool stop = (ck.ctl & SH_STOP) != 0;
            const bool all_done = ck.done_waves == NW;
            if (!stop && all_done) break;  // workgroup-uniform
            __syncthreads();  // (B) everybody has read ctl and done_waves
            if (__builtin_expect(stop, 0)) {
                // mid-scan reduction: the tables make room for the selection buffers, every query keeps its k best exact
                // entries (back at the head of its spill area), then the tables come back
                if (wave < QT) {
                    const int q = wave;
                    int n = ck.cnt[q];
                    n = n < SH_CAPG ? n : SH_CAPG;
                    const int keep = scanh_select_q<false>(tk, q, a.k, spill + (size_t)q * SH_CAPG, n, ck.exact_n[q], ck.thr_x[q], fixb, thrx);
                    for (int i = lane; i < keep; i += 64) spill[(size_t)q * SH_CAPG + i] = tk.buf[q][i];
                    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk.thr_x[q]);
                    if (lane == 0) {
                        ck.cnt[q] = keep;
                        ck.exact_n[q] = keep;
                        const uint32_t old = ck.thr_x[q];
                        const uint32_t nt = t < old ? t : old;
                        ck.thr_x[q] = nt;
                        reinterpret_cast<uint16_t *>(ck.thr_pk)[q] = (uint16_t)(nt < 32767u ? nt : 32767u);
                        const int qi = group * QT + q;
                        if (a.gthr && qi < a.nq) atomicMin(&a.gthr[qi], nt);
                    }
                }
                __syncthreads();
                load_tables();
                if (tid == 0) ck.ctl = (ck.ctl & ~SH_STOP) + 1u;
                __syncthreads();
            }
        }
        SQ_T(2);  // look-ups + candidates

        // ---- the segment's k best: one selection per query, on the dead tables' space ----
        if (wave < QT) {
            const int q = wave;
            int n = ck.cnt[q];
            n = n < SH_CAPG ? n : SH_CAPG;
            const int keep = scanh_select_q<true>(tk, q, a.k, spill + (size_t)q * SH_CAPG, n, ck.exact_n[q], ck.thr_x[q], fixb, thrx);
            if (lane == 0) {
                tk.cnt[q] = keep;
                const int qi = group * QT + q;
                if (a.gthr && qi < a.nq) {  // the exact k-th distance of this segment bounds the other segments' filters too
                    const uint32_t t = tk.thr_x[q];
                    if (t < 32767u) atomicMin(&a.gthr[qi], t);
                }
            }
        }
        __syncthreads();
        SQ_T(3);  // final selection
#pragma unroll 1
        for (int q = 0; q < QT; ++q) {
            const int qi = group * QT + q;
            if (qi >= a.nq) break;
            const int cnt = tk.cnt[q];
            const int64_t o = ((int64_t)qi * a.stride + item.sidx) * a.k;
            for (int i = tid; i < a.k; i += NT) {
                if (i < cnt) {
                    const unsigned long long e = tk.buf[q][i];
                    a.part_d[o + i] = __uint_as_float((uint32_t)(e >> 32));
                    a.part_id[o + i] = a.id_base + (int64_t)(uint32_t)e;
                } else {
                    a.part_d[o + i] = __uint_as_float(0x7f800000u);
                    a.part_id[o + i] = -1;
                }
            }
            if (item.sidx == 0 && item.nseg < a.stride) {  // a group with fewer segments than the partial stride: the other slots stay empty
                const int64_t o2 = ((int64_t)qi * a.stride + item.nseg) * a.k;
                for (int i = tid; i < (a.stride - item.nseg) * a.k; i += NT) {
                    a.part_d[o2 + i] = __uint_as_float(0x7f800000u);
                    a.part_id[o2 + i] = -1;
                }
            }
        }
        __syncthreads();  // the next item's tables overwrite the buffers
        SQ_T(4);  // output
    }
    SQ_TEND();
}

// ---------------------------------------------------------------------------------------------------------------------
// host side: the item table
// ---------------------------------------------------------------------------------------------------------------------
static int g_scanh_balance = 0;       // 0 = choose, 1 = equal shares of the flat (group x row) space, 2 = (group, split) blocks
static int64_t g_scanh_min_rows = 16384;  // smallest share of a workgroup in the balanced plan
void set_scanh_balance(int v) { g_scanh_balance = v; }
void set_scanh_min_rows(int64_t v) { g_scanh_min_rows = v < 2048 ? 2048 : v; }

static int g_scanh_cus = 0;  // cvtmi_opq_scan_plan: planning for a given CU count (0 = the device's)
static int scanh_slots()
{
    if (g_scanh_cus > 0) return 2 * g_scanh_cus;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return 2 * cus;  // two 1024-thread workgroups per CU (LDS: 2 x 75 KB)
}

// Builds the item table of one search.  splits > 0 forces (group, split) blocks with that many row splits.
void scanh_plan(int64_t n_rows, int64_t nq, int splits, ScanHPlan &p)
{
    constexpr int64_t TILE = 2048, MINT = 4, MAX_SEG = (1LL << 28) - 4096;  // rows: segment granule; 32-bit byte offsets inside a segment
    const int64_t groups = (nq + SQ_QT - 1) / SQ_QT;
    const int64_t slots = scanh_slots();
    p.items.clear();
    p.grid = 0; p.rounds = 0; p.stride = 1;
    if (groups <= 0 || n_rows <= 0) return;
    const bool resident = n_rows * 16 <= (96LL << 20);  // the pre-rotated rows stay in the Infinity Cache (and mostly in L2)
    const bool balanced = splits <= 0 && n_rows <= MAX_SEG && (g_scanh_balance == 1 || (g_scanh_balance == 0 && resident));
    struct Seg { int64_t group, row0, rows; int wg; };
    std::vector<Seg> segs;
    if (balanced) {
        const int64_t tg = (n_rows + TILE - 1) / TILE, total = groups * tg;
        const int64_t min_tiles = std::max<int64_t>(1, g_scanh_min_rows / TILE);
        const int64_t P = std::max<int64_t>(1, std::min<int64_t>(slots, total / min_tiles));
        const auto bound = [&](int64_t w) {  // share boundary of workgroup w, in tiles; never closer than MINT tiles to a group boundary
            int64_t b = (int64_t)(((__int128)w * total) / P);
            const int64_t r = b % tg;
            if (r != 0 && r < MINT) b -= r;
            else if (r != 0 && tg - r < MINT) b += tg - r;
            return b;
        };
        for (int64_t w = 0; w < P; ++w) {
            int64_t u = bound(w);
            const int64_t end = w + 1 == P ? total : bound(w + 1);
            while (u < end) {
                const int64_t g = u / tg, t0 = u - g * tg, t1 = std::min(tg, t0 + (end - u));
                const int64_t r0 = t0 * TILE, r1 = std::min(n_rows, t1 * TILE);
                if (r1 > r0) segs.push_back({ g, r0, r1 - r0, (int)w });
                u += t1 - t0;
            }
        }
        p.grid = (int)P;
    } else {
        int64_t S = splits > 0 ? splits : 1;
        const int64_t min_splits = (n_rows + MAX_SEG - 1) / MAX_SEG;
        if (splits <= 0) {
            // blocks of one round share their rows through L2 when a row split stays on one XCD (multiples of 8); enough of them
            // to fill the slots, at least 16 K rows each
            S = min_splits;
            while (groups * S < slots && n_rows / (S * 2) >= 16384) S *= 2;
            if (S > 1 && S < 8 && min_splits > 1) S = 8;
        }
        if (S < min_splits) S = min_splits;
        int64_t rps = (n_rows + S - 1) / S;
        rps = ((rps + TILE - 1) / TILE) * TILE;
        const int64_t blocks = groups * S;
        const int64_t P = std::min<int64_t>(slots, blocks);
        for (int64_t b = 0; b < blocks; ++b) {
            int64_t split, group;
            if ((S & 7) == 0) {
                const int64_t s8 = S >> 3, xcd = b & 7, i = b >> 3;
                split = xcd + 8 * (i % s8);
                group = i / s8;
            } else {
                split = b % S;
                group = b / S;
            }
            const int64_t r0 = split * rps, r1 = std::min(n_rows, r0 + rps);
            // (an empty split still reports: its slot of the partial lists must be written)
            segs.push_back({ group, std::min(r0, n_rows), r1 > r0 ? r1 - r0 : 0, (int)(b % P) });
        }
        p.grid = (int)P;
    }
    // segment index inside its group (ascending rows = ascending ids: the merge's tie rule), segments per group
    std::vector<int> nseg((size_t)groups, 0), sidx(segs.size());
    {
        std::vector<size_t> order(segs.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
            return segs[x].group != segs[y].group ? segs[x].group < segs[y].group : segs[x].row0 < segs[y].row0;
        });
        for (size_t i : order) sidx[i] = nseg[(size_t)segs[i].group]++;
    }
    int stride = 1;
    for (int64_t g = 0; g < groups; ++g) stride = std::max(stride, nseg[(size_t)g]);
    std::vector<int> per_wg((size_t)p.grid, 0);
    for (const Seg &s : segs) per_wg[(size_t)s.wg]++;
    int rounds = 0;
    for (int c : per_wg) rounds = std::max(rounds, c);
    p.items.assign((size_t)rounds * p.grid, ScanItem{ 0, 0, 0, 0, 0 });
    std::fill(per_wg.begin(), per_wg.end(), 0);
    for (size_t i = 0; i < segs.size(); ++i) {
        const Seg &s = segs[i];
        ScanItem it;
        it.group = (int32_t)s.group;
        it.row0_64 = (uint32_t)(s.row0 / 64);
        it.rows = (uint32_t)s.rows;
        it.sidx = (uint16_t)sidx[i];
        it.nseg = (uint16_t)nseg[(size_t)s.group];
        p.items[(size_t)per_wg[(size_t)s.wg]++ * p.grid + s.wg] = it;
    }
    p.rounds = rounds;
    p.stride = stride;
}

size_t scanh_spill_bytes(int grid) { return (size_t)grid * SQ_QT * SH_CAPG * sizeof(unsigned long long); }
size_t scanh_qlut_bytes(int64_t nq) { return (size_t)((nq + SQ_QT - 1) / SQ_QT) * 4096 * sizeof(uint4); }
size_t scanh_qp_bytes(int64_t nq) { return (size_t)((nq + SQ_QT - 1) / SQ_QT) * sizeof(QuantParams); }

int launch_adc_scan_h(const OpqModelDev &m, const uint8_t *codes, const uint8_t *codes_rot, int64_t n_rows, int64_t id_base,
                      const float *q_rot, int64_t nq, int k, const ScanHPlan &plan, const ScanItem *items_dev, float *part_d,
                      int64_t *part_id, float *lut_g, void *qlut, void *qp_g, void *spill, uint32_t *gthr, int lazy, int seed,
                      hipStream_t st)
{
    if (nq <= 0 || plan.grid <= 0) return CVTMI_OK;
    if (m.M != 16 || m.D > 256 || m.K > 256 || m.K < 1) return fail(CVTMI_EUNSUPPORTED, "adc_scan16h: M=%d D=%d K=%d not covered", m.M, m.D, m.K);
    if (k < 1 || k > 128) return fail(CVTMI_EUNSUPPORTED, "adc_scan16h: k=%d outside 1..128", k);
    if (n_rows > 0xfffffffeLL) return fail(CVTMI_EUNSUPPORTED, "adc_scan16h: more than 2^32-2 rows per shard");
    if (nq > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "adc_scan16h: nq too large");
    const int64_t groups = (nq + SQ_QT - 1) / SQ_QT;
    hipLaunchKernelGGL(scan16h_prep_kernel, dim3((unsigned)groups), dim3(1024), 0, st, q_rot, (int)nq, m.D, m.step, m.K, m.books, m.coarse,
                       lut_g, reinterpret_cast<uint4 *>(qlut), reinterpret_cast<QuantParams *>(qp_g), lazy);
    CVTMI_HIP(hipGetLastError());
    if (gthr && plan.stride > 1) CVTMI_HIP(hipMemsetAsync(gthr, 0xff, (size_t)nq * sizeof(uint32_t), st));
    ScanHArgs a;
    a.codes = codes; a.codes_rot = codes_rot; a.id_base = id_base; a.nq = (int)nq; a.k = k; a.K = m.K;
    a.items = items_dev; a.rounds = plan.rounds;
    a.qlut = reinterpret_cast<const uint4 *>(qlut); a.qp_g = reinterpret_cast<const QuantParams *>(qp_g); a.lut_g = lut_g;
    a.spill = reinterpret_cast<unsigned long long *>(spill);
    a.gthr = (gthr && plan.stride > 1) ? gthr : nullptr;
    a.stride = plan.stride; a.part_d = part_d; a.part_id = part_id; a.seed = seed;
    if (codes_rot) hipLaunchKernelGGL((adc_scan16h_kernel<true>), dim3((unsigned)plan.grid), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((adc_scan16h_kernel<false>), dim3((unsigned)plan.grid), dim3(1024), 0, st, a);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi

extern "C" int64_t cvtmi_opq_scan_plan(int64_t n_rows, int64_t nq, int splits, int cus, int64_t *items, int64_t cap, int *grid, int *rounds,
                                       int *stride)
{
    using namespace cvtmi;
    if (n_rows < 0 || nq < 0 || cus < 0 || cap < 0 || (cap > 0 && !items)) return fail(CVTMI_EINVAL, "cvtmi_opq_scan_plan: bad arguments");
    ScanHPlan p;
    g_scanh_cus = cus;
    scanh_plan(n_rows, nq, splits, p);
    g_scanh_cus = 0;
    if (grid) *grid = p.grid;
    if (rounds) *rounds = p.rounds;
    if (stride) *stride = p.stride;
    const int64_t n = (int64_t)p.items.size();
    for (int64_t i = 0; i < n && i < cap; ++i) {
        const ScanItem &it = p.items[(size_t)i];
        items[5 * i + 0] = it.group; items[5 * i + 1] = it.row0_64; items[5 * i + 2] = it.rows; items[5 * i + 3] = it.sidx; items[5 * i + 4] = it.nseg;
    }
    return n;
}

namespace cvtmi {

#ifdef CVTMI_SCAN_TIMING
extern "C" int cvtmi_debug_scanh_timing(unsigned long long *out, int reset)
{
    unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_dbg), sizeof z) != hipSuccess) return -3;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_scan_dbg), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
#endif

}  // namespace cvtmi
