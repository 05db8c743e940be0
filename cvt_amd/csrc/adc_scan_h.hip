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
    uint32_t *ghist;            // [nq][SH_BINS] candidates of ALL segments of a query by bin (groups cut into segments), or null
    int stride;                 // partial lists per query
    float *part_d;
    int64_t *part_id;
    float *out_d;               // final lists [nq][k]: groups with one segment write here when stride > 1 (with stride 1, part_* ARE the final lists)
    int64_t *out_id;
    int seed;
};

// minimum / maximum over the wave in the DPP network (no LDS traffic; __shfl_xor is a ds_bpermute, a dependent LDS round trip per step:
// the 384 of them a thread of the table kernel used to issue were 18 of its 37 us).  Result in every lane.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    const int id = (int)0xffffffffu;
#define CVTMI_DPP_MIN(ctrl, rmask) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, ctrl, rmask, 0xf, false); v = t < v ? t : v; }
    CVTMI_DPP_MIN(0x111, 0xf) CVTMI_DPP_MIN(0x112, 0xf) CVTMI_DPP_MIN(0x114, 0xf) CVTMI_DPP_MIN(0x118, 0xf)
    CVTMI_DPP_MIN(0x142, 0xa) CVTMI_DPP_MIN(0x143, 0xc)
#undef CVTMI_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#define CVTMI_DPP_MAX(ctrl, rmask) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false); v = t > v ? t : v; }
    CVTMI_DPP_MAX(0x111, 0xf) CVTMI_DPP_MAX(0x112, 0xf) CVTMI_DPP_MAX(0x114, 0xf) CVTMI_DPP_MAX(0x118, 0xf)
    CVTMI_DPP_MAX(0x142, 0xa) CVTMI_DPP_MAX(0x143, 0xc)
#undef CVTMI_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

#ifdef CVTMI_SCAN_TIMING
static __device__ unsigned long long g_prep_dbg[8];
#define PREP_T(i) do { if (threadIdx.x == 0) { const unsigned long long n__ = clock64(); atomicAdd(&g_prep_dbg[i], n__ - pt__); pt__ = n__; } } while (0)
#define PREP_T0() unsigned long long pt__ = clock64()
#else
#define PREP_T(i) do { } while (0)
#define PREP_T0() do { } while (0)
#endif
// ---- tables of one query group, once: fp32 (IVFOPQ.cpp:279-291 arithmetic) + the quantised LDS image + its parameters ----
// rot_R / rot_perm (at most one non-null): q_rot holds the RAW queries and the rotation is applied here -- y[i] = the k-ordered fmaf
// chain over R[i][k] x[k] (the arithmetic of the MFMA rotation kernel, rotate.hip) or x[perm[i]] (IVFOPQ.cpp:424-439): the small-batch
// path saves a launch.  zero / zero_words: a scratch area cleared by workgroup 0 (the small-batch path's counters and histograms).
__global__ __launch_bounds__(1024) void scan16h_prep_kernel(const float *__restrict__ q_rot, int nq, int D, int step, int K,
                                                            const float *__restrict__ books, const float *__restrict__ centroid,
                                                            float *__restrict__ lut_g, uint4 *__restrict__ qlut,
                                                            QuantParams *__restrict__ qp_g, int lazy_on,
                                                            const float *__restrict__ rot_R = nullptr, const int32_t *__restrict__ rot_perm = nullptr,
                                                            uint32_t *__restrict__ zero = nullptr, int zero_words = 0)
{
    constexpr int NT = 1024, M = 16, QT = SQ_QT;
    __shared__ __attribute__((aligned(16))) uint4 stage[256 * 16];
    __shared__ float res[QT * 256];
    __shared__ QuantParams qp;
    __shared__ uint32_t mx_bits[QT][16];
    __shared__ int nonfinite[QT];
    const int tid = threadIdx.x, lane = tid & 63, group = blockIdx.x;
    PREP_T0();
    if (rot_R) {
        // R (D <= 128: 64 KB) and the raw queries go through LDS: entry (i, k) of R sits at i * D + ((k + i) & (D - 1)), so the lanes
        // of a wave (consecutive i) read column k from different banks; the k-ordered fmaf chain per output is the arithmetic of the
        // MFMA rotation kernel (rotate.hip).  (Read straight from memory, lane i streaming row i, the chain cost 90 us.)
        float *Rs = reinterpret_cast<float *>(stage);   // the stage area is not needed before the quantised image is built
        {   // (all loads first: one load, one wait, one store at a time this loop alone took 16 us of a workgroup running by itself)
            constexpr int R4 = 128 * 128 / 4 / NT;   // float4 pieces per thread at D = 128
            float4 rv[R4];
            const int n4 = D * D / 4, dsh = 31 - __clz(D);   // D is a power of two (scans_fuses_rotation)
#pragma unroll
            for (int j = 0; j < R4; ++j) rv[j] = tid + j * NT < n4 ? reinterpret_cast<const float4 *>(rot_R)[tid + j * NT] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < R4; ++j) {
                const int e = 4 * (tid + j * NT);
                if (e < D * D) {
                    const int i = e >> dsh, kk = e & (D - 1);
                    const float v4[4] = { rv[j].x, rv[j].y, rv[j].z, rv[j].w };
#pragma unroll
                    for (int t = 0; t < 4; ++t) Rs[i * D + ((kk + t + i) & (D - 1))] = v4[t];
                }
            }
        }
        for (int i = tid; i < QT * D; i += NT) {
            const int q = i / D, d = i - q * D;
            int qi = group * QT + q;
            qi = qi < nq ? qi : nq - 1;
            res[q * 256 + d] = q_rot[(int64_t)qi * D + d];   // raw, for now
        }
        __syncthreads();
        float y[(QT * 256 + NT - 1) / NT];
        int cnt = 0;
        for (int i = tid; i < QT * D; i += NT, ++cnt) {
            const int q = i / D, d = i - q * D;
            float acc = 0.0f;
#pragma unroll 8
            for (int kk = 0; kk < D; ++kk) acc = __fmaf_rn(Rs[d * D + ((kk + d) & (D - 1))], res[q * 256 + kk], acc);
            y[cnt] = acc;
        }
        __syncthreads();
        cnt = 0;
        for (int i = tid; i < QT * D; i += NT, ++cnt) {
            const int q = i / D, d = i - q * D;
            res[q * 256 + d] = __fsub_rn(y[cnt], centroid[d]);
        }
        __syncthreads();
    } else {
        for (int i = tid; i < QT * D; i += NT) {
            const int q = i / D, d = i - q * D;
            int qi = group * QT + q;
            qi = qi < nq ? qi : nq - 1;  // ragged last group: the last query again (its slots are never reported)
            const float *x = q_rot + (int64_t)qi * D;
            const float y = rot_perm ? x[rot_perm[d]] : x[d];
            res[q * 256 + d] = __fsub_rn(y, centroid[d]);
        }
    }
    if (zero)   // (small-batch path: the control words of this query group)
        for (int i = tid; i < zero_words; i += NT) zero[(size_t)blockIdx.x * zero_words + i] = 0u;
    if (tid < QT * 16) {
        qp.mn_bits[tid >> 4][tid & 15] = 0x7f7fffffu;
        mx_bits[tid >> 4][tid & 15] = 0u;
    }
    if (tid < QT) nonfinite[tid] = 0;
    __syncthreads();
    PREP_T(0);  // queries in (rotated)
    float acc[4][QT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * NT, m = e >> 8, j = e & 255;  // a wave covers 64 consecutive j of one m
        if (j < K) {
            const float *cb = books + ((int64_t)m * K + j) * step;
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[i][q] = 0.0f;
            if (step == 8) {  // the usual sub-vector: the centroid in two 16-byte loads (a dword at a time, each waited for, the eight
                              // loads of a pair were most of this kernel's 39 us when one workgroup runs alone)
                const float4 c0 = reinterpret_cast<const float4 *>(cb)[0], c1 = reinterpret_cast<const float4 *>(cb)[1];
                const float cv[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const float t = __fsub_rn(res[q * 256 + m * 8 + kk], cv[kk]);
                        acc[i][q] = __fadd_rn(acc[i][q], __fmul_rn(t, t));
                    }
                }
            } else {
                for (int kk = 0; kk < step; ++kk) {
                    const float c = cb[kk];
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const float t = __fsub_rn(res[q * 256 + m * step + kk], c);
                        acc[i][q] = __fadd_rn(acc[i][q], __fmul_rn(t, t));
                    }
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
            lo = wave_min_u32(lo);
            hi = wave_max_u32(hi);
            if (lane == 0) {
                atomicMin(&qp.mn_bits[q][m], lo);
                atomicMax(&mx_bits[q][m], hi);
            }
        }
    }
    PREP_T(1);  // table entries, fp32 tables out, ranges
    __syncthreads();
    PREP_T(2);
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
    PREP_T(3);  // scales + quantisation into the stage
    __syncthreads();
    uint4 *dst = qlut + (size_t)group * 4096;
    for (int i = tid; i < 4096; i += NT) dst[i] = stage[i];
    static_assert(sizeof(QuantParams) % 4 == 0, "QuantParams is copied word by word");
    if (tid < (int)(sizeof(QuantParams) / 4))
        reinterpret_cast<uint32_t *>(qp_g + group)[tid] = reinterpret_cast<const uint32_t *>(&qp)[tid];
    PREP_T(4);  // image out
}

// control words of one workgroup
struct ScanHCtl {
    __attribute__((aligned(16))) uint32_t thr_pk[SQ_QT / 2];  // 15-bit bounds, two per word (what the loop compares with)
    uint32_t epoch;       // bumped whenever thr_pk (or stop) changes: the scanning waves look at it once per 64 rows
    uint32_t stop;        // a spill area is full: everybody meets at the barrier
    uint32_t next_chunk;
    int done_waves;
    int cnt[SQ_QT];       // spill positions handed out per query (beyond SH_CAPG: refused)
    uint32_t thr_x[SQ_QT];
    int exact_n[SQ_QT];   // leading spill entries that carry exact keys (after a mid-scan reduction)
    int lazy[SQ_QT];      // the integer sums bound the real sums from both sides (slack != 0)
};
constexpr int SH_STAGE = 32;  // candidates a wave collects in its own LDS area before it moves them to the spill areas
struct ScanHShared {
    __attribute__((aligned(16))) uint32_t hist[SQ_QT][SH_BINS];
    unsigned long long stage[16][SH_STAGE];
    QuantParams qp;
    ScanHCtl ck;
};

// inclusive prefix sum over the wave in the DPP network (no LDS traffic: under the scan's load an LDS round trip of one wave
// takes ~0.4 us); the sequence of LLVM's AMDGPUAtomicOptimizer::buildScan for gfx9
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// bound of one query from its histogram: (upper edge of the first bin at which the cumulative count reaches k) + slack,
// 0xffffffff while fewer than k rows are counted.  Whole wave, wave-uniform result; counts only grow, so any snapshot of
// the bins is a valid "at least".
__device__ __forceinline__ uint32_t scanh_hist_bound(const uint32_t *hq, int k, uint32_t slack)
{
    const int lane = threadIdx.x & 63;
    const uint4 c = *reinterpret_cast<const uint4 *>(hq + lane * 4);
    const uint32_t mine = c.x + c.y + c.z + c.w;
    const uint32_t incl = wave_incl_scan_add(mine);
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

// The same bound from the histogram the segments of a query share in HBM (device-scope atomic adds at L2; read around the vector L1).
// Every counted entry is a row of the query, so the k-th smallest integer sum over ALL its rows is below the returned bound.
__device__ __forceinline__ uint32_t scanh_hist_bound_shared(const uint32_t *hq, int k, uint32_t slack)
{
    const int lane = threadIdx.x & 63;
    uint4 c;
    c.x = __hip_atomic_load(hq + lane * 4 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.y = __hip_atomic_load(hq + lane * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.z = __hip_atomic_load(hq + lane * 4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.w = __hip_atomic_load(hq + lane * 4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t mine = c.x + c.y + c.z + c.w;
    const uint32_t incl = wave_incl_scan_add(mine);
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

// publish a (possibly) tighter bound of query q: thr_x, its half of thr_pk, the epoch.  Whole wave, t wave-uniform.  Plain loads and stores: two waves that publish at once leave one of two
// valid bounds, and an epoch that moved at least once.
__device__ __forceinline__ bool scanh_publish(ScanHCtl &ck, int q, uint32_t t)
{
    const uint32_t old = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ck.thr_x[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (t >= old) return false;  // wave-uniform
    __hip_atomic_store(&ck.thr_x[q], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(reinterpret_cast<uint16_t *>(ck.thr_pk) + q, (uint16_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ck.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    __hip_atomic_store(&ck.epoch, e + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return true;
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
    for (int base0 = ex; base0 < n; base0 += 256) {  // wave-uniform trip count; four loads per lane in flight
        unsigned long long e4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = base0 + r * 64 + lane;
            e4[r] = i < n ? sp[i] : ~0ull;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (base0 + r * 64 >= n) break;  // wave-uniform
            const unsigned long long e = e4[r];
            bool in = e != ~0ull && (uint32_t)(e >> 32) < T;
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
    }
    return topk_compact_wave_q<SQ_QT, SQ_CAP, SORTED>(tk, q, k, fixb, thrx, have);
}

#ifdef CVTMI_SCAN_TIMING
static __device__ unsigned long long g_scanh_cnt[8];  // candidates stored, rare-path entries (wave 0), rare-path cycles (wave 0), bound updates (wave 0), update cycles, items, stops
// (accumulated in thread 0's registers, flushed once per item: an atomic per event would sit in the loop's vmcnt queue)
#define SH_CNT(i, v) do { if (threadIdx.x == 0) sh_cnt__[i] += (unsigned long long)(v); } while (0)
#define SH_CNT_DECL() unsigned long long sh_cnt__[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define SH_CNT_FLUSH() do { if (threadIdx.x == 0) { for (int i__ = 0; i__ < 8; ++i__) { if (sh_cnt__[i__]) atomicAdd(&g_scanh_cnt[i__], sh_cnt__[i__]); sh_cnt__[i__] = 0; } } } while (0)
#else
#define SH_CNT(i, v) do { } while (0)
#define SH_CNT_DECL() do { } while (0)
#define SH_CNT_FLUSH() do { } while (0)
#endif

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
    SH_CNT_DECL();
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
    const uint32_t next_chunk_addr = (uint32_t)(uintptr_t)&ck.next_chunk;

    for (int round = 0; round < a.rounds; ++round) {
        ScanItem item = a.items[(size_t)round * gridDim.x + blockIdx.x];
        {  // (workgroup-uniform: keep everything derived from it in scalar registers)
            item.group = __builtin_amdgcn_readfirstlane(item.group);
            item.row0_64 = (uint32_t)__builtin_amdgcn_readfirstlane((int)item.row0_64);
            item.rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)item.rows);
            item.chunk0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)item.chunk0);
            const int sn = __builtin_amdgcn_readfirstlane((int)item.sidx | ((int)item.nseg << 16));
            item.sidx = (uint16_t)sn; item.nseg = (uint16_t)(sn >> 16);
        }
        if (item.nseg == 0) continue;  // no item in this round (workgroup-uniform); an EMPTY segment still writes its empty lists
        const int group = item.group;
        const int64_t row_begin = (int64_t)item.row0_64 * 64;
        const uint32_t n_local = item.rows;
        const ExactFromLutBatch fixb{ rows, a.lut_g, a.K, a.nq, group };
        // bounds shared by the segments of a group: read when a segment starts, written when it ends (a global atomic per published bound
        // sat in the scan loop's vmcnt queue: the row prefetch waited for its acknowledgement)
        uint32_t *const gthr = item.nseg > 1 ? a.gthr : nullptr;
        // candidates of all segments of these queries by bin: a segment's bounds come from the union, i.e. from (segments) times the
        // rows it has seen itself -- the candidates it has to store while its bound is still loose fall accordingly
        uint32_t *const gh = (item.nseg > 1 && a.ghist) ? a.ghist + (size_t)group * QT * SH_BINS : nullptr;
        const int q_valid = a.nq - group * QT;   // queries of this group that exist (the rest are copies of the last one)

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
            if (gthr) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int qi = group * QT + 2 * tid + h;
                    if (qi < a.nq) {  // a stale value is an older, looser, still valid bound
                        const uint32_t g = __hip_atomic_load(&gthr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        t2[h] = g < t2[h] ? g : t2[h];
                    }
                }
            }
            ck.thr_x[2 * tid] = t2[0]; ck.thr_x[2 * tid + 1] = t2[1];
            ck.thr_pk[tid] = t2[0] | (t2[1] << 16);
        }
        if (tid < QT) { ck.cnt[tid] = 0; ck.exact_n[tid] = 0; }
        if (tid == 0) {
            ck.epoch = 0; ck.stop = 0; ck.next_chunk = 0; ck.done_waves = 0;
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
        // walk position -> chunk of the segment: the walk starts at chunk0 and wraps around (positions past the end, the prefetch of a
        // wave that is about to finish, stay on the last position)
        const uint32_t chunk0 = item.chunk0;
        auto chunk_at = [&](uint32_t pos) -> uint32_t {  // wave-uniform
            uint32_t cc = (pos < last_chunk ? pos : last_chunk) + chunk0;
            return cc >= n_chunks && n_chunks ? cc - n_chunks : cc;
        };
        auto load_rows = [&](uint32_t pos) -> uint4 {
            const uint32_t cc = chunk_at(pos);
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
        if (gh) {   // what the other segments of these queries have counted so far
            __syncthreads();
            if (wave < QT && wave < q_valid && __builtin_amdgcn_readfirstlane(ck.lazy[wave])) {
                const uint32_t t = scanh_hist_bound_shared(gh + wave * SH_BINS, a.k, qp.slack[wave]);
                if (t != 0xffffffffu) (void)scanh_publish(ck, wave, t);
            }
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
        uint32_t it = grab_pair(), it_next = it + 1u;
        bool counted = false;
        int wn = 0;  // candidates in this wave's staging area (wave-uniform)
        unsigned long long *stage_w = sh.stage[wave];
        // a chunk whose candidates could not all be staged because a spill area was full (it is walked again after the reduction),
        // and per query the lanes whose candidate was still to come
        uint32_t fail_it = 0xffffffffu;
        unsigned long long fail_mq[QT] = { 0, 0, 0, 0, 0, 0, 0, 0 };

        // Moves the wave's staged candidates to the spill areas: ONE LDS round trip hands out the positions of all eight queries
        // (lane q adds query q's count to its counter), then one 8-byte global store per entry.  Queries whose count crossed a multiple
        // of SH_UPD get their bound recomputed from the histogram.  Returns false when a spill area is full: the refused entries stay
        // (at the head of the staging area) and the workgroup has to stop.  `moved`: a bound was published.
        auto flush = [&](bool &moved) -> bool {
            const bool have = lane < wn;
            const unsigned long long e = have ? stage_w[lane] : 0ull;
            const uint32_t eq = (uint32_t)(e >> 47) & 7u;
            uint32_t cntv = 0, rank = 0;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const unsigned long long m = __ballot(have && eq == (uint32_t)q);
                cntv = lane == q ? (uint32_t)__popcll(m) : cntv;
                const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                rank = eq == (uint32_t)q ? r : rank;
            }
            uint32_t basev = 0;
            if (lane < QT && cntv) basev = (uint32_t)atomicAdd(&ck.cnt[lane], (int)cntv);
            uint32_t mybase = 0, upd = 0;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const uint32_t bq = (uint32_t)__builtin_amdgcn_readlane((int)basev, q), cq_ = (uint32_t)__builtin_amdgcn_readlane((int)cntv, q);
                mybase = eq == (uint32_t)q ? bq : mybase;
                if (cq_ && ((bq + cq_) / SH_UPD != bq / SH_UPD)) upd |= 1u << q;  // scalar
            }
            const uint32_t pos = mybase + rank;
            const bool ok = have && pos < (uint32_t)SH_CAPG;
            if (ok) spill[(size_t)eq * SH_CAPG + pos] = e & 0x00007fffffffffffull;  // (the query tag leaves the key)
            if (gh && ok && (int)eq < q_valid) atomicAdd(&gh[eq * SH_BINS + (((uint32_t)(e >> 32) & 0x7fffu) >> 7)], 1u);   // (no return value: fire and forget)
            const unsigned long long bad = __ballot(have && !ok);
            if (upd) {  // wave-uniform
                [[maybe_unused]] const long long t_u0 = SQA_NOW();
#pragma unroll 1
                for (int q = 0; q < QT; ++q) {
                    if (!((upd >> q) & 1u)) continue;
                    if (!__builtin_amdgcn_readfirstlane(ck.lazy[q])) continue;
                    const uint32_t t = (gh && q < q_valid) ? scanh_hist_bound_shared(gh + q * SH_BINS, a.k, qp.slack[q]) : scanh_hist_bound(hist[q], a.k, qp.slack[q]);
                    if (t != 0xffffffffu && scanh_publish(ck, q, t)) moved = true;
                }
                SH_CNT(3, 1); SH_CNT(4, SQA_NOW() - t_u0);
            }
            if (!bad) { wn = 0; return true; }
            const int r2 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bad >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bad, 0u));
            if (have && !ok) stage_w[r2] = e;
            wn = __popcll(bad);
            __hip_atomic_store(&ck.stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t ep = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ck.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            __hip_atomic_store(&ck.epoch, ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        };

        for (;;) {
            // (after a stop the wave comes back to chunk `it`: its rows are loaded again rather than kept across the reduction's calls)
            uint32_t moffp[4], cr8, cq;
            lane_consts(moffp, cr8, cq);
            uint4 cur = make_uint4(0, 0, 0, 0), nxt;
            if (it < n_chunks) cur = load_rows(it);
            bool raised = false;  // this wave asked for a stop
            if (wn) {  // entries a full spill area refused before the reduction
                bool mv = false;
                raised = !flush(mv);
            }
            while (!raised) {  // one round per set of bounds: inside the chunk loop they are constants (re-read here whenever the epoch moved)
                uint32_t tpk[QT / 2];
#pragma unroll
                for (int i = 0; i < QT / 2; ++i) tpk[i] = __hip_atomic_load(&ck.thr_pk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t ep_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ck.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (it >= n_chunks || __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ck.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) break;
                while (it < n_chunks) {
                    const uint32_t ep_now = __hip_atomic_load(&ck.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // read early, used after the chunk
                    const uint32_t base = chunk_at(it) * 64u;
                    nxt = load_rows(it_next);
                    uint32_t s0, s1, s2, s3;
                    scan16q_row_sums<PREROT, 16>(cur, moffp, cr8, cq, lut_b, s0, s1, s2, s3);
                    // sum < T for any of the 8 queries  <=>  a sign bit in the packed (sum - T)  (both < 2^15)
                    const uint32_t d0 = pk_sub_i16(s0, tpk[0]), d1 = pk_sub_i16(s1, tpk[1]), d2 = pk_sub_i16(s2, tpk[2]), d3 = pk_sub_i16(s3, tpk[3]);
                    const uint32_t sg = (d0 | d1 | d2 | d3) & 0x80008000u;
                    bool moved = false;  // wave-uniform
                    if (__builtin_expect(__ballot(sg != 0) != 0, 0)) {  // one chunk in five (and the register allocator must know: the loops in here are not the hot ones)
                        // Every vector instruction in here is paid for twice: the wave issues one per ~32 cycles beside seven others, and the
                        // look-up loops of those seven are bound by the same issue port.  So: which queries have candidates is decided on
                        // scalar masks (eight compares), and only those queries cost vector work -- rank by lane count (no atomic, no round
                        // trip), one 8-byte LDS store into the wave's own staging area, one histogram increment.
                        [[maybe_unused]] const long long t_r0 = SQA_NOW();
                        unsigned long long tail = ~0ull;
                        if (base + 64u > n_local) tail = n_local > base ? ((1ull << (n_local - base)) - 1ull) : 0ull;  // scalar: lanes inside the segment
                        const bool redo = it == fail_it;  // scalar: second walk over a chunk a stop interrupted
                        // (the lane number is recomputed: kept in a register across the scan loop it ends up in scratch, and its reload would wait for the row prefetch)
                        uint32_t ones = ~0u;
                        asm volatile("" : "+s"(ones));  // (or hipcc recognises the lane number and reloads the spilled one)
                        const uint32_t row = (uint32_t)row_begin + base + __builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
                        const uint32_t dd[4] = { d0, d1, d2, d3 }, ss[4] = { s0, s1, s2, s3 };
                        unsigned long long mq[QT];  // per query: lanes whose row is a candidate (scalar masks)
#pragma unroll
                        for (int i = 0; i < QT / 2; ++i) {
                            mq[2 * i] = __ballot((dd[i] & 0x8000u) != 0) & tail;
                            mq[2 * i + 1] = __ballot((int32_t)dd[i] < 0) & tail;
                        }
                        if (redo) {
#pragma unroll
                            for (int q = 0; q < QT; ++q) mq[q] &= fail_mq[q];
                        }
                        for (;;) {  // one round, unless the chunk holds more candidates than the staging area has room for
                            bool left = false;
#pragma unroll
                            for (int q = 0; q < QT; ++q) {
                                if (!mq[q]) continue;  // scalar
                                const unsigned long long m = mq[q];
                                const int room = SH_STAGE - wn;
                                const int n = __popcll(m);
                                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                                unsigned long long take = m;
                                if (n > room) take = __ballot(__builtin_amdgcn_inverse_ballot_w64(m) && rank < (uint32_t)room);
                                if (__builtin_amdgcn_inverse_ballot_w64(take)) {  // exec = take: no vector compare
                                    uint32_t sw = ss[q >> 1];
                                    asm volatile("" : "+v"(sw));  // (keeps the entry's arithmetic inside this branch: hipcc otherwise computes all eight queries' entries up front)
                                    const uint32_t sq = (q & 1) ? sw >> 16 : sw & 0xffffu;
                                    stage_w[(uint32_t)wn + rank] = ((unsigned long long)(((uint32_t)q << 15) | sq) << 32) | row;
                                    atomicAdd(&hist[q][sq >> 7], 1u);
                                }
                                wn += n > room ? room : n;
                                mq[q] = m & ~take;
                                left |= mq[q] != 0;
                            }
                            if (left || wn > SH_STAGE - 6) {
                                if (!flush(moved)) {
                                    raised = true;
                                    fail_it = it;
#pragma unroll
                                    for (int q = 0; q < QT; ++q) fail_mq[q] = mq[q];
                                    break;
                                }
                            }
                            if (!left) break;
                        }
                        if (redo && !raised) fail_it = 0xffffffffu;
                        SH_CNT(1, 1); SH_CNT(2, SQA_NOW() - t_r0);
                    }
                    if (raised) break;  // a spill area is full: everybody stops, this wave comes back to this chunk after the reduction
                    cur = nxt;
                    it = it_next;
                    it_next = after(it);
                    SH_CNT(7, 1);
                    // scalar compare: somebody (this wave included) published a bound or asked for a stop -> next round
                    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)ep_now) != ep_seen || moved) break;
                }
            }
            if (!raised && it >= n_chunks && wn) {  // out of rows: the rest of the staging area
                bool mv = false;
                raised = !flush(mv);
            }
            if (it >= n_chunks && wn == 0 && !counted) {
                counted = true;
                atomicAdd(&ck.done_waves, lane == 0 ? 1 : 0);
            }
            __syncthreads();  // (A) every wave is out of rows, or a stop is up
            const bool stop = ck.stop != 0;
            const bool all_done = ck.done_waves == NW;
            if (!stop && all_done) break;  // workgroup-uniform
            __syncthreads();  // (B) everybody has read stop and done_waves
            if (__builtin_expect(stop, 0)) {
                SH_CNT(6, 1);
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
                        if (gthr && qi < a.nq) atomicMin(&gthr[qi], nt);
                    }
                }
                __syncthreads();
                load_tables();
                if (tid == 0) { ck.stop = 0; ck.epoch = ck.epoch + 1u; }
                __syncthreads();
            }
        }
        SQ_T(2);  // look-ups + candidates
#ifdef CVTMI_SCAN_TIMING
        if (tid < QT) atomicAdd(&g_scanh_cnt[0], (unsigned long long)ck.cnt[tid]);
        SH_CNT(5, 1);
        SH_CNT_FLUSH();
#endif

        // ---- the segment's k best: one selection per query, on the dead tables' space ----
        if (wave < QT) {
            const int q = wave;
            int n = ck.cnt[q];
            n = n < SH_CAPG ? n : SH_CAPG;
            const int keep = scanh_select_q<true>(tk, q, a.k, spill + (size_t)q * SH_CAPG, n, ck.exact_n[q], ck.thr_x[q], fixb, thrx);
            if (lane == 0) {
                tk.cnt[q] = keep;
                const int qi = group * QT + q;
                if (gthr && qi < a.nq) {  // the other segments of the group start from this one's bound (integer or from its exact k-th distance)
                    const uint32_t t = tk.thr_x[q], t0 = ck.thr_x[q];
                    const uint32_t tm = t < t0 ? t : t0;
                    if (tm < 32767u) atomicMin(&gthr[qi], tm);
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
            // a group scanned in one piece reports straight into the result arrays (the merge skips its queries)
            const bool whole = item.nseg == 1 && a.stride > 1;
            float *od = whole ? a.out_d : a.part_d;
            int64_t *oi = whole ? a.out_id : a.part_id;
            const int64_t o = whole ? (int64_t)qi * a.k : ((int64_t)qi * a.stride + item.sidx) * a.k;
            for (int i = tid; i < a.k; i += NT) {
                if (i < cnt) {
                    const unsigned long long e = tk.buf[q][i];
                    od[o + i] = __uint_as_float((uint32_t)(e >> 32));
                    oi[o + i] = a.id_base + (int64_t)(uint32_t)e;
                } else {
                    od[o + i] = __uint_as_float(0x7f800000u);
                    oi[o + i] = -1;
                }
            }
            if (whole) continue;
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

// =====================================================================================================================
// Small batches (1 .. 8 queries: the reference's own call pattern is 1-9 query frames per Query, multi_frame_index_test.cpp:45-54).
//
// One query group cannot amortise adc_scan16h's per-segment costs: with the rows cut into a few hundred segments every segment
// would warm its bounds up on its own (~500 candidates per query each).  Here the bound is global and comes first:
//   scan16s_hist_kernel     G workgroups, each over its row block: integer sums of every 4th 64-row chunk into a histogram per
//                           query (bin = sum >> 7), added to one global histogram.  The first bin at which the cumulative count
//                           reaches k proves k rows below its edge, so T = edge + slack admits every row that can be among the
//                           k best (the lazy-selection argument, adc_scan16.h) -- about 4 k rows pass it.
//   scan16s_collect_kernel  the same row blocks again, all chunks, against T: the rows below it go to one global list per query
//                           (one atomic per wave and query).  The workgroup that finishes last selects: k-th smallest integer
//                           sum of the list by two histogram passes, exact reference-order sums of the rows within `slack` of
//                           it, sort, results.  A list that overflows (masses of equal rows) or a query whose sums bound nothing
//                           (non-finite tables) is answered by that workgroup with an exact pass over all rows.
// Three launches with scan16h_prep_kernel (rotation folded in), none of them waiting for another workgroup.
// =====================================================================================================================
static int scanh_slots();
constexpr int SS_WCAP = 64;        // candidates per (workgroup, query) of the collect pass; more: the query takes the exact fall-back
constexpr int SS_NMAX = 2048;      // candidates per query the selection takes (LDS); more: exact fall-back
constexpr int SS_SAMPLE_SHIFT = 2; // the histogram pass looks at every 4th chunk
// scratch words: [0, 2048) histograms, [2048, 2056) the bounds T (written by workgroup 0 of the collect pass)
constexpr int SS_WORDS = SQ_QT * SH_BINS + SQ_QT + 8;

struct ScanSArgs {
    const uint8_t *codes, *codes_rot;
    int64_t n_rows, id_base, rows_per_wg, rows_per_hist_wg;
    int nq, k, K, G;              // G = workgroups of the collect pass
    const uint4 *qlut;
    const QuantParams *qp_g;
    const float *lut_g;
    uint32_t *ctl;                // SS_WORDS, zeroed by the prep kernel
    uint32_t *wcnt;               // [G][8]: candidates workgroup g found for query q (> SS_WCAP: it dropped some)
    unsigned long long *gcand;    // [G][8][SS_WCAP]
    float *out_d;
    int64_t *out_id;
    int dbg;   // timing experiments (results wrong when non-zero): 1 = no selection, 2 = no row pass either
};

template <bool PREROT>
__global__ __launch_bounds__(1024) void scan16s_hist_kernel(const ScanSArgs a)   // (one workgroup per CU: 128 registers per lane)
{
    constexpr int NT = 1024, QT = SQ_QT;
    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * 16 * QT / 2];
    __shared__ uint32_t hist[QT][SH_BINS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = blockIdx.y;   // one grid row per query group
    const uint4 *qlut = a.qlut + (size_t)grp * 4096;
    uint32_t *ctl = a.ctl + (size_t)grp * SS_WORDS;
    {
        uint4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = qlut[tid + i * NT];
#pragma unroll
        for (int i = 0; i < 4; ++i) reinterpret_cast<uint4 *>(lut)[tid + i * NT] = t[i];
    }
    for (int i = tid; i < QT * SH_BINS; i += NT) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t c = tid & 15, cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_hist_wg;
    int64_t r1 = r0 + a.rows_per_hist_wg;
    r1 = r1 < a.n_rows ? r1 : a.n_rows;
    const uint4 *rows = reinterpret_cast<const uint4 *>(PREROT ? a.codes_rot : a.codes);
    for (int64_t base = r0 + ((int64_t)wave << (6 + SS_SAMPLE_SHIFT)); base < r1; base += (int64_t)(NT / 64) << (6 + SS_SAMPLE_SHIFT)) {
        const int64_t row = base + lane;
        const uint4 v = rows[row < r1 ? row : r1 - 1];
        uint32_t sm[4];
        scan16q_row_sums<PREROT>(v, moffp, cr8, cq, lut_b, sm[0], sm[1], sm[2], sm[3]);
        if (row < r1) {
#pragma unroll
            for (int q = 0; q < QT; ++q) atomicAdd(&hist[q][((sm[q >> 1] >> (16 * (q & 1))) & 0xffffu) >> 7], 1u);
        }
    }
    __syncthreads();
    for (int i = tid; i < QT * SH_BINS; i += NT) {
        const uint32_t v = (&hist[0][0])[i];
        if (v) atomicAdd(&ctl[i], v);
    }
}

// exact answer of ONE query by the calling workgroup alone: every row, fp32 sums in the reference's order from the query's table
// (IVFOPQ.cpp:302-306), k smallest (distance, id) through block_topk.h -- the fall-back of the small-batch path.  lds: >= 16 KB of
// table + the selection buffer; ends with the results written.
template <int NT>
__device__ void scans_exact_query(const ScanSArgs &a, int q, uint32_t *lds)
{
    constexpr int CAPX = 384, TRIGX = 256, RX = 2;
    float *lutf = reinterpret_cast<float *>(lds);
    TopKShared<1, CAPX> &tk = *reinterpret_cast<TopKShared<1, CAPX> *>(lds + 16 * 256);
    const int tid = threadIdx.x;
    __syncthreads();
    for (int i = tid; i < 16 * 256; i += NT) lutf[i] = a.lut_g[(int64_t)q * 16 * 256 + i];
    topk_init(tk);
    __syncthreads();
    const uint4 *rows = reinterpret_cast<const uint4 *>(a.codes);
    int tile = 0;
    for (int64_t base = 0; base < a.n_rows; base += (int64_t)NT * RX, ++tile) {
        uint32_t key[RX][1], pay[RX];
#pragma unroll
        for (int r = 0; r < RX; ++r) {
            const int64_t row = base + r * NT + tid;
            const bool valid = row < a.n_rows;
            const uint4 cw = rows[valid ? row : 0];
            const uint32_t w[4] = { cw.x, cw.y, cw.z, cw.w };
            float sum = 0.0f;
#pragma unroll
            for (int m = 0; m < 16; ++m) sum = __fadd_rn(sum, lutf[m * 256 + ((w[m >> 2] >> (8 * (m & 3))) & 0xffu)]);
            pay[r] = (uint32_t)row;
            key[r][0] = valid ? __float_as_uint(sum) : KEY_MAX;  // sums are >= +0
        }
        topk_tile<1, RX, CAPX, TRIGX, NT>(tk, a.k, tile, key, pay);
    }
    __syncthreads();
    topk_compact<1, CAPX, NT>(tk, a.k);
    const int cnt = tk.cnt[0];
    for (int i = tid; i < a.k; i += NT) {
        if (i < cnt) {
            const unsigned long long e = tk.buf[0][i];
            a.out_d[(int64_t)q * a.k + i] = __uint_as_float((uint32_t)(e >> 32));
            a.out_id[(int64_t)q * a.k + i] = a.id_base + (int64_t)(uint32_t)e;
        } else {
            a.out_d[(int64_t)q * a.k + i] = __uint_as_float(0x7f800000u);
            a.out_id[(int64_t)q * a.k + i] = -1;
        }
    }
    __syncthreads();
}

// the rows of the workgroup's block below the bounds -> its own candidate lists (no global atomic: 256 workgroups bumping eight shared
// counters, and one "who is last" ticket, cost 70 us of a 130 us kernel)
template <bool PREROT>
__global__ __launch_bounds__(1024) void scan16s_collect_kernel(const ScanSArgs a)
{
    constexpr int NT = 1024, QT = SQ_QT;
    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * 16 * QT / 2];
    __shared__ __attribute__((aligned(16))) uint32_t hist[QT][SH_BINS];
    __shared__ QuantParams qp;
    __shared__ uint32_t T[QT], tpk_s[QT / 2];
    __shared__ uint32_t cnt[QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = blockIdx.y;   // one grid row per query group
    const uint4 *qlut = a.qlut + (size_t)grp * 4096;
    uint32_t *ctl = a.ctl + (size_t)grp * SS_WORDS;
    {
        uint4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = qlut[tid + i * NT];
#pragma unroll
        for (int i = 0; i < 4; ++i) reinterpret_cast<uint4 *>(lut)[tid + i * NT] = t[i];
    }
    if (tid < (int)(sizeof(QuantParams) / 4)) reinterpret_cast<uint32_t *>(&qp)[tid] = reinterpret_cast<const uint32_t *>(a.qp_g + grp)[tid];
    for (int i = tid; i < QT * SH_BINS; i += NT) (&hist[0][0])[i] = ctl[i];
    if (tid < QT) cnt[tid] = 0;
    __syncthreads();
    if (wave < QT) {  // the bound of query `wave` from the global histogram (a sample: every 4th chunk of every row block)
        const uint32_t sl = qp.slack[wave];
        uint32_t t = sl ? scanh_hist_bound(hist[wave], a.k, sl) : 0xffffffffu;
        if (t > 32767u) t = 32767u;  // fewer than k rows sampled, or sums that bound nothing: every row is a candidate
        if (lane == 0) {
            T[wave] = t;
            if (blockIdx.x == 0) ctl[QT * SH_BINS + wave] = t;   // for the selection kernel
        }
    }
    __syncthreads();
    if (tid < QT / 2) tpk_s[tid] = T[2 * tid] | (T[2 * tid + 1] << 16);
    __syncthreads();
    const uint32_t c = tid & 15, cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);
    uint32_t tpk[QT / 2];
#pragma unroll
    for (int i = 0; i < QT / 2; ++i) tpk[i] = tpk_s[i];
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_wg;
    int64_t r1 = r0 + a.rows_per_wg;
    r1 = r1 < a.n_rows ? r1 : a.n_rows;
    const uint4 *rows = reinterpret_cast<const uint4 *>(PREROT ? a.codes_rot : a.codes);
    unsigned long long *mine = a.gcand + ((size_t)grp * a.G + blockIdx.x) * QT * SS_WCAP;
    for (int64_t base = r0 + ((int64_t)wave << 6); base < r1 && a.dbg < 2; base += NT) {
        const int64_t row = base + lane;
        const uint4 v = rows[row < r1 ? row : r1 - 1];
        uint32_t s4[4];
        scan16q_row_sums<PREROT, 16>(v, moffp, cr8, cq, lut_b, s4[0], s4[1], s4[2], s4[3]);
        uint32_t d4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d4[i] = pk_sub_i16(s4[i], tpk[i]);
        const bool inr = row < r1;
        if (__ballot(inr && ((d4[0] | d4[1] | d4[2] | d4[3]) & 0x80008000u) != 0) == 0) continue;  // wave-uniform
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const bool cand = inr && ((q & 1) ? (int32_t)d4[q >> 1] < 0 : (d4[q >> 1] & 0x8000u) != 0);
            const unsigned long long m = __ballot(cand);
            if (!m) continue;  // scalar
            uint32_t basep = 0;
            if (lane == 0) basep = atomicAdd(&cnt[q], (uint32_t)__popcll(m));   // LDS
            basep = (uint32_t)__builtin_amdgcn_readfirstlane((int)basep);
            const uint32_t pos = basep + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (cand && pos < (uint32_t)SS_WCAP) {
                const uint32_t sq = (q & 1) ? s4[q >> 1] >> 16 : s4[q >> 1] & 0xffffu;
                mine[q * SS_WCAP + pos] = ((unsigned long long)sq << 32) | (uint32_t)row;
            }
        }
    }
    __syncthreads();
    if (tid < QT) a.wcnt[((size_t)grp * a.G + blockIdx.x) * QT + tid] = cnt[tid];
}

// one workgroup per query: the candidates of all collect workgroups, the k-th smallest integer sum among them (two histogram
// passes), exact reference-order sums of the rows within `slack` of it, sort, results -- or, when a workgroup dropped candidates
// (masses of equal rows), the list is too long, or the query's sums bound nothing (non-finite tables), the exact pass over all rows
// (the launch bounds of adc_scan16h_kernel: the out-of-line selection code is shared with it, and the callee is compiled for the loosest
//  bound among its callers -- with 128 registers allowed here the scan kernel grew to 120 and lost its second workgroup per CU: +15 %)
__global__ __launch_bounds__(1024, 8) void scan16s_select_kernel(const ScanSArgs a)
{
    constexpr int NT = 1024, QT = SQ_QT;
    using TopK = TopKShared<QT, SQ_CAP>;
    __shared__ __attribute__((aligned(16))) uint32_t lds[16 * 256 + 1024];   // exact fall-back: table + selection buffer
    __shared__ __attribute__((aligned(16))) unsigned long long cand[SS_NMAX];
    __shared__ __attribute__((aligned(16))) uint32_t h[SH_BINS];
    __shared__ TopK tk;
    __shared__ QuantParams qp;
    __shared__ uint32_t off[1024 + 1];
    __shared__ int s_slow;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x, grp = q / QT, ql = q % QT;   // query, its group, its place there
    if (a.dbg) return;
    if (tid < (int)(sizeof(QuantParams) / 4)) reinterpret_cast<uint32_t *>(&qp)[tid] = reinterpret_cast<const uint32_t *>(a.qp_g + grp)[tid];
    // candidates per collect workgroup (G <= 1024) -> exclusive offsets
    uint32_t c = tid < a.G ? a.wcnt[((size_t)grp * a.G + tid) * QT + ql] : 0u;
    const bool dropped = c > (uint32_t)SS_WCAP;
    off[tid] = c;
    if (tid == 0) s_slow = 0;
    __syncthreads();
    if (dropped) s_slow = 1;
    if (tid == 0) {   // (G values: a serial prefix is a few microseconds at most; workgroups are few)
        uint32_t run = 0;
        for (int g = 0; g < a.G; ++g) { const uint32_t v = off[g]; off[g] = run; run += v; }
        off[a.G] = run;
    }
    __syncthreads();
    const uint32_t n = off[a.G];
    const uint32_t Tq = a.ctl[(size_t)grp * SS_WORDS + QT * SH_BINS + ql];
    if (s_slow || n > (uint32_t)SS_NMAX || Tq >= 32767u) {   // workgroup-uniform
        scans_exact_query<NT>(a, q, lds);
        return;
    }
    if (tid < a.G) {
        const unsigned long long *src = a.gcand + (((size_t)grp * a.G + tid) * QT + ql) * SS_WCAP;
        for (uint32_t j = 0; j < c; ++j) cand[off[tid] + j] = src[j];
    }
    __syncthreads();
    if (wave != 0) return;
    uint32_t Tsel = Tq;
    if ((int)n > a.k + 64 && qp.slack[ql]) {
        // k-th smallest integer sum of the list: bin of 128 by one histogram pass, position inside the bin by a second; rows at
        // S_k + slack and beyond are beaten by k rows (adc_scan16.h), the few below it get exact sums
        for (int i = lane; i < SH_BINS; i += 64) h[i] = 0;
        for (uint32_t i = lane; i < n; i += 64) atomicAdd(&h[(uint32_t)(cand[i] >> 32) >> 7], 1u);
        const uint4 cb = *reinterpret_cast<const uint4 *>(h + lane * 4);
        const uint32_t mine = cb.x + cb.y + cb.z + cb.w, incl = wave_incl_scan_add(mine);
        const unsigned long long reach = __ballot(incl >= (uint32_t)a.k);
        if (reach) {
            const int l0 = __ffsll((long long)reach) - 1;
            uint32_t cum = (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), l0);
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cb.x, l0), c1 = (uint32_t)__builtin_amdgcn_readlane((int)cb.y, l0),
                           c2 = (uint32_t)__builtin_amdgcn_readlane((int)cb.z, l0);
            uint32_t b = (uint32_t)l0 * 4u;
            if (cum + c0 < (uint32_t)a.k) { cum += c0; ++b; if (cum + c1 < (uint32_t)a.k) { cum += c1; ++b; if (cum + c2 < (uint32_t)a.k) { cum += c2; ++b; } } }
            // cum rows lie below bin b; the k-th is the (k - cum)-th smallest inside it
            for (int i = lane; i < 128; i += 64) h[i] = 0;
            for (uint32_t i = lane; i < n; i += 64) {
                const uint32_t sv = (uint32_t)(cand[i] >> 32);
                if ((sv >> 7) == b) atomicAdd(&h[sv & 127u], 1u);
            }
            const uint32_t f0 = h[lane * 2], f1 = h[lane * 2 + 1];
            const uint32_t inc2 = wave_incl_scan_add(f0 + f1);
            const uint32_t need = (uint32_t)a.k - cum;
            const unsigned long long r2 = __ballot(inc2 >= need);
            if (r2) {
                const int l2 = __ffsll((long long)r2) - 1;
                const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)(inc2 - f0 - f1), l2);
                const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)f0, l2);
                const uint32_t sk = (b << 7) + (uint32_t)l2 * 2u + (before + g0 >= need ? 0u : 1u);
                const uint32_t t2 = sk + qp.slack[ql];
                Tsel = t2 < Tsel ? t2 : Tsel;
            }
        }
    }
    const QuantThr thrx{ &qp };
    const ExactFromLutBatch fixb{ reinterpret_cast<const uint4 *>(a.codes), a.lut_g, a.K, a.nq, grp };
    const int keep = scanh_select_q<true>(tk, ql, a.k, cand, (int)n, 0, Tsel, fixb, thrx);
    for (int i = lane; i < a.k; i += 64) {
        if (i < keep) {
            const unsigned long long e = tk.buf[ql][i];
            a.out_d[(int64_t)q * a.k + i] = __uint_as_float((uint32_t)(e >> 32));
            a.out_id[(int64_t)q * a.k + i] = a.id_base + (int64_t)(uint32_t)e;
        } else {
            a.out_d[(int64_t)q * a.k + i] = __uint_as_float(0x7f800000u);
            a.out_id[(int64_t)q * a.k + i] = -1;
        }
    }
}

static int g_scans_dbg = 0;
void set_scans_dbg(int v) { g_scans_dbg = v; }
constexpr int SS_GMAX = 1024;   // collect workgroups at most (the selection kernel's prefix over them)
constexpr int SS_GROUPS = 16;   // query groups the small-batch path takes (128 queries; the dispatch stops earlier, where the persistent grid catches up)
static int scans_collect_max() { const int g = scanh_slots() / 2; return g < SS_GMAX ? g : SS_GMAX; }   // collect workgroups per group: one per CU
size_t scans_scratch_bytes()
{
    const size_t gm = (size_t)scans_collect_max();
    return ((size_t)SS_GROUPS * SS_WORDS * 4 + 63) / 64 * 64 + (size_t)SS_GROUPS * gm * SQ_QT * 4 + (size_t)SS_GROUPS * gm * SQ_QT * SS_WCAP * sizeof(unsigned long long);
}
// the preparation kernel rotates in LDS: a dense R needs D a power of two <= 128 (64 KB); a permutation or no rotation: any D
bool scans_fuses_rotation(const OpqModelDev &m) { return !m.R || (m.D <= 128 && m.D >= 4 && (m.D & (m.D - 1)) == 0); }
bool scans_applies(const OpqModelDev &m, int64_t n_rows, int64_t nq, int k)
{
    return m.M == 16 && m.D <= 256 && m.K >= 1 && m.K <= 256 && nq >= 1 && nq <= SS_GROUPS * SQ_QT && k >= 1 && k <= 128 && n_rows >= 65536 && n_rows <= 0xfffffffeLL;
}

// q: RAW queries when rotate != 0 (the model's rotation is applied by the preparation kernel), rotated ones otherwise
int launch_adc_scan_small(const OpqModelDev &m, const uint8_t *codes, const uint8_t *codes_rot, int64_t n_rows, int64_t id_base, const float *q,
                          int rotate, int64_t nq, int k, float *dist, int64_t *ids, float *lut_g, void *qlut, void *qp_g, void *scratch, int lazy,
                          hipStream_t st)
{
    if (!scans_applies(m, n_rows, nq, k)) return fail(CVTMI_EUNSUPPORTED, "adc_scan16s: shape not covered");
    uint32_t *ctl = reinterpret_cast<uint32_t *>(scratch);
    const unsigned ng = (unsigned)((nq + SQ_QT - 1) / SQ_QT);   // query groups: a grid row each in the two row passes
    hipLaunchKernelGGL(scan16h_prep_kernel, dim3(ng), dim3(1024), 0, st, q, (int)nq, m.D, m.step, m.K, m.books, m.coarse, lut_g,
                       reinterpret_cast<uint4 *>(qlut), reinterpret_cast<QuantParams *>(qp_g), lazy, rotate ? m.R : nullptr, rotate ? m.perm : nullptr,
                       ctl, SS_WORDS);   // (a dense rotation: D a power of two <= 128, checked by the caller through scans_fuses_rotation)
    CVTMI_HIP(hipGetLastError());
    ScanSArgs a;
    a.codes = codes; a.codes_rot = codes_rot; a.n_rows = n_rows; a.id_base = id_base;
    const int64_t grid = std::min<int64_t>(scanh_slots() / 2, (n_rows + 4095) / 4096);   // one workgroup per CU, at least 4096 rows each
    a.rows_per_wg = ((n_rows + grid - 1) / grid + 255) / 256 * 256;                       // (whole groups of four chunks: the histogram's sample)
    a.nq = (int)nq; a.k = k; a.K = m.K;
    a.qlut = reinterpret_cast<const uint4 *>(qlut); a.qp_g = reinterpret_cast<const QuantParams *>(qp_g); a.lut_g = lut_g;
    a.ctl = ctl;
    char *sc = reinterpret_cast<char *>(scratch) + ((size_t)SS_GROUPS * SS_WORDS * 4 + 63) / 64 * 64;
    a.wcnt = reinterpret_cast<uint32_t *>(sc);
    a.gcand = reinterpret_cast<unsigned long long *>(sc + (size_t)SS_GROUPS * scans_collect_max() * SQ_QT * 4);
    a.out_d = dist; a.out_id = ids; a.dbg = g_scans_dbg;
    const unsigned g = (unsigned)((n_rows + a.rows_per_wg - 1) / a.rows_per_wg);
    a.G = (int)g;
    // the histogram pass: few workgroups (every bin they share costs a global atomic each: 256 of them on ~240 hot bins took 70 us),
    // each sampling every 4th chunk of a 1/32 slab
    const int64_t hg = std::min<int64_t>(32, (n_rows + 16383) / 16384);
    a.rows_per_hist_wg = ((n_rows + hg - 1) / hg + 255) / 256 * 256;
    const unsigned gh = (unsigned)((n_rows + a.rows_per_hist_wg - 1) / a.rows_per_hist_wg);
    if (codes_rot) {
        hipLaunchKernelGGL((scan16s_hist_kernel<true>), dim3(gh, ng), dim3(1024), 0, st, a);
        hipLaunchKernelGGL((scan16s_collect_kernel<true>), dim3(g, ng), dim3(1024), 0, st, a);
    } else {
        hipLaunchKernelGGL((scan16s_hist_kernel<false>), dim3(gh, ng), dim3(1024), 0, st, a);
        hipLaunchKernelGGL((scan16s_collect_kernel<false>), dim3(g, ng), dim3(1024), 0, st, a);
    }
    hipLaunchKernelGGL(scan16s_select_kernel, dim3((unsigned)nq), dim3(1024), 0, st, a);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// =====================================================================================================================
// k = 129 .. CVTMI_K_MAX through the filter scan (round 6; get_sort_results(score_total, num_show) takes any num_show:
// opq/src/common.h:25-37).  The checkpoint / spill protocols above keep k + band entries per query in 15 KB of LDS: k <= 128.  Beyond
// that the exact kernel answered with ONE query per workgroup and a 4096-entry selection buffer -- 11x slower at 10 000 x 1 M rows.
// The bound-first form of the small-batch path needs no selection while it scans, so it takes any k:
//   scan16h_prep_kernel      tables of every query group (fp32 + the quantised LDS image), as for adc_scan16h
//   scan16s_hist_kernel      every 4th chunk of the rows, all groups: one 256-bin histogram of the integer sums per query.  The first
//                            bin at which the cumulative count reaches k proves k rows below its edge: T = edge + slack admits every row
//                            that can be among the k best (~4 k rows pass)
//   scan16k_collect_kernel   all rows against T: the rows below it to per-(workgroup, query) lists in HBM (wave-aggregated LDS counter)
//   scan16k_select_kernel    one workgroup per query: k-th smallest integer sum of its list (two histogram passes) -> S_k; the rows
//                            below S_k + slack (k + a band) get exact reference-order sums (IVFOPQ.cpp:302-306) and are sorted by
//                            (distance, id) in LDS (bitonic, 4096 entries); the first k are the answer -- bit-identical to the
//                            reference's.  A list that overflowed, more than 4096 rows inside the band (masses of equal rows) or sums
//                            that bound nothing (non-finite tables) raise the query's flag: the exact kernel answers those queries.
// =====================================================================================================================
constexpr int SK_NSEL = 4096;   // entries the selection sorts (k <= 2048 + the band)

// histogram pass of the big-k pipeline: scan16s_hist_kernel's sample (every 4th chunk), two workgroups per CU, and a CUT-OFF -- as soon
// as the workgroup's own histogram proves k rows below a bin edge, rows above that edge cannot move the final bound (it can only lie at
// or below it): they are skipped by one packed compare per chunk instead of eight LDS atomics per row (the atomics were 60 % of the pass).
template <bool PREROT>
__global__ __launch_bounds__(1024, 8) void scan16k_hist_kernel(const ScanSArgs a)
{
    constexpr int NT = 1024, QT = SQ_QT;
    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * 16 * QT / 2];
    __shared__ __attribute__((aligned(16))) uint32_t hist[QT][SH_BINS];
    __shared__ uint32_t cut_pk[QT / 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = blockIdx.y;
    const uint4 *qlut = a.qlut + (size_t)grp * 4096;
    uint32_t *ctl = a.ctl + (size_t)grp * SS_WORDS;
    {
        uint4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = qlut[tid + i * NT];
#pragma unroll
        for (int i = 0; i < 4; ++i) reinterpret_cast<uint4 *>(lut)[tid + i * NT] = t[i];
    }
    for (int i = tid; i < QT * SH_BINS; i += NT) (&hist[0][0])[i] = 0;
    if (tid < QT / 2) cut_pk[tid] = 0x7fff7fffu;
    __syncthreads();
    const uint32_t c = tid & 15, cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_hist_wg;
    int64_t r1 = r0 + a.rows_per_hist_wg;
    r1 = r1 < a.n_rows ? r1 : a.n_rows;
    const uint4 *rows = reinterpret_cast<const uint4 *>(PREROT ? a.codes_rot : a.codes);
    const int64_t stride = (int64_t)(NT / 64) << (6 + SS_SAMPLE_SHIFT);
    uint32_t cpk[QT / 2];
#pragma unroll
    for (int i = 0; i < QT / 2; ++i) cpk[i] = 0x7fff7fffu;
    int round = 0;
    for (int64_t base = r0 + ((int64_t)wave << (6 + SS_SAMPLE_SHIFT)); base < r1; base += stride, ++round) {
        const int64_t row = base + lane;
        const uint4 v = rows[row < r1 ? row : r1 - 1];
        uint32_t sm[4];
        scan16q_row_sums<PREROT>(v, moffp, cr8, cq, lut_b, sm[0], sm[1], sm[2], sm[3]);
        uint32_t d4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d4[i] = pk_sub_i16(sm[i], cpk[i]);   // sign per 16-bit field: sum < cut-off
        const bool inr = row < r1;
        if (__ballot(inr && ((d4[0] | d4[1] | d4[2] | d4[3]) & 0x80008000u) != 0)) {  // wave-uniform
            if (inr) {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const bool below = (q & 1) ? (int32_t)d4[q >> 1] < 0 : (d4[q >> 1] & 0x8000u) != 0;
                    if (below) atomicAdd(&hist[q][((sm[q >> 1] >> (16 * (q & 1))) & 0xffffu) >> 7], 1u);
                }
            }
        }
        if ((round & 15) == 15) {   // every 16 sampled chunks per wave (1024 rows): the cut-offs the workgroup's counts allow so far
            if (wave < QT) {        // (no barrier: counts only grow, so a snapshot's bound is valid; a stale cut-off is only looser)
                const uint32_t t = scanh_hist_bound(hist[wave], a.k, 0u);   // edge of the bin that holds the k-th of the rows counted so far
                if (lane == 0 && t < 32767u) reinterpret_cast<uint16_t *>(cut_pk)[wave] = (uint16_t)t;
            }
#pragma unroll
            for (int i = 0; i < QT / 2; ++i) cpk[i] = __hip_atomic_load(&cut_pk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    for (int i = tid; i < QT * SH_BINS; i += NT) {
        const uint32_t v = (&hist[0][0])[i];
        if (v) atomicAdd(&ctl[i], v);
    }
}

template <bool PREROT>
__global__ __launch_bounds__(1024, 8) void scan16k_collect_kernel(const ScanSArgs a, int wcap)
{
    constexpr int NT = 1024, QT = SQ_QT;
    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * 16 * QT / 2];
    __shared__ __attribute__((aligned(16))) uint32_t hist[QT][SH_BINS];
    __shared__ QuantParams qp;
    __shared__ uint32_t T[QT], tpk_s[QT / 2];
    __shared__ uint32_t cnt[QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = blockIdx.y;
    const uint4 *qlut = a.qlut + (size_t)grp * 4096;
    uint32_t *ctl = a.ctl + (size_t)grp * SS_WORDS;
    {
        uint4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = qlut[tid + i * NT];
#pragma unroll
        for (int i = 0; i < 4; ++i) reinterpret_cast<uint4 *>(lut)[tid + i * NT] = t[i];
    }
    if (tid < (int)(sizeof(QuantParams) / 4)) reinterpret_cast<uint32_t *>(&qp)[tid] = reinterpret_cast<const uint32_t *>(a.qp_g + grp)[tid];
    for (int i = tid; i < QT * SH_BINS; i += NT) (&hist[0][0])[i] = ctl[i];
    if (tid < QT) cnt[tid] = 0;
    __syncthreads();
    if (wave < QT) {  // the bound of query `wave` from the global histogram (a sample: every 4th chunk of every row block)
        const uint32_t sl = qp.slack[wave];
        uint32_t t = sl ? scanh_hist_bound(hist[wave], a.k, sl) : 0xffffffffu;
        if (t > 32767u) t = 32767u;  // fewer than k rows sampled, or sums that bound nothing: the exact kernel answers (select raises the flag)
        if (lane == 0) {
            T[wave] = t;
            if (blockIdx.x == 0) ctl[QT * SH_BINS + wave] = t;
        }
    }
    __syncthreads();
    if (tid < QT / 2) tpk_s[tid] = T[2 * tid] | (T[2 * tid + 1] << 16);
    __syncthreads();
    bool any_open = false;   // a query without a bound would send every row to its list: nothing is collected for it
#pragma unroll
    for (int q = 0; q < QT; ++q) any_open |= T[q] >= 32767u;
    const uint32_t c = tid & 15, cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);
    uint32_t tpk[QT / 2];
#pragma unroll
    for (int i = 0; i < QT / 2; ++i) {
        uint32_t t2 = tpk_s[i];
        if (any_open) {   // (rare) open queries compare against 0: no candidate
            if ((t2 & 0xffffu) >= 32767u) t2 &= 0xffff0000u;
            if ((t2 >> 16) >= 32767u) t2 &= 0x0000ffffu;
        }
        tpk[i] = t2;
    }
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_wg;
    int64_t r1 = r0 + a.rows_per_wg;
    r1 = r1 < a.n_rows ? r1 : a.n_rows;
    const uint4 *rows = reinterpret_cast<const uint4 *>(PREROT ? a.codes_rot : a.codes);
    unsigned long long *mine = a.gcand + ((size_t)grp * a.G + blockIdx.x) * QT * (size_t)wcap;
    uint4 v = rows[r0 + ((int64_t)wave << 6) + lane < r1 ? r0 + ((int64_t)wave << 6) + lane : (r1 > 0 ? r1 - 1 : 0)];
    for (int64_t base = r0 + ((int64_t)wave << 6); base < r1; base += NT) {
        const int64_t row = base + lane, nrow = row + NT;
        const uint4 vn = rows[nrow < r1 ? nrow : r1 - 1];   // next chunk of this wave, in flight during the look-ups
        uint32_t s4[4];
        scan16q_row_sums<PREROT, 16>(v, moffp, cr8, cq, lut_b, s4[0], s4[1], s4[2], s4[3]);
        v = vn;
        uint32_t d4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d4[i] = pk_sub_i16(s4[i], tpk[i]);
        const bool inr = row < r1;
        if (__ballot(inr && ((d4[0] | d4[1] | d4[2] | d4[3]) & 0x80008000u) != 0) == 0) continue;  // wave-uniform
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const bool cand = inr && ((q & 1) ? (int32_t)d4[q >> 1] < 0 : (d4[q >> 1] & 0x8000u) != 0);
            const unsigned long long m = __ballot(cand);
            if (!m) continue;  // scalar
            uint32_t basep = 0;
            if (lane == 0) basep = atomicAdd(&cnt[q], (uint32_t)__popcll(m));   // LDS
            basep = (uint32_t)__builtin_amdgcn_readfirstlane((int)basep);
            const uint32_t pos = basep + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (cand && pos < (uint32_t)wcap) {
                const uint32_t sq = (q & 1) ? s4[q >> 1] >> 16 : s4[q >> 1] & 0xffffu;
                mine[(size_t)q * wcap + pos] = ((unsigned long long)sq << 32) | (uint32_t)row;
            }
        }
    }
    __syncthreads();
    if (tid < QT) a.wcnt[((size_t)grp * a.G + blockIdx.x) * QT + tid] = cnt[tid];
}

// ascending bitonic sort of n (a power of two, <= SK_NSEL) 64-bit keys in LDS by the whole workgroup
template <int NT>
__device__ __forceinline__ void block_bitonic_u64(unsigned long long *e, int n)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n / 2; i += NT) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;   // lo = (i / stride) * 2 stride + i % stride
                const bool up = (lo & size) == 0;
                const unsigned long long a0 = e[lo], a1 = e[hi];
                if ((a0 > a1) == up) { e[lo] = a1; e[hi] = a0; }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void scan16k_select_kernel(const ScanSArgs a, int wcap, uint32_t *__restrict__ flags)
{
    constexpr int NT = 1024, QT = SQ_QT;
    __shared__ __attribute__((aligned(16))) unsigned long long sel[SK_NSEL];
    __shared__ __attribute__((aligned(16))) uint32_t h[SH_BINS];
    __shared__ QuantParams qp;
    __shared__ uint32_t off[64 + 1];
    __shared__ uint32_t s_n, s_bin, s_cum, s_sk;
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, q = blockIdx.x, grp = q / QT, ql = q % QT;
    if (tid < (int)(sizeof(QuantParams) / 4)) reinterpret_cast<uint32_t *>(&qp)[tid] = reinterpret_cast<const uint32_t *>(a.qp_g + grp)[tid];
    if (tid == 0) { s_bad = 0; s_n = 0; }
    for (int i = tid; i < SH_BINS; i += NT) h[i] = 0;
    __syncthreads();
    if (tid < a.G) {   // G <= 64 collect workgroups per group
        const uint32_t c = a.wcnt[((size_t)grp * a.G + tid) * QT + ql];
        off[tid] = c;
        if (c > (uint32_t)wcap) s_bad = 1;   // that workgroup dropped candidates
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int g = 0; g < a.G; ++g) { const uint32_t v = off[g]; off[g] = run; run += v; }
        off[a.G] = run;
    }
    __syncthreads();
    const uint32_t n = off[a.G];
    const uint32_t Tq = a.ctl[(size_t)grp * SS_WORDS + QT * SH_BINS + ql];
    const uint32_t slack = qp.slack[ql];
    if (s_bad || Tq >= 32767u || !slack || n < (uint32_t)(a.k < a.n_rows ? a.k : a.n_rows)) {   // workgroup-uniform: the exact kernel answers this query
        if (tid == 0) flags[q] = 1u;
        return;
    }
    const auto list_of = [&](int g) -> const unsigned long long * { return a.gcand + (((size_t)grp * a.G + g) * QT + ql) * (size_t)wcap; };
    // pass 1: bins of 128 units
    for (int g = 0; g < a.G; ++g) {
        const unsigned long long *src = list_of(g);
        const uint32_t cg = off[g + 1] - off[g];
        for (uint32_t i = tid; i < cg; i += NT) atomicAdd(&h[(uint32_t)(src[i] >> 32) >> 7], 1u);
    }
    __syncthreads();
    if (tid < 64) {
        const uint4 cb = *reinterpret_cast<const uint4 *>(h + lane * 4);
        const uint32_t mine = cb.x + cb.y + cb.z + cb.w, incl = wave_incl_scan_add(mine);
        const unsigned long long reach = __ballot(incl >= (uint32_t)a.k);
        const int l0 = reach ? __ffsll((long long)reach) - 1 : 63;
        uint32_t cum = (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), l0);
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cb.x, l0), c1 = (uint32_t)__builtin_amdgcn_readlane((int)cb.y, l0),
                       c2 = (uint32_t)__builtin_amdgcn_readlane((int)cb.z, l0);
        uint32_t b = (uint32_t)l0 * 4u;
        if (cum + c0 < (uint32_t)a.k) { cum += c0; ++b; if (cum + c1 < (uint32_t)a.k) { cum += c1; ++b; if (cum + c2 < (uint32_t)a.k) { cum += c2; ++b; } } }
        if (lane == 0) { s_bin = b; s_cum = cum; }   // cum rows lie below bin b; the k-th is the (k - cum)-th smallest inside it
    }
    __syncthreads();
    for (int i = tid; i < 128; i += NT) h[i] = 0;
    __syncthreads();
    const uint32_t bin = s_bin;
    for (int g = 0; g < a.G; ++g) {   // pass 2: position inside the bin
        const unsigned long long *src = list_of(g);
        const uint32_t cg = off[g + 1] - off[g];
        for (uint32_t i = tid; i < cg; i += NT) {
            const uint32_t sv = (uint32_t)(src[i] >> 32);
            if ((sv >> 7) == bin) atomicAdd(&h[sv & 127u], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {
        const uint32_t f0 = h[lane * 2], f1 = h[lane * 2 + 1];
        const uint32_t inc2 = wave_incl_scan_add(f0 + f1);
        const uint32_t need = (uint32_t)a.k - s_cum;
        const unsigned long long r2 = __ballot(inc2 >= need);
        const int l2 = r2 ? __ffsll((long long)r2) - 1 : 63;
        const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)(inc2 - f0 - f1), l2);
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)f0, l2);
        if (lane == 0) s_sk = (bin << 7) + (uint32_t)l2 * 2u + (before + g0 >= need ? 0u : 1u);
    }
    __syncthreads();
    uint32_t Tsel = s_sk + slack;
    Tsel = Tsel < Tq ? Tsel : Tq;
    // the rows inside the band -> LDS (order is irrelevant: the sort decides)
    for (int g = 0; g < a.G; ++g) {
        const unsigned long long *src = list_of(g);
        const uint32_t cg = off[g + 1] - off[g];
        for (uint32_t i0 = 0; i0 < cg; i0 += NT) {
            const uint32_t i = i0 + tid;
            const unsigned long long e = i < cg ? src[i] : ~0ull;
            const bool in = i < cg && (uint32_t)(e >> 32) < Tsel;
            const unsigned long long m = __ballot(in);
            uint32_t basep = 0;
            if (lane == 0 && m) basep = atomicAdd(&s_n, (uint32_t)__popcll(m));
            basep = (uint32_t)__builtin_amdgcn_readfirstlane((int)basep);
            const uint32_t pos = basep + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (in && pos < (uint32_t)SK_NSEL) sel[pos] = e;
        }
    }
    __syncthreads();
    const uint32_t ns = s_n;
    if (ns > (uint32_t)SK_NSEL) {   // a crowded band (masses of equal rows): the exact kernel
        if (tid == 0) flags[q] = 1u;
        return;
    }
    const ExactFromLut fix{ reinterpret_cast<const uint4 *>(a.codes), a.lut_g, a.K, a.nq, grp };
    int np = 256;
    while (np < (int)ns) np <<= 1;
    for (int i = tid; i < np; i += NT) sel[i] = (uint32_t)i < ns ? fix(ql, sel[i]) : ~0ull;   // exact (distance bits, row); padding sorts last
    block_bitonic_u64<NT>(sel, np);
    for (int i = tid; i < a.k; i += NT) {
        if ((uint32_t)i < ns) {
            const unsigned long long e = sel[i];
            a.out_d[(int64_t)q * a.k + i] = __uint_as_float((uint32_t)(e >> 32));
            a.out_id[(int64_t)q * a.k + i] = a.id_base + (int64_t)(uint32_t)e;
        } else {
            a.out_d[(int64_t)q * a.k + i] = __uint_as_float(0x7f800000u);
            a.out_id[(int64_t)q * a.k + i] = -1;
        }
    }
}

// collect workgroups per query group and list capacity per (workgroup, query) of a big-k search
static void scank_shape(int64_t n_rows, int64_t nq, int k, int *G, int *wcap)
{
    const int64_t groups = (nq + SQ_QT - 1) / SQ_QT;
    int64_t g = (scanh_slots() / 2 + groups - 1) / groups;            // enough workgroups for one per CU
    g = std::max<int64_t>(1, std::min<int64_t>(g, std::min<int64_t>(64, (n_rows + 16383) / 16384)));
    *G = (int)g;
    *wcap = (int)((8LL * k + g - 1) / g + 1024);   // ~4 k rows pass the sampled bound in all; twice that plus a margin, spread over the workgroups
}
bool scank_applies(const OpqModelDev &m, int64_t n_rows, int64_t nq, int k)
{
    return m.M == 16 && m.D <= 256 && m.K >= 1 && m.K <= 256 && nq >= 1 && k > 128 && k <= 2048 && n_rows >= 65536 && n_rows <= 0xfffffffeLL &&
           (int64_t)k * 8 <= n_rows;
}
size_t scank_scratch_bytes(int64_t n_rows, int64_t nq, int k)
{
    int G, wcap;
    scank_shape(n_rows, nq, k, &G, &wcap);
    const size_t groups = (size_t)((nq + SQ_QT - 1) / SQ_QT);
    return (groups * SS_WORDS * 4 + 63) / 64 * 64 + (groups * G * SQ_QT * 4 + 63) / 64 * 64 + ((size_t)nq * 4 + 63) / 64 * 64 +
           groups * G * SQ_QT * (size_t)wcap * sizeof(unsigned long long);
}
// q_rot: rotated queries.  flags_out: [nq] words, 1 = the query was NOT answered (the caller runs the exact kernel for those)
int launch_adc_scan_bigk(const OpqModelDev &m, const uint8_t *codes, const uint8_t *codes_rot, int64_t n_rows, int64_t id_base, const float *q_rot,
                         int64_t nq, int k, float *dist, int64_t *ids, float *lut_g, void *qlut, void *qp_g, void *scratch, int lazy,
                         uint32_t **flags_out, hipStream_t st)
{
    if (!scank_applies(m, n_rows, nq, k)) return fail(CVTMI_EUNSUPPORTED, "adc_scan16k: shape not covered");
    int G, wcap;
    scank_shape(n_rows, nq, k, &G, &wcap);
    const size_t groups = (size_t)((nq + SQ_QT - 1) / SQ_QT);
    if (groups > 65535) return fail(CVTMI_EUNSUPPORTED, "adc_scan16k: more than 65535 query groups per call");
    char *sc = reinterpret_cast<char *>(scratch);
    uint32_t *ctl = reinterpret_cast<uint32_t *>(sc);
    sc += (groups * SS_WORDS * 4 + 63) / 64 * 64;
    ScanSArgs a;
    a.wcnt = reinterpret_cast<uint32_t *>(sc);
    sc += (groups * G * SQ_QT * 4 + 63) / 64 * 64;
    uint32_t *flags = reinterpret_cast<uint32_t *>(sc);
    sc += ((size_t)nq * 4 + 63) / 64 * 64;
    a.gcand = reinterpret_cast<unsigned long long *>(sc);
    CVTMI_HIP(hipMemsetAsync(flags, 0, (size_t)nq * 4, st));
    hipLaunchKernelGGL(scan16h_prep_kernel, dim3((unsigned)groups), dim3(1024), 0, st, q_rot, (int)nq, m.D, m.step, m.K, m.books, m.coarse, lut_g,
                       reinterpret_cast<uint4 *>(qlut), reinterpret_cast<QuantParams *>(qp_g), lazy, (const float *)nullptr, (const int32_t *)nullptr,
                       ctl, SS_WORDS);
    CVTMI_HIP(hipGetLastError());
    a.codes = codes; a.codes_rot = codes_rot; a.n_rows = n_rows; a.id_base = id_base;
    a.rows_per_wg = ((n_rows + G - 1) / G + 255) / 256 * 256;   // whole groups of four chunks: the histogram's sample
    a.rows_per_hist_wg = a.rows_per_wg;
    a.nq = (int)nq; a.k = k; a.K = m.K; a.G = (int)((n_rows + a.rows_per_wg - 1) / a.rows_per_wg);
    a.qlut = reinterpret_cast<const uint4 *>(qlut); a.qp_g = reinterpret_cast<const QuantParams *>(qp_g); a.lut_g = lut_g;
    a.ctl = ctl; a.out_d = dist; a.out_id = ids; a.dbg = 0;
    const dim3 grid((unsigned)a.G, (unsigned)groups);
    if (codes_rot) {
        hipLaunchKernelGGL((scan16k_hist_kernel<true>), grid, dim3(1024), 0, st, a);
        hipLaunchKernelGGL((scan16k_collect_kernel<true>), grid, dim3(1024), 0, st, a, wcap);
    } else {
        hipLaunchKernelGGL((scan16k_hist_kernel<false>), grid, dim3(1024), 0, st, a);
        hipLaunchKernelGGL((scan16k_collect_kernel<false>), grid, dim3(1024), 0, st, a, wcap);
    }
    hipLaunchKernelGGL(scan16k_select_kernel, dim3((unsigned)nq), dim3(1024), 0, st, a, wcap, flags);
    CVTMI_HIP(hipGetLastError());
    *flags_out = flags;
    return CVTMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side: the item table
// ---------------------------------------------------------------------------------------------------------------------
static int g_scanh_share_hist = 1;   // cvtmi_set_tuning("scanh_share_hist"): the segments of a query count their candidates in one histogram (bounds from the union)
void set_scanh_share_hist(int v) { g_scanh_share_hist = v != 0; }
size_t scanh_gthr_bytes(int64_t nq) { return ((size_t)(nq + 3) / 4 * 4 + (size_t)nq * SH_BINS) * sizeof(uint32_t); }
static double g_scanh_fix = 160000.0;   // cvtmi_set_tuning("scanh_fix"): what an item of a group kept whole costs besides its rows, in row-equivalents (planner)
void set_scanh_fix(double v) { g_scanh_fix = v > 0 ? v : 160000.0; }
static int g_scanh_balance = 0;       // 0 = choose, 1 = equal shares of the flat (group x row) space, 2 = (group, split) blocks
static int64_t g_scanh_min_rows = 16384;  // smallest share of a workgroup in the balanced plan
static int g_scanh_tail = 0;              // (group, split) blocks: the groups of the last, partly filled round may be cut finer (measured: no gain, off)
void set_scanh_tail(int v) { g_scanh_tail = v != 0; }
void set_scanh_balance(int v) { g_scanh_balance = v; }
void set_scanh_min_rows(int64_t v) { g_scanh_min_rows = v < 2048 ? 2048 : v; }

// Workgroup slots of the device: a constant of the process once read (every grid and scratch size of a search derives from this ONE
// value; cvtmi_opq_scan_plan plans for another CU count through scanh_plan's own parameter, never through shared state).
static int scanh_slots()
{
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
void scanh_plan(int64_t n_rows, int64_t nq, int splits, ScanHPlan &p, int cus)
{
    constexpr int64_t TILE = 2048, MINT = 4, MAX_SEG = (1LL << 28) - 4096;  // rows: segment granule; 32-bit byte offsets inside a segment
    const int64_t groups = (nq + SQ_QT - 1) / SQ_QT;
    const int64_t slots = cus > 0 ? 2 * (int64_t)cus : scanh_slots();   // cus > 0: cvtmi_opq_scan_plan planning for a given CU count
    p.items.clear();
    p.multi.clear();
    p.grid = 0; p.rounds = 0; p.stride = 1;
    if (groups <= 0 || n_rows <= 0) return;
    const bool resident = n_rows * 16 <= (96LL << 20);  // the pre-rotated rows stay in the Infinity Cache (and mostly in L2)
    const bool balanced = splits <= 0 && n_rows <= MAX_SEG && g_scanh_balance == 1;   // (equal shares: only on request since the segments share their histograms -- row splits measure better, tools/sweep_scan_h.py)
    (void)resident;
    // (planner's own choice: equal shares when S = 1 would fill the slots between half and whole -- 1 M rows x 2500 queries: 1.02-1.07 ms
    //  against 1.24 for whole groups and 1.18 for adc_scan16q; with more groups than slots the shares cost an item more per workgroup
    //  than blocks do and measured 2-10 % behind them, tools/sweep_scan_h.py)
    struct Seg { int64_t group, row0, rows; int wg; int64_t chunk0; };
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
        // Shares of at least one group (the shape the planner itself picks: as many groups as slots, or more) are laid out in "row time":
        // a workgroup that has worked through t tiles of its share is at tile t mod tg of whatever group it is in -- a whole group is walked
        // from that phase on, wrapping around; a group cut by a share boundary gives its FIRST rows to the workgroup that starts with it
        // (time 0) and its last rows to the one that ends with it.  So the workgroups of an XCD read the same rows at the same time.
        const bool row_time = total / P >= tg;
        for (int64_t w = 0; w < P; ++w) {
            const int64_t begin = bound(w), end = w + 1 == P ? total : bound(w + 1);
            int64_t u = begin;
            while (u < end) {
                const int64_t g = u / tg, t0 = u - g * tg, t1 = std::min(tg, t0 + (end - u));
                int64_t a = t0, b = t1, c0 = 0;
                if (row_time) {
                    if (t0 == 0 && t1 == tg) c0 = ((u - begin) % tg) * (TILE / 64);   // whole group: start at the share's row time
                    else if (t0 > 0) { a = 0; b = t1 - t0; }                            // the group began in the previous share: its first rows
                    else { a = tg - t1; b = tg; }                                       // the group goes on in the next share: its last rows
                }
                const int64_t r0 = a * TILE, r1 = std::min(n_rows, b * TILE);
                if (r1 > r0) segs.push_back({ g, r0, r1 - r0, (int)w, c0 });
                u += t1 - t0;
            }
        }
        p.grid = (int)P;
    } else {
        int64_t S = splits > 0 ? splits : 1;
        const int64_t min_splits = (n_rows + MAX_SEG - 1) / MAX_SEG;
        if (splits <= 0) {
            // Cost of cutting every group into `cand` row splits = the makespan of the persistent grid: workgroup w walks items w, w + P, ...;
            // CU c hosts workgroups c and c + slots / 2; an item costs its rows plus what it costs besides them (tables, seed, the candidates
            // of its warm-up -- which the segments of a group now share through one histogram, so that part falls with the split count --,
            // the final selection); while both workgroups of a CU are busy each runs at the shared look-up rate, the one that is left runs
            // 1.65 x faster.  Constants fitted to tools/sweep_scan_h.py at 1 M rows (nq = 600 .. 5000, 1 .. 4 splits: the model picks the
            // measured best everywhere but at nq = 2500, where it is 2 % off).  Round 3's model -- rounds x (0.17 M + rows) with a discount
            // for a half-empty last round -- kept groups whole where two or four splits measure 15-20 % faster (nq = 1500, 2500, 3000).
            const auto makespan = [&](int64_t cand) {
                const double W = g_scanh_fix * (0.5 + 0.5 / (double)cand) + (double)n_rows / (double)cand;
                const int64_t items = groups * cand, P = std::min<int64_t>(slots, items), cus = std::max<int64_t>(1, slots / 2);
                const auto cnt = [&](int64_t w) -> int64_t { return w < P ? items / P + (w < items % P ? 1 : 0) : 0; };
                double T = 0.0;
                for (int64_t c = 0; c < cus; ++c) {
                    const int64_t x = cnt(c), y = cnt(c + cus), lo = std::min(x, y), hi = std::max(x, y);
                    T = std::max(T, (double)lo * W + (double)(hi - lo) * W / 1.65);
                }
                return T;
            };
            double best_cost = 1e300;
            S = min_splits;
            for (int64_t cand = min_splits; cand <= 64; ++cand) {
                if (cand > min_splits && n_rows / cand < 16384) break;
                const double cost = makespan(cand);
                if (cost < best_cost * 0.99) { best_cost = cost; S = cand; }
            }
        }
        if (S < min_splits) S = min_splits;
        // blocks of `cnt` groups from group g0 on, each cut into Sx row splits, appended in the order a grid of blocks would run them
        const auto add_blocks = [&](int64_t g0, int64_t cnt, int64_t Sx, int64_t first_slot, int64_t P) {
            int64_t rps = (n_rows + Sx - 1) / Sx;
            rps = ((rps + TILE - 1) / TILE) * TILE;
            for (int64_t b = 0; b < cnt * Sx; ++b) {
                int64_t split, group;
                if ((Sx & 7) == 0) {
                    const int64_t s8 = Sx >> 3, xcd = b & 7, i = b >> 3;
                    split = xcd + 8 * (i % s8);
                    group = i / s8;
                } else {
                    split = b % Sx;
                    group = b / Sx;
                }
                const int64_t r0 = split * rps, r1 = std::min(n_rows, r0 + rps);
                // (an empty split still reports: its slot of the partial lists must be written)
                segs.push_back({ g0 + group, std::min(r0, n_rows), r1 > r0 ? r1 - r0 : 0, (int)((first_slot + b) % P), 0 });
            }
        };
        const int64_t P = std::min<int64_t>(slots, groups * S);
        // Two regions (planner's choice only, whole groups in the first): the groups that fill whole rounds of slots stay whole, those
        // of the last, partly filled round are cut finer, so that every CU keeps two workgroups until the end instead of running the
        // tail at half its look-up rate.  (adc_scan16q's planner has the same option and rarely takes it: there a split costs 0.38 M
        // row-equivalents.)
        int64_t ga = groups, Sb = S;
        if (splits <= 0 && S == 1 && groups > slots && groups % slots != 0 && g_scanh_tail) {
            const int64_t rem = groups % slots;
            const double fix = g_scanh_fix;
            double best = 1e300;
            for (int64_t cand = 1; cand <= 32 && n_rows / cand >= 16384; ++cand) {
                const int64_t blocks = rem * cand;
                const double rounds = (double)((blocks + slots - 1) / slots);
                const double solo = blocks <= slots / 2 ? 0.78 : 1.0;   // one workgroup per CU runs 1.27x faster
                const double cost = rounds * solo * (fix + (double)n_rows / (double)cand);
                if (cost < best * 0.98) { best = cost; Sb = cand; }
            }
            if (Sb > 1) ga = groups - rem;
        }
        add_blocks(0, ga, S, 0, P);
        if (ga < groups) add_blocks(ga, groups - ga, Sb, ga * S, P);
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
    p.items.assign((size_t)rounds * p.grid, ScanItem{ 0, 0, 0, 0, 0, 0, 0 });
    std::fill(per_wg.begin(), per_wg.end(), 0);
    for (size_t i = 0; i < segs.size(); ++i) {
        const Seg &s = segs[i];
        ScanItem it;
        it.group = (int32_t)s.group;
        it.row0_64 = (uint32_t)(s.row0 / 64);
        it.rows = (uint32_t)s.rows;
        it.chunk0 = (uint32_t)s.chunk0;
        it.pad = 0;
        it.sidx = (uint16_t)sidx[i];
        it.nseg = (uint16_t)nseg[(size_t)s.group];
        p.items[(size_t)per_wg[(size_t)s.wg]++ * p.grid + s.wg] = it;
    }
    p.rounds = rounds;
    p.stride = stride;
    p.multi.assign((size_t)nq, 0u);
    for (int64_t qi = 0; qi < nq; ++qi) p.multi[(size_t)qi] = nseg[(size_t)(qi / SQ_QT)] > 1 ? 1u : 0u;
}

size_t scanh_spill_bytes(int grid) { return (size_t)grid * SQ_QT * SH_CAPG * sizeof(unsigned long long); }
size_t scanh_qlut_bytes(int64_t nq) { return (size_t)((nq + SQ_QT - 1) / SQ_QT) * 4096 * sizeof(uint4); }
size_t scanh_qp_bytes(int64_t nq) { return (size_t)((nq + SQ_QT - 1) / SQ_QT) * sizeof(QuantParams); }

int launch_adc_scan_h(const OpqModelDev &m, const uint8_t *codes, const uint8_t *codes_rot, int64_t n_rows, int64_t id_base,
                      const float *q_rot, int64_t nq, int k, const ScanHPlan &plan, const ScanItem *items_dev, float *part_d,
                      int64_t *part_id, float *out_d, int64_t *out_id, float *lut_g, void *qlut, void *qp_g, void *spill, uint32_t *gthr, int lazy, int seed,
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
    uint32_t *ghist = (gthr && plan.stride > 1 && g_scanh_share_hist) ? gthr + (nq + 3) / 4 * 4 : nullptr;
    if (ghist) CVTMI_HIP(hipMemsetAsync(ghist, 0, (size_t)nq * SH_BINS * sizeof(uint32_t), st));
    ScanHArgs a;
    a.ghist = ghist;
    a.codes = codes; a.codes_rot = codes_rot; a.id_base = id_base; a.nq = (int)nq; a.k = k; a.K = m.K;
    a.items = items_dev; a.rounds = plan.rounds;
    a.qlut = reinterpret_cast<const uint4 *>(qlut); a.qp_g = reinterpret_cast<const QuantParams *>(qp_g); a.lut_g = lut_g;
    a.spill = reinterpret_cast<unsigned long long *>(spill);
    a.gthr = (gthr && plan.stride > 1) ? gthr : nullptr;
    a.stride = plan.stride; a.part_d = part_d; a.part_id = part_id; a.out_d = out_d; a.out_id = out_id; a.seed = seed;
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
    scanh_plan(n_rows, nq, splits, p, cus);
    if (grid) *grid = p.grid;
    if (rounds) *rounds = p.rounds;
    if (stride) *stride = p.stride;
    const int64_t n = (int64_t)p.items.size();
    for (int64_t i = 0; i < n && i < cap; ++i) {
        const ScanItem &it = p.items[(size_t)i];
        items[6 * i + 0] = it.group; items[6 * i + 1] = it.row0_64; items[6 * i + 2] = it.rows; items[6 * i + 3] = it.sidx; items[6 * i + 4] = it.nseg;
        items[6 * i + 5] = it.chunk0;
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
extern "C" int cvtmi_debug_prep_timing(unsigned long long *out, int reset)
{
    unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prep_dbg), sizeof z) != hipSuccess) return -3;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_prep_dbg), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
extern "C" int cvtmi_debug_scanh_counters(unsigned long long *out, int reset)
{
    unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scanh_cnt), sizeof z) != hipSuccess) return -3;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_scanh_cnt), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
#endif

}  // namespace cvtmi
