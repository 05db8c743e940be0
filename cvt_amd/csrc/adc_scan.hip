// adc_scan.hip -- the bandwidth-/LDS-bound core: per-query distance tables in LDS, packed uint8
// code rows streamed with 16-byte loads, k smallest (distance, id) kept per workgroup.
//
// Reference arithmetic being reproduced (opq/src/IVFOPQ.cpp):
//   LUT   :273-291  PQ_table[m][j] = sum_k (res[m*step+k] - book[m][j][k])^2, fp32, k ascending,
//                   separate multiply and add; res = q - coarse[vw] (:273-276)
//   scan  :300-306  score = sum_{m<M} PQ_table[m][code[m]], fp32, m ascending, from 0.0f
//   top-k common.h:25-37  k smallest (score, id)
//
// MI355X mapping
//   * one 256-thread workgroup serves QT queries ("query tile") over one row split; the QT tables
//     are built straight into LDS, interleaved by query: lut[m][j][q], so ONE ds_read_b{32,64,128}
//     per code byte returns the table entries of all QT queries (QT=4 -> ds_read_b128, 64 KB).
//   * each lane owns R code rows per tile; a row (M=16) is one global_load_dwordx4, 64 lanes read
//     1 KiB contiguous; the next tile's rows are in flight while the current one is looked up.
//   * per-row sums stay in registers in the reference's order (bit-exact); candidates below the
//     running k-th distance go to the LDS selection buffer (block_topk.h).
//   * block -> (query group, row split) mapping keeps a row split on one XCD (block b runs on XCD
//     b % 8) so each XCD's 4 MiB L2 only ever caches 1/8 of the code matrix.
#include <algorithm>
#include <atomic>

#include "adc_scan16.h"

namespace cvtmi {

constexpr int SCAN_CAP = 384;   // selection buffer entries per query
constexpr int SCAN_TRIG = 256;  // compaction is requested beyond this fill
// code rows per lane per tile: 4 where the register file allows it, 2 for the widest variants
__host__ __device__ constexpr int scan_rows(int M, int QT) { return (M * QT > 32) ? 2 : 4; }


template <int M> struct CodeRow;
template <> struct CodeRow<16> { using type = uint4; };
template <> struct CodeRow<8> { using type = uint2; };
template <> struct CodeRow<4> { using type = uint32_t; };

template <int M>
__device__ __forceinline__ uint32_t code_byte(const typename CodeRow<M>::type &c, int m);
template <>
__device__ __forceinline__ uint32_t code_byte<16>(const uint4 &c, int m)
{
    const uint32_t w = (m < 4) ? c.x : (m < 8) ? c.y : (m < 12) ? c.z : c.w;
    return (w >> (8 * (m & 3))) & 0xffu;
}
template <>
__device__ __forceinline__ uint32_t code_byte<8>(const uint2 &c, int m)
{
    const uint32_t w = (m < 4) ? c.x : c.y;
    return (w >> (8 * (m & 3))) & 0xffu;
}
template <>
__device__ __forceinline__ uint32_t code_byte<4>(const uint32_t &c, int m)
{
    return (c >> (8 * m)) & 0xffu;
}

template <int QT> struct LutVec;
template <> struct LutVec<1> { using type = float; };
template <> struct LutVec<2> { using type = float2; };
template <> struct LutVec<4> { using type = float4; };
template <> struct LutVec<8> { struct alignas(32) type { float4 a, b; }; };

template <int QT>
__device__ __forceinline__ void lut_get(const float *lut, uint32_t entry, float (&v)[QT])
{
    // entry = m*256 + code; table is [entry][QT] floats -> one aligned vector LDS read
    if constexpr (QT == 1) {
        v[0] = lut[entry];
    } else if constexpr (QT == 2) {
        const float2 t = reinterpret_cast<const float2 *>(lut)[entry];
        v[0] = t.x; v[1] = t.y;
    } else if constexpr (QT == 4) {
        const float4 t = reinterpret_cast<const float4 *>(lut)[entry];
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const float4 t = reinterpret_cast<const float4 *>(lut)[entry * 2];
        const float4 u = reinterpret_cast<const float4 *>(lut)[entry * 2 + 1];
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        v[4] = u.x; v[5] = u.y; v[6] = u.z; v[7] = u.w;
    }
}

// Build the QT interleaved tables into LDS.  `res` is an LDS scratch of QT*D floats.
template <int M, int QT>
__device__ __forceinline__ void build_lut_lds(float *lut, float *res, const ScanArgs &a, int group)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < QT * a.D; i += kBlock) {
        const int q = i / a.D, d = i - q * a.D;
        int qi = group * QT + q;
        qi = qi < a.nq ? qi : a.nq - 1;  // ragged last group: repeat the last query, output is skipped
        res[i] = __fsub_rn(a.q_rot[(int64_t)qi * a.D + d], a.centroid[d]);
    }
    __syncthreads();
    const int j = tid;  // one centroid per thread (K <= 256)
#pragma unroll 1
    for (int m = 0; m < M; ++m) {
        float acc[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = 0.0f;
        if (j < a.K) {
            const float *cb = a.books + ((int64_t)m * a.K + j) * a.step;
            for (int kk = 0; kk < a.step; ++kk) {
                const float c = cb[kk];
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const float t = __fsub_rn(res[q * a.D + m * a.step + kk], c);
                    acc[q] = __fadd_rn(acc[q], __fmul_rn(t, t));
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[q] = __uint_as_float(0x7f800000u);  // code >= K: never a match
        }
#pragma unroll
        for (int q = 0; q < QT; ++q) lut[(m * 256 + j) * QT + q] = acc[q];
    }
    __syncthreads();
}

// waves per SIMD the register allocator must leave room for: the LDS footprint admits 8 / 4 / 2 / 1
// workgroups per CU at QT = 1 / 2 / 4 / 8 (M = 16), i.e. that many waves per SIMD.
template <int QT> struct ScanOcc { static constexpr int waves = QT == 1 ? 6 : QT == 2 ? 3 : QT == 4 ? 2 : 1; };

template <int M, int QT, int CAP = SCAN_CAP, int TRIG = SCAN_TRIG>
__global__ __launch_bounds__(kBlock, CAP > SCAN_CAP ? 1 : ScanOcc<QT>::waves) void adc_scan_kernel(const ScanArgs a)
{
    using Row = typename CodeRow<M>::type;
    constexpr int R = scan_rows(M, QT);
    __shared__ __attribute__((aligned(32))) float lut[M * 256 * QT];
    __shared__ TopKShared<QT, CAP> tk;

    // ---- block -> (query group, row split); a row split stays on one XCD when splits % 8 == 0 ----
    int group, split;
    {
        const int b = blockIdx.x;
        if ((a.splits & 7) == 0) {
            const int s8 = a.splits >> 3;
            const int xcd = b & 7, i = b >> 3;
            split = xcd + 8 * (i % s8);
            group = i / s8;
        } else {
            split = b % a.splits;
            group = b / a.splits;
        }
    }

    if (QT == 1 && a.only && a.only[group] == 0u) return;   // workgroup-uniform: this query has its answer already (big-k pipeline)

    topk_init(tk);
    build_lut_lds<M, QT>(lut, reinterpret_cast<float *>(&tk.buf[0][0]), a, group);  // ends with a barrier

    const int64_t row_begin = (int64_t)split * a.rows_per_split;
    int64_t row_end = row_begin + a.rows_per_split;
    row_end = row_end < a.n_rows ? row_end : a.n_rows;
    const Row *rows = reinterpret_cast<const Row *>(a.codes);
    const int tid = threadIdx.x;

    Row cur[R], nxt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row_begin + r * kBlock + tid;
        if (row < row_end) cur[r] = rows[row];
        else cur[r] = Row{};
    }
    int tile = 0;
    for (int64_t base = row_begin; base < row_end; base += (int64_t)kBlock * R, ++tile) {
        // prefetch the next tile's rows
        const int64_t nbase = base + (int64_t)kBlock * R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = nbase + r * kBlock + tid;
            if (row < row_end) nxt[r] = rows[row];
            else nxt[r] = Row{};
        }
        uint32_t key[R][QT];
        uint32_t pay[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[q] = 0.0f;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float v[QT];
                lut_get<QT>(lut, m * 256 + code_byte<M>(cur[r], m), v);
#pragma unroll
                for (int q = 0; q < QT; ++q) acc[q] = __fadd_rn(acc[q], v[q]);
            }
            // keep the scheduler from hoisting every row's 16 table reads at once (register blow-up);
            // one row = M reads in flight per lane is already enough to cover the LDS latency
            __builtin_amdgcn_sched_barrier(0);
            const int64_t row = base + r * kBlock + tid;
            const bool valid = row < row_end;
            pay[r] = (uint32_t)row;
#pragma unroll
            for (int q = 0; q < QT; ++q) key[r][q] = valid ? __float_as_uint(acc[q]) : KEY_MAX;  // sums are >= +0
        }
        topk_tile<QT, R, CAP, TRIG>(tk, a.k, tile, key, pay);
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = nxt[r];
    }

    __syncthreads();
    topk_compact(tk, a.k);
    // ---- sorted partial result of this (query, split) ----
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi >= a.nq) break;
        const int cnt = tk.cnt[q];
        const int64_t o = ((int64_t)qi * a.splits + split) * a.k;
        for (int i = tid; i < a.k; i += kBlock) {
            if (i < cnt) {
                const unsigned long long e = tk.buf[q][i];
                a.part_d[o + i] = __uint_as_float((uint32_t)(e >> 32));
                a.part_id[o + i] = a.id_base + (int64_t)(uint32_t)e;
            } else {
                a.part_d[o + i] = __uint_as_float(0x7f800000u);
                a.part_id[o + i] = -1;
            }
        }
    }
}

// ==========================================================================================
// adc_scan16: the M = 16 scan with CONFLICT-FREE table reads.
//
// Problem measured on the kernel above (rocprofv3, profiles/r01_*): with one code row per lane, the 64
// lanes of a ds_read look up the SAME sub-quantiser with random codes, so 63-66 % of all LDS cycles
// are bank-conflict replays and the LDS pipe, ~80 % busy, is the bound.
//
// Fix: ds_read_b128 is serviced in groups of 16 lanes over 16 slots of 16 bytes.  With M = 16 and
// QT = 4, one table entry (4 queries x fp32) IS one slot, so if the 16 lanes of a group look up 16
// DIFFERENT sub-quantisers and the table is laid out [code][m][q] (slot = m), every group is
// conflict-free regardless of the codes.  Lane l therefore walks its own row in the rotated order
// m = (t + l) & 15, t = 0..15 (its 16 code bytes are rotated once in registers so the byte for step t
// sits at a compile-time position).
//
// The rotated order changes the fp32 rounding of the sum, so it is used ONLY as a filter: a row is
// pushed to the selection buffer when its rotated-order sum is below thr * (1 + 2^-17), a bound that
// provably admits every row whose reference-order sum is below thr (both are sums of the same 16
// non-negative floats: each is within 15 ulp-steps, 8.9e-7 relative, of the real sum).  At
// compaction each newly pushed row is re-read and summed in the REFERENCE order (m ascending,
// IVFOPQ.cpp:302-306) before the sort, so thr, the kept set and the reported distances are exactly
// the reference's.  Pushes are ~0.1 % of rows, the exact pass is noise.
// ==========================================================================================
constexpr int S16_CAP = 384;
constexpr int S16_TRIG = 256;

// thr key -> the bound the rotated-order sums are compared with
struct InflateThr {
    __device__ __forceinline__ uint32_t operator()(int /*q*/, uint32_t t) const
    {
        if (t >= 0x7f800000u) return t;  // no finite threshold yet (KEY_MAX) or +inf: compare as is
        const float f = __uint_as_float(t);
        return __float_as_uint(fmaf(f, 7.62939453125e-06f /* 2^-17 */, f));
    }
};

// exact reference-order distance of a pushed row, read through the table layout of adc_scan16
// (float index of (code j, sub-quantiser m, query q) = j*16*QT + (q>>2)*64 + m*4 + (q&3))
template <int QT>
struct ExactFix {
    static constexpr bool enabled = true;
    const float *lut;
    const uint4 *rows;
    __device__ __forceinline__ unsigned long long operator()(int q, unsigned long long e) const
    {
        const uint32_t row = (uint32_t)e;
        const uint4 c = rows[row];
        const uint32_t w[4] = { c.x, c.y, c.z, c.w };
        float s = 0.0f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const uint32_t j = (w[m >> 2] >> (8 * (m & 3))) & 0xffu;
            s = __fadd_rn(s, lut[j * (16 * QT) + (q >> 2) * 64 + m * 4 + (q & 3)]);
        }
        return ((unsigned long long)__float_as_uint(s) << 32) | row;
    }
};

// NT threads per workgroup.  The 64 KB of tables are shared by NT/64 waves; two workgroups fit a CU's
// LDS, so NT = 512 gives 4 waves per SIMD, NT = 1024 gives 8 (register budget 128 / 64 VGPRs).
// Measured on the bare loop (tools/ubench/scan_loop.hip): 256 threads 15.8, 512 threads 21.2,
// 1024 threads 24.9 T table reads/s -- more waves hide the LDS latency the skewed loop exposes.
template <int NT>
__global__ __launch_bounds__(NT, NT / 128) void adc_scan16_kernel(const ScanArgs a)
{
    constexpr int M = 16, QT = 4;
    constexpr int R = NT >= 1024 ? 1 : 2;  // rows per lane per tile (64-VGPR budget at 8 waves/SIMD)
    __shared__ __attribute__((aligned(16))) float lut[256 * 16 * QT];  // [code j][m][q]: one 16-byte slot per (j, m)
    __shared__ TopKShared<QT, S16_CAP> tk;

    int group, split;
    {
        const int b = blockIdx.x;
        if ((a.splits & 7) == 0) {
            const int s8 = a.splits >> 3;
            const int xcd = b & 7, i = b >> 3;
            split = xcd + 8 * (i % s8);
            group = i / s8;
        } else {
            split = b % a.splits;
            group = b / a.splits;
        }
    }
    const int tid = threadIdx.x;
    topk_init(tk);

    // ---- tables: the arithmetic of build_lut_lds (IVFOPQ.cpp:273-291), stored [j][m][q] ----
    {
        float *res = reinterpret_cast<float *>(&tk.buf[0][0]);
        for (int i = tid; i < QT * a.D; i += NT) {
            const int q = i / a.D, d = i - q * a.D;
            int qi = group * QT + q;
            qi = qi < a.nq ? qi : a.nq - 1;
            res[i] = __fsub_rn(a.q_rot[(int64_t)qi * a.D + d], a.centroid[d]);
        }
        __syncthreads();
        for (int e = tid; e < M * 256; e += NT) {  // (m, j) pairs; consecutive lanes -> consecutive j
            const int m = e >> 8, j = e & 255;
            float acc[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[q] = 0.0f;
            if (j < a.K) {
                const float *cb = a.books + ((int64_t)m * a.K + j) * a.step;
                for (int kk = 0; kk < a.step; ++kk) {
                    const float c = cb[kk];
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const float t = __fsub_rn(res[q * a.D + m * a.step + kk], c);
                        acc[q] = __fadd_rn(acc[q], __fmul_rn(t, t));
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < QT; ++q) acc[q] = __uint_as_float(0x7f800000u);
            }
            *reinterpret_cast<float4 *>(&lut[(j * 16 + m) * QT]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
    }
    const uint4 *rows = reinterpret_cast<const uint4 *>(a.codes);
    using Fix = ExactFix<QT>;
    const Fix fix{ lut, rows };
    const InflateThr thrx;

    const int64_t row_begin = (int64_t)split * a.rows_per_split;
    int64_t row_end = row_begin + a.rows_per_split;
    row_end = row_end < a.n_rows ? row_end : a.n_rows;

    // lane constants: rotation amount c and, packed one byte per step, the byte offset m*16 of
    // sub-quantiser m = (t + c) & 15 inside a 256-byte table row
    const uint32_t c = tid & 15;
    const uint32_t cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);

    // Row indices inside the split are 32-bit (the launcher keeps a split below 2^28 rows) so that the
    // loads are "scalar base + 32-bit lane offset".  Rows past the end are CLAMPED to the last row
    // instead of predicated off: an exec-masked load made the compiler wait vmcnt(0) on the prefetch
    // it had just issued; clamped rows are rejected by the range check in the (rare) push path.
    const uint32_t n_local = (uint32_t)(row_end > row_begin ? row_end - row_begin : 0);
    const char *rows_b = reinterpret_cast<const char *>(rows + row_begin);
    const uint32_t last = n_local ? n_local - 1 : 0;
    auto load_row = [&](uint32_t lrow) -> uint4 {
        const uint32_t cl = lrow < last ? lrow : last;
        return *reinterpret_cast<const uint4 *>(rows_b + (size_t)(cl * 16u));
    };
    uint4 cur[R], nxt[R];
    if (n_local) {
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = load_row(r * NT + tid);
    }
    int tile = 0;
    for (uint32_t base = 0; base < n_local; base += NT * R, ++tile) {
#pragma unroll
        for (int r = 0; r < R; ++r) nxt[r] = load_row(base + NT * R + r * NT + tid);
        uint32_t thr[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) thr[q] = tk.thr_x[q];
        uint32_t key[R][QT];
        bool want = false;
        uint32_t pending = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // rotate the 16 code bytes right by c: byte t of `rot` = code[(t + c) & 15]
            uint32_t d0 = __builtin_amdgcn_alignbit(cur[r].y, cur[r].x, cr8);
            uint32_t d1 = __builtin_amdgcn_alignbit(cur[r].z, cur[r].y, cr8);
            uint32_t d2 = __builtin_amdgcn_alignbit(cur[r].w, cur[r].z, cr8);
            uint32_t d3 = __builtin_amdgcn_alignbit(cur[r].x, cur[r].w, cr8);
            {
                const bool b0 = cq & 1;
                const uint32_t e0 = b0 ? d1 : d0, e1 = b0 ? d2 : d1, e2 = b0 ? d3 : d2, e3 = b0 ? d0 : d3;
                const bool b1 = cq & 2;
                d0 = b1 ? e2 : e0; d1 = b1 ? e3 : e1; d2 = b1 ? e0 : e2; d3 = b1 ? e1 : e3;
            }
            const uint32_t rot[4] = { d0, d1, d2, d3 };
            float a0, a1, a2, a3;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                // LDS byte address = code*256 + m*16, formed by ONE v_perm_b32:
                // byte0 <- moffp[t>>2].byte[t&3], byte1 <- rot[t>>2].byte[t&3], bytes 2,3 <- 0
                const uint32_t sel = 0x0c0c0000u | ((4u + (t & 3)) << 8) | (uint32_t)(t & 3);
                const uint32_t addr = __builtin_amdgcn_perm(rot[t >> 2], moffp[t >> 2], sel);
                const float4 v = *reinterpret_cast<const float4 *>(lut_b + addr);
                if (t == 0) { a0 = v.x; a1 = v.y; a2 = v.z; a3 = v.w; }
                else { a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w; }
            }
            __builtin_amdgcn_sched_barrier(0);
            key[r][0] = __float_as_uint(a0); key[r][1] = __float_as_uint(a1);
            key[r][2] = __float_as_uint(a2); key[r][3] = __float_as_uint(a3);
            // wave-level test on scalar masks: 4 v_cmp + 3 s_or, no per-lane boolean materialised
            const unsigned long long pm = __ballot(key[r][0] < thr[0]) | __ballot(key[r][1] < thr[1]) |
                                          __ballot(key[r][2] < thr[2]) | __ballot(key[r][3] < thr[3]);
            if (pm) {  // rare once the threshold has tightened
                const uint32_t lrow = base + r * NT + tid;
                if (lrow < n_local) {
#pragma unroll
                    for (int q = 0; q < QT; ++q)
                        if (key[r][q] < thr[q])
                            if (!topk_push<QT, S16_CAP, S16_TRIG>(tk, q, key[r][q], (uint32_t)(row_begin + lrow), want))
                                pending |= 1u << (r * QT + q);
                }
            }
        }
        topk_tile_end<QT, S16_CAP, NT>(tk, a.k, tile, want, pending, fix, thrx, [&](uint32_t pend) {
            uint32_t still = 0;
            bool dummy = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const uint32_t bit = 1u << (r * QT + q);
                    if ((pend & bit) && key[r][q] <= tk.thr_x[q]) {
                        if (!topk_push<QT, S16_CAP, S16_TRIG>(tk, q, key[r][q], (uint32_t)(row_begin + base + r * NT + tid), dummy))
                            still |= bit;
                    }
                }
            }
            return still;
        });
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = nxt[r];
    }

    __syncthreads();
    topk_compact<QT, S16_CAP, NT, Fix, InflateThr>(tk, a.k, fix, thrx);
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi >= a.nq) break;
        const int cnt = tk.cnt[q];
        const int64_t o = ((int64_t)qi * a.splits + split) * a.k;
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
    }
}

// ==========================================================================================
// adc_scan16q: 8 queries per pass over the codes, filter on 15-bit FIXED-POINT LOWER BOUNDS.
//
// adc_scan16 above is bound by LDS bandwidth: 4 bytes of table per look-up, 256 B/clk/CU.  Since its
// rotated-order sums are only a filter anyway, the filter table does not have to be fp32.  Here each
// table entry is a 16-bit integer  qv = floor((LUT[m][j] - min_m) / scale) - 1  (>= 0, per-query scale
// chosen so that sum_m max_j qv <= 32766), i.e. a rigorous LOWER bound of the entry in units of `scale`:
//     sum_m qv_m  <=  (exact_real_sum - sum_m min_m) / scale .
// One ds_read_b128 now serves EIGHT queries, sums are exact integers (v_pk_add_u16, order-free), and a
// row is a candidate when  sum < T  with  T = floor((thr*(1+1e-6) - bias)/scale) + 2  -- every row whose
// reference-order fp32 distance is below thr passes.  Candidates (~0.1 % of rows) are re-evaluated
// EXACTLY at compaction straight from the codebooks in the reference's operation order
// (IVFOPQ.cpp:279-291 + :302-306), so results stay bit-identical; no fp32 table is kept in LDS.
// Measured on the bare loop (tools/ubench/scan_loop_u16.hip): 36-45 T look-ups/s vs 21-25 for fp32.
// ==========================================================================================

// The QT queries' tables of `group`, quantised into LDS: lut[code j][m][q] u16, with the per-query scale / bias /
// lazy-selection parameters in qp.  mx_bits = QT x 16 words of scratch; nonfinite / lazy = QT flags each.
// Called by the whole workgroup; ends with a barrier.
template <int NT>
__device__ __forceinline__ void scan16q_build_tables(const ScanArgs &a, int group, uint32_t *lut, QuantParams &qp, uint32_t (*mx_bits)[16],
                                                     int *nonfinite, int *lazy)
{
    constexpr int M = 16, QT = SQ_QT;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < QT * 16) {
        qp.mn_bits[tid >> 4][tid & 15] = 0x7f7fffffu;  // FLT_MAX
        mx_bits[tid >> 4][tid & 15] = 0u;
    }
    if (tid < QT) nonfinite[tid] = 0;
    __syncthreads();
    // fp32 table entries of (m, j) for the QT queries, from the per-query tables a.lut_g
    // (lut_kernel: IVFOPQ.cpp:279-291); lanes walk consecutive j -> coalesced
    int qis[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        qis[q] = qi < a.nq ? qi : a.nq - 1;
    }
    auto entry = [&](int m, int j, float (&acc)[QT]) {
#pragma unroll
        for (int q = 0; q < QT; ++q)
            acc[q] = a.lut_g[((int64_t)qis[q] * M + m) * 256 + j];  // padded with +inf past K
    };
    // pass A: per (query, m) range of the finite entries (a wave covers 64 consecutive j of one m)
    for (int e = tid; e < M * 256; e += NT) {
        const int m = e >> 8, j = e & 255;
        float acc[QT];
        entry(m, j, acc);
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const uint32_t bits = __float_as_uint(acc[q]);
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
    if (tid < QT) {
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
        // lazy selection (scan_compact_lazy_q): usable when every table entry is finite and the band stays narrow
        const double sl = 34.0 + ceil(4e-6 * (32767.0 + bias * (double)inv));
        const bool lazy_ok = a.lazy && !nonfinite[q] && sl < 1024.0 && bias >= 0.0;
        qp.slack[q] = lazy_ok ? (uint32_t)sl : 0u;
        lazy[q] = lazy_ok ? 1 : 0;
    }
    __syncthreads();
    // pass B: quantise  qv = max(0, floor((v - min_m) * inv) - 1), non-finite entries -> 0 (a lower bound of anything)
    for (int e = tid; e < M * 256; e += NT) {
        const int m = e >> 8, j = e & 255;
        float acc[QT];
        entry(m, j, acc);
        uint32_t qv[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const float v = acc[q];
            int iv = 0;
            if (__float_as_uint(v) < 0x7f800000u) {
                const float f = __fmul_rn(__fsub_rn(v, qp.mn[q][m]), qp.inv_scale[q]);
                iv = (int)floorf(f) - 1;
                iv = iv < 0 ? 0 : (iv > SQ_MAXSUM ? SQ_MAXSUM : iv);
            }
            qv[q] = (uint32_t)iv;
        }
        *reinterpret_cast<uint4 *>(&lut[(j * 16 + m) * (QT / 2)]) =
            make_uint4(qv[0] | (qv[1] << 16), qv[2] | (qv[3] << 16), qv[4] | (qv[5] << 16), qv[6] | (qv[7] << 16));
    }
    __syncthreads();  // tables ready; the scratch words on tk.buf are dead from here on
}


template <int NT, int R, bool PREROT>
__global__ __launch_bounds__(NT, NT / 128) void adc_scan16q_kernel(const ScanArgs a)
{
    constexpr int QT = SQ_QT;
    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * 16 * QT / 2];  // u16 [code j][m][q]: 16 B per (j, m)
    __shared__ TopKShared<QT, SQ_CAP> tk;
    __shared__ QuantParams qp;
    __shared__ struct { int stop, done_waves; uint32_t next_chunk; uint32_t thr_pk[QT / 2]; int lazy[QT]; int nonfinite[QT]; } ck;

    SQ_T0();
    int group, split, my_splits = a.splits;
    int64_t my_rps = a.rows_per_split;
    {
        int b = blockIdx.x, g0 = 0;
        if (b >= a.groups_a * a.splits) {  // region B: the tail groups, split finer (dispatched last)
            b -= a.groups_a * a.splits;
            g0 = a.groups_a;
            my_splits = a.splits_b;
            my_rps = a.rows_per_split_b;
        }
        if ((my_splits & 7) == 0) {
            const int s8 = my_splits >> 3;
            const int xcd = b & 7, i = b >> 3;
            split = xcd + 8 * (i % s8);
            group = g0 + i / s8;
        } else {
            split = b % my_splits;
            group = g0 + b / my_splits;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    topk_init(tk);
    scan16q_build_tables<NT>(a, group, lut, qp, reinterpret_cast<uint32_t(*)[16]>(&tk.buf[0][0]), ck.nonfinite, ck.lazy);  // scratch on the (still unused) buffers
    SQ_T(0);  // prologue

    const uint4 *rows = reinterpret_cast<const uint4 *>(a.codes);
    const ExactFromLutBatch fixb{ rows, a.lut_g, a.K, a.nq, group };
    const QuantThr thrx{ &qp };
    if (tid < QT / 2) {  // pass-all until k rows are known -- or what the other row splits of these queries have already established
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
        tk.thr_x[2 * tid] = t2[0]; tk.thr_x[2 * tid + 1] = t2[1];
        ck.thr_pk[tid] = t2[0] | (t2[1] << 16);
    }
    if (tid == 0) { ck.stop = 0; ck.done_waves = 0; ck.next_chunk = 0; }
    __syncthreads();

    const int64_t row_begin = (int64_t)split * my_rps;
    int64_t row_end = row_begin + my_rps;
    row_end = row_end < a.n_rows ? row_end : a.n_rows;
    const uint32_t n_local = (uint32_t)(row_end > row_begin ? row_end - row_begin : 0);
    // PREROT: the main loop streams the pre-rotated copy (lane l needs its row rotated by l & 15 = row & 15 bytes;
    // doing that in registers costs 12 of the loop's ~83 VALU instructions per row, and VALU is what bounds it)
    const char *rows_b = PREROT ? reinterpret_cast<const char *>(reinterpret_cast<const uint4 *>(a.codes_rot) + row_begin)
                                : reinterpret_cast<const char *>(rows + row_begin);
    // 64 rows of chunk c (row group r): one 16-byte load per lane off a scalar base.  Chunks past the end (the prefetch of a wave that
    // is about to finish) read the last one; there is NO per-lane clamp: the last chunk may read up to 64 R - 1 rows past the
    // split -- the next split's rows, or the kDevSlack bytes every device buffer carries (host_util.h) -- and what is read there is
    // never used (lrow < n_local where candidates are taken).  (The clamp was 2 of the loop's ~66 vector instructions per row.)
    constexpr uint32_t WROWS = 64 * R;
    const uint32_t n_chunks = (n_local + WROWS - 1) / WROWS;
    const uint32_t last_chunk = n_chunks ? n_chunks - 1 : 0;
    const uint32_t lane16 = (uint32_t)(threadIdx.x & 63) * 16u;
    auto load_rows = [&](uint32_t chunk, int r) -> uint4 {
        const uint32_t cc = chunk < last_chunk ? chunk : last_chunk;  // wave-uniform
        return *reinterpret_cast<const uint4 *>(rows_b + (size_t)cc * (WROWS * 16u) + (uint32_t)(r * 1024) + lane16);
    };
    auto load64 = [&](uint32_t c64) -> uint4 {  // rows 64 c64 .. 64 c64 + 63, all inside the split (the seed's reads)
        return *reinterpret_cast<const uint4 *>(rows_b + (size_t)c64 * 1024u + lane16);
    };
    const uint32_t c = tid & 15;
    const uint32_t cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);
    if (a.seed && (n_local + 63) / 64 >= 4 * SQ_SEED_CHUNKS)  // workgroup-uniform
        scan16q_seed<NT, PREROT>(a.k, load64, moffp, cr8, cq, lut_b, reinterpret_cast<uint32_t *>(&tk.buf[0][0]), qp, ck.lazy, tk.thr_x, ck.thr_pk);
    SQ_T(0);

    // ---- main loop: waves run on their own between checkpoints ----
    // Between two checkpoints a wave processes 64*R-row chunks with NO workgroup barrier: candidates go to the shared buffers with LDS atomics.  A push that
    // finds its buffer full raises `stop`, the wave keeps its position (the `done` bits remember what it
    // already stored from that chunk) and everybody meets at the checkpoint, where the buffers are
    // compacted when one is full or past TRIG.  Row order across waves is arbitrary: legal here because
    // the filter is a superset test and the exact (distance, id) sort decides (block_topk.h).
    constexpr int NW = NT / 64;
    // chunks are handed out dynamically (one LDS atomic per 64*R rows): waves that run faster take more,
    // so nobody waits long at the final checkpoint.  `it` = chunk in hand, `it_next` = already reserved
    // and being prefetched.
    // (the atomic is spelled out: for `if (lane == 0) atomicAdd(...)` hipcc emits its wave-aggregation sequence -- two mbcnt, a
    //  compare, a population count and two moves -- around the one-lane atomic: 6 of the loop's 66 vector instructions per row)
    const uint32_t next_chunk_addr = (uint32_t)(uintptr_t)&ck.next_chunk;
    // Chunks go out in PAIRS (2 p, 2 p + 1): one atomic per two chunks, the odd one follows without asking.
    auto grab_pair = [&]() -> uint32_t {
        uint32_t v;
        asm volatile("" : "=v"(v));  // (only lane 0's value is read: no instruction to define the others)
        if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(next_chunk_addr), "v"(1u) : "memory");
        return 2u * (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    auto after = [&](uint32_t chunk) -> uint32_t { return (chunk & 1u) ? grab_pair() : chunk + 1u; };  // wave-uniform branch
    uint32_t it = grab_pair(), it_next = it + 1u, done = 0, done_for = 0xffffffffu;
    bool counted = false;
    uint4 cur[R], nxt[R];
    if (n_local) {
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = load_rows(it, r);
    }
    for (;;) {
        uint32_t tpk[QT / 2], bpk[QT / 2];
#pragma unroll
        for (int i = 0; i < QT / 2; ++i) { tpk[i] = ck.thr_pk[i]; bpk[i] = 0x80008000u - tpk[i]; }   // (both fields of tpk <= 0x7fff: no borrow)
        while (it < n_chunks) {
            const int stop_seen = ck.stop;  // read early, used after the chunk
            const uint32_t base = it * WROWS;
#pragma unroll
            for (int r = 0; r < R; ++r) nxt[r] = load_rows(it_next, r);
            bool failed_any = false;  // wave-uniform (a ballot inside the rare path): the common path tests a scalar
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // 16 look-ups, packed 15-bit sums (scan16q_row_sums; all 16 reads in flight here: this kernel has the registers)
                // the sums start at 0x8000 - T per field (T <= 32767, sum <= 32766: no carry): sum < T for any of the 8 queries  <=>  a CLEAR
                // bit 15 in one of the eight fields  <=>  the AND of the four words lacks a bit of 0x80008000 -- three ANDs, one AND with
                // the mask and a compare instead of four packed subtractions, three ORs, the mask and the compare (round 5)
                uint32_t s0 = bpk[0], s1 = bpk[1], s2 = bpk[2], s3 = bpk[3];
                scan16q_row_sums<PREROT, 16, true>(cur[r], moffp, cr8, cq, lut_b, s0, s1, s2, s3);
                const uint32_t sg = (~((s0 & s1) & (s2 & s3))) & 0x80008000u;
                if (__ballot(sg != 0)) {  // rare once the threshold has tightened
                    s0 -= bpk[0]; s1 -= bpk[1]; s2 -= bpk[2]; s3 -= bpk[3];   // the plain sums (field-wise: no borrow, each field >= its bias)
                    if (done_for != it) { done = 0; done_for = it; }  // (the bits belong to one chunk; only this path sets or reads them)
                    bool failed = false;
                    uint32_t lrow = base + r * 64 + lane;
                    asm volatile("" : "+v"(lrow));  // (keeps the tail test in here: hipcc otherwise folds it into the common path's branch)
                    if (sg != 0 && lrow < n_local) {
                        const uint32_t sums[4] = { s0, s1, s2, s3 };
#pragma unroll
                        for (int q = 0; q < QT; ++q) {
                            const uint32_t bit = 1u << (r * QT + q);
                            const uint32_t sq = (sums[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                            const uint32_t tq = (tpk[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                            if (sq < tq && !(done & bit)) {
                                bool dummy = false;
                                if (topk_push<QT, SQ_CAP, SQ_TRIG>(tk, q, sq, (uint32_t)(row_begin + lrow), dummy)) done |= bit;
                                else failed = true;
                            }
                        }
                    }
                    failed_any |= __ballot(failed) != 0;
                }
            }
            if (failed_any) {  // some buffer is full: stop everyone, come back to this chunk after the compaction
                if (lane == 0) ck.stop = 1;
                break;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) cur[r] = nxt[r];
            it = it_next;
            it_next = after(it);
            if (__builtin_amdgcn_readfirstlane(stop_seen)) break;
        }
        SQ_T(1);  // look-ups + pushes
        if (it >= n_chunks && !counted) {
            counted = true;
            if (lane == 0) atomicAdd(&ck.done_waves, 1);
        }
        __syncthreads();  // checkpoint (A)
        SQ_T(2);  // wait for the other waves
        bool need = ck.stop != 0;
#pragma unroll
        for (int q = 0; q < QT; ++q) need |= tk.cnt[q] >= SQ_TRIG;
        const bool all_done = ck.done_waves == NW;
        if (need) {  // workgroup-uniform
            {
                const int wv = tid >> 6;
                const int keep_max = a.k + 48 < SQ_TRIG - 24 ? a.k + 48 : SQ_TRIG - 24;
                for (int q = wv; q < QT; q += NW) {  // wave-uniform: one wave per query, in registers
                    const int keep = ck.lazy[q] ? scan_compact_lazy_q<QT, SQ_CAP>(tk, q, a.k, fixb, thrx, qp.slack[q], keep_max, &ck.lazy[q])
                                                : topk_compact_wave_q<QT, SQ_CAP, false>(tk, q, a.k, fixb, thrx);
                    if (lane == 0) {
                        tk.cnt[q] = keep;
                        if (a.gthr) {  // the row splits of a query tighten each other's filter (same tables -> same units)
                            const int qi = group * QT + q;
                            if (qi < a.nq) {
                                const uint32_t mine = tk.thr_x[q];
                                const uint32_t seen = atomicMin(&a.gthr[qi], mine);
                                tk.thr_x[q] = seen < mine ? seen : mine;
                            }
                        }
                    }
                }
                __syncthreads();
            }
            if (tid < QT / 2) ck.thr_pk[tid] = tk.thr_x[2 * tid] | (tk.thr_x[2 * tid + 1] << 16);
            if (tid == 0) ck.stop = 0;
        }
        __syncthreads();  // checkpoint (B)
        SQ_T(3);  // checkpoint protocol + compactions
        if (all_done && !need) break;
    }

    __syncthreads();
    topk_compact_wave<QT, SQ_CAP, NT, true>(tk, a.k, fixb, thrx);
    SQ_T(4);  // final compaction
    SQ_TEND();
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi >= a.nq) break;
        const int cnt = tk.cnt[q];
        // a group of the first region that was scanned in one piece holds its queries' final lists: they go where the caller wants them
        const bool in_place = a.out_d != nullptr && group < a.groups_a;   // (out_d is only set when that region has one split)
        float *const pd = in_place ? a.out_d : a.part_d;
        int64_t *const pi = in_place ? a.out_id : a.part_id;
        const int64_t o = in_place ? (int64_t)qi * a.k : ((int64_t)qi * a.stride + split) * a.k;
        for (int i = tid; i < a.k; i += NT) {
            if (i < cnt) {
                const unsigned long long e = tk.buf[q][i];
                pd[o + i] = __uint_as_float((uint32_t)(e >> 32));
                pi[o + i] = a.id_base + (int64_t)(uint32_t)e;
            } else {
                pd[o + i] = __uint_as_float(0x7f800000u);
                pi[o + i] = -1;
            }
        }
        // a region with fewer splits than the partial stride leaves the other slots empty for the merge
        if (!in_place && split == 0 && my_splits < a.stride) {
            const int64_t o2 = ((int64_t)qi * a.stride + my_splits) * a.k;
            for (int i = tid; i < (a.stride - my_splits) * a.k; i += NT) {
                a.part_d[o2 + i] = __uint_as_float(0x7f800000u);
                a.part_id[o2 + i] = -1;
            }
        }
    }
}

// ==========================================================================================
// adc_scan16a: adc_scan16q without checkpoints -- compactions run BESIDE the scan.
//
// In adc_scan16q every compaction parks the whole workgroup: all waves meet at a barrier, one wave per query
// compacts, a second barrier releases them (13 such stops per workgroup over 1 M rows: 15 % of its life, and the
// reason a row split costs as much as 0.38 M rows).  Here nobody waits for a compaction that is not in his way:
//   * a push takes its slot with one LDS atomic (cnt[q]) and stores the entry with one 8-byte store (free slots hold SQA_EMPTY);
//   * the pusher that takes slot TRIG-1 raises bit q of `duty`; the workgroup's last wave does not scan: it sleeps on
//     `duty` and compacts query q ON ITS OWN while the 15 scanning waves keep scanning and keep pushing into the
//     CAP-TRIG slots above the trigger (the loop is bound by VALU issue, not by the number of waves: the scanning waves
//     take the slots the 16th would have used);
//   * the service wave LOCKS the counter (cnt[q] += LOCK: every later push sees "full"), waits until none of the
//     slots handed out before the lock reads SQA_EMPTY any more, selects (scan_compact_lazy_q / topk_compact_wave_q as
//     in adc_scan16q), resets the slots it freed, publishes the new 15-bit threshold (a 2-byte store into thr_pk,
//     then epoch++), and unlocks with cnt[q] = kept;
//   * a scanning wave reads `epoch` once per chunk (where adc_scan16q read `stop`) and reloads the packed
//     thresholds only when it moved;
//   * a lane whose push found the buffer full keeps its candidate, sleeps, re-reads the threshold (the candidate may
//     have dropped out) and tries again -- rare, because a histogram of the first 2048 rows seeds the thresholds
//     (see "seed" in the kernel), so the scan never goes through the phase in which every row passes.
// Progress: the trigger slot lies below the capacity, so a full buffer always has its duty bit up or its compaction
// running; a compaction only waits for stores that follow their atomic unconditionally.  Results: the buffers hold a
// superset of the rows below the final threshold, exactly as before (a push under an older, looser threshold is still
// a valid candidate), and the final compaction re-sums in the reference's order.
// ==========================================================================================
constexpr int SQA_LOCK = 1 << 20;
#ifndef SQA_SERVERS
#define SQA_SERVERS 2
#endif


struct ScanAsyncCtl {
    __attribute__((aligned(16))) uint32_t thr_pk[SQ_QT / 2];  // 15-bit thresholds, two per word
    uint32_t next_chunk, epoch;
    uint32_t duty;  // bit q: query q wants a compaction
    int done_waves;  // scanning waves that are out of rows
    int lazy[SQ_QT], nonfinite[SQ_QT];
    // what the slow path needs of the kernel's arguments (uniform; kept here so that its call passes three pointers)
    const uint4 *rows;
    const float *lut_g;
    uint32_t *gthr;
    int K, nq, group, k;
#ifdef CVTMI_SCAN_TIMING
    unsigned long long dbg[8];
#endif
};
using ScanAsyncTopK = TopKShared<SQ_QT, SQ_CAP>;
constexpr unsigned long long SQA_EMPTY = ~0ull;  // a slot nobody has stored into since the last compaction (no entry looks like it: row ids stop at 2^32 - 2)

__device__ __forceinline__ void scan16a_load_thr(const ScanAsyncCtl &ck, uint32_t (&tpk)[SQ_QT / 2])
{
#pragma unroll
    for (int i = 0; i < SQ_QT / 2; ++i) tpk[i] = __hip_atomic_load(&ck.thr_pk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one attempt to store the candidates in `pend` (bit q = this lane's row is a candidate of query q); the stored ones leave
// `pend`; the pusher that takes the trigger slot raises the query's duty bit.  Returns the duties this lane raised.
__device__ __forceinline__ uint32_t scan16a_push(ScanAsyncTopK &tk, ScanAsyncCtl &ck, uint32_t &pend, const uint32_t (&sums)[4], uint32_t row)
{
    uint32_t trig = 0;
#pragma unroll
    for (int q = 0; q < SQ_QT; ++q) {
        if (!(pend & (1u << q))) continue;
        const int pos = atomicAdd(&tk.cnt[q], 1);
        if (pos < SQ_CAP) {
            const uint32_t sq = (sums[q >> 1] >> (16 * (q & 1))) & 0xffffu;
            tk.buf[q][pos] = ((unsigned long long)sq << 32) | row;  // one 8-byte store: a reader sees SQA_EMPTY or the entry
            pend &= ~(1u << q);
            if (pos == SQ_TRIG - 1) trig |= 1u << q;
        }
    }
    if (trig) atomicOr(&ck.duty, trig);
    return trig;
}

// ---- the compaction side of adc_scan16a ----
// Coding rule for these wave-level loops: no `if (lane == 0)` region next to a loop's back edge.  Steps that only one lane
// may take effect in are atomics whose operand is neutral in the other lanes (add 0, and ~0) or stores of a wave-uniform
// value.  (With one-lane regions at the end of the service loop hipcc once structured it so that lanes 1..63 left the
// loop after the first compaction and lane 0 compacted alone from then on: four entries seen, threshold back at "pass all",
// a compaction every 40 rows -- 100x the run time, results still exact.)

// takes one raised duty bit: the query to compact, -1 if none is up, -2 if another wave was faster (look again)
__device__ __forceinline__ int scan16a_claim(ScanAsyncCtl &ck)
{
    const bool l0 = (threadIdx.x & 63) == 0;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ck.duty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (m == 0) return -1;
    const int q = __ffs((int)m) - 1;
    const uint32_t old = (uint32_t)__builtin_amdgcn_readfirstlane((int)atomicAnd(&ck.duty, l0 ? ~(1u << q) : ~0u));
    return (old >> q) & 1u ? q : -2;
}

// query q is compacted by the calling wave alone (q wave-uniform, its duty bit taken by the caller)
__device__ __forceinline__ void scan16a_compact(ScanAsyncTopK &tk, ScanAsyncCtl &ck, QuantParams &qp, int q)
{
    constexpr int QT = SQ_QT;
    const int lane = threadIdx.x & 63;
    const bool l0 = lane == 0;
    const int k = ck.k, group = ck.group;
    const ExactFromLutBatch fixb{ ck.rows, ck.lut_g, ck.K, ck.nq, group };
    const QuantThr thrx{ &qp };
    const int keep_max = k + 48 < SQ_TRIG - 24 ? k + 48 : SQ_TRIG - 24;
    [[maybe_unused]] const long long t_c0 = SQA_NOW();
    const int n_old = __builtin_amdgcn_readfirstlane(atomicAdd(&tk.cnt[q], l0 ? SQA_LOCK : 0));  // from here on every push to q fails
    const int n = n_old < SQ_CAP ? n_old : SQ_CAP;
    for (;;) {  // the slots handed out before the lock may still be on their way
        bool hole = false;
#pragma unroll
        for (int r = 0; r < (SQ_CAP + 63) / 64; ++r) {
            const int p = r * 64 + lane;
            const unsigned long long e = __hip_atomic_load(&tk.buf[q][p < n ? p : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            hole |= p < n && e == SQA_EMPTY;
        }
        if (!__ballot(hole)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int lazy = __builtin_amdgcn_readfirstlane(ck.lazy[q]);
    const int keep = lazy ? scan_compact_lazy_q<QT, SQ_CAP>(tk, q, k, fixb, thrx, qp.slack[q], keep_max, &ck.lazy[q], n)
                          : topk_compact_wave_q<QT, SQ_CAP, false>(tk, q, k, fixb, thrx, n);
#pragma unroll
    for (int r = 0; r < (SQ_CAP + 63) / 64; ++r) {  // the slots above the kept entries are free again
        const int p = r * 64 + lane;
        if (p >= keep && p < n) tk.buf[q][p] = SQA_EMPTY;
    }
    uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk.thr_x[q]);
    const int qi = group * QT + q;
    if (ck.gthr && qi < ck.nq) {  // (wave-uniform) the row splits of a query tighten each other's filter (same tables -> same units)
        uint32_t seen = 0xffffffffu;
        if (l0) seen = atomicMin(&ck.gthr[qi], t);  // one lane: 64 lanes on one global address cost 10 us
        seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)seen);
        t = seen < t ? seen : t;
        tk.thr_x[q] = t;
    }
    t = t < 32767u ? t : 32767u;
    __hip_atomic_store(reinterpret_cast<uint16_t *>(ck.thr_pk) + q, (uint16_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    atomicAdd(&ck.epoch, l0 ? 1u : 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    atomicExch(&tk.cnt[q], keep);  // unlock (every lane writes the same value)
    SQA_ADD(2, 1);
    SQA_ADD(3, SQA_NOW() - t_c0);
    SQA_ADD(0, n);
    SQA_ADD(1, keep);
    SQA_ADD(6, lazy);
    SQA_ADD(7, n_old > SQ_CAP ? 1 : 0);
}

// The service wave: compacts the queries whose duty bit is up until the `scanners` scanning waves have all run out of rows.
// At raised priority: under eight waves per SIMD an instruction of this wave would otherwise issue every ~32 cycles, and every
// counter it holds locked meanwhile stalls pushes.
__device__ __attribute__((noinline)) void scan16a_serve(ScanAsyncTopK &tk, ScanAsyncCtl &ck, QuantParams &qp, int scanners)
{
    __builtin_amdgcn_s_setprio(3);
    for (;;) {
        const int q = scan16a_claim(ck);
        if (q >= 0) {
            scan16a_compact(tk, ck, qp, q);
        } else if (q == -1) {
            // (a bit raised after this look by a wave that then finishes is left to the final compaction: nobody waits for it)
            const int done = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ck.done_waves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (done == scanners) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

template <int NT, int R, bool PREROT>
__global__ __launch_bounds__(NT, NT / 128) void adc_scan16a_kernel(const ScanArgs a)
{
    constexpr int QT = SQ_QT;
    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * 16 * QT / 2];  // u16 [code j][m][q]: 16 B per (j, m)
    __shared__ TopKShared<QT, SQ_CAP> tk;
    __shared__ QuantParams qp;
    __shared__ ScanAsyncCtl ck;

    SQ_T0();
    int group, split, my_splits = a.splits;
    int64_t my_rps = a.rows_per_split;
    {
        int b = blockIdx.x, g0 = 0;
        if (b >= a.groups_a * a.splits) {  // region B: the tail groups, split finer (dispatched last)
            b -= a.groups_a * a.splits;
            g0 = a.groups_a;
            my_splits = a.splits_b;
            my_rps = a.rows_per_split_b;
        }
        if ((my_splits & 7) == 0) {
            const int s8 = my_splits >> 3;
            const int xcd = b & 7, i = b >> 3;
            split = xcd + 8 * (i % s8);
            group = g0 + i / s8;
        } else {
            split = b % my_splits;
            group = g0 + b / my_splits;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    topk_init(tk);
    scan16q_build_tables<NT>(a, group, lut, qp, reinterpret_cast<uint32_t(*)[16]>(&tk.buf[0][0]), ck.nonfinite, ck.lazy);
    SQ_T(0);  // prologue

    const uint4 *rows = reinterpret_cast<const uint4 *>(a.codes);
    const ExactFromLutBatch fixb{ rows, a.lut_g, a.K, a.nq, group };
    const QuantThr thrx{ &qp };
    if (tid < QT / 2) {  // pass-all until k rows are known -- or what the other row splits of these queries have already established
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
        tk.thr_x[2 * tid] = t2[0]; tk.thr_x[2 * tid + 1] = t2[1];
        ck.thr_pk[tid] = t2[0] | (t2[1] << 16);
    }
#ifdef CVTMI_SCAN_TIMING
    if (tid < 8) ck.dbg[tid] = 0;
#endif
    if (tid == 0) {
        ck.next_chunk = 0; ck.epoch = 0; ck.duty = 0; ck.done_waves = 0;
        ck.rows = rows; ck.lut_g = a.lut_g; ck.gthr = a.gthr; ck.K = a.K; ck.nq = a.nq; ck.group = group; ck.k = a.k;
    }
    __syncthreads();

    const int64_t row_begin = (int64_t)split * my_rps;
    int64_t row_end = row_begin + my_rps;
    row_end = row_end < a.n_rows ? row_end : a.n_rows;
    const uint32_t n_local = (uint32_t)(row_end > row_begin ? row_end - row_begin : 0);
    const char *rows_b = PREROT ? reinterpret_cast<const char *>(reinterpret_cast<const uint4 *>(a.codes_rot) + row_begin)
                                : reinterpret_cast<const char *>(rows + row_begin);
    // 64 rows of chunk c (row group r): one 16-byte load per lane off a scalar base.  Chunks past the end (the prefetch of a wave that
    // is about to finish) read the last one; there is NO per-lane clamp: the last chunk may read up to 64 R - 1 rows past the
    // split -- the next split's rows, or the kDevSlack bytes every device buffer carries (host_util.h) -- and what is read there is
    // never used (lrow < n_local where candidates are taken).  (The clamp was 2 of the loop's ~66 vector instructions per row.)
    constexpr uint32_t WROWS = 64 * R;
    const uint32_t n_chunks = (n_local + WROWS - 1) / WROWS;
    const uint32_t last_chunk = n_chunks ? n_chunks - 1 : 0;
    const uint32_t lane16 = (uint32_t)(threadIdx.x & 63) * 16u;
    auto load_rows = [&](uint32_t chunk, int r) -> uint4 {
        const uint32_t cc = chunk < last_chunk ? chunk : last_chunk;  // wave-uniform
        return *reinterpret_cast<const uint4 *>(rows_b + (size_t)cc * (WROWS * 16u) + (uint32_t)(r * 1024) + lane16);
    };
    auto load64 = [&](uint32_t c64) -> uint4 {  // rows 64 c64 .. 64 c64 + 63, all inside the split (the seed's reads)
        return *reinterpret_cast<const uint4 *>(rows_b + (size_t)c64 * 1024u + lane16);
    };
    const uint32_t c = tid & 15;
    const uint32_t cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);
    static_assert(R == 1, "adc_scan16a: one row per lane and chunk");
    constexpr int NW = NT / 64, NSV = SQA_SERVERS, SV = NW - NSV;  // waves SV.. serve the compactions, the others scan
    const int wave = tid >> 6;

    if (a.seed && n_chunks >= 4 * SQ_SEED_CHUNKS)  // workgroup-uniform
        scan16q_seed<NT, PREROT>(a.k, load64, moffp, cr8, cq, lut_b, reinterpret_cast<uint32_t *>(&tk.buf[0][0]), qp, ck.lazy, tk.thr_x, ck.thr_pk);
    for (int i = tid; i < QT * SQ_CAP; i += NT) (&tk.buf[0][0])[i] = SQA_EMPTY;
    __syncthreads();
    SQ_T(3);  // seed

    // ---- main loop: no workgroup barrier until the rows are exhausted ----
    if (wave >= SV) {
        scan16a_serve(tk, ck, qp, SV);
    } else {
        auto grab = [&]() -> uint32_t {
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(&ck.next_chunk, 1u);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        };
        uint32_t it = grab(), it_next = grab();
        uint4 cur = make_uint4(0, 0, 0, 0), nxt;
        if (n_local) cur = load_rows(it, 0);
        uint32_t tpk[QT / 2];
        scan16a_load_thr(ck, tpk);
        uint32_t epoch_seen = 0;
        while (it < n_chunks) {
            const uint32_t ep = __hip_atomic_load(&ck.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // read early, used after the chunk
            const uint32_t base = it * WROWS;
            nxt = load_rows(it_next, 0);
            uint32_t s0, s1, s2, s3;
            scan16q_row_sums<PREROT>(cur, moffp, cr8, cq, lut_b, s0, s1, s2, s3);
            // sum < T for any of the 8 queries  <=>  a sign bit in the packed (sum - T)  (both < 2^15)
            const uint32_t sg = (pk_sub_i16(s0, tpk[0]) | pk_sub_i16(s1, tpk[1]) | pk_sub_i16(s2, tpk[2]) | pk_sub_i16(s3, tpk[3])) & 0x80008000u;
            if (__ballot(sg != 0)) {  // rare once the threshold has tightened
                const uint32_t lrow = base + lane;
                const uint32_t sums[4] = { s0, s1, s2, s3 };
                uint32_t cand = 0;
                if (sg != 0 && lrow < n_local) {
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const uint32_t sq = (sums[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                        const uint32_t tq = (tpk[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                        if (sq < tq) cand |= 1u << q;
                    }
                }
                const uint32_t row = (uint32_t)(row_begin + lrow);
                scan16a_push(tk, ck, cand, sums, row);
                while (__ballot(cand != 0)) {  // a buffer was full: its compaction is under way (the trigger slot lies below) -- wait for room
                    SQA_ADD(5, 1);
                    __builtin_amdgcn_s_sleep(4);
                    int cnt[QT];
                    scan16a_load_thr(ck, tpk);  // a waiting candidate may have dropped out meanwhile
#pragma unroll
                    for (int q = 0; q < QT; ++q) cnt[q] = __hip_atomic_load(&tk.cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    uint32_t tryq = 0;
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const uint32_t sq = (sums[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                        const uint32_t tq = (tpk[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                        if (!(sq < tq)) cand &= ~(1u << q);
                        else if ((cand & (1u << q)) && cnt[q] < SQ_CAP) tryq |= 1u << q;  // (a full or locked counter is left alone)
                    }
                    const uint32_t before = tryq;
                    scan16a_push(tk, ck, tryq, sums, row);
                    cand &= ~(before & ~tryq);
                }
            }
            cur = nxt;
            it = it_next;
            it_next = grab();
            const uint32_t ep_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)ep);
            if (ep_u != epoch_seen) {
                epoch_seen = ep_u;
                scan16a_load_thr(ck, tpk);
            }
        }
        if (lane == 0) atomicAdd(&ck.done_waves, 1);
    }
    SQ_T(1);  // look-ups + pushes
    __syncthreads();  // every scanning wave is out of rows, the service wave has left its loop: no counter is locked
    SQ_T(2);
    topk_compact_wave<QT, SQ_CAP, NT, true>(tk, a.k, fixb, thrx);
    SQ_T(4);  // final compaction
    SQ_TEND();
#ifdef CVTMI_SCAN_TIMING
    if (tid < 8) atomicAdd(&g_scan_dbg2[tid], ck.dbg[tid]);
#endif
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi >= a.nq) break;
        const int cnt = tk.cnt[q];
        const int64_t o = ((int64_t)qi * a.stride + split) * a.k;
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
        if (split == 0 && my_splits < a.stride) {  // a region with fewer splits than the partial stride: the other slots stay empty
            const int64_t o2 = ((int64_t)qi * a.stride + my_splits) * a.k;
            for (int i = tid; i < (a.stride - my_splits) * a.k; i += NT) {
                a.part_d[o2 + i] = __uint_as_float(0x7f800000u);
                a.part_id[o2 + i] = -1;
            }
        }
    }
}

static int g_scan_seed = 1;
void set_scan_seed(int v) { g_scan_seed = v != 0; }
static std::atomic<int> g_scan_tail_splits{0};   // cvtmi_set_tuning("scan_tail_splits"): see plan_scan
void set_scan_tail_splits(int v) { g_scan_tail_splits = v; }
int scan_seed_enabled() { return g_scan_seed; }

// Row ids travel as 32-bit payloads: one launch covers at most 2^32-1 rows.
static int cu_count()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

ScanPlan plan_scan(const OpqModelDev &m, int64_t n_rows, int64_t nq, int k, int want_qtile, int want_splits,
                   int want_variant)
{
    ScanPlan p;
    int qt = want_qtile;
    if (qt != 1 && qt != 2 && qt != 4 && qt != 8) qt = (nq >= 4) ? 4 : (nq >= 2 ? 2 : 1);
    if (k > 128) qt = 1;
    // M = 16 kernels with conflict-free skewed table reads:
    //   variant 1 / 2: fp32 tables, 4 queries per pass, 512- / 1024-thread workgroups (adc_scan16)
    //   variant 3 / 4: 15-bit lower-bound tables, 8 queries per pass, 1024- / 512-thread workgroups (adc_scan16q)
    p.variant = 0;
    //   variant 5: adc_scan16a (compactions beside the scan); variant 6: adc_scan16h (adc_scan_h.hip: histogram bounds, spilled
    //   candidates, persistent grid) -- takes any number of queries
    // want_variant 7 = the library's own choice between 3 and 6: adc_scan16h where it measured ahead on a cache-resident index
    // (63 ... 500 query groups at 1 M rows: -6 % at 1000 queries, -10 ... -14 % at 2500; tools/sweep_scan_h.py), adc_scan16q elsewhere
    if (k > 128) { want_variant = 0; want_qtile = 1; }  // the exact row-per-lane kernel, one query per workgroup (kernels.h: kBigK)
    // 7 = choose: the persistent grid (6) wherever a query group is cut into row segments -- its segments share one histogram per query, so
    // the candidates a segment has to store while its bound is loose fall with the split count (1 M rows: nq = 128 0.24 -> 0.20 ms wall,
    // 1000 0.59 -> 0.47, 3000 1.23 -> 1.10); whole groups (nq > 3200 at 1 M rows) run 4 % faster on adc_scan16q
    // Round 5 swept the table size as well (tools/sweep_scan_dispatch.py, profiles/r05_scan_dispatch_sweep.txt): on a code matrix that
    // still fits the Infinity Cache (<= 256 MB: 16 M rows) the persistent grid stays ahead up to 512 queries (10 M rows: 64 / 128 / 256 /
    // 512 queries 0.37 / 0.57 / 0.84 / 1.63 ms against 0.53 / 0.67 / 0.96 / 1.81), level from 1000; and it is ahead from ~33 queries on, not
    // 100 (what the small-batch form does not take: api.hip scans_chosen).  Beyond 16 M rows the two are within a few per cent.
    if (want_variant == 7) {
        const bool resident = n_rows * 16 <= (96LL << 20), near = n_rows * 16 <= (256LL << 20);
        want_variant = (m.M == 16 && n_rows >= 131072 && nq >= 33 && ((resident && nq <= 3200) || (near && nq <= 512))) ? 6 : 3;
        // one to three queries on a table too large for the small-batch form (api.hip scans_chosen): adc_scan16q takes four queries or
        // more, and what is left below it -- the fp32-table / row-per-lane kernels -- needs 1.9-2.9 ms on 100 M rows against 1.2 ms here
        if (m.M == 16 && !near && nq < 4) want_variant = 6;
    }
    if (m.M == 16 && want_variant >= 3 && (nq >= 4 || want_variant == 6) && m.D <= 256) { p.variant = want_variant; qt = 8; }
    else if (m.M == 16 && want_variant >= 1) {
        if (want_variant <= 2 && (want_qtile == 0 || want_qtile == 4) && nq >= 4) { p.variant = want_variant; qt = 4; }
        else if (want_variant >= 3 && nq >= 2) { p.variant = 1; qt = 4; }
    }
    if (qt == 8 && m.M == 16 && p.variant < 3) qt = 4;  // row-per-lane: 8 x 16 KB fp32 tables do not fit beside the buffers
    p.qtile = qt;
    const int64_t groups = (nq + qt - 1) / qt;
    int s = want_splits;
    if (s <= 0) {
        // Cost model fitted to tools/sweep_scan.py runs on 1 M rows (nq = 1250 ... 10000, splits 1-3) and
        // tools/scan_trace.py: a workgroup costs a fixed ~0.38 M row-equivalents (table build, selection warm-up,
        // compactions, and the slowdown of sharing its CU) plus its rows; `slots` workgroups run at a time
        // (2 per CU); a partly filled last round is not as bad as its occupancy, because a workgroup alone on
        // its CU runs ~1.27x faster: at <= 50 % occupancy the round takes 0.78 of a full one.
        // Pick the S that minimises rounds(groups * S) * (FIX + rows / S), at least 16 K rows per workgroup.
        const int64_t slots = 2 * (int64_t)cu_count();
        const double fix = 380000.0;
        const auto wg_cost = [&](int64_t S) { return fix + (double)n_rows / (double)S; };
        const auto rounds_of = [&](int64_t blocks) {
            const int64_t full = blocks / slots, last = blocks % slots;
            if (!last) return (double)full;
            const double f = (double)last / (double)slots;
            return (double)full + (f <= 0.5 ? 0.78 : 0.78 + 0.44 * (f - 0.5));
        };
        int64_t best = 1, best_ga = 0, best_sb = 0;
        double best_cost = 1e300;
        // shards past 2^28 rows must be split anyway (32-bit byte offsets, below): the search then starts at that count, and
        // multiples of 8 get 3 % of credit -- they keep a row split on one XCD, whose 64 workgroups share each row through its
        // L2 (2^30 rows x 10 000 queries: 5 splits 3.32 s, 8 splits 3.09 s, 9 splits 3.45 s)
        const int64_t lo = p.variant >= 1 ? std::max<int64_t>(1, (n_rows + ((1LL << 28) - 4096) - 1) / ((1LL << 28) - 4096)) : 1;
        best = lo;
        for (int64_t cand = lo; cand <= 64; ++cand) {
            if (cand > lo && n_rows / cand < 16384) break;
            double cost = rounds_of(groups * cand) * wg_cost(cand);
            if (lo > 1 && cand % 8 == 0) cost *= 0.97;
            if (cost < best_cost * 0.98) { best_cost = cost; best = cand; }
        }
        // adc_scan16q can also run two regions: `full` whole rounds of S-split workgroups, then the remaining groups split finer
        // (S2 > S).  Round 5 measured what the last round really costs (tools/sweep_tail.py, 1 M rows): ONE workgroup keeps a CU's LDS
        // pipe as busy as two do (6000 queries = 512 + 238 groups run at the per-query rate of 4096 = one exact round), so a last
        // round is only expensive while it leaves CUs EMPTY: 101 groups past two full rounds (9000 queries) 2.89 -> 3.16 M queries/s
        // in 2 splits (4: 3.12, 3: 3.01), 226 or 238 groups (10 000 / 6000 queries): nothing to gain (+0.6 % / -6 %).  Rule: cut the
        // remainder into the power of two of splits that still leaves at most one workgroup per CU.
        // cvtmi_set_tuning("scan_tail_splits"): 0 = this rule, -1 = never, S > 0 = S splits whenever there is a remainder.
        const int64_t full = groups * best / slots;  // whole rounds of region A
        const int64_t rem = groups * best % slots;
        const int tail_req = g_scan_tail_splits.load();
        if (p.variant >= 3 && full >= 1 && rem != 0 && best == 1 && tail_req >= 0) {
            int64_t sb = 0;
            if (tail_req > 0) sb = tail_req;
            else if (rem * 2 <= slots / 2) { sb = 2; while (sb < 8 && rem * sb * 2 <= slots / 2) sb *= 2; }
            while (sb > best && n_rows / sb < 16384) sb /= 2;   // at least 16 K rows per workgroup
            if (sb > best) { best_ga = full * slots / best; best_sb = sb; }
        }
        s = (int)best;
        p.groups_a = (int)best_ga;
        p.splits_b = (int)best_sb;
    }
    // the skewed kernels address a split's rows with 32-bit byte offsets (row * 16): at most 2^28 - 4096 rows per split
    if (p.variant >= 1) {
        const int64_t max_rows = (1LL << 28) - 4096;
        const int64_t min_splits = (n_rows + max_rows - 1) / max_rows;
        if (s < min_splits) { s = (int)min_splits; p.groups_a = 0; p.splits_b = 0; }
    }
    p.splits = s;
    (void)k;
    return p;
}

template <int M, int QT>
static int launch_t(const ScanArgs &a, hipStream_t st)
{
    const int64_t blocks = (int64_t)a.groups * a.splits;
    if (blocks <= 0) return CVTMI_OK;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "adc_scan: grid too large (%lld)", (long long)blocks);
    hipLaunchKernelGGL((adc_scan_kernel<M, QT>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

template <int M>
static int launch_m(const ScanArgs &a, int qt, hipStream_t st)
{
    if (a.k > 128) {  // one query per workgroup, the large selection buffer (kernels.h: kBigK)
        const int64_t blocks = (int64_t)a.groups * a.splits;
        if (qt != 1) return fail(CVTMI_EINVAL, "adc_scan: k=%d needs qtile 1", a.k);
        if (blocks <= 0) return CVTMI_OK;
        if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "adc_scan: grid too large (%lld)", (long long)blocks);
        hipLaunchKernelGGL((adc_scan_kernel<M, 1, kBigCap, kBigTrig>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    switch (qt) {
        case 1: return launch_t<M, 1>(a, st);
        case 2: return launch_t<M, 2>(a, st);
        case 4: return launch_t<M, 4>(a, st);
        case 8:
            if constexpr (M <= 8) return launch_t<M, 8>(a, st);
            break;
        default: break;
    }
    return fail(CVTMI_EUNSUPPORTED, "adc_scan: qtile %d not built for M=%d", qt, M);
}

// codes_rot[r] = codes[r] rotated left by r & 15 bytes: stored byte j = code byte (j + r) & 15   (M = 16 rows)
__global__ __launch_bounds__(kBlock) void rotate_codes_kernel(const uint4 *__restrict__ codes, uint4 *__restrict__ out,
                                                              int64_t row0, int64_t n)
{
    const int64_t r = row0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const uint4 v = codes[r];
    const uint32_t c = (uint32_t)r & 15u, cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t d0 = __builtin_amdgcn_alignbit(v.y, v.x, cr8), d1 = __builtin_amdgcn_alignbit(v.z, v.y, cr8);
    uint32_t d2 = __builtin_amdgcn_alignbit(v.w, v.z, cr8), d3 = __builtin_amdgcn_alignbit(v.x, v.w, cr8);
    const bool b0 = cq & 1;
    const uint32_t e0 = b0 ? d1 : d0, e1 = b0 ? d2 : d1, e2 = b0 ? d3 : d2, e3 = b0 ? d0 : d3;
    const bool b1 = cq & 2;
    d0 = b1 ? e2 : e0; d1 = b1 ? e3 : e1; d2 = b1 ? e0 : e2; d3 = b1 ? e1 : e3;
    out[r] = make_uint4(d0, d1, d2, d3);
}

__global__ __launch_bounds__(kBlock) void pad_codes_kernel(const uint8_t *__restrict__ codes, int M, uint4 *__restrict__ out, int64_t row0, int64_t n)
{
    const int64_t r = row0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    uint32_t w[4] = { 0u, 0u, 0u, 0u };
    for (int b = 0; b < M; ++b) w[b >> 2] |= (uint32_t)codes[r * M + b] << (8 * (b & 3));
    out[r] = make_uint4(w[0], w[1], w[2], w[3]);
}
int launch_pad_codes(const uint8_t *codes, int M, uint8_t *codes16, int64_t row0, int64_t n, hipStream_t st)
{
    if (n <= row0) return CVTMI_OK;
    if (M < 1 || M > 16) return fail(CVTMI_EINVAL, "pad_codes: M=%d", M);
    const int64_t blocks = (n - row0 + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "pad_codes: too many rows");
    hipLaunchKernelGGL(pad_codes_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, codes, M, reinterpret_cast<uint4 *>(codes16), row0, n);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_rotate_codes(const uint8_t *codes, uint8_t *codes_rot, int64_t row0, int64_t n, hipStream_t st)
{
    if (n <= row0) return CVTMI_OK;
    const int64_t blocks = (n - row0 + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "rotate_codes: too many rows");
    hipLaunchKernelGGL(rotate_codes_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, reinterpret_cast<const uint4 *>(codes),
                       reinterpret_cast<uint4 *>(codes_rot), row0, n);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_adc_scan(const OpqModelDev &m, const uint8_t *codes, int64_t n_rows, int64_t id_base, const float *q_rot,
                    int64_t nq, int k, const ScanPlan &plan, float *part_d, int64_t *part_id, float *lut_scratch,
                    const uint8_t *codes_rot, hipStream_t st, uint32_t *gthr, int lazy, float *final_d, int64_t *final_id, const uint32_t *only)
{
    if (nq <= 0) return CVTMI_OK;
    if (k < 1 || k > kBigK) return fail(CVTMI_EUNSUPPORTED, "adc_scan: k=%d outside 1..%d", k, kBigK);
    if (m.K > 256 || m.K < 1) return fail(CVTMI_EUNSUPPORTED, "adc_scan: K=%d outside 1..256", m.K);
    if (n_rows > 0xfffffffeLL) return fail(CVTMI_EUNSUPPORTED, "adc_scan: more than 2^32-2 rows per shard");
    if (nq > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "adc_scan: nq too large");
    if (m.D * 8 * (int)sizeof(float) > SCAN_CAP * 8 * 1) {
        // residual scratch aliases the selection buffer: QT*D floats must fit QT*CAP*8 bytes
        if (m.D > SCAN_CAP * 2) return fail(CVTMI_EUNSUPPORTED, "adc_scan: D=%d too large", m.D);
    }
    ScanArgs a;
    a.codes = codes; a.n_rows = n_rows; a.id_base = id_base; a.q_rot = q_rot; a.nq = (int)nq;
    a.books = m.books; a.centroid = m.coarse; a.D = m.D; a.step = m.step; a.K = m.K; a.k = k;
    a.splits = plan.splits;
    a.groups = (int)((nq + plan.qtile - 1) / plan.qtile);
    // split boundaries on whole tiles so that every workgroup's rows are 16-byte-row aligned tiles
    int64_t rps = (n_rows + plan.splits - 1) / plan.splits;
    const int64_t tile_rows = 2048;  // multiple of every kernel's tile (NT x R rows)
    rps = ((rps + tile_rows - 1) / tile_rows) * tile_rows;
    if (rps < tile_rows) rps = tile_rows;
    a.rows_per_split = rps;
    a.groups_a = a.groups; a.splits_b = 0; a.stride = plan.splits; a.rows_per_split_b = rps;
    a.part_d = part_d; a.part_id = part_id; a.out_d = nullptr; a.out_id = nullptr; a.lut_g = lut_scratch; a.codes_rot = codes_rot;
    a.gthr = nullptr; a.lazy = lazy; a.seed = g_scan_seed; a.only = only;
    if (plan.variant >= 3 && m.M == 16 && plan.qtile == 8) {
        if (!lut_scratch) return fail(CVTMI_EINVAL, "adc_scan16q: table scratch missing");
        if (plan.real_M > 0) {   // M < 16 through these kernels: the model's own tables, all-zero ones behind them (codes = the padded rows)
            OpqModelDev mr = m;
            mr.M = plan.real_M;
            CVTMI_TRY(launch_lut(mr, q_rot, nq, nullptr, lut_scratch, st, 256, 16));
        } else
        CVTMI_TRY(launch_lut(m, q_rot, nq, nullptr, lut_scratch, st, 256));  // [nq][16][256] fp32 (+inf past K), once per query
        int64_t blocks = (int64_t)a.groups * a.splits;
        if (plan.splits_b > plan.splits && plan.groups_a > 0 && plan.groups_a < a.groups) {
            a.groups_a = plan.groups_a; a.splits_b = plan.splits_b; a.stride = plan.stride();
            int64_t rb = (n_rows + plan.splits_b - 1) / plan.splits_b;
            rb = ((rb + tile_rows - 1) / tile_rows) * tile_rows;
            a.rows_per_split_b = rb < tile_rows ? tile_rows : rb;
            blocks = (int64_t)a.groups_a * a.splits + (int64_t)(a.groups - a.groups_a) * a.splits_b;
        }
        if (final_d && final_id && scan_in_place_queries(plan, m.M, nq) > 0) {   // (one region without splits: groups_a == groups, all in place)
            a.out_d = final_d; a.out_id = final_id;
        }
        if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "adc_scan: grid too large (%lld)", (long long)blocks);
        if (gthr && a.stride > 1) {  // the row splits of a query share their filter threshold
            CVTMI_HIP(hipMemsetAsync(gthr, 0xff, (size_t)nq * sizeof(uint32_t), st));
            a.gthr = gthr;
        }
        if (plan.packed) return launch_adc_scan16p(a, plan.real_M, blocks, st);   // (codes = the model's own rows, codes_rot = their packed rotation)
        const dim3 g((unsigned)blocks);
        if (codes_rot) {
            if (plan.variant == 3) hipLaunchKernelGGL((adc_scan16q_kernel<1024, 1, true>), g, dim3(1024), 0, st, a);
            else if (plan.variant == 4) hipLaunchKernelGGL((adc_scan16q_kernel<512, 2, true>), g, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((adc_scan16a_kernel<1024, 1, true>), g, dim3(1024), 0, st, a);
        } else {
            if (plan.variant == 3) hipLaunchKernelGGL((adc_scan16q_kernel<1024, 1, false>), g, dim3(1024), 0, st, a);
            else if (plan.variant == 4) hipLaunchKernelGGL((adc_scan16q_kernel<512, 2, false>), g, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((adc_scan16a_kernel<1024, 1, false>), g, dim3(1024), 0, st, a);
        }
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    if (plan.variant >= 1 && m.M == 16 && plan.qtile == 4) {
        const int64_t blocks = (int64_t)a.groups * a.splits;
        if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "adc_scan: grid too large (%lld)", (long long)blocks);
        if (plan.variant == 2) hipLaunchKernelGGL((adc_scan16_kernel<1024>), dim3((unsigned)blocks), dim3(1024), 0, st, a);
        else hipLaunchKernelGGL((adc_scan16_kernel<512>), dim3((unsigned)blocks), dim3(512), 0, st, a);
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    switch (m.M) {
        case 16: return launch_m<16>(a, plan.qtile, st);
        case 8: return launch_m<8>(a, plan.qtile, st);
        case 4: return launch_m<4>(a, plan.qtile, st);
        default: break;
    }
    return fail(CVTMI_EUNSUPPORTED, "adc_scan: M=%d not built (4, 8, 16)", m.M);
}

#ifdef CVTMI_SCAN_TIMING
extern "C" int cvtmi_debug_scan_timing(unsigned long long *out, int reset)
{
    unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_dbg), sizeof z) != hipSuccess) return -3;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_scan_dbg), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
extern "C" int cvtmi_debug_scan_async(unsigned long long *out, int reset)
{
    unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_dbg2), sizeof z) != hipSuccess) return -3;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_scan_dbg2), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
extern "C" int cvtmi_debug_topk(unsigned long long *out, int reset)
{
    unsigned long long z[4] = { 0, 0, 0, 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_topk_dbg), sizeof z) != hipSuccess) return -3;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_topk_dbg), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
extern "C" int cvtmi_debug_scan_trace(unsigned long long *out, int n_blocks)
{
    if (n_blocks > 16384) n_blocks = 16384;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_trace), (size_t)n_blocks * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

}  // namespace cvtmi
