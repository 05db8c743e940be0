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
#include "block_topk.h"
#include "kernels.h"

namespace cvtmi {

constexpr int SCAN_CAP = 384;   // selection buffer entries per query
constexpr int SCAN_TRIG = 256;  // compaction is requested beyond this fill
// code rows per lane per tile: 4 where the register file allows it, 2 for the widest variants
__host__ __device__ constexpr int scan_rows(int M, int QT) { return (M * QT > 32) ? 2 : 4; }

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
    float *part_d;
    int64_t *part_id;
};

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

template <int M, int QT>
__global__ __launch_bounds__(kBlock, ScanOcc<QT>::waves) void adc_scan_kernel(const ScanArgs a)
{
    using Row = typename CodeRow<M>::type;
    constexpr int R = scan_rows(M, QT);
    __shared__ __attribute__((aligned(32))) float lut[M * 256 * QT];
    __shared__ TopKShared<QT, SCAN_CAP> tk;

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
        topk_tile<QT, R, SCAN_CAP, SCAN_TRIG>(tk, a.k, tile, key, pay);
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

// Row ids travel as 32-bit payloads: one launch covers at most 2^32-1 rows.
ScanPlan plan_scan(const OpqModelDev &m, int64_t n_rows, int64_t nq, int k, int want_qtile, int want_splits)
{
    ScanPlan p;
    int qt = want_qtile;
    if (qt != 1 && qt != 2 && qt != 4 && qt != 8) qt = (nq >= 4) ? 4 : (nq >= 2 ? 2 : 1);
    if (qt == 8 && m.M == 16) qt = 4;  // 8 x 16 KB tables + buffers would leave one workgroup per CU
    p.qtile = qt;
    const int64_t groups = (nq + qt - 1) / qt;
    int s = want_splits;
    if (s <= 0) {
        // enough workgroups to fill 256 CUs x 2 several times over, a multiple of 8 so that each XCD
        // keeps to its own slice of the codes, but never fewer than ~4K rows per workgroup.
        const int64_t target = 4096;
        int64_t need = (target + groups - 1) / groups;
        need = ((need + 7) / 8) * 8;
        int64_t max_by_rows = n_rows / 4096;
        if (max_by_rows < 1) max_by_rows = 1;
        if (need > max_by_rows) need = max_by_rows >= 8 ? (max_by_rows / 8) * 8 : max_by_rows;
        if (need < 1) need = 1;
        if (need > 4096) need = 4096;
        s = (int)need;
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

int launch_adc_scan(const OpqModelDev &m, const uint8_t *codes, int64_t n_rows, int64_t id_base, const float *q_rot,
                    int64_t nq, int k, const ScanPlan &plan, float *part_d, int64_t *part_id, hipStream_t st)
{
    if (nq <= 0) return CVTMI_OK;
    if (k < 1 || k > 128) return fail(CVTMI_EUNSUPPORTED, "adc_scan: k=%d outside 1..128", k);
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
    const int64_t tile_rows = (int64_t)kBlock * 4;
    rps = ((rps + tile_rows - 1) / tile_rows) * tile_rows;
    if (rps < tile_rows) rps = tile_rows;
    a.rows_per_split = rps;
    a.part_d = part_d; a.part_id = part_id;
    switch (m.M) {
        case 16: return launch_m<16>(a, plan.qtile, st);
        case 8: return launch_m<8>(a, plan.qtile, st);
        case 4: return launch_m<4>(a, plan.qtile, st);
        default: break;
    }
    return fail(CVTMI_EUNSUPPORTED, "adc_scan: M=%d not built (4, 8, 16)", m.M);
}

}  // namespace cvtmi
