// flat.hip -- exhaustive nearest-neighbour search over un-compressed rows:
// hnswlib::BruteforceSearch<dist_t>::searchKnn (brute_force_search/src/brutoforce.hpp:73-93) with
//   InnerProductSpace (space_ip.hpp:211-239)      dist = 1 - sum q*x          fp32
//   L2Space           (space_l2.h:153-184)         dist = sum (q-x)^2          fp32
//   L2SpaceI          (space_l2.h:186-245)         dist = sum (q-x)^2          uint8 -> int32
//
// Summation order (so that fp32 distances are BIT-EXACT against the reference as its own build
// flags compile it, see oracle/Makefile):
//   IP,  D % 4 == 0 : the SSE branch (space_ip.hpp:84-131, :168-206): four lane accumulators,
//                     acc[l] += q[i+l]*x[i+l] (separate mul/add), result 1 - (((a0+a1)+a2)+a3)
//   L2F, D % 16 == 0: the AVX branch hard-enabled by `#define USE_AVX` (space_l2.h:12, :46-73):
//                     eight lane accumulators, summed left to right
//   L2F, D % 4 == 0 : L2SqrSIMD4Ext (:123-151): four lanes
//   otherwise       : the scalar loops (space_ip.hpp:25-34, space_l2.h:26-37)
//   uint8           : exact integers, any order; groups of four bytes, dim % 4 tail dropped (:198-215)
//
// Round-1 mapping: one lane per row, QT queries staged in LDS share every row read; selection by
// the shared LDS top-k (block_topk.h).  The uint8 metric runs on v_dot4_u32_u8:
// sum (q-x)^2 = |q|^2 + |x|^2 - 2<q,x>, all three exact in 32-bit integers (<= 512*255^2).
#include "block_topk.h"
#include "kernels.h"
#include "dist_f32.h"

namespace cvtmi {

constexpr int FLAT_CAP = 384;
constexpr int FLAT_TRIG = 256;
constexpr int FLAT_R = 2;

struct FlatArgs {
    const void *data;
    int64_t n;
    const void *q;
    int nq;
    int D, k, splits;
    int64_t rows_per_split;
    float *part_d;
    int64_t *part_id;
    const uint32_t *only_if;   // [nq] or null: a workgroup none of whose queries is flagged exits at once
};

template <int QT>
__device__ __forceinline__ bool flat_skip(const FlatArgs &a, int group)
{
    if (!a.only_if) return false;
    uint32_t any = 0;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi < a.nq) any |= a.only_if[qi];
    }
    return any == 0;
}

template <bool IP, int LANES, int QT, int CAP = FLAT_CAP, int TRIG = FLAT_TRIG>
__global__ __launch_bounds__(kBlock) void flat_f32_kernel(const FlatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [QT][D]
    __shared__ TopKShared<QT, CAP> tk;
    const int split = blockIdx.x % a.splits, group = blockIdx.x / a.splits;
    if (flat_skip<QT>(a, group)) return;
    const int tid = threadIdx.x;
    const float *Q = reinterpret_cast<const float *>(a.q);
    for (int i = tid; i < QT * a.D; i += kBlock) {
        const int q = i / a.D, d = i - q * a.D;
        int qi = group * QT + q;
        qi = qi < a.nq ? qi : a.nq - 1;
        qs[i] = Q[(int64_t)qi * a.D + d];
    }
    topk_init(tk);
    __syncthreads();
    const int64_t row_begin = (int64_t)split * a.rows_per_split;
    int64_t row_end = row_begin + a.rows_per_split;
    row_end = row_end < a.n ? row_end : a.n;
    const float *X = reinterpret_cast<const float *>(a.data);
    int tile = 0;
    for (int64_t base = row_begin; base < row_end; base += (int64_t)kBlock * FLAT_R, ++tile) {
        uint32_t key[FLAT_R][QT];
        uint32_t pay[FLAT_R];
#pragma unroll
        for (int r = 0; r < FLAT_R; ++r) {
            const int64_t row = base + r * kBlock + tid;
            const bool valid = row < row_end;
            float d[QT];
            dist_f32_row<IP, LANES, QT>(X + (valid ? row : row_begin) * a.D, qs, a.D, d);
            pay[r] = (uint32_t)row;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                uint32_t kk = f32_key(d[q]);
                kk = kk == KEY_MAX ? KEY_MAX - 1 : kk;
                key[r][q] = valid ? kk : KEY_MAX;
            }
        }
        topk_tile<QT, FLAT_R, CAP, TRIG>(tk, a.k, tile, key, pay);
    }
    __syncthreads();
    topk_compact(tk, a.k);
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi >= a.nq) break;
        const int cnt = tk.cnt[q];
        const int64_t o = ((int64_t)qi * a.splits + split) * a.k;
        for (int i = tid; i < a.k; i += kBlock) {
            if (i < cnt) {
                const unsigned long long e = tk.buf[q][i];
                a.part_d[o + i] = key_f32((uint32_t)(e >> 32));
                a.part_id[o + i] = (int64_t)(uint32_t)e;
            } else {
                a.part_d[o + i] = __uint_as_float(0x7f800000u);
                a.part_id[o + i] = -1;
            }
        }
    }
}

// Same search with the queries in SCALAR registers.  The kernel above re-reads every query value from LDS for
// every row (QT * D / 4 broadcast ds_read_b128 per row and lane: the LDS pipe, not the VALU, sets its pace).
// A query value is the same for all 64 lanes, so it belongs in an SGPR: the queries are read with uniform
// addresses straight from global memory (s_load through the scalar cache), the rows stay one per lane, and the
// multiply / subtract take the SGPR as an operand -- no LDS traffic at all in the distance loop.  Same
// per-lane accumulation order as dist_f32.h, hence the same bits.
template <bool IP, int LANES, int QT, int CAP = FLAT_CAP, int TRIG = FLAT_TRIG>
__global__ __launch_bounds__(kBlock) void flat_f32_sq_kernel(const FlatArgs a)
{
    __shared__ TopKShared<QT, CAP> tk;
    const int split = blockIdx.x % a.splits, group = blockIdx.x / a.splits;
    if (flat_skip<QT>(a, group)) return;
    const int tid = threadIdx.x;
    // constant address space + uniform address = s_load_dword*: the values land in SGPRs
    typedef const __attribute__((address_space(4))) float cfloat;
    cfloat *Q = (cfloat *)(uintptr_t)a.q;
    cfloat *qp[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        int qi = group * QT + q;
        qi = qi < a.nq ? qi : a.nq - 1;
        qp[q] = Q + (int64_t)qi * a.D;  // wave-uniform
    }
    topk_init(tk);
    __syncthreads();
    const int64_t row_begin = (int64_t)split * a.rows_per_split;
    int64_t row_end = row_begin + a.rows_per_split;
    row_end = row_end < a.n ? row_end : a.n;
    const float *X = reinterpret_cast<const float *>(a.data);
    const int D = a.D;
    int tile = 0;
    for (int64_t base = row_begin; base < row_end; base += (int64_t)kBlock * FLAT_R, ++tile) {
        uint32_t key[FLAT_R][QT];
        uint32_t pay[FLAT_R];
        // blocked layout (flat_block_kernel): float4 c of the 64 rows of block b are contiguous, so the 64 lanes
        // of a wave read 1 KB in one piece; row = 64 b + lane (tiles start on multiples of 64)
        const float4 *rowp[FLAT_R];
        bool valid[FLAT_R];
#pragma unroll
        for (int r = 0; r < FLAT_R; ++r) {
            const int64_t row = base + r * kBlock + tid;
            valid[r] = row < row_end;
            const int64_t rc = valid[r] ? row : row_begin;  // rows past the end read a valid row, rejected by the key
            rowp[r] = reinterpret_cast<const float4 *>(X) + (rc >> 6) * (int64_t)(D >> 2) * 64 + (rc & 63);
            pay[r] = (uint32_t)row;
        }
        float acc[FLAT_R][QT][LANES];
#pragma unroll
        for (int r = 0; r < FLAT_R; ++r)
#pragma unroll
            for (int q = 0; q < QT; ++q)
#pragma unroll
                for (int l = 0; l < LANES; ++l) acc[r][q][l] = 0.0f;
        // unrolled so that the scalar loads of 16 dimensions x QT queries (64 SGPRs) are in flight together
#pragma unroll(16 / LANES)
        for (int i = 0; i < D; i += LANES) {  // D % LANES == 0 (the launcher picks LANES)
            float xv[FLAT_R][LANES];
#pragma unroll
            for (int r = 0; r < FLAT_R; ++r)
#pragma unroll
                for (int l4 = 0; l4 < LANES / 4; ++l4) {
                    const float4 v = rowp[r][(int64_t)((i >> 2) + l4) * 64];
                    xv[r][4 * l4 + 0] = v.x; xv[r][4 * l4 + 1] = v.y; xv[r][4 * l4 + 2] = v.z; xv[r][4 * l4 + 3] = v.w;
                }
#pragma unroll
            for (int q = 0; q < QT; ++q) {
#pragma unroll
                for (int l = 0; l < LANES; ++l) {
                    const float qv = qp[q][i + l];  // uniform address: scalar load
#pragma unroll
                    for (int r = 0; r < FLAT_R; ++r) {
                        if constexpr (IP) {
                            acc[r][q][l] = __fadd_rn(acc[r][q][l], __fmul_rn(qv, xv[r][l]));
                        } else {
                            const float t = __fsub_rn(qv, xv[r][l]);
                            acc[r][q][l] = __fadd_rn(acc[r][q][l], __fmul_rn(t, t));
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < FLAT_R; ++r)
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                float sum = acc[r][q][0];
#pragma unroll
                for (int l = 1; l < LANES; ++l) sum = __fadd_rn(sum, acc[r][q][l]);
                const float d = IP ? __fsub_rn(1.0f, sum) : sum;
                uint32_t kk = f32_key(d);
                kk = kk == KEY_MAX ? KEY_MAX - 1 : kk;
                key[r][q] = valid[r] ? kk : KEY_MAX;
            }
        topk_tile<QT, FLAT_R, CAP, TRIG>(tk, a.k, tile, key, pay);
    }
    __syncthreads();
    topk_compact(tk, a.k);
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi >= a.nq) break;
        const int cnt = tk.cnt[q];
        const int64_t o = ((int64_t)qi * a.splits + split) * a.k;
        for (int i = tid; i < a.k; i += kBlock) {
            if (i < cnt) {
                const unsigned long long e = tk.buf[q][i];
                a.part_d[o + i] = key_f32((uint32_t)(e >> 32));
                a.part_id[o + i] = (int64_t)(uint32_t)e;
            } else {
                a.part_d[o + i] = __uint_as_float(0x7f800000u);
                a.part_id[o + i] = -1;
            }
        }
    }
}

// rows [row0, row0 + n) of a row-major [n][D] fp32 block -> the blocked layout above (D % 4 == 0):
// float4 index of (row r, float4 c) = ((r >> 6) * (D / 4) + c) * 64 + (r & 63)
__global__ __launch_bounds__(kBlock) void flat_block_kernel(const float4 *__restrict__ src, int64_t n, int D4, int64_t row0,
                                                            float4 *__restrict__ dst)
{
    const int64_t total = n * D4;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t r = e / D4;
        const int c = (int)(e - r * D4);
        const int64_t row = row0 + r;
        dst[((row >> 6) * D4 + c) * 64 + (row & 63)] = src[e];  // coalesced read, 16-byte scattered write (add time only)
    }
}

int launch_flat_block(const float *src, int64_t n, int D, int64_t row0, float *dst, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    int64_t blocks = (n * (D / 4) + kBlock - 1) / kBlock;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(flat_block_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, reinterpret_cast<const float4 *>(src), n, D / 4,
                       row0, reinterpret_cast<float4 *>(dst));
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// the inverse, for the threshold filter's exact finish (round 6): rows [row0 / 64 * 64, n) of the blocked layout -> row-major dst.  A workgroup moves 16
// pieces of a 64-row block through LDS: 1 KB reads, 256-byte writes
__global__ __launch_bounds__(kBlock) void flat_unblock_kernel(const float4 *__restrict__ src, int64_t b0, int64_t n, int D4, float4 *__restrict__ dst)
{
    __shared__ float4 t_s[64][17];
    const int cgroups = (D4 + 15) / 16;
    const int64_t b = b0 + blockIdx.x / cgroups;
    const int c0 = (int)(blockIdx.x % cgroups) * 16;
    for (int i = threadIdx.x; i < 64 * 16; i += kBlock) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < D4) t_s[r][c] = src[(b * D4 + c0 + c) * 64 + r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += kBlock) {
        const int r = i >> 4, c = i & 15;
        const int64_t row = b * 64 + r;
        if (c0 + c < D4 && row < n) dst[row * D4 + c0 + c] = t_s[r][c];
    }
}
int launch_flat_unblock(const float *blocked, int64_t row0, int64_t n, int D, float *dst, hipStream_t st)
{
    const int64_t b0 = row0 / 64, b1 = (n + 63) / 64;
    if (b1 <= b0) return CVTMI_OK;
    const int64_t blocks = (b1 - b0) * ((D / 4 + 15) / 16);
    if (blocks > 0x7fffffffLL) return fail(CVTMI_EINVAL, "flat_unblock: too many blocks");
    hipLaunchKernelGGL(flat_unblock_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, reinterpret_cast<const float4 *>(blocked), b0, n, D / 4,
                       reinterpret_cast<float4 *>(dst));
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// uint8 rows.  VEC: rows are 16-byte aligned multiples of 16 (D % 16 == 0) -> dwordx4 loads + dot4
template <bool VEC, int QT, int CAP = FLAT_CAP, int TRIG = FLAT_TRIG>
__global__ __launch_bounds__(kBlock) void flat_u8_kernel(const FlatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t qw[];  // [QT][ceil(D/4)] packed query bytes
    __shared__ TopKShared<QT, CAP> tk;
    __shared__ uint32_t qq[QT];
    const int split = blockIdx.x % a.splits, group = blockIdx.x / a.splits;
    if (flat_skip<QT>(a, group)) return;   // (round 6: the uint8 threshold filter's flagged queries come here under a predicate)
    const int tid = threadIdx.x;
    const int Dq = (a.D >> 2) << 2;  // the reference drops a dim % 4 tail (space_l2.h:198)
    const int W = Dq >> 2;           // 32-bit words per row that take part
    const uint8_t *Q = reinterpret_cast<const uint8_t *>(a.q);
    for (int i = tid; i < QT * W; i += kBlock) {
        const int q = i / W, w = i - q * W;
        int qi = group * QT + q;
        qi = qi < a.nq ? qi : a.nq - 1;
        const uint8_t *p = Q + (int64_t)qi * a.D + 4 * w;
        qw[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    }
    topk_init(tk);
    __syncthreads();
    if (tid < QT) {
        uint32_t s = 0;
        for (int w = 0; w < W; ++w) s = __builtin_amdgcn_udot4(qw[tid * W + w], qw[tid * W + w], s, false);
        qq[tid] = s;
    }
    __syncthreads();
    const int64_t row_begin = (int64_t)split * a.rows_per_split;
    int64_t row_end = row_begin + a.rows_per_split;
    row_end = row_end < a.n ? row_end : a.n;
    const uint8_t *X = reinterpret_cast<const uint8_t *>(a.data);
    int tile = 0;
    for (int64_t base = row_begin; base < row_end; base += (int64_t)kBlock * FLAT_R, ++tile) {
        uint32_t key[FLAT_R][QT];
        uint32_t pay[FLAT_R];
#pragma unroll
        for (int r = 0; r < FLAT_R; ++r) {
            const int64_t row = base + r * kBlock + tid;
            const bool valid = row < row_end;
            const uint8_t *xr = X + (valid ? row : row_begin) * a.D;
            uint32_t xx = 0, qx[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) qx[q] = 0;
            if constexpr (VEC) {
                for (int w = 0; w < W; w += 4) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(xr + 4 * w);
                    const uint32_t xv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xx = __builtin_amdgcn_udot4(xv[e], xv[e], xx, false);
#pragma unroll
                        for (int q = 0; q < QT; ++q) qx[q] = __builtin_amdgcn_udot4(qw[q * W + w + e], xv[e], qx[q], false);
                    }
                }
            } else {
                for (int w = 0; w < W; ++w) {
                    const uint8_t *p = xr + 4 * w;
                    const uint32_t xv = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
                    xx = __builtin_amdgcn_udot4(xv, xv, xx, false);
#pragma unroll
                    for (int q = 0; q < QT; ++q) qx[q] = __builtin_amdgcn_udot4(qw[q * W + w], xv, qx[q], false);
                }
            }
            pay[r] = (uint32_t)row;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const uint32_t d = qq[q] + xx - 2u * qx[q];  // exact: every term < 2^26
                key[r][q] = valid ? d : KEY_MAX;
            }
        }
        topk_tile<QT, FLAT_R, CAP, TRIG>(tk, a.k, tile, key, pay);
    }
    __syncthreads();
    topk_compact(tk, a.k);
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi >= a.nq) break;
        const int cnt = tk.cnt[q];
        const int64_t o = ((int64_t)qi * a.splits + split) * a.k;
        for (int i = tid; i < a.k; i += kBlock) {
            if (i < cnt) {
                const unsigned long long e = tk.buf[q][i];
                a.part_d[o + i] = __uint_as_float((uint32_t)(e >> 32));  // int32 distance bits
                a.part_id[o + i] = (int64_t)(uint32_t)e;
            } else {
                a.part_d[o + i] = __uint_as_float(0x7f800000u);
                a.part_id[o + i] = -1;
            }
        }
    }
}

// ---- selection behind the streaming matrix-core kernel (flat_u8_mstream_kernel, flat_mfma.hip; 1 .. 128 queries) ---------------
// That kernel streams the rows once and keeps only MINIMA of the exact distances: per 32-row tile and query (tmin[tile][query]) and
// per wave and query (gmin[query][wave]).  The k-th smallest wave minimum, theta, bounds the true k-th distance from above (the k
// smallest minima are k distinct rows), and with a thousand waves against k <= 128 the tiles whose minimum is <= theta are the ~k
// that hold the answer plus the rare collision.  The finish kernel derives theta, revisits only those tiles and recomputes their 32
// distances from the rows.  Ties at theta are all revisited, so the result is exact whatever the data (all-equal distances degrade
// to a pass over the tile minima and a recomputation of every row, not to a wrong answer).
// grid = nq * S workgroups; workgroup (q, s) derives theta from all G wave minima of its query, then selects among the rows the
// qualifying waves wrote inside ITS slice of the rows.  Wave w owns rows (w + t G) rpi + j, t = 0.., j < rpi; the slice's entries are fed
// in ascending row order (t outermost, then wave, then j), which is the order block_topk.h's tie rule asks for.
constexpr int FIN_CAP = 1024, FIN_TRIG = 768, FIN_R = 4, FIN_MAXG = 8192, FIN_MAXR = 160, FIN_CL = 1024;
struct StreamFinishArgs {
    int64_t n;
    const int32_t *gmin;
    int G, S, rpi_log2, k;
    float *part_d; int64_t *part_id;
    const int32_t *tmin; int nqp; int tile_group; const uint8_t *X; const uint8_t *Q; int D;   // tmin[((round / tile_group) * G + wave) * nqp + q]
};
__global__ __launch_bounds__(kBlock) void flat_u8_stream_finish_kernel(const StreamFinishArgs a)
{
    __shared__ TopKShared<1, FIN_CAP> tk;
    __shared__ uint16_t ql[FIN_MAXG];          // qualifying waves, ascending
    __shared__ int r_first[FIN_MAXR], r_off[FIN_MAXR + 1];
    __shared__ int qn_s;
    __shared__ int32_t cl_s[FIN_CL];             // this slice's chunks whose tile minimum is under theta, ascending
    __shared__ __attribute__((aligned(16))) uint32_t q_s[128];   // the query's bytes
    const int64_t n = a.n;
    const int G = a.G, S = a.S, rpi_log2 = a.rpi_log2, k = a.k;
    const int32_t *gmin = a.gmin;
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x / S;
    const int sl = blockIdx.x % S;
    int qq_t = 0;
    {
        if (tid < a.D / 4) q_s[tid] = reinterpret_cast<const uint32_t *>(a.Q + q * a.D)[tid];
        __syncthreads();
        for (int w = 0; w < a.D / 4; ++w) qq_t = (int)__builtin_amdgcn_udot4(q_s[w], q_s[w], (uint32_t)qq_t, false);
    }
    const int32_t *gm = gmin + q * G;
    // theta = the k-th smallest of the G wave minima, in two rounds of the same bound: the k-th smallest of the 256 per-thread minima
    // (k <= 128 distinct waves) is an upper bound theta1 that already excludes nearly every wave; the few minima under it are
    // gathered in LDS and ONE wave selects among them in registers (radix select on lane masks, block_topk.h) -- the two LDS
    // bitonic sorts this used to take were most of the kernel (64 queries over 10 M rows: 112 us of a 0.88 ms search).
    uint32_t theta;
    {
        uint32_t *sel_v = reinterpret_cast<uint32_t *>(&tk.buf[0][0]);   // 1024 words of the (not yet used) selection buffer
        uint32_t mymin = KEY_MAX;
        for (int i = tid; i < G; i += kBlock) {
            const uint32_t v = (uint32_t)gm[i];
            mymin = v < mymin ? v : mymin;
        }
        sel_v[tid] = mymin;
        if (tid == 0) qn_s = 0;
        __syncthreads();
        if (tid < 64) {
            unsigned long long e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = r * 64 + tid;
                e[r] = p < G ? ((unsigned long long)sel_v[p] << 32) | (uint32_t)p : ~0ull;   // (thread p has an element iff p < G)
            }
            const uint32_t t1 = (G < kBlock ? G : kBlock) >= k ? (uint32_t)(wave_select<4>(e, k) >> 32) : KEY_MAX;
            if (tid == 0) r_first[0] = (int)t1;
        }
        __syncthreads();
        const uint32_t theta1 = (uint32_t)r_first[0];
        __syncthreads();   // (sel_v is written again below)
        for (int i = tid; i < G; i += kBlock) {
            const uint32_t v = (uint32_t)gm[i];
            if (v <= theta1) {
                const int pos = atomicAdd(&qn_s, 1);
                if (pos < FIN_CAP) sel_v[pos] = v;
            }
        }
        __syncthreads();
        const int n2 = qn_s;
        if (n2 <= FIN_CAP) {   // workgroup-uniform; the usual case by far (n2 is about k)
            if (tid < 64) {
                unsigned long long e[FIN_CAP / 64];
#pragma unroll
                for (int r = 0; r < FIN_CAP / 64; ++r) {
                    const int p = r * 64 + tid;
                    e[r] = p < n2 ? ((unsigned long long)sel_v[p] << 32) | (uint32_t)p : ~0ull;
                }
                const uint32_t t = n2 >= k ? (uint32_t)(wave_select<FIN_CAP / 64>(e, k) >> 32) : KEY_MAX;   // fewer than k waves: everything qualifies
                if (tid == 0) r_first[0] = (int)t;
            }
            __syncthreads();
            theta = (uint32_t)r_first[0];
            __syncthreads();
        } else {               // masses of equal minima: the sorting path
            __syncthreads();
            topk_init(tk);
            __syncthreads();
            int tile = 0;
            for (int base = 0; base < G; base += kBlock * FIN_R, ++tile) {
                uint32_t key[FIN_R][1], pay[FIN_R];
#pragma unroll
                for (int r = 0; r < FIN_R; ++r) {
                    const int i = base + r * kBlock + tid;
                    pay[r] = (uint32_t)i;
                    const uint32_t v = i < G ? (uint32_t)gm[i] : KEY_MAX;
                    key[r][0] = v <= theta1 ? v : KEY_MAX;
                }
                topk_tile<1, FIN_R, FIN_CAP, FIN_TRIG>(tk, k, tile, key, pay);
            }
            __syncthreads();
            topk_compact(tk, k);
            theta = tk.thr[0];
            __syncthreads();
        }
    }
    // all qualifying waves, ascending (wave 0: one ballot per 64 waves)
    if (tid < 64) {
        int cnt = 0;
        for (int c = 0; c < G; c += 64) {
            const int g = c + tid;
            const bool yes = g < G && (uint32_t)gm[g] <= theta;
            const unsigned long long m = __ballot(yes);
            if (yes) ql[cnt + __popcll(m & ((1ull << tid) - 1))] = (uint16_t)g;
            cnt += __popcll(m);
        }
        if (tid == 0) qn_s = cnt;
    }
    topk_init(tk);
    __syncthreads();
    const int qn = qn_s;
    // This workgroup's slice is a CONTIGUOUS range of row chunks (chunk c = rpi rows, written by wave c % G in its round c / G), so
    // the S lists of a query cover ascending row ranges and topk_merge_kernel's tie rule (position order = row order) holds.  Per
    // round of the slice, the qualifying waves inside the slice's window are a sub-range of ql (two binary searches).
    const int64_t c_total = (n + ((int64_t)1 << rpi_log2) - 1) >> rpi_log2;
    const int64_t c_lo = c_total * sl / S, c_hi = c_total * (sl + 1) / S;
    const int64_t t_a = c_lo / G;
    const int n_rounds = c_hi > c_lo ? (int)((c_hi - 1) / G - t_a + 1) : 0;   // <= FIN_MAXR: n < 2^31, G * rpi >= 2^16, S = 64
    if (tid < n_rounds) {
        const int64_t first = (t_a + tid) * G;
        const int lo_w = (int)(c_lo > first ? c_lo - first : 0), hi_w = (int)(c_hi - first < G ? c_hi - first : G);
        auto lower = [&](int w) { int a_ = 0, b_ = qn; while (a_ < b_) { const int m_ = (a_ + b_) >> 1; if ((int)ql[m_] < w) a_ = m_ + 1; else b_ = m_; } return a_; };
        const int i0 = lower(lo_w), i1 = lower(hi_w);
        r_first[tid] = i0;
        r_off[tid + 1] = i1 - i0;
    }
    __syncthreads();
    if (tid == 0) {
        r_off[0] = 0;
        for (int r = 0; r < n_rounds; ++r) r_off[r + 1] += r_off[r];
    }
    __syncthreads();
    // The slice's candidate chunks are the (round, qualifying wave) pairs; what decides is the TILE minimum of a chunk, and only
    // ~k / S of the slice's few hundred chunks have one under theta.  Wave 0 lists those first (in ascending order: ballots, no
    // atomics -- the selection below wants ascending rows), then the workgroup computes exact distances for the listed chunks' rows
    // only.  (Checking the tile minimum per ROW cost ten rounds of the selection protocol per slice: 78 us of a 0.86 ms search.)
    const int nch = n_rounds ? r_off[n_rounds] : 0;
    auto chunk_of = [&](int ci, int64_t &chunk, int64_t &tgrp) {
        int a_ = 0, b_ = n_rounds - 1;   // the round whose [r_off, r_off + 1) holds ci
        while (a_ < b_) { const int m_ = (a_ + b_ + 1) >> 1; if (r_off[m_] <= ci) a_ = m_; else b_ = m_ - 1; }
        const int wv = ql[r_first[a_] + (ci - r_off[a_])];
        chunk = (t_a + a_) * G + wv;
        tgrp = ((t_a + a_) / a.tile_group) * G + wv;
    };
    if (tid < 64) {
        int cnt = 0;
        for (int c0 = 0; c0 < nch; c0 += 64) {
            const int ci = c0 + tid;
            bool yes = false;
            int64_t chunk = 0, tgrp = 0;
            if (ci < nch) {
                chunk_of(ci, chunk, tgrp);
                yes = (chunk << rpi_log2) < n && (uint32_t)a.tmin[tgrp * a.nqp + q] <= theta;
            }
            const unsigned long long m = __ballot(yes);
            const int pos = cnt + __popcll(m & ((1ull << tid) - 1));
            if (yes && pos < FIN_CL) cl_s[pos] = (int32_t)chunk;   // chunk < 2^31 / 32
            cnt += __popcll(m);
        }
        if (tid == 0) qn_s = cnt;
    }
    __syncthreads();
    const int n_cl = qn_s;
    int tile = 0;
    if (n_cl <= FIN_CL) {   // workgroup-uniform
        const int64_t total = (int64_t)n_cl << rpi_log2;
        for (int64_t base = 0; base < total; base += kBlock * FIN_R, ++tile) {
            uint32_t key[FIN_R][1], pay[FIN_R];
#pragma unroll
            for (int r = 0; r < FIN_R; ++r) {
                const int64_t e = base + r * kBlock + tid;
                key[r][0] = KEY_MAX;
                pay[r] = 0;
                if (e < total) {
                    const int64_t row = ((int64_t)cl_s[e >> rpi_log2] << rpi_log2) + (e & ((1 << rpi_log2) - 1));
                    if (row < n) {   // exact sum (q - x)^2 = |q|^2 + |x|^2 - 2 <q, x>, every term < 2^27
                        const uint4 *xr = reinterpret_cast<const uint4 *>(a.X + row * a.D);
                        uint32_t xx = 0, qx = 0;
                        for (int c = 0; c < a.D / 16; ++c) {
                            const uint4 v = xr[c];
                            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                            for (int e2 = 0; e2 < 4; ++e2) {
                                xx = __builtin_amdgcn_udot4(w[e2], w[e2], xx, false);
                                qx = __builtin_amdgcn_udot4(w[e2], q_s[4 * c + e2], qx, false);
                            }
                        }
                        const uint32_t d = (uint32_t)qq_t + xx - 2u * qx;
                        pay[r] = (uint32_t)row;
                        if (d <= theta) key[r][0] = d;
                    }
                }
            }
            topk_tile<1, FIN_R, FIN_CAP, FIN_TRIG>(tk, k, tile, key, pay);
        }
    } else {               // a slice full of tiles under theta (masses of equal rows): every chunk, tile minimum checked per row
        const int64_t total = (int64_t)nch << rpi_log2;
        for (int64_t base = 0; base < total; base += kBlock * FIN_R, ++tile) {
            uint32_t key[FIN_R][1], pay[FIN_R];
#pragma unroll
            for (int r = 0; r < FIN_R; ++r) {
                const int64_t e = base + r * kBlock + tid;
                key[r][0] = KEY_MAX;
                pay[r] = 0;
                if (e < total) {
                    int64_t chunk, tgrp;
                    chunk_of((int)(e >> rpi_log2), chunk, tgrp);
                    const int64_t row = (chunk << rpi_log2) + (e & ((1 << rpi_log2) - 1));
                    if (row < n) {
                        uint32_t d = KEY_MAX;
                        if ((uint32_t)a.tmin[tgrp * a.nqp + q] <= theta) {
                            const uint4 *xr = reinterpret_cast<const uint4 *>(a.X + row * a.D);
                            uint32_t xx = 0, qx = 0;
                            for (int c = 0; c < a.D / 16; ++c) {
                                const uint4 v = xr[c];
                                const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                                for (int e2 = 0; e2 < 4; ++e2) {
                                    xx = __builtin_amdgcn_udot4(w[e2], w[e2], xx, false);
                                    qx = __builtin_amdgcn_udot4(w[e2], q_s[4 * c + e2], qx, false);
                                }
                            }
                            d = (uint32_t)qq_t + xx - 2u * qx;
                        }
                        pay[r] = (uint32_t)row;
                        if (d <= theta) key[r][0] = d;
                    }
                }
            }
            topk_tile<1, FIN_R, FIN_CAP, FIN_TRIG>(tk, k, tile, key, pay);
        }
    }
    __syncthreads();
    topk_compact(tk, k);
    const int cnt = tk.cnt[0];
    float *pd = a.part_d + ((int64_t)blockIdx.x) * k;
    int64_t *pi = a.part_id + ((int64_t)blockIdx.x) * k;
    for (int i = tid; i < k; i += kBlock) {
        if (i < cnt) {
            pd[i] = __uint_as_float((uint32_t)(tk.buf[0][i] >> 32));   // the int32 distance's bits: what the uint8 metric returns
            pi[i] = (int64_t)(uint32_t)tk.buf[0][i];
        } else {
            pd[i] = __uint_as_float(0x7f800000u);
            pi[i] = -1;
        }
    }
}

// ==========================================================================================
// uint8 L2 on the matrix cores (D % 32 == 0): sum (q-x)^2 = |q'|^2 + |x'|^2 - 2 <q',x'> with q' = q-128,
// x' = x-128 in [-128,127] (one v_xor with 0x80808080 per dword), every term exact in int32
// (<= 512*128^2*... < 2^31).  v_mfma_i32_32x32x32_i8: A = 32 queries x 32 dims, B = 32 rows x 32 dims;
// lane l supplies 16 consecutive dims (l>>5)*16.. of query / row (l&31) and receives, for ITS row, the dot
// products with 16 of the 32 queries.  A workgroup serves QT = 32*QB queries over one row split; a wave owns
// 32 rows per tile, rows come straight from HBM as 16-byte loads (next tile in flight while this one is
// multiplied), queries sit in LDS with a +16-byte row pad (conflict-free ds_read_b128).
// |x'|^2 per row is computed once when rows are added (flat_u8_norms_kernel).
// ==========================================================================================
typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(kBlock) void flat_u8_norms_kernel(const uint8_t *__restrict__ x, int64_t n, int D,
                                                               int32_t *__restrict__ norms)
{
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= n) return;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(x + row * D);  // D % 32 == 0
    int s = 0;
    for (int w = 0; w < D / 4; ++w) {
        const int v = (int)(p[w] ^ 0x80808080u);
        s = __builtin_amdgcn_sdot4(v, v, s, false);
    }
    norms[row] = s;
}

// D / 16 a power of two (D = 32 ... 512): a row is read by D / 16 adjacent lanes, 16 bytes each (coalesced), and
// folded with shuffles -- the one-thread-per-row form above walks memory at a D-byte stride (0.24 TB/s).
__global__ __launch_bounds__(kBlock) void flat_u8_norms_coalesced_kernel(const uint8_t *__restrict__ x, int64_t n, int D,
                                                                         int32_t *__restrict__ norms)
{
    const int lpr = D >> 4;  // lanes per row
    const int64_t f = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t row = f / lpr;
    int s = 0;
    if (row < n) {
        const uint4 v = reinterpret_cast<const uint4 *>(x)[f];
        const int w[4] = { (int)(v.x ^ 0x80808080u), (int)(v.y ^ 0x80808080u), (int)(v.z ^ 0x80808080u), (int)(v.w ^ 0x80808080u) };
#pragma unroll
        for (int c = 0; c < 4; ++c) s = __builtin_amdgcn_sdot4(w[c], w[c], s, false);
    }
    for (int o = lpr >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (row < n && (f & (lpr - 1)) == 0) norms[row] = s;
}

int launch_flat_u8_norms(const uint8_t *x, int64_t n, int D, int32_t *norms, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    const int lpr = D >> 4;
    if (D % 32 == 0 && D <= 1024 && (lpr & (lpr - 1)) == 0 && ((uintptr_t)x & 15) == 0) {
        const int64_t threads = n * lpr;
        hipLaunchKernelGGL(flat_u8_norms_coalesced_kernel, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           st, x, n, D, norms);
    } else {
        hipLaunchKernelGGL(flat_u8_norms_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, x, n, D, norms);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

struct FlatMfmaArgs {
    const uint8_t *data;
    const int32_t *norms;
    int64_t n;
    const uint8_t *q;
    int nq, D, k, splits;
    int64_t rows_per_split;
    float *part_d;
    int64_t *part_id;
    uint32_t *gthr;  // [nq] smallest k-th-best distance any row split of the query has reached (row-tile kernel)
    uint32_t *gslot; // [nq][nslot] slot j = smallest distance seen by any split with split % nslot == j (nslot = k, or 0)
    int nslot;
};

template <int QB, int CAP, int TRIG, int DMAX>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void flat_u8_mfma_kernel(const FlatMfmaArgs a)
{
    constexpr int QT = 32 * QB;
    constexpr int KS = DMAX / 32;  // k-steps held in registers per row (D <= DMAX)
    extern __shared__ __attribute__((aligned(16))) uint8_t q8s[];  // [QT][D + 16] query bytes ^ 0x80
    __shared__ TopKShared<QT, CAP> tk;
    __shared__ int qq[QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int group, split;
    {
        const int b = blockIdx.x;
        if ((a.splits & 7) == 0) {  // a row split stays on one XCD: the query groups of that split share its L2
            const int s8 = a.splits >> 3;
            const int xcd = b & 7, i = b >> 3;
            split = xcd + 8 * (i % s8);
            group = i / s8;
        } else {
            split = b % a.splits;
            group = b / a.splits;
        }
    }
    const int D = a.D, LDQ = D + 16, nks = D / 32;
    for (int i = tid; i < QT * (D / 4); i += kBlock) {
        const int q = i / (D / 4), w = i - q * (D / 4);
        int qi = group * QT + q;
        qi = qi < a.nq ? qi : a.nq - 1;
        const uint32_t v = reinterpret_cast<const uint32_t *>(a.q + (int64_t)qi * D)[w];
        *reinterpret_cast<uint32_t *>(q8s + q * LDQ + 4 * w) = v ^ 0x80808080u;
    }
    topk_init(tk);
    __syncthreads();
    for (int q = tid; q < QT; q += kBlock) {
        int s = 0;
        for (int w = 0; w < D / 4; ++w) {
            const int v = *reinterpret_cast<const int *>(q8s + q * LDQ + 4 * w);
            s = __builtin_amdgcn_sdot4(v, v, s, false);
        }
        qq[q] = s;
    }
    __syncthreads();

    const int64_t row_begin = (int64_t)split * a.rows_per_split;
    int64_t row_end = row_begin + a.rows_per_split;
    row_end = row_end < a.n ? row_end : a.n;
    const uint32_t n_local = (uint32_t)(row_end > row_begin ? row_end - row_begin : 0);
    const uint32_t last = n_local ? n_local - 1 : 0;
    const int lj = lane & 31, lh = lane >> 5;
    // (rows_per_split is rounded up to whole tiles, so the last splits of a plan can start past the end: their clamped loads read row 0)
    const int64_t row_base = n_local ? row_begin : 0;
    const uint8_t *xb = a.data + row_base * D;
    const int32_t *nb = a.norms + row_base;
    auto fetch = [&](uint32_t tile_base, mf_v4i (&v)[KS], int &xx) {
        uint32_t r = tile_base + wave * 32 + lj;
        r = r < last ? r : last;  // clamped, rejected at push time
        const uint8_t *p = xb + (size_t)r * D + lh * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (s < nks) v[s] = *reinterpret_cast<const mf_v4i *>(p + 32 * s);
        xx = nb[r];
    };
    mf_v4i cur[KS], nxt[KS];
    int xx_cur = 0, xx_nxt = 0;
    if (n_local) fetch(0, cur, xx_cur);
    int tile = 0;
    for (uint32_t base = 0; base < n_local; base += 128, ++tile) {
        fetch(base + 128, nxt, xx_nxt);
        mf_v16i acc[QB];
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s < nks) {
                mf_v4i bv = cur[s];
                bv ^= (int)0x80808080;
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    const mf_v4i av = *reinterpret_cast<const mf_v4i *>(q8s + (b * 32 + lj) * LDQ + 32 * s + lh * 16);
                    acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc[b], 0, 0, 0);
                }
            }
        }
        const uint32_t lrow = base + wave * 32 + lj;
        const bool valid = lrow < n_local;
        uint32_t key[QB][16];
        bool want = false;
        uint32_t pending = 0;
#pragma unroll
        for (int b = 0; b < QB; ++b) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                // per-query norm and threshold come from LDS (half-wave broadcast reads): holding them in
                // registers next to two row buffers and the accumulators spilled
                const int q = b * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                key[b][e] = (uint32_t)(qq[q] + xx_cur - 2 * acc[b][e]);  // exact, >= 0
                if (valid && key[b][e] < tk.thr_x[q]) {
                    if (!topk_push<QT, CAP, TRIG>(tk, q, key[b][e], (uint32_t)(row_begin + lrow), want)) pending |= 1u << (b * 16 + e);
                }
            }
        }
        const int f = tile % 3;
        if (want) tk.flag[f] = 1;
        __syncthreads();
        if (tk.flag[f]) {  // workgroup-uniform
            for (;;) {
                topk_compact<QT, CAP, kBlock>(tk, a.k);
                if (pending) tk.flag[3] = 1;
                __syncthreads();
                const int again = tk.flag[3];
                __syncthreads();
                if (!again) break;
                if (tid == 0) tk.flag[3] = 0;
                uint32_t still = 0;
                bool dummy = false;
#pragma unroll
                for (int b = 0; b < QB; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t bit = 1u << (b * 16 + e);
                        const int q = b * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        if ((pending & bit) && key[b][e] <= tk.thr_x[q])
                            if (!topk_push<QT, CAP, TRIG>(tk, q, key[b][e], (uint32_t)(row_begin + lrow), dummy)) still |= bit;
                    }
                pending = still;
                __syncthreads();
            }
        }
        if (tid == 0) tk.flag[(tile + 2) % 3] = 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
        xx_cur = xx_nxt;
    }
    __syncthreads();
    topk_compact<QT, CAP, kBlock>(tk, a.k);
    for (int q = 0; q < QT; ++q) {
        const int qi = group * QT + q;
        if (qi >= a.nq) break;
        const int cnt = tk.cnt[q];
        const int64_t o = ((int64_t)qi * a.splits + split) * a.k;
        for (int i = tid; i < a.k; i += kBlock) {
            if (i < cnt) {
                const unsigned long long e = tk.buf[q][i];
                a.part_d[o + i] = __uint_as_float((uint32_t)(e >> 32));  // int32 distance bits
                a.part_id[o + i] = (int64_t)(uint32_t)e;
            } else {
                a.part_d[o + i] = __uint_as_float(0x7f800000u);
                a.part_id[o + i] = -1;
            }
        }
    }
}

// ---- row-tile variant: the workgroup shares ROWS, every wave owns 32 QUERIES ----------------------------
// The kernel above lets each wave fetch its own 32 rows straight into the MFMA operand layout: a lane reads
// 16 bytes at a 512-byte stride, i.e. every load instruction touches 32 different cache lines for 1 KB of
// data, and the L1/TA path -- not HBM, not the matrix cores -- sets the pace (4.9 TB/s of row traffic,
// 6 % of the i8 MFMA peak).  Here a 32-row tile is fetched ONCE per workgroup with fully coalesced 16-byte
// loads, parked in LDS (double-buffered, one barrier per tile) and read by all NW waves in operand layout;
// each wave keeps its own 32 queries in registers for the whole launch (64 VGPRs at D = 512) and owns their
// selection buffers outright, so compaction is wave-local (register radix select, block_topk.h) and needs no
// workgroup protocol.  One pass over the rows now serves 32 * NW queries.
#ifdef CVTMI_FLAT_TIMING
__device__ unsigned long long g_flat_dbg[8];
#define FT_T(i) do { if (threadIdx.x == 0) { const unsigned long long now__ = clock64(); ft_acc__[i] += now__ - ft_last__; ft_last__ = now__; } } while (0)
#define FT_T0() unsigned long long ft_last__ = clock64(); unsigned long long ft_acc__[5] = { 0, 0, 0, 0, 0 }
#define FT_TEND() do { if (threadIdx.x == 0) for (int i__ = 0; i__ < 5; ++i__) atomicAdd(&g_flat_dbg[i__], ft_acc__[i__]); } while (0)
extern "C" int cvtmi_debug_flat_timing(unsigned long long *out, int reset)
{
    unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flat_dbg), sizeof z) != hipSuccess) return -3;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_flat_dbg), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
#else
#define FT_T(i) do { } while (0)
#define FT_T0() do { } while (0)
#define FT_TEND() do { } while (0)
#endif

struct NoFixBatch {
    template <int NR>
    __device__ __forceinline__ void operator()(int, unsigned long long (&)[NR], const bool (&)[NR]) const {}
};

// OPT (measurement switch, cvtmi_set_tuning("flat_u8_opt")): bit 0 = two accumulator chains per tile (even / odd K steps, summed at
// the end) so that back-to-back matrix instructions never wait for each other's result; bit 1 = all K-step operands of a tile are
// read from LDS before its first matrix instruction (64 VGPRs at D = 512) instead of one read in front of each.
template <int NW, int CAP, int DMAX, int OPT>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW >= 4 ? NW / 4 : 1, NW >= 4 ? NW / 4 : 1)))
void flat_u8_rowtile_kernel(const FlatMfmaArgs a)
{
    constexpr int NT = 64 * NW, QT = 32 * NW;
    constexpr int KS = DMAX / 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t rows_s[];  // [2][32][D + 16] row bytes ^ 0x80
    __shared__ TopKShared<QT, CAP> tk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int group, split;
    {
        const int b = blockIdx.x;
        if ((a.splits & 7) == 0) {  // a row split stays on one XCD: the query groups of that split share its L2
            const int s8 = a.splits >> 3;
            const int xcd = b & 7, i = b >> 3;
            split = xcd + 8 * (i % s8);
            group = i / s8;
        } else {
            split = b % a.splits;
            group = b / a.splits;
        }
    }
    const int D = a.D, LDR = D + 16, nks = D / 32;
    const int lj = lane & 31, lh = lane >> 5;
    topk_init(tk);

    // this wave's 32 queries in MFMA A layout: lane (lj, lh) holds dims 32 s + 16 lh .. + 16 of query lj
    mf_v4i qreg[KS];
    int qq_l = 0;  // |q'|^2 of query lj (both half-waves compute it)
    {
        int qi = group * QT + wave * 32 + lj;
        qi = qi < a.nq ? qi : a.nq - 1;
        const uint8_t *qp = a.q + (int64_t)qi * D;
        // unconditional (clamped) loads: predicated ones make hipcc wait for each before issuing the next
#pragma unroll
        for (int s = 0; s < KS; ++s) qreg[s] = *reinterpret_cast<const mf_v4i *>(qp + 32 * (s < nks ? s : 0) + 16 * lh);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qreg[s] ^= (int)0x80808080;
            if (s >= nks) qreg[s] = mf_v4i{ 0, 0, 0, 0 };
#pragma unroll
            for (int c = 0; c < 4; ++c) qq_l = __builtin_amdgcn_sdot4(qreg[s][c], qreg[s][c], qq_l, false);
        }
        qq_l += __shfl_xor(qq_l, 32);
    }
    // The 16 queries this lane scores: q_e = (e & 3) + 8 (e >> 2) + 4 lh  (MFMA C layout).
    // dist = |q'|^2 + |x'|^2 - 2<q',x'> < thr  <=>  |x'|^2 - 2<q',x'> < thr - |q'|^2: one register per query
    // (all terms are below 2^26, so the signed compare is exact; "no threshold yet" = INT_MAX).
    // Row splits of a query run concurrently and each would warm its own threshold up from scratch
    // (~k (1 + ln(rows/k)) candidates per split, and handling candidates is what this kernel's time goes to).
    // They share it instead: after a compaction a wave publishes its k-th best distance with atomicMin on
    // gthr[query]; every PD tiles it reads the value back and filters with min(own k-th, shared k-th + 1).
    // "+ 1": a row that ties the shared k-th may win the (distance, id) tie against rows of another split, so
    // it must survive; every row of the final top-k is <= every published k-th, hence never dropped.
    __shared__ int qq_s[QT];
    __shared__ int thq_s[QT];  // filter bound per query in the 'partial distance' domain (see below), INT_MAX = none yet
    // A tighter shared bound than "some split's k-th best": every split also publishes the SMALLEST distance it
    // has seen into slot (split % k) of its query (atomicMin).  The k slots then hold k distances of k different
    // rows (different splits), so their maximum bounds the global k-th best -- with S splits that is roughly the
    // k-th best of everything scanned so far, S/k times tighter than a single split's k-th.  Needs S >= k.
    __shared__ uint32_t min_s[QT];   // smallest key this workgroup has pushed per query
    __shared__ uint32_t bmax_s[QT];  // max over the k slots as last read (KEY_MAX until every slot is filled)
    if (lane < 32) { min_s[wave * 32 + lj] = KEY_MAX; bmax_s[wave * 32 + lj] = KEY_MAX; }
    if (lane < 32) qq_s[wave * 32 + lj] = qq_l;
    int qi_l = group * QT + wave * 32 + lj;
    qi_l = qi_l < a.nq ? qi_l : a.nq - 1;
    uint32_t g_l = __hip_atomic_load(&a.gthr[qi_l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // shared threshold of query lj as last read 
    // queries past nq are copies of the last one (clamped loads): left alone they select, compact and hammer the same gthr / gslot words as
    // the original (nq = 65 took 6.4 ms against 1.7 for 64, nq = 257 22.7 against 2.5 for 256).  Their bound is INT_MIN: nothing ever passes.
    const bool pad_l = group * QT + wave * 32 + lj >= a.nq;
    if (lane < 32) thq_s[wave * 32 + lj] = pad_l ? (int)0x80000000 : 0x7fffffff;
    __syncthreads();
    const int64_t row_begin = (int64_t)split * a.rows_per_split;
    int64_t row_end = row_begin + a.rows_per_split;
    row_end = row_end < a.n ? row_end : a.n;
    const uint32_t n_local = (uint32_t)(row_end > row_begin ? row_end - row_begin : 0);
    const uint32_t last = n_local ? n_local - 1 : 0;
    // (rows_per_split is rounded up to whole tiles, so the last splits of a plan can start past the end -- 1 M rows in 120 splits of 8448:
    //  their clamped loads read row 0 instead of memory past the index)
    const int64_t row_base = n_local ? row_begin : 0;
    const uint8_t *xb = a.data + row_base * D;
    const int32_t *nb = a.norms + row_base;
    // tile loader: the 32 x D bytes of a tile are 2 D 16-byte pieces, piece f = row f / (D/16), column f % (D/16)
    const int P16 = D >> 4, pieces = 32 * P16;
    constexpr int LPT = (32 * (DMAX / 16) + NT - 1) / NT;  // pieces per thread
    // (piece and row indices are clamped instead of predicated: every load is issued unconditionally, so the
    //  compiler can keep PD tiles in flight with counted waits)
    int pr[LPT], pc[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        int f = tid + i * NT;
        f = f < pieces ? f : pieces - 1;
        pr[i] = f / P16;
        pc[i] = f - pr[i] * P16;
    }
    auto fetch = [&](uint32_t base, mf_v4i (&v)[LPT], int &xx) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            uint32_t row = base + pr[i];
            row = row < last ? row : last;  // clamped, rejected at push time
            v[i] = *reinterpret_cast<const mf_v4i *>(xb + (size_t)row * D + 16 * pc[i]);
        }
        uint32_t row = base + lj;
        row = row < last ? row : last;
        xx = nb[row];
    };
    auto park = [&](int buf, const mf_v4i (&v)[LPT]) {
        uint8_t *dst = rows_s + buf * 32 * LDR;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            mf_v4i t = v[i];
            t ^= (int)0x80808080;
            if (tid + i * NT < pieces) *reinterpret_cast<mf_v4i *>(dst + pr[i] * LDR + 16 * pc[i]) = t;
        }
    };
    const NoFixBatch nofix;
    const IdThr idthr;
    // Row tiles are small (32 x D bytes), HBM latency is ~2 us: PD tiles are kept in flight in a register ring
    // (tile t sits in slot t % PD until it is parked in LDS one iteration before its turn).
    constexpr int PD = LPT <= 4 ? 4 : 2;
    const uint32_t n_tiles = (n_local + 31) / 32;
    mf_v4i pf[PD][LPT];
    int xxr[PD];
#pragma unroll
    for (int u = 0; u < PD; ++u) fetch(32u * u, pf[u], xxr[u]);  // rows past the split are clamped
    if (n_tiles) park(0, pf[0]);
    __syncthreads();
    FT_T0();
    // The tile loop runs to a multiple of PD and has no memory-related control flow (tiles past the end are
    // clamped duplicates whose rows are rejected at push time): with branches around the loads the compiler
    // falls back to s_waitcnt vmcnt(0) and the ring drains every iteration.
    // Software pipeline inside the wave: the 16 MFMAs of tile t are issued, then the test of tile t-1 (whose
    // accumulator is the other register set) runs on the VALU while the matrix pipe works -- the two used to
    // take turns (1.9 K + 2.0 K clocks per tile, tools/flat_timing.py).
    // epilogue(acc, xx, t): test the 16 (row, query) results of one tile, push the hits, compact on overflow.
    auto epilogue = [&](const mf_v16i &acc, int xx_t, uint32_t t, bool refresh, uint32_t hit) {
        const uint32_t lrow = 32u * t + lj;
        const bool valid = lrow < n_local;
        hit = valid ? hit : 0u;
        uint32_t pend = 0;
        if (__any(hit != 0)) {
            // Candidates go straight to the wave's own buffers; a buffer is compacted only when a push finds it
            // full (the lane keeps that candidate and offers it again afterwards with '<=': it may tie the new
            // k-th entry and carry the smaller id).  All of it is wave-local: no workgroup protocol.
            bool dummy = false;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (hit & (1u << e)) {
                    const int q = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const uint32_t key = (uint32_t)(qq_s[q] + xx_t - 2 * acc[e]);  // exact, >= 0
                    if (a.nslot) atomicMin(&min_s[q], key);
                    if (!topk_push<QT, CAP, CAP>(tk, q, key, (uint32_t)(row_begin + lrow), dummy)) pend |= 1u << e;
                }
            }
            while (__any(pend != 0)) {
                unsigned long long m = __ballot(lane < 32 && tk.cnt[wave * 32 + lj] >= CAP);
                while (m) {
                    const int ql = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int keep = topk_compact_wave_q_inl<QT, CAP, false>(tk, wave * 32 + ql, a.k, nofix, idthr);
                    if (lane == 0) tk.cnt[wave * 32 + ql] = keep;
                }
                if (lane < 32) {
                    const uint32_t th = tk.thr[wave * 32 + lj];
                    if (th != KEY_MAX && th < g_l) atomicMin(&a.gthr[qi_l], th);
                }
                uint32_t still = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (pend & (1u << e)) {
                        const int q = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        const uint32_t key = (uint32_t)(qq_s[q] + xx_t - 2 * acc[e]);
                        if (key <= tk.thr_x[q])
                            if (!topk_push<QT, CAP, CAP>(tk, q, key, (uint32_t)(row_begin + lrow), dummy)) still |= 1u << e;
                    }
                }
                pend = still;
                refresh = true;
            }
        }
        if (refresh) {  // wave-uniform
            if (lane < 32) {
                const uint32_t own = tk.thr_x[wave * 32 + lj];
                uint32_t sh = g_l == KEY_MAX ? KEY_MAX : g_l + 1u;
                const uint32_t bm = bmax_s[wave * 32 + lj];
                if (bm != KEY_MAX && bm + 1u < sh) sh = bm + 1u;
                const uint32_t th = own < sh ? own : sh;
                thq_s[wave * 32 + lj] = pad_l ? (int)0x80000000 : (th == KEY_MAX ? 0x7fffffff : (int)th - qq_s[wave * 32 + lj]);
            }
        }
    };
    // branch-free test of result e of a finished tile (bounds come from LDS: 2 distinct addresses per wave,
    // broadcast); issued between the MFMAs of the next tile so that it runs in their shadow
    auto hit_bit = [&](const mf_v16i &acc, int xx_t, int e) -> uint32_t {
        return ((xx_t - 2 * acc[e]) < thq_s[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh] ? 1u : 0u) << e;
    };
    mf_v16i accs[2];
    int xx_prev = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const uint32_t t = t0 + u;
            const int buf = u & 1;    // PD is even: tile t lives in LDS buffer t & 1 = u & 1
            const int xx_cur = xxr[u];
            fetch(32u * (t + PD), pf[u], xxr[u]);  // slot u is free: tile t is in LDS (clamped past the end)
            const uint8_t *rt = rows_s + buf * 32 * LDR + lj * LDR + 16 * lh;
            mf_v16i acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0;
            const mf_v16i &prev = accs[(u & 1) ^ 1];
            uint32_t hit = 0;
            constexpr int EPS = 16 / KS;  // results of the previous tile tested per MFMA of this one
            if constexpr (OPT == 0) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s < nks) {
                        const mf_v4i bv = *reinterpret_cast<const mf_v4i *>(rt + 32 * s);
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(qreg[s], bv, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < EPS; ++j) hit |= hit_bit(prev, xx_prev, s * EPS + j);
                    // keep the order written here: operand reads a few MFMAs ahead at most (64 VGPRs if all 16 were
                    // hoisted), the tests of the previous tile between the MFMAs
                    if ((s & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                mf_v16i acc2;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[e] = 0;
                mf_v4i bvs[KS];
                if constexpr ((OPT & 2) != 0) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) bvs[s] = *reinterpret_cast<const mf_v4i *>(rt + 32 * (s < nks ? s : 0));
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s < nks) {
                        const mf_v4i bv = (OPT & 2) ? bvs[s] : *reinterpret_cast<const mf_v4i *>(rt + 32 * s);
                        if ((OPT & 1) && (s & 1)) acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(qreg[s], bv, acc2, 0, 0, 0);
                        else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(qreg[s], bv, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < EPS; ++j) hit |= hit_bit(prev, xx_prev, s * EPS + j);
                }
                if constexpr ((OPT & 1) != 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
                }
            }
            accs[u & 1] = acc;
            FT_T(0);  // LDS reads + MFMAs issued, previous tile tested
            if (t > 0) epilogue(prev, xx_prev, t - 1, u == 0, hit);  // pushes (rare) + threshold refresh
            xx_prev = xx_cur;
            if (u == 0) g_l = __hip_atomic_load(&a.gthr[qi_l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // used at the next refresh, PD tiles on
            if (u == 0 && a.nslot && (t0 & 63) == 0) {  // every 64 tiles: publish this split's minima, fold last round's slot reads
                if (lane < 32) {
                    const uint32_t mine = min_s[wave * 32 + lj];
                    if (mine != KEY_MAX) atomicMin(&a.gslot[(int64_t)qi_l * a.nslot + split % a.nslot], mine);
                }
                // per-query maximum over the k slots (KEY_MAX while any slot is empty).  The reads are consumed at
                // once -- this drains the prefetch ring, once per 16 tiles -- rather than held in registers the
                // kernel does not have
                if (lane < 32) bmax_s[wave * 32 + lj] = 0u;
                for (int idx = lane; idx < 32 * a.nslot; idx += 64) {
                    const int ql = idx / a.nslot, j = idx - ql * a.nslot;
                    int qq2 = group * QT + wave * 32 + ql;
                    qq2 = qq2 < a.nq ? qq2 : a.nq - 1;
                    const uint32_t v = __hip_atomic_load(&a.gslot[(int64_t)qq2 * a.nslot + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicMax(&bmax_s[wave * 32 + ql], v);
                }
            }
            FT_T(1);  // test + pushes + compactions
            park(buf ^ 1, pf[(u + 1) % PD]);
            FT_T(2);  // wait for the prefetched tile + LDS store
            lds_barrier();  // next tile parked by everyone, this tile read by everyone (global prefetch stays in flight)
            FT_T(3);  // barrier
        }
    }
    if (n_tiles) {  // the last tile processed by the loop (the loop runs to a multiple of PD: its parity is known)
        const uint32_t t_last = ((n_tiles + PD - 1) / PD) * PD - 1;
        uint32_t hit = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) hit |= hit_bit(accs[(PD - 1) & 1], xx_prev, e);
        epilogue(accs[(PD - 1) & 1], xx_prev, t_last, false, hit);
    }
    FT_TEND();
    // sorted result of this wave's queries
    for (int ql = 0; ql < 32; ++ql) {
        const int q = wave * 32 + ql, qi = group * QT + q;
        if (qi >= a.nq) break;  // wave-uniform
        const int cnt = topk_compact_wave_q<QT, CAP, true>(tk, q, a.k, nofix, idthr);
        if (lane == 0 && tk.thr[q] != KEY_MAX) atomicMin(&a.gthr[qi], tk.thr[q]);
        const int64_t o = ((int64_t)qi * a.splits + split) * a.k;
        for (int i = lane; i < a.k; i += 64) {
            if (i < cnt) {
                const unsigned long long e = tk.buf[q][i];
                a.part_d[o + i] = __uint_as_float((uint32_t)(e >> 32));  // int32 distance bits
                a.part_id[o + i] = (int64_t)(uint32_t)e;
            } else {
                a.part_d[o + i] = __uint_as_float(0x7f800000u);
                a.part_id[o + i] = -1;
            }
        }
    }
}

static int g_flat_u8_opt = 0;
void set_flat_u8_opt(int v) { g_flat_u8_opt = v; }

// Queries per workgroup of the MFMA paths, 0 = shape not covered (caller falls back to the dot4 kernel).
// More queries per workgroup = fewer passes over the rows; what bounds it is LDS: two row tiles
// (32 x (D + 16) bytes each) next to the selection buffers (8 * CAP bytes per query, CAP >= k + 32).
int flat_u8_mfma_qtile(int D, int k, int64_t nq)
{
    if (D % 32 != 0 || D > 512 || nq < 5 || k > 128) return 0;   // 1..4 queries: streaming / row-per-lane kernels (5 on those took 3.4 ms against 1.7 here)
    // measured on 2 M x 512-d (tools/bench_flat_u8.py): the row-tile kernel wins wherever its selection buffers
    // leave room for >= 4 waves of queries; large k falls back to one query block per workgroup
    if (k <= 24) return nq > 128 ? 256 : (nq > 64 ? 128 : (nq > 32 ? 64 : 32));
    if (k <= 80) return nq > 64 ? 128 : 32;
    return 32;
}

int flat_u8_mfma_splits(int64_t n, int64_t nq, int qt)
{
    if (qt <= 32) return flat_plan_splits(n, nq, qt);
    const int64_t groups = (nq + qt - 1) / qt;
    int64_t s = (512 + groups - 1) / groups;
    const int64_t max_by_rows = n / 8192 > 1 ? n / 8192 : 1;
    if (s > max_by_rows) s = max_by_rows;
    if (s >= 8) s = (s / 8) * 8;  // a row split per XCD: the query groups of a split share its L2
    return (int)(s < 1 ? 1 : s);
}

int launch_flat_u8_mfma(int D, const uint8_t *data, const int32_t *norms, int64_t n, const uint8_t *q, int64_t nq, int k,
                        int splits, float *part_d, int64_t *part_id, uint32_t *gthr, hipStream_t st)
{
    const int qt = flat_u8_mfma_qtile(D, k, nq);
    if (!qt) return fail(CVTMI_EUNSUPPORTED, "flat_u8_mfma: shape not covered");
    FlatMfmaArgs a;
    a.data = data; a.norms = norms; a.n = n; a.q = q; a.nq = (int)nq; a.D = D; a.k = k; a.splits = splits;
    int64_t rps = (n + splits - 1) / splits;
    rps = ((rps + 127) / 128) * 128;
    if (rps < 128) rps = 128;
    a.rows_per_split = rps;
    a.part_d = part_d; a.part_id = part_id; a.gthr = gthr;
    a.nslot = (qt > 32 && k <= 16 && splits >= 2 * k) ? k : 0;
    a.gslot = gthr ? gthr + nq : nullptr;
    const int64_t groups = (nq + qt - 1) / qt;
    const int64_t blocks = groups * splits;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "flat_u8_mfma: grid too large");
    if (qt > 32) {
        if (!gthr) return fail(CVTMI_EINVAL, "flat_u8_mfma: threshold scratch missing");
        CVTMI_HIP(hipMemsetAsync(gthr, 0xff, (size_t)nq * (1 + 16) * sizeof(uint32_t), st));  // k-th bests + up to 16 slots per query
    }
    if (qt == 32) {
        const size_t lds = (size_t)qt * (D + 16);
#define CVTMI_FM(QB, CAP, TRIG, DMAX)                                                                                   \
    do {                                                                                                                \
        CVTMI_HIP(hipFuncSetAttribute((const void *)flat_u8_mfma_kernel<QB, CAP, TRIG, DMAX>,                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        hipLaunchKernelGGL((flat_u8_mfma_kernel<QB, CAP, TRIG, DMAX>), dim3((unsigned)blocks), dim3(kBlock), lds, st, a); \
    } while (0)
        if (D <= 128) CVTMI_FM(1, 208, 176, 128); else CVTMI_FM(1, 208, 176, 512);
#undef CVTMI_FM
    } else {
        const size_t lds = (size_t)2 * 32 * (D + 16);
#define CVTMI_RT1(NW, CAP, DMAX, OPT)                                                                                   \
    do {                                                                                                                \
        CVTMI_HIP(hipFuncSetAttribute((const void *)flat_u8_rowtile_kernel<NW, CAP, DMAX, OPT>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        hipLaunchKernelGGL((flat_u8_rowtile_kernel<NW, CAP, DMAX, OPT>), dim3((unsigned)blocks), dim3(64 * NW), lds, st, a); \
    } while (0)
#define CVTMI_RT(NW, CAP, DMAX)                                                                                         \
    do {                                                                                                                \
        if (NW == 8 && DMAX == 512 && g_flat_u8_opt == 0) CVTMI_RT1(NW, CAP, DMAX, 0);                                  \
        else if (NW == 8 && DMAX == 512 && g_flat_u8_opt == 1) CVTMI_RT1(NW, CAP, DMAX, 1);                             \
        else if (NW == 8 && DMAX == 512 && g_flat_u8_opt == 2) CVTMI_RT1(NW, CAP, DMAX, 2);                             \
        else if (NW == 8 && DMAX == 512) CVTMI_RT1(NW, CAP, DMAX, 3);                                                   \
        else CVTMI_RT1(NW, CAP, DMAX, 0);                                                                               \
    } while (0)
        // CAP by k: k <= 24 -> 56, k <= 80 -> 112 (CAP - k >= 32: a compacted buffer takes a whole tile)
        if (k <= 24) {
            if (qt == 256) { if (D <= 128) CVTMI_RT(8, 56, 128); else CVTMI_RT(8, 56, 512); }
            else if (qt == 128) { if (D <= 128) CVTMI_RT(4, 56, 128); else CVTMI_RT(4, 56, 512); }
            else { if (D <= 128) CVTMI_RT(2, 56, 128); else CVTMI_RT(2, 56, 512); }
        } else {  // k <= 80, 128 queries
            if (D <= 128) CVTMI_RT(4, 112, 128); else CVTMI_RT(4, 112, 512);
        }
#undef CVTMI_RT
#undef CVTMI_RT1
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

__global__ __launch_bounds__(kBlock) void gather_labels_kernel(int64_t *ids, int64_t count, const int64_t *labels)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < count) {
        const int64_t r = ids[i];
        ids[i] = r >= 0 ? labels[r] : -1;
    }
}

int launch_gather_labels(int64_t *ids, int64_t count, const int64_t *labels, hipStream_t st)
{
    if (count <= 0) return CVTMI_OK;
    hipLaunchKernelGGL(gather_labels_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, ids,
                       count, labels);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

__global__ __launch_bounds__(kBlock) void offset_labels_kernel(int64_t *ids, int64_t count, int64_t base)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < count && ids[i] >= 0) ids[i] += base;
}
int launch_offset_labels(int64_t *ids, int64_t count, int64_t base, hipStream_t st)
{
    if (count <= 0) return CVTMI_OK;
    hipLaunchKernelGGL(offset_labels_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, ids, count, base);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

constexpr int STREAM_SLICES = 64;
// selection after flat_u8_mstream_kernel (flat_mfma.hip): wave minima wmin[nq][G], tile minima tmin[tiles][nqp] -> part [nq][slices][k] -> merge
int launch_flat_u8_mstream_finish(int D, const uint8_t *data, int64_t n, const uint8_t *q, int64_t nq, int k, const int32_t *wmin, int G,
                                  const int32_t *tmin, int nqp, int tile_group, float *part_d, int64_t *part_id, float *out_d, int64_t *out_rows, hipStream_t st)
{
    // slices per query: every slice repeats the query's threshold selection, so no more of them than fill the chip once
    // (64 queries x 64 slices = 4096 workgroups took 112 us; x 16 ...); long indexes keep 64 (rounds per slice <= FIN_MAXR)
    int S = STREAM_SLICES;
    if (n <= ((int64_t)64 << 20))
        while (S > 4 && nq * S > 1024) S >>= 1;
    StreamFinishArgs fa = {};
    fa.n = n; fa.gmin = wmin; fa.G = G; fa.S = S; fa.rpi_log2 = 5; fa.k = k; fa.part_d = part_d; fa.part_id = part_id;
    fa.tmin = tmin; fa.nqp = nqp; fa.tile_group = tile_group; fa.X = data; fa.Q = q; fa.D = D;
    hipLaunchKernelGGL(flat_u8_stream_finish_kernel, dim3((unsigned)(nq * S)), dim3(kBlock), 0, st, fa);
    CVTMI_HIP(hipGetLastError());
    return launch_topk_merge(part_d, part_id, nq, S, k, out_d, out_rows, st);
}
int flat_u8_stream_slices() { return STREAM_SLICES; }

int flat_qtile(int64_t nq) { return nq >= 4 ? 4 : 1; }

int flat_plan_splits(int64_t n, int64_t nq, int qtile)
{
    // 8 workgroups of 256 threads fit a CU: 2048 run at a time on 256 CUs.  One round of long workgroups beats
    // two rounds of short ones (tools/bench_flat_f32.py: 2250 workgroups took as long as 2 x their own length),
    // so the row splits fill the machine once and no more: the largest split count with groups * splits <= 2048.
    const int64_t groups = (nq + qtile - 1) / qtile;
    const int64_t slots = 2048;
    int64_t need = slots / groups;
    int64_t max_by_rows = n / 2048;
    if (max_by_rows < 1) max_by_rows = 1;
    if (need > max_by_rows) need = max_by_rows;
    if (need < 1) need = 1;
    if (need > 1024) need = 1024;
    return (int)need;
}

template <bool IP, int LANES>
static int launch_f32(const FlatArgs &a, int qtile, unsigned blocks, size_t lds, hipStream_t st)
{
    if (a.k > 128) {  // one query per workgroup, the large selection buffer (kernels.h: kBigK)
        if constexpr (LANES >= 4) hipLaunchKernelGGL((flat_f32_sq_kernel<IP, LANES, 1, kBigCap, kBigTrig>), dim3(blocks), dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((flat_f32_kernel<IP, LANES, 1, kBigCap, kBigTrig>), dim3(blocks), dim3(kBlock), lds, st, a);
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    if constexpr (LANES >= 4) {  // queries in SGPRs (rows are 16-byte aligned: D % 4 == 0)
        if (qtile == 4) hipLaunchKernelGGL((flat_f32_sq_kernel<IP, LANES, 4>), dim3(blocks), dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((flat_f32_sq_kernel<IP, LANES, 1>), dim3(blocks), dim3(kBlock), 0, st, a);
    } else {
        if (qtile == 4) hipLaunchKernelGGL((flat_f32_kernel<IP, LANES, 4>), dim3(blocks), dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((flat_f32_kernel<IP, LANES, 1>), dim3(blocks), dim3(kBlock), lds, st, a);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_flat_search(int metric, int D, const void *data, int64_t n, const void *q, int64_t nq, int k, int qtile,
                       int splits, float *part_d, int64_t *part_id, hipStream_t st, const uint32_t *only_if)
{
    if (nq <= 0) return CVTMI_OK;
    if (k < 1 || k > kBigK) return fail(CVTMI_EUNSUPPORTED, "flat_search: k=%d outside 1..%d", k, kBigK);
    if (n > 0xfffffffeLL || nq > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "flat_search: index too large");
    if (qtile != 1 && qtile != 4) return fail(CVTMI_EINVAL, "flat_search: qtile %d", qtile);
    if (k > 128 && qtile != 1) return fail(CVTMI_EINVAL, "flat_search: k=%d needs qtile 1", k);
    if (D < 1 || D > 4096) return fail(CVTMI_EUNSUPPORTED, "flat_search: D=%d outside 1..4096", D);
    FlatArgs a;
    a.data = data; a.n = n; a.q = q; a.nq = (int)nq; a.D = D; a.k = k; a.splits = splits;
    int64_t rps = (n + splits - 1) / splits;
    const int64_t tile_rows = (int64_t)kBlock * FLAT_R;
    rps = ((rps + tile_rows - 1) / tile_rows) * tile_rows;
    if (rps < tile_rows) rps = tile_rows;
    a.rows_per_split = rps;
    a.part_d = part_d; a.part_id = part_id;
    a.only_if = only_if;
    const int64_t groups = (nq + qtile - 1) / qtile;
    const int64_t blocks = groups * splits;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "flat_search: grid too large");
    const size_t lds_f = (size_t)qtile * D * sizeof(float);
    switch (metric) {
        case CVTMI_METRIC_IP:
            if (D % 4 == 0) return launch_f32<true, 4>(a, qtile, (unsigned)blocks, lds_f, st);
            return launch_f32<true, 1>(a, qtile, (unsigned)blocks, lds_f, st);
        case CVTMI_METRIC_L2F:
            if (D % 16 == 0) return launch_f32<false, 8>(a, qtile, (unsigned)blocks, lds_f, st);
            if (D % 4 == 0) return launch_f32<false, 4>(a, qtile, (unsigned)blocks, lds_f, st);
            return launch_f32<false, 1>(a, qtile, (unsigned)blocks, lds_f, st);
        case CVTMI_METRIC_L2U8: {
            const size_t lds_u = (size_t)qtile * ((D >> 2) + 1) * sizeof(uint32_t);
            if (k > 128) {
                if (D % 16 == 0) hipLaunchKernelGGL((flat_u8_kernel<true, 1, kBigCap, kBigTrig>), dim3((unsigned)blocks), dim3(kBlock), lds_u, st, a);
                else hipLaunchKernelGGL((flat_u8_kernel<false, 1, kBigCap, kBigTrig>), dim3((unsigned)blocks), dim3(kBlock), lds_u, st, a);
            } else if (D % 16 == 0) {
                if (qtile == 4) hipLaunchKernelGGL((flat_u8_kernel<true, 4>), dim3((unsigned)blocks), dim3(kBlock), lds_u, st, a);
                else hipLaunchKernelGGL((flat_u8_kernel<true, 1>), dim3((unsigned)blocks), dim3(kBlock), lds_u, st, a);
            } else {
                if (qtile == 4) hipLaunchKernelGGL((flat_u8_kernel<false, 4>), dim3((unsigned)blocks), dim3(kBlock), lds_u, st, a);
                else hipLaunchKernelGGL((flat_u8_kernel<false, 1>), dim3((unsigned)blocks), dim3(kBlock), lds_u, st, a);
            }
            CVTMI_HIP(hipGetLastError());
            return CVTMI_OK;
        }
        default: break;
    }
    return fail(CVTMI_EINVAL, "flat_search: unknown metric %d", metric);
}

}  // namespace cvtmi
