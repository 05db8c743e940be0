// opq_encode.hip -- index-build side of the OPQ path: coarse assignment, PQ encode, and the
// stand-alone distance-table kernel.
//
// Reference arithmetic (opq/src/IVFOPQ.cpp):
//   coarse argmin  :110-129  dismin = float(UINT_MAX); sum_j (x_j - c_ij)^2 sequential fp32;
//                            strict '<' => first minimum; vw = -1 when nothing beats the start
//   residual       :135-139  r = x - coarse[vw]
//   PQ argmin      :141-161  per sub-quantiser m: argmin_j sum_k (r[m*step+k] - book[m][j][k])^2,
//                            same start value and tie rule; code = (uchar)vw1  (-1 -> 255)
//   LUT            :273-291
// Bit-exactness rules out the |x|^2 - 2xc + |c|^2 GEMM form: every distance is evaluated in the
// reference's operation order with __fsub_rn/__fmul_rn/__fadd_rn (no contraction), on the VALU.
//
// MI355X mapping of the encode: a workgroup owns 256*V rows; for each sub-quantiser the 256 x step
// codebook (8 KB at step 8) is staged in LDS once and every lane walks all K centroids for its V
// rows (LDS broadcast reads, row sub-vectors in registers).  Codes are packed in registers and
// leave as one 16-byte (M=16) store per row.
#include "kernels.h"

namespace cvtmi {

constexpr float kStartDist = 4294967296.0f;  // (float)UINT_MAX

// ------------------------------------------------------------------------------------------
// coarse assignment: each lane keeps CT running sums (one per centroid of the LDS tile) and
// streams over the dimensions, so every sum is accumulated in ascending j as the reference does.
// ------------------------------------------------------------------------------------------
constexpr int COARSE_CT = 32;   // centroids per LDS tile
constexpr int COARSE_DC = 128;  // dimensions per LDS chunk

__global__ __launch_bounds__(kBlock) void coarse_assign_kernel(const float *__restrict__ x, int64_t n, int D,
                                                               const float *__restrict__ coarse, int coarseK,
                                                               int32_t *__restrict__ out)
{
    __shared__ float cen[COARSE_DC][COARSE_CT];  // transposed: [dim][centroid]
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = row < n;
    const float *xr = x + (valid ? row : 0) * D;
    float best = kStartDist;
    int bi = -1;
    for (int c0 = 0; c0 < coarseK; c0 += COARSE_CT) {
        float acc[COARSE_CT];
#pragma unroll
        for (int c = 0; c < COARSE_CT; ++c) acc[c] = 0.0f;
        for (int d0 = 0; d0 < D; d0 += COARSE_DC) {
            __syncthreads();
            for (int i = threadIdx.x; i < COARSE_DC * COARSE_CT; i += kBlock) {
                const int c = i / COARSE_DC, d = i - c * COARSE_DC;  // coalesced along d
                float v = 0.0f;
                if (c0 + c < coarseK && d0 + d < D) v = coarse[(int64_t)(c0 + c) * D + d0 + d];
                cen[d][c] = v;
            }
            __syncthreads();
            const int dl = (D - d0) < COARSE_DC ? (D - d0) : COARSE_DC;
            for (int d = 0; d < dl; ++d) {
                const float xv = xr[d0 + d];
#pragma unroll
                for (int c = 0; c < COARSE_CT; ++c) {
                    const float t = __fsub_rn(xv, cen[d][c]);
                    acc[c] = __fadd_rn(acc[c], __fmul_rn(t, t));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < COARSE_CT; ++c) {
            if (c0 + c < coarseK && acc[c] < best) {
                best = acc[c];
                bi = c0 + c;
            }
        }
    }
    if (valid) out[row] = bi;
}

int launch_coarse_assign(const OpqModelDev &m, const float *x_rot, int64_t n, int32_t *list_id, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    // the k-means assignment kernel is this computation (same arithmetic, same first-minimum rule); for D <= 128
    // it keeps the row in registers and packs two centroids per instruction (kmeans.hip)
    if (m.D <= 128) return launch_kmeans_assign(x_rot, m.D, n, m.D, m.coarse, m.coarseK, list_id, nullptr, st);
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "coarse_assign: n too large");
    hipLaunchKernelGGL(coarse_assign_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, x_rot, n, m.D, m.coarse,
                       m.coarseK, list_id);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// ------------------------------------------------------------------------------------------
// PQ encode
// ------------------------------------------------------------------------------------------
constexpr int ENC_V = 4;  // rows per lane

template <int STEP>
__global__ __launch_bounds__(kBlock) void pq_encode_kernel(const float *__restrict__ x, int64_t n, int D, int M, int K,
                                                           const float *__restrict__ coarse,
                                                           const int32_t *__restrict__ list_id,
                                                           const float *__restrict__ books,
                                                           uint8_t *__restrict__ codes)
{
    __shared__ __attribute__((aligned(16))) float cb[256 * STEP];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * (kBlock * ENC_V) + tid;
    const float *cen[ENC_V];
    bool valid[ENC_V];
#pragma unroll
    for (int v = 0; v < ENC_V; ++v) {
        const int64_t row = row0 + (int64_t)v * kBlock;
        valid[v] = row < n;
        int l = 0;
        if (valid[v] && list_id) l = list_id[row];
        if (l < 0) l = 0;  // all-NaN row: the reference reads coarse[-1]; list 0 is used here (see oracle)
        cen[v] = coarse + (int64_t)l * D;
    }
    uint32_t packed[ENC_V][4];
#pragma unroll
    for (int v = 0; v < ENC_V; ++v)
#pragma unroll
        for (int w = 0; w < 4; ++w) packed[v][w] = 0;

#pragma unroll 1
    for (int m = 0; m < M; ++m) {
        __syncthreads();
        for (int i = tid; i < K * STEP; i += kBlock) cb[i] = books[(int64_t)m * K * STEP + i];
        __syncthreads();
        // Two rows ride in the two halves of every packed-fp32 register: x2[p][kk] = (row 2p, row 2p+1) at
        // dimension kk, the centroid value is broadcast to both halves.  Every v_pk_add/v_pk_mul then does
        // useful work in both halves and each row still sees sub, mul, add in the reference's order
        // (measured: scalar code or hipcc's own SLP packing of one row are 10-45 % slower).
        float2 x2[ENC_V / 2][STEP];
#pragma unroll
        for (int p = 0; p < ENC_V / 2; ++p) {
            const int64_t ra = valid[2 * p] ? row0 + (int64_t)(2 * p) * kBlock : 0;
            const int64_t rb = valid[2 * p + 1] ? row0 + (int64_t)(2 * p + 1) * kBlock : 0;
#pragma unroll
            for (int kk = 0; kk < STEP; ++kk)
                x2[p][kk] = make_float2(__fsub_rn(x[ra * D + m * STEP + kk], cen[2 * p][m * STEP + kk]),
                                        __fsub_rn(x[rb * D + m * STEP + kk], cen[2 * p + 1][m * STEP + kk]));
        }
        float best[ENC_V];
        int bj[ENC_V];
#pragma unroll
        for (int v = 0; v < ENC_V; ++v) { best[v] = kStartDist; bj[v] = -1; }
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            float c[STEP];
#pragma unroll
            for (int kk = 0; kk < STEP; ++kk) c[kk] = cb[j * STEP + kk];
#pragma unroll
            for (int p = 0; p < ENC_V / 2; ++p) {
                float2 d = make_float2(0.0f, 0.0f);
#pragma unroll
                for (int kk = 0; kk < STEP; ++kk) {
                    const float2 t = x2[p][kk] - make_float2(c[kk], c[kk]);
                    d = d + t * t;
                }
                if (d.x < best[2 * p]) { best[2 * p] = d.x; bj[2 * p] = j; }
                if (d.y < best[2 * p + 1]) { best[2 * p + 1] = d.y; bj[2 * p + 1] = j; }
            }
        }
        if (M <= 16) {
#pragma unroll
            for (int v = 0; v < ENC_V; ++v) {
                const uint32_t b = (uint32_t)(bj[v] & 0xff) << (8 * (m & 3));
                // m>>2 is runtime: select without dynamic register indexing
                packed[v][0] |= (m >> 2) == 0 ? b : 0u;
                packed[v][1] |= (m >> 2) == 1 ? b : 0u;
                packed[v][2] |= (m >> 2) == 2 ? b : 0u;
                packed[v][3] |= (m >> 2) == 3 ? b : 0u;
            }
        } else {
#pragma unroll
            for (int v = 0; v < ENC_V; ++v)
                if (valid[v]) codes[(row0 + (int64_t)v * kBlock) * M + m] = (uint8_t)bj[v];
        }
    }
    if (M <= 16) {
#pragma unroll
        for (int v = 0; v < ENC_V; ++v) {
            if (!valid[v]) continue;
            uint8_t *dst = codes + (row0 + (int64_t)v * kBlock) * M;
            if (M == 16) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4(packed[v][0], packed[v][1], packed[v][2], packed[v][3]);
            } else if (M == 8) {
                *reinterpret_cast<uint2 *>(dst) = make_uint2(packed[v][0], packed[v][1]);
            } else if (M == 4) {
                *reinterpret_cast<uint32_t *>(dst) = packed[v][0];
            } else {
                for (int mm = 0; mm < M; ++mm) dst[mm] = (uint8_t)(packed[v][mm >> 2] >> (8 * (mm & 3)));
            }
        }
    }
}

// generic sub-vector length (any step): one row per lane, codebook read through the caches
__global__ __launch_bounds__(kBlock) void pq_encode_generic_kernel(const float *__restrict__ x, int64_t n, int D, int M,
                                                                   int K, int step, const float *__restrict__ coarse,
                                                                   const int32_t *__restrict__ list_id,
                                                                   const float *__restrict__ books,
                                                                   uint8_t *__restrict__ codes)
{
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= n) return;
    int l = list_id ? list_id[row] : 0;
    if (l < 0) l = 0;
    const float *cen = coarse + (int64_t)l * D;
    const float *xr = x + row * D;
    for (int m = 0; m < M; ++m) {
        float best = kStartDist;
        int bj = -1;
        for (int j = 0; j < K; ++j) {
            const float *c = books + ((int64_t)m * K + j) * step;
            float d = 0.0f;
            for (int kk = 0; kk < step; ++kk) {
                const float r = __fsub_rn(xr[m * step + kk], cen[m * step + kk]);
                const float t = __fsub_rn(r, c[kk]);
                d = __fadd_rn(d, __fmul_rn(t, t));
            }
            if (d < best) { best = d; bj = j; }
        }
        codes[row * M + m] = (uint8_t)bj;
    }
}

int launch_pq_encode(const OpqModelDev &m, const float *x_rot, int64_t n, const int32_t *list_id, uint8_t *codes,
                     hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    if (m.K > 256) return fail(CVTMI_EUNSUPPORTED, "pq_encode: K=%d > 256", m.K);
    const int64_t rows_per_block = (int64_t)kBlock * ENC_V;
    const int64_t blocks = (n + rows_per_block - 1) / rows_per_block;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "pq_encode: n too large");
#define CVTMI_ENC(S)                                                                                               \
    case S:                                                                                                        \
        hipLaunchKernelGGL((pq_encode_kernel<S>), dim3((unsigned)blocks), dim3(kBlock), 0, st, x_rot, n, m.D, m.M, \
                           m.K, m.coarse, list_id, m.books, codes);                                                \
        break;
    switch (m.step) {
        CVTMI_ENC(2) CVTMI_ENC(4) CVTMI_ENC(8) CVTMI_ENC(16) CVTMI_ENC(32)
        default: {
            const int64_t gb = (n + kBlock - 1) / kBlock;
            if (gb > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "pq_encode: n too large");
            hipLaunchKernelGGL(pq_encode_generic_kernel, dim3((unsigned)gb), dim3(kBlock), 0, st, x_rot, n, m.D, m.M,
                               m.K, m.step, m.coarse, list_id, m.books, codes);
        }
    }
#undef CVTMI_ENC
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// ------------------------------------------------------------------------------------------
// stand-alone LUT: lut[nq][M][K]  (the scan kernel builds its own copy straight into LDS)
// ------------------------------------------------------------------------------------------
// ld = entries per (query, m) row in the output: K (the API layout) or 256 with +inf padding (scan scratch)
__global__ __launch_bounds__(kBlock) void lut_kernel(const float *__restrict__ q_rot, int D, int M, int K, int step,
                                                     const float *__restrict__ coarse,
                                                     const int32_t *__restrict__ list_id,
                                                     const float *__restrict__ books, float *__restrict__ lut, int ld)
{
    extern __shared__ __attribute__((aligned(16))) float res[];  // D floats
    const int64_t qi = blockIdx.x;
    int l = list_id ? list_id[qi] : 0;
    if (l < 0) l = 0;
    for (int d = threadIdx.x; d < D; d += kBlock) res[d] = __fsub_rn(q_rot[qi * D + d], coarse[(int64_t)l * D + d]);
    __syncthreads();
    for (int e = threadIdx.x; e < M * ld; e += kBlock) {
        const int m = e / ld, j = e - m * ld;
        float acc = __uint_as_float(0x7f800000u);
        if (j < K) {
            const float *c = books + ((int64_t)m * K + j) * step;
            acc = 0.0f;
            for (int kk = 0; kk < step; ++kk) {
                const float t = __fsub_rn(res[m * step + kk], c[kk]);
                acc = __fadd_rn(acc, __fmul_rn(t, t));
            }
        }
        lut[qi * M * ld + e] = acc;
    }
}

int launch_lut(const OpqModelDev &m, const float *q_rot, int64_t nq, const int32_t *list_id, float *lut, hipStream_t st,
               int ld)
{
    if (ld <= 0) ld = m.K;
    if (nq <= 0) return CVTMI_OK;
    if (nq > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "lut: nq too large");
    hipLaunchKernelGGL(lut_kernel, dim3((unsigned)nq), dim3(kBlock), (size_t)m.D * sizeof(float), st, q_rot, m.D, m.M,
                       m.K, m.step, m.coarse, list_id, m.books, lut, ld);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
