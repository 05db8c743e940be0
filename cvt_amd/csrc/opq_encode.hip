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
// Bit-exactness rules out taking distances from the |x|^2 - 2xc + |c|^2 GEMM form.  Two encode kernels:
//   pq_encode_kernel       every distance in the reference's operation order (__fsub_rn/__fmul_rn/__fadd_rn, no
//                          contraction) on the VALU: a workgroup owns 256*V rows; per sub-quantiser the 256 x step
//                          codebook slice (8 KB at step 8) is staged in LDS and every lane walks all K centroids for
//                          its V rows (LDS broadcast reads, row sub-vectors in registers)
//   pq_encode_mfma_kernel  the GEMM form on the bf16 matrix cores as a FILTER with a proven error bound; pairs it
//                          cannot separate go through the reference's chain (K = 256, step 8 / 16, D <= 128)
// Codes are packed in registers and leave as one 16-byte (M=16) store per row.
#include <algorithm>

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

// ------------------------------------------------------------------------------------------
// PQ encode, matrix-core filter + exact resolution (K = 256, D <= 128, step 8 or 16)
//
// The code byte is an argmin: only WHICH centroid wins has to equal the reference, not the distances.  For
// row sub-vector r and centroid c_j, |r - c_j|^2 = |r|^2 - 2 T_j with T_j = r.c_j - |c_j|^2 / 2, so the winner
// is the largest T_j -- a (256 x step) x (step x 32) product per (32 rows, sub-quantiser).  The product is only
// a FILTER: T_j carries rounding error and the reference's own sum carries its own.  Both are bounded (below),
// so when the best T beats the runner-up by more than the bound the reference's argmin is that centroid, bit
// for bit; otherwise (near ties, duplicate codewords, non-finite input: ~1 % of the (row, m) pairs on
// SIFT-shaped data) the wave evaluates all 256 centroids of that pair in the reference's operation order with
// the reference's first-minimum rule.  Output = the reference's codes for every input.
//
// Which matrix instruction.  v_mfma_f32_32x32x2_f32 is exact enough by itself, but the fp32 matrix rate equals
// the fp32 vector rate and (measured: SQ_VALU_MFMA_BUSY + SQ_ACTIVE_INST_VALU add up to the kernel's cycles) it
// does not run beside VALU work -- and the reduction below is VALU work.  The bf16 matrix pipe does run beside
// the VALU and is 16x faster, so the fp32 operands are split into two bf16 terms each (c = c1 + c2 + O(2^-18 |c|),
// r likewise): (c1 + c2).(r1 + r2) costs two v_mfma_f32_32x32x16_bf16 per 32 x 32 tile at step 8 (the 16-deep K
// holds c_x.r1 | c_x.r2 side by side), four at step 16; products of bf16 pairs are exact in the fp32 accumulator.
// One more product adds -|c|^2/2 (two bf16 terms against ones).  The codebook is split once per workgroup, into LDS.
//
// Layout: one persistent workgroup per CU keeps both bf16 halves of the whole codebook (D x 1 KB) and -|c|^2/2 in
// LDS; a wave owns 32 rows at a time.  Centroids are the A side of the product, rows the B side, so a lane ends
// up holding 16 centroids x 8 tiles of ITS row (lane & 31) and the running best / second best stay in registers:
// key = T with its low 6 bits replaced by the position (v_and_or_b32), second = v_med3_f32(best, second, key),
// best = max(best, key) -- 3 VALU ops per (row, centroid) instead of the 24 of the exact chain, issued between
// the products of the NEXT tile group.  The two lane halves (centroid rows 4*(lane>>5) of the output) merge with
// one cross-half exchange per (32 rows, m).
//
// Bound.  u = 2^-24, Q = |r|^2 + max_j |c_j|^2, so |T_j| <= Q and d_j = |r - c_j|^2 <= 2Q.  Per key:
//   bf16 split of r and of c (round to nearest, twice): 2^-18 |r||c| each             <=  64 u Q
//   split of |c|^2/2 into two bf16 terms, and its own fp32 rounding                   <=  24 u Q
//   accumulation inside the matrix unit: 3 (5) products of <= 17 terms, taken as 2u per term even if it truncates <= 102 (170) u Q
//   position bits (7 low mantissa bits overwritten)                                   <= 256 u Q
// and the reference's D_j (sub, mul, step-1 adds): |D_j - d_j| <= (step + 3) u d_j    <=  19 u Q in T units.
// best - second > 2 (514 + 19) u Q = 1066 u Q (step 16)  =>  D_second - D_best > 0 for every other centroid.
// The kernel asks for 1536 u Q = 1.5 * 2^-14 Q, and for 2^-60 < Q < 2^30 (no distance reaches the reference's start
// value float(UINT_MAX), no split term is flushed); anything else, NaN included, takes the exact path.
// ------------------------------------------------------------------------------------------
constexpr int ENCM_THREADS = 512;
#ifdef CVTMI_ENC_STATS
__device__ unsigned long long g_enc_stats[2];  // (row, m) pairs seen / resolved by the exact chain
__device__ int g_enc_mode;  // timing experiments: 1 = skip the exact chain, 2 = skip the reduction, 8 = count
#endif

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// (best, second) of the union of two (best, second) pairs
__device__ __forceinline__ void top2_merge(float &a1, float &a2, float b1, float b2)
{
    const float lo = fminf(a1, b1);
    a1 = fmaxf(a1, b1);
    a2 = fmaxf(lo, fmaxf(a2, b2));
}

// PERM (STEP == 8): x holds the rows BEFORE the reference's reorder_ (IVFOPQ.cpp:424-439, y[i] = x[perm[i]]); the eight values of a
// sub-quantiser are gathered from the row through perm (wave-uniform indices: scalar loads) instead of read as two float4 -- the
// permuted copy of the rows, 2 x 4 D bytes per row of traffic and a launch, never exists (cvtmi_opq_rotate_encode on a perm model).
template <int STEP, bool PERM = false>
__global__ __launch_bounds__(ENCM_THREADS) void pq_encode_mfma_kernel(const float *__restrict__ x, int64_t n, int D, int M,
                                                                       const float *__restrict__ coarse,
                                                                       const int32_t *__restrict__ list_id,
                                                                       const float *__restrict__ books,
                                                                       uint8_t *__restrict__ codes, int32_t *__restrict__ list_out,
                                                                       const int32_t *__restrict__ perm = nullptr)
{
    static_assert(!PERM || STEP == 8, "the gathered read is wave-uniform only when a lane owns all 8 values of a sub-quantiser");
    constexpr int WAVES = ENCM_THREADS / 64;
    constexpr int LOFF = STEP == 16 ? 8 : 0;  // step 16: a lane half owns dimensions [8 * (lane >> 5), +8); step 8: all 8
    using f32x16 = __attribute__((ext_vector_type(16))) float;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16 *c1 = reinterpret_cast<__bf16 *>(smem_raw);                 // [M][256][STEP]  first bf16 term of the codebook
    __bf16 *c2 = c1 + (size_t)M * 256 * STEP;                          // [M][256][STEP]  second term
    uint32_t *nhc = reinterpret_cast<uint32_t *>(c2 + (size_t)M * 256 * STEP);  // [M][256]  -|c|^2/2 as two bf16 terms
    uint32_t *cmax = nhc + M * 256;                                    // [M]  max_j |c|^2 (non-negative: bits order as uint)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    if (tid < M) cmax[tid] = 0u;
    __syncthreads();
    for (int c = tid; c < M * 256; c += ENCM_THREADS) {
        float s = 0.0f;
#pragma unroll
        for (int kk = 0; kk < STEP; ++kk) {
            const float v = books[(size_t)c * STEP + kk];
            s = __fmaf_rn(v, v, s);
            const __bf16 h = (__bf16)v;
            c1[(size_t)c * STEP + kk] = h;
            c2[(size_t)c * STEP + kk] = (__bf16)(v - (float)h);
        }
        const float nh = -0.5f * s;
        const __bf16 h1 = (__bf16)nh, h2 = (__bf16)(nh - (float)h1);
        nhc[c] = (uint32_t)__builtin_bit_cast(unsigned short, h1) | ((uint32_t)__builtin_bit_cast(unsigned short, h2) << 16);
        atomicMax(&cmax[c >> 8], __float_as_uint(s));  // a NaN / inf codebook ends up as NaN / inf here: exact path
    }
    __syncthreads();

    const int64_t nbatch = (n + 31) / 32;
    for (int64_t batch = (int64_t)blockIdx.x * WAVES + wave; batch < nbatch; batch += (int64_t)gridDim.x * WAVES) {
        const int64_t row = batch * 32 + li;
        const int64_t rowc = row < n ? row : n - 1;  // clamped: tail rows are computed, never stored
        int l = list_id ? list_id[rowc] : 0;
        if (l < 0) l = 0;  // all-NaN row: see pq_encode_kernel
        const float *xp = x + rowc * D + lk * LOFF;
        const float *cp = coarse + (int64_t)l * D + lk * LOFF;
        uint32_t packed[4] = { 0u, 0u, 0u, 0u };
        float norm2 = 0.0f;  // |x - coarse[0]|^2 up to rounding: decides the single-list assignment (list_out)
#ifdef CVTMI_ENC_STATS
        const int dbg_mode = g_enc_mode;
#endif
        // Software pipeline over the 2 M units (sub-quantiser m, centroid half h): the products of the NEXT unit are
        // issued between the ~200 VALU instructions that reduce the CURRENT unit's 64 accumulators.
        float xv[8], cv[8], r[8], rn[8];
        auto fetch = [&](int m) {
            if constexpr (PERM) {
                // Both lane halves need all eight values (K = 0..7 against r1, 8..15 against r2), but a gathered dword load costs the
                // texture path a cycle per row whatever its width: each half fetches four (its own scalar-selected index) and
                // v_permlane32_swap hands them across -- 4 loads + 4 swaps per sub-quantiser instead of 8 loads
                const int32_t *pm = perm + m * STEP;   // uniform address
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i0 = pm[j], i1 = pm[4 + j];
                    const uint32_t a = __float_as_uint(xp[lk ? i1 : i0]);
                    const auto sw = __builtin_amdgcn_permlane32_swap(a, a, false, false);   // [0]: low half's value everywhere, [1]: high half's
                    xv[j] = __uint_as_float(sw[0]);
                    xv[4 + j] = __uint_as_float(sw[1]);
                }
#pragma unroll
                for (int q = 0; q < 8; q += 4) *reinterpret_cast<float4 *>(&cv[q]) = *reinterpret_cast<const float4 *>(cp + m * STEP + q);
            } else {
#pragma unroll
                for (int q = 0; q < 8; q += 4) {
                    *reinterpret_cast<float4 *>(&xv[q]) = *reinterpret_cast<const float4 *>(xp + m * STEP + q);
                    *reinterpret_cast<float4 *>(&cv[q]) = *reinterpret_cast<const float4 *>(cp + m * STEP + q);
                }
            }
        };
        // the row side of the products: r = r1 + r2 (+ 2^-18 |r|) in bf16
        auto split_row = [&](const float (&rv)[8], bf16x8 &r1, bf16x8 &r2) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const __bf16 h = (__bf16)rv[kk];
                r1[kk] = h;
                r2[kk] = (__bf16)(rv[kk] - (float)h);
            }
        };
        // D[centroid i][row j]: A = centroids (i = lane & 31, k = 8 * (lane >> 5) ..), B = rows
        auto products = [&](int m, int half, const bf16x8 &r1, const bf16x8 &r2, f32x16 (&acc)[4]) {
            const f32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            const bf16x8 bzero = { 0, 0, 0, 0, 0, 0, 0, 0 };
            bf16x8 ones = bzero;  // the row side of the -|c|^2/2 product: (1, 1, 0, ...) in the low lane half only
            ones[0] = lk ? (__bf16)0.0f : (__bf16)1.0f;
            ones[1] = ones[0];
            bf16x8 a1[4], a2[4], ab[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = (half * 4 + t) * 32 + li;
                const size_t off = (size_t)(m * 256 + j) * STEP + lk * LOFF;
                a1[t] = *reinterpret_cast<const bf16x8 *>(c1 + off);
                a2[t] = *reinterpret_cast<const bf16x8 *>(c2 + off);
                const uint32_t hv = nhc[m * 256 + j];  // unconditional read, then a select: no branch inside the pipeline
                union { uint32_t u[4]; bf16x8 v; } cvt = { { lk ? 0u : hv, 0u, 0u, 0u } };
                ab[t] = cvt.v;
            }
            if constexpr (STEP == 8) {
                const bf16x8 rb = lk ? r2 : r1;  // K = 0..7: c_x . r1, K = 8..15: c_x . r2
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], rb, zero, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[t], rb, acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], r1, zero, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], r2, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[t], r1, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[t], r2, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[t], ones, acc[t], 0, 0, 0);  // - |c|^2 / 2
        };
        const float pinf = __uint_as_float(0x7f800000u | (uint32_t)(n < 0));  // +inf the optimiser cannot see through:
        const float ninf = -pinf;                                             // v_med3(a, b, +inf) = max without a canonicalising pre-pass
        auto reduce = [&](const f32x16 (&acc)[4], float &b1, float &b2) {
            // two independent (best, second) chains, tiles {0, 1} and {2, 3}, interleaved: back-to-back VALU
            // instructions never depend on each other
            float d1 = ninf, d2 = ninf;
            b1 = ninf;
            b2 = ninf;
#ifdef CVTMI_ENC_STATS
            if (dbg_mode & 2) { b1 = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]; return; }
#endif
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float k0 = __uint_as_float((__float_as_uint(acc[t][e]) & 0xffffffc0u) | (uint32_t)(t * 16 + e));
                    const float k1 = __uint_as_float((__float_as_uint(acc[t + 2][e]) & 0xffffffc0u) | (uint32_t)((t + 2) * 16 + e));
                    b2 = __builtin_amdgcn_fmed3f(b1, b2, k0);
                    d2 = __builtin_amdgcn_fmed3f(d1, d2, k1);
                    b1 = __builtin_amdgcn_fmed3f(b1, k0, pinf);
                    d1 = __builtin_amdgcn_fmed3f(d1, k1, pinf);
                }
            // union of the two chains: second = max(min(b1, d1), max(b2, d2)), via med3 with the opaque infinities
            const float lo = __builtin_amdgcn_fmed3f(b1, d1, ninf), s2 = __builtin_amdgcn_fmed3f(b2, d2, pinf);
            b1 = __builtin_amdgcn_fmed3f(b1, d1, pinf);
            b2 = __builtin_amdgcn_fmed3f(lo, s2, pinf);
        };
        constexpr int NPROD = 4 * (STEP == 8 ? 3 : 5);
        auto interleave = [&]() {  // one product, then its share of the 3 x 64 reduction instructions
#pragma unroll
            for (int i = 0; i < NPROD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (200 + NPROD - 1) / NPROD, 0);
            }
        };
        f32x16 accA[4], accB[4];
        bf16x8 r1, r2, n1, n2;
        fetch(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) r[kk] = __fsub_rn(xv[kk], cv[kk]);  // the reference's residual (:135-139)
        split_row(r, r1, r2);
        fetch(M > 1 ? 1 : 0);
        products(0, 0, r1, r2, accA);
#pragma unroll 1
        for (int m = 0; m < M; ++m) {
            float best[2], second[2];
            products(m, 1, r1, r2, accB);
            reduce(accA, best[0], second[0]);
            interleave();
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) rn[kk] = __fsub_rn(xv[kk], cv[kk]);
            split_row(rn, n1, n2);
            fetch(m + 2 < M ? m + 2 : M - 1);                     // clamped instead of branched: the pipeline stays one basic block
            products(m + 1 < M ? m + 1 : M - 1, 0, n1, n2, accA);  // (the last round's products are discarded)
            reduce(accB, best[1], second[1]);
            interleave();

            float rr = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) rr = __fmaf_rn(r[kk], r[kk], rr);
            if (STEP == 16) rr += __shfl_xor(rr, 32, 64);
            norm2 += rr;
            const float Q = (rr + __uint_as_float(cmax[m])) * 1.001f;
            // halves of the centroid range, then the two lane halves; the winner's origin rides in bit 6 / is compared
            const bool hiwin = best[1] > best[0];
            const float w1 = __uint_as_float((__float_as_uint(hiwin ? best[1] : best[0]) & ~64u) | (hiwin ? 64u : 0u));
            const float w2 = fmaxf(fminf(best[0], best[1]), fmaxf(second[0], second[1]));
            const float p1 = __shfl_xor(w1, 32, 64), p2 = __shfl_xor(w2, 32, 64);
            const int win_lk = p1 > w1 ? (lk ^ 1) : lk;
            float f1 = w1, f2 = w2;
            top2_merge(f1, f2, p1, p2);
            const uint32_t pos = __float_as_uint(f1) & 127u;  // bit 6: centroid half, bits 5..4: tile, bits 3..0: accumulator element
            const int e = (int)(pos & 15u);
            int code = (int)(pos >> 4) * 32 + (e & 3) + 8 * (e >> 2) + 4 * win_lk;
            const bool sure = (f1 - f2 > Q * 0x1.8p-14f) && (Q < 0x1p30f) && (Q > 0x1p-60f);  // false for NaN anywhere
            uint64_t todo = __ballot(!sure) & 0xffffffffull;  // both lane halves agree: rows = low half
#ifdef CVTMI_ENC_STATS
            if (lane == 0 && (dbg_mode & 8)) { atomicAdd(&g_enc_stats[0], 32ull); atomicAdd(&g_enc_stats[1], (unsigned long long)__popcll(todo)); }
            if (dbg_mode & 1) todo = 0;
#endif
            while (todo) {
                const int rl = __ffsll((unsigned long long)todo) - 1;
                todo &= todo - 1;
                // the reference's loop for row rl (IVFOPQ.cpp:141-161): 4 centroids per lane (fp32 codebook through
                // the caches: the LDS copy is the bf16 split), then the first minimum
                float rs[STEP];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    rs[kk] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r[kk]), rl));
                    if (STEP == 16) rs[(8 + kk) % STEP] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r[kk]), rl + 32));
                }
                // lane owns centroids 4 * lane .. 4 * lane + 3: lane order = centroid order, so "first minimum" is the
                // lowest lane holding the wave-wide minimum, and inside the lane the strict '<' in ascending j
                float bd = kStartDist;
                int bi = 0;
                const float *c = books + (size_t)(m * 256 + 4 * lane) * STEP;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float cf[STEP];
#pragma unroll
                    for (int q = 0; q < STEP; q += 4) *reinterpret_cast<float4 *>(&cf[q]) = *reinterpret_cast<const float4 *>(c + i * STEP + q);
                    float d = 0.0f;
#pragma unroll
                    for (int kk = 0; kk < STEP; ++kk) {
                        const float t = __fsub_rn(rs[kk], cf[kk]);
                        d = __fadd_rn(d, __fmul_rn(t, t));
                    }
                    if (d < bd) { bd = d; bi = i; }  // NaN never wins, as in the reference
                }
                // wave minimum of bd (no NaN among them): 4 DPP steps inside each row of 16 lanes, then the 4 rows
                float v = bd;
                v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
                v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
                v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xf, 0xf, false)));  // row_half_mirror
                v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xf, 0xf, false)));  // row_mirror
                const float dmin = fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)),
                                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))),
                                         fminf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)),
                                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48))));
                int exact = 255;  // nothing beat the start value: (uchar)-1
                if (dmin < kStartDist) {
                    const int wl = __ffsll((unsigned long long)__ballot(bd == dmin)) - 1;
                    exact = 4 * wl + __builtin_amdgcn_readlane(bi, wl);
                }
                if (li == rl) code = exact;
            }
            const uint32_t b = (uint32_t)(code & 0xff) << (8 * (m & 3));
            packed[0] |= (m >> 2) == 0 ? b : 0u;
            packed[1] |= (m >> 2) == 1 ? b : 0u;
            packed[2] |= (m >> 2) == 2 ? b : 0u;
            packed[3] |= (m >> 2) == 3 ? b : 0u;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) r[kk] = rn[kk];
            r1 = n1;
            r2 = n2;
        }
        if (list_out && lk == 0 && row < n) {
            // coarseK == 1 (IVFOPQ.cpp:110-129 with one centroid): list 0 when the sequential fp32 sum of (x - c0)^2 stays
            // below the start value float(UINT_MAX), else -1.  norm2 is that sum up to ~1e-5 relative: far from 2^32 it
            // decides; otherwise (huge or non-finite rows) the lane walks the reference's chain.
            int lst = 0;
            if (!(norm2 * 1.01f < 0x1p31f)) {
                float acc = 0.0f;
                for (int kk = 0; kk < D; ++kk) {
                    const float t = __fsub_rn(x[row * D + kk], coarse[kk]);
                    acc = __fadd_rn(acc, __fmul_rn(t, t));
                }
                lst = acc < kStartDist ? 0 : -1;
            }
            list_out[row] = lst;
        }
        if (lk == 0 && row < n) {
            uint8_t *dst = codes + row * M;
            if (M == 16) *reinterpret_cast<uint4 *>(dst) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            else if (M == 8) *reinterpret_cast<uint2 *>(dst) = make_uint2(packed[0], packed[1]);
            else
                for (int mm = 0; mm < M; ++mm) dst[mm] = (uint8_t)(packed[mm >> 2] >> (8 * (mm & 3)));
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

static int encm_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    return cus;
}

static size_t encm_lds(const OpqModelDev &m) { return ((size_t)m.M * 256 * m.step + (size_t)m.M * 256 + m.M) * sizeof(float); }
static bool encm_ok(const OpqModelDev &m, const float *x_rot)
{
    return m.K == 256 && (m.step == 8 || m.step == 16) && m.M <= 16 && m.D == m.M * m.step && encm_lds(m) <= 160 * 1024 &&
           ((((uintptr_t)x_rot) | ((uintptr_t)m.books) | ((uintptr_t)m.coarse)) & 15) == 0;
}
// does launch_pq_encode take the matrix-core kernel (which can also write the single-list assignment)?
// Smallest call the matrix-core encode takes when the dispatch chooses.  It was 8192 rows -- and the reference's own call shape is
// IVFOPQ::Add of ONE video, a few hundred frames (opq/src/IVFOPQ.cpp:135-163): those went to the VALU kernel, whose 1024-row
// workgroups cost 0.7 ms whatever the row count, against 0.04-0.06 ms here (round 5, tools/sweep_encode_dispatch.py: 64 ... 4096 rows,
// M = 16 and 8: 13-19x).
constexpr int64_t ENCM_MIN_ROWS = 1;
bool pq_encode_fuses_lists(const OpqModelDev &m, const float *x_rot, int64_t n, int variant)
{
    return m.coarseK == 1 && encm_ok(m, x_rot) && (variant == 2 || (variant == 0 && n >= ENCM_MIN_ROWS));
}

// variant: 0 = choose, 1 = the VALU kernel (reference chain for every centroid), 2 = matrix-core filter + exact resolution
// single_list_out: coarseK == 1 and pq_encode_fuses_lists(): the kernel also writes the list assignment (0 / -1)
// can launch_pq_encode read the rows through the model's permutation (x = rows before reorder_)?
bool pq_encode_takes_perm(const OpqModelDev &m, const float *x, int64_t n, int variant)
{
    return m.perm && m.step == 8 && encm_ok(m, x) && (variant == 2 || (variant == 0 && n >= ENCM_MIN_ROWS));
}

int launch_pq_encode(const OpqModelDev &m, const float *x_rot, int64_t n, const int32_t *list_id, uint8_t *codes,
                     hipStream_t st, int variant, int32_t *single_list_out, const int32_t *perm)
{
    if (n <= 0) return CVTMI_OK;
    if (perm && !pq_encode_takes_perm(m, x_rot, n, variant)) return fail(CVTMI_EINVAL, "pq_encode: gathered rows without the matrix-core kernel");
    if (m.K > 256) return fail(CVTMI_EUNSUPPORTED, "pq_encode: K=%d > 256", m.K);
    const size_t lds_mfma = encm_lds(m);
    const bool mfma_ok = encm_ok(m, x_rot);
    if (variant == 2 && !mfma_ok) return fail(CVTMI_EUNSUPPORTED, "pq_encode: the matrix-core encode needs K = 256, step 8 or 16, M <= 16, D <= 128");
    if (single_list_out && !pq_encode_fuses_lists(m, x_rot, n, variant)) return fail(CVTMI_EINVAL, "pq_encode: list output without the fused kernel");
    if (mfma_ok && (variant == 2 || (variant == 0 && n >= ENCM_MIN_ROWS))) {
        constexpr int waves = ENCM_THREADS / 64;
        const int64_t nbatch = (n + 31) / 32;
        const int64_t blocks = std::min<int64_t>(encm_cus(), (nbatch + waves - 1) / waves);
        if (perm) {
            CVTMI_HIP(hipFuncSetAttribute((const void *)pq_encode_mfma_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma));
            hipLaunchKernelGGL((pq_encode_mfma_kernel<8, true>), dim3((unsigned)blocks), dim3(ENCM_THREADS), lds_mfma, st, x_rot, n, m.D, m.M,
                               m.coarse, list_id, m.books, codes, single_list_out, perm);
        } else if (m.step == 8) {
            CVTMI_HIP(hipFuncSetAttribute((const void *)pq_encode_mfma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma));
            hipLaunchKernelGGL((pq_encode_mfma_kernel<8>), dim3((unsigned)blocks), dim3(ENCM_THREADS), lds_mfma, st, x_rot, n, m.D, m.M,
                               m.coarse, list_id, m.books, codes, single_list_out, nullptr);
        } else {
            CVTMI_HIP(hipFuncSetAttribute((const void *)pq_encode_mfma_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma));
            hipLaunchKernelGGL((pq_encode_mfma_kernel<16>), dim3((unsigned)blocks), dim3(ENCM_THREADS), lds_mfma, st, x_rot, n, m.D, m.M,
                               m.coarse, list_id, m.books, codes, single_list_out, nullptr);
        }
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
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

#ifdef CVTMI_ENC_STATS
extern "C" int cvtmi_debug_encode_mode(int mode) { return hipMemcpyToSymbol(HIP_SYMBOL(g_enc_mode), &mode, sizeof mode) == hipSuccess ? 0 : -3; }
extern "C" int cvtmi_debug_encode_stats(unsigned long long *out, int reset)
{
    unsigned long long z[2] = { 0, 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_enc_stats), sizeof z) != hipSuccess) return -3;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_enc_stats), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------
// stand-alone LUT: lut[nq][M][K]  (the scan kernel builds its own copy straight into LDS)
// ------------------------------------------------------------------------------------------
// ld = entries per (query, m) row in the output: K (the API layout) or 256 with +inf padding (scan scratch)
__global__ __launch_bounds__(kBlock) void lut_kernel(const float *__restrict__ q_rot, int D, int M, int K, int step,
                                                     const float *__restrict__ coarse,
                                                     const int32_t *__restrict__ list_id,
                                                     const float *__restrict__ books, float *__restrict__ lut, int ld, int tables)
{
    extern __shared__ __attribute__((aligned(16))) float res[];  // D floats
    const int64_t qi = blockIdx.x;
    int l = list_id ? list_id[qi] : 0;
    if (l < 0) l = 0;
    for (int d = threadIdx.x; d < D; d += kBlock) res[d] = __fsub_rn(q_rot[qi * D + d], coarse[(int64_t)l * D + d]);
    __syncthreads();
    for (int e = threadIdx.x; e < tables * ld; e += kBlock) {
        const int m = e / ld, j = e - m * ld;
        float acc = __uint_as_float(0x7f800000u);
        if (m >= M) {
            acc = 0.0f;   // appended table: all zeros
        } else if (j < K) {
            const float *c = books + ((int64_t)m * K + j) * step;
            acc = 0.0f;
            for (int kk = 0; kk < step; ++kk) {
                const float t = __fsub_rn(res[m * step + kk], c[kk]);
                acc = __fadd_rn(acc, __fmul_rn(t, t));
            }
        }
        lut[qi * tables * ld + e] = acc;
    }
}

// The same tables for sub-vectors of 8 (the usual shape; round 6: also 16 and 32 = M = 8 / M = 4 at D = 128), QB queries per workgroup: a
// thread fetches its centroid once, in 16-byte loads, and uses it for all QB queries -- the kernel above reads the 128 KB of codebooks once
// per query, a dword at a time (10 000 queries: 134 us, 1.2 TB/s of table writes; M = 4: 177 us).  Same operations in the same order per entry.
template <int QB, int STEP = 8>
__global__ __launch_bounds__(kBlock) void lut8_kernel(const float *__restrict__ q_rot, int64_t nq, int D, int M, int K, const float *__restrict__ coarse,
                                                      const int32_t *__restrict__ list_id, const float *__restrict__ books, float *__restrict__ lut, int ld,
                                                      int tables)
{
    extern __shared__ __attribute__((aligned(16))) float res[];  // [QB][D]
    const int64_t q0 = (int64_t)blockIdx.x * QB;
    for (int i = threadIdx.x; i < QB * D; i += kBlock) {
        const int q = i / D, d = i - q * D;
        const int64_t qi = q0 + q < nq ? q0 + q : nq - 1;
        int l = list_id ? list_id[qi] : 0;
        if (l < 0) l = 0;
        res[i] = __fsub_rn(q_rot[qi * D + d], coarse[(int64_t)l * D + d]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < tables * ld; e += kBlock) {
        const int m = e / ld, j = e - m * ld;
        float acc[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] = m >= M ? 0.0f : __uint_as_float(0x7f800000u);   // (appended tables: all zeros)
        if (m < M && j < K) {
            const float4 *c = reinterpret_cast<const float4 *>(books + ((int64_t)m * K + j) * STEP);
            float cv[STEP];
#pragma unroll
            for (int w = 0; w < STEP / 4; ++w) {
                const float4 c4 = c[w];
                cv[4 * w] = c4.x; cv[4 * w + 1] = c4.y; cv[4 * w + 2] = c4.z; cv[4 * w + 3] = c4.w;
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                float a = 0.0f;
#pragma unroll
                for (int kk = 0; kk < STEP; ++kk) {
                    const float t = __fsub_rn(res[q * D + m * STEP + kk], cv[kk]);
                    a = __fadd_rn(a, __fmul_rn(t, t));
                }
                acc[q] = a;
            }
        }
#pragma unroll
        for (int q = 0; q < QB; ++q)
            if (q0 + q < nq) lut[(q0 + q) * tables * ld + e] = acc[q];
    }
}

int launch_lut(const OpqModelDev &m, const float *q_rot, int64_t nq, const int32_t *list_id, float *lut, hipStream_t st,
               int ld, int tables)
{
    if (ld <= 0) ld = m.K;
    if (tables < m.M) tables = m.M;
    if (nq <= 0) return CVTMI_OK;
    if (nq > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "lut: nq too large");
    if ((m.step == 8 || m.step == 16 || m.step == 32) && nq >= 64 && (((uintptr_t)m.books) & 15) == 0) {
        constexpr int QB = 4;
        const dim3 g((unsigned)((nq + QB - 1) / QB));
        const size_t lds = (size_t)QB * m.D * sizeof(float);
        if (m.step == 8) hipLaunchKernelGGL((lut8_kernel<QB, 8>), g, dim3(kBlock), lds, st, q_rot, nq, m.D, m.M, m.K, m.coarse, list_id, m.books, lut, ld, tables);
        else if (m.step == 16) hipLaunchKernelGGL((lut8_kernel<QB, 16>), g, dim3(kBlock), lds, st, q_rot, nq, m.D, m.M, m.K, m.coarse, list_id, m.books, lut, ld, tables);
        else hipLaunchKernelGGL((lut8_kernel<QB, 32>), g, dim3(kBlock), lds, st, q_rot, nq, m.D, m.M, m.K, m.coarse, list_id, m.books, lut, ld, tables);
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    hipLaunchKernelGGL(lut_kernel, dim3((unsigned)nq), dim3(kBlock), (size_t)m.D * sizeof(float), st, q_rot, m.D, m.M,
                       m.K, m.step, m.coarse, list_id, m.books, lut, ld, tables);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
