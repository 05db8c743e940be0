// kmeans.hip -- codebook training: TrainPQ::CoarseQuan / ProdQuan
// (opq/train_codebook/train_PQ_codebook.cpp:150-244).  The reference calls yael's kmeans(), which is not
// vendored (parity unpinned); what is implemented is the fully specified Lloyd iteration of include/cvtmi.h
// (the tests hold it bit-exact against the CPU checker): splitmix64 seeding, assignment with the arithmetic of
// IVFOPQ::Add (sequential fp32 distance, strict '<'), centroid = float(double sum in ascending row order /
// count), empty clusters keep their centroid, stop when an assignment pass changes nothing.
//
//  * assign: one lane per row, CT centroids at a time from an LDS tile transposed [dim][centroid]
//    (broadcast reads), squared distances kept in registers -- VALU fp32 bound, 3 d k flop per row.
//  * update: the order of the double additions is part of bit-exactness, so there is no atomic scatter:
//    one wave owns one centroid, walks the assignment array 64 rows at a time (ballot of the matches) and
//    folds the matching rows in ascending order, lane = dimension.  k waves re-read n assignments from
//    L2 (k n 4 bytes): for coarseK = 8192, n = 1 M that is ~3 ms per iteration, noise next to the assign pass.
#include "kernels.h"

namespace cvtmi {

constexpr int KM_CT = 16;    // centroids per register tile
constexpr int KM_DC = 128;   // dimensions per LDS chunk
constexpr float kKmStart = 4294967296.0f;  // float(UINT_MAX), IVFOPQ.cpp:114

__global__ __launch_bounds__(kBlock) void kmeans_assign_kernel(const float *__restrict__ x, int64_t ld, int64_t n, int d,
                                                               const float *__restrict__ cent, int k,
                                                               int32_t *__restrict__ assign,
                                                               unsigned long long *__restrict__ changed)
{
    __shared__ float cen[KM_DC][KM_CT];  // transposed: [dim][centroid]
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = row < n;
    const float *xr = x + (valid ? row : 0) * ld;
    float best = kKmStart;
    int bi = -1;
    for (int c0 = 0; c0 < k; c0 += KM_CT) {
        float acc[KM_CT];
#pragma unroll
        for (int c = 0; c < KM_CT; ++c) acc[c] = 0.0f;
        for (int d0 = 0; d0 < d; d0 += KM_DC) {
            __syncthreads();
            for (int i = threadIdx.x; i < KM_DC * KM_CT; i += kBlock) {
                const int c = i / KM_DC, dd = i - c * KM_DC;  // coalesced along the dimension
                float v = 0.0f;
                if (c0 + c < k && d0 + dd < d) v = cent[(int64_t)(c0 + c) * d + d0 + dd];
                cen[dd][c] = v;
            }
            __syncthreads();
            const int dl = (d - d0) < KM_DC ? (d - d0) : KM_DC;
            for (int dd = 0; dd < dl; ++dd) {
                const float xv = xr[d0 + dd];
#pragma unroll
                for (int c = 0; c < KM_CT; ++c) {
                    const float t = __fsub_rn(xv, cen[dd][c]);
                    acc[c] = __fadd_rn(acc[c], __fmul_rn(t, t));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < KM_CT; ++c) {
            if (c0 + c < k && acc[c] < best) {
                best = acc[c];
                bi = c0 + c;
            }
        }
    }
    bool ch = false;
    if (valid) {
        ch = assign[row] != bi;
        assign[row] = bi;
    }
    const unsigned long long m = __ballot(ch);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(changed, (unsigned long long)__popcll(m));
}

constexpr int KM_DPL = 8;  // dimensions per lane of the update: d <= 512

__global__ __launch_bounds__(kBlock) void kmeans_update_kernel(const float *__restrict__ x, int64_t ld, int64_t n, int d,
                                                               const int32_t *__restrict__ assign, int k,
                                                               float *__restrict__ cent)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);  // one wave per centroid
    if (c >= k) return;
    double sum[KM_DPL];
#pragma unroll
    for (int i = 0; i < KM_DPL; ++i) sum[i] = 0.0;
    long long cnt = 0;
    for (int64_t base = 0; base < n; base += 64) {
        const int64_t r = base + lane;
        const int a = r < n ? assign[r] : -1;
        unsigned long long m = __ballot(a == c);
        while (m) {  // matching rows in ascending order
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float *xr = x + (base + b) * ld;
#pragma unroll
            for (int i = 0; i < KM_DPL; ++i) {
                const int dd = lane + 64 * i;
                if (dd < d) sum[i] = __dadd_rn(sum[i], (double)xr[dd]);
            }
            ++cnt;
        }
    }
    if (cnt > 0) {
#pragma unroll
        for (int i = 0; i < KM_DPL; ++i) {
            const int dd = lane + 64 * i;
            if (dd < d) cent[(int64_t)c * d + dd] = (float)__ddiv_rn(sum[i], (double)cnt);
        }
    }
}

__global__ __launch_bounds__(kBlock) void kmeans_gather_kernel(const float *__restrict__ x, int64_t ld, int d,
                                                               const int64_t *__restrict__ rows, int k,
                                                               float *__restrict__ cent)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= (int64_t)k * d) return;
    const int c = (int)(e / d), dd = (int)(e - (int64_t)c * d);
    cent[e] = x[rows[c] * ld + dd];
}

__global__ __launch_bounds__(kBlock) void kmeans_fill_kernel(int32_t *p, int64_t n, int32_t v)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = v;
}

// res[r][i] = x[r][i] - cent[assign[r]][i]   (train_PQ_codebook.cpp:190-197; a row nobody claims uses centroid 0)
__global__ __launch_bounds__(kBlock) void kmeans_residual_kernel(const float *__restrict__ x, int64_t n, int d,
                                                                 const float *__restrict__ cent,
                                                                 const int32_t *__restrict__ assign,
                                                                 float *__restrict__ res)
{
    const int rows_per_block = kBlock / 64;
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * rows_per_block) {
        int c = assign[r];
        c = c < 0 ? 0 : c;
        for (int i = lane; i < d; i += 64) res[r * d + i] = __fsub_rn(x[r * d + i], cent[(int64_t)c * d + i]);
    }
}

int launch_kmeans_assign(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k, int32_t *assign,
                         unsigned long long *changed, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "kmeans: n too large");
    hipLaunchKernelGGL(kmeans_assign_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, ld, n, d, cent, k, assign, changed);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_kmeans_update(const float *x, int64_t ld, int64_t n, int d, const int32_t *assign, int k, float *cent,
                         hipStream_t st)
{
    if (d > 64 * KM_DPL) return fail(CVTMI_EUNSUPPORTED, "kmeans: d=%d > %d", d, 64 * KM_DPL);
    const int wpb = kBlock / 64;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)((k + wpb - 1) / wpb)), dim3(kBlock), 0, st, x, ld, n, d, assign, k, cent);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_kmeans_gather(const float *x, int64_t ld, int d, const int64_t *rows, int k, float *cent, hipStream_t st)
{
    const int64_t total = (int64_t)k * d;
    hipLaunchKernelGGL(kmeans_gather_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, x, ld, d, rows, k, cent);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_kmeans_fill(int32_t *p, int64_t n, int32_t v, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    hipLaunchKernelGGL(kmeans_fill_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, p, n, v);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_kmeans_residual(const float *x, int64_t n, int d, const float *cent, const int32_t *assign, float *res,
                           hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    hipLaunchKernelGGL(kmeans_residual_kernel, dim3(2048), dim3(kBlock), 0, st, x, n, d, cent, assign, res);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
