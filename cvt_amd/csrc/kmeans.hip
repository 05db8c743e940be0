// kmeans.hip -- codebook training: TrainPQ::CoarseQuan / ProdQuan
// (opq/train_codebook/train_PQ_codebook.cpp:150-244).  The reference calls yael's kmeans(), which is not
// vendored (parity unpinned); what is implemented is the fully specified Lloyd iteration of include/cvtmi.h
// (the tests hold it bit-exact against the CPU checker): splitmix64 seeding, assignment with the arithmetic of
// IVFOPQ::Add (sequential fp32 distance, strict '<'), centroid = float(double sum in ascending row order /
// count), empty clusters keep their centroid, stop when an assignment pass changes nothing.
//
//  * assign, 32 <= d <= 128: bf16 matrix-core filter with a proven bound, exact kernel for the rows it cannot decide
//    (assign_mfma.hip).  Otherwise, and for those rows:
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
    if (changed) {
        const unsigned long long m = __ballot(ch);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(changed, (unsigned long long)__popcll(m));
    }
}

// d <= 128: the row stays in registers for the whole launch (the kernel above re-reads it from memory for every
// tile of 16 centroids, one 4-byte load per dimension and lane at a row stride: as many loads as arithmetic),
// and two centroids ride in the halves of every packed-fp32 register (x broadcast to both).  Same sub, mul, add
// per (row, centroid, dimension) in ascending dimension order, same first-minimum rule: same bits.
template <int DMAX>
__global__ __launch_bounds__(kBlock) void kmeans_assign_reg_kernel(const float *__restrict__ x, int64_t ld, int64_t n, int d,
                                                                   const float *__restrict__ cent, int k,
                                                                   int32_t *__restrict__ assign,
                                                                   unsigned long long *__restrict__ changed, int csplit,
                                                                   float *__restrict__ part_d, int32_t *__restrict__ part_i)
{
    // part_d != null: blockIdx.y walks only centroids [y * csplit, (y + 1) * csplit) and leaves its (best distance, index)
    // in part_*[y][row]; the caller folds the ranges in ascending order with the same strict '<' (few rows, many centroids)
    __shared__ __attribute__((aligned(16))) float cen[DMAX][KM_CT];  // transposed: [dim][centroid]
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = row < n;
    const float *xr = x + (valid ? row : 0) * ld;
    float xv[DMAX];
#pragma unroll
    for (int dd = 0; dd < DMAX; ++dd) xv[dd] = dd < d ? xr[dd] : 0.0f;
    float best = kKmStart;
    int bi = -1;
    const int c_lo = part_d ? (int)blockIdx.y * csplit : 0;
    if (part_d) k = k < c_lo + csplit ? k : c_lo + csplit;
    for (int c0 = c_lo; c0 < k; c0 += KM_CT) {
        __syncthreads();
        for (int i = threadIdx.x; i < DMAX * KM_CT; i += kBlock) {
            const int c = i / DMAX, dd = i - c * DMAX;  // coalesced along the dimension
            float v = 0.0f;
            if (c0 + c < k && dd < d) v = cent[(int64_t)(c0 + c) * d + dd];
            cen[dd][c] = v;
        }
        __syncthreads();
        float2 acc[KM_CT / 2];
#pragma unroll
        for (int p = 0; p < KM_CT / 2; ++p) acc[p] = make_float2(0.0f, 0.0f);
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd) {
            if (dd < d) {  // wave-uniform
                const float2 xb = make_float2(xv[dd], xv[dd]);
                const float4 *cp = reinterpret_cast<const float4 *>(&cen[dd][0]);
#pragma unroll
                for (int g = 0; g < KM_CT / 4; ++g) {
                    const float4 c4 = cp[g];  // broadcast read: 4 centroids of this dimension
                    const float2 t0 = xb - make_float2(c4.x, c4.y), t1 = xb - make_float2(c4.z, c4.w);
                    acc[2 * g] = acc[2 * g] + t0 * t0;
                    acc[2 * g + 1] = acc[2 * g + 1] + t1 * t1;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < KM_CT / 2; ++p) {
            if (c0 + 2 * p < k && acc[p].x < best) { best = acc[p].x; bi = c0 + 2 * p; }
            if (c0 + 2 * p + 1 < k && acc[p].y < best) { best = acc[p].y; bi = c0 + 2 * p + 1; }
        }
    }
    if (part_d) {
        if (valid) {
            part_d[(int64_t)blockIdx.y * n + row] = best;
            part_i[(int64_t)blockIdx.y * n + row] = bi;
        }
        return;
    }
    bool ch = false;
    if (valid) {
        ch = assign[row] != bi;
        assign[row] = bi;
    }
    if (changed) {
        const unsigned long long m = __ballot(ch);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(changed, (unsigned long long)__popcll(m));
    }
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

static int g_assign_variant = 0;  // 0 = choose, 1 = exact kernels for every row, 2 = matrix-core filter wherever it applies
void set_assign_variant(int v) { g_assign_variant = v; }

int launch_kmeans_assign(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k, int32_t *assign,
                         unsigned long long *changed, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    // 32 <= d <= 128: matrix-core filter + exact resolution of the rows it cannot decide (assign_mfma.hip); same result
    if (g_assign_variant != 1 && assign_filter_applies(x, ld, g_assign_variant == 2 && n < 4096 ? 4096 : n, d, cent, k) && n < 0x7fffffff)
        return launch_assign_filtered(x, ld, n, d, cent, k, assign, changed, st);
    return launch_kmeans_assign_exact(x, ld, n, d, cent, k, assign, changed, st);
}

int launch_kmeans_assign_exact(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k, int32_t *assign,
                               unsigned long long *changed, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "kmeans: n too large");
    const dim3 g((unsigned)blocks), b(kBlock);
    if (d <= 8) hipLaunchKernelGGL(kmeans_assign_reg_kernel<8>, g, b, 0, st, x, ld, n, d, cent, k, assign, changed, 0, nullptr, nullptr);
    else if (d <= 16) hipLaunchKernelGGL(kmeans_assign_reg_kernel<16>, g, b, 0, st, x, ld, n, d, cent, k, assign, changed, 0, nullptr, nullptr);
    else if (d <= 32) hipLaunchKernelGGL(kmeans_assign_reg_kernel<32>, g, b, 0, st, x, ld, n, d, cent, k, assign, changed, 0, nullptr, nullptr);
    else if (d <= 64) hipLaunchKernelGGL(kmeans_assign_reg_kernel<64>, g, b, 0, st, x, ld, n, d, cent, k, assign, changed, 0, nullptr, nullptr);
    else if (d <= 128) hipLaunchKernelGGL(kmeans_assign_reg_kernel<128>, g, b, 0, st, x, ld, n, d, cent, k, assign, changed, 0, nullptr, nullptr);
    else hipLaunchKernelGGL(kmeans_assign_kernel, g, b, 0, st, x, ld, n, d, cent, k, assign, changed);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// few rows against many centroids (the rows the matrix-core filter of assign_mfma.hip could not decide): the centroid
// range is cut into `splits` pieces that run as separate workgroups, then folded in ascending order -- the same strict '<'
__global__ __launch_bounds__(kBlock) void kmeans_assign_fold_kernel(const float *__restrict__ part_d, const int32_t *__restrict__ part_i,
                                                                    int64_t n, int splits, int32_t *__restrict__ assign)
{
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= n) return;
    float best = kKmStart;
    int bi = -1;
    for (int s = 0; s < splits; ++s) {
        const float dd = part_d[(int64_t)s * n + row];
        if (dd < best) { best = dd; bi = part_i[(int64_t)s * n + row]; }
    }
    assign[row] = bi;
}

int launch_kmeans_assign_split(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k, int32_t *assign, int splits,
                               float *part_d, int32_t *part_i, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    if (d > 128) return fail(CVTMI_EUNSUPPORTED, "kmeans_assign_split: d=%d > 128", d);
    const int csplit = ((k + splits - 1) / splits + KM_CT - 1) / KM_CT * KM_CT;
    splits = (k + csplit - 1) / csplit;
    const dim3 g((unsigned)((n + kBlock - 1) / kBlock), (unsigned)splits), b(kBlock);
    if (d <= 8) hipLaunchKernelGGL(kmeans_assign_reg_kernel<8>, g, b, 0, st, x, ld, n, d, cent, k, assign, nullptr, csplit, part_d, part_i);
    else if (d <= 16) hipLaunchKernelGGL(kmeans_assign_reg_kernel<16>, g, b, 0, st, x, ld, n, d, cent, k, assign, nullptr, csplit, part_d, part_i);
    else if (d <= 32) hipLaunchKernelGGL(kmeans_assign_reg_kernel<32>, g, b, 0, st, x, ld, n, d, cent, k, assign, nullptr, csplit, part_d, part_i);
    else if (d <= 64) hipLaunchKernelGGL(kmeans_assign_reg_kernel<64>, g, b, 0, st, x, ld, n, d, cent, k, assign, nullptr, csplit, part_d, part_i);
    else hipLaunchKernelGGL(kmeans_assign_reg_kernel<128>, g, b, 0, st, x, ld, n, d, cent, k, assign, nullptr, csplit, part_d, part_i);
    hipLaunchKernelGGL(kmeans_assign_fold_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), b, 0, st, part_d, part_i, n, splits, assign);
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
