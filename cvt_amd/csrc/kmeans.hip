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
//  * update, k <= 512: scatter in any order + a proof that the float result is the index-order one (below); otherwise:
//  * update: the order of the double additions is part of bit-exactness, so there is no atomic scatter:
//    one wave owns one centroid, walks the assignment array 64 rows at a time (ballot of the matches) and
//    folds the matching rows in ascending order, lane = dimension.  k waves re-read n assignments from
//    L2 (k n 4 bytes): for coarseK = 8192, n = 1 M that is ~3 ms per iteration, noise next to the assign pass.
#include <algorithm>

#include <mutex>
#include "host_util.h"
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
    // Few rows against many centroids -- the reference's index build is IVFOPQ::Add of ONE video, a few hundred frames over 8192 lists
    // (opq/src/IVFOPQ.cpp:135-163): with one thread per row walking every centroid such a call took 7.6 ms whatever its size (round 5,
    // tools/sweep_add_video.py).  The centroid range is cut over enough workgroups to fill the chip and folded in ascending order (the
    // same strict '<': same assignment), as the filter does for the rows it cannot decide.  The scratch is shared: the stream is drained
    // before the lock is given back, as in launch_assign_filtered.
    if (g_assign_variant != 1 && changed == nullptr && n < 4096 && k >= 256 && d <= 128) {
        // One grow-only scratch PER DEVICE with its own lock (round 6): handles on different devices (IVFOPQ::SetDevices adds
        // to them in turn) neither wait for each other nor free and reallocate each other's buffer.  Callers on one device still take
        // turns, and the stream is drained under the lock -- this entry is not capturable into a graph, like launch_assign_filtered.
        constexpr int kFewDevs = 16;
        static std::mutex few_mu[kFewDevs + 1];
        static DevBuf few_scr[kFewDevs + 1];   // the last slot: any device number past the table (shared, reallocated on a change of device)
        static int few_other_dev = -1;
        int cur = 0;
        CVTMI_HIP(hipGetDevice(&cur));
        const int slot = cur >= 0 && cur < kFewDevs ? cur : kFewDevs;
        std::lock_guard<std::mutex> guard(few_mu[slot]);
        DevBuf &scr = few_scr[slot];
        if (slot == kFewDevs && cur != few_other_dev) { scr.release(); few_other_dev = cur; }
        const int row_blocks = (int)((n + kBlock - 1) / kBlock);
        const int splits = std::max(1, std::min((k + 63) / 64, (1024 + row_blocks - 1) / row_blocks));
        const size_t b_part = (((size_t)n * splits * sizeof(float)) + 255) & ~(size_t)255;
        CVTMI_TRY(scr.reserve(2 * b_part));
        const int rc = launch_kmeans_assign_split(x, ld, n, d, cent, k, assign, splits, scr.as<float>(),
                                                  reinterpret_cast<int32_t *>(scr.as<char>() + b_part), st);
        CVTMI_HIP(hipStreamSynchronize(st));
        return rc;
    }
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

// ---- update, fast path -------------------------------------------------------------------------------------
// centroid[c][j] = float(double sum over the members of c in ascending row order / count).  The order only matters through
// the final rounding to float: sums accumulated in ANY order differ from the index-order sum by at most
// 2 (count) 2^-53 sum |x| (each of the <= count - 1 additions of either order rounds within 2^-53 of a partial sum that
// is bounded by sum |x|).  So rows are scattered with double atomics in LDS (sum, sum of |x|, count), and where
// float((S - E) / count) == float((S + E) / count) that float is the reference's centroid, proven; where it is not (and
// for non-finite sums) the centroid is flagged and recomputed in index order by the wave-per-centroid kernel.
// k <= 512: the partial sums of a block of rows are kept in LDS (8 dimensions per workgroup column: k x 8 doubles twice),
// one LDS atomic per element, and reach the global arrays once per workgroup
constexpr int KM_LK = 512;   // centroids the LDS variant holds
constexpr int KM_DG = 8;     // dimensions per workgroup column
__global__ __launch_bounds__(kBlock) void kmeans_scatter_lds_kernel(const float *__restrict__ x, int64_t ld, int64_t n, int d,
                                                                    const int32_t *__restrict__ assign, int k, int64_t rows_per_block,
                                                                    double *__restrict__ sum, double *__restrict__ asum,
                                                                    unsigned int *__restrict__ cnt)
{
    extern __shared__ __attribute__((aligned(16))) double km_s[];  // [k][KM_DG] sums, [k][KM_DG] sums of |x|, then k counts
    double *s_sum = km_s, *s_abs = km_s + (size_t)k * KM_DG;
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_abs + (size_t)k * KM_DG);
    const int d0 = blockIdx.y * KM_DG;
    for (int i = threadIdx.x; i < k * KM_DG; i += kBlock) { s_sum[i] = 0.0; s_abs[i] = 0.0; }
    for (int i = threadIdx.x; i < k; i += kBlock) s_cnt[i] = 0u;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < n ? r1 : n;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += kBlock) {
        const int c = assign[r];
        if (c < 0) continue;  // a row no centroid claims
        if (blockIdx.y == 0) atomicAdd(&s_cnt[c], 1u);
#pragma unroll
        for (int j = 0; j < KM_DG; ++j) {
            if (d0 + j < d) {
                const double v = (double)x[r * ld + d0 + j];
                atomicAdd(&s_sum[c * KM_DG + j], v);
                atomicAdd(&s_abs[c * KM_DG + j], fabs(v));
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k * KM_DG; i += kBlock) {
        const int c = i / KM_DG, j = i - c * KM_DG;
        if (d0 + j < d && s_abs[i] != 0.0) {
            atomicAdd(&sum[(int64_t)c * d + d0 + j], s_sum[i]);
            atomicAdd(&asum[(int64_t)c * d + d0 + j], s_abs[i]);
        }
    }
    if (blockIdx.y == 0)
        for (int i = threadIdx.x; i < k; i += kBlock)
            if (s_cnt[i]) atomicAdd(&cnt[i], s_cnt[i]);
}

__global__ __launch_bounds__(kBlock) void kmeans_finalize_kernel(const double *__restrict__ sum, const double *__restrict__ asum,
                                                                 const unsigned int *__restrict__ cnt, int k, int d,
                                                                 float *__restrict__ cent, int *__restrict__ redo)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= (int64_t)k * d) return;
    const int c = (int)(e / d);
    const unsigned int m = cnt[c];
    if (m == 0) return;  // an empty cluster keeps its centroid
    const double S = sum[e], E = 2.0 * (double)m * 0x1p-53 * asum[e] * 1.0000001, dm = (double)m;
    const float lo = (float)__ddiv_rn(S - E, dm), hi = (float)__ddiv_rn(S + E, dm);
    if (lo == hi) cent[e] = lo;  // proven: the index-order sum lies in [S - E, S + E], division and rounding are monotone
    else redo[c] = 1;            // (also NaN / inf): index order decides
}

// index-order recomputation of the flagged centroids (the kernel above, one wave per centroid)
__global__ __launch_bounds__(kBlock) void kmeans_update_flagged_kernel(const float *__restrict__ x, int64_t ld, int64_t n, int d,
                                                                       const int32_t *__restrict__ assign, int k,
                                                                       const int *__restrict__ redo, float *__restrict__ cent)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (c >= k || !redo[c]) return;
    double sum[KM_DPL];
#pragma unroll
    for (int i = 0; i < KM_DPL; ++i) sum[i] = 0.0;
    long long cnt = 0;
    // one wave walks all n assignments: 8 chunks of 64 are loaded at once so that the walk is not one memory latency per chunk
    for (int64_t base0 = 0; base0 < n; base0 += 512) {
        int a8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r = base0 + 64 * u + lane;
            a8[u] = r < n ? assign[r] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t base = base0 + 64 * u;
            unsigned long long m = __ballot(a8[u] == c);
            while (m) {
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
    }
    if (cnt > 0) {
#pragma unroll
        for (int i = 0; i < KM_DPL; ++i) {
            const int dd = lane + 64 * i;
            if (dd < d) cent[(int64_t)c * d + dd] = (float)__ddiv_rn(sum[i], (double)cnt);
        }
    }
}

int launch_kmeans_update(const float *x, int64_t ld, int64_t n, int d, const int32_t *assign, int k, float *cent,
                         hipStream_t st)
{
    if (d > 64 * KM_DPL) return fail(CVTMI_EUNSUPPORTED, "kmeans: d=%d > %d", d, 64 * KM_DPL);
    const int wpb = kBlock / 64;
    const size_t kd = (size_t)k * d;
    const size_t need = kd * 16 + (size_t)k * 8 + 256;
    if (n >= 4096 && k <= KM_LK) {  // fast path: scatter + proof, index order only where the proof fails (larger k: the
                                    // wave-per-centroid kernel below measures faster than scattering with global atomics)
        void *scratch = nullptr;  // stream-ordered: lives exactly as long as the kernels below
        CVTMI_HIP(hipMallocAsync(&scratch, need, st));
        double *sum = static_cast<double *>(scratch), *asum = sum + kd;
        unsigned int *cnt = reinterpret_cast<unsigned int *>(asum + kd);
        int *redo = reinterpret_cast<int *>(cnt + k);
        CVTMI_HIP(hipMemsetAsync(scratch, 0, need, st));
        const int64_t rpb = std::max<int64_t>(2048, (n + 255) / 256);
        const dim3 g((unsigned)((n + rpb - 1) / rpb), (unsigned)((d + KM_DG - 1) / KM_DG));
        const size_t lds = (size_t)k * KM_DG * 16 + (size_t)k * 4;
        CVTMI_HIP(hipFuncSetAttribute((const void *)kmeans_scatter_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kmeans_scatter_lds_kernel, g, dim3(kBlock), lds, st, x, ld, n, d, assign, k, rpb, sum, asum, cnt);
        hipLaunchKernelGGL(kmeans_finalize_kernel, dim3((unsigned)((kd + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, sum, asum, cnt, k, d, cent, redo);
        hipLaunchKernelGGL(kmeans_update_flagged_kernel, dim3((unsigned)((k + wpb - 1) / wpb)), dim3(kBlock), 0, st, x, ld, n, d, assign, k, redo, cent);
        const hipError_t le = hipGetLastError();
        CVTMI_HIP(hipFreeAsync(scratch, st));
        CVTMI_HIP(le);
        return CVTMI_OK;
    }
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
