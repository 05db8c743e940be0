// opq_learn.hip -- OPQ rotation learning ("optional"): the dense rotation R that the a-R GEMM applies, LEARNED from a sample
// instead of handed in.  Not in the reference -- opq/ only permutes dimensions (reorder_, opq/src/IVFOPQ.cpp:424-439) and reads that
// permutation from a file; the specification is the oracle's orc_opq_learn_rotation, held bit for bit (tests/test_gpu_train.py).
//
// The non-parametric alternation of Ge et al. ("Optimized Product Quantization", 2013) over pieces the path already has:
//     R = I;  repeat `outer` times:
//         Xr = X R^T                                   launch_rotate_gemm  (fp32 MFMA, == the k-ordered fmaf chain)
//         per sub-space m: books[m], assign[m]         cvtmi_kmeans_dev    (the fully specified Lloyd iteration of csrc/kmeans.hip)
//         Y[r] = concat_m books[m][assign[m][r]]       opq_reconstruct_kernel
//         C = X^T Y in double                          opq_xty_kernel: one workgroup per block of 1024 rows (rows ascending inside a
//                                                      block), then opq_xty_reduce_kernel adds the blocks in ascending order -- the
//                                                      summation order IS the specification, so the device reproduces it exactly
//         C = U S V^T, R = V U^T                       orthogonal Procrustes, one-sided Jacobi in double ON THE HOST (a 128 x 128 matrix:
//                                                      ~30 MFLOP; the device has nothing to add)
//     books = k-means of the final Xr.
// The heavy parts (rotation, assignment, update, X^T Y) stay on the device; only D x D doubles cross PCIe per outer step.
#include <cmath>
#include <vector>

#include "common.h"
#include "host_util.h"
#include "kernels.h"
#include "../../include/cvtmi.h"

namespace cvtmi {

constexpr int XTY_BLOCK = 1024;   // rows per partial sum (the oracle's ORC_XTY_BLOCK)

__global__ __launch_bounds__(kBlock) void opq_reconstruct_kernel(const float *__restrict__ books, const int32_t *__restrict__ assign, int64_t n,
                                                                int D, int M, int K, float *__restrict__ y)
{
    const int step = D / M;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n * D; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / D;
        const int c = (int)(i - r * D), m = c / step, j = c - m * step;
        const int32_t a = assign[(int64_t)m * n + r];
        y[i] = a < 0 ? 0.0f : books[((int64_t)m * K + a) * step + j];
    }
}

// partial[b][i][j] = sum over the rows r of block b, ascending, of (double) x[r][i] * (double) y[r][j]   (the product is exact)
// 256 threads as a 16 x 16 grid of (D / 16) x (D / 16) tiles of C; 32-row slices of x and y pass through LDS.
template <int T>   // T = D / 16: 2, 4, 6, 8
__global__ __launch_bounds__(256) void opq_xty_kernel(const float *__restrict__ x, const float *__restrict__ y, int64_t n, double *__restrict__ part)
{
    constexpr int D = 16 * T, SL = 32;
    __shared__ float xs[SL][D], ys[SL][D];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int64_t r0 = (int64_t)blockIdx.x * XTY_BLOCK, r1 = r0 + XTY_BLOCK < n ? r0 + XTY_BLOCK : n;
    double acc[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = 0.0;
    for (int64_t s0 = r0; s0 < r1; s0 += SL) {
        const int rows = (int)(r1 - s0 < SL ? r1 - s0 : SL);
        __syncthreads();
        for (int i = tid; i < rows * D; i += 256) {
            xs[i / D][i % D] = x[s0 * D + i];
            ys[i / D][i % D] = y[s0 * D + i];
        }
        __syncthreads();
        for (int r = 0; r < rows; ++r) {   // ascending rows: the order of the specification
#pragma unroll
            for (int a = 0; a < T; ++a) {
                const double xi = (double)xs[r][ti * T + a];
#pragma unroll
                for (int b = 0; b < T; ++b) acc[a][b] = __dadd_rn(acc[a][b], __dmul_rn(xi, (double)ys[r][tj * T + b]));
            }
        }
    }
    double *out = part + (int64_t)blockIdx.x * D * D;
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) out[(ti * T + a) * D + tj * T + b] = acc[a][b];
}
// C[e] = sum over the blocks, ascending, of partial[b][e]
__global__ __launch_bounds__(kBlock) void opq_xty_reduce_kernel(const double *__restrict__ part, int64_t nblocks, int DD, double *__restrict__ C)
{
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= DD) return;
    double s = 0.0;
    for (int64_t b = 0; b < nblocks; ++b) s = __dadd_rn(s, part[b * DD + e]);
    C[e] = s;
}

// R = V U^T for C = U S V^T: the oracle's orc_procrustes, operation for operation (one-sided Jacobi on the columns of C, cyclic sweeps,
// a pair is rotated when |a_p . a_q| > 1e-15 sqrt(|a_p|^2 |a_q|^2), at most 60 sweeps).  false (R untouched): zero / non-finite C.
static bool procrustes_host(const double *C, int D, float *R)
{
    std::vector<double> A((size_t)D * D), V((size_t)D * D);
    for (int i = 0; i < D * D; ++i) {
        A[(size_t)i] = C[i];
        if (!(C[i] == C[i]) || C[i] - C[i] != 0.0) return false;
    }
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) V[(size_t)i * D + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < D - 1; ++p)
            for (int q = p + 1; q < D; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int i = 0; i < D; ++i) {
                    const double ap = A[(size_t)i * D + p], aq = A[(size_t)i * D + q];
                    alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
                }
                if (!(std::fabs(gamma) > 1e-15 * std::sqrt(alpha * beta))) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < D; ++i) {
                    const double ap = A[(size_t)i * D + p], aq = A[(size_t)i * D + q];
                    A[(size_t)i * D + p] = c * ap - sn * aq; A[(size_t)i * D + q] = sn * ap + c * aq;
                    const double vp = V[(size_t)i * D + p], vq = V[(size_t)i * D + q];
                    V[(size_t)i * D + p] = c * vp - sn * vq; V[(size_t)i * D + q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double smax = 0.0;
    for (int j = 0; j < D; ++j) {
        double s2 = 0.0;
        for (int i = 0; i < D; ++i) s2 += A[(size_t)i * D + j] * A[(size_t)i * D + j];
        const double sj = std::sqrt(s2);
        if (!(sj == sj) || sj - sj != 0.0) return false;
        if (sj > smax) smax = sj;
    }
    if (!(smax > 0.0)) return false;   // C = 0: nothing to align
    // U: columns with a length are normalised; a rank-deficient C leaves columns without one -- any orthonormal completion maximises
    // tr(R C) equally, so they are completed deterministically (the oracle's rule: the unit vector e_k with the least of its length inside
    // the span of the columns already final, first minimum; two rounds of Gram-Schmidt against them; normalised)
    std::vector<unsigned char> fin((size_t)D, 0);
    for (int j = 0; j < D; ++j) {
        double s2 = 0.0;
        for (int i = 0; i < D; ++i) s2 += A[(size_t)i * D + j] * A[(size_t)i * D + j];
        const double sj = std::sqrt(s2);
        if (sj > 1e-12 * smax) {
            for (int i = 0; i < D; ++i) A[(size_t)i * D + j] = A[(size_t)i * D + j] / sj;
            fin[(size_t)j] = 1;
        }
    }
    for (int j = 0; j < D; ++j) {
        if (fin[(size_t)j]) continue;
        int kbest = 0;   // the unit vector with the least of its length inside the span of the final columns
        double ebest = 0.0;
        for (int k = 0; k < D; ++k) {
            double e2 = 0.0;
            for (int c = 0; c < D; ++c)
                if (fin[(size_t)c]) e2 += A[(size_t)k * D + c] * A[(size_t)k * D + c];
            if (k == 0 || e2 < ebest) { ebest = e2; kbest = k; }
        }
        for (int i = 0; i < D; ++i) A[(size_t)i * D + j] = i == kbest ? 1.0 : 0.0;
        for (int round = 0; round < 2; ++round)
            for (int c = 0; c < D; ++c) {
                if (!fin[(size_t)c]) continue;
                double dot = 0.0;
                for (int i = 0; i < D; ++i) dot += A[(size_t)i * D + j] * A[(size_t)i * D + c];
                for (int i = 0; i < D; ++i) A[(size_t)i * D + j] = A[(size_t)i * D + j] - dot * A[(size_t)i * D + c];
            }
        double s2 = 0.0;
        for (int i = 0; i < D; ++i) s2 += A[(size_t)i * D + j] * A[(size_t)i * D + j];
        const double sj = std::sqrt(s2);
        if (!(sj > 1e-8)) return false;
        for (int i = 0; i < D; ++i) A[(size_t)i * D + j] = A[(size_t)i * D + j] / sj;
        fin[(size_t)j] = 1;
    }
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {   // R[i][j] = sum_k V[i][k] U[j][k]
            double acc = 0.0;
            for (int k = 0; k < D; ++k) acc += V[(size_t)i * D + k] * A[(size_t)j * D + k];
            R[(size_t)i * D + j] = (float)acc;
        }
    return true;
}

}  // namespace cvtmi

using namespace cvtmi;

extern "C" {

int cvtmi_opq_learn_rotation_dev(const float *x, int64_t n, int D, int M, int K, int outer, int niter, uint64_t seed, float *R, float *books,
                                 void *stream)
{
    if (!x || !R || !books || n < 1 || M < 1 || M > 16 || D < 1 || D % M != 0 || K < 1 || K > 256 || outer < 0)
        return fail(CVTMI_EINVAL, "cvtmi_opq_learn_rotation: bad arguments");
    if (D % 32 != 0 || D > 128) return fail(CVTMI_EUNSUPPORTED, "cvtmi_opq_learn_rotation: D=%d (the dense rotation is built for 32, 64, 96, 128)", D);
    if (n < K) return fail(CVTMI_EINVAL, "cvtmi_opq_learn_rotation: fewer rows (%lld) than centroids (%d)", (long long)n, K);
    hipStream_t st = (hipStream_t)stream;
    const int step = D / M;
    const int64_t nblocks = (n + XTY_BLOCK - 1) / XTY_BLOCK;
    Tmp xr, y, assign, part, Cd;
    CVTMI_TRY(xr.alloc((size_t)n * D * sizeof(float)));
    CVTMI_TRY(y.alloc((size_t)n * D * sizeof(float)));
    CVTMI_TRY(assign.alloc((size_t)n * M * sizeof(int32_t)));
    CVTMI_TRY(part.alloc((size_t)nblocks * D * D * sizeof(double)));
    CVTMI_TRY(Cd.alloc((size_t)D * D * sizeof(double)));
    std::vector<float> Rh((size_t)D * D, 0.0f);
    for (int i = 0; i < D; ++i) Rh[(size_t)i * D + i] = 1.0f;
    std::vector<double> Ch((size_t)D * D);
    for (int t = 0; t <= outer; ++t) {
        CVTMI_HIP(hipMemcpyAsync(R, Rh.data(), Rh.size() * sizeof(float), hipMemcpyHostToDevice, st));
        CVTMI_HIP(stream_wait(st));   // (Rh is reused below)
        CVTMI_TRY(launch_rotate_gemm(R, D, x, n, xr.as<float>(), st));
        for (int m = 0; m < M; ++m)
            CVTMI_TRY(cvtmi_kmeans_dev(xr.as<float>() + m * step, D, n, step, K, niter, seed, books + (size_t)m * K * step,
                                       assign.as<int32_t>() + (size_t)m * n, nullptr, stream));
        if (t == outer) break;
        const unsigned rb = (unsigned)std::min<int64_t>((n * D + kBlock - 1) / kBlock, 65535);
        hipLaunchKernelGGL(opq_reconstruct_kernel, dim3(rb), dim3(kBlock), 0, st, books, assign.as<int32_t>(), n, D, M, K, y.as<float>());
        switch (D / 16) {
        case 2: hipLaunchKernelGGL((opq_xty_kernel<2>), dim3((unsigned)nblocks), dim3(256), 0, st, x, y.as<float>(), n, part.as<double>()); break;
        case 4: hipLaunchKernelGGL((opq_xty_kernel<4>), dim3((unsigned)nblocks), dim3(256), 0, st, x, y.as<float>(), n, part.as<double>()); break;
        case 6: hipLaunchKernelGGL((opq_xty_kernel<6>), dim3((unsigned)nblocks), dim3(256), 0, st, x, y.as<float>(), n, part.as<double>()); break;
        default: hipLaunchKernelGGL((opq_xty_kernel<8>), dim3((unsigned)nblocks), dim3(256), 0, st, x, y.as<float>(), n, part.as<double>()); break;
        }
        hipLaunchKernelGGL(opq_xty_reduce_kernel, dim3((unsigned)((D * D + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, part.as<double>(), nblocks, D * D,
                           Cd.as<double>());
        CVTMI_HIP(hipGetLastError());
        CVTMI_HIP(hipMemcpyAsync(Ch.data(), Cd.p, Ch.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        CVTMI_HIP(stream_wait(st));
        (void)procrustes_host(Ch.data(), D, Rh.data());   // a degenerate step keeps R
    }
    CVTMI_HIP(stream_wait(st));   // the temporaries die with this frame
    return CVTMI_OK;
}

int cvtmi_opq_learn_rotation(const float *x, int64_t n, int D, int M, int K, int outer, int niter, uint64_t seed, float *R, float *books)
{
    if (!x || !R || !books || n < 1 || D < 1 || K < 1) return fail(CVTMI_EINVAL, "cvtmi_opq_learn_rotation: bad arguments");
    Tmp dx, dR, db;
    CVTMI_TRY(dx.upload(x, (size_t)n * D * sizeof(float)));
    CVTMI_TRY(dR.alloc((size_t)D * D * sizeof(float)));
    CVTMI_TRY(db.alloc((size_t)K * D * sizeof(float)));
    CVTMI_TRY(cvtmi_opq_learn_rotation_dev(dx.as<float>(), n, D, M, K, outer, niter, seed, dR.as<float>(), db.as<float>(), nullptr));
    CVTMI_HIP(hipMemcpy(R, dR.p, (size_t)D * D * sizeof(float), hipMemcpyDeviceToHost));
    CVTMI_HIP(hipMemcpy(books, db.p, (size_t)K * D * sizeof(float), hipMemcpyDeviceToHost));
    return CVTMI_OK;
}

}  // extern "C"
