// sq8.hip -- int8 scalar quantisation kernels (cvtk::quant::Int8Quan + sq_train).
//
// Reference arithmetic (scalar_quantization/scalar_quantization/int8_quan.cc):
//   L2NormalizeVector :46-56   accum(double) += float(v*v) sequentially; sqrt in double;
//                              den = float(max(1e-12, norm)); v = v / den (fp32 divide)
//   Int8Encode        :79-92   xi = (x - vmin)/vdiff (0 when vdiff == 0); clamp to [0,1];
//                              byte = (int)(255 * xi)
//   Int8Decode(string):126-130 x = vmin + vdiff * (b + 0.5) / 255.0, evaluated in double
// sq_train.cpp:84-103: rows L2-normalised, then per-dimension min and max - min (faiss
// ScalarQuantizer default RS_minmax; that library is not vendored: parity unpinned).
//
// All three are streaming, HBM-bound kernels (2.5 KB moved per 512-d row on encode).  The only
// serial part is the double-precision norm, whose addition order is part of bit-exactness: rows are
// staged through LDS with coalesced loads and one lane per row folds them in index order.
#include "kernels.h"

namespace cvtmi {

constexpr int NORM_ROWS = 64;   // rows per workgroup
constexpr int NORM_COLS = 256;  // columns staged per step

__global__ __launch_bounds__(kBlock) void sq8_rownorm_kernel(const float *__restrict__ x, int64_t n, int d,
                                                             float *__restrict__ den)
{
    __shared__ float tile[NORM_ROWS][NORM_COLS + 1];
    const int64_t row0 = (int64_t)blockIdx.x * NORM_ROWS;
    const int tid = threadIdx.x;
    double accum = 0.0;
    for (int c0 = 0; c0 < d; c0 += NORM_COLS) {
        const int cl = (d - c0) < NORM_COLS ? (d - c0) : NORM_COLS;
        __syncthreads();
        for (int i = tid; i < NORM_ROWS * NORM_COLS; i += kBlock) {
            const int r = i / NORM_COLS, c = i - r * NORM_COLS;
            float v = 0.0f;
            if (c < cl && row0 + r < n) v = x[(row0 + r) * d + c0 + c];
            tile[r][c] = v;
        }
        __syncthreads();
        if (tid < NORM_ROWS) {
            for (int c = 0; c < cl; ++c) {
                const float v = tile[tid][c];
                accum += (double)__fmul_rn(v, v);
            }
        }
    }
    if (tid < NORM_ROWS && row0 + tid < n) {
        const double nrm = __dsqrt_rn(accum);
        den[row0 + tid] = (float)(nrm > 1e-12 ? nrm : 1e-12);
    }
}

int launch_sq8_rownorm(const float *x, int64_t n, int d, float *den, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    const int64_t blocks = (n + NORM_ROWS - 1) / NORM_ROWS;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "sq8: n too large");
    hipLaunchKernelGGL(sq8_rownorm_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, n, d, den);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

__global__ __launch_bounds__(kBlock) void sq8_encode_kernel(const float *__restrict__ vmin,
                                                            const float *__restrict__ vdiff, int d, float *x,
                                                            int64_t total, const float *__restrict__ den,
                                                            int write_back, uint8_t *__restrict__ codes)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t r = e / d;
        const int i = (int)(e - r * d);
        float v = x[e];
        if (den) {
            v = __fdiv_rn(v, den[r]);
            if (write_back) x[e] = v;
        }
        float xi = 0.0f;
        const float df = vdiff[i];
        if (df != 0.0f) xi = __fdiv_rn(__fsub_rn(v, vmin[i]), df);
        if (xi < 0.0f) xi = 0.0f;
        if (xi > 1.0f) xi = 1.0f;
        codes[e] = (uint8_t)(int)__fmul_rn(255.0f, xi);
    }
}

int launch_sq8_encode(const float *vmin, const float *vdiff, int d, float *x, int64_t n, const float *den,
                      int write_back, uint8_t *codes, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    const int64_t total = n * d;
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(sq8_encode_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, vmin, vdiff, d, x, total, den,
                       write_back, codes);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

__global__ __launch_bounds__(kBlock) void sq8_decode_kernel(const float *__restrict__ vmin,
                                                            const float *__restrict__ vdiff, int d,
                                                            const uint8_t *__restrict__ codes, int64_t total,
                                                            float *__restrict__ x)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int i = (int)(e % d);
        const double t = __ddiv_rn(__dmul_rn((double)vdiff[i], __dadd_rn((double)codes[e], 0.5)), 255.0);
        x[e] = (float)__dadd_rn((double)vmin[i], t);
    }
}

int launch_sq8_decode(const float *vmin, const float *vdiff, int d, const uint8_t *codes, int64_t n, float *x,
                      hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    const int64_t total = n * d;
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(sq8_decode_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, vmin, vdiff, d, codes, total, x);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// ---- train: per-dimension min / max over (normalised) rows; min and max are order-independent ----
__global__ void sq8_train_init_kernel(uint32_t *kmin, uint32_t *kmax, int d)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < d) {
        kmin[i] = f32_key(__uint_as_float(0x7f800000u));  // +inf
        kmax[i] = f32_key(__uint_as_float(0xff800000u));  // -inf
    }
}

constexpr int TRAIN_ROWS = 512;  // rows per workgroup

__global__ __launch_bounds__(kBlock) void sq8_train_kernel(const float *__restrict__ x, int64_t n, int d,
                                                           const float *__restrict__ den, uint32_t *kmin,
                                                           uint32_t *kmax)
{
    const int64_t r0 = (int64_t)blockIdx.x * TRAIN_ROWS;
    int64_t r1 = r0 + TRAIN_ROWS;
    r1 = r1 < n ? r1 : n;
    for (int c = threadIdx.x; c < d; c += kBlock) {  // lanes walk adjacent columns: coalesced rows
        float lo = __uint_as_float(0x7f800000u), hi = __uint_as_float(0xff800000u);
        for (int64_t r = r0; r < r1; ++r) {
            float v = x[r * d + c];
            if (den) v = __fdiv_rn(v, den[r]);
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
        atomicMin(&kmin[c], f32_key(lo));
        atomicMax(&kmax[c], f32_key(hi));
    }
}

__global__ void sq8_train_finish_kernel(const uint32_t *kmin, const uint32_t *kmax, int d, float *vmin, float *vdiff)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < d) {
        const float lo = key_f32(kmin[i]), hi = key_f32(kmax[i]);
        vmin[i] = lo;
        vdiff[i] = __fsub_rn(hi, lo);
    }
}

int launch_sq8_train(const float *x, int64_t n, int d, const float *den, uint32_t *kmin, uint32_t *kmax, float *vmin,
                     float *vdiff, hipStream_t st)
{
    const unsigned db = (unsigned)((d + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(sq8_train_init_kernel, dim3(db), dim3(kBlock), 0, st, kmin, kmax, d);
    if (n > 0) {
        const int64_t blocks = (n + TRAIN_ROWS - 1) / TRAIN_ROWS;
        if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "sq8_train: n too large");
        hipLaunchKernelGGL(sq8_train_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, n, d, den, kmin, kmax);
    }
    hipLaunchKernelGGL(sq8_train_finish_kernel, dim3(db), dim3(kBlock), 0, st, kmin, kmax, d, vmin, vdiff);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
