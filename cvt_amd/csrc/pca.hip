// pca.hip -- PCA projection with the L2-normalise epilogue.
//
// Reference: cvtk::PCAUtils::reduceDim (pca_train_project/pca_online/pca_utils.cc:25-35) and
// PCAModel::reductDimension (pca_train_project/project/pca_dimension.h:47-58):
//     y = cv::PCA::project(x) = (x - mean) * vectors^T      (fp32 subtract, then a gemm)
//     y /= float(max(1e-12, sqrt(y . y)))                   per row
// The models in the tree are 1024 -> 128 (model/pca_1024_128_300w_googlenet.yml) and 2048 -> 256.
//
//   launch_pca_project   Y[n x dout] = normalise((X[n x din] - mean) * E^T), E = vectors [dout][din] row-major
//
// This one IS GEMM-shaped (262 KFLOP and 4.5 KB per row at 1024 -> 128: 58 flop/B, the fp32 matrix roof is hit
// before the HBM one), so it runs on v_mfma_f32_32x32x2_f32.  A workgroup of 4 waves owns 128 rows x all dout
// outputs; K is walked in chunks of 32 columns: the X chunk (mean subtracted on the way in) and the E chunk
// are staged through a double-buffered LDS tile (stride 33 floats: the 32 lanes that read one k column hit 32
// banks), chunk c+1 is in flight from HBM/L2 while chunk c feeds 16 x NT MFMAs per wave.  One LDS-only barrier
// per chunk.  k ascends with the instruction order and every product is rounded once, so a projection is
// the k-ordered fmaf chain of its row, bit for bit.  The epilogue reduces y.y in fp64 across the 32 lanes that
// hold a row, takes the fp32 square root and divides (correctly rounded), as the reference does.
#include "common.h"
#include "div_rn.h"
#include "kernels.h"

namespace cvtmi {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PCA_THREADS = 256;
constexpr int PCA_ROWS = 128;        // rows per workgroup (32 per wave)
constexpr int PCA_KC = 32;           // K columns per chunk
constexpr int PCA_LD = PCA_KC + 1;   // padded chunk stride (floats)

template <int NT>
__global__ __launch_bounds__(PCA_THREADS) void pca_project_kernel(const float *__restrict__ mean, const float *__restrict__ E,
                                                                   int din, int dout, const float *__restrict__ x, int64_t n,
                                                                   int l2norm, float *__restrict__ y)
{
    constexpr int EROWS = 32 * NT;
    constexpr int BUF = (PCA_ROWS + EROWS) * PCA_LD;  // floats per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * PCA_ROWS;
    const int nch = (din + PCA_KC - 1) / PCA_KC;
    // loader: float4 number f = p*256 + tid of a [rows][32] chunk: row f/8, columns 4*(f%8) -- the column offset is
    // the same for every p, so one mean float4 per chunk and thread
    const int c4 = (tid & 7) * 4, r0 = tid >> 3;
    float4 px[4], pe[NT], pm;  // raw loads of the chunk in flight; the arithmetic on them waits until stash()
    float keep = 1.0f;
    auto fetch = [&](int c) {
        const int k = c * PCA_KC + c4;
        const bool live = k < din;  // din % 4 == 0: a float4 is inside or outside as a whole
        const int kc = live ? k : 0;  // loads are unconditional (clamped address); dead values are zeroed below
        keep = live ? 1.0f : 0.0f;
        pm = *reinterpret_cast<const float4 *>(mean + kc);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int64_t row = row0 + p * 32 + r0;
            row = row < n ? row : n - 1;  // clamped: tail rows are never stored
            px[p] = *reinterpret_cast<const float4 *>(x + row * din + kc);
        }
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int j = p * 32 + r0;
            pe[p] = *reinterpret_cast<const float4 *>(E + (int64_t)(j < dout ? j : dout - 1) * din + kc);
        }
    };
    auto stash = [&](float *buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float *d = buf + (p * 32 + r0) * PCA_LD + c4;
            d[0] = px[p].x - pm.x; d[1] = px[p].y - pm.y; d[2] = px[p].z - pm.z; d[3] = px[p].w - pm.w;
        }
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            float *d = buf + (PCA_ROWS + p * 32 + r0) * PCA_LD + c4;
            const float w = (p * 32 + r0) < dout ? keep : 0.0f;  // rows past dout and columns past din enter as exact zeros
            d[0] = pe[p].x * w; d[1] = pe[p].y * w; d[2] = pe[p].z * w; d[3] = pe[p].w * w;
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    fetch(0);
    for (int c = 0; c < nch; ++c) {
        float *buf = smem + (c & 1) * BUF;
        stash(buf);                    // this stage was last read two chunks ago: every wave is past that barrier
        if (c + 1 < nch) fetch(c + 1); // lands while the MFMAs below run
        lds_barrier();
        // A[i][k] = X[wave*32 + i][k0 + k] (lane: i = lane & 31, k = lane >> 5);  B[k][j] = E[j][k0 + k]
        const float *xa = buf + (wave * 32 + li) * PCA_LD + lk;
        const float *eb = buf + (PCA_ROWS + li) * PCA_LD + lk;
#pragma unroll
        for (int kk = 0; kk < PCA_KC; kk += 2) {
            const float a = xa[kk];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, eb[t * 32 * PCA_LD + kk], acc[t], 0, 0, 0);
        }
    }
    // C/D layout of 32x32: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5)
    DivBy den[16];  // 16 rows per lane, NT quotients each: one reciprocal per row
    if (l2norm) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) s += (double)acc[t][e] * (double)acc[t][e];  // pad columns hold exact zeros
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) s += __shfl_xor(s, o, 64);  // stays inside the 32-lane half that owns the row
            // sqrtf, correctly rounded: the fp64 root of an fp32 value rounds to it (__fsqrt_rn is the 1-ulp native one)
            const float nrm = (float)__dsqrt_rn((double)(float)s);
            den[e] = div_by((float)fmax(1e-12, (double)nrm));
        }
    }
    const int64_t wrow0 = row0 + wave * 32;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = t * 32 + li;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t r = wrow0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
            if (r < n && col < dout) y[r * dout + col] = l2norm ? div_rn(acc[t][e], den[e]) : acc[t][e];
        }
    }
}

int launch_pca_project(const float *mean, const float *E, int din, int dout, const float *x, int64_t n, int l2norm, float *y,
                       hipStream_t st)
{
    if (din < 4 || din % 4 != 0) return fail(CVTMI_EUNSUPPORTED, "pca_project: input dimension %d (needs a multiple of 4)", din);
    if (dout < 1 || dout > 256) return fail(CVTMI_EUNSUPPORTED, "pca_project: output dimension %d (built for 1..256)", dout);
    if ((((uintptr_t)x) | ((uintptr_t)E) | ((uintptr_t)mean)) & 15) return fail(CVTMI_EINVAL, "pca_project: pointers must be 16-byte aligned");
    if (n == 0) return CVTMI_OK;
    const int nt = (dout + 31) / 32;
    const int64_t blocks = (n + PCA_ROWS - 1) / PCA_ROWS;
    if (blocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "pca_project: %lld rows in one call", (long long)n);
    const size_t lds = (size_t)2 * (PCA_ROWS + 32 * nt) * PCA_LD * sizeof(float);
#define CVTMI_PCA_CASE(T)                                                                                                   \
    case T:                                                                                                                 \
        CVTMI_HIP(hipFuncSetAttribute((const void *)pca_project_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((pca_project_kernel<T>), dim3((unsigned)blocks), dim3(PCA_THREADS), lds, st, mean, E, din, dout, x, n, \
                           l2norm, y);                                                                                      \
        break;
    switch (nt) {
        CVTMI_PCA_CASE(1) CVTMI_PCA_CASE(2) CVTMI_PCA_CASE(3) CVTMI_PCA_CASE(4)
        CVTMI_PCA_CASE(5) CVTMI_PCA_CASE(6) CVTMI_PCA_CASE(7) CVTMI_PCA_CASE(8)
        default: return fail(CVTMI_EUNSUPPORTED, "pca_project: dout=%d", dout);
    }
#undef CVTMI_PCA_CASE
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
