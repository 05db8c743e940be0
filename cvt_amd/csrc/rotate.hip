// rotate.hip -- the "OPQ rotation" applied to every database and query vector.
//
// Reference: IVFOPQ::reorder (opq/src/IVFOPQ.cpp:424-439), run over each row by
// LoadSingleFeatFile (:459-461): y[i] = x[reorder_[i]] -- a permutation, i.e. the special case
// R[i][reorder_[i]] = 1 of a d x d rotation y = R x.
//
//   launch_permute      the reference's gather, verbatim (exact for every input, NaN/Inf included).
//   launch_rotate_gemm  the general rotation as an fp32 MFMA GEMM  Y[n x D] = X[n x D] * R^T.
//                       v_mfma_f32_32x32x2_f32 accumulates one product per k in ascending k with a
//                       single rounding each (== an fmaf chain), so with a 0/1 permutation matrix
//                       it reproduces the gather bit for bit on finite inputs, and with a dense R
//                       it matches a k-ordered fmaf chain (the test oracle restates that) bit for bit.
//
// MI355X mapping of the GEMM (D = 32*NT, NT <= 4): a workgroup keeps R in LDS (row stride D+1
// floats: the 32 lanes that read one k column hit 32 different banks) and loops over 128-row
// slabs; each wave owns 32 rows x D outputs = NT accumulators of 32x32 (16 VGPRs each), stages its
// 32 x D slab of X through LDS with coalesced 16-byte loads, and issues NT MFMAs per k-pair.
#include "kernels.h"

namespace cvtmi {

__global__ __launch_bounds__(kBlock) void permute_kernel(const int32_t *__restrict__ perm, int D,
                                                         const float *__restrict__ x, int64_t total,
                                                         float *__restrict__ y)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t r = e / D;
        const int i = (int)(e - r * D);
        y[e] = x[r * D + perm[i]];
    }
}

int launch_permute(const int32_t *perm, int D, const float *x, int64_t n, float *y, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    const int64_t total = n * D;
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(permute_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, perm, D, x, total, y);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NT>
__global__ __launch_bounds__(kBlock, 1) void rotate_gemm_kernel(const float *__restrict__ R,
                                                                const float *__restrict__ x, int64_t n,
                                                                float *__restrict__ y)
{
    constexpr int D = 32 * NT;
    constexpr int LD = D + 1;  // padded leading dimension (floats)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Rs = smem;                 // [D][LD]   Rs[j][k] = R[j][k]
    float *Xs = smem + D * LD;        // 4 waves x [32][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < D * D; i += kBlock) {
        const int j = i / D, k = i - j * D;
        Rs[j * LD + k] = R[i];
    }
    __syncthreads();
    float *Xw = Xs + wave * 32 * LD;
    const int64_t n_slabs = (n + 127) / 128;
    for (int64_t slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
        const int64_t row0 = slab * 128 + wave * 32;
        // stage this wave's 32 x D slab (coalesced float4 loads; ragged tail rows read as zero)
        for (int i = lane; i < 32 * (D / 4); i += 64) {
            const int r = i / (D / 4), c4 = i - r * (D / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < n) v = reinterpret_cast<const float4 *>(x + (row0 + r) * D)[c4];
            float *dst = Xw + r * LD + c4 * 4;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        // wave-private LDS slab: make the stores visible to the wave's own reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
        const int li = lane & 31, lk = lane >> 5;
        // A[i][k] = X[row0+i][k0+k]  (lane: i = lane&31, k = lane>>5)
        // B[k][j] = R^T[k0+k][j0+j] = R[j0+j][k0+k]  (lane: k = lane>>5, j = lane&31)
#pragma unroll 4
        for (int k0 = 0; k0 < D; k0 += 2) {
            const float a = Xw[li * LD + k0 + lk];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float b = Rs[(t * 32 + li) * LD + k0 + lk];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        // C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (row0 + r < n) y[(row0 + r) * D + t * 32 + li] = acc[t][e];
            }
        }
        __builtin_amdgcn_wave_barrier();  // all lanes done reading Xw before the next slab overwrites it
    }
}

int launch_rotate_gemm(const float *R, int D, const float *x, int64_t n, float *y, hipStream_t st)
{
    if (n <= 0) return CVTMI_OK;
    if (D % 32 != 0 || D < 32 || D > 128)
        return fail(CVTMI_EUNSUPPORTED, "rotate_gemm: D=%d (built for 32, 64, 96, 128)", D);
    const int NT = D / 32;
    const size_t lds = (size_t)(D + 4 * 32) * (D + 1) * sizeof(float);
    const int64_t n_slabs = (n + 127) / 128;
    int64_t blocks = n_slabs < 256 ? n_slabs : 256;  // one persistent workgroup per CU (LDS-bound occupancy)
#define CVTMI_ROT(T)                                                                                              \
    case T:                                                                                                       \
        CVTMI_HIP(hipFuncSetAttribute((const void *)rotate_gemm_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds));                                                                 \
        hipLaunchKernelGGL((rotate_gemm_kernel<T>), dim3((unsigned)blocks), dim3(kBlock), lds, st, R, x, n, y);   \
        break;
    switch (NT) {
        CVTMI_ROT(1) CVTMI_ROT(2) CVTMI_ROT(3) CVTMI_ROT(4)
        default: return fail(CVTMI_EUNSUPPORTED, "rotate_gemm: D=%d", D);
    }
#undef CVTMI_ROT
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
