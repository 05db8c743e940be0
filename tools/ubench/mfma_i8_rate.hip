// issue rate of the two gfx950 int8 matrix instructions (one wave per SIMD, 4 independent accumulators)
// hipcc --offload-arch=gfx950 -O3 -o mfma_i8_rate mfma_i8_rate.hip && ./mfma_i8_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int KIND> __global__ __launch_bounds__(256) void k(int *out, int iters)
{
    v4i a = { (int)threadIdx.x, 1, 2, 3 }, b = { 4, 5, 6, (int)blockIdx.x };
    v16i c32[4] = {};
    v4i c16[8] = {};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) c32[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c32[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) c16[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c16[t], 0, 0, 0);
        }
    }
    int s = 0;
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) s += c32[t][e];
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 4; ++e) s += c16[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    int *out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 1024;  // 256 CUs x 4 blocks x 4 waves -> 4 waves per SIMD
    for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n_inst = (double)blocks * 4 * iters * (kind == 0 ? 4 : 8);
            const double ops = n_inst * (kind == 0 ? 32.0 * 32 * 32 * 2 : 16.0 * 16 * 64 * 2);
            printf("%s: %.3f ms, %.0f TOPS, %.1f cycles/instr/SIMD at 2.4 GHz\n", kind == 0 ? "v_mfma_i32_32x32x32_i8" : "v_mfma_i32_16x16x64_i8", ms,
                   ops / ms / 1e9, ms * 1e-3 * 2.4e9 / (n_inst / 1024.0));
        }
    }
    return 0;
}
