// mfma_f32_rate.hip -- sustained rate of v_mfma_f32_32x32x2_f32 (the rotation kernel's instruction) on constant and on random operands:
// how much of the 157 TF instruction-rate peak does the power budget allow?   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
template <int WPS>
__global__ __launch_bounds__(256 * WPS) void k(const float *src, float *out, int iters)
{
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = src[(i * 64 + threadIdx.x) & 65535]; b[i] = src[(i * 64 + threadIdx.x + 4096) & 65535]; }
    v16f acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + c) & 15], b[i], acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s;
}
int main()
{
    float *src, *out; hipMalloc(&src, 65536 * 4); hipMalloc(&out, 256 * 512 * 4);
    static float h[65536];
    for (int mode = 0; mode < 2; ++mode) {
        unsigned x = 777;
        for (auto &v : h) { x = x * 1664525u + 1013904223u; v = mode ? ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 1.0f; }
        hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
        for (int wps = 1; wps <= 2; ++wps) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 4000;
            if (wps == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, src, out, 10); else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, src, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (wps == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, src, out, iters); else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, src, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)iters * 64 * wps * 1024.0 * 32 * 32 * 2 * 2;
            printf("%s operands, %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s (fp32 matrix peak 157.3)\n", mode ? "random" : "constant", wps, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
