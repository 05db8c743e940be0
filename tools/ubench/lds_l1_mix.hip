// lds_l1_mix.hip -- can the vector L1 serve some of the ADC scan's table look-ups beside the LDS?
// Each lane holds 16 random code bytes per "row"; per row it does NL 16-byte look-ups in an LDS table (conflict-free skew, as
// adc_scan16q) and NG 16-byte look-ups in a 4 KB-per-sub-quantiser table in global memory (L1-resident after first touch).
// Prints cycles per row-iteration per CU for (NL, NG) mixes.   hipcc --offload-arch=gfx950 -O3 lds_l1_mix.hip -o lds_l1_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int NL, int NG>
__global__ __launch_bounds__(1024) void mix_kernel(const uint4 *__restrict__ gtab, const uint4 *__restrict__ codes, int iters, u32 *out)
{
    extern __shared__ uint4 tab[];   // [256 codes][16 slots]
    for (int i = threadIdx.x; i < 4096; i += 1024) tab[i] = gtab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint4 *gt = gtab + 4096 + (size_t)(blockIdx.x & 1) * 4096;   // this workgroup's global tables: [m][256] entries of 16 B
    uint4 c = codes[(size_t)blockIdx.x * 1024 + threadIdx.x];
    u32x4 acc = { 0, 0, 0, 0 };
    for (int it = 0; it < iters; ++it) {
        const u32 w[4] = { c.x, c.y, c.z, c.w };
#pragma unroll
        for (int t = 0; t < NL; ++t) {
            const u32 code = (w[t >> 2] >> ((t & 3) * 8)) & 255u;
            const uint4 v = tab[code * 16 + ((t + lane) & 15)];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int tt = 15 - t;
            const u32 code = (w[tt >> 2] >> ((tt & 3) * 8)) & 255u;
            const uint4 v = gt[t * 256 + code];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        // next "row": scramble the codes (cheap, keeps addresses random)
        c.x = c.x * 1664525u + 1013904223u; c.y = c.y * 22695477u + 1u; c.z ^= c.x >> 7; c.w += c.y ^ (c.z << 3);
    }
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int NL, int NG>
static void run(const uint4 *gtab, const uint4 *codes, u32 *out, int iters)
{
    const int blocks = 512;   // 2 per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)mix_kernel<NL, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((mix_kernel<NL, NG>), dim3(blocks), dim3(1024), 65536, 0, gtab, codes, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_kernel<NL, NG>), dim3(blocks), dim3(1024), 65536, 0, gtab, codes, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per CU: 2 workgroups x 16 waves x iters row-iterations (64 rows each)
    const double wave_iters_per_cu = 2.0 * 16 * iters;
    printf("NL=%2d NG=%2d: %.3f ms, %.1f ns per wave-iteration per CU (x ~2.0 GHz = %.0f cycles); LDS floor for NL reads = %d cycles\n", NL, NG, ms,
           ms * 1e6 / wave_iters_per_cu, ms * 1e6 / wave_iters_per_cu * 2.0, NL * 4);
}

int main()
{
    const int iters = 4000;
    std::vector<uint4> h(4096 + 2 * 4096);
    for (auto &v : h) { v.x = rand(); v.y = rand(); v.z = rand(); v.w = rand(); }
    std::vector<uint4> hc(512 * 1024);
    for (auto &v : hc) { v.x = rand() * 2654435761u; v.y = rand() * 40503u; v.z = rand(); v.w = rand() * 7u; }
    uint4 *gtab, *codes; u32 *out;
    hipMalloc(&gtab, h.size() * 16); hipMalloc(&codes, hc.size() * 16); hipMalloc(&out, 512 * 1024 * 4);
    hipMemcpy(gtab, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    hipMemcpy(codes, hc.data(), hc.size() * 16, hipMemcpyHostToDevice);
    run<16, 0>(gtab, codes, out, iters);
    run<15, 1>(gtab, codes, out, iters);
    run<14, 2>(gtab, codes, out, iters);
    run<13, 3>(gtab, codes, out, iters);
    run<12, 4>(gtab, codes, out, iters);
    run<8, 8>(gtab, codes, out, iters);
    run<0, 4>(gtab, codes, out, iters);
    run<0, 16>(gtab, codes, out, iters);
    return 0;
}
