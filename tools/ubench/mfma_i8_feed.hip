// mfma_i8_feed.hip -- what keeps v_mfma_i32_32x32x32_i8 from its issue rate when the operands are real?
// Variants (all: KS = 16 K steps per "tile", CH chains = query sets sharing each B operand, WPS waves per SIMD):
//   feed 0: A and B constant registers                      (pure issue rate at this occupancy / chain count)
//   feed 1: A = CH x 16 distinct register operands, B constant
//   feed 2: A distinct, B read from LDS (ds_read_b128 per K step, PD steps ahead), no barrier
//   feed 3: feed 2 + one s_barrier per tile
// Each variant runs twice: operands all ones, then random bytes (same instruction stream, different switching power).
// hipcc --offload-arch=gfx950 -O3 -o mfma_i8_feed mfma_i8_feed.hip && ./mfma_i8_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int CH, int FEED, int WPS>
__global__ __launch_bounds__(256 * WPS) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void k(const v4i *src, int *out, int tiles)
{
    constexpr int KS = 16, PD = 4;
    __shared__ v4i ring[4][KS * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4 * KS * 64; i += 256 * WPS) (&ring[0][0])[i] = src[i & 1023];
    v4i a[CH][KS];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int s = 0; s < KS; ++s) a[c][s] = src[(c * KS + s) * 64 + lane];
    __syncthreads();
    v16i acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = e;
    const v4i bconst = src[lane + 7];
    for (int t = 0; t < tiles; ++t) {
        if (FEED == 3) __builtin_amdgcn_s_barrier();
        const v4i *pb = &ring[t & 3][lane];
        v4i bv[KS];
        if (FEED >= 2) {
#pragma unroll
            for (int s = 0; s < PD; ++s) bv[s] = pb[s * 64];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (FEED >= 2 && s + PD < KS) bv[s + PD] = pb[(s + PD) * 64];
#pragma unroll
            for (int c = 0; c < CH; ++c)
                acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(FEED >= 1 ? a[c][s] : a[0][0], FEED >= 2 ? bv[s] : bconst, acc[c], 0, 0, 0);
        }
        if (FEED >= 2) {
            __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + PD < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CH, 0);
            }
        }
    }
    int sum = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[c][e];
    out[blockIdx.x * 256 * WPS + threadIdx.x] = sum;
}

template <int CH, int FEED, int WPS>
static void run(const v4i *src, int *out)
{
    const int tiles = getenv("TILES") ? atoi(getenv("TILES")) : 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CH, FEED, WPS>), dim3(blocks), dim3(256 * WPS), 0, 0, src, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CH, FEED, WPS>), dim3(blocks), dim3(256 * WPS), 0, 0, src, out, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_inst_per_simd = (double)tiles * 16 * CH * WPS;
    const double ops = n_inst_per_simd * 1024.0 * 32.0 * 32 * 32 * 2;
    printf("chains %d feed %d waves/SIMD %d: %.3f ms, %.0f TOPS, %.1f ns per matrix instruction per SIMD (32 cycles = %.1f ns at 2.0 GHz)\n", CH, FEED, WPS, ms,
           ops / ms / 1e9, ms * 1e6 / n_inst_per_simd, 16.0);
}

int main()
{
    v4i *src; int *out;
    hipMalloc(&src, 1 << 20); hipMalloc(&out, 256 * 512 * 4);
    for (int mode = 0; mode < 2; ++mode) {
    if (mode == 0) { hipMemset(src, 1, 1 << 20); printf("-- operands: every byte 1 (few bits toggle)\n"); }
    else {
        static unsigned char h[1 << 20];
        unsigned x = 12345;
        for (auto &b : h) { x = x * 1664525u + 1013904223u; b = (unsigned char)(x >> 24); }
        hipMemcpy(src, h, 1 << 20, hipMemcpyHostToDevice);
        printf("-- operands: random bytes\n");
    }
    run<4, 0, 1>(src, out);
    run<3, 0, 1>(src, out);
    run<2, 0, 1>(src, out);
    run<2, 0, 2>(src, out);
    run<3, 1, 1>(src, out);
    run<3, 2, 1>(src, out);
    run<3, 3, 1>(src, out);
    run<2, 1, 2>(src, out);
    run<2, 2, 2>(src, out);
    run<2, 3, 2>(src, out);
    run<4, 2, 1>(src, out);
    }
    return 0;
}
