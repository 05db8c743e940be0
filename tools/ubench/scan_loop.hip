// Micro-benchmark of the ADC-scan inner loop variants (development aid, not part of the product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scan_loop.hip -o scan_loop && ./scan_loop
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE bits: 1 = v_perm address, 2 = barrier per tile, 4 = threshold compare + (never taken) push,
//            8 = scalar adds instead of letting the compiler pack, 16 = no rotation (conflicting reads),
//           32 = software-pipeline rows (issue next row's reads before consuming)
template <int NT, int R, int MODE, int MINW>
__global__ __launch_bounds__(NT, MINW) void scan_kernel(const uint4 *__restrict__ rows, int64_t n_rows, int rows_per_block,
                                                        const float *__restrict__ lut_g, uint32_t thr, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) float lut[256 * 16 * 4];
    __shared__ int cnt;
    for (int i = threadIdx.x; i < 256 * 16 * 4; i += NT) lut[i] = lut_g[i];
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int tid = threadIdx.x;
    const uint32_t c = (MODE & 16) ? 0u : (tid & 15);
    const uint32_t cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t moff[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) moff[t] = ((t + c) & 15u) * 16u;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) moffp[w] = moff[4 * w] | (moff[4 * w + 1] << 8) | (moff[4 * w + 2] << 16) | (moff[4 * w + 3] << 24);
    const char *lut_b = reinterpret_cast<const char *>(lut);
    const int64_t row_begin = 0;  // every block = one query group scanning all rows
    const int64_t row_end = n_rows; (void)rows_per_block;
    uint32_t chk = 0;
    uint4 cur[R], nxt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row_begin + r * NT + tid;
        cur[r] = row < row_end ? rows[row] : make_uint4(0, 0, 0, 0);
    }
    for (int64_t base = row_begin; base < row_end; base += (int64_t)NT * R) {
        const int64_t nbase = base + (int64_t)NT * R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = nbase + r * NT + tid;
            nxt[r] = row < row_end ? rows[row] : make_uint4(0, 0, 0, 0);
        }
        bool want = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t d0 = cur[r].x, d1 = cur[r].y, d2 = cur[r].z, d3 = cur[r].w;
            if (!(MODE & 16)) {
                d0 = __builtin_amdgcn_alignbit(cur[r].y, cur[r].x, cr8);
                d1 = __builtin_amdgcn_alignbit(cur[r].z, cur[r].y, cr8);
                d2 = __builtin_amdgcn_alignbit(cur[r].w, cur[r].z, cr8);
                d3 = __builtin_amdgcn_alignbit(cur[r].x, cur[r].w, cr8);
                const bool b0 = cq & 1;
                const uint32_t e0 = b0 ? d1 : d0, e1 = b0 ? d2 : d1, e2 = b0 ? d3 : d2, e3 = b0 ? d0 : d3;
                const bool b1 = cq & 2;
                d0 = b1 ? e2 : e0; d1 = b1 ? e3 : e1; d2 = b1 ? e0 : e2; d3 = b1 ? e1 : e3;
            }
            const uint32_t rot[4] = { d0, d1, d2, d3 };
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                uint32_t addr;
                if (MODE & 1) {
                    // byte0 = moff (from moffp[t>>2] byte t&3), byte1 = code byte (rot[t>>2] byte t&3), bytes 2,3 = 0
                    const uint32_t sel = 0x0c0c0000u | ((4u + (t & 3)) << 8) | (uint32_t)(t & 3);
                    addr = __builtin_amdgcn_perm(rot[t >> 2], moffp[t >> 2], sel);
                } else {
                    const uint32_t j = (rot[t >> 2] >> (8 * (t & 3))) & 0xffu;
                    addr = j * 256u + moff[t];
                }
                const float4 v = *reinterpret_cast<const float4 *>(lut_b + addr);
                if (MODE & 8) {
                    a0 = __fadd_rn(a0, v.x); a1 = __fadd_rn(a1, v.y); a2 = __fadd_rn(a2, v.z); a3 = __fadd_rn(a3, v.w);
                } else {
                    a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t k0 = __float_as_uint(a0), k1 = __float_as_uint(a1), k2 = __float_as_uint(a2), k3 = __float_as_uint(a3);
            if (MODE & 4) {
                if (k0 < thr || k1 < thr || k2 < thr || k3 < thr) {
                    const int pos = atomicAdd(&cnt, 1);
                    if (pos >= 256) want = true;
                    chk += pos;
                }
            } else {
                chk ^= k0 ^ k1 ^ k2 ^ k3;
            }
        }
        if (MODE & 2) {
            if (want) cnt = 0;
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = nxt[r];
    }
    out[(int64_t)blockIdx.x * NT + tid] = chk + ((MODE & 4) ? cnt : 0);
}

template <int NT, int R, int MODE, int MINW>
static void run(const char *name, const uint4 *rows, int64_t n_rows, int groups, const float *lut, uint32_t *out)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rpb = (int)n_rows;  // every block scans all rows (= one query group)
    for (int it = 0; it < 2; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((scan_kernel<NT, R, MODE, MINW>), dim3(groups), dim3(NT), 0, 0, rows, n_rows, rpb, lut, 0x00000001u, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ql = (double)groups * 4 * n_rows * 16;
    printf("%-44s NT=%4d R=%d MODE=%2d minw=%d : %8.3f ms  %6.2f T q-lookups/s  (%.0f K QPS at 1M x 10K)\n", name, NT, R, MODE, MINW, ms,
           ql / ms / 1e9, groups * 4 / ms);
    fflush(stdout);
}

int main()
{
    const int64_t n = 1000000;
    const int groups = 2500;
    std::vector<uint32_t> h((size_t)n * 4);
    uint64_t s = 88172645463325252ull;
    for (auto &w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
    std::vector<float> hl(256 * 16 * 4);
    for (auto &f : hl) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; f = (float)((s >> 40) & 0xffff) / 65536.0f + 0.5f; }
    uint4 *rows; float *lut; uint32_t *out;
    CK(hipMalloc(&rows, n * 16)); CK(hipMalloc(&lut, hl.size() * 4)); CK(hipMalloc(&out, (size_t)groups * 1024 * 4));
    CK(hipMemcpy(rows, h.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(lut, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    //            NT   R MODE MINW
    run<256, 2, 0, 2>("base (shift/and addr, pk add)", rows, n, groups, lut, out);
    run<256, 2, 16, 2>("no rotation (bank conflicts)", rows, n, groups, lut, out);
    run<256, 2, 1, 2>("v_perm addr", rows, n, groups, lut, out);
    run<256, 2, 1 | 8, 2>("v_perm addr, scalar adds", rows, n, groups, lut, out);
    run<256, 2, 1 | 2, 2>("v_perm + barrier/tile", rows, n, groups, lut, out);
    run<256, 2, 1 | 2 | 4, 2>("v_perm + barrier + thr compare", rows, n, groups, lut, out);
    run<256, 1, 1 | 2 | 4, 2>("  same R=1", rows, n, groups, lut, out);
    run<256, 4, 1 | 2 | 4, 2>("  same R=4", rows, n, groups, lut, out);
    run<512, 1, 1 | 2 | 4, 4>("512 threads, 4 waves/SIMD", rows, n, groups, lut, out);
    run<512, 2, 1 | 2 | 4, 4>("512 threads, 4 waves/SIMD", rows, n, groups, lut, out);
    run<512, 1, 1, 4>("512 threads, no barrier no cmp", rows, n, groups, lut, out);
    run<1024, 1, 1 | 2 | 4, 4>("1024 threads (1 block/CU), 4 waves/SIMD", rows, n, groups, lut, out);
    run<1024, 2, 1 | 2 | 4, 4>("1024 threads (1 block/CU), 4 waves/SIMD", rows, n, groups, lut, out);
    run<256, 1, 1, 2>("256 R=1 no barrier", rows, n, groups, lut, out);
    run<256, 4, 1, 2>("256 R=4 no barrier", rows, n, groups, lut, out);
    return 0;
}
