// Micro-benchmark: 16-bit fixed-point filter tables, 8 queries per ds_read_b128 (development aid).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b)
{
    us2 x = __builtin_bit_cast(us2, a), y = __builtin_bit_cast(us2, b);
    return __builtin_bit_cast(uint32_t, (us2)(x + y));
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b)
{
    s2 x = __builtin_bit_cast(s2, a), y = __builtin_bit_cast(s2, b);
    return __builtin_bit_cast(uint32_t, (s2)(x - y));
}

// QT = 8 (one b128 per code byte) or 16 (two: second at +256)
template <int NT, int R, int QT, int MINW, int ADD3 = 0, int NLK = 16>
__global__ __launch_bounds__(NT, MINW) void scan_kernel(const uint4 *__restrict__ rows, int64_t n_rows, const uint32_t *__restrict__ lut_g,
                                                        uint32_t thr, uint32_t *out)
{
    constexpr int ROWB = 16 * QT * 2;  // bytes per code value: 16 sub-quantisers x QT x u16
    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * ROWB / 4];
    __shared__ int cnt;
    for (int i = threadIdx.x; i < 256 * ROWB / 4; i += NT) lut[i] = lut_g[i % (256 * 16 * 4)];
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int tid = threadIdx.x;
    const uint32_t c = tid & 15;
    const uint32_t cr8 = (c & 3) * 8, cq = c >> 2;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);
    const uint32_t n_local = (uint32_t)n_rows, last = n_local - 1;
    const char *rows_b = reinterpret_cast<const char *>(rows);
    auto load_row = [&](uint32_t lrow) -> uint4 {
        const uint32_t cl = lrow < last ? lrow : last;
        return *reinterpret_cast<const uint4 *>(rows_b + (size_t)(cl * 16u));
    };
    uint32_t chk = 0;
    uint4 cur[R], nxt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cur[r] = load_row(r * NT + tid);
    uint32_t thr_pk[QT / 2];
#pragma unroll
    for (int i = 0; i < QT / 2; ++i) thr_pk[i] = thr | (thr << 16);
    for (uint32_t base = 0; base < n_local; base += NT * R) {
#pragma unroll
        for (int r = 0; r < R; ++r) nxt[r] = load_row(base + NT * R + r * NT + tid);
        bool want = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t d0 = __builtin_amdgcn_alignbit(cur[r].y, cur[r].x, cr8);
            uint32_t d1 = __builtin_amdgcn_alignbit(cur[r].z, cur[r].y, cr8);
            uint32_t d2 = __builtin_amdgcn_alignbit(cur[r].w, cur[r].z, cr8);
            uint32_t d3 = __builtin_amdgcn_alignbit(cur[r].x, cur[r].w, cr8);
            const bool b0 = cq & 1;
            const uint32_t e0 = b0 ? d1 : d0, e1 = b0 ? d2 : d1, e2 = b0 ? d3 : d2, e3 = b0 ? d0 : d3;
            const bool b1 = cq & 2;
            d0 = b1 ? e2 : e0; d1 = b1 ? e3 : e1; d2 = b1 ? e0 : e2; d3 = b1 ? e1 : e3;
            const uint32_t rot[4] = { d0, d1, d2, d3 };
            uint32_t acc[QT / 2];
            if (ADD3 && QT == 8) {
                // packed 15-bit sums never carry across the 16-bit halves: plain 32-bit adds, two look-ups per v_add3_u32
                uint4 v[NLK];
#pragma unroll
                for (int t = 0; t < NLK; ++t) {
                    const uint32_t sel = 0x0c0c0000u | ((4u + (t & 3)) << 8) | (uint32_t)(t & 3);
                    const uint32_t addr = __builtin_amdgcn_perm(rot[t >> 2], moffp[t >> 2], sel);
                    v[t] = *reinterpret_cast<const uint4 *>(lut_b + addr);
                }
                acc[0] = 0; acc[1] = 0; acc[2] = 0; acc[3] = 0;
#pragma unroll
                for (int t = 0; t < NLK; t += 2) {
                    acc[0] = acc[0] + v[t].x + v[t + 1].x; acc[1] = acc[1] + v[t].y + v[t + 1].y;
                    acc[2] = acc[2] + v[t].z + v[t + 1].z; acc[3] = acc[3] + v[t].w + v[t + 1].w;
                }
            } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint32_t sel = 0x0c0c0000u | ((4u + (t & 3)) << 8) | (uint32_t)(t & 3);
                uint32_t addr = __builtin_amdgcn_perm(rot[t >> 2], moffp[t >> 2], sel);  // code*256 + m*16
                if (QT == 16) addr = (addr & 0xffu) | ((addr & 0xff00u) << 1);            // code*512 + m*16
                const uint4 v = *reinterpret_cast<const uint4 *>(lut_b + addr);
                if (t == 0) { acc[0] = v.x; acc[1] = v.y; acc[2] = v.z; acc[3] = v.w; }
                else { acc[0] = pk_add_u16(acc[0], v.x); acc[1] = pk_add_u16(acc[1], v.y); acc[2] = pk_add_u16(acc[2], v.z); acc[3] = pk_add_u16(acc[3], v.w); }
                if (QT == 16) {
                    const uint4 u = *reinterpret_cast<const uint4 *>(lut_b + addr + 256);
                    if (t == 0) { acc[4] = u.x; acc[5] = u.y; acc[6] = u.z; acc[7] = u.w; }
                    else { acc[4] = pk_add_u16(acc[4], u.x); acc[5] = pk_add_u16(acc[5], u.y); acc[6] = pk_add_u16(acc[6], u.z); acc[7] = pk_add_u16(acc[7], u.w); }
                }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            // any (sum < thr) over the QT packed 15-bit sums: sign bits of (sum - thr)
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < QT / 2; ++i) m |= pk_sub_i16(acc[i], thr_pk[i]);
            if (__ballot((m & 0x80008000u) != 0)) {
                if (m & 0x80008000u) {
                    const int pos = atomicAdd(&cnt, 1);
                    if (pos >= 256) want = true;
                    chk += pos;
                }
            }
        }
        if (want) cnt = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = nxt[r];
    }
    out[(int64_t)blockIdx.x * NT + tid] = chk + cnt;
}

template <int NT, int R, int QT, int MINW, int ADD3 = 0, int NLK = 16>
static void run(const char *name, const uint4 *rows, int64_t n_rows, int nq, const uint32_t *lut, uint32_t *out)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int groups = nq / QT;
    for (int it = 0; it < 2; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((scan_kernel<NT, R, QT, MINW, ADD3, NLK>), dim3(groups), dim3(NT), 0, 0, rows, n_rows, lut, 1u, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s NT=%4d R=%d QT=%2d minw=%d groups=%d: %8.3f ms  %6.2f T q-lookups/s  (%.0f K QPS at 1M x 10K)\n", name, NT, R, QT, MINW, groups, ms,
           (double)nq * n_rows * 16 / ms / 1e9, nq / ms);
    fflush(stdout);
}

int main()
{
    const int64_t n = 1000000;
    std::vector<uint32_t> h((size_t)n * 4);
    uint64_t s = 88172645463325252ull;
    for (auto &w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
    std::vector<uint32_t> hl(256 * 16 * 4);
    for (auto &f : hl) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; f = (uint32_t)((s >> 40) & 0x03ff03ff) + 0x00100010; }
    uint4 *rows; uint32_t *lut; uint32_t *out;
    CK(hipMalloc(&rows, n * 16)); CK(hipMalloc(&lut, hl.size() * 4)); CK(hipMalloc(&out, (size_t)4096 * 1024 * 4));
    CK(hipMemcpy(rows, h.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(lut, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    const int nq = 10240;
    run<1024, 1, 8, 8, 1, 16>("add3 16 look-ups", rows, n, nq, lut, out);
    run<1024, 1, 8, 8, 1, 14>("add3 14 look-ups", rows, n, nq, lut, out);
    run<1024, 1, 8, 8, 1, 12>("add3 12 look-ups", rows, n, nq, lut, out);
    run<1024, 1, 8, 8, 1, 8>("add3 8 look-ups", rows, n, nq, lut, out);
    run<1024, 1, 8, 8, 1, 4>("add3 4 look-ups", rows, n, nq, lut, out);
    run<1024, 1, 8, 8, 1, 16>("add3 16 look-ups", rows, n, nq, lut, out);
    if (getenv("ONLY_NLK")) return 0;
    run<512, 2, 8, 4>("u16 QT=8", rows, n, nq, lut, out);
    run<512, 1, 8, 4>("u16 QT=8", rows, n, nq, lut, out);
    run<512, 4, 8, 4>("u16 QT=8", rows, n, nq, lut, out);
    run<1024, 1, 8, 4>("u16 QT=8", rows, n, nq, lut, out);
    run<1024, 2, 8, 4>("u16 QT=8", rows, n, nq, lut, out);
    run<256, 2, 8, 2>("u16 QT=8", rows, n, nq, lut, out);
    run<512, 2, 8, 4, 1>("u16 QT=8 add3", rows, n, nq, lut, out);
    run<512, 4, 8, 4, 1>("u16 QT=8 add3", rows, n, nq, lut, out);
    run<1024, 1, 8, 4, 1>("u16 QT=8 add3", rows, n, nq, lut, out);
    run<1024, 2, 8, 4, 1>("u16 QT=8 add3", rows, n, nq, lut, out);
    run<512, 2, 16, 2>("u16 QT=16 (1 WG/CU)", rows, n, nq, lut, out);
    run<1024, 1, 16, 4>("u16 QT=16 (1 WG/CU)", rows, n, nq, lut, out);
    run<1024, 2, 16, 4>("u16 QT=16 (1 WG/CU)", rows, n, nq, lut, out);
    return 0;
}
