// Micro-benchmark (round 6): 8-bit filter tables, SIXTEEN queries per ds_read_b128 (development aid, not product code).
//
//   u16  : today's loop (adc_scan16.h scan16q_row_sums): 15-bit entries, 8 queries per read, v_add3 on packed halves, bit-15 test
//   u8   : 4-bit entries (<= 15), 16 queries per read, v_add3 on packed bytes (16 entries sum <= 240: no carry), bit-7 test
//   mfma : 6-bit entries (<= 42), 16 queries per read; look-ups are summed three at a time on packed bytes (<= 126: no carry and
//          non-negative as int8), and each group sum is WIDENED into int32 accumulators by one v_mfma_i32_32x32x32_i8 against a
//          constant selector operand:  D[i][j] += sum_k SEL[i][k] * DATA[k][j],  SEL[i][k] = 1 iff k = (half i>>2&1 ... see sel_operand)
//          -> lane l, register r = sum of row l's entries for query r.  The thresholds ride in the accumulators' start value, the
//          test is the sign bit of the OR of the 16 registers.
// Every variant counts survivors; `mfma` also checks its sums of the first chunk against a scalar re-computation (layout proof).
//
// hipcc --offload-arch=gfx950 -O3 -o scan_loop_u8 scan_loop_u8.hip && ./scan_loop_u8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

enum { K_U16 = 0, K_U8 = 1, K_MFMA = 2, K_MFMA2 = 3 };

// A-operand of the selector: lane (i = l % 32, h = l / 32) holds SEL[i][k(h, b)], b = 0..15.  Output row i of the 32x32 product lands
// in lane (j, H) register r with i = 8 (r / 4) + 4 H + (r % 4); we want that register to be "query r of the row held by lane j + 32 H",
// i.e. output row i takes DATA row k(h = H(i), b = r(i)) with H(i) = (i >> 2) & 1, r(i) = 4 (i >> 3) + (i & 3).
__device__ __forceinline__ v4i sel_operand()
{
    const int l = threadIdx.x & 63, i = l & 31, h = l >> 5;
    const int H = (i >> 2) & 1, r = 4 * (i >> 3) + (i & 3);
    uint32_t w[4] = { 0, 0, 0, 0 };
    if (h == H) w[r >> 2] = 1u << (8 * (r & 3));
    return v4i{ (int)w[0], (int)w[1], (int)w[2], (int)w[3] };
}

template <int KIND, int NT, int MINW, int PAD_KB, bool CHECK = false, int NMF = 6>
__global__ __launch_bounds__(NT, MINW) void scan_kernel(const uint4 *__restrict__ rows, int64_t n_rows, const uint32_t *__restrict__ lut_g,
                                                        int thr, uint32_t *out, int *check)
{
    // MFMA variants: one block, the table first: its LDS address is 0, so a look-up's address is the v_perm_b32 result itself
    // (the packed-add variants keep separate arrays: at 64 registers the struct form spills)
    struct ShA { uint32_t lut[256 * 16 * 4]; uint32_t queue[16][192]; uint32_t qcnt[16]; int cnt; char pad[PAD_KB * 1024 + 16]; };
    __shared__ __attribute__((aligned(16))) char sh_raw[KIND == K_MFMA ? sizeof(ShA) : 16];
    __shared__ __attribute__((aligned(16))) uint32_t lut_s[KIND == K_MFMA ? 4 : 256 * 16 * 4];
    __shared__ uint32_t queue_s[KIND == K_MFMA ? 1 : 16][192];
    __shared__ uint32_t qcnt_s[16];
    __shared__ int cnt_s;
    __shared__ char pad_s[KIND == K_MFMA ? 16 : PAD_KB * 1024 + 16];
    ShA &sh = *reinterpret_cast<ShA *>(sh_raw);
    uint32_t *lut = KIND == K_MFMA ? sh.lut : lut_s;
    uint32_t (*queue)[192] = KIND == K_MFMA ? sh.queue : queue_s;
    uint32_t *qcnt = KIND == K_MFMA ? sh.qcnt : qcnt_s;
    int &cnt = KIND == K_MFMA ? sh.cnt : cnt_s;
    char *pad = KIND == K_MFMA ? sh.pad : pad_s;
    for (int i = threadIdx.x; i < 256 * 16 * 4; i += NT) lut[i] = lut_g[i];
    if (threadIdx.x == 0) cnt = 0;
    if (threadIdx.x < 16) qcnt[threadIdx.x] = 0;
    if (PAD_KB && thr == -12345) pad[threadIdx.x] = 1;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;
    const uint32_t c = tid & 15;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const char *lut_b = reinterpret_cast<const char *>(lut);
    const uint32_t n_chunks = (uint32_t)(n_rows / 64);
    const char *rows_b = reinterpret_cast<const char *>(rows);
    const uint32_t lane16 = lane * 16u;
    auto load_rows = [&](uint32_t chunk) -> uint4 {
        const uint32_t cc = chunk < n_chunks - 1 ? chunk : n_chunks - 1;
        return *reinterpret_cast<const uint4 *>(rows_b + (size_t)cc * 1024u + lane16);
    };
    uint32_t chk = 0;
    uint4 cur = load_rows(wave), nxt;
    const v4i sel = sel_operand();
    v16i cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = -thr;
    const uint32_t b16 = 0x80008000u - ((uint32_t)thr | ((uint32_t)thr << 16));
    const uint32_t b8 = 0x80808080u - (uint32_t)thr * 0x01010101u;
    for (uint32_t it = wave; it < n_chunks; it += NW) {
        nxt = load_rows(it + NW);
        const uint32_t rot[4] = { cur.x, cur.y, cur.z, cur.w };  // pre-rotated rows (the product's layout): byte t = code of sub-space (t + lane) & 15
        uint4 v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t sel_b = 0x0c0c0000u | ((4u + (t & 3)) << 8) | (uint32_t)(t & 3);
            const uint32_t addr = __builtin_amdgcn_perm(rot[t >> 2], moffp[t >> 2], sel_b);  // code*256 + m*16
            v[t] = *reinterpret_cast<const uint4 *>(lut_b + addr);
        }
        if constexpr (KIND == K_MFMA) __builtin_amdgcn_sched_barrier(0);
        if constexpr (KIND == K_MFMA) {
            v16i acc = cinit;
            auto q4 = [&](int t) { return v4i{ (int)v[t].x, (int)v[t].y, (int)v[t].z, (int)v[t].w }; };
            if constexpr (NMF == 6) {
#pragma unroll
                for (int g = 0; g < 5; ++g) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, q4(3 * g) + q4(3 * g + 1) + q4(3 * g + 2), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, q4(15), acc, 0, 0, 0);
            } else if constexpr (NMF == 4) {
#pragma unroll
                for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, q4(4 * g) + q4(4 * g + 1) + q4(4 * g + 2) + q4(4 * g + 3), acc, 0, 0, 0);
            } else if constexpr (NMF == 3) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, q4(0) + q4(1) + q4(2) + q4(3) + q4(4), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, q4(5) + q4(6) + q4(7) + q4(8) + q4(9), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, q4(10) + q4(11) + q4(12) + q4(13) + q4(14) + q4(15), acc, 0, 0, 0);
            } else if constexpr (NMF == 2) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, q4(0) + q4(1) + q4(2) + q4(3) + q4(4) + q4(5) + q4(6) + q4(7), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, q4(8) + q4(9) + q4(10) + q4(11) + q4(12) + q4(13) + q4(14) + q4(15), acc, 0, 0, 0);
            } else {  // 1: everything added on packed bytes, one widening (timing only: random 6-bit entries overflow their bytes)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, q4(0) + q4(1) + q4(2) + q4(3) + q4(4) + q4(5) + q4(6) + q4(7) + q4(8) + q4(9) + q4(10) + q4(11) + q4(12) + q4(13) + q4(14) + q4(15), acc, 0, 0, 0);
            }
            if (CHECK && check && blockIdx.x == 0 && it == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) check[lane * 16 + r] = acc[r] + thr;
            }
            uint32_t o = ((uint32_t)acc[0] | (uint32_t)acc[1] | (uint32_t)acc[2]);
            o = o | (uint32_t)acc[3] | (uint32_t)acc[4];
            o = o | (uint32_t)acc[5] | (uint32_t)acc[6];
            o = o | (uint32_t)acc[7] | (uint32_t)acc[8];
            o = o | (uint32_t)acc[9] | (uint32_t)acc[10];
            o = o | (uint32_t)acc[11] | (uint32_t)acc[12];
            o = o | (uint32_t)acc[13] | (uint32_t)acc[14];
            o = o | (uint32_t)acc[15];
            if (__ballot((int)o < 0)) {
                if ((int)o < 0) {
                    uint32_t m = 0;  // sign bits of the 16 registers, register 15 in bit 0
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = __builtin_amdgcn_alignbit(m, (uint32_t)acc[r], 31);
                    m &= 0xffffu;
                    while (m) {
                        const int bit = __ffs((int)m) - 1;
                        m &= m - 1;
                        const int q = 15 - bit;
                        const uint32_t pos = atomicAdd(&qcnt[q], 1u);
                        queue[q][pos % 192u] = it * 64 + lane;
                        ++chk;
                    }
                }
            }
        } else if constexpr (KIND == K_U8) {
            uint32_t s0 = b8, s1 = b8, s2 = b8, s3 = b8;
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                s0 = s0 + v[t].x + v[t + 1].x; s1 = s1 + v[t].y + v[t + 1].y;
                s2 = s2 + v[t].z + v[t + 1].z; s3 = s3 + v[t].w + v[t + 1].w;
            }
            const uint32_t sg = (~((s0 & s1) & (s2 & s3))) & 0x80808080u;
            if (__ballot(sg != 0)) {
                if (sg != 0) {
                    const uint32_t sums[4] = { s0, s1, s2, s3 };
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        uint32_t m = ~sums[w] & 0x80808080u;
                        while (m) {
                            const int bit = __ffs((int)m) - 1;
                            m &= m - 1;
                            const int q = 4 * w + (bit >> 3);
                            const uint32_t pos = atomicAdd(&qcnt[q], 1u);
                            queue[q][pos % 192u] = it * 64 + lane;
                            ++chk;
                        }
                    }
                }
            }
        } else {
            uint32_t s0 = b16, s1 = b16, s2 = b16, s3 = b16;
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                s0 = s0 + v[t].x + v[t + 1].x; s1 = s1 + v[t].y + v[t + 1].y;
                s2 = s2 + v[t].z + v[t + 1].z; s3 = s3 + v[t].w + v[t + 1].w;
            }
            const uint32_t sg = (~((s0 & s1) & (s2 & s3))) & 0x80008000u;
            if (__ballot(sg != 0)) {
                if (sg != 0) {
                    const uint32_t sums[4] = { s0, s1, s2, s3 };
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        uint32_t m = ~sums[w] & 0x80008000u;
                        while (m) {
                            const int bit = __ffs((int)m) - 1;
                            m &= m - 1;
                            const int q = 2 * w + (bit >> 4);
                            const uint32_t pos = atomicAdd(&qcnt[q], 1u);
                            queue[q][pos % 192u] = it * 64 + lane;
                            ++chk;
                        }
                    }
                }
            }
        }
        cur = nxt;
    }
    out[(int64_t)blockIdx.x * NT + tid] = chk;
}


// mfma, software-pipelined inside the wave: the 16 reads of chunk i + 1 are issued between the first matrix instruction of chunk i
// and the other five, so a wave's matrix chain runs while its next reads are in flight (the compiler alone waits for each batch).
template <int NT, int MINW, int PAD_KB>
__global__ __launch_bounds__(NT, MINW) void scan_kernel_p(const uint4 *__restrict__ rows, int64_t n_rows, const uint32_t *__restrict__ lut_g,
                                                          int thr, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) struct {
        uint32_t lut[256 * 16 * 4];
        uint32_t queue[16][192];
        uint32_t qcnt[16];
        char pad[PAD_KB * 1024 + 16];
    } sh;
    for (int i = threadIdx.x; i < 256 * 16 * 4; i += NT) sh.lut[i] = lut_g[i];
    if (threadIdx.x < 16) sh.qcnt[threadIdx.x] = 0;
    if (PAD_KB && thr == -12345) sh.pad[threadIdx.x] = 1;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;
    const uint32_t c = tid & 15;
    uint32_t moffp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        moffp[w] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) moffp[w] |= (((4 * w + b + c) & 15u) * 16u) << (8 * b);
    }
    const uint32_t n_chunks = (uint32_t)(n_rows / 64);
    const char *rows_b = reinterpret_cast<const char *>(rows);
    const uint32_t lane16 = lane * 16u;
    auto load_rows = [&](uint32_t chunk) -> uint4 {
        const uint32_t cc = chunk < n_chunks - 1 ? chunk : n_chunks - 1;
        return *reinterpret_cast<const uint4 *>(rows_b + (size_t)cc * 1024u + lane16);
    };
    uint32_t chk = 0;
    const v4i sel = sel_operand();
    v16i cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = -thr;
    v4i v[16];
    auto issue = [&](const uint4 &row) {
        const uint32_t rot[4] = { row.x, row.y, row.z, row.w };
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t sel_b = 0x0c0c0000u | ((4u + (t & 3)) << 8) | (uint32_t)(t & 3);
            const uint32_t addr = __builtin_amdgcn_perm(rot[t >> 2], moffp[t >> 2], sel_b);
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[t]) : "v"(addr));
        }
    };
    uint4 cur = load_rows(wave), nxt = load_rows(wave + NW);
    issue(cur);
    for (uint32_t it = wave; it < n_chunks; it += NW) {
        const uint4 nxt2 = load_rows(it + 2 * NW);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                       "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
        v4i s[5];
#pragma unroll
        for (int g = 0; g < 5; ++g) s[g] = v[3 * g] + v[3 * g + 1] + v[3 * g + 2];
        v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, v[15], cinit, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue(nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 5; ++g) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sel, s[g], acc, 0, 0, 0);
        uint32_t o = ((uint32_t)acc[0] | (uint32_t)acc[1] | (uint32_t)acc[2]);
        o = o | (uint32_t)acc[3] | (uint32_t)acc[4];
        o = o | (uint32_t)acc[5] | (uint32_t)acc[6];
        o = o | (uint32_t)acc[7] | (uint32_t)acc[8];
        o = o | (uint32_t)acc[9] | (uint32_t)acc[10];
        o = o | (uint32_t)acc[11] | (uint32_t)acc[12];
        o = o | (uint32_t)acc[13] | (uint32_t)acc[14];
        o = o | (uint32_t)acc[15];
        if (__ballot((int)o < 0)) {
            if ((int)o < 0) {
                uint32_t m = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) m = __builtin_amdgcn_alignbit(m, (uint32_t)acc[r], 31);
                m &= 0xffffu;
                while (m) {
                    const int bit = __ffs((int)m) - 1;
                    m &= m - 1;
                    const int q = 15 - bit;
                    const uint32_t pos = atomicAdd(&sh.qcnt[q], 1u);
                    sh.queue[q][pos % 192u] = it * 64 + lane;
                    ++chk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the atomics' returns; the next reads are counted again from zero)
        }
        cur = nxt; nxt = nxt2;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[(int64_t)blockIdx.x * NT + tid] = chk;
}

template <int NT, int MINW, int PAD_KB>
static void run_p(const char *name, const uint4 *rows, int64_t n_rows, int nq, const uint32_t *lut, uint32_t *out, int thr)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int groups = nq / 16;
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((scan_kernel_p<NT, MINW, PAD_KB>), dim3(groups), dim3(NT), 0, 0, rows, n_rows, lut, thr, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<uint32_t> ho((size_t)NT);
    CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    uint64_t surv = 0;
    for (auto x : ho) surv += x;
    printf("%-34s NT=%4d minw=%d pad=%3dKB groups=%4d thr=%5d: %8.3f ms  %6.2f T look-ups/s  %7.0f K q/s at 1M rows; survivors/(row,query) of group 0 %.2e\n",
           name, NT, MINW, PAD_KB, groups, thr, best, (double)nq * n_rows * 16 / best / 1e9, nq / best, (double)surv / ((double)n_rows * 16));
    fflush(stdout);
}

static uint64_t rng_s = 88172645463325252ull;
static inline uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 16); }

template <int KIND, int NT, int MINW, int PAD_KB, bool CHECK = false, int NMF = 6>
static void run(const char *name, const uint4 *rows, const std::vector<uint32_t> &hrows, int64_t n_rows, int nq, const uint32_t *lut,
                const std::vector<uint32_t> &hlut, uint32_t *out, int thr, int *check)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int QT = KIND == K_U16 ? 8 : 16;
    const int groups = nq / QT;
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((scan_kernel<KIND, NT, MINW, PAD_KB, CHECK, NMF>), dim3(groups), dim3(NT), 0, 0, rows, n_rows, lut, thr, out, it == 0 ? check : nullptr);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<uint32_t> ho((size_t)NT);
    CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    uint64_t surv = 0;
    for (auto x : ho) surv += x;
    printf("%-34s NT=%4d minw=%d pad=%3dKB groups=%4d thr=%5d: %8.3f ms  %6.2f T look-ups/s  %7.0f K q/s at 1M rows; survivors/(row,query) of group 0 %.2e\n",
           name, NT, MINW, PAD_KB, groups, thr, best, (double)nq * n_rows * 16 / best / 1e9, nq / best, (double)surv / ((double)n_rows * QT));
    if (KIND == K_MFMA && CHECK) {
        std::vector<int> hc(64 * 16);
        CK(hipMemcpy(hc.data(), check, hc.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        const uint8_t *lb = reinterpret_cast<const uint8_t *>(hlut.data());
        const uint8_t *rb = reinterpret_cast<const uint8_t *>(hrows.data());
        for (int l = 0; l < 64; ++l)
            for (int q = 0; q < 16; ++q) {
                int s = 0;
                for (int t = 0; t < 16; ++t) {
                    const int code = rb[l * 16 + t], m = (t + l) & 15;
                    s += lb[code * 256 + m * 16 + q];
                }
                if (s != hc[l * 16 + q]) { if (bad < 4) printf("  MISMATCH lane %d query %d: device %d expected %d\n", l, q, hc[l * 16 + q], s); ++bad; }
            }
        printf("  layout check of the selector product (64 rows x 16 queries): %s\n", bad ? "FAILED" : "ok");
    }
    fflush(stdout);
}

int main()
{
    const int64_t n = 1000000;
    std::vector<uint32_t> h((size_t)n * 4);
    for (auto &w : h) w = rnd();
    std::vector<uint32_t> l16(256 * 16 * 4), l4(256 * 16 * 4), l6(256 * 16 * 4);
    for (auto &f : l16) f = (rnd() & 0x03ff03ff) + 0x00100010;
    for (auto &f : l4) f = rnd() & 0x0f0f0f0f;
    for (auto &f : l6) { uint32_t x = 0; for (int b = 0; b < 4; ++b) x |= (rnd() % 43u) << (8 * b); f = x; }
    uint4 *rows; uint32_t *lut16, *lut4, *lut6; uint32_t *out; int *check;
    CK(hipMalloc(&rows, n * 16 + 65536)); CK(hipMalloc(&lut16, l16.size() * 4)); CK(hipMalloc(&lut4, l4.size() * 4)); CK(hipMalloc(&lut6, l6.size() * 4));
    CK(hipMalloc(&out, (size_t)4096 * 1024 * 4)); CK(hipMalloc(&check, 64 * 16 * 4));
    CK(hipMemcpy(rows, h.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(lut16, l16.data(), l16.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lut4, l4.data(), l4.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lut6, l6.data(), l6.size() * 4, hipMemcpyHostToDevice));
    const int nq = 10240;
    // thresholds: u16 sums ~ 16 x 527 = 8400 +- 1200; u8 sums ~ 120 +- 18; mfma sums ~ 336 +- 50
    for (int rep = 0; rep < 2; ++rep) {
        run<K_U16, 1024, 8, 0>("u16 8q/read, 2 WG/CU", rows, h, n, nq, lut16, l16, out, 1, nullptr);
        run<K_U16, 1024, 8, 0>("u16 8q/read, 2 WG/CU, 3e-4", rows, h, n, nq, lut16, l16, out, 4300, nullptr);
        run<K_U8, 1024, 8, 0>("u8 16q/read, 2 WG/CU, none pass", rows, h, n, nq, lut4, l4, out, 20, nullptr);
        run<K_U8, 1024, 4, 70>("u8 16q/read, 1 WG/CU, none pass", rows, h, n, nq, lut4, l4, out, 20, nullptr);
        run<K_U8, 1024, 8, 0>("u8 16q/read, 2 WG/CU, ~3e-5", rows, h, n, nq, lut4, l4, out, 48, nullptr);
        run<K_U8, 1024, 8, 0>("u8 16q/read, 2 WG/CU, ~3e-4", rows, h, n, nq, lut4, l4, out, 58, nullptr);
        run<K_MFMA, 1024, 4, 70, true>("mfma 16q/read, 1 WG/CU (+check)", rows, h, n, nq, lut6, l6, out, 1, check);
        run<K_MFMA, 1024, 4, 70>("mfma 16q/read, 1 WG/CU", rows, h, n, nq, lut6, l6, out, 1, nullptr);
        run<K_MFMA, 1024, 4, 70>("mfma 16q/read, 1 WG/CU, ~3e-4", rows, h, n, nq, lut6, l6, out, 165, nullptr);
        run<K_MFMA, 1024, 4, 70>("mfma 16q/read, 1 WG/CU, ~3e-3", rows, h, n, nq, lut6, l6, out, 200, nullptr);
        run<K_MFMA, 1024, 4, 70, false, 4>("mfma (4,4,4,4) 1 WG/CU", rows, h, n, nq, lut6, l6, out, 1, nullptr);
        run<K_MFMA, 1024, 4, 70, false, 3>("mfma (5,5,6) 1 WG/CU", rows, h, n, nq, lut6, l6, out, 1, nullptr);
        run<K_MFMA, 1024, 4, 70, false, 2>("mfma (8,8) 1 WG/CU", rows, h, n, nq, lut6, l6, out, 1, nullptr);
        run<K_MFMA, 1024, 4, 70, false, 1>("mfma (16) 1 WG/CU", rows, h, n, nq, lut6, l6, out, 1, nullptr);
        run<K_MFMA, 512, 4, 0, false, 4>("mfma (4,4,4,4) 2 WG/CU x 8", rows, h, n, nq, lut6, l6, out, 1, nullptr);
        run<K_MFMA, 512, 4, 0, false, 2>("mfma (8,8) 2 WG/CU x 8", rows, h, n, nq, lut6, l6, out, 1, nullptr);
        run<K_MFMA, 512, 4, 0, false, 1>("mfma (16) 2 WG/CU x 8", rows, h, n, nq, lut6, l6, out, 1, nullptr);
        run_p<768, 3, 70>("mfma pipelined, 1 WG/CU x 12 waves", rows, n, nq, lut6, out, 1);
        run_p<768, 3, 70>("mfma pipelined, 1x12, ~2e-4", rows, n, nq, lut6, out, 165);
        run<K_MFMA, 512, 4, 0>("mfma 16q/read, 2 WG/CU x 8 waves", rows, h, n, nq, lut6, l6, out, 1, nullptr);
        run<K_MFMA, 512, 4, 0>("mfma 16q/read, 2x8 waves, ~3e-4", rows, h, n, nq, lut6, l6, out, 165, nullptr);
    }
    return 0;
}
