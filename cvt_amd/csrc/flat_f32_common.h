// flat_f32_common.h -- helpers shared by the fp32 flat-search kernels (flat_f32_stream.hip: the stream over the rows;
// flat_f32_tfilter.hip: large batches as a threshold filter).  gfx950 only.
#pragma once
#include <atomic>

#include "common.h"

namespace cvtmi {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float FS_PAD_BIAS = -3.0e38f;  // rows past n inside the last 64-row block: their keys never win
constexpr float FS_EMPTY = -1.0e37f;     // best <= this: the lane saw no valid row in that group

// v = hi + lo + O(2^-17 |v|), both bf16 (round to nearest even)
__device__ __forceinline__ void fs_split(const float (&v)[8], bf16x8 &hi, bf16x8 &lo)
{
    union { bf16x8 v; uint32_t u[4]; } h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2 x = { v[2 * p], v[2 * p + 1] };
        const bf16x2 hp = __builtin_convertvector(x, bf16x2);
        const uint32_t hu = __builtin_bit_cast(uint32_t, hp);
        const f32x2 r = { v[2 * p] - __uint_as_float(hu << 16), v[2 * p + 1] - __uint_as_float(hu & 0xffff0000u) };
        const bf16x2 lp = __builtin_convertvector(r, bf16x2);
        h.u[p] = hu;
        l.u[p] = __builtin_bit_cast(uint32_t, lp);
    }
    hi = h.v;
    lo = l.v;
}

__device__ __forceinline__ float fs_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t fs_wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64); v = v > w ? v : w; }
    return v;
}
__device__ __forceinline__ uint32_t fs_wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64); v = v < w ? v : w; }
    return v;
}
// |q|^2 by one wave (any order: it only scales the margin)
__device__ __forceinline__ float fs_qnorm(const float *q, int D)
{
    float s = 0.0f;
    for (int e = threadIdx.x & 63; e < D; e += 64) s = __fmaf_rn(q[e], q[e], s);
    return fs_wave_sum(s);
}

// exact distance of blocked row `row` to the query qv (LDS) in the reference's summation order (dist_f32.h): the row's D / 4
// pieces (16 bytes each, 1 KB apart in the blocked layout) are requested 32 at a time
template <bool IP, int LANES, int NLOAD = 32>   // NLOAD pieces requested at a time (a 1024-thread workgroup has 128 registers per lane: 16)
__device__ __forceinline__ float fs_exact(const float *X, int D, int64_t row, const float4 *qv, const bool rowmajor = false)
{
    float acc[LANES];
#pragma unroll
    for (int l = 0; l < LANES; ++l) acc[l] = 0.0f;
    // rowmajor: X is the row-major copy of the rows (flat_unblock_kernel) -- the same pieces in the same order out of whole cache lines
    // (a piece of the blocked layout shares its 128-byte line with seven other rows: a gathered row costs eight times its bytes)
    const int64_t stride = rowmajor ? 1 : 64;
    const float4 *xr = reinterpret_cast<const float4 *>(X) + (rowmajor ? row * (int64_t)(D >> 2) : (row >> 6) * (int64_t)(D >> 2) * 64 + (row & 63));
    for (int c0 = 0; c0 < D / 4; c0 += NLOAD) {
        float4 xv[NLOAD];
#pragma unroll
        for (int c = 0; c < NLOAD; ++c) xv[c] = xr[(int64_t)(c0 + c < D / 4 ? c0 + c : 0) * stride];
#pragma unroll
        for (int c = 0; c < NLOAD; ++c) {
            if (c0 + c < D / 4) {
                const float4 qw = qv[c0 + c];
                const float xs[4] = { xv[c].x, xv[c].y, xv[c].z, xv[c].w }, qs[4] = { qw.x, qw.y, qw.z, qw.w };
                const int l0 = 4 * (c % (LANES / 4));   // (c0 is a multiple of NLOAD, NLOAD of LANES / 4)
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    if constexpr (IP) {
                        acc[l0 + l] = __fadd_rn(acc[l0 + l], __fmul_rn(qs[l], xs[l]));
                    } else {
                        const float t = __fsub_rn(qs[l], xs[l]);
                        acc[l0 + l] = __fadd_rn(acc[l0 + l], __fmul_rn(t, t));
                    }
                }
            }
        }
    }
    float sum = acc[0];
#pragma unroll
    for (int l = 1; l < LANES; ++l) sum = __fadd_rn(sum, acc[l]);
    return IP ? __fsub_rn(1.0f, sum) : sum;
}

// The k-th largest of a wave's keys (order-preserving uint32 keys, NK per lane; absent ones = 0), by ONE wave and without a
// barrier: MSB-first, a bit of the answer per round (keep the bit if at least k keys reach the trial value), starting below the
// common prefix of the largest and the smallest key.  Stops early at a trial value that between k and kmax keys reach -- any such
// value serves as a first threshold.  Returns the largest c found with count(keys >= c) >= k (0 if fewer than k keys are non-zero).
template <int NK>
__device__ __forceinline__ uint32_t fs_wave_select(const uint32_t (&key)[NK], int k, int kmax)
{
    uint32_t hi = 0, lo = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        hi = key[j] > hi ? key[j] : hi;
        lo = (key[j] < lo && key[j] != 0u) ? key[j] : lo;
    }
    hi = fs_wave_max_u32(hi);
    lo = fs_wave_min_u32(lo);
    if (hi <= lo) return hi;   // all present keys equal (or none present: 0)
    const int top = 31 - __builtin_clz(hi ^ lo);             // highest bit in which two keys differ
    uint32_t c = top == 31 ? 0u : (hi >> (top + 1)) << (top + 1);   // the common prefix (every key reaches it)
    for (int bit = top; bit >= 0; --bit) {
        const uint32_t trial = c | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < NK; ++j) cnt += __popcll(__ballot(key[j] >= trial));
        if (cnt >= k) {
            c = trial;
            if (cnt <= kmax) break;
        }
    }
    return c;
}

// |q - q1|^2 by one wave, q1 = the bf16 operand of fs_split (the same conversion, value for value)
__device__ __forceinline__ float fs_qlow(const float *q, int D)
{
    float s_ = 0.0f;
    for (int e0 = 8 * (threadIdx.x & 63); e0 < D; e0 += 512) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = q[e0 + e];
        bf16x8 h, l;
        fs_split(v, h, l);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float r = v[e] - (float)h[e];   // exact (Sterbenz-like: q1 is q rounded to 8 bits)
            s_ = __fmaf_rn(r, r, s_);
        }
    }
    return fs_wave_sum(s_);
}

// (host) the dynamic-LDS attribute of a kernel, once per device
static int fs_set_lds(const void *fn, size_t lds, std::atomic<bool> (&done)[16])
{
    int dev = 0;
    CVTMI_HIP(hipGetDevice(&dev));
    if (dev >= 16 || !done[dev].load(std::memory_order_acquire)) {   // the attribute is per device (several host threads may search at once: setting it twice is harmless)
        CVTMI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev < 16) done[dev].store(true, std::memory_order_release);
    }
    return CVTMI_OK;
}

}  // namespace
}  // namespace cvtmi
