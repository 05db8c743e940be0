// common.h -- shared host/device helpers of libcvtmi (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/cvtmi.h"

namespace cvtmi {

// ---- error plumbing: thread-local message + status codes, nothing throws across the C ABI ----
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define CVTMI_HIP(expr)                                                                          \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return ::cvtmi::fail(CVTMI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                 __FILE__, __LINE__);                                            \
    } while (0)

#define CVTMI_TRY(expr)                 \
    do {                                \
        int rc__ = (expr);              \
        if (rc__ != CVTMI_OK) return rc__; \
    } while (0)

// ---- order-preserving float <-> uint32 keys ----
constexpr uint32_t KEY_MAX = 0xFFFFFFFFu;
// any float (negative distances of the inner-product metric included)
__device__ __forceinline__ uint32_t f32_key(float f)
{
    uint32_t u = __float_as_uint(f);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(u);
}

constexpr int kBlock = 256;  // default workgroup: 256 threads (4 waves); the scan, row-tile and SQ8 tile kernels pick their own

#ifdef __HIPCC__
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load
// (s_waitcnt vmcnt(0)), which throws away a register prefetch that is meant to stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#endif

__host__ __device__ constexpr int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace cvtmi
