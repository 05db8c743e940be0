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


// ---- non-temporal streaming accesses -------------------------------------------------------------------------------
// For data one CU touches once per launch (row streams in, result streams out): no place in L2 / Infinity Cache is asked for,
// which shortens issued -> landed by ~18 % on this chip (MI355X_MICROARCH.md "nt-weights") -- measured gains are quoted where
// a kernel uses them.  Never for data that concurrent workgroups share through L2 (the ADC scan's code rows, the filter
// kernels' row tiles: those lose 6-10 % with it).
#ifdef __HIPCC__
typedef float cvt_nt_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt(const float4 *p)
{
    const cvt_nt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const cvt_nt_f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t ld_nt(const uint32_t *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_nt(float4 *p, const float4 &v)
{
    cvt_nt_f32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<cvt_nt_f32x4 *>(p));
}
__device__ __forceinline__ void st_nt(uint32_t *p, uint32_t v) { __builtin_nontemporal_store(v, p); }
#endif

}  // namespace cvtmi
