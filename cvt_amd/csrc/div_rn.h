// div_rn.h -- correctly rounded fp32 a / b for MANY dividends per divisor (SQ8 encode, PCA normalisation).
// With y = RN(1 / b) computed once, each quotient costs three instructions: q = RN(a*y), e = fma(-b, q, a),
// RN(q + e*y).  That is RN(a / b) whenever b's significand is not all ones and nothing leaves the normal range
// (Markstein's theorem; tools/ubench/div_check.c brute-forces 1.1e9 quotients incl. every significand of b);
// operands outside the guarded range take the compiler's IEEE division (out of line: it is rare).
#pragma once
#include "common.h"

namespace cvtmi {

struct DivBy {
    float b, y;  // divisor, RN(1 / b)
    bool ok;     // fast path allowed for this divisor
};
__device__ __forceinline__ DivBy div_by(float b)
{
    DivBy d;
    d.b = b;
    d.ok = b >= 0x1p-40f && b <= 0x1p40f && (__float_as_uint(b) & 0x7fffffu) != 0x7fffffu;
    d.y = d.ok ? __fdiv_rn(1.0f, b) : 0.0f;
    return d;
}
static __device__ __attribute__((noinline)) float div_slow(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, const DivBy &d)
{
    const float aa = fabsf(a);
    const float q = __fmul_rn(a, d.y);
    const float e = __fmaf_rn(-d.b, q, a);
    float r = __fmaf_rn(e, d.y, q);
    r = aa == 0.0f ? a : r;  // +-0 / positive b keeps its sign
    if (!(d.ok && (aa <= 0x1p60f && (aa >= 0x1p-60f || aa == 0.0f)))) r = div_slow(a, d.b);  // rare: out of line
    return r;
}

}  // namespace cvtmi
