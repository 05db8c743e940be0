// dist_f32.h -- fp32 distances in the summation order of the reference's own SIMD builds, so that results are
// bit-exact: inner product = 1 - sum in 4-lane SSE order (space_ip.hpp / space_ip.h: the AVX branches are not
// compiled), squared L2 in 8-lane AVX order when D % 16 == 0, 4-lane when D % 4 == 0 (space_l2.h:40-151), the
// scalar loop otherwise.  Shared by the flat search and the HNSW search.
#ifndef CVTMI_DIST_F32_H
#define CVTMI_DIST_F32_H
#include "common.h"

namespace cvtmi {

// LANES = 1 (scalar loop), 4 or 8;  IP = inner product, else squared L2
template <bool IP, int LANES, int QT>
__device__ __forceinline__ void dist_f32_row(const float *__restrict__ row, const float *qs, int D, float (&out)[QT])
{
    float acc[QT][LANES];
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int l = 0; l < LANES; ++l) acc[q][l] = 0.0f;
    if constexpr (LANES == 1) {
        for (int i = 0; i < D; ++i) {
            const float xv = row[i];
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                if constexpr (IP) {
                    acc[q][0] = __fadd_rn(acc[q][0], __fmul_rn(qs[q * D + i], xv));
                } else {
                    const float t = __fsub_rn(qs[q * D + i], xv);
                    acc[q][0] = __fadd_rn(acc[q][0], __fmul_rn(t, t));
                }
            }
        }
    } else {
        for (int i = 0; i < D; i += LANES) {
            float xv[LANES];
#pragma unroll
            for (int l4 = 0; l4 < LANES / 4; ++l4) {
                const float4 v = *reinterpret_cast<const float4 *>(row + i + 4 * l4);
                xv[4 * l4 + 0] = v.x; xv[4 * l4 + 1] = v.y; xv[4 * l4 + 2] = v.z; xv[4 * l4 + 3] = v.w;
            }
#pragma unroll
            for (int q = 0; q < QT; ++q) {
#pragma unroll
                for (int l = 0; l < LANES; ++l) {
                    if constexpr (IP) {
                        acc[q][l] = __fadd_rn(acc[q][l], __fmul_rn(qs[q * D + i + l], xv[l]));
                    } else {
                        const float t = __fsub_rn(qs[q * D + i + l], xv[l]);
                        acc[q][l] = __fadd_rn(acc[q][l], __fmul_rn(t, t));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        float s = acc[q][0];
#pragma unroll
        for (int l = 1; l < LANES; ++l) s = __fadd_rn(s, acc[q][l]);
        out[q] = IP ? __fsub_rn(1.0f, s) : s;
    }
}


}  // namespace cvtmi
#endif
