// assign_mfma.hip -- nearest-centroid assignment (coarse argmin of IVFOPQ::Add, IVFOPQ.cpp:110-129; the assignment
// pass of the k-means in kmeans.hip) through a matrix-core FILTER, for 32 <= d <= 128 (d % 16 == 0).
//
// Same idea as pq_encode_mfma_kernel (opq_encode.hip), with a d-deep product instead of a step-deep one: the
// assignment is an argmin, T_j = x.c_j - |c_j|^2 / 2 ranks the centroids like the distances do, and a bf16
// two-term split of both operands gives T to ~2^-17 on v_mfma_f32_32x32x16_bf16 (which runs beside the VALU at
// 16x the fp32 rate).  When the best T beats the second best by more than the proven bound, the reference's
// sequential fp32 argmin is that centroid; rows where it does not (near ties, duplicates, non-finite values,
// distances near the reference's start value) are collected and go through the exact kernels of kmeans.hip.
//
//   pack      centroids [k][d] fp32 -> operand-ordered bf16 halves [tile][chunk][half][lane] (one coalesced 1 KB read per
//             matrix operand and wave), -|c|^2/2 as two bf16 terms per centroid, max |c|^2
//   filter    a wave owns 32 rows (B side; both bf16 halves of the row stay in registers: 16 d bytes); the 8 waves of a
//             workgroup sweep the centroid tiles together (A side: tile t+1 HBM/L2 -> registers -> LDS under the products
//             of tile t, one LDS-only barrier per tile) with 3 d/16 + 1 products per 32 x 32 tile (c1.r1, c1.r2, c2.r1; c2.r2 is below the bound), keeps (best, second, tile of best)
//             per lane: key = T with its low 4 bits replaced by the accumulator element
//   resolve   flagged rows: gathered, assigned by the exact kernel, scattered back
//
// Bound (u = 2^-24, Q = |x|^2 + max |c|^2, |T| <= Q, d_j <= 2Q), d = 128: accumulation of 3 d + 2 terms taken as
// 2u per term 388 uQ, bf16 splits 64 uQ, the omitted c2.r2 term 32 uQ, |c|^2/2 split 24 uQ, position bits 16 uQ: 524 uQ per
// key; the reference's chain (d + 3) u d_j = 131 uQ in T units.  best - second > 2 * 524 + 131 = 1179 uQ proves the argmin; the kernel asks
// for 2048 uQ = 2^-13 Q and 2^-60 < Q < 2^30.
#include <algorithm>
#include <mutex>

#include "block_topk.h"
#include "common.h"
#include "kernels.h"

namespace cvtmi {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float kPadT = -1.0e30f;  // -|c|^2/2 of the padding centroids of the last tile: they never win

// one thread per (tile, lane): lane (li, lk) of tile t owns centroid 32 t + li, dimensions 16 c + 8 lk .. + 8 of chunk c
__global__ __launch_bounds__(kBlock) void assign_pack_kernel(const float *__restrict__ cent, int k, int d, int ntiles,
                                                             uint4 *__restrict__ packA, uint32_t *__restrict__ nhcp,
                                                             uint32_t *__restrict__ cmax2)
{
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= ntiles * 64) return;
    const int t = g >> 6, lane = g & 63, li = lane & 31, lk = lane >> 5;
    const int j = t * 32 + li, nch = d / 16;
    float s = 0.0f;
    for (int c = 0; c < nch; ++c) {
        union { bf16x8 v; uint4 u; } h1, h2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = j < k ? cent[(int64_t)j * d + 16 * c + 8 * lk + e] : 0.0f;
            const __bf16 a = (__bf16)v;
            h1.v[e] = a;
            h2.v[e] = (__bf16)(v - (float)a);
        }
        packA[((int64_t)(t * nch + c) * 2 + 0) * 64 + lane] = h1.u;
        packA[((int64_t)(t * nch + c) * 2 + 1) * 64 + lane] = h2.u;
    }
    if (lk == 0) {  // |c|^2 as the fp32 fma chain over all d dimensions
        float nh = kPadT;
        if (j < k) {
            for (int e = 0; e < d; ++e) s = __fmaf_rn(cent[(int64_t)j * d + e], cent[(int64_t)j * d + e], s);
            nh = -0.5f * s;
            atomicMax(cmax2, __float_as_uint(s));  // NaN / inf end up as NaN / inf: every row takes the exact path
        }
        const __bf16 a = (__bf16)nh, b = (__bf16)(nh - (float)a);
        nhcp[j] = (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
    }
}

constexpr int AF_THREADS = 512;  // 8 waves = 256 rows per workgroup pass; the centroid tile is shared through LDS

template <int NCH>
__global__ __launch_bounds__(AF_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void assign_filter_kernel(const float *__restrict__ x, int64_t ld, int64_t n,
                                                                   const uint4 *__restrict__ packA, const uint32_t *__restrict__ nhcp,
                                                                   const uint32_t *__restrict__ cmax2, int ntiles,
                                                                   int32_t *__restrict__ assign, unsigned long long *__restrict__ changed,
                                                                   int32_t *__restrict__ flag_list, unsigned int *__restrict__ flag_cnt)
{
    constexpr int WAVES = AF_THREADS / 64;
    constexpr int TILE = NCH * 2 * 64;                          // uint4 per centroid tile (both bf16 halves, operand order)
    constexpr int LPT = (TILE + AF_THREADS - 1) / AF_THREADS;   // uint4 per thread and tile
    __shared__ uint4 tile_s[2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const float pinf = __uint_as_float(0x7f800000u | (uint32_t)(n < 0));  // opaque +inf: v_med3(a, b, +inf) = max, no canonicalising pre-pass
    const float ninf = -pinf;
    const float cm = __uint_as_float(*cmax2);
    const int64_t nblk = (n + 32 * WAVES - 1) / (32 * WAVES);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {  // every wave of the workgroup makes the same trips (barriers inside)
        const int64_t row = (blk * WAVES + wave) * 32 + li;
        const int64_t rowc = row < n ? row : n - 1;  // clamped: tail rows are computed, never stored
        const float *xp = x + rowc * ld + 8 * lk;
        bf16x8 r1[NCH], r2[NCH];
        float rr = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float v[8];
            *reinterpret_cast<float4 *>(&v[0]) = *reinterpret_cast<const float4 *>(xp + 16 * c);
            *reinterpret_cast<float4 *>(&v[4]) = *reinterpret_cast<const float4 *>(xp + 16 * c + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                rr = __fmaf_rn(v[e], v[e], rr);
                const __bf16 h = (__bf16)v[e];
                r1[c][e] = h;
                r2[c][e] = (__bf16)(v[e] - (float)h);
            }
        }
        rr += __shfl_xor(rr, 32, 64);
        const float Q = (rr + cm) * 1.001f;
        const bf16x8 bzero = { 0, 0, 0, 0, 0, 0, 0, 0 };
        bf16x8 ones = bzero;
        ones[0] = lk ? (__bf16)0.0f : (__bf16)1.0f;
        ones[1] = ones[0];
        float B1 = ninf, B2 = ninf;
        int Bt = 0;
        // tile t+1 travels HBM/L2 -> registers while tile t (in LDS) feeds the products; one LDS-only barrier per tile
        uint4 pre[LPT];
        auto fetch = [&](int t) {
            t = t < ntiles ? t : ntiles - 1;
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                int f = tid + i * AF_THREADS;
                f = f < TILE ? f : TILE - 1;
                pre[i] = packA[(int64_t)t * TILE + f];
            }
        };
        fetch(0);
        __syncthreads();  // the previous pass is done with both stages
        for (int t = 0; t < ntiles; ++t) {
            uint4 *stage = tile_s[t & 1];
#pragma unroll
            for (int i = 0; i < LPT; ++i)
                if (tid + i * AF_THREADS < TILE) stage[tid + i * AF_THREADS] = pre[i];
            fetch(t + 1);
            const uint32_t hv = nhcp[t * 32 + li];
            lds_barrier();
            const uint4 *pa = stage + lane;
            const f32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            f32x16 acc0 = zero, acc1 = zero;  // two chains (x.r1 / x.r2 terms): dependent products are two apart
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                union { uint4 u; bf16x8 v; } a1, a2;
                a1.u = pa[(c * 2 + 0) * 64];
                a2.u = pa[(c * 2 + 1) * 64];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, r1[c], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, r2[c], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, r1[c], acc0, 0, 0, 0);  // c2.r2 <= 2^-18 |c||r| is left out (bound)
            }
            {
                union { uint32_t u[4]; bf16x8 v; } ab = { { lk ? 0u : hv, 0u, 0u, 0u } };
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab.v, ones, acc0, 0, 0, 0);  // - |c|^2 / 2
            }
            float b1 = ninf, b2 = ninf;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float key = __uint_as_float((__float_as_uint(acc0[e] + acc1[e]) & 0xfffffff0u) | (uint32_t)e);
                b2 = __builtin_amdgcn_fmed3f(b1, b2, key);
                b1 = __builtin_amdgcn_fmed3f(b1, key, pinf);
            }
            // (B1, B2) <- top two of {B1, B2, b1, b2}; the tile of the best travels with it
            const float lo = __builtin_amdgcn_fmed3f(B1, b1, ninf), s2 = __builtin_amdgcn_fmed3f(B2, b2, pinf);
            Bt = b1 > B1 ? t : Bt;
            B1 = __builtin_amdgcn_fmed3f(B1, b1, pinf);
            B2 = __builtin_amdgcn_fmed3f(lo, s2, pinf);
        }
        // the two lane halves hold different centroid rows of every tile
        const float p1 = __shfl_xor(B1, 32, 64), p2 = __shfl_xor(B2, 32, 64);
        const int pt = __shfl_xor(Bt, 32, 64);
        const bool other = p1 > B1;
        const int win_lk = other ? (lk ^ 1) : lk, win_t = other ? pt : Bt;
        const float f1 = fmaxf(B1, p1), f2 = fmaxf(fminf(B1, p1), fmaxf(B2, p2));
        const int e = (int)(__float_as_uint(f1) & 15u);
        const int best = win_t * 32 + (e & 3) + 8 * (e >> 2) + 4 * win_lk;
        const bool sure = (f1 - f2 > Q * 0x1p-13f) && (Q < 0x1p30f) && (Q > 0x1p-60f);  // false for NaN anywhere
        const bool mine = lk == 0 && row < n;
        bool ch = false;
        if (mine && sure) {
            ch = assign[row] != best;
            assign[row] = best;
        }
        const unsigned long long fm = __ballot(mine && !sure);
        if (fm) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(flag_cnt, (unsigned int)__popcll(fm));
            base = __shfl(base, 0, 64);
            if (mine && !sure) flag_list[base + __popcll(fm & ((1ull << lane) - 1))] = (int32_t)row;
        }
        if (changed) {
            const unsigned long long m = __ballot(ch);
            if (lane == 0 && m) atomicAdd(changed, (unsigned long long)__popcll(m));
        }
    }
}

__global__ __launch_bounds__(kBlock) void assign_gather_rows_kernel(const float *__restrict__ x, int64_t ld, int d,
                                                                    const int32_t *__restrict__ flag_list, int nf,
                                                                    float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)nf * d) return;
    const int r = (int)(i / d), e = (int)(i - (int64_t)r * d);
    out[i] = x[(int64_t)flag_list[r] * ld + e];
}

__global__ __launch_bounds__(kBlock) void assign_scatter_kernel(const int32_t *__restrict__ flag_list, const int32_t *__restrict__ exact,
                                                                int nf, int32_t *__restrict__ assign,
                                                                unsigned long long *__restrict__ changed)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    bool ch = false;
    if (i < nf) {
        const int32_t row = flag_list[i];
        ch = assign[row] != exact[i];
        assign[row] = exact[i];
    }
    if (changed) {
        const unsigned long long m = __ballot(ch);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(changed, (unsigned long long)__popcll(m));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Coarse top-nk of IVFOPQ::Query (IVFOPQ.cpp:238-260) through the same filter: the nk nearest of coarseK centroids per query
// frame, by the reference's sequential fp32 distance, (distance, index) order.
//   score    T[query][centroid] = q.c - |c|^2/2 for every pair, on the matrix cores (the products of assign_filter_kernel with the
//            operand roles swapped: D[query i][centroid j], so a lane holds one centroid column and 16 queries, and a store
//            instruction writes two 128-byte runs of one query's row)
//   select   one workgroup per query: theta = the nk-th largest T; a centroid with T < theta - 2^-13 Q is beaten by nk others in
//            the reference's own arithmetic (the assignment's pairwise bound: 2 x 524 uQ of filter error + 131 uQ of reference
//            rounding < 2048 uQ), so only the few with T >= theta - 2^-13 Q get the reference's distance chain; those are sorted
//            by (distance, index).  A query whose candidate list overflows (near-equidistant centroids), or whose magnitudes
//            leave the bound's range (non-finite values), walks all centroids exactly inside the same workgroup.
// ------------------------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(AF_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe_score_kernel(const float *__restrict__ x, int64_t ld, int64_t n,
                                                                 const uint4 *__restrict__ packA, const uint32_t *__restrict__ nhcp, int ntiles,
                                                                 float *__restrict__ T, int64_t ldT)
{
    constexpr int WAVES = AF_THREADS / 64;
    constexpr int TILE = NCH * 2 * 64;
    constexpr int LPT = (TILE + AF_THREADS - 1) / AF_THREADS;
    __shared__ uint4 tile_s[2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int64_t nblk = (n + 32 * WAVES - 1) / (32 * WAVES);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t row0 = (blk * WAVES + wave) * 32;
        const int64_t row = row0 + li;
        const int64_t rowc = row < n ? row : n - 1;
        const float *xp = x + rowc * ld + 8 * lk;
        bf16x8 r1[NCH], r2[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float v[8];
            *reinterpret_cast<float4 *>(&v[0]) = *reinterpret_cast<const float4 *>(xp + 16 * c);
            *reinterpret_cast<float4 *>(&v[4]) = *reinterpret_cast<const float4 *>(xp + 16 * c + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 h = (__bf16)v[e];
                r1[c][e] = h;
                r2[c][e] = (__bf16)(v[e] - (float)h);
            }
        }
        const bf16x8 bzero = { 0, 0, 0, 0, 0, 0, 0, 0 };
        bf16x8 ones = bzero;
        ones[0] = lk ? (__bf16)0.0f : (__bf16)1.0f;
        ones[1] = ones[0];
        uint4 pre[LPT];
        auto fetch = [&](int t) {
            t = t < ntiles ? t : ntiles - 1;
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                int f = tid + i * AF_THREADS;
                f = f < TILE ? f : TILE - 1;
                pre[i] = packA[(int64_t)t * TILE + f];
            }
        };
        // (grid row y scores its own range of centroid tiles: with the queries alone -- 256 per workgroup -- 10 000 frames made 40 workgroups)
        const int tpb = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y, t0 = (int)blockIdx.y * tpb, t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
        fetch(t0);
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            uint4 *stage = tile_s[t & 1];
#pragma unroll
            for (int i = 0; i < LPT; ++i)
                if (tid + i * AF_THREADS < TILE) stage[tid + i * AF_THREADS] = pre[i];
            fetch(t + 1);
            const uint32_t hv = nhcp[t * 32 + li];
            lds_barrier();
            const uint4 *pa = stage + lane;
            const f32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            f32x16 acc0 = zero, acc1 = zero;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {  // queries are the A side here: D[query i][centroid j], lane = centroid li
                union { uint4 u; bf16x8 v; } a1, a2;
                a1.u = pa[(c * 2 + 0) * 64];
                a2.u = pa[(c * 2 + 1) * 64];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1[c], a1.v, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r2[c], a1.v, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1[c], a2.v, acc0, 0, 0, 0);
            }
            {
                union { uint32_t u[4]; bf16x8 v; } ab = { { lk ? 0u : hv, 0u, 0u, 0u } };
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, ab.v, acc0, 0, 0, 0);  // - |c|^2 / 2 of centroid li
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t qrow = row0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (qrow < n) T[qrow * ldT + t * 32 + li] = acc0[e] + acc1[e];
            }
        }
    }
}

constexpr int PS_CAND = 96;    // candidates per query that get an exact distance
constexpr int PS_CAP = 1024;   // selection buffer of the theta pass
constexpr int PS_TRIG = 768;

__global__ __launch_bounds__(kBlock) void probe_select_kernel(const float *__restrict__ T, int64_t ldT, const float *__restrict__ q, int D,
                                                              const float *__restrict__ coarse, int coarseK, const uint32_t *__restrict__ cmax2,
                                                              int nprobe, int32_t *__restrict__ probe)
{
    extern __shared__ __attribute__((aligned(16))) float ps_q[];  // D floats
    __shared__ TopKShared<1, PS_CAP> tk;
    __shared__ int cand_s[PS_CAND];
    __shared__ unsigned long long key_s[128];
    __shared__ int ncand_s;
    __shared__ float red_s[kBlock / 64];
    const int64_t qi = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    float part = 0.0f;
    for (int d = tid; d < D; d += kBlock) {
        const float v = q[qi * D + d];
        ps_q[d] = v;
        part = __fmaf_rn(v, v, part);
    }
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) red_s[tid >> 6] = part;
    if (tid == 0) ncand_s = 0;
    topk_init(tk);
    __syncthreads();
    float qq = 0.0f;
    for (int w = 0; w < kBlock / 64; ++w) qq += red_s[w];
    const float Q = (qq + __uint_as_float(*cmax2)) * 1.001f;
    const bool bounded = Q < 0x1p30f && Q > 0x1p-60f;  // false for NaN / inf anywhere
    auto exact_dist = [&](int c) -> float {  // the reference's chain (IVFOPQ.cpp:244-248)
        const float *cp = coarse + (int64_t)c * D;
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float t = __fsub_rn(ps_q[d], cp[d]);
            acc = __fadd_rn(acc, __fmul_rn(t, t));
        }
        return acc;
    };
    bool exact_all = !bounded || nprobe > PS_CAND / 2;
    if (!exact_all) {
        // theta = the nprobe-th largest T of this query's row (keys ascending = T descending)
        const float *Tr = T + qi * ldT;
        int tile = 0;
        for (int base = 0; base < coarseK; base += kBlock * 4, ++tile) {
            uint32_t key[4][1];
            uint32_t pay[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = base + r * kBlock + tid;
                pay[r] = (uint32_t)c;
                key[r][0] = KEY_MAX;
                if (c < coarseK) {
                    const uint32_t kk = ~f32_key(Tr[c]);
                    key[r][0] = kk == KEY_MAX ? KEY_MAX - 1 : kk;
                }
            }
            topk_tile<1, 4, PS_CAP, PS_TRIG>(tk, nprobe, tile, key, pay);
        }
        __syncthreads();
        topk_compact<1, PS_CAP>(tk, nprobe);
        const int cnt = tk.cnt[0];
        const float theta = cnt >= nprobe ? key_f32(~(uint32_t)(tk.buf[0][nprobe - 1] >> 32)) : -__uint_as_float(0x7f800000u);
        const float cut = theta - Q * 0x1p-13f;
        for (int c = tid; c < coarseK; c += kBlock) {
            if (Tr[c] >= cut) {
                const int slot = atomicAdd(&ncand_s, 1);
                if (slot < PS_CAND) cand_s[slot] = c;
            }
        }
        __syncthreads();
        const int nc = ncand_s;
        if (nc > PS_CAND || nc < (nprobe < coarseK ? nprobe : coarseK)) exact_all = true;  // crowded band (or a NaN score hid a centroid)
        else {
            if (tid < 128) {
                unsigned long long e = ~0ull;
                if (tid < nc) e = ((unsigned long long)__float_as_uint(exact_dist(cand_s[tid])) << 32) | (uint32_t)cand_s[tid];
                key_s[tid] = e;
            }
            __syncthreads();
            for (int k2 = 2; k2 <= 128; k2 <<= 1)
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    if (tid < 128) {
                        const int p = tid ^ j;
                        if (p > tid) {
                            const unsigned long long a = key_s[tid], b = key_s[p];
                            const bool up = (tid & k2) == 0;
                            if (up ? a > b : a < b) { key_s[tid] = b; key_s[p] = a; }
                        }
                    }
                    __syncthreads();
                }
            for (int i = tid; i < nprobe; i += kBlock) probe[qi * nprobe + i] = i < nc ? (int32_t)(uint32_t)key_s[i] : -1;
        }
    }
    if (exact_all) {  // workgroup-uniform: every centroid through the reference's chain
        __syncthreads();
        topk_init(tk);
        __syncthreads();
        int tile = 0;
        for (int base = 0; base < coarseK; base += kBlock, ++tile) {
            const int c = base + tid;
            uint32_t key[1][1] = { { KEY_MAX } };
            uint32_t pay[1] = { (uint32_t)c };
            if (c < coarseK) {
                const uint32_t kk = __float_as_uint(exact_dist(c));
                key[0][0] = kk == KEY_MAX ? KEY_MAX - 1 : kk;
            }
            topk_tile<1, 1, PS_CAP, PS_TRIG>(tk, nprobe, tile, key, pay);
        }
        __syncthreads();
        topk_compact<1, PS_CAP>(tk, nprobe);
        const int cnt = tk.cnt[0];
        for (int i = tid; i < nprobe; i += kBlock) probe[qi * nprobe + i] = i < cnt ? (int32_t)(uint32_t)tk.buf[0][i] : -1;
    }
}

// grow-only device scratch of this translation unit; calls that use it are serialised (they end in a stream
// synchronisation anyway)
struct AssignScratch {
    void *p = nullptr;
    size_t bytes = 0;
    int dev = -1;
    int reserve(size_t need)
    {
        int cur = 0;
        CVTMI_HIP(hipGetDevice(&cur));
        if (p && cur == dev && bytes >= need) return CVTMI_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        CVTMI_HIP(hipMalloc(&p, need));
        bytes = need;
        dev = cur;
        return CVTMI_OK;
    }
};
static AssignScratch g_scr;
static std::mutex g_scr_mutex;

bool assign_filter_applies(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k)
{
    return d >= 32 && d <= 128 && d % 16 == 0 && k >= 64 && n >= 4096 && n < 0x7fffffff && ld % 4 == 0 &&
           ((((uintptr_t)x) | ((uintptr_t)cent)) & 15) == 0;
}

// exact: the kernels of kmeans.hip (reference chain for every centroid)
int launch_assign_filtered(const float *x, int64_t ld, int64_t n, int d, const float *cent, int k, int32_t *assign,
                           unsigned long long *changed, hipStream_t st)
{
    std::lock_guard<std::mutex> guard(g_scr_mutex);
    const int nch = d / 16, ntiles = (k + 31) / 32;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t b_pack = up((size_t)ntiles * nch * 2 * 64 * sizeof(uint4)), b_nhc = up((size_t)ntiles * 32 * 4), b_misc = 256,
                 b_flag = up((size_t)n * 4);
    CVTMI_TRY(g_scr.reserve(b_pack + b_nhc + b_misc + b_flag));
    char *base = static_cast<char *>(g_scr.p);
    uint4 *packA = reinterpret_cast<uint4 *>(base);
    uint32_t *nhcp = reinterpret_cast<uint32_t *>(base + b_pack);
    uint32_t *cmax2 = reinterpret_cast<uint32_t *>(base + b_pack + b_nhc);
    unsigned int *flag_cnt = cmax2 + 1;
    int32_t *flag_list = reinterpret_cast<int32_t *>(base + b_pack + b_nhc + b_misc);
    CVTMI_HIP(hipMemsetAsync(cmax2, 0, 8, st));
    hipLaunchKernelGGL(assign_pack_kernel, dim3((unsigned)((ntiles * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, cent, k, d, ntiles,
                       packA, nhcp, cmax2);
    constexpr int rows_per_block = 32 * (AF_THREADS / 64);
    const int64_t blocks = std::min<int64_t>((n + rows_per_block - 1) / rows_per_block, 256 * 4);
#define CVTMI_AF(N)                                                                                                                \
    case N:                                                                                                                        \
        hipLaunchKernelGGL((assign_filter_kernel<N>), dim3((unsigned)blocks), dim3(AF_THREADS), 0, st, x, ld, n, packA, nhcp, cmax2, ntiles, \
                           assign, changed, flag_list, flag_cnt);                                                                  \
        break;
    switch (nch) {
        CVTMI_AF(2) CVTMI_AF(3) CVTMI_AF(4) CVTMI_AF(5) CVTMI_AF(6) CVTMI_AF(7) CVTMI_AF(8)
        default: return fail(CVTMI_EUNSUPPORTED, "assign filter: d=%d", d);
    }
#undef CVTMI_AF
    CVTMI_HIP(hipGetLastError());
    unsigned int nf = 0;
    CVTMI_HIP(hipMemcpyAsync(&nf, flag_cnt, sizeof nf, hipMemcpyDeviceToHost, st));
    CVTMI_HIP(hipStreamSynchronize(st));
    if (nf == 0) return CVTMI_OK;
    // the rows the filter could not decide: exact kernel on a compact copy, the centroid range spread over enough
    // workgroups to fill the chip (a handful of row blocks would otherwise walk all k centroids on a handful of CUs)
    const int row_blocks = (int)((nf + kBlock - 1) / kBlock);
    int splits = std::max(1, std::min((k + 63) / 64, (1024 + row_blocks - 1) / row_blocks));
    float *rows = nullptr, *part_d = nullptr;
    int32_t *exact = nullptr, *part_i = nullptr;
    const size_t b_rows = up((size_t)nf * d * sizeof(float)), b_ex = up((size_t)nf * 4), b_part = up((size_t)nf * splits * 4);
    char *tmp = nullptr;
    CVTMI_HIP(hipMalloc(reinterpret_cast<void **>(&tmp), b_rows + b_ex + 2 * b_part));
    rows = reinterpret_cast<float *>(tmp);
    exact = reinterpret_cast<int32_t *>(tmp + b_rows);
    part_d = reinterpret_cast<float *>(tmp + b_rows + b_ex);
    part_i = reinterpret_cast<int32_t *>(tmp + b_rows + b_ex + b_part);
    hipLaunchKernelGGL(assign_gather_rows_kernel, dim3((unsigned)(((int64_t)nf * d + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, x, ld, d,
                       flag_list, (int)nf, rows);
    int rc = launch_kmeans_assign_split(rows, d, nf, d, cent, k, exact, splits, part_d, part_i, st);
    if (rc == CVTMI_OK) {
        hipLaunchKernelGGL(assign_scatter_kernel, dim3((nf + kBlock - 1) / kBlock), dim3(kBlock), 0, st, flag_list, exact, (int)nf, assign, changed);
        if (hipGetLastError() != hipSuccess) rc = fail(CVTMI_EHIP, "assign filter: scatter launch failed");
    }
    (void)hipStreamSynchronize(st);  // the temporaries die with this frame
    (void)hipFree(tmp);
    return rc;
}

bool coarse_probe_filter_applies(const float *q, int64_t nq, int d, const float *cent, int k, int nprobe)
{
    return d >= 32 && d <= 128 && d % 16 == 0 && k >= 256 && nq >= 256 && nprobe <= PS_CAND / 2 && nq < 0x7fffffff &&
           ((((uintptr_t)q) | ((uintptr_t)cent)) & 15) == 0;
}

// probe[nq][nprobe] = the nprobe nearest centroids per query, (distance, index) order, -1 padding
int launch_coarse_probe_filtered(const float *q, int64_t nq, int d, const float *cent, int k, int nprobe, int32_t *probe, hipStream_t st)
{
    std::lock_guard<std::mutex> guard(g_scr_mutex);
    const int nch = d / 16, ntiles = (k + 31) / 32;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int64_t ldT = (int64_t)ntiles * 32;
    const int64_t chunk = std::max<int64_t>(256, std::min<int64_t>(nq, (int64_t)((512ull << 20) / ((size_t)ldT * sizeof(float)))));
    const size_t b_pack = up((size_t)ntiles * nch * 2 * 64 * sizeof(uint4)), b_nhc = up((size_t)ntiles * 32 * 4), b_misc = 256,
                 b_T = up((size_t)chunk * ldT * sizeof(float));
    CVTMI_TRY(g_scr.reserve(b_pack + b_nhc + b_misc + b_T));
    char *base = static_cast<char *>(g_scr.p);
    uint4 *packA = reinterpret_cast<uint4 *>(base);
    uint32_t *nhcp = reinterpret_cast<uint32_t *>(base + b_pack);
    uint32_t *cmax2 = reinterpret_cast<uint32_t *>(base + b_pack + b_nhc);
    float *T = reinterpret_cast<float *>(base + b_pack + b_nhc + b_misc);
    CVTMI_HIP(hipMemsetAsync(cmax2, 0, 8, st));
    hipLaunchKernelGGL(assign_pack_kernel, dim3((unsigned)((ntiles * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, cent, k, d, ntiles,
                       packA, nhcp, cmax2);
    constexpr int rows_per_block = 32 * (AF_THREADS / 64);
    for (int64_t a = 0; a < nq; a += chunk) {
        const int64_t m = std::min(chunk, nq - a);
        const int64_t blocks = std::min<int64_t>((m + rows_per_block - 1) / rows_per_block, 256 * 4);
        // centroid-tile ranges per query block: enough workgroups for two per CU, at least 8 tiles (256 centroids) each
        const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ntiles / 8, 32), (512 + blocks - 1) / blocks));
#define CVTMI_PS(N)                                                                                                                   \
    case N:                                                                                                                           \
        hipLaunchKernelGGL((probe_score_kernel<N>), dim3((unsigned)blocks, gy), dim3(AF_THREADS), 0, st, q + a * d, (int64_t)d, m, packA, nhcp, ntiles, T, ldT); \
        break;
        switch (nch) {
            CVTMI_PS(2) CVTMI_PS(3) CVTMI_PS(4) CVTMI_PS(5) CVTMI_PS(6) CVTMI_PS(7) CVTMI_PS(8)
            default: return fail(CVTMI_EUNSUPPORTED, "probe filter: d=%d", d);
        }
#undef CVTMI_PS
        hipLaunchKernelGGL(probe_select_kernel, dim3((unsigned)m), dim3(kBlock), (size_t)d * sizeof(float), st, T, ldT, q + a * d, d, cent, k,
                           cmax2, nprobe, probe + a * nprobe);
        CVTMI_HIP(hipGetLastError());
    }
    CVTMI_HIP(hipStreamSynchronize(st));  // the shared scratch is free for the next caller when the lock is dropped
    return CVTMI_OK;
}

}  // namespace cvtmi
