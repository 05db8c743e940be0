// flat_mfma.hip -- exhaustive fp32 search (BruteforceSearch + InnerProductSpace / L2Space, brutoforce.hpp:73-93,
// space_ip.hpp, space_l2.h) through a matrix-core FILTER, for 32 <= D <= 128 (D % 16 == 0) and batches of queries.
//
// The result -- k smallest (distance, row) per query with the distances in the reference's summation order --
// must come out bit for bit, so the distances themselves are evaluated by the exact code (dist_f32.h).  What the
// matrix cores do is decide WHICH rows need an exact distance:
//   1. the first n/16 rows (n/32 for k <= 16) are searched exactly (flat.hip; for large batches in two levels: the exact
//      kernels see a sixteenth of that sample and one filter stage extends it).  The k-th best of that sample, tau_q, bounds the k-th
//      best of the whole set from above.
//   2. T = q.x + b_x ranks the remaining rows (b_x = -|x|^2/2 for L2, 0 for the inner product); a bf16 two-term
//      split of both operands evaluates it to ~2^-17 on v_mfma_f32_32x32x16_bf16.  A row can only be in the
//      top k if its reference distance is <= tau_q, which implies T >= thr_q once every rounding (the split,
//      the accumulation inside the matrix unit, the reference's own sum) is bounded -- all other rows are
//      provably out.  Survivors (~15 k per query) are appended to a per-query candidate list.
//   3. a second cut on the approximate scores (a survivor far enough below the k-th best survivor is beaten by k rows),
//      then the few rows left get their exact distance (same code, same order as flat.hip) and are sorted with the
//      sample's results by (distance, row).
// A query whose list overflows (adversarial row order, masses of duplicates), non-finite rows or queries, or
// magnitudes outside the bound's range send the whole call down the exact path: same answer, old speed.
//
// Bound (u = 2^-24, Q = |q|^2 + max |x|^2, |T| <= Q, distances <= 2Q): accumulation of 3 D + 2 terms (x1.q1, x1.q2, x2.q1)
// taken as 2u per term 388 uQ, bf16 splits 64 uQ, the omitted x2.q2 term 32 uQ, |x|^2/2 split and its fp32 rounding 24 uQ:
// 508 uQ on T; the reference's sum
// (D + 4) u 2Q and |q|^2 (D + 1) u Q in distance units, i.e. ~200 uQ in T units.  thr_q is lowered by 2048 uQ = 2^-13 Q.
#include <algorithm>

#include <atomic>
#include "common.h"
#include "kernels.h"

namespace cvtmi {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float kPadBias = -1.0e30f;  // rows past n in the last tile: never candidates (ids >= n are skipped anyway)

// element (row, dim) of the blocked fp32 layout of flat.hip
__device__ __forceinline__ float blocked_at(const float *X, int D, int64_t row, int dim)
{
    return X[(((row >> 6) * (D >> 2) + (dim >> 2)) * 64 + (row & 63)) * 4 + (dim & 3)];
}

// one thread per (tile, lane): lane (li, lk) of tile t owns row 32 t + li, dimensions 16 c + 8 lk .. + 8 of chunk c.
// stats[0] = max |x|^2 (uint bits), stats[1] = number of rows holding a non-finite value, stats[2] = max |x - x1|^2 (x1: the first bf16 term)
__global__ __launch_bounds__(kBlock) void flat_pack_kernel(const float *__restrict__ X, int64_t n, int D, int nch, int l2, int64_t tile0, int64_t ntiles,
                                                           uint4 *__restrict__ pack, uint32_t *__restrict__ bias,
                                                           uint32_t *__restrict__ stats)
{
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x + tile0 * 64;   // tiles [tile0, ntiles)
    if (g >= ntiles * 64) return;
    const int64_t t = g >> 6;
    const int lane = (int)(g & 63), li = lane & 31, lk = lane >> 5;   // nch K steps of 16 dimensions (>= D / 16: zeros beyond D)
    const int64_t row = t * 32 + li;
    bool finite = true;
    for (int c = 0; c < nch; ++c) {
        union { bf16x8 v; uint4 u; } h1, h2;
        // the lane's eight dimensions are two 16-byte pieces of the blocked layout (consecutive lanes = consecutive rows: 512 contiguous bytes per piece)
        float xv[8] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int d0 = 16 * c + 8 * lk + 4 * p;
            if (row < n && d0 < D) {   // (D % 4 == 0: a piece is inside the row or past it)
                const float4 t4 = reinterpret_cast<const float4 *>(X)[((row >> 6) * (int64_t)(D >> 2) + (d0 >> 2)) * 64 + (row & 63)];
                xv[4 * p] = t4.x; xv[4 * p + 1] = t4.y; xv[4 * p + 2] = t4.z; xv[4 * p + 3] = t4.w;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = xv[e];
            finite = finite && (fabsf(v) <= 3.0e38f);
            const __bf16 a = (__bf16)v;
            h1.v[e] = a;
            h2.v[e] = (__bf16)(v - (float)a);
        }
        pack[((t * nch + c) * 2 + 0) * 64 + lane] = h1.u;
        pack[((t * nch + c) * 2 + 1) * 64 + lane] = h2.u;
    }
    if (!finite) atomicAdd(&stats[1], 1u);
    if (lk == 0) {
        float b = kPadBias;
        if (row < n) {
            float s = 0.0f, r2 = 0.0f;   // |x|^2 and |x - x1|^2 (what the first bf16 term leaves out: flat_f32_tfilter.hip, one product), both in dimension order
            const float4 *xr = reinterpret_cast<const float4 *>(X) + (row >> 6) * (int64_t)(D >> 2) * 64 + (row & 63);
            for (int c4 = 0; c4 < (D >> 2); ++c4) {
                const float4 t4 = xr[(int64_t)c4 * 64];
                const float tv[4] = { t4.x, t4.y, t4.z, t4.w };
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = tv[e], r = v - (float)(__bf16)v;
                    s = __fmaf_rn(v, v, s);
                    r2 = __fmaf_rn(r, r, r2);
                }
            }
            b = l2 ? -0.5f * s : 0.0f;
            atomicMax(&stats[0], __float_as_uint(s));  // NaN / inf show up as such: the call takes the exact path
            atomicMax(&stats[2], __float_as_uint(r2));
        }
        const __bf16 a = (__bf16)b, c2 = (__bf16)(b - (float)a);
        bias[row] = (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, c2) << 16);
    }
}

// thr[q]: a row whose reference distance is <= tau_q = sample_d[q][k-1] has T >= thr[q] (see the bound above);
// -inf (everything passes -> overflow -> exact path) whenever the bound does not apply
__global__ __launch_bounds__(kBlock) void flat_thr_kernel(const float *__restrict__ q, int64_t nq, int D, int l2,
                                                          const float *__restrict__ sample_d, int k,
                                                          uint32_t *__restrict__ stats, float *__restrict__ thr, float *__restrict__ margin)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nq) return;
    float qq = 0.0f;
    for (int e = 0; e < D; ++e) qq = __fmaf_rn(q[i * D + e], q[i * D + e], qq);
    if (!(qq <= 3.0e38f)) atomicMax(&stats[2], 0xffffffffu);  // a non-finite query: reads as an overflow, the exact path answers
    const float tau = sample_d[i * k + k - 1];
    const float Q = (qq + __uint_as_float(stats[0])) * 1.001f;
    float t = l2 ? 0.5f * (qq - tau) - Q * 0x1p-13f : (1.0f - tau) - Q * 0x1p-13f - (1.0f + fabsf(tau)) * 0x1p-20f;
    if (!(Q > 0x1p-60f && Q < 0x1p60f) || !(fabsf(t) <= 3.0e38f) || !(fabsf(tau) <= 3.0e38f)) {
        // the bound does not apply: every row would pass.  Flag the overflow directly -- the 32-bit pair counter could wrap
        // (nq * n >= 2^32) and land below pair_cap, which would read as a complete candidate list
        t = -__uint_as_float(0x7f800000u);
        atomicMax(&stats[2], 0xffffffffu);
    }
    thr[i] = t;
    margin[i] = Q * 0x1p-13f + (l2 ? 0.0f : (1.0f + fabsf(tau)) * 0x1p-20f);  // the pairwise cut of flat_finish_kernel
}

constexpr int FF_THREADS = 512;  // 8 waves = 256 queries per workgroup; the row tile is shared through LDS

template <int NCH>
__global__ __launch_bounds__(FF_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void flat_filter_kernel(const float *__restrict__ q, int64_t nq, const uint4 *__restrict__ pack,
                                                                 const uint32_t *__restrict__ bias, const float *__restrict__ thr,
                                                                 int64_t tile_begin, int64_t tile_end, int64_t tiles_per_split,
                                                                 uint32_t pair_cap, uint32_t *__restrict__ pair_cnt, uint4 *__restrict__ pairs,
                                                                 int qblocks)
{
    constexpr int D = 16 * NCH;
    // 1-D grid, XCD-aware: workgroups are dealt to the 8 XCDs round-robin, so b % 8 is the XCD.  All query blocks of one row
    // split sit on the same XCD next to each other in dispatch order: they stream the same row tiles through that XCD's L2
    // (otherwise every query block re-reads its rows from HBM: 16 x 5 GB at the C3 shape).
    const int64_t bi = blockIdx.x >> 3;
    const int split_id = (int)(blockIdx.x & 7) + 8 * (int)(bi / qblocks), qb_id = (int)(bi % qblocks);

    constexpr int TILE = NCH * 2 * 64;                          // uint4 per row tile (both bf16 halves, operand order)
    constexpr int LPT = (TILE + FF_THREADS - 1) / FF_THREADS;   // uint4 per thread and tile
    __shared__ uint4 tile_s[2][TILE];
    // survivors are parked per wave in LDS (slots handed out with ballots: no atomics) and leave in batches: one global
    // atomic and one coalesced write per ~190 survivors instead of a 2 us round trip per survivor in the tile loop
    constexpr int PBUF = 256;
    __shared__ uint4 park_s[FF_THREADS / 64][PBUF];  // (query, row, T bits, -)
    int parked = 0;  // wave-uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int64_t qi = ((int64_t)qb_id * (FF_THREADS / 64) + wave) * 32 + li;
    const int64_t qc = qi < nq ? qi : nq - 1;  // clamped: padding queries compute, never push
    const float *qp = q + qc * D + 8 * lk;
    bf16x8 q1[NCH], q2[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float v[8];
        *reinterpret_cast<float4 *>(&v[0]) = *reinterpret_cast<const float4 *>(qp + 16 * c);
        *reinterpret_cast<float4 *>(&v[4]) = *reinterpret_cast<const float4 *>(qp + 16 * c + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 h = (__bf16)v[e];
            q1[c][e] = h;
            q2[c][e] = (__bf16)(v[e] - (float)h);
        }
    }
    const float th = qi < nq ? thr[qc] : __uint_as_float(0x7f800000u);  // +inf: nothing passes
    const bf16x8 bzero = { 0, 0, 0, 0, 0, 0, 0, 0 };
    bf16x8 ones = bzero;
    ones[0] = lk ? (__bf16)0.0f : (__bf16)1.0f;
    ones[1] = ones[0];
    const int64_t t0 = tile_begin + (int64_t)split_id * tiles_per_split;
    int64_t t1 = t0 + tiles_per_split;
    t1 = t1 < tile_end ? t1 : tile_end;
    if (t0 >= t1) return;
    // tile t+1 travels HBM -> registers while tile t (in LDS) feeds the products; one LDS-only barrier per tile
    uint4 pre[LPT];
    auto fetch = [&](int64_t t) {
        t = t < t1 ? t : t1 - 1;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            int f = tid + i * FF_THREADS;
            f = f < TILE ? f : TILE - 1;
            pre[i] = pack[t * TILE + f];
        }
    };
    fetch(t0);
    for (int64_t t = t0; t < t1; ++t) {
        uint4 *stage = tile_s[(t - t0) & 1];
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            if (tid + i * FF_THREADS < TILE) stage[tid + i * FF_THREADS] = pre[i];
        fetch(t + 1);
        const uint32_t hv = bias[t * 32 + li];
        lds_barrier();
        const uint4 *pa = stage + lane;
        const f32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        f32x16 acc0 = zero, acc1 = zero;  // two chains: dependent products are two apart
        // D[row i of the tile][query j]: A = rows (i = lane & 31, k = 8 * (lane >> 5) ..), B = queries
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            union { uint4 u; bf16x8 v; } a1, a2;
            a1.u = pa[(c * 2 + 0) * 64];
            a2.u = pa[(c * 2 + 1) * 64];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, q1[c], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, q2[c], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, q1[c], acc0, 0, 0, 0);  // x2.q2 <= 2^-18 |x||q| is left out (bound)
        }
        {
            union { uint32_t u[4]; bf16x8 v; } ab = { { lk ? 0u : hv, 0u, 0u, 0u } };
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab.v, ones, acc0, 0, 0, 0);  // + b_x
        }
        uint32_t hit = 0;
        float tv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            tv[e] = acc0[e] + acc1[e];
            hit |= (tv[e] >= th ? 1u : 0u) << e;
        }
        if (qi >= nq) hit = 0;
        while (__any(hit != 0)) {  // ~15 k survivors per query over the whole scan: a couple per wave and tile
            const unsigned long long m = __ballot(hit != 0);
            const int cnt = __popcll(m);
            if (parked + cnt > PBUF) {  // flush (wave-uniform branch)
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(pair_cnt, (uint32_t)parked);
                base = __shfl(base, 0, 64);
                for (int i = lane; i < parked; i += 64)
                    if (base + i < pair_cap) pairs[base + i] = park_s[wave][i];
                parked = 0;
            }
            if (hit) {
                const int e = __ffs((int)hit) - 1;
                hit &= hit - 1;
                float tsel = tv[0];  // tv[e] without dynamic register indexing
#pragma unroll
                for (int j = 1; j < 16; ++j) tsel = e == j ? tv[j] : tsel;
                park_s[wave][parked + __popcll(m & ((1ull << lane) - 1))] =
                    make_uint4((uint32_t)qi, (uint32_t)(t * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk), __float_as_uint(tsel), 0u);
            }
            parked += cnt;
        }
    }
    if (parked) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(pair_cnt, (uint32_t)parked);
        base = __shfl(base, 0, 64);
        for (int i = lane; i < parked; i += 64)
            if (base + i < pair_cap) pairs[base + i] = park_s[wave][i];
    }
}

// survivors -> per-query lists (approximate score, row); the list order is arbitrary
__global__ __launch_bounds__(kBlock) void flat_scatter_kernel(const uint32_t *__restrict__ pair_cnt, uint32_t pair_cap,
                                                              const uint4 *__restrict__ pairs, int cap, uint32_t *__restrict__ cand_cnt,
                                                              float *__restrict__ cand_t, int32_t *__restrict__ cand_row,
                                                              uint32_t *__restrict__ overflow)
{
    uint32_t total = *pair_cnt;
    if (total > pair_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(overflow, 0xffffffffu);
        total = pair_cap;
    }
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < total; p += gridDim.x * kBlock) {
        const uint4 pr = pairs[p];
        const uint32_t slot = atomicAdd(&cand_cnt[pr.x], 1u);
        if (slot < (uint32_t)cap) {
            cand_t[(int64_t)pr.x * cap + slot] = __uint_as_float(pr.z);
            cand_row[(int64_t)pr.x * cap + slot] = (int32_t)pr.y;
        } else {
            atomicMax(overflow, slot + 1u);
        }
    }
}

// in-place bitonic sort of s[0 .. n2) (n2 a power of two) by the whole workgroup
template <typename T, bool DESC>
__device__ __forceinline__ void block_bitonic(T *s, int n2)
{
    for (int k2 = 2; k2 <= n2; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += kBlock) {
                const int p = i ^ j;
                if (p > i) {
                    const T a = s[i], b = s[p];
                    const bool up = ((i & k2) == 0) != DESC;
                    if (up ? a > b : a < b) { s[i] = b; s[p] = a; }
                }
            }
            __syncthreads();
        }
}

constexpr int FIN_CAND = 4096;  // survivors per query the finishing kernel takes (>= the caller's cap)
constexpr int FIN_KEEP = 896;   // survivors that get an exact distance

// One workgroup per query.  Second cut, still on approximate scores: with theta = the k-th largest T among the survivors,
// a survivor with T < theta - 2^-13 Q is beaten -- in the reference's own arithmetic -- by k others (pairwise: 2 x 508 uQ
// of filter error + 2 x ~200 uQ of reference rounding < 2048 uQ), so it is out.  What is left (k and a few) gets its exact
// distance in the reference's summation order (dist_f32.h) and meets the sample's k results in a (distance, row) sort.
template <bool IP, int LANES>
__global__ __launch_bounds__(kBlock) void flat_finish_kernel(const float *__restrict__ X, int64_t n, int D, const float *__restrict__ q,
                                                             const uint32_t *__restrict__ cand_cnt, const float *__restrict__ cand_t,
                                                             const int32_t *__restrict__ cand_row, int cap, int k,
                                                             const float *__restrict__ margin, const float *__restrict__ sample_d,
                                                             const int64_t *__restrict__ sample_i, float *__restrict__ out_d,
                                                             int64_t *__restrict__ out_i, uint32_t *__restrict__ overflow)
{
    __shared__ uint32_t tk_s[FIN_CAND];                    // order-preserving keys of T, sorted descending
    __shared__ unsigned long long fin_s[1024];             // (distance key << 32 | row), sorted ascending
    __shared__ int kept_s[FIN_KEEP];
    __shared__ int nkeep_s;
    const int64_t qi = blockIdx.x;
    const int tid = threadIdx.x;
    uint32_t c = cand_cnt[qi];
    if (tid == 0) atomicMax(overflow, c);  // the largest list of the call (> cap: ran over)
    c = c < (uint32_t)cap ? c : (uint32_t)cap;
    const float *ct = cand_t + qi * cap;
    const int32_t *cr = cand_row + qi * cap;
    int n2 = 1;
    while (n2 < (int)c) n2 <<= 1;
    if (tid == 0) nkeep_s = 0;
    float theta = -__uint_as_float(0x7f800000u);
    if ((int)c > k) {  // workgroup-uniform
        for (int i = tid; i < n2; i += kBlock) tk_s[i] = i < (int)c ? f32_key(ct[i]) : 0u;
        __syncthreads();
        block_bitonic<uint32_t, true>(tk_s, n2);
        theta = key_f32(tk_s[k - 1]);
    }
    __syncthreads();
    const float cut = theta - margin[qi];
    for (int i = tid; i < (int)c; i += kBlock) {
        if (ct[i] >= cut && cr[i] < n) {
            const int slot = atomicAdd(&nkeep_s, 1);
            if (slot < FIN_KEEP) kept_s[slot] = cr[i];
        }
    }
    __syncthreads();
    int nk = nkeep_s;
    if (nk > FIN_KEEP) {  // masses of near ties: the exact path answers the call
        if (tid == 0) atomicMax(overflow, 0xffffffffu);
        nk = FIN_KEEP;
    }
    const int total = nk + k;  // <= 1024
    int m2 = 1;
    while (m2 < total) m2 <<= 1;
    for (int i = tid; i < m2; i += kBlock) {
        unsigned long long e = ~0ull;
        if (i < nk) {
            const int64_t row = kept_s[i];
            float acc[LANES];
#pragma unroll
            for (int l = 0; l < LANES; ++l) acc[l] = 0.0f;
            const float4 *qv = reinterpret_cast<const float4 *>(q + qi * D);
            // blocked layout: float4 c of a row sits at ((row >> 6) * (D / 4) + c) * 64 + (row & 63)
            const float4 *xr = reinterpret_cast<const float4 *>(X) + (row >> 6) * (int64_t)(D >> 2) * 64 + (row & 63);
            for (int i4 = 0; i4 < D / 4; i4 += LANES / 4) {
#pragma unroll
                for (int g = 0; g < LANES / 4; ++g) {
                    const float4 xv = xr[(int64_t)(i4 + g) * 64], qq = qv[i4 + g];
                    const float xs[4] = { xv.x, xv.y, xv.z, xv.w }, qs[4] = { qq.x, qq.y, qq.z, qq.w };
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        if constexpr (IP) {
                            acc[4 * g + l] = __fadd_rn(acc[4 * g + l], __fmul_rn(qs[l], xs[l]));
                        } else {
                            const float t = __fsub_rn(qs[l], xs[l]);
                            acc[4 * g + l] = __fadd_rn(acc[4 * g + l], __fmul_rn(t, t));
                        }
                    }
                }
            }
            float sum = acc[0];
#pragma unroll
            for (int l = 1; l < LANES; ++l) sum = __fadd_rn(sum, acc[l]);
            const float d = IP ? __fsub_rn(1.0f, sum) : sum;
            e = ((unsigned long long)f32_key(d) << 32) | (uint32_t)row;
        } else if (i < total) {
            const int64_t id = sample_i[qi * k + (i - nk)];
            if (id >= 0) e = ((unsigned long long)f32_key(sample_d[qi * k + (i - nk)]) << 32) | (uint32_t)id;
        }
        fin_s[i] = e;
    }
    __syncthreads();
    block_bitonic<unsigned long long, false>(fin_s, m2);
    for (int i = tid; i < k; i += kBlock) {
        const unsigned long long e = fin_s[i];
        const bool ok = e != ~0ull;
        out_d[qi * k + i] = ok ? key_f32((uint32_t)(e >> 32)) : __uint_as_float(0x7f800000u);
        out_i[qi * k + i] = ok ? (int64_t)(uint32_t)e : -1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// uint8 L2 (L2SpaceI, space_l2.h:186-245) through the same pipeline.  Here the matrix cores (v_mfma_i32_32x32x32_i8 on
// x - 128) give the EXACT integer distance, so there is no bound to prove and no second cut: the k-th best of an exactly
// searched leading sample is the threshold, the filter kernel streams the remaining rows with thresholds in registers (no
// selection state, no threshold traffic between workgroups), survivors are sorted with the sample's results.
// ------------------------------------------------------------------------------------------------------------------
using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;

// [tile][s][lane] 16 bytes: row 32 t + (lane & 31), dimensions 32 s + 16 (lane >> 5) .. + 16, each byte ^ 0x80 (x - 128 as int8)
__global__ __launch_bounds__(kBlock) void flat_u8_pack_kernel(const uint8_t *__restrict__ X, int64_t n, int D, int64_t tile0, int64_t ntiles,
                                                              uint4 *__restrict__ pack)
{
    const int ks = D / 32;
    const int64_t g = tile0 * ks * 64 + (int64_t)blockIdx.x * kBlock + threadIdx.x;   // tiles [tile0, ntiles)
    if (g >= ntiles * ks * 64) return;
    const int lane = (int)(g & 63);
    const int64_t ts = g >> 6, t = ts / ks;
    const int s_ = (int)(ts - t * ks);
    const int64_t row = t * 32 + (lane & 31);
    uint4 v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);  // rows past n: x = 0 (never reported)
    if (row < n) v = *reinterpret_cast<const uint4 *>(X + row * D + 32 * s_ + 16 * (lane >> 5));
    v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
    pack[g] = v;
}

// A workgroup of NT / 64 waves scores 32 QS queries per wave against its whole row range.  QS = 2: every 1 KB row operand
// read from LDS feeds two matrix instructions (LDS delivers ~1 KB per 8 cycles and CU, which is what four SIMDs of
// v_mfma_i32_32x32x32_i8 consume with one query set per wave: that, not HBM, bounded the one-set kernels at ~35 % of the
// matrix pipe), and the rows are streamed nq / (16 QS NT / 32) times in all instead of twice as often.
template <int KS, int NT, int QS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(QS == 1 ? 4 : 2, QS == 1 ? 4 : 2))) void flat_u8_filter_kernel(
    const uint8_t *__restrict__ q, int64_t nq, int D, const uint4 *__restrict__ pack, const int32_t *__restrict__ norms, int64_t n,
    const float *__restrict__ sample_d, int k, int64_t tile_begin, int64_t tile_end, int64_t tiles_per_split, uint32_t pair_cap,
    uint32_t *__restrict__ pair_cnt, uint4 *__restrict__ pairs, int qblocks)
{
    constexpr int TILE = KS * 64;             // uint4 per row tile at most (D = 32 KS)
    constexpr int LPT = (TILE + NT - 1) / NT;
    constexpr int PBUF = 256;
    constexpr int QPB = (NT / 64) * 32 * QS;  // queries per workgroup
    // 1-D grid, XCD-aware: workgroups are dealt to the 8 XCDs round-robin, so b % 8 is the XCD.  All query blocks of one row
    // split sit on the same XCD next to each other in dispatch order and share its L2.
    const int64_t bi = blockIdx.x >> 3;
    const int split_id = (int)(blockIdx.x & 7) + 8 * (int)(bi / qblocks), qb_id = (int)(bi % qblocks);
    const int tile_len = (D / 32) * 64;       // uint4 per row tile of this index (<= TILE)
    __shared__ uint4 tile_s[2][TILE];
    __shared__ uint4 park_s[NT / 64][PBUF];   // (query, row, distance, -)
    __shared__ int qq_s[QPB], thr_s[QPB];     // per query of the workgroup: |q'|^2, tau - |q'|^2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lj = lane & 31, lk = lane >> 5;
    const int nks = D / 32;
    i32x4 qreg[QS][KS];
#pragma unroll
    for (int u = 0; u < QS; ++u) {
        const int ql = (wave * QS + u) * 32 + lj;
        const int64_t qi = (int64_t)qb_id * QPB + ql;
        const int64_t qc = qi < nq ? qi : nq - 1;
        const uint8_t *qp = q + qc * D;
        int qq = 0;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) qreg[u][s_] = *reinterpret_cast<const i32x4 *>(qp + 32 * (s_ < nks ? s_ : 0) + 16 * lk);
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            qreg[u][s_] ^= (int)0x80808080;
            if (s_ >= nks) qreg[u][s_] = i32x4{ 0, 0, 0, 0 };
#pragma unroll
            for (int c = 0; c < 4; ++c) qq = __builtin_amdgcn_sdot4(qreg[u][s_][c], qreg[u][s_][c], qq, false);
        }
        qq += __shfl_xor(qq, 32, 64);
        if (lk == 0) {
            const int tau = (int)__float_as_uint(sample_d[qc * k + k - 1]);  // integer distance bits (0x7f800000 = none: everything passes)
            qq_s[ql] = qq;
            thr_s[ql] = qi < nq ? tau - qq : (int)0x80000000;               // padding queries: nothing passes
        }
    }
    __syncthreads();
    // the 16 queries of set u this lane scores: q_e = (e & 3) + 8 (e >> 2) + 4 lk  (MFMA C layout: D[query][row], lane = row)
    // (with one query set the 16 thresholds sit in registers; with two they are read from LDS per tile -- two distinct
    //  addresses per wave, broadcast -- because 128 query registers leave no room)
    int thr[QS == 1 ? 16 : 1];
    if constexpr (QS == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) thr[e] = thr_s[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk];
    }
    const int64_t t0 = tile_begin + (int64_t)split_id * tiles_per_split;
    int64_t t1 = t0 + tiles_per_split;
    t1 = t1 < tile_end ? t1 : tile_end;
    if (t0 >= t1) return;
    int parked = 0;  // wave-uniform
    uint4 pre[LPT];
    int xx_next = 0;
    auto fetch = [&](int64_t t) {
        t = t < t1 ? t : t1 - 1;
        int64_t row = t * 32 + lj;
        row = row < n ? row : n - 1;
        xx_next = norms[row];
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            int f = tid + i * NT;
            f = f < tile_len ? f : tile_len - 1;
            pre[i] = pack[t * tile_len + f];
        }
    };
    fetch(t0);
    for (int64_t t = t0; t < t1; ++t) {
        uint4 *stage = tile_s[(t - t0) & 1];
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            if (tid + i * NT < tile_len) stage[tid + i * NT] = pre[i];
        const int xx = xx_next;
        fetch(t + 1);
        lds_barrier();
        const i32x4 *pb = reinterpret_cast<const i32x4 *>(stage) + lane;
        i32x16 acc[QS];
#pragma unroll
        for (int u = 0; u < QS; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[u][e] = 0;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_)
            if (s_ < nks) {
                const i32x4 bv = pb[s_ * 64];
#pragma unroll
                for (int u = 0; u < QS; ++u) acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qreg[u][s_], bv, acc[u], 0, 0, 0);
            }
        const int64_t row = t * 32 + lj;
#pragma unroll
        for (int u = 0; u < QS; ++u) {
            uint32_t hit = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int th = QS == 1 ? thr[QS == 1 ? e : 0] : thr_s[(wave * QS + u) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk];
                hit |= (xx - 2 * acc[u][e] <= th ? 1u : 0u) << e;  // distance - |q'|^2, exact
            }
            if (row >= n) hit = 0;
            while (__any(hit != 0)) {
                const unsigned long long m = __ballot(hit != 0);
                const int cnt = __popcll(m);
                if (parked + cnt > PBUF) {  // flush (wave-uniform branch)
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(pair_cnt, (uint32_t)parked);
                    base = __shfl(base, 0, 64);
                    for (int i = lane; i < parked; i += 64)
                        if (base + i < pair_cap) pairs[base + i] = park_s[wave][i];
                    parked = 0;
                }
                if (hit) {
                    const int e = __ffs((int)hit) - 1;
                    hit &= hit - 1;
                    int asel = acc[u][0];
#pragma unroll
                    for (int j = 1; j < 16; ++j) asel = e == j ? acc[u][j] : asel;
                    const int ql = (wave * QS + u) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                    park_s[wave][parked + __popcll(m & ((1ull << lane) - 1))] =
                        make_uint4((uint32_t)(qb_id * QPB + ql), (uint32_t)row, (uint32_t)(xx - 2 * asel + qq_s[ql]), 0u);
                }
                parked += cnt;
            }
        }
    }
    if (parked) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(pair_cnt, (uint32_t)parked);
        base = __shfl(base, 0, 64);
        for (int i = lane; i < parked; i += 64)
            if (base + i < pair_cap) pairs[base + i] = park_s[wave][i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same filter as a software-pipelined GEMM (flat_u8_gfilter_kernel): 8 waves, 32 queries per wave in registers (A side),
// row tiles streamed HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: the packed copy IS the operand order, one 1 KB piece
// per wave-instruction, lane-linear), no staging registers and no LDS store pass.  What the kernels above leave on the table
// (v_mfma_i32_32x32x32_i8 busy a third of the time) is structure, not bandwidth:
//   * one barrier per 32-row tile and 16 matrix instructions between barriers          -> G tiles (64 rows) per barrier;
//   * every matrix instruction waited for the LDS read issued right in front of it, and for its predecessor's result (all 16
//     K steps of a tile accumulate into ONE register set)                              -> the G tiles of a group are G
//     independent accumulator chains fed alternately, sharing each query operand;
//   * the register prefetch ring (32 VGPRs) plus its LDS stores                         -> DMA into a ring of 3 groups, counted
//     s_waitcnt vmcnt(OPS): a group's data is requested two barriers before it is read.
// Synchronisation (one s_barrier per group): at the top of iteration g a wave waits until at most the OPS requests of group g + 1
// are outstanding (its own share of group g has landed: requests retire in order), then the barrier (everyone's share of g has
// landed, everyone has finished reading group g - 1), then it requests group g + 2 into the slot group g - 1 occupied, then reads
// group g.  The DMA is issued from inline asm, so hipcc neither counts it nor drains it (it waits vmcnt(0) in front of every
// LDS read when it sees the builtin); ordinary memory operations in the loop are confined to the rare survivor flush.
// Row norms |x'|^2 travel the same way (one 4-byte-per-lane piece per group, requested by every wave: same request count in
// every wave, same bytes to the same place).
// ------------------------------------------------------------------------------------------------------------------
// NT: non-temporal hint for rows a CU reads once per launch (MI355X_MICROARCH.md "nt-weights")
template <bool NT = false>
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// measured (10 M x 512-d): the streaming kernel, whose rows are read once per pass, 0.908 -> 0.83 ms for one query (6.2 TB/s end
// to end), 1.27 -> 1.22 ms for 128; the filter kernels, whose query blocks share each row tile through L2, lose 6-10 % with it
constexpr bool U8_NT_GF = false, U8_NT_MS = true;
__device__ __forceinline__ void glds4(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// NW waves (32 queries each) per workgroup, G row tiles per barrier.  <8, 2>: one workgroup per CU, its two waves per SIMD move in
// lock step (same barrier).  <4, 1>: two independent workgroups per CU, one wave per SIMD each -- the pair on a SIMD drifts apart,
// so one wave's matrix instructions run beside the other's reads / waits / barrier.
template <int KS, int NW, int G, int NB>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void flat_u8_gfilter_kernel(
    const uint8_t *__restrict__ q, int64_t nq, const uint4 *__restrict__ pack, const int32_t *__restrict__ norms, int64_t n,
    const float *__restrict__ sample_d, int k, int64_t tile_begin, int64_t tile_end, int64_t tiles_per_split, uint32_t pair_cap,
    uint32_t *__restrict__ pair_cnt, uint4 *__restrict__ pairs, int qblocks, int dbg)
{
#ifndef CVTMI_GF_DBG
    dbg = 0;   // the experiment switches exist in -DCVTMI_GF_DBG builds only
#endif
    constexpr int D = 32 * KS;
    constexpr int PPW = G * KS / NW;          // 1 KB pieces each wave requests per group
    constexpr int OPS = PPW + 1;              // + the norms piece
    constexpr int PBUF = 128;
    constexpr int QPB = 32 * NW;
    constexpr int C = G == 1 ? 2 : G;         // accumulator chains: the G tiles, or the even / odd K steps of a single tile
    static_assert(G * KS % NW == 0 && (G == 1 || G == 2 || G == 4), "pieces per group must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) uint4 gf_ring[];   // [NB][G][KS * 64]
    __shared__ int xx_s[NB][G * 32 < 64 ? 64 : G * 32];
    __shared__ uint4 park_s[NW][PBUF];         // (query, row, distance, -)
    __shared__ int qq_s[QPB], thr_s[QPB];
    const int tid = threadIdx.x, lane = tid & 63, lj = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t bi = blockIdx.x >> 3;
    const int split_id = (int)(blockIdx.x & 7) + 8 * (int)(bi / qblocks), qb_id = (int)(bi % qblocks);
    // ---- this wave's 32 queries: operands in registers, |q'|^2 and tau - |q'|^2 in LDS ----
    i32x4 qreg[KS];
    {
        const int ql = wave * 32 + lj;
        const int64_t qi = (int64_t)qb_id * QPB + ql;
        const int64_t qc = qi < nq ? qi : nq - 1;
        const uint8_t *qp = q + qc * D;
        int qq = 0;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) qreg[s_] = *reinterpret_cast<const i32x4 *>(qp + 32 * s_ + 16 * lk);
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            qreg[s_] ^= (int)0x80808080;
#pragma unroll
            for (int c = 0; c < 4; ++c) qq = __builtin_amdgcn_sdot4(qreg[s_][c], qreg[s_][c], qq, false);
        }
        qq += __shfl_xor(qq, 32, 64);
        if (lk == 0) {
            const int tau = (int)__float_as_uint(sample_d[qc * k + k - 1]);  // integer distance bits (0x7f800000 = none: everything passes)
            qq_s[ql] = qq;
            thr_s[ql] = qi < nq ? tau - qq : (int)0x80000000;               // padding queries: nothing passes
        }
    }
    __syncthreads();   // (also drains the ordinary loads above: from here on the loop's only memory traffic is the DMA)
    // Threshold test folded into the accumulation.  A (row, query) pair survives when  xx - 2 dot <= thr  (xx = |x'|^2, thr = tau - |q'|^2).
    // With thr = 2 a + b and xx = 2 c + d (b, d in {0, 1}) that is  dot + a >= c + [d = 1 and b = 0], so  dot + a >= c  is a superset
    // test -- and dot + a is what the matrix unit delivers when a = thr >> 1 is the initial accumulator.  A tile then costs 8 v_max3
    // and one compare per lane (max over the lane's 16 queries against c) instead of four instructions per result; the rare lane
    // that passes re-tests its 16 results exactly.
    i32x16 thrh;
#pragma unroll
    for (int e = 0; e < 16; ++e) thrh[e] = thr_s[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk] >> 1;
    const int64_t t0 = tile_begin + (int64_t)split_id * tiles_per_split;
    int64_t t1 = t0 + tiles_per_split;
    t1 = t1 < tile_end ? t1 : tile_end;
    if (t0 >= t1) return;   // workgroup-uniform
    const int64_t n_groups = (t1 - t0 + G - 1) / G;
    const uint32_t ring_b = (uint32_t)(uintptr_t)gf_ring, xx_b = (uint32_t)(uintptr_t)&xx_s[0][0];
    // requests of group g: pieces p = wave * PPW + i (tile g G + p / KS, K step p % KS); tiles past the split are clamped duplicates
    auto request = [&](int64_t g, int slot) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave * PPW + i;
            int64_t t = t0 + g * G + p / KS;
            t = t < t1 ? t : t1 - 1;
            glds16<U8_NT_GF>(pack + (t * KS + p % KS) * 64 + lane, ring_b + (uint32_t)(((slot * G * KS) + p) * 64 * 16));
        }
        int64_t t = t0 + g * G + (lane >> 5);   // norms of the group's first two tiles per piece; G = 4 takes two pieces' worth in one
        int64_t row;
        if (G <= 2) {
            if (G == 1) t = t0 + g;
            t = t < t1 ? t : t1 - 1;
            row = t * 32 + lj;
            row = row < n ? row : n - 1;
            glds4(norms + row, xx_b + (uint32_t)(slot * (G * 32 < 64 ? 64 : G * 32) * 4));
        } else {  // G = 4: lanes cover tiles 0..1 here, tiles 2..3 are fetched by the second half of the waves' pieces
            const int half = wave & 1;
            t += 2 * half;
            t = t < t1 ? t : t1 - 1;
            row = t * 32 + lj;
            row = row < n ? row : n - 1;
            glds4(norms + row, xx_b + (uint32_t)((slot * (G * 32 < 64 ? 64 : G * 32) + half * 64) * 4));
        }
    };
    int parked = 0;  // wave-uniform
    // survivors of one tile (rare: ~k ln(n / sample) rows per query in all).  accq = dot + (thr >> 1) per result.
    auto collect = [&](const i32x16 &accq, int xx, int64_t row, bool cand) {
        uint32_t hit = 0;
        if (cand && row < n) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int th = thr_s[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk];
                const int dot = accq[e] - (th >> 1);
                hit |= (xx - 2 * dot <= th ? 1u : 0u) << e;   // distance - |q'|^2 <= tau - |q'|^2, exact
            }
        }
        while (__any(hit != 0)) {
            const unsigned long long m = __ballot(hit != 0);
            const int cnt = __popcll(m);
            if (parked + cnt > PBUF) {  // flush (wave-uniform branch)
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(pair_cnt, (uint32_t)parked);
                base = __shfl(base, 0, 64);
                for (int i = lane; i < parked; i += 64)
                    if (base + i < pair_cap) pairs[base + i] = park_s[wave][i];
                parked = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ordinary traffic must not sit between counted DMA requests
            }
            if (hit) {
                const int e = __ffs((int)hit) - 1;
                hit &= hit - 1;
                int asel = accq[0];
#pragma unroll
                for (int j = 1; j < 16; ++j) asel = e == j ? accq[j] : asel;
                const int ql = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                const int dot = asel - (thr_s[ql] >> 1);
                park_s[wave][parked + __popcll(m & ((1ull << lane) - 1))] =
                    make_uint4((uint32_t)(qb_id * QPB + ql), (uint32_t)row, (uint32_t)(xx - 2 * dot + qq_s[ql]), 0u);
            }
            parked += cnt;
        }
    };
    // NB - 1 groups of requests go out up front, one more per iteration: the counted waits below rely on exactly that
#pragma unroll
    for (int p = 0; p < NB - 1; ++p) request(p < n_groups ? p : n_groups - 1, p);
    // The reduction of group g - 1's results (8 v_max3 per tile) is issued between the matrix instructions of group g (second
    // accumulator set): the matrix pipe of a SIMD is shared by two waves, what a wave does between its own matrix instructions is free.
    constexpr int PD = KS >= 8 ? 4 : (KS >= 4 ? 2 : 1);   // K steps of operand reads in flight ahead of the matrix instructions
    constexpr int MPS = (8 * G + KS - 1) / KS;            // v_max3 per K step
    i32x16 prev[G];
    int xx_prev[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        xx_prev[u] = 0x7fffffff;
        prev[u] = thrh;
    }
    for (int64_t g = 0; g < n_groups; ++g) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * OPS) : "memory");  // own share of group g has landed (later groups may be in flight)
        __builtin_amdgcn_s_barrier();                                // everyone's share has; everyone is done with group g - 1
        if (!(dbg & 1)) request(g + NB - 1 < n_groups ? g + NB - 1 : n_groups - 1, (int)((g + NB - 1) % NB));  // into the slot group g - 1 occupied (past the end: a duplicate nobody reads)
        const int slot = (int)(g % NB);
        const i32x4 *pb = reinterpret_cast<const i32x4 *>(gf_ring) + (size_t)slot * G * KS * 64 + lane;
        int xx_cur[G];
#pragma unroll
        for (int u = 0; u < G; ++u) xx_cur[u] = xx_s[slot][u * 32 + lj];
        i32x16 acc[C];
        int mx[G];
#pragma unroll
        for (int u = 0; u < G; ++u) mx[u] = (int)0x80000000;
        i32x4 bv[G][KS];
#pragma unroll
        for (int s_ = 0; s_ < PD; ++s_)
#pragma unroll
            for (int u = 0; u < G; ++u) bv[u][s_] = pb[(u * KS + s_) * 64];
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            if (s_ + PD < KS) {
#pragma unroll
                for (int u = 0; u < G; ++u) bv[u][s_ + PD] = pb[(u * KS + s_ + PD) * 64];
            }
            if constexpr (G == 1) {   // one tile: even and odd K steps are the two chains (summed below)
                const i32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
                acc[s_ & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qreg[s_], bv[0][s_], s_ == 0 ? thrh : (s_ == 1 ? zero : acc[s_ & 1]), 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < G; ++u)   // G independent chains: consecutive matrix instructions never wait for each other
                    acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qreg[s_], bv[u][s_], s_ == 0 ? thrh : acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < MPS; ++j) {
                const int idx = s_ * MPS + j;   // max3 number idx of the 8 G: tile idx / 8, results 2 (idx % 8), + 1
                if (idx < 8 * G) {
                    const int u2 = idx >> 3, e = (idx & 7) * 2;
                    const int m2 = prev[u2][e] > prev[u2][e + 1] ? prev[u2][e] : prev[u2][e + 1];
                    mx[u2] = mx[u2] > m2 ? mx[u2] : m2;
                }
            }
        }
        // pin the issue order: PD steps of operand reads up front, then per K step {G reads, G matrix instructions, the max3s}
        __builtin_amdgcn_sched_group_barrier(0x100, G * PD, 0);
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            if (s_ + PD < KS) __builtin_amdgcn_sched_group_barrier(0x100, G, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, MPS, 0);
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int64_t t = t0 + (g - 1) * G + u;
            const bool cand = mx[u] >= (xx_prev[u] >> 1) && t < t1 && !(dbg & 2);     // xx_prev = INT_MAX before the first group: nothing passes
            if (__any(cand)) collect(prev[u], xx_prev[u], t * 32 + lj, cand);
            if constexpr (G == 1) {
#pragma unroll
                for (int e = 0; e < 16; ++e) prev[0][e] = acc[0][e] + acc[1][e];
            } else {
                prev[u] = acc[u];
            }
            xx_prev[u] = xx_cur[u];
        }
    }
    {   // the last group's tests
#pragma unroll
        for (int u = 0; u < G; ++u) {
            int m = prev[u][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) m = m > prev[u][e] ? m : prev[u][e];
            const int64_t t = t0 + (n_groups - 1) * G + u;
            const bool cand = m >= (xx_prev[u] >> 1) && t < t1;
            if (__any(cand)) collect(prev[u], xx_prev[u], t * 32 + lj, cand);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (parked) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(pair_cnt, (uint32_t)parked);
        base = __shfl(base, 0, 64);
        for (int i = lane; i < parked; i += 64)
            if (base + i < pair_cap) pairs[base + i] = park_s[wave][i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The GEMM-shaped form of the same filter (flat_u8_gfilter_wide_kernel): ONE wave per SIMD that owns the SIMD's whole register file.
// The 8-wave kernel above left the matrix pipe half idle for structural reasons: every wave reads every row tile from LDS (8 x 16 KB
// per 32-row tile = half of the LDS bandwidth a tile's matrix time offers, arriving in lock-step bursts after the shared barrier), and
// a tile gives a wave only 16 matrix instructions between barriers.  Here a wave keeps AQ x 32 queries in registers (AQ = 4: 256 of its
// 512 registers), so one 16 KB tile read from LDS feeds AQ x KS matrix instructions in AQ independent chains (64 per barrier at
// D = 512), LDS traffic per matrix instruction drops 4x, L2 -> LDS traffic 2x (512 queries per workgroup), and nothing on the SIMD
// competes with the wave for the matrix pipe.  Synchronisation, DMA ring, threshold folding and the survivor path are the 8-wave
// kernel's; the thr >> 1 start values come from LDS straight into the accumulators at the top of each tile (no registers held), and
// the two accumulator sets swap roles from tile to tile (the previous tile's results are reduced while this tile accumulates).
// ------------------------------------------------------------------------------------------------------------------
template <int KS, int AQ, int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flat_u8_gfilter_wide_kernel(
    const uint8_t *__restrict__ q, int64_t nq, const uint4 *__restrict__ pack, const int32_t *__restrict__ norms, int64_t n,
    const float *__restrict__ sample_d, int k, int64_t tile_begin, int64_t tile_end, int64_t tiles_per_split, uint32_t pair_cap,
    uint32_t *__restrict__ pair_cnt, uint4 *__restrict__ pairs, int qblocks, int dbg)
{
#ifndef CVTMI_GF_DBG
    dbg = 0;
#endif
    constexpr int D = 32 * KS, NW = 4;
    constexpr int PPW = KS / NW;              // 1 KB pieces each wave requests per tile
    constexpr int OPS = PPW + 1;              // + the norms piece
    constexpr int PBUF = 128;
    constexpr int QPB = 32 * NW * AQ;
    static_assert(KS % NW == 0, "pieces per tile must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) uint4 gf_ring[];   // [NB][KS * 64]
    __shared__ int xx_s[NB][64];
    __shared__ uint4 park_s[NW][PBUF];         // (query, row, distance, -)
    __shared__ int qq_s[QPB], thr_s[QPB];
    __shared__ __attribute__((aligned(16))) int thh_s[NW][AQ][2][16];   // thr >> 1 in accumulator order, per (wave, query set, lane half)
    const int tid = threadIdx.x, lane = tid & 63, lj = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t bi = blockIdx.x >> 3;
    const int split_id = (int)(blockIdx.x & 7) + 8 * (int)(bi / qblocks), qb_id = (int)(bi % qblocks);
    i32x4 qreg[AQ][KS];
#pragma unroll
    for (int a = 0; a < AQ; ++a) {
        const int ql = (wave * AQ + a) * 32 + lj;
        const int64_t qi = (int64_t)qb_id * QPB + ql;
        const int64_t qc = qi < nq ? qi : nq - 1;
        const uint8_t *qp = q + qc * D;
        int qq = 0;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) qreg[a][s_] = *reinterpret_cast<const i32x4 *>(qp + 32 * s_ + 16 * lk);
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            qreg[a][s_] ^= (int)0x80808080;
#pragma unroll
            for (int c = 0; c < 4; ++c) qq = __builtin_amdgcn_sdot4(qreg[a][s_][c], qreg[a][s_][c], qq, false);
        }
        qq += __shfl_xor(qq, 32, 64);
        if (lk == 0) {
            const int tau = (int)__float_as_uint(sample_d[qc * k + k - 1]);  // integer distance bits (0x7f800000 = none: everything passes)
            qq_s[ql] = qq;
            thr_s[ql] = qi < nq ? tau - qq : (int)0x80000000;               // padding queries: nothing passes
        }
    }
    __syncthreads();
    if (lj < 16) {
#pragma unroll
        for (int a = 0; a < AQ; ++a) thh_s[wave][a][lk][lj] = thr_s[(wave * AQ + a) * 32 + (lj & 3) + 8 * (lj >> 2) + 4 * lk] >> 1;
    }
    __syncthreads();   // (also drains the ordinary loads above: from here on the loop's only memory traffic is the DMA)
    const int64_t t0 = tile_begin + (int64_t)split_id * tiles_per_split;
    int64_t t1 = t0 + tiles_per_split;
    t1 = t1 < tile_end ? t1 : tile_end;
    if (t0 >= t1) return;   // workgroup-uniform
    const int64_t n_groups = t1 - t0;
    const uint32_t ring_b = (uint32_t)(uintptr_t)gf_ring, xx_b = (uint32_t)(uintptr_t)&xx_s[0][0];
    auto request = [&](int64_t g, int slot) {
        int64_t t = t0 + g;
        t = t < t1 ? t : t1 - 1;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave * PPW + i;
            glds16<U8_NT_GF>(pack + (t * KS + p) * 64 + lane, ring_b + (uint32_t)(((slot * KS) + p) * 64 * 16));
        }
        int64_t row = t * 32 + lj;
        row = row < n ? row : n - 1;
        glds4(norms + row, xx_b + (uint32_t)(slot * 64 * 4));
    };
    int parked = 0;  // wave-uniform
    auto collect = [&](const i32x16 &accq, int a, int xx, int64_t row, bool cand) {
        uint32_t hit = 0;
        const int qbase = (wave * AQ + a) * 32;
        if (cand && row < n) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int th = thr_s[qbase + (e & 3) + 8 * (e >> 2) + 4 * lk];
                const int dot = accq[e] - (th >> 1);
                hit |= (xx - 2 * dot <= th ? 1u : 0u) << e;   // distance - |q'|^2 <= tau - |q'|^2, exact
            }
        }
        while (__any(hit != 0)) {
            const unsigned long long m = __ballot(hit != 0);
            const int cnt = __popcll(m);
            if (parked + cnt > PBUF) {  // flush (wave-uniform branch)
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(pair_cnt, (uint32_t)parked);
                base = __shfl(base, 0, 64);
                for (int i = lane; i < parked; i += 64)
                    if (base + i < pair_cap) pairs[base + i] = park_s[wave][i];
                parked = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ordinary traffic must not sit between counted DMA requests
            }
            if (hit) {
                const int e = __ffs((int)hit) - 1;
                hit &= hit - 1;
                int asel = accq[0];
#pragma unroll
                for (int j = 1; j < 16; ++j) asel = e == j ? accq[j] : asel;
                const int ql = qbase + (e & 3) + 8 * (e >> 2) + 4 * lk;
                const int dot = asel - (thr_s[ql] >> 1);
                park_s[wave][parked + __popcll(m & ((1ull << lane) - 1))] =
                    make_uint4((uint32_t)(qb_id * QPB + ql), (uint32_t)row, (uint32_t)(xx - 2 * dot + qq_s[ql]), 0u);
            }
            parked += cnt;
        }
    };
#pragma unroll
    for (int p = 0; p < NB - 1; ++p) request(p < n_groups ? p : n_groups - 1, p);
    constexpr int PD = KS >= 8 ? 4 : 2;                   // K steps of operand reads in flight ahead of the matrix instructions
    constexpr int MPS = (8 * AQ + KS - 1) / KS;           // v_max3 per K step
    int xx_prev = 0x7fffffff;                             // nothing passes before the first tile
    // one tile: accumulate into acc while prev (the tile before) is reduced and tested
    auto body = [&](i32x16 (&acc)[AQ], i32x16 (&prev)[AQ], int64_t g) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * OPS) : "memory");  // own share of tile g has landed (later tiles may be in flight)
        __builtin_amdgcn_s_barrier();                                // everyone's share has; everyone is done with tile g - 1
        if (!(dbg & 1)) request(g + NB - 1 < n_groups ? g + NB - 1 : n_groups - 1, (int)((g + NB - 1) % NB));
        const int slot = (int)(g % NB);
        const i32x4 *pb = reinterpret_cast<const i32x4 *>(gf_ring) + (size_t)slot * KS * 64 + lane;
        const int xx_cur = xx_s[slot][lj];
#pragma unroll
        for (int a = 0; a < AQ; ++a) {
            const i32x4 *tp = reinterpret_cast<const i32x4 *>(&thh_s[wave][a][lk][0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const i32x4 v = tp[c];
                acc[a][4 * c] = v[0]; acc[a][4 * c + 1] = v[1]; acc[a][4 * c + 2] = v[2]; acc[a][4 * c + 3] = v[3];
            }
        }
        int mx[AQ];
#pragma unroll
        for (int a = 0; a < AQ; ++a) mx[a] = (int)0x80000000;
        i32x4 bv[KS];
#pragma unroll
        for (int s_ = 0; s_ < PD; ++s_) bv[s_] = pb[s_ * 64];
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            if (s_ + PD < KS) bv[s_ + PD] = pb[(s_ + PD) * 64];
#pragma unroll
            for (int a = 0; a < AQ; ++a) acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qreg[a][s_], bv[s_], acc[a], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < MPS; ++j) {
                const int idx = s_ * MPS + j;   // max3 number idx of the 8 AQ: query set idx / 8, results 2 (idx % 8), + 1
                if (idx < 8 * AQ) {
                    const int a2 = idx >> 3, e = (idx & 7) * 2;
                    const int m2 = prev[a2][e] > prev[a2][e + 1] ? prev[a2][e] : prev[a2][e + 1];
                    mx[a2] = mx[a2] > m2 ? mx[a2] : m2;
                }
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * AQ + PD + 1, 0);
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            if (s_ + PD < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, AQ, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, MPS, 0);
        }
        const int64_t t = t0 + g - 1;
#pragma unroll
        for (int a = 0; a < AQ; ++a) {
            const bool cand = mx[a] >= (xx_prev >> 1) && g > 0 && !(dbg & 2);
            if (__any(cand)) collect(prev[a], a, xx_prev, t * 32 + lj, cand);
        }
        xx_prev = xx_cur;
    };
    auto tail = [&](i32x16 (&prev)[AQ]) __attribute__((always_inline)) {   // the last tile's tests
        const int64_t t = t0 + n_groups - 1;
#pragma unroll
        for (int a = 0; a < AQ; ++a) {
            int m = prev[a][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) m = m > prev[a][e] ? m : prev[a][e];
            const bool cand = m >= (xx_prev >> 1);
            if (__any(cand)) collect(prev[a], a, xx_prev, t * 32 + lj, cand);
        }
    };
    i32x16 ra[AQ], rb[AQ];
#pragma unroll
    for (int a = 0; a < AQ; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) rb[a][e] = (int)0x80000000;
    int64_t g = 0;
    for (; g + 1 < n_groups; g += 2) {
        body(ra, rb, g);
        body(rb, ra, g + 1);
    }
    if (g < n_groups) {
        body(ra, rb, g);
        tail(ra);
    } else {
        tail(rb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (parked) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(pair_cnt, (uint32_t)parked);
        base = __shfl(base, 0, 64);
        for (int i = lane; i < parked; i += 64)
            if (base + i < pair_cap) pairs[base + i] = park_s[wave][i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Small and mid-size batches (1 .. 128 queries) as a STREAM over the raw rows (flat_u8_mstream_kernel): the row-tile kernels read the rows
// lane-per-row (64 cache lines per load instruction) and top out near 3 TB/s on 512-byte rows and 1 TB/s on 128-byte rows; the
// filter pipeline needs the packed copy and a sample stage and only pays from ~256 queries.  Here each wave streams whole 32-row
// tiles (32 D contiguous bytes) into a wave-private LDS ring by LDS-DMA -- coalesced 1 KB pieces, no barrier anywhere: the ring, the
// waits and the matrix chain all belong to one wave --, reads them back as v_mfma_i32_32x32x32_i8 B operands (rows of a piece keep
// their 256-byte alignment, so the reads run 2- to 4-way bank-conflicted: 64-128 LDS cycles per tile, a few per cent of a tile's
// time), multiplies against QB x 32 queries held in registers, and keeps only MINIMA: per tile and query (one coalesced 128 QB-byte
// store per tile) and per wave and query.  Selection is flat_u8_stream_finish_kernel<true>: theta from the wave minima, then the
// ~k tiles whose minimum is <= theta get their 32 distances recomputed from the rows.  One wave per SIMD (QB x D/8 operand registers).
// ------------------------------------------------------------------------------------------------------------------
template <int O, int CNT, int V>
__device__ __forceinline__ void fold_min(int (&s_)[V], int sub)
{
    if constexpr (O >= 1) {
        if constexpr (CNT > 1) {
            const bool hi = (sub & O) != 0;
#pragma unroll
            for (int i = 0; i < CNT / 2; ++i) {
                const int keep = hi ? s_[i + CNT / 2] : s_[i];
                const int send = hi ? s_[i] : s_[i + CNT / 2];
                const int got = __shfl_xor(send, O, 64);
                s_[i] = keep < got ? keep : got;
            }
            fold_min<O / 2, CNT / 2, V>(s_, sub);
        } else {
            const int got = __shfl_xor(s_[0], O, 64);
            s_[0] = s_[0] < got ? s_[0] : got;
            fold_min<O / 2, 1, V>(s_, sub);
        }
    }
}
constexpr int ms_log2(int v) { return v <= 1 ? 0 : 1 + ms_log2(v / 2); }
constexpr int MS_PAD = 32;   // bytes between the 1 KB pieces of a tile in LDS
constexpr int MS_GROUP = 4;  // tiles of a wave whose minima are kept as one (tmin[(i / MS_GROUP) * waves + wave])

template <int KS, int QB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flat_u8_mstream_kernel(
    const uint8_t *__restrict__ X, const int32_t *__restrict__ norms, int64_t n, const uint8_t *__restrict__ Q, int nq,
    int32_t *__restrict__ tmin, int32_t *__restrict__ wmin)
{
    constexpr int D = 32 * KS, NB = 32 / KS >= 2 ? 32 / KS : 2;       // ring slots per wave: 32 KB of tiles
    constexpr int PIECE = 1024 + MS_PAD, SLOT = KS * PIECE + 256;      // + the tile's 32 norms (glds4: 64 lanes x 4 B)
    constexpr int OPS = KS + 1;
    constexpr int V = 16 * QB, LB = 5, VB = ms_log2(V), T = LB < VB ? LB : VB, R = V >> T;
    constexpr int NQP = 32 * QB;
    extern __shared__ __attribute__((aligned(16))) uint8_t ms_ring[];  // [4 waves][NB][SLOT]
    __shared__ int qq_s[NQP];
    const int tid = threadIdx.x, lane = tid & 63, lj = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    i32x4 qreg[QB][KS];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int ql = b * 32 + lj;
        const int qc = ql < nq ? ql : nq - 1;
        const uint8_t *qp = Q + (int64_t)qc * D;
        int qq = 0;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) qreg[b][s_] = *reinterpret_cast<const i32x4 *>(qp + 32 * s_ + 16 * lk);
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            qreg[b][s_] ^= (int)0x80808080;
#pragma unroll
            for (int c = 0; c < 4; ++c) qq = __builtin_amdgcn_sdot4(qreg[b][s_][c], qreg[b][s_][c], qq, false);
        }
        qq += __shfl_xor(qq, 32, 64);
        if (wave == 0 && lk == 0) qq_s[ql] = ql < nq ? qq : 0x3fffffff;   // padding queries: minima that never qualify
    }
    __syncthreads();   // (drains the ordinary loads: from here on the wave's loads are the DMA)
    // what this lane holds after the min-fold over the 32 rows of its half: values v = pfx * R + i, v = 16 b + e, query = 32 b + (e & 3) + 8 (e >> 2) + 4 lk
    const int sub = lj;
    const bool writer = (sub & ((1 << (LB - T)) - 1)) == 0;
    const int pfx = sub >> (LB - T);
    int my_q[R], my_qq[R], mn[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int v = pfx * R + i, b = v >> 4, e = v & 15;
        my_q[i] = 32 * b + (e & 3) + 8 * (e >> 2) + 4 * lk;
        my_qq[i] = qq_s[my_q[i]];
        mn[i] = 0x7fffffff;
    }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t G = (int64_t)gridDim.x * 4, wave_g = (int64_t)blockIdx.x * 4 + wave;
    const int64_t my_tiles = wave_g < n_tiles ? (n_tiles - wave_g + G - 1) / G : 0;
    const uint32_t ring_b = (uint32_t)(uintptr_t)ms_ring + (uint32_t)(wave * NB * SLOT);
    // this lane's share of a tile: piece p covers tile bytes [1024 p + 16 lane, +16) = row (1024 p + 16 lane) / D; rows past n are clamped
    auto request = [&](int64_t i, int slot) {
        const int64_t t = wave_g + (i < my_tiles ? i : my_tiles - 1) * G;
#pragma unroll
        for (int p = 0; p < KS; ++p) {
            const int byte = 1024 * p + 16 * lane;
            int64_t row = t * 32 + byte / D;
            row = row < n ? row : n - 1;
            glds16<U8_NT_MS>(X + row * D + (byte % D), ring_b + (uint32_t)(slot * SLOT + p * PIECE));
        }
        int64_t row = t * 32 + lj;
        row = row < n ? row : n - 1;
        glds4(norms + row, ring_b + (uint32_t)(slot * SLOT + KS * PIECE));
    };
    if (my_tiles > 0) {
#pragma unroll
        for (int p = 0; p < NB - 1; ++p) request(p, p);
    }
    // B operand of row lj, K step s: tile byte lj D + 32 s + 16 lk -> piece (lj D) >> 10, offset (lj D) & 1023
    const uint32_t rd_off = (uint32_t)(((lj * D) >> 10) * PIECE + ((lj * D) & 1023) + 16 * lk);
    int dmin[V];
#pragma unroll
    for (int v = 0; v < V; ++v) dmin[v] = 0x7fffffff;
    for (int64_t i = 0; i < my_tiles; ++i) {
        request(i + NB - 1, (int)((i + NB - 1) % NB));
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * OPS) : "memory");   // tile i has landed (loads complete in order; the stores below can only add to the wait)
        const int slot = (int)(i % NB);
        const uint8_t *base = ms_ring + (size_t)wave * NB * SLOT + (size_t)slot * SLOT;
        const int64_t t = wave_g + i * G;
        const int64_t row = t * 32 + lj;
        const int xx = reinterpret_cast<const int *>(base + KS * PIECE)[lj];
        i32x16 acc[QB];
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            i32x4 bv = *reinterpret_cast<const i32x4 *>(base + rd_off + 32 * s_);
            bv ^= (int)0x80808080;
#pragma unroll
            for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qreg[b][s_], bv, acc[b], 0, 0, 0);
        }
        const int xr = row < n ? xx : 0x3fffffff;   // rows past the end never set a minimum
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int d = xr - 2 * acc[b][e];   // + |q'|^2 after the fold (constant per query)
                dmin[16 * b + e] = d < dmin[16 * b + e] ? d : dmin[16 * b + e];
            }
        // the butterfly across the 32 rows costs ~5 instructions per value: it runs once per MS_GROUP of the wave's tiles (their
        // minima are first combined lane by lane, one v_min each), and the finish kernel treats those tiles as one unit
        if ((i & (MS_GROUP - 1)) == MS_GROUP - 1 || i == my_tiles - 1) {
            fold_min<16, V, V>(dmin, sub);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int d = dmin[r] + my_qq[r];
                if (writer) tmin[((i / MS_GROUP) * G + wave_g) * NQP + my_q[r]] = d;
                mn[r] = d < mn[r] ? d : mn[r];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) dmin[v] = 0x7fffffff;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (writer && my_q[r] < nq) wmin[(int64_t)my_q[r] * G + wave_g] = mn[r];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one workgroup per query: (distance, row) sort of the survivors and the sample's k results; distances are int32 bits
__global__ __launch_bounds__(kBlock) void flat_u8_finish_kernel(const uint32_t *__restrict__ cand_cnt, const float *__restrict__ cand_d,
                                                                const int32_t *__restrict__ cand_row, int cap, int k,
                                                                const float *__restrict__ sample_d, const int64_t *__restrict__ sample_i,
                                                                float *__restrict__ out_d, int64_t *__restrict__ out_i,
                                                                uint32_t *__restrict__ overflow)
{
    __shared__ unsigned long long key_s[4096];
    const int64_t qi = blockIdx.x;
    const int tid = threadIdx.x;
    uint32_t c = cand_cnt[qi];
    if (tid == 0) atomicMax(overflow, c);
    c = c < (uint32_t)cap ? c : (uint32_t)cap;
    const int total = (int)c + k;  // <= 4096
    int n2 = 1;
    while (n2 < total) n2 <<= 1;
    for (int i = tid; i < n2; i += kBlock) {
        unsigned long long e = ~0ull;
        if (i < (int)c) e = ((unsigned long long)__float_as_uint(cand_d[qi * cap + i]) << 32) | (uint32_t)cand_row[qi * cap + i];
        else if (i < total) {
            const int64_t id = sample_i[qi * k + (i - (int)c)];
            if (id >= 0) e = ((unsigned long long)__float_as_uint(sample_d[qi * k + (i - (int)c)]) << 32) | (uint32_t)id;
        }
        key_s[i] = e;
    }
    __syncthreads();
    block_bitonic<unsigned long long, false>(key_s, n2);
    for (int i = tid; i < k; i += kBlock) {
        const unsigned long long e = key_s[i];
        const bool ok = e != ~0ull;
        out_d[qi * k + i] = ok ? __uint_as_float((uint32_t)(e >> 32)) : __uint_as_float(0x7f800000u);
        out_i[qi * k + i] = ok ? (int64_t)(uint32_t)e : -1;
    }
}

// timing experiments (results are WRONG when non-zero; honoured by -DCVTMI_GF_DBG builds only): bit 0 no DMA after the first groups,
// bit 1 no survivor tests.  What they showed on 10 M x 512-d, nq = 4096 (tools/bench_flat_u8_opt.py, DBG=0,2,3): filter kernel 18.7 ms,
// without the tests 17.6, without tests and DMA 14.4 = 2.8 P int-op/s -- against 3.2-3.5 P for a loop of nothing but matrix
// instructions on random bytes (tools/ubench/mfma_i8_feed.hip; 4.4-4.9 P on constant bytes: the ceiling is the power the operands draw)
static int g_u8_dbg = 0;
int set_flat_u8_dbg(int v)
{
#ifdef CVTMI_GF_DBG
    g_u8_dbg = v;
    return CVTMI_OK;
#else
    (void)v;
    return CVTMI_EUNSUPPORTED;
#endif
}
static int g_u8_gfilter = 1;  // cvtmi_set_tuning("flat_u8_gfilter"): 1 = the software-pipelined filter kernel where it applies (D = 64 .. 512, power of two)
void set_flat_u8_gfilter(int v) { g_u8_gfilter = v; }  // 0 off, 1 choose, 2 two 4-wave workgroups per CU, 3 one 8-wave workgroup per CU, 4 one wave per SIMD x 128 queries
bool flat_u8_gfilter_shape(int D) { return g_u8_gfilter && (D == 64 || D == 128 || D == 256 || D == 512); }

// 1 .. 128 queries over raw rows: stream + minima (the selection is launch_flat_u8_mstream_finish, flat.hip)
constexpr int MSTREAM_BLOCKS = 256;   // one 4-wave workgroup per CU (one wave per SIMD); 192 / 240 / 252 / 255 measured the same or worse
static int g_mstream_min_nq = 1;       // measurement hook (flat_u8_mstream_min): below it the row-per-lane / row-tile kernels answer
void set_flat_u8_mstream_min(int v) { g_mstream_min_nq = v; }
// Smallest table the stream takes.  Structural bound: the selection needs k waves with at least one 32-row tile each (k <= 128: 4096 rows);
// waves without a tile publish "no minimum" and are never looked at.  It was 262 144 (eight tiles per wave) until a sweep over table
// sizes (round 5, tools/sweep_flat_small_tables.py) showed the row-tile kernels 2-20x behind on everything smaller: 65 536 x 512-d,
// 16 / 100 / 1000 queries 0.39 / 0.84 / 1.06 ms against 0.046 / 0.060 / 0.47.
static std::atomic<int64_t> g_mstream_min_rows{4096};   // cvtmi_set_tuning("flat_u8_mstream_min_rows")
void set_flat_u8_mstream_min_rows(int64_t v) { g_mstream_min_rows = v < 4096 ? 4096 : v; }
bool flat_u8_mstream_applies(int D, int64_t n, int64_t nq, int k)
{
    return (D == 128 || D == 256 || D == 512) && nq >= g_mstream_min_nq && nq <= 128 && n >= g_mstream_min_rows.load() && n < 0x7fffffff && k <= 128 &&
           n / 32 / 64 / (MSTREAM_BLOCKS * 4) + 2 <= 160;   // rounds a finish slice can span (FIN_MAXR, flat.hip): 331 M rows
}
// entries of the tile-group minima array: groups of MS_GROUP tiles per wave, (group, wave) major
int64_t flat_u8_mstream_groups(int64_t n)
{
    const int64_t waves = (int64_t)MSTREAM_BLOCKS * 4, tiles = (n + 31) / 32;
    const int64_t per_wave = (tiles + waves - 1) / waves;
    return (per_wave + MS_GROUP - 1) / MS_GROUP * waves;
}
int flat_u8_mstream_group() { return MS_GROUP; }
// scratch (int32): tile-group minima [groups][32 QB] then wave minima [nq][waves]
size_t flat_u8_mstream_scratch(int64_t n, int64_t nq, int *nqp, int *waves)
{
    const int qb = nq <= 32 ? 1 : (nq <= 64 ? 2 : 4);
    if (nqp) *nqp = 32 * qb;
    if (waves) *waves = MSTREAM_BLOCKS * 4;
    return ((size_t)flat_u8_mstream_groups(n) * 32 * qb + (size_t)nq * MSTREAM_BLOCKS * 4) * sizeof(int32_t);
}
int launch_flat_u8_mstream(int D, const uint8_t *data, const int32_t *norms, int64_t n, const uint8_t *q, int64_t nq, int32_t *tmin,
                           int32_t *wmin, hipStream_t st)
{
    const int qb = nq <= 32 ? 1 : (nq <= 64 ? 2 : 4);
#define CVTMI_MS(KS_, QB_)                                                                                                         \
    do {                                                                                                                          \
        constexpr int NB_ = 32 / (KS_) >= 2 ? 32 / (KS_) : 2;                                                                     \
        const size_t lds = (size_t)4 * NB_ * ((KS_) * (1024 + MS_PAD) + 256);                                                     \
        CVTMI_HIP(hipFuncSetAttribute((const void *)flat_u8_mstream_kernel<KS_, QB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((flat_u8_mstream_kernel<KS_, QB_>), dim3(MSTREAM_BLOCKS), dim3(256), lds, st, data, norms, n, q, (int)nq, tmin, wmin); \
    } while (0)
#define CVTMI_MSQ(KS_)                                                                                                             \
    do {                                                                                                                          \
        if (qb == 1) CVTMI_MS(KS_, 1); else if (qb == 2) CVTMI_MS(KS_, 2); else CVTMI_MS(KS_, 4);                                   \
    } while (0)
    switch (D) {
        case 128: CVTMI_MSQ(4); break;
        case 256: CVTMI_MSQ(8); break;
        case 512: CVTMI_MSQ(16); break;
        default: return fail(CVTMI_EUNSUPPORTED, "flat_u8_mstream: D=%d", D);
    }
#undef CVTMI_MSQ
#undef CVTMI_MS
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

bool flat_u8_filter_applies(int D, int64_t n, int64_t nq, int k)
{
    return D % 32 == 0 && D >= 32 && D <= 512 && nq >= 256 && n >= 262144 && k <= 64;
}
size_t flat_u8_pack_bytes(int D, int64_t n) { return (size_t)((n + 31) / 32) * (D / 32) * 64 * sizeof(uint4); }

// rows [row0, n) (row0 a multiple of 32; the tiles before it stay as they are)
int launch_flat_u8_pack(const uint8_t *X, int64_t n, int D, uint4 *pack, hipStream_t st, int64_t row0)
{
    const int64_t ntiles = (n + 31) / 32, tile0 = row0 / 32, total = (ntiles - tile0) * (D / 32) * 64;
    if (total <= 0) return CVTMI_OK;
    hipLaunchKernelGGL(flat_u8_pack_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, X, n, D, tile0, ntiles, pack);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// rows [row_begin, n) against all queries; thresholds = the sample's k-th best; *pair_cnt zeroed by the caller
int launch_flat_u8_filter(const uint8_t *q, int64_t nq, int D, const uint4 *pack, const int32_t *norms, const float *sample_d, int k,
                          int64_t row_begin, int64_t n, uint32_t pair_cap, uint32_t *pair_cnt, uint4 *pairs, hipStream_t st)
{
    const int64_t tile_begin = row_begin / 32, tile_end = (n + 31) / 32;
    if (tile_end <= tile_begin) return CVTMI_OK;
    if (flat_u8_gfilter_shape(D)) {  // the software-pipelined kernel
        const bool two = g_u8_gfilter == 2;   // two 4-wave workgroups per CU: measured 15 % slower than one 8-wave workgroup (16 matrix instructions per barrier)
        const bool wide4 = g_u8_gfilter == 4 && D >= 128;   // one wave per SIMD, 96-128 queries per wave (flat_u8_gfilter_wide_kernel): measured equal at nq = 4096, 10 % slower at nq = 1000
        const int qpb = wide4 ? (D == 512 ? 384 : 512) : (two ? 128 : 256);
        const int64_t qblocks = (nq + qpb - 1) / qpb;
        const int64_t want = two ? 512 : 256;                                     // workgroups resident at a time
        // row splits: a multiple of 8 (one split per XCD at a time), chosen so that rounds(grid / resident) x tiles per split is smallest --
        // qblocks x splits rarely lands on a multiple of the resident count, and a 3 % overhang used to cost a whole second round
        // (nq = 3840: 15 x 24 = 360 workgroups on 256 slots, 33 ms against 21 ms for nq = 4096)
        const int64_t tiles = tile_end - tile_begin;
        int64_t tps = tiles, best_cost = INT64_MAX;
        for (int64_t m = 1; m <= 256; ++m) {
            const int64_t t = std::max<int64_t>(64, (tiles + 8 * m - 1) / (8 * m));
            const int64_t s_used = (tiles + t - 1) / t, grid = (s_used + 7) / 8 * 8 * qblocks;
            const int64_t cost = (grid + want - 1) / want * (t + 24);              // + start-up of a workgroup, in tiles
            if (cost < best_cost) { best_cost = cost; tps = t; }
            if (t == 64) break;
        }
        const int64_t splits = (tiles + tps - 1) / tps;
        const int64_t splits8 = (splits + 7) / 8 * 8;
        if (splits8 * qblocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "flat u8 filter: nq too large");
        const dim3 g((unsigned)(splits8 * qblocks));
#define CVTMI_GF(N, NW, G, NB)                                                                                                    \
    do {                                                                                                                         \
        const size_t lds = (size_t)(NB) * (G) * (N) * 64 * sizeof(uint4);                                                        \
        CVTMI_HIP(hipFuncSetAttribute((const void *)flat_u8_gfilter_kernel<N, NW, G, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((flat_u8_gfilter_kernel<N, NW, G, NB>), g, dim3(64 * (NW)), lds, st, q, nq, pack, norms, n, sample_d, k, tile_begin, tile_end, \
                           tps, pair_cap, pair_cnt, pairs, (int)qblocks, g_u8_dbg);                                              \
    } while (0)
#define CVTMI_GW(N, AQ, NB)                                                                                                      \
    do {                                                                                                                         \
        const size_t lds = (size_t)(NB) * (N) * 64 * sizeof(uint4);                                                              \
        CVTMI_HIP(hipFuncSetAttribute((const void *)flat_u8_gfilter_wide_kernel<N, AQ, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((flat_u8_gfilter_wide_kernel<N, AQ, NB>), g, dim3(256), lds, st, q, nq, pack, norms, n, sample_d, k, tile_begin, tile_end, \
                           tps, pair_cap, pair_cnt, pairs, (int)qblocks, g_u8_dbg);                                              \
    } while (0)
        if (wide4) {
            switch (D / 32) {
                case 4: CVTMI_GW(4, 4, 4); break;
                case 8: CVTMI_GW(8, 4, 4); break;
                default: CVTMI_GW(16, 3, 4); break;   // 3 x 64 operand registers (an 8-slot ring measured the same as 4: the DMA is not latency-bound): the fourth query set does not fit beside two accumulator sets
            }
            CVTMI_HIP(hipGetLastError());
            return CVTMI_OK;
        }
#undef CVTMI_GW
        switch (D / 32) {
            case 2: CVTMI_GF(2, 8, 4, 3); break;
            case 4: if (two) CVTMI_GF(4, 4, 2, 3); else CVTMI_GF(4, 8, 2, 3); break;
            case 8: if (two) CVTMI_GF(8, 4, 2, 3); else CVTMI_GF(8, 8, 2, 3); break;
            case 16: if (two) CVTMI_GF(16, 4, 2, 2); else CVTMI_GF(16, 8, 2, 3); break;   // 4 waves: ring of 2 groups, two workgroups fit a CU's LDS
            default: return fail(CVTMI_EUNSUPPORTED, "flat u8 pipelined filter: D=%d (64, 128, 256 or 512)", D);
        }
#undef CVTMI_GF
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    }
    const bool wide = nq >= 1024;  // 16 waves: 512 queries per workgroup
    const int qpb = wide ? 512 : 256;
    const int64_t qblocks = (nq + qpb - 1) / qpb;
    int64_t splits = std::max<int64_t>(1, (256 * 2 + qblocks - 1) / qblocks);
    int64_t tps = std::max<int64_t>(16, (tile_end - tile_begin + splits - 1) / splits);
    splits = (tile_end - tile_begin + tps - 1) / tps;
    const int64_t splits8 = (splits + 7) / 8 * 8;
    if (splits8 * qblocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "flat u8 filter: nq too large");
    const dim3 g((unsigned)(splits8 * qblocks));
#define CVTMI_FU(N)                                                                                                              \
    do {                                                                                                                         \
        if (wide)                                                                                                                \
            hipLaunchKernelGGL((flat_u8_filter_kernel<N, 1024, 1>), g, dim3(1024), 0, st, q, nq, D, pack, norms, n, sample_d, k, tile_begin,   \
                               tile_end, tps, pair_cap, pair_cnt, pairs, (int)qblocks);                                          \
        else                                                                                                                     \
            hipLaunchKernelGGL((flat_u8_filter_kernel<N, 512, 1>), g, dim3(512), 0, st, q, nq, D, pack, norms, n, sample_d, k, tile_begin,   \
                               tile_end, tps, pair_cap, pair_cnt, pairs, (int)qblocks);                                          \
    } while (0)
    const int ks = D / 32;
    if (ks <= 2) CVTMI_FU(2);
    else if (ks <= 4) CVTMI_FU(4);
    else if (ks <= 8) CVTMI_FU(8);
    else CVTMI_FU(16);
#undef CVTMI_FU
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_flat_u8_finish(int64_t nq, const uint32_t *pair_cnt, uint32_t pair_cap, const uint4 *pairs, int cap, int k, const float *sample_d,
                          const int64_t *sample_i, uint32_t *cand_cnt, float *cand_d, int32_t *cand_row, float *out_d, int64_t *out_i,
                          uint32_t *overflow, hipStream_t st)
{
    if (cap + k > 4096) return fail(CVTMI_EUNSUPPORTED, "flat u8 finish: cap=%d k=%d", cap, k);
    hipLaunchKernelGGL(flat_scatter_kernel, dim3(256 * 8), dim3(kBlock), 0, st, pair_cnt, pair_cap, pairs, cap, cand_cnt, cand_d, cand_row, overflow);
    hipLaunchKernelGGL(flat_u8_finish_kernel, dim3((unsigned)nq), dim3(kBlock), 0, st, cand_cnt, cand_d, cand_row, cap, k, sample_d, sample_i, out_d,
                       out_i, overflow);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

bool flat_filter_applies(int metric, int D, int64_t n, int64_t nq, int k)
{
    return (metric == CVTMI_METRIC_IP || metric == CVTMI_METRIC_L2F) && D >= 32 && D <= 128 && D % 16 == 0 && nq >= 16 && n >= 131072 &&
           k <= 128;   // from 16 queries: 1 M x 128-d nq = 16 / 32 took 0.40-0.46 / 0.63-0.71 ms on the exact kernels, 0.42-0.44 through the filter
}

size_t flat_pack_bytes(int nch, int64_t n) { return (size_t)((n + 31) / 32) * nch * 2 * 64 * sizeof(uint4); }

// rows [row0, n) (row0 > 0: appended rows -- the tile that holds row0 is packed again, the statistics keep accumulating)
int launch_flat_pack(const float *X, int64_t n, int D, int nch, int metric, uint4 *pack, uint32_t *bias, uint32_t *stats, hipStream_t st, int64_t row0)
{
    const int64_t ntiles = (n + 31) / 32, tile0 = row0 / 32;
    if (row0 == 0) CVTMI_HIP(hipMemsetAsync(stats, 0, 12, st));
    if (tile0 >= ntiles) return CVTMI_OK;
    hipLaunchKernelGGL(flat_pack_kernel, dim3((unsigned)(((ntiles - tile0) * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, X, n, D, nch,
                       metric == CVTMI_METRIC_L2F ? 1 : 0, tile0, ntiles, pack, bias, stats);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

int launch_flat_thr(const float *q, int64_t nq, int D, int metric, const float *sample_d, int k, uint32_t *stats, float *thr,
                    float *margin, hipStream_t st)
{
    hipLaunchKernelGGL(flat_thr_kernel, dim3((unsigned)((nq + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, q, nq, D,
                       metric == CVTMI_METRIC_L2F ? 1 : 0, sample_d, k, stats, thr, margin);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// rows [row_begin, n) (row_begin % 32 == 0) against all queries; *pair_cnt must be zeroed by the caller
int launch_flat_filter(const float *q, int64_t nq, int D, const uint4 *pack, const uint32_t *bias, const float *thr, int64_t row_begin,
                       int64_t n, uint32_t pair_cap, uint32_t *pair_cnt, uint4 *pairs, hipStream_t st)
{
    const int64_t tile_begin = row_begin / 32, tile_end = (n + 31) / 32;
    if (tile_end <= tile_begin) return CVTMI_OK;
    const int64_t qblocks = (nq + 255) / 256;
    int64_t splits = std::max<int64_t>(1, (256 * 4 + qblocks - 1) / qblocks);   // ~4 workgroups per CU in flight
    int64_t tps = std::max<int64_t>(16, (tile_end - tile_begin + splits - 1) / splits);
    splits = (tile_end - tile_begin + tps - 1) / tps;
    const int64_t splits8 = (splits + 7) / 8 * 8;  // empty splits return at once
    if (splits8 * qblocks > 0x7fffffff) return fail(CVTMI_EUNSUPPORTED, "flat filter: nq too large");
    const dim3 g((unsigned)(splits8 * qblocks)), b(FF_THREADS);
#define CVTMI_FF(N)                                                                                                               \
    case N:                                                                                                                       \
        hipLaunchKernelGGL((flat_filter_kernel<N>), g, b, 0, st, q, nq, pack, bias, thr, tile_begin, tile_end, tps, pair_cap, pair_cnt, pairs, (int)qblocks); \
        break;
    switch (D / 16) {
        CVTMI_FF(2) CVTMI_FF(3) CVTMI_FF(4) CVTMI_FF(5) CVTMI_FF(6) CVTMI_FF(7) CVTMI_FF(8)
        default: return fail(CVTMI_EUNSUPPORTED, "flat filter: D=%d", D);
    }
#undef CVTMI_FF
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// cand_cnt [nq] must be zeroed by the caller; *overflow > cap afterwards: a list ran over, the results are not to be used
int launch_flat_finish(int metric, const float *X, int64_t n, int D, const float *q, int64_t nq, const uint32_t *pair_cnt, uint32_t pair_cap,
                       const uint4 *pairs, int cap, int k, const float *margin, const float *sample_d, const int64_t *sample_i,
                       uint32_t *cand_cnt, float *cand_t, int32_t *cand_row, float *out_d, int64_t *out_i, uint32_t *overflow,
                       hipStream_t st)
{
    if (cap > FIN_CAND || k > 128) return fail(CVTMI_EUNSUPPORTED, "flat finish: cap=%d k=%d", cap, k);
    hipLaunchKernelGGL(flat_scatter_kernel, dim3(256 * 8), dim3(kBlock), 0, st, pair_cnt, pair_cap, pairs, cap, cand_cnt, cand_t, cand_row, overflow);
    if (metric == CVTMI_METRIC_IP)
        hipLaunchKernelGGL((flat_finish_kernel<true, 4>), dim3((unsigned)nq), dim3(kBlock), 0, st, X, n, D, q, cand_cnt, cand_t, cand_row, cap, k,
                           margin, sample_d, sample_i, out_d, out_i, overflow);
    else
        hipLaunchKernelGGL((flat_finish_kernel<false, 8>), dim3((unsigned)nq), dim3(kBlock), 0, st, X, n, D, q, cand_cnt, cand_t, cand_row, cap, k,
                           margin, sample_d, sample_i, out_d, out_i, overflow);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
