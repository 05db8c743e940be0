// flat_f32_stream.hip -- exhaustive fp32 search (BruteforceSearch<float>::searchKnn, brutoforce.hpp:73-93, with
// InnerProductSpace, space_ip.hpp:211-239, or L2Space, space_l2.h:153-184) as ONE stream over the fp32 rows.
//
// The answer -- the k smallest (distance, row) per query, distances in the reference's own summation order -- has to come out
// bit for bit, so every distance that is reported is evaluated by the exact code (dist_f32.h order) in the finishing kernel.
// The stream only decides WHICH rows need one:
//   * flat_f32_mstream_kernel: every wave streams 32-row tiles of the blocked fp32 layout (flat.hip) into a wave-private LDS
//     ring by LDS-DMA (no barrier anywhere: ring, waits and matrix chain belong to one wave), splits the fp32 values into two
//     bf16 terms on the fly (no operand copy of the rows in HBM), and scores T = q.x + b_x (b_x = -|x|^2/2 for L2, 0 for the
//     inner product; rides in as the accumulator's start value) for 32 QB queries held in registers on
//     v_mfma_f32_32x32x16_bf16 (x1.q1 + x2.q1 + x1.q2).  A lane ends up with the scores of ITS row against 16 QB queries, so
//     the running best and second best of a lane's rows over a GROUP of up to 32 consecutive tiles cost three VALU
//     instructions per score and no cross-lane traffic: key = T with its low five bits replaced by the tile's position
//     in the group, second = med3(best, second, key), best = max(best, key).  Per group, lane and query one (best, second)
//     pair goes to HBM: 8 bytes per 32 rows and query.
//   * flat_f32_stream_finish_kernel (one workgroup per query): theta = the k-th largest best (k distinct groups = k distinct
//     rows); a row whose score is below theta - margin is beaten, in the reference's own arithmetic, by k rows (bound below),
//     so the candidates are the best rows of the groups with best >= theta - margin, plus ALL rows of a group whose second best
//     also reaches the cut (two answers in one group: rare).  The k-and-a-few candidates get exact distances and are sorted by
//     (distance, row).
// A query the bound does not cover (non-finite values, magnitudes outside 2^-60 .. 2^60) or whose lists run over (masses of
// near ties) is flagged in redo[]; the caller then launches the exact kernels of flat.hip with that array as a predicate --
// they exit at once for every query that is not flagged.  No host synchronisation anywhere on the path.
//
// Bound (u = 2^-24, Q = |q|^2 + max |x|^2, |T| <= Q; same accounting as flat_mfma.hip): accumulation of 3 D + 1 terms taken as
// 2u per term <= 388 uQ (D <= 128; 772 uQ at D = 256), bf16 splits 64 uQ, the omitted x2.q2 term 32 uQ, b_x in fp32 8 uQ, the
// position bits 2^-18 |T| = 64 uQ: <= 556 uQ (940 uQ at D = 256) on a key; the reference's own sum ~200 uQ (400 uQ) in T units.
// Two rows are ordered the same way by their keys and by the reference whenever the keys differ by more than
// 2 x (556 + 200) uQ = 1512 uQ (D <= 128) / 2 x (940 + 400) = 2680 uQ (D = 256); margin = 2^-13 Q = 2048 uQ, 2^-12 Q for D > 128.
#include <algorithm>
#include <type_traits>

#include <atomic>

#include "common.h"
#include "flat_f32_common.h"
#include "kernels.h"

namespace cvtmi {

namespace {


constexpr int FS_BLOCKS = 256;           // one 4-wave workgroup per CU, one wave per SIMD
constexpr int FS_WAVES = FS_BLOCKS * 4;
constexpr int FS_POS_BITS = 5;           // tiles per group <= 32

#ifdef CVTMI_FS_TIMING
// phase clocks (100 MHz wall clock) of workgroup 0 of the collect ([0..7]) and finish ([8..15]) kernels: tools/fs_timing.py
__device__ unsigned long long g_fs_dbg[16];
#define FS_T0() unsigned long long fs_last__ = wall_clock64()
#define FS_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now__ = wall_clock64(); g_fs_dbg[i] += now__ - fs_last__; fs_last__ = now__; } } while (0)
__device__ unsigned long long g_fss_dbg[2][8];   // shader clocks of waves 0 and 4 of workgroup 0 of the shared-ring kernel, per section
#define FSS_T0() unsigned long long fss_last__ = clock64(); unsigned long long fss_acc__[8] = {}
#define FSS_T(i) do { const unsigned long long now__ = clock64(); fss_acc__[i] += now__ - fss_last__; fss_last__ = now__; } while (0)
#define FSS_TEND() do { if (blockIdx.x == 0 && (wave & 3) == 0 && wave_id < 8 && lane == 0) for (int i__ = 0; i__ < 8; ++i__) g_fss_dbg[wave_id >> 2][i__] = fss_acc__[i__]; } while (0)
#else
#define FS_T0() do { } while (0)
#define FS_T(i) do { } while (0)
#define FSS_T0() do { } while (0)
#define FSS_T(i) do { } while (0)
#define FSS_TEND() do { } while (0)
#endif

// nt: non-temporal hint -- rows that one CU reads once per launch need no place in L2 / Infinity Cache (issued -> landed
// 18 % sooner, MI355X_MICROARCH.md "nt-weights"; 1 M x 128-d, one query: 0.119 -> 0.106 ms)
__device__ __forceinline__ void fs_glds16(const void *gsrc, uint32_t lds_dst, int nt)
{
    lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);   // wave-uniform by construction: keep it in an SGPR
    unsigned keep;
    if (nt)  // kernel argument: scalar branch
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void fs_glds4(const void *gsrc, uint32_t lds_dst)
{
    lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}


// max-fold of V values per lane across the 32 lanes of a half wave (lanes differ in sub = lane & 31): every step halves the
// values a lane keeps, so after the steps lane sub holds R = V >> T of them, maxima over all 32 lanes: value pfx * R + i with
// pfx = sub >> (5 - T), T = min(5, log2 V); when V < 32 the last 5 - T steps are plain butterflies
template <int O, int CNT, int V>
__device__ __forceinline__ void fs_fold_max(float (&s_)[V], int sub)
{
    if constexpr (O >= 1) {
        if constexpr (CNT > 1) {
            const bool hi = (sub & O) != 0;
#pragma unroll
            for (int i = 0; i < CNT / 2; ++i) {
                const float keep = hi ? s_[i + CNT / 2] : s_[i];
                const float send = hi ? s_[i] : s_[i + CNT / 2];
                s_[i] = fmaxf(keep, __shfl_xor(send, O, 64));
            }
            fs_fold_max<O / 2, CNT / 2, V>(s_, sub);
        } else {
            s_[0] = fmaxf(s_[0], __shfl_xor(s_[0], O, 64));
            fs_fold_max<O / 2, 1, V>(s_, sub);
        }
    }
}
constexpr int fs_log2(int v) { return v <= 1 ? 0 : 1 + fs_log2(v / 2); }

// ring geometry: a UNIT is UK K steps (16 dimensions each) of one tile = 2 UK pieces of 1 KB
// PACKED (round 6): the rows come from the bf16 operand copy of the threshold filter (flat_pack_kernel) -- ONE ready-made 1 KB piece
// per K step (the first term) instead of two fp32 pieces that are split on the fly: half the bytes of a pass, one product per term.
template <int NCH, bool PACKED = false> struct FsGeom {
    static constexpr int UK = NCH % 4 == 0 ? 4 : (NCH % 2 == 0 ? 2 : 1);
    static constexpr int NU = NCH / UK;                        // units per tile
    static constexpr int UNIT = UK * (PACKED ? 1024 : 2048);   // bytes
    static constexpr int RU = 32768 / UNIT;                    // ring units per wave (32 KB)
    static constexpr int RB = RU / NU + 3;                     // bias slots (256 B each): tiles that can be in flight + 1
    static constexpr int WAVE_LDS = RU * UNIT + RB * 256;
    static constexpr int OPS = (PACKED ? UK : 2 * UK) + 1;     // DMA requests per unit
};

// X: blocked fp32 rows (float4 c of the 64 rows of block b contiguous: ((b * D/4 + c) * 64 + r) float4); bias[n_tiles * 32];
// gb[((q * NG + g) * FS_WAVES + wave) * 32 + j] = (best, second) of rows (wave + (g G + p) FS_WAVES) * 32 + j, p < G;
// wm[q * FS_WAVES + wave] = the largest best of the wave (the finish derives its first threshold from these 1024 values)
template <int NCH, int QB, bool PACKED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flat_f32_mstream_kernel(
    const float *__restrict__ X, const float *__restrict__ bias, int64_t n_tiles, const float *__restrict__ Q, int nq, int G, int NG,
    float2 *__restrict__ gb, float *__restrict__ wm, uint32_t *__restrict__ zero_a, uint32_t *__restrict__ zero_b, int nt)
{
    using Ge = FsGeom<NCH, PACKED>;
    constexpr int D = 16 * NCH, UK = Ge::UK, NU = Ge::NU, UNIT = Ge::UNIT, RU = Ge::RU, RB = Ge::RB, OPS = Ge::OPS;
    constexpr int NACC = QB == 1 ? 2 : QB;                     // QB == 1: two chains so that back-to-back products are independent
    extern __shared__ __attribute__((aligned(16))) uint8_t fs_ring[];   // [4 waves][RU units | RB bias slots]
    const int tid = threadIdx.x, lane = tid & 63, lj = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x == 0 && tid < nq) zero_a[tid] = zero_b[tid] = 0u;   // redo flags / list counters of this pass (read by later launches)
    // queries: lane (i, half) holds dimensions 16 s + 8 half .. + 8 of query 32 b + i, as two bf16 terms
    bf16x8 qh[QB][NCH], ql[QB][NCH];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qi = b * 32 + lj;
        const float *qp = Q + (int64_t)(qi < nq ? qi : nq - 1) * D + 8 * lk;
#pragma unroll
        for (int s = 0; s < NCH; ++s) {
            float v[8];
            *reinterpret_cast<float4 *>(&v[0]) = *reinterpret_cast<const float4 *>(qp + 16 * s);
            *reinterpret_cast<float4 *>(&v[4]) = *reinterpret_cast<const float4 *>(qp + 16 * s + 4);
            fs_split(v, qh[b][s], ql[b][s]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // from here on the wave's outstanding loads are the DMA requests only
    const int64_t wave_g = (int64_t)blockIdx.x * 4 + wave;
    const int my_tiles = wave_g < n_tiles ? (int)((n_tiles - wave_g + FS_WAVES - 1) / FS_WAVES) : 0;
    const int my_units = my_tiles * NU;
    const uint32_t ring_b = (uint32_t)(uintptr_t)fs_ring + (uint32_t)(wave * Ge::WAVE_LDS);
    const uint32_t bias_b = ring_b + (uint32_t)(RU * UNIT);
    // piece (s, h) of tile t: lanes 0-31 fetch float4 c = 4 s + h of rows 0-31 of the tile, lanes 32-63 float4 c = 4 s + 2 + h:
    // lane (j, half) then finds dimensions 16 s + 8 half .. + 8 of row j at its own slot of pieces (s, 0) and (s, 1)
    auto request = [&](int u) {
        const int uc = u < my_units ? u : my_units - 1;
        const int i = uc / NU;
        const int part = uc - i * NU;
        const int64_t t = wave_g + (int64_t)i * FS_WAVES;
        const uint32_t dst = ring_b + (uint32_t)((uc % RU) * UNIT);
        if constexpr (PACKED) {   // X = the operand copy: [tile][K step][first | second term][64 lanes] x 16 bytes
            const uint8_t *tile = reinterpret_cast<const uint8_t *>(X) + ((t * NCH + part * UK) * 2) * 1024 + lane * 16;
#pragma unroll
            for (int s = 0; s < UK; ++s) fs_glds16(tile + (size_t)s * 2048, dst + (uint32_t)(s * 1024), nt);
        } else {
            const float *tile = X + ((t >> 1) * (int64_t)(D / 4) * 64 + (t & 1) * 32 + lj) * 4;   // row j of the tile, float4 0
#pragma unroll
            for (int s = 0; s < UK; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    fs_glds16(tile + (int64_t)(4 * (part * UK + s) + 2 * lk + h) * 256, dst + (uint32_t)((2 * s + h) * 1024), nt);
        }
        fs_glds4(bias + t * 32 + lj, bias_b + (uint32_t)((i % RB) * 256));
    };
    if (my_tiles > 0) {
#pragma unroll
        for (int u = 0; u < RU - 1; ++u) request(u);
    }
    // wave maxima: the fold leaves lane sub with R of the V values (see fs_fold_max)
    constexpr int QBP = QB == 3 ? 4 : QB, V = 16 * QBP, T = fs_log2(V) < 5 ? fs_log2(V) : 5, R = V >> T;
    float wmax[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wmax[r] = FS_PAD_BIAS;
    float best[QB][16], second[QB][16];
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) best[b][e] = second[b][e] = FS_PAD_BIAS;
    f32x16 acc[NACC];
    const int gmask = G - 1, glog = 31 - __builtin_clz(G);
    for (int i = 0; i < my_tiles; ++i) {
#pragma unroll
        for (int part = 0; part < NU; ++part) {
            const int u = i * NU + part;
            request(u + RU - 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RU - 1) * OPS) : "memory");   // unit u has landed (requests retire in order)
            const uint8_t *ub = fs_ring + (size_t)wave * Ge::WAVE_LDS + (size_t)(u % RU) * UNIT + lane * 16;
            if (part == 0) {
                const float bx = reinterpret_cast<const float *>(fs_ring + (size_t)wave * Ge::WAVE_LDS + RU * UNIT + (size_t)(i % RB) * 256)[lj];
#pragma unroll
                for (int a = 0; a < NACC; ++a)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][e] = (QB == 1 && a == 1) ? 0.0f : bx;
            }
#pragma unroll
            for (int s = 0; s < UK; ++s) {
                const int ks = part * UK + s;
                if constexpr (PACKED) {
                    const bf16x8 xp = *reinterpret_cast<const bf16x8 *>(ub + s * 1024);
                    if constexpr (QB == 1) {   // two chains over alternate K steps: back-to-back products stay independent
                        acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[0][ks], xp, acc[s & 1], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[b][ks], xp, acc[b], 0, 0, 0);
                    }
                    continue;
                }
                float v[8];
                *reinterpret_cast<float4 *>(&v[0]) = *reinterpret_cast<const float4 *>(ub + (2 * s) * 1024);
                *reinterpret_cast<float4 *>(&v[4]) = *reinterpret_cast<const float4 *>(ub + (2 * s + 1) * 1024);
                bf16x8 xh, xl;
                fs_split(v, xh, xl);
                if constexpr (QB == 1) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[0][ks], xh, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[0][ks], xl, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql[0][ks], xh, acc[0], 0, 0, 0);
                } else {
#pragma unroll
                    for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[b][ks], xh, acc[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[b][ks], xl, acc[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql[b][ks], xh, acc[b], 0, 0, 0);
                }
            }
        }
        const uint32_t pos = (uint32_t)(i & gmask);
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float t = QB == 1 ? acc[0][e] + acc[1][e] : acc[b][e];
                const float key = __uint_as_float((__float_as_uint(t) & ~((1u << FS_POS_BITS) - 1)) | pos);
                second[b][e] = __builtin_amdgcn_fmed3f(best[b][e], second[b][e], key);
                best[b][e] = fmaxf(best[b][e], key);
            }
        if (pos == (uint32_t)gmask || i == my_tiles - 1) {
            const int g = i >> glog;
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int qi = 32 * b + (e & 3) + 8 * (e >> 2) + 4 * lk;
                    if (qi < nq) gb[(((int64_t)qi * NG + g) * FS_WAVES + wave_g) * 32 + lj] = make_float2(best[b][e], second[b][e]);
                }
            float fold[V];
#pragma unroll
            for (int v = 0; v < V; ++v) fold[v] = (v >> 4) < QB ? best[(v >> 4) < QB ? (v >> 4) : 0][v & 15] : FS_PAD_BIAS;
            fs_fold_max<16, V, V>(fold, lj);
#pragma unroll
            for (int r = 0; r < R; ++r) wmax[r] = fmaxf(wmax[r], fold[r]);
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) best[b][e] = second[b][e] = FS_PAD_BIAS;
        }
    }
    if ((lj & ((1 << (5 - T)) - 1)) == 0) {
        const int pfx = lj >> (5 - T);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int v = pfx * R + r, qi = 32 * (v >> 4) + (v & 3) + 8 * ((v & 15) >> 2) + 4 * lk;
            if (qi < nq) wm[(int64_t)qi * FS_WAVES + wave_g] = wmax[r];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- batches beyond one wave's registers: the four waves of a workgroup hold DIFFERENT queries and share the row tiles -------
// One workgroup per CU streams tiles t = block + 256 i through ONE ring in LDS (RU whole tiles, 128 KB): every wave requests a
// quarter of a tile's pieces, one s_barrier per tile says "tile i has landed for everyone and tile i - 1 has been read by
// everyone", the request for tile i + RU - 1 goes into the slot tile i - 1 leaves.  A wave scores the tile against ITS 32 QB
// queries, so a pass serves 128 QB queries (384 at D = 128) per read of the rows and the matrix pipe, not HBM, sets the pace.
// Groups, (best, second) entries and the finish are those of the kernel above with 256 row streams instead of 1024; the first
// threshold's 1024 maxima are per (workgroup, lane & 3): the lanes are folded down to four classes, disjoint row sets.
constexpr int FSS_STREAMS = 256;
template <int NCH, int NW> struct FssGeom {
    static constexpr int TILE = NCH * 2048;                    // bytes
    static constexpr int RU = 131072 / TILE;                   // tiles in the ring
    static constexpr int RB = RU + 1;                          // bias slots
    static constexpr int LDS = RU * TILE + RB * 256;
    static constexpr int PW = 2 * NCH / NW;                    // pieces a wave requests per tile (even: whole K steps)
    static constexpr int OPS = PW + 1;
};
// score key = t with its low FS_POS_BITS bits replaced by pos (one v_bfi_b32)
__device__ __forceinline__ float fs_key(float t, uint32_t pos)
{
    constexpr uint32_t m = (1u << FS_POS_BITS) - 1;
    return __uint_as_float((pos & m) | (__float_as_uint(t) & ~m));
}
// max without the canonicalising v_max the compiler puts in front of fmaxf (the operands are never signalling NaNs)
__device__ __forceinline__ float fs_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// NW waves of 32 QB queries each, the first NF of them also request and convert the tiles (NF divides the 2 NCH pieces of a
// tile into whole K steps).  <QB, 4, 4>: one wave per SIMD with up to 96 queries in its registers.  <1, 12, NCH>: three waves
// per SIMD with 32 queries each -- a wave issues a vector instruction every ~5 cycles at best, so the vector work of a tile
// (folding the scores into best / second, the conversion) only disappears behind the matrix instructions when several
// waves share the SIMD.  Measured per pass on 1 M x 128-d: one wave per SIMD 254 us (256 queries) / 340 us (384), two waves per
// SIMD in lock step 210 us (256), two waves alternating between a matrix and a vector phase 254 us, two waves with separate
// jobs (four multiply, four feed) 280 us, one wave with the vector work pinned between its matrix instructions by
// sched_group_barrier 290 us, three waves per SIMD (this form) 265 us for 384 queries = 0.69 us per query against 0.82-0.89.
template <int NCH, int QB, int NW, int NF>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void flat_f32_mshare_kernel(
    const float *__restrict__ X, const float *__restrict__ bias, int64_t n_tiles, const float *__restrict__ Q, int nq, int G, int NG,
    float2 *__restrict__ gb, float *__restrict__ wm, uint32_t *__restrict__ zero_a, uint32_t *__restrict__ zero_b, int dbg)
{
    using Ge = FssGeom<NCH, NF>;
    const int nt = dbg & 16;   // (not a timing experiment: the non-temporal hint of fs_glds16)
    constexpr int D = 16 * NCH, TILE = Ge::TILE, RU = Ge::RU, RB = Ge::RB, PW = Ge::PW, OPS = Ge::OPS;
    static_assert(PW >= 2 && PW % 2 == 0, "a wave converts whole K steps");
    constexpr bool TWO_CHAINS = QB == 1 && NW <= 8;          // one wave per SIMD: two chains so that back-to-back products are independent
    constexpr int NACC = TWO_CHAINS ? 2 : QB;
    extern __shared__ __attribute__((aligned(16))) uint8_t fs_ring[];   // [RU tiles | RB bias slots]
    const int tid = threadIdx.x, lane = tid & 63, lj = lane & 31, lk = lane >> 5;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool computes = true, feeds = wave_id < NF;
    const int wave = wave_id;
    if (blockIdx.x == 0)
        for (int i = tid; i < nq; i += 64 * NW) zero_a[i] = zero_b[i] = 0u;
    bf16x8 qh[QB][NCH], ql[QB][NCH];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qi = (wave * QB + b) * 32 + lj;
        const float *qp = Q + (int64_t)(qi < nq ? qi : nq - 1) * D + 8 * lk;
#pragma unroll
        for (int s = 0; s < NCH; ++s) {
            float v[8];
            *reinterpret_cast<float4 *>(&v[0]) = *reinterpret_cast<const float4 *>(qp + 16 * s);
            *reinterpret_cast<float4 *>(&v[4]) = *reinterpret_cast<const float4 *>(qp + 16 * s + 4);
            fs_split(v, qh[b][s], ql[b][s]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int64_t stream = blockIdx.x;
    const int my_tiles = stream < n_tiles ? (int)((n_tiles - stream + FSS_STREAMS - 1) / FSS_STREAMS) : 0;
    const uint32_t ring_b = (uint32_t)(uintptr_t)fs_ring;
    const uint32_t bias_b = ring_b + (uint32_t)(RU * TILE);
    // this wave's share of tile i: pieces p = wave PW .. + PW, piece p = 2 s + h as in the kernel above
    auto request = [&](int i) __attribute__((always_inline)) {
        const int ic = i < my_tiles ? i : my_tiles - 1;
        const int64_t t = stream + (int64_t)ic * FSS_STREAMS;
        const float *tile = X + ((t >> 1) * (int64_t)(D / 4) * 64 + (t & 1) * 32 + lj) * 4;
        // (past the end: the last tile again, into the slot that is free anyway -- never the slot of a converted tile)
        const uint32_t dst = ring_b + (uint32_t)((i % RU) * TILE);
#pragma unroll
        for (int pp = 0; pp < PW; ++pp) {
            const int p = wave * PW + pp, s = p >> 1, h = p & 1;
            fs_glds16(tile + (int64_t)(4 * s + 2 * lk + h) * 256, dst + (uint32_t)(p * 1024), nt);
        }
        fs_glds4(bias + t * 32 + lj, bias_b + (uint32_t)((i % RB) * 256));
    };
    // Operand conversion happens ONCE per workgroup: the pieces a wave requested are exactly K steps wave PW/2 .. of the tile,
    // so the wave that fetched them splits them into the two bf16 terms IN PLACE (piece 2 s keeps the high terms of its lanes,
    // piece 2 s + 1 the low terms: same bytes) as soon as its own requests have landed -- one tile ahead of the products, no
    // extra barrier -- and the product loop of every wave reads ready-made operands.
    auto convert = [&](int i) __attribute__((always_inline)) {   // own share of tile i (requests retired: the caller waited)
        uint8_t *tb = fs_ring + (size_t)(i % RU) * TILE + lane * 16;
#pragma unroll
        for (int ss = 0; ss < PW / 2; ++ss) {
            const int s = wave * (PW / 2) + ss;
            float v[8];
            *reinterpret_cast<float4 *>(&v[0]) = *reinterpret_cast<const float4 *>(tb + (2 * s) * 1024);
            *reinterpret_cast<float4 *>(&v[4]) = *reinterpret_cast<const float4 *>(tb + (2 * s + 1) * 1024);
            bf16x8 xh, xl;
            fs_split(v, xh, xl);
            *reinterpret_cast<bf16x8 *>(tb + (2 * s) * 1024) = xh;
            *reinterpret_cast<bf16x8 *>(tb + (2 * s + 1) * 1024) = xl;
        }
    };
    if (my_tiles > 0 && feeds) {
#pragma unroll
        for (int i = 0; i < RU - 1; ++i) request(i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RU - 2) * OPS) : "memory");
        convert(0);
    }
    constexpr int QBP = QB == 3 ? 4 : QB, V = 16 * QBP, R = V / 8;   // three fold steps: 32 lanes -> the 4 classes lane & 3
    float wmax[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wmax[r] = FS_PAD_BIAS;
    float best[QB][16], second[QB][16];
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) best[b][e] = second[b][e] = FS_PAD_BIAS;
    f32x16 acc[NACC];
    const int gmask = G - 1, glog = 31 - __builtin_clz(G);
    // the products of tile i (operands ready in LDS) into acc
    auto products = [&](int i) __attribute__((always_inline)) {
        const uint8_t *ub = fs_ring + (size_t)(i % RU) * TILE + lane * 16;
        {
            const float bx = reinterpret_cast<const float *>(fs_ring + RU * TILE + (size_t)(i % RB) * 256)[lj];
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][e] = (TWO_CHAINS && a == 1) ? 0.0f : bx;
        }
        // operands of K step s + 1 are read while the products of K step s run (the order is pinned below)
        bf16x8 xh[2], xl[2];
        xh[0] = *reinterpret_cast<const bf16x8 *>(ub);
        xl[0] = *reinterpret_cast<const bf16x8 *>(ub + 1024);
#pragma unroll
        for (int s = 0; s < NCH; ++s) {
            if (s + 1 < NCH) {
                xh[(s + 1) & 1] = *reinterpret_cast<const bf16x8 *>(ub + (2 * s + 2) * 1024);
                xl[(s + 1) & 1] = *reinterpret_cast<const bf16x8 *>(ub + (2 * s + 3) * 1024);
            }
            const bf16x8 ch = xh[s & 1], cl = xl[s & 1];
            if constexpr (TWO_CHAINS) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[0][s], ch, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[0][s], cl, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql[0][s], ch, acc[0], 0, 0, 0);
            } else {
#pragma unroll
                for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[b][s], ch, acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[b][s], cl, acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql[b][s], ch, acc[b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int s = 0; s < NCH; ++s) {
            if (s + 1 < NCH) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * QB, 0);
        }
    };
    // the scores of tile i into the group's best / second
    float tv[QB][16];
    auto take = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) tv[b][e] = TWO_CHAINS ? acc[0][e] + acc[1][e] : acc[b][e];
    };
    auto fold_in = [&](int i) __attribute__((always_inline)) {
        const uint32_t pos = (uint32_t)(i & gmask);
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float key = fs_key(tv[b][e], pos);
                second[b][e] = __builtin_amdgcn_fmed3f(best[b][e], second[b][e], key);
                best[b][e] = fs_max(best[b][e], key);
            }
    };
    // at the end of a group the entries leave
    auto group_end = [&](int i) __attribute__((always_inline)) {
        const uint32_t pos = (uint32_t)(i & gmask);
        if (pos == (uint32_t)gmask || i == my_tiles - 1) {
            const int g = i >> glog;
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int qi = 32 * (wave * QB + b) + (e & 3) + 8 * (e >> 2) + 4 * lk;
                    if (qi < nq) gb[(((int64_t)qi * NG + g) * FSS_STREAMS + stream) * 32 + lj] = make_float2(best[b][e], second[b][e]);
                }
            float fold[V];
#pragma unroll
            for (int v = 0; v < V; ++v) fold[v] = (v >> 4) < QB ? best[(v >> 4) < QB ? (v >> 4) : 0][v & 15] : FS_PAD_BIAS;
            // three halving steps (lane bits 4, 3, 2): lane sub keeps the R values of prefix sub >> 2, maxima over its class sub & 3
            {
                const bool h4 = (lj & 16) != 0, h3 = (lj & 8) != 0, h2 = (lj & 4) != 0;
#pragma unroll
                for (int i2 = 0; i2 < V / 2; ++i2) {
                    const float keep = h4 ? fold[i2 + V / 2] : fold[i2], send = h4 ? fold[i2] : fold[i2 + V / 2];
                    fold[i2] = fmaxf(keep, __shfl_xor(send, 16, 64));
                }
#pragma unroll
                for (int i2 = 0; i2 < V / 4; ++i2) {
                    const float keep = h3 ? fold[i2 + V / 4] : fold[i2], send = h3 ? fold[i2] : fold[i2 + V / 4];
                    fold[i2] = fmaxf(keep, __shfl_xor(send, 8, 64));
                }
#pragma unroll
                for (int i2 = 0; i2 < V / 8; ++i2) {
                    const float keep = h2 ? fold[i2 + V / 8] : fold[i2], send = h2 ? fold[i2] : fold[i2 + V / 8];
                    fold[i2] = fmaxf(keep, __shfl_xor(send, 4, 64));
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) wmax[r] = fmaxf(wmax[r], fold[r]);
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) best[b][e] = second[b][e] = FS_PAD_BIAS;
        }
    };
    auto update = [&](int i) __attribute__((always_inline)) {
        take();
        fold_in(i);
        group_end(i);
    };
    // the request for tile i + RU - 1 (into the slot tile i - 1 has left) and the conversion of the own share of tile i + 1
    FSS_T0();
    auto feed = [&](int i) __attribute__((always_inline)) {
        if (!(dbg & 1)) request(i + RU - 1);
        FSS_T(3);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RU - 2) * OPS) : "memory");   // own share of tile i + 1 (requested RU - 2 tiles ago) has landed
        FSS_T(4);
        if (i + 1 < my_tiles && !(dbg & 8)) convert(i + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FSS_T(5);
    };
    // All waves move in lock step, one barrier per tile.  (Measured on 1 M x 128-d, 256 queries per pass: eight waves of 32
    // queries 210 us per pass, four waves of 64 queries 254 us; the matrix instructions of a pass take 89 us at the pipe's
    // rate -- the vector work of a tile (folding 16 scores per 32 queries into best / second, the operand conversion, the
    // requests) does not run beside them: letting the two waves of a SIMD alternate between a matrix phase and a vector phase
    // (two barriers per tile) measured 254 us, a single wave per SIMD with the vector work pinned between its matrix
    // instructions by sched_group_barrier 290 us.)
    // The waves of one SIMD (w, w + 4, w + 8) do not walk the phases of a tile in the same order: the first feeds (requests +
    // conversion), then multiplies and folds; the others multiply and fold first -- one of them feeds afterwards.  One barrier
    // per tile for all of them.
    const bool feed_first = NW <= 4 || (wave_id >> 2) == 0;
    for (int i = 0; i < my_tiles; ++i) {
        // everyone's converted share of tile i is in LDS, everyone has read tile i - 1
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        FSS_T(0);
        if (feeds && feed_first) feed(i);
        if (!(dbg & 2)) products(i);
        FSS_T(1);
        if (!(dbg & 4)) update(i);
        FSS_T(2);
        if (feeds && !feed_first) feed(i);
    }
    FSS_TEND();
    if (computes) {
        const int pfx = lj >> 2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int v = pfx * R + r, qi = 32 * (wave * QB + (v >> 4)) + (v & 3) + 8 * ((v & 15) >> 2) + 4 * lk;
            if ((v >> 4) < QB && qi < nq) wm[(int64_t)qi * FS_WAVES + stream * 4 + (lj & 3)] = wmax[r];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// bias[r] for rows [row0, row1) of the blocked layout (+ the padding rows up to the end of the last 64-row block, whose
// values are zeroed): -|x|^2 / 2 (L2) or 0 (inner product); stats[0] = max |x|^2 (bits), stats[1] = rows with a non-finite value
__global__ __launch_bounds__(kBlock) void flat_f32_bias_kernel(float *__restrict__ X, int D, int l2, int64_t row0, int64_t row1,
                                                               int64_t row_pad, float *__restrict__ bias, uint32_t *__restrict__ stats)
{
    const int64_t row = row0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= row_pad) return;
    float4 *xr = reinterpret_cast<float4 *>(X) + (row >> 6) * (int64_t)(D >> 2) * 64 + (row & 63);
    if (row >= row1) {
        for (int c = 0; c < D / 4; ++c) xr[(int64_t)c * 64] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        bias[row] = FS_PAD_BIAS;
        return;
    }
    float s = 0.0f;
    bool finite = true;
    for (int c = 0; c < D / 4; ++c) {
        const float4 v = xr[(int64_t)c * 64];
        const float w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            finite = finite && fabsf(w[e]) <= 3.0e38f;
            s = __fmaf_rn(w[e], w[e], s);
        }
    }
    bias[row] = l2 ? -0.5f * s : 0.0f;
    if (!finite || !(s <= 3.0e38f)) atomicAdd(&stats[1], 1u);
    else atomicMax(&stats[0], __float_as_uint(s));
}

constexpr int FSF_LIST = 2048;   // groups that may reach the first cut
constexpr int FSF_KEEP = 1024;   // rows that get an exact distance

struct FsFinishArgs {
    const float *X; int64_t n; int D;
    const float *Xr;             // row-major copy of the rows for the exact distances, or null (they are gathered from the blocked rows)
    const float *Q; int64_t nq; int k;
    const float2 *gb; const float *wm; int G, NG, S;
    int ns_log;                  // log2 of the row streams of the pass (1024 waves, or 256 workgroups of the shared-ring kernel)
    const uint32_t *stats;       // [0] max |x|^2
    const uint32_t *pstats;      // packed form (one product over the operand copy's first terms): [2] max |x - x1|^2; null otherwise
    uint32_t *cnt;               // [nq] listed groups (zeroed by the caller)
    float *qb;                   // [nq] Q = (|q|^2 + max |x|^2) * 1.001 of the query, [nq] its margin term W (collect -> finish)
    uint4 *list;                 // [nq][FSF_LIST] (best bits, second bits, entry, -)
    float *out_d; int64_t *out_i;
    uint32_t *redo;              // [nq]: 1 = the exact kernels must answer this query
};

template <bool IP>
__device__ __forceinline__ float fs_margin(float Qb, int D, float theta)
{
    float m = Qb * (D > 128 ? 0x1p-12f : 0x1p-13f);
    if (IP) m += (2.0f + fabsf(theta)) * 0x1p-20f;   // the rounding of 1 - sum in the reference
    return m;
}


// workgroup (q, s): theta1 = (about) the k-th largest of the query's 1024 wave maxima -- at least k distinct waves = k distinct
// rows reach it, so it bounds theta from below; every group of slice s whose best reaches theta1 - margin joins the query's list.
// Every wave derives theta1 for itself (registers only, no barrier).
template <bool IP, int LANES>
__global__ __launch_bounds__(kBlock) void flat_f32_stream_collect_kernel(const FsFinishArgs a)
{
    constexpr int U = 8;                    // entries per thread: one batch of loads in flight
    __shared__ uint4 hit_s[kBlock * U];     // this workgroup's hits (all of its entries may qualify when rows tie in masses)
    __shared__ __attribute__((aligned(16))) float q_s[256];
    __shared__ float cut_s;
    __shared__ int nh_s;
    __shared__ uint32_t base_s;
    const int64_t qi = blockIdx.x / a.S;
    const int sl = blockIdx.x % a.S, tid = threadIdx.x, lane = tid & 63;
    FS_T0();
    const int64_t E = ((int64_t)a.NG << a.ns_log) * 32;
    const int64_t e0 = E * sl / a.S, e1 = E * (sl + 1) / a.S;
    const float2 *gb = a.gb + qi * E;
    // the slice's first batch is requested before the threshold is known
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t e = e0 + tid + (int64_t)u * kBlock;
        v[u] = e < e1 ? gb[e] : make_float2(FS_PAD_BIAS, FS_PAD_BIAS);
    }
    if (tid < a.D) q_s[tid] = a.Q[qi * a.D + tid];   // D <= 256
    if (tid < 64) {   // wave 0: theta1 (registers only)
        uint32_t key[FS_WAVES / 64];
#pragma unroll
        for (int j = 0; j < FS_WAVES / 64; ++j) key[j] = f32_key(a.wm[qi * FS_WAVES + j * 64 + lane]);
        const float qq = fs_qnorm(a.Q + qi * a.D, a.D);
        const float Qb = (qq + __uint_as_float(a.stats[0])) * 1.001f;
        float cut1 = __uint_as_float(0x7fc00000u);   // NaN: give up
        FS_T(0);
        if (Qb > 0x1p-60f && Qb < 0x1p60f) {   // (false for a non-finite query as well)
            const float theta1 = key_f32(fs_wave_select(key, a.k, a.k + a.k / 4 + 8));
            if (theta1 > FS_EMPTY) {           // else: fewer than k waves saw a row, not this kernel's case
                // packed form: what the omitted low terms can add up to for this query (flat_f32_tfilter.hip, NPROD 1)
                float w = 0.0f;
                if (a.pstats) w = (sqrtf(__uint_as_float(a.stats[0]) * fs_qlow(a.Q + qi * a.D, a.D)) + sqrtf(__uint_as_float(a.pstats[2]) * qq)) * 1.01f;
                cut1 = theta1 - fs_margin<IP>(Qb, a.D, theta1) - 2.0f * w;
                if (sl == 0 && lane == 0) { a.qb[qi] = Qb; a.qb[a.nq + qi] = w; }
            }
        }
        if (lane == 0) {
            cut_s = cut1;
            nh_s = 0;
            if (!(cut1 == cut1) && sl == 0) a.redo[qi] = 1u;
        }
        FS_T(1);
    }
    __syncthreads();
    const float cut1 = cut_s;
    if (!(cut1 == cut1)) return;
    uint4 *list = a.list + qi * FSF_LIST;
    // hits go to LDS as they are found (no barrier between batches); more than the list could take anyway: the exact kernels answer
    for (int64_t base = e0 + tid; base < e1; base += (int64_t)kBlock * U) {
        if (base != e0 + tid) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t e = base + (int64_t)u * kBlock;
                v[u] = e < e1 ? gb[e] : make_float2(FS_PAD_BIAS, FS_PAD_BIAS);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (v[u].x >= cut1) {
                const int slot = atomicAdd(&nh_s, 1);
                if (slot < kBlock * U) hit_s[slot] = make_uint4(__float_as_uint(v[u].x), __float_as_uint(v[u].y), (uint32_t)(base + (int64_t)u * kBlock), 0u);
            }
    }
    __syncthreads();
    const int nh = nh_s;
    if (nh > kBlock * U) {
        if (tid == 0) a.redo[qi] = 1u;
        return;
    }
    if (tid == 0 && nh > 0) base_s = atomicAdd(&a.cnt[qi], (uint32_t)nh);   // one global atomic per workgroup
    __syncthreads();
    // the exact distance of every hit's best row rides along (the hits are spread over many workgroups here: their rows'
    // pieces are fetched side by side; the finish only computes for the rare group whose second best qualifies too)
    for (int i = tid; i < nh; i += kBlock) {
        if (base_s + i >= (uint32_t)FSF_LIST) break;
        uint4 h = hit_s[i];
        const uint32_t e = h.z;
        const int64_t j = e & 31, wv = (e >> 5) & ((1u << a.ns_log) - 1), g = e >> (5 + a.ns_log), pp = h.x & ((1u << FS_POS_BITS) - 1);
        const int64_t row = (wv + ((g * a.G + pp) << a.ns_log)) * 32 + j;
        h.w = row < a.n ? __float_as_uint(fs_exact<IP, LANES>(a.Xr ? a.Xr : a.X, a.D, row, reinterpret_cast<const float4 *>(q_s), a.Xr != nullptr)) : 0x7fc00000u;
        list[base_s + i] = h;
    }
    FS_T(2);
}

// one workgroup per query: theta = the k-th largest listed best, cut = theta - margin, candidate rows, exact distances, the k
// smallest (distance, row) in order
template <bool IP, int LANES>
__global__ __launch_bounds__(kBlock) void flat_f32_stream_finish_kernel(const FsFinishArgs a)
{
    __shared__ unsigned long long sort_s[FSF_KEEP];   // (distance key) << 32 | row of a candidate
    __shared__ uint32_t row_s[FSF_KEEP];
    // the listed groups (best key, entry, second best, exact distance of the best row); later the query
    __shared__ __attribute__((aligned(16))) float work_s[4 * FSF_LIST];
    __shared__ int cnt_s;
    __shared__ uint32_t theta_s;
    uint32_t *best_s = reinterpret_cast<uint32_t *>(work_s), *ent_s = best_s + FSF_LIST;
    float *sec_s = work_s + 2 * FSF_LIST, *dist_s = work_s + 3 * FSF_LIST;
    const int64_t qi = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int D = a.D, k = a.k, G = a.G;
    const float *q = a.Q + qi * D;
    FS_T0();
    if (a.redo[qi]) return;   // (set by the collect kernel: workgroup-uniform)
    auto give_up = [&]() {
        if (tid == 0) a.redo[qi] = 1u;
    };
    const int nl = (int)a.cnt[qi];
    if (nl > FSF_LIST || nl < k) {   // the list ran over (nl >= k holds whenever theta1 was valid)
        give_up();
        return;
    }
    const float Qb = a.qb[qi];
    if (tid == 0) cnt_s = 0;
    const uint4 *list = a.list + qi * FSF_LIST;
    for (int i = tid; i < nl; i += kBlock) {
        const uint4 v = list[i];
        best_s[i] = f32_key(__uint_as_float(v.x));
        sec_s[i] = __uint_as_float(v.y);
        ent_s[i] = v.z;
        dist_s[i] = __uint_as_float(v.w);
    }
    for (int i = tid; i < k; i += kBlock) {   // fewer candidates than k (cannot happen for n >= k): the tail stays empty
        a.out_d[qi * k + i] = __uint_as_float(0x7f800000u);
        a.out_i[qi * k + i] = -1;
    }
    __syncthreads();
    FS_T(8);
    if (tid < 64) {   // theta: exact k-th largest, by one wave
        auto sel = [&](auto nk_tag) {
            constexpr int NK = decltype(nk_tag)::value;
            uint32_t key[NK];
#pragma unroll
            for (int j = 0; j < NK; ++j) key[j] = (j * 64 + lane < nl) ? best_s[j * 64 + lane] : 0u;
            const uint32_t c = fs_wave_select(key, k, k);
            if (lane == 0) theta_s = c;
        };
        if (nl <= 256) sel(std::integral_constant<int, 4>());
        else if (nl <= 512) sel(std::integral_constant<int, 8>());
        else sel(std::integral_constant<int, FSF_LIST / 64>());
    }
    __syncthreads();
    FS_T(9);
    const float theta = key_f32(theta_s);
    const float cut = theta - fs_margin<IP>(Qb, D, theta) - (a.pstats ? 2.0f * a.qb[a.nq + qi] : 0.0f);
    // candidates: the best row of a qualifying group comes with its exact distance; a group whose second best qualifies as well
    // has all its rows evaluated here
    __shared__ int ndone_s;
    if (tid == 0) ndone_s = 0;
    __syncthreads();
    auto dist_key = [](float d) {
        uint32_t dk = f32_key(d);
        return dk >= 0xfffffff0u ? 0xffffffefu : dk;   // (NaN patterns: keep the absent-row codes free)
    };
    for (int i = tid; i < nl; i += kBlock) {
        const float bestv = key_f32(best_s[i]);
        if (!(bestv >= cut)) continue;
        const uint32_t e = ent_s[i];
        const int64_t j = e & 31, wv = (e >> 5) & ((1u << a.ns_log) - 1), g = e >> (5 + a.ns_log);
        if (sec_s[i] >= cut) {
            const int base = atomicAdd(&cnt_s, G);
            for (int p = 0; p < G; ++p) {
                const int64_t row = (wv + ((g * G + p) << a.ns_log)) * 32 + j;
                if (base + p < FSF_KEEP) row_s[base + p] = row < a.n ? (uint32_t)row : 0xffffffffu;
            }
        } else {
            const uint32_t p = __float_as_uint(bestv) & ((1u << FS_POS_BITS) - 1);
            const int64_t row = (wv + ((g * G + p) << a.ns_log)) * 32 + j;
            const int slot = atomicAdd(&ndone_s, 1);
            if (slot < FSF_KEEP && row < a.n) sort_s[slot] = ((unsigned long long)dist_key(dist_s[i]) << 32) | (uint32_t)row;
            else if (slot < FSF_KEEP) sort_s[slot] = ~0ull - (unsigned)slot;
        }
    }
    __syncthreads();
    const int nrow = cnt_s, ndone = ndone_s, nc = nrow + ndone;
    if (nc > FSF_KEEP) {
        give_up();
        return;
    }
    FS_T(10);
    if (nrow > 0) {   // (workgroup-uniform)
        for (int e = tid; e < D; e += kBlock) work_s[e] = q[e];   // (the listed groups are not needed any more)
        __syncthreads();
        for (int i = tid; i < nrow; i += kBlock) {
            unsigned long long ekey = ~0ull - (unsigned)(ndone + i);   // an absent row: a key above every real one, distinct per slot
            if (row_s[i] != 0xffffffffu)
                ekey = ((unsigned long long)dist_key(fs_exact<IP, LANES>(a.Xr ? a.Xr : a.X, D, row_s[i], reinterpret_cast<const float4 *>(work_s), a.Xr != nullptr)) << 32) | row_s[i];
            sort_s[ndone + i] = ekey;
        }
        __syncthreads();
    }
    FS_T(11);
    // rank of every candidate among the (distinct) keys: its place in the output
    for (int i = tid; i < nc; i += kBlock) {
        const unsigned long long e = sort_s[i];
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < nc; ++j) rank += sort_s[j] < e ? 1 : 0;
        if (rank < k && (uint32_t)(e >> 32) < 0xfffffff0u) {
            a.out_d[qi * k + rank] = key_f32((uint32_t)(e >> 32));
            a.out_i[qi * k + rank] = (int64_t)(uint32_t)e;
        }
    }
    FS_T(12);
}

#ifdef CVTMI_FS_TIMING
extern "C" int cvtmi_debug_fs_timing(unsigned long long *out, int reset)
{
    unsigned long long z[16] = {};
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fs_dbg), sizeof z) != hipSuccess) return -3;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_fs_dbg), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
extern "C" int cvtmi_debug_fss_timing(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fss_dbg), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

}  // namespace

// ---- host side -----------------------------------------------------------------------------------------------------------
// D in {32, 64, 96, 128, 192, 256}; a wave holds 32 QB queries, QB the largest of 1..4 whose registers fit one wave per SIMD
// (8 NCH operand registers + 48 of accumulators / best / second per 32 queries).  A pass of the private-ring kernel serves
// 32 QB queries, a pass of the shared-ring kernel 128 QB.
static int fs_qb_max(int nch)
{
    int qb = 4;
    while (qb > 1 && qb * (8 * nch + 48) > 368) --qb;
    return qb;
}
static std::atomic<int> g_fs_dbgflags{0};     // timing experiments (results wrong when non-zero): cvtmi_set_tuning("flat_f32_dbg")
void set_flat_f32_dbg(int v) { g_fs_dbgflags = v; }
int get_flat_f32_dbg() { return g_fs_dbgflags.load(); }
static std::atomic<int> g_fs_packed{1};  // cvtmi_set_tuning("flat_f32_packed"): 1 = small batches stream the bf16 operand copy when there is one
void set_flat_f32_packed(int v) { g_fs_packed = v != 0; }
static std::atomic<int> g_fs_nt{1};      // cvtmi_set_tuning("flat_f32_nt"): 0 = never, 1 = choose, 2 = always
void set_flat_f32_nt(int v) { g_fs_nt = v; }
// non-temporal row loads: measured better wherever the stream kernel is bound by the rows (one query 0.119 -> 0.106 ms,
// 64 queries 0.147 -> 0.14, the shared ring 1.09 -> 1.07 at 1000), worse at three query blocks per wave (0.155 -> 0.165)
static bool fs_nt(bool shared, int qb) { return g_fs_nt == 2 || (g_fs_nt == 1 && (shared || qb <= 2)); }
static std::atomic<int> g_fs_share{0};   // cvtmi_set_tuning("flat_f32_share"): 0 = choose, 1 = four waves x 32 QB queries, 2 = FS_MANY waves x 32 queries
void set_flat_f32_share(int v) { g_fs_share = v; }
// the shared-ring kernel wants whole K steps per wave: D / 16 a multiple of the wave count
constexpr int FS_MANY = 12;   // waves of the many-wave form of the shared-ring kernel (three per SIMD)
// round 6, "flat_f32_share" 3: eight waves (two per SIMD) of 64 queries each -- 512 queries per pass instead of 384 (1000 queries: two
// passes over the rows instead of three)
static bool fs_wide() { return g_fs_share == 3; }
static int fs_many_queries() { return fs_wide() ? 8 * 64 : 32 * FS_MANY; }
// Round 6: the eight- / twelve-wave forms are no longer chosen, only asked for ("flat_f32_share" 2 / 3).  A width sweep against the
// exact kernels found ONE query of ~10^5 (2 M x 64-d rows, 1000 queries, L2: the row at rank 99 missing, tools/f32_tfilter_widths.py)
// that both of them answer wrongly and the four-wave form, the private rings and the threshold filter answer correctly -- not
// understood, so not trusted.  What is known: deterministic (the same query and row on every run, whatever the non-temporal hints), it
// needs the whole 2 M rows (the first 1.9 M: right) AND the query at its place in a pass of 334 or 500 queries -- the same query in the same
// wave of a pass of 200 queries is answered correctly, as it is in passes that start at query 60 or 64.  Batches of 16 queries and more over 262 144
// rows and more go through flat_f32_tfilter.hip anyway; what is left to the shared ring are large batches on smaller tables.
static bool fs_eight(int D) { return g_fs_share >= 2 && (D == 64 || D == 128); }
static bool fs_four(int D) { return (D / 16) % 4 == 0; }
int flat_f32_stream_qmax(int D)
{
    if (D != 32 && D != 64 && D != 96 && D != 128 && D != 192 && D != 256) return 0;
    return fs_eight(D) ? fs_many_queries() : (fs_four(D) ? 128 : 32) * fs_qb_max(D / 16);
}
static bool fs_shared(int D, int64_t nq) { return nq > 32 * fs_qb_max(D / 16); }
// the most queries a pass of the private-ring kernel takes (more go to the shared ring, up to flat_f32_stream_qmax)
int flat_f32_stream_private_max(int D) { return flat_f32_stream_qmax(D) > 0 ? 32 * fs_qb_max(D / 16) : 0; }
bool flat_f32_stream_applies(int metric, int D, int64_t n, int k)
{
    return (metric == CVTMI_METRIC_IP || metric == CVTMI_METRIC_L2F) && flat_f32_stream_qmax(D) > 0 && n >= 32768 && n < 0xffffffffLL &&
           k >= 1 && k <= 128 && (n + 31) / 32 < ((int64_t)1 << 17) * FSS_STREAMS * 32;   // entry index: 32 bits
}
// tiles per group and groups per row stream for n rows on `streams` streams
static void fs_groups(int64_t n, int streams, int *G, int *NG)
{
    const int64_t n_tiles = (n + 31) / 32, per = (n_tiles + streams - 1) / streams;
    int g = 32;
    while (g > 1 && g / 2 >= per) g >>= 1;   // short indexes: one group, no wider than the tiles a stream sees
    *G = g;
    *NG = (int)((per + g - 1) / g);
}
// scratch of one pass: group entries [nq][NG][streams][32] float2, 1024 maxima per query, lists [nq][FSF_LIST] uint4, Q per query
static size_t fs_gb_bytes(int D, int64_t n, int64_t nq_pass)
{
    int G, NG;
    const int streams = fs_shared(D, nq_pass) ? FSS_STREAMS : FS_WAVES;
    fs_groups(n, streams, &G, &NG);
    return (size_t)nq_pass * NG * streams * 32 * sizeof(float2);
}
size_t flat_f32_stream_scratch(int D, int64_t n, int64_t nq_pass)
{
    return fs_gb_bytes(D, n, nq_pass) + (size_t)nq_pass * (FS_WAVES + 4) * sizeof(float) + (size_t)nq_pass * FSF_LIST * sizeof(uint4);
}

int launch_flat_f32_bias(float *X, int D, int metric, int64_t row0, int64_t row1, float *bias, uint32_t *stats, hipStream_t st)
{
    const int64_t row_pad = (row1 + 63) / 64 * 64;
    if (row_pad <= row0) return CVTMI_OK;
    hipLaunchKernelGGL(flat_f32_bias_kernel, dim3((unsigned)((row_pad - row0 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, X, D,
                       metric == CVTMI_METRIC_L2F ? 1 : 0, row0, row1, row_pad, bias, stats);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

struct FsStreamArgs {
    const float *X, *bias; int64_t n_tiles; const float *q; int nq, G, NG; float2 *gb; float *wm; uint32_t *redo, *cnt;
};
template <int NCH>
static int fs_launch_eight(const FsStreamArgs &a, hipStream_t st)
{
    if constexpr (NCH == 4 || NCH == 8) {
        static std::atomic<bool> attr_set[16] = {};
        const size_t lds = FssGeom<NCH, NCH>::LDS;
        if (fs_wide()) {
            static std::atomic<bool> attr_set_w[16] = {};
            CVTMI_TRY(fs_set_lds((const void *)flat_f32_mshare_kernel<NCH, 2, 8, NCH>, lds, attr_set_w));
            hipLaunchKernelGGL((flat_f32_mshare_kernel<NCH, 2, 8, NCH>), dim3(FSS_STREAMS), dim3(64 * 8), lds, st, a.X, a.bias, a.n_tiles, a.q,
                               a.nq, a.G, a.NG, a.gb, a.wm, a.redo, a.cnt, g_fs_dbgflags | (fs_nt(true, 1) ? 16 : 0));
            CVTMI_HIP(hipGetLastError());
            return CVTMI_OK;
        }
        CVTMI_TRY(fs_set_lds((const void *)flat_f32_mshare_kernel<NCH, 1, FS_MANY, NCH>, lds, attr_set));
        hipLaunchKernelGGL((flat_f32_mshare_kernel<NCH, 1, FS_MANY, NCH>), dim3(FSS_STREAMS), dim3(64 * FS_MANY), lds, st, a.X, a.bias, a.n_tiles, a.q,
                           a.nq, a.G, a.NG, a.gb, a.wm, a.redo, a.cnt, g_fs_dbgflags | (fs_nt(true, 1) ? 16 : 0));
        CVTMI_HIP(hipGetLastError());
        return CVTMI_OK;
    } else {
        return fail(CVTMI_EINVAL, "flat_f32_stream: eight waves at D=%d", 16 * NCH);
    }
}
template <int NCH, int QB>
static int fs_launch_packed_qb(const FsStreamArgs &a, const void *pack, hipStream_t st)
{
    static std::atomic<bool> attr_set[16] = {};
    const size_t lds = (size_t)4 * FsGeom<NCH, true>::WAVE_LDS;
    CVTMI_TRY(fs_set_lds((const void *)flat_f32_mstream_kernel<NCH, QB, true>, lds, attr_set));
    hipLaunchKernelGGL((flat_f32_mstream_kernel<NCH, QB, true>), dim3(FS_BLOCKS), dim3(256), lds, st, reinterpret_cast<const float *>(pack), a.bias, a.n_tiles, a.q, a.nq,
                       a.G, a.NG, a.gb, a.wm, a.redo, a.cnt, fs_nt(false, QB) ? 1 : 0);
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}
template <int NCH>
static int fs_launch_packed(int qb, const FsStreamArgs &a, const void *pack, hipStream_t st)
{
    if constexpr (3 * (8 * NCH + 48) <= 368) {
        if (qb == 3) return fs_launch_packed_qb<NCH, 3>(a, pack, st);
    }
    if (qb == 2) return fs_launch_packed_qb<NCH, 2>(a, pack, st);
    return fs_launch_packed_qb<NCH, 1>(a, pack, st);
}
template <int NCH, int QB, bool SHARED>
static int fs_launch_stream(const FsStreamArgs &a, hipStream_t st)
{
    static std::atomic<bool> attr_set[16] = {};
    if constexpr (SHARED && NCH % 4 != 0) {
        return fail(CVTMI_EINVAL, "flat_f32_stream: shared ring at D=%d", 16 * NCH);
    } else if constexpr (SHARED) {
        const size_t lds = FssGeom<NCH, 4>::LDS;
        CVTMI_TRY(fs_set_lds((const void *)flat_f32_mshare_kernel<NCH, QB, 4, 4>, lds, attr_set));
        hipLaunchKernelGGL((flat_f32_mshare_kernel<NCH, QB, 4, 4>), dim3(FSS_STREAMS), dim3(256), lds, st, a.X, a.bias, a.n_tiles, a.q, a.nq, a.G, a.NG,
                           a.gb, a.wm, a.redo, a.cnt, g_fs_dbgflags | (fs_nt(true, QB) ? 16 : 0));
    } else {
        const size_t lds = (size_t)4 * FsGeom<NCH>::WAVE_LDS;
        CVTMI_TRY(fs_set_lds((const void *)flat_f32_mstream_kernel<NCH, QB>, lds, attr_set));
        hipLaunchKernelGGL((flat_f32_mstream_kernel<NCH, QB>), dim3(FS_BLOCKS), dim3(256), lds, st, a.X, a.bias, a.n_tiles, a.q, a.nq, a.G, a.NG,
                           a.gb, a.wm, a.redo, a.cnt, fs_nt(false, QB) ? 1 : 0);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}
template <int NCH, bool SHARED>
static int fs_launch_qb(int qb, const FsStreamArgs &a, hipStream_t st)
{
    if constexpr (4 * (8 * NCH + 48) <= 368) {
        if (qb == 4) return fs_launch_stream<NCH, 4, SHARED>(a, st);
    }
    if constexpr (3 * (8 * NCH + 48) <= 368) {
        if (qb == 3) return fs_launch_stream<NCH, 3, SHARED>(a, st);
    }
    if (qb == 2) return fs_launch_stream<NCH, 2, SHARED>(a, st);
    if (qb == 1) return fs_launch_stream<NCH, 1, SHARED>(a, st);
    return fail(CVTMI_EINVAL, "flat_f32_stream: %d query blocks per wave at D=%d", qb, 16 * NCH);
}

// one pass: nq <= flat_f32_stream_qmax(D) queries against rows [0, n); results for queries whose redo flag stays 0.
// redo[nq] and cnt[nq] are zeroed inside
int launch_flat_f32_stream(int metric, int D, const float *X, const float *bias, const uint32_t *stats, int64_t n, const float *q, int64_t nq,
                           int k, void *scratch, float *out_d, int64_t *out_i, uint32_t *redo, uint32_t *cnt, hipStream_t st, const void *pack,
                           const uint32_t *pstats, const float *Xrows)
{
    const int qmax = flat_f32_stream_qmax(D);
    if (qmax == 0 || nq < 1 || nq > qmax) return fail(CVTMI_EINVAL, "flat_f32_stream: D=%d nq=%lld", D, (long long)nq);
    const bool shared = fs_shared(D, nq);
    const int streams = shared ? FSS_STREAMS : FS_WAVES;
    int G, NG;
    fs_groups(n, streams, &G, &NG);
    float2 *gb = reinterpret_cast<float2 *>(scratch);
    float *wm = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(scratch) + fs_gb_bytes(D, n, nq));
    uint4 *list = reinterpret_cast<uint4 *>(wm + (size_t)nq * FS_WAVES);
    float *qbuf = reinterpret_cast<float *>(list + (size_t)nq * FSF_LIST);
    const FsStreamArgs sa = { X, bias, (n + 31) / 32, q, (int)nq, G, NG, gb, wm, redo, cnt };
    const int qb = shared ? (int)((nq + 127) / 128) : (int)((nq + 31) / 32);
    const bool eight = shared && fs_eight(D);
    // round 6: up to 32 queries over the bf16 operand copy of the rows when the handle keeps one (half the bytes, one product per term)
    const bool packed = pack != nullptr && pstats != nullptr && !shared && qb <= 3 && g_fs_packed.load() != 0;
#define CVTMI_FS(NCH_) \
    case NCH_: CVTMI_TRY(packed ? fs_launch_packed<NCH_>(qb, sa, pack, st) : eight ? fs_launch_eight<NCH_>(sa, st) : shared ? (fs_launch_qb<NCH_, true>(qb, sa, st)) : (fs_launch_qb<NCH_, false>(qb, sa, st))); break;
    switch (D / 16) {
        CVTMI_FS(2) CVTMI_FS(4) CVTMI_FS(6) CVTMI_FS(8) CVTMI_FS(12) CVTMI_FS(16)
        default: return fail(CVTMI_EUNSUPPORTED, "flat_f32_stream: D=%d", D);
    }
#undef CVTMI_FS
    FsFinishArgs fa;
    fa.X = X; fa.Xr = Xrows; fa.n = n; fa.D = D; fa.Q = q; fa.nq = nq; fa.k = k; fa.gb = gb; fa.wm = wm; fa.G = G; fa.NG = NG; fa.stats = stats;
    fa.pstats = packed ? pstats : nullptr;
    fa.ns_log = shared ? 8 : 10;
    fa.cnt = cnt; fa.list = list; fa.qb = qbuf; fa.out_d = out_d; fa.out_i = out_i; fa.redo = redo;
    // slices per query: one batch of loads per thread (2048 entries per slice) while that keeps the grid near one round of workgroups
    const int64_t E = (int64_t)NG * streams * 32;
    fa.S = (int)std::min<int64_t>(std::max<int64_t>(1, 1024 / nq), std::max<int64_t>(1, E / 2048));
    const unsigned cgrid = (unsigned)(nq * fa.S);
    if (metric == CVTMI_METRIC_IP) {
        hipLaunchKernelGGL((flat_f32_stream_collect_kernel<true, 4>), dim3(cgrid), dim3(kBlock), 0, st, fa);
        hipLaunchKernelGGL((flat_f32_stream_finish_kernel<true, 4>), dim3((unsigned)nq), dim3(kBlock), 0, st, fa);
    } else {
        hipLaunchKernelGGL((flat_f32_stream_collect_kernel<false, 8>), dim3(cgrid), dim3(kBlock), 0, st, fa);
        hipLaunchKernelGGL((flat_f32_stream_finish_kernel<false, 8>), dim3((unsigned)nq), dim3(kBlock), 0, st, fa);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

}  // namespace cvtmi
