// block_topk.h -- streaming "k smallest (key, payload)" selection shared by one workgroup.
//
// Semantics to reproduce: std::partial_sort_copy on pair<float,uint> (opq/src/common.h:25-37)
// and the max-heap of pair<dist,label> in BruteforceSearch::searchKnn
// (brute_force_search/src/brutoforce.hpp:73-93): the k smallest pairs in LEXICOGRAPHIC
// (distance, id) order.
//
// Design (MI355X): candidates are 64-bit words (ordered-distance key << 32 | payload) appended to a
// per-query LDS buffer with one LDS atomic per candidate.  A candidate is appended only if its key
// beats the current k-th key ("thr"), so after a short warm-up almost no row reaches the buffer and
// the scan kernels stay bound by their data path, not by selection.  When a buffer passes TRIG
// entries the workgroup sorts it in LDS (bitonic network, all comparators ascending so the
// power-of-two padding never has to be stored), keeps the k smallest and tightens thr.
//
// Tie rule: payloads must grow with the id order and tiles must be fed in ascending payload order.
// Then a later candidate whose key EQUALS thr can never displace the current k-th entry (it has a
// larger id), so the fast path compares with '<'.  Only candidates of the tile that overflowed the
// buffer are retried with '<=', because those may precede entries already stored.
// Kernels that push under a LOOSER bound than thr (adc_scan16: approximate keys) are exempt from the
// ordering rule: every tie reaches the buffer and the sort of exact (key, payload) words decides.
#pragma once
#include "common.h"

namespace cvtmi {

template <int QT, int CAP>
struct TopKShared {
    unsigned long long buf[QT][CAP];
    int cnt[QT];
    int exact_n[QT];    // entries [0, exact_n) carry final keys; later ones may still need Fix
    uint32_t thr[QT];   // key of the current k-th entry (KEY_MAX while fewer than k are known)
    uint32_t thr_x[QT]; // ThrX(thr): the bound the fast path compares with
    int flag[4];        // [0..2] rotating "compaction wanted" flags, [3] "some candidate still pending"
};

// Hooks for kernels whose fast path pushes candidates under an APPROXIMATE key (adc_scan16):
//   Fix   rewrites a freshly pushed entry with its exact key before the sort;
//   ThrX  maps the exact k-th key to the (looser) bound the approximate keys are compared with.
struct NoFix {
    static constexpr bool enabled = false;
    __device__ __forceinline__ unsigned long long operator()(int, unsigned long long e) const { return e; }
};
struct IdThr {
    __device__ __forceinline__ uint32_t operator()(int /*q*/, uint32_t t) const { return t; }
};

template <int QT, int CAP>
__device__ __forceinline__ void topk_init(TopKShared<QT, CAP> &s)
{
    if (threadIdx.x < QT) {
        s.cnt[threadIdx.x] = 0;
        s.exact_n[threadIdx.x] = 0;
        s.thr[threadIdx.x] = KEY_MAX;
        s.thr_x[threadIdx.x] = KEY_MAX;
    }
    if (threadIdx.x < 4) s.flag[threadIdx.x] = 0;
}

// Append one candidate.  Returns false when the buffer was full (caller keeps it pending).
template <int QT, int CAP, int TRIG>
__device__ __forceinline__ bool topk_push(TopKShared<QT, CAP> &s, int q, uint32_t key, uint32_t payload,
                                          bool &want_compact)
{
    const int pos = atomicAdd(&s.cnt[q], 1);
    if (pos >= TRIG) want_compact = true;
    if (pos < CAP) {
        s.buf[q][pos] = ((unsigned long long)key << 32) | payload;
        return true;
    }
    return false;
}

__device__ __forceinline__ void lds_cmpx(unsigned long long *b, int i, int j)
{
    const unsigned long long x = b[i], y = b[j];
    if (x > y) {
        b[i] = y;
        b[j] = x;
    }
}

// Sort every query's buffer, keep the k smallest entries, refresh thr.  Must be called by all NT
// threads after a barrier that made the pushes visible; ends with a barrier.
template <int QT, int CAP, int NT = kBlock, class Fix = NoFix, class ThrX = IdThr>
__device__ void topk_compact(TopKShared<QT, CAP> &s, int k, const Fix &fix = Fix(), const ThrX &thrx = ThrX())
{
    constexpr int NW = NT / 64;
    constexpr int G = QT < NW ? QT : NW;   // queries sorted concurrently
    constexpr int TPQ = NT / G;            // threads cooperating on one query
    constexpr int ROUNDS = (QT + G - 1) / G;

    // uniform network size: smallest power of two covering the fullest buffer
    int nmax = 0;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        int c = s.cnt[q];
        c = c < CAP ? c : CAP;
        nmax = c > nmax ? c : nmax;
    }
    int np = 2;
    while (np < nmax) np <<= 1;
    const int lim = np < CAP ? np : CAP;  // stored slots that take part

    const int grp = threadIdx.x / TPQ, t = threadIdx.x % TPQ;
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int q = rd * G + grp;
        const bool act = q < QT;
        unsigned long long *b = s.buf[act ? q : 0];
        int n = 0;
        if (act) {
            n = s.cnt[q];
            n = n < CAP ? n : CAP;
            if constexpr (Fix::enabled)
                for (int i = s.exact_n[q] + t; i < n; i += TPQ) b[i] = fix(q, b[i]);
            for (int i = n + t; i < lim; i += TPQ) b[i] = ~0ull;
        }
        __syncthreads();
        for (int size = 2; size <= np; size <<= 1) {
            const int half = size >> 1;
            if (act) {
                for (int c = t; c < (np >> 1); c += TPQ) {
                    const int blk = c / half, off = c - blk * half;
                    const int i = blk * size + off, j = blk * size + size - 1 - off;
                    if (j < lim) lds_cmpx(b, i, j);
                }
            }
            __syncthreads();
            for (int stride = half >> 1; stride >= 1; stride >>= 1) {
                if (act) {
                    for (int c = t; c < (np >> 1); c += TPQ) {
                        const int blk = c / stride, off = c - blk * stride;
                        const int i = blk * 2 * stride + off, j = i + stride;
                        if (j < lim) lds_cmpx(b, i, j);
                    }
                }
                __syncthreads();
            }
        }
        if (act && t == 0) {
            s.cnt[q] = n < k ? n : k;
            s.exact_n[q] = n < k ? n : k;
            const uint32_t th = (n >= k) ? (uint32_t)(b[k - 1] >> 32) : KEY_MAX;
            s.thr[q] = th;
            s.thr_x[q] = thrx(q, th);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Barrier-free compaction for buffers of at most 256 entries: ONE WAVE per query keeps the buffer in
// registers (entry idx = r*64 + lane, r < 4) and sorts it with a bitonic network whose exchanges at
// distance >= 64 are register swaps and below 64 lane shuffles -- no LDS traffic, no s_barrier inside.
// FixB(q, e[4], need[4]) rewrites, in one batch, the entries that still carry approximate keys.
// Same contract as topk_compact: call after a barrier that made the pushes visible; ends with a barrier.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, mask);
    hi = __shfl_xor(hi, mask);
    return ((unsigned long long)hi << 32) | lo;
}

// ascending bitonic sort of the 64 * NR entries a wave holds in registers (entry 64 r + lane in e[r]); NR = 1, 2, 4
template <int NR>
__device__ __forceinline__ void wave_bitonic_sort(unsigned long long (&e)[NR])
{
    static_assert(NR == 1 || NR == 2 || NR == 4, "register sort: 64, 128 or 256 entries");
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 2; k <= 64 * NR; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j >= 64) {
                const int dr = j >> 6;  // partner register r ^ dr of the same lane
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int pr = r ^ dr;
                    if (pr > r) {
                        const bool asc = ((r * 64 + lane) & k) == 0;
                        const unsigned long long x = e[r], y = e[pr];
                        const bool sw = asc ? (x > y) : (x < y);
                        e[r] = sw ? y : x;
                        e[pr] = sw ? x : y;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const unsigned long long mine = e[r];
                    const unsigned long long other = shfl_xor_u64(mine, j);
                    const bool asc = ((r * 64 + lane) & k) == 0;
                    const bool lower = (lane & j) == 0;
                    const bool take_min = lower == asc;
                    const unsigned long long mn = mine < other ? mine : other, mx = mine < other ? other : mine;
                    e[r] = take_min ? mn : mx;
                }
            }
        }
    }
}
__device__ __forceinline__ void wave_bitonic_sort256(unsigned long long (&e)[4]) { wave_bitonic_sort<4>(e); }

// k-th smallest (k = 1..number of valid entries) of the wave's 64 * NR register entries (NR per lane), by an MSB-first
// radix select on lane masks: per bit one compare + ballot per register and scalar popcounts, no cross-lane data
// movement, and it stops as soon as a single candidate is left (~20 of 64 bits for distinct fp32 keys).
// Slots that hold no entry must be ~0ull.
template <int NR>
__device__ __forceinline__ unsigned long long wave_select(const unsigned long long (&e)[NR], int k)
{
    unsigned long long alive[NR];  // wave-uniform lane masks
#pragma unroll
    for (int r = 0; r < NR; ++r) alive[r] = ~0ull;
    int kk = k, n_alive = 64 * NR;
    unsigned long long prefix = 0;
#pragma unroll
    for (int half = 1; half >= 0; --half) {
        uint32_t w[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) w[r] = half ? (uint32_t)(e[r] >> 32) : (uint32_t)e[r];
        for (int b = 31; b >= 0 && n_alive > 1; --b) {  // wave-uniform
            const uint32_t m = 1u << b;
            unsigned long long z[NR];
            int c0 = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                z[r] = __ballot((w[r] & m) == 0) & alive[r];
                c0 += __popcll(z[r]);
            }
            if (kk <= c0) {
#pragma unroll
                for (int r = 0; r < NR; ++r) alive[r] = z[r];
                n_alive = c0;
            } else {
#pragma unroll
                for (int r = 0; r < NR; ++r) alive[r] &= ~z[r];
                kk -= c0;
                n_alive -= c0;
                prefix |= (unsigned long long)m << (half * 32);
            }
        }
    }
    if (n_alive == 1) {  // the survivor's value (the loop may have stopped before its low bits were walked)
        unsigned long long v = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (alive[r]) {  // wave-uniform
                const int src = __ffsll((long long)alive[r]) - 1;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e[r], src);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e[r] >> 32), src);
                v = ((unsigned long long)hi << 32) | lo;
            }
        }
        return v;
    }
    return prefix;  // duplicates of the k-th value: every bit was walked
}

// k-th smallest VALUE of the NBITS-bit field that starts at bit 32 of the wave's 64 * NR register entries (the key of an entry
// whose keys are small integers): the radix select above restricted to that field -- NBITS rounds, ties need no resolving
// because only the value is wanted.  Slots that hold no entry must be ~0ull (their field reads as all ones: never below a real key).
template <int NR, int NBITS>
__device__ __forceinline__ uint32_t wave_select_field(const unsigned long long (&e)[NR], int k)
{
    unsigned long long alive[NR];  // wave-uniform lane masks; entries wider than the field (empty slots) never take part
    int kk = k;
    uint32_t prefix = 0;
    uint32_t w[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        w[r] = (uint32_t)(e[r] >> 32);
        alive[r] = __ballot((w[r] >> NBITS) == 0);
    }
    for (int b = NBITS - 1; b >= 0; --b) {  // wave-uniform
        const uint32_t m = 1u << b;
        unsigned long long z[NR];
        int c0 = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            z[r] = __ballot((w[r] & m) == 0) & alive[r];
            c0 += __popcll(z[r]);
        }
        if (kk <= c0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) alive[r] = z[r];
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) alive[r] &= ~z[r];
            kk -= c0;
            prefix |= m;
        }
    }
    return prefix;
}

#ifdef CVTMI_SCAN_TIMING
static __device__ unsigned long long g_topk_dbg[4];  // compaction rounds, fix cycles, sort cycles, new entries (thread 0's view)
#define TK_T(i, t0) do { if (threadIdx.x == 0) atomicAdd(&g_topk_dbg[i], (unsigned long long)(clock64() - (t0))); } while (0)
#else
#define TK_T(i, t0) do { } while (0)
#endif

// one query, one wave, no barrier: returns the number of entries kept.
// SORTED: the kept entries are left in ascending order (needed once, for the result); otherwise they are the k
// smallest in arbitrary order, found by selection instead of a sort (the intermediate compactions only need the
// set and the k-th value).
// Register slot p = 64 r + lane (NE = 64 NR slots) holds new entry ex + p for p < n - ex and exact entry
// p - (NE - ex) for p >= NE - ex: the entries to be fixed sit in the first registers, so fixb can skip its second batch when
// there are at most 128 of them.
// n_in >= 0: the number of entries to take (callers that keep s.cnt[q] locked while they compact), otherwise s.cnt[q]
template <int QT, int CAP, bool SORTED, class FixB, class ThrX>
__device__ __forceinline__ int topk_compact_wave_q_inl(TopKShared<QT, CAP> &s, int q, int k, const FixB &fixb, const ThrX &thrx,
                                                       int n_in = -1)
{
    static_assert(CAP <= 256, "register sort holds 256 entries per wave");
    constexpr int NR = CAP <= 64 ? 1 : (CAP <= 128 ? 2 : 4);  // registers per lane
    constexpr int NE = 64 * NR;
    const int lane = threadIdx.x & 63;
    unsigned long long *b = s.buf[q];
    int n = n_in >= 0 ? n_in : s.cnt[q];
    n = n < CAP ? n : CAP;
    const int ex = s.exact_n[q];
    unsigned long long e[NR];
    bool need[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int p = r * 64 + lane;
        need[r] = p < n - ex;
        const bool old = p >= NE - ex;
        const int idx = need[r] ? ex + p : p - (NE - ex);
        e[r] = (need[r] || old) ? b[idx] : ~0ull;
    }
#ifdef CVTMI_SCAN_TIMING
    const long long tk0 = clock64();
    if (threadIdx.x == 0) { atomicAdd(&g_topk_dbg[0], 1ull); atomicAdd(&g_topk_dbg[3], (unsigned long long)(n - ex)); }
#endif
    fixb(q, e, need);
#ifdef CVTMI_SCAN_TIMING
    TK_T(1, tk0);
    const long long tk1 = clock64();
#endif
    const int keep = n < k ? n : k;
    uint32_t th_k;
    if constexpr (SORTED) {
        wave_bitonic_sort<NR>(e);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int idx = r * 64 + lane;
            if (idx < keep) b[idx] = e[r];
        }
        // k-th entry: element index k-1 lives in register (k-1)>>6 of lane (k-1)&63
        unsigned long long kth = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (((k - 1) >> 6) == r) kth = e[r];
        const uint32_t th_lane = (uint32_t)(kth >> 32);
        th_k = (uint32_t)__shfl((int)th_lane, (k - 1) & 63);
    } else {
        // n < k: everything stays (all entries are below ~0ull - 1)
        const unsigned long long kth = n >= k ? wave_select<NR>(e, k) : 0xfffffffffffffffeull;
        th_k = (uint32_t)(kth >> 32);
        int base = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool in = e[r] <= kth;
            const unsigned long long m = __ballot(in);
            const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (in && pos < keep) b[pos] = e[r];
            base += __popcll(m);
        }
    }
#ifdef CVTMI_SCAN_TIMING
    TK_T(2, tk1);
#endif
    if (lane == 0) {
        s.exact_n[q] = keep;
        const uint32_t th = (n >= k) ? th_k : KEY_MAX;
        s.thr[q] = th;
        s.thr_x[q] = thrx(q, th);
    }
    return keep;
}

// out-of-line form: a call keeps the caller's hot loop small, but values live across it must sit in the
// callee-saved half of the register file (kernels near the VGPR limit use the _inl form instead)
template <int QT, int CAP, bool SORTED, class FixB, class ThrX>
__device__ __attribute__((noinline)) int topk_compact_wave_q(TopKShared<QT, CAP> &s, int q, int k, const FixB &fixb,
                                                            const ThrX &thrx, int n_in = -1)
{
    return topk_compact_wave_q_inl<QT, CAP, SORTED>(s, q, k, fixb, thrx, n_in);
}

template <int QT, int CAP, int NT, bool SORTED, class FixB, class ThrX>
__device__ __forceinline__ void topk_compact_wave(TopKShared<QT, CAP> &s, int k, const FixB &fixb, const ThrX &thrx)
{
    constexpr int NW = NT / 64;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = wv; q < QT; q += NW) {  // wave-uniform
        const int keep = topk_compact_wave_q<QT, CAP, SORTED>(s, q, k, fixb, thrx);
        if (lane == 0) s.cnt[q] = keep;
    }
    __syncthreads();
}

// End-of-tile protocol.  `want` = some push of this thread crossed TRIG; `pending` = bit mask of this
// thread's candidates that found the buffer full; retry(pending) re-offers them (comparing against
// the refreshed s.thr_x with '<=') and returns the mask of those still not stored.
// One barrier on the fast path.  `tile` is the workgroup-uniform tile counter.
template <int QT, int CAP, int NT, class Fix, class ThrX, class Retry>
__device__ __forceinline__ void topk_tile_end(TopKShared<QT, CAP> &s, int k, int tile, bool want, uint32_t pending,
                                              const Fix &fix, const ThrX &thrx, Retry &&retry)
{
    const int f = tile % 3;
    if (want) s.flag[f] = 1;
    __syncthreads();
    if (s.flag[f]) {  // workgroup-uniform
        for (;;) {
            topk_compact<QT, CAP, NT, Fix, ThrX>(s, k, fix, thrx);
            if (pending) s.flag[3] = 1;
            __syncthreads();
            const int again = s.flag[3];
            __syncthreads();
            if (!again) break;
            if (threadIdx.x == 0) s.flag[3] = 0;
            pending = retry(pending);
            __syncthreads();
        }
    }
    // flag (tile-1)%3 was last read before this tile's barrier: safe to clear for tile+2
    if (threadIdx.x == 0) s.flag[(tile + 2) % 3] = 0;
}

// Convenience form for kernels that hold R x QT exact keys (KEY_MAX = not a candidate) and R payloads
// per thread and obey the ordering rule above.
template <int QT, int R, int CAP, int TRIG, int NT = kBlock>
__device__ __forceinline__ void topk_tile(TopKShared<QT, CAP> &s, int k, int tile, const uint32_t (&key)[R][QT],
                                          const uint32_t (&pay)[R])
{
    uint32_t thr[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) thr[q] = s.thr_x[q];
    bool want = false;
    uint32_t pending = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            if (key[r][q] < thr[q]) {
                if (!topk_push<QT, CAP, TRIG>(s, q, key[r][q], pay[r], want)) pending |= 1u << (r * QT + q);
            }
        }
    }
    topk_tile_end<QT, CAP, NT>(s, k, tile, want, pending, NoFix(), IdThr(), [&](uint32_t pend) {
        uint32_t still = 0;
        bool dummy = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const uint32_t bit = 1u << (r * QT + q);
                if ((pend & bit) && key[r][q] <= s.thr_x[q] && key[r][q] != KEY_MAX) {
                    if (!topk_push<QT, CAP, TRIG>(s, q, key[r][q], pay[r], dummy)) still |= bit;
                }
            }
        }
        return still;
    });
}

}  // namespace cvtmi
