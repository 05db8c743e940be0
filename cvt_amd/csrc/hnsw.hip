// hnsw.hip -- batched HNSW search over a graph built and saved by the reference:
// HierarchicalNSW::searchKnn (hnsw_sifts_retrieval/hnswlib/hnswalg.h:688-729) = greedy descent through the
// upper levels (:692-712), searchBaseLayerST (:217-280) with ef = max(ef_, k), then the k best.
//
// One wave per query (the traversal is a chain of dependent gathers: latency-bound, not roofline-bound;
// throughput comes from thousands of queries in flight).  What a wave does per expanded node:
//   * its <= 64 level-0 neighbours are taken one per lane: visited test-and-set on a per-query bitmap in
//     HBM, then every unvisited lane computes its neighbour's full distance in the reference's summation
//     order (dist_f32.h), reading its 4D-byte vector with 16-byte loads;
//   * the accept / push / pop decisions are replayed in list order, exactly as the scalar loop makes them.
// Both queues of the reference are std::priority_queue with CompareByFirst (:78-83), so ties between equal
// distances are decided by the binary-heap mechanics: push_heap / pop_heap of libstdc++ (__push_heap,
// __adjust_heap) are restated literally on arrays in LDS (the candidate queue spills to HBM past LCAP).
// With that, labels and distances are bit-identical to the reference's on the same graph, duplicates included.
#include <atomic>

#include "dist_f32.h"
#include "kernels.h"

namespace cvtmi {

constexpr int HN_EF_MAX = 1024;  // top queue: ef + 1 entries in LDS
constexpr int HN_LCAP = 256;    // candidate queue entries kept in LDS; the rest lives in HBM

struct HnEnt { float d; uint32_t id; };

// max-heap on d over an array addressed through A (get / set), n entries
template <class A>
__device__ __forceinline__ void hn_push_heap(A &a, int hole, HnEnt v)
{
    int parent = (hole - 1) / 2;
    while (hole > 0) {
        const HnEnt p = a.get(parent);
        if (!(p.d < v.d)) break;
        a.set(hole, p);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a.set(hole, v);
}
// The same sift-up with the whole ancestor chain taken at once: lane i reads ancestor i of the hole (the chain is known before
// anything is compared: positions (hole + 1) >> (i + 1), 1-based), one ballot finds the first ancestor that stays, the ancestors
// below it move down one step each and v lands above them -- two memory round trips whatever the depth, instead of one per level.
// Same comparisons, same final layout as __push_heap.
template <class A>
__device__ __forceinline__ void hn_push_heap_wave(A &a, int hole, HnEnt v, int lane)
{
    const int h1 = hole + 1;
    const int anc1 = lane < 31 ? (h1 >> (lane + 1)) : 0;      // 1-based ancestor i, 0 = past the root
    const bool valid = anc1 >= 1;
    HnEnt p; p.d = 0.0f; p.id = 0u;
    if (valid) p = a.get_l(anc1 - 1);
    const unsigned long long up = __ballot(valid && p.d < v.d);   // ancestor i is passed
    const int s_ = __ffsll((long long)~up) - 1;                    // first ancestor that stays (or the first lane past the root)
    const int below = lane == 0 ? hole : (h1 >> lane) - 1;         // the position one step below ancestor `lane` on the chain
    if (lane < s_) a.set_l(below, p);
    else if (lane == s_) a.set_l(below, v);
}
template <class A>
__device__ __forceinline__ void hn_push(A &a, int &n, float d, uint32_t id, int lane)
{
    HnEnt v; v.d = d; v.id = id;
    hn_push_heap_wave(a, n, v, lane);
    ++n;
}
// __adjust_heap walks from the root to a leaf, one comparison of two children per level: a chain of dependent reads, eleven deep for a
// queue of 2000 entries, the lower three or four of them in HBM.  Here the wave fetches the next FIVE levels below the hole at once --
// level j of that subtree is the 2^j consecutive entries from (hole + 1) 2^j - 1, lanes 2^j - 2 .. 2^(j+1) - 3 take it -- and the five
// comparisons run on register values (v_readlane with a uniform lane number): one memory round trip per five levels, same
// comparisons, same moves, same final layout.
__device__ __forceinline__ HnEnt hn_lane(const HnEnt &e, int src)
{
    HnEnt r;
    r.d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.d), src));
    r.id = (uint32_t)__builtin_amdgcn_readlane((int)e.id, src);
    return r;
}
template <class A>
__device__ __forceinline__ void hn_pop(A &a, int &n, int lane)
{
    if (n > 1) {
        const int len = n - 1;
        int hole = 0, second = 0;
        const int limit = (len - 1) / 2;
        const int j = 31 - __clz(lane + 2);          // level of this lane's entry in the subtree (1 .. 5; lanes 62, 63 idle)
        const int off = lane + 2 - (1 << j);
        HnEnt value; value.d = 0.0f; value.id = 0u;
        bool have_value = false;
        while (second < limit) {
            const int pos = ((second + 1) << j) - 1 + off;
            HnEnt e; e.d = 0.0f; e.id = 0u;
            if (j <= 5 && pos < len) e = a.get_l(pos);
            if (!have_value) {                         // the entry that leaves the end of the array rides along with the first fetch (lane 63)
                if (lane == 63) e = a.get_l(len);
                value = hn_lane(e, 63);
                have_value = true;
            }
            int rel = 0;                               // offset of the hole within its level of the subtree
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (second >= limit) break;
                second = 2 * (second + 1);
                const int lr = __builtin_amdgcn_readfirstlane((2 << t) - 2 + 2 * rel + 1);   // lane of the right child
                const HnEnt r = hn_lane(e, lr), l = hn_lane(e, lr - 1);
                HnEnt pick = r;
                rel = 2 * rel + 1;
                if (r.d < l.d) { --second; --rel; pick = l; }
                a.set(hole, pick);
                hole = second;
            }
        }
        if (!have_value) value = a.get(len);           // (len <= 2: no level to walk)
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            a.set(hole, a.get(second - 1));
            hole = second - 1;
        }
        hn_push_heap_wave(a, hole, value, lane);
    }
    --n;
}

// Every lane runs the heap code with wave-uniform values; loads broadcast, lane 0 stores.
struct LdsArr {
    HnEnt *p; bool w;
    __device__ __forceinline__ HnEnt get(int i) const { return p[i]; }
    __device__ __forceinline__ void set(int i, HnEnt v) const { if (w) p[i] = v; }
    __device__ __forceinline__ HnEnt get_l(int i) const { return p[i]; }            // per-lane index
    __device__ __forceinline__ void set_l(int i, HnEnt v) const { p[i] = v; }
};
// first `cap` entries (the upper heap levels, touched by every operation) in LDS, the rest in HBM
struct SplitArr {
    HnEnt *l; HnEnt *g; bool w; int cap;
    __device__ __forceinline__ HnEnt get(int i) const { return i < cap ? l[i] : g[i - cap]; }
    __device__ __forceinline__ void set(int i, HnEnt v) const { if (w) { if (i < cap) l[i] = v; else g[i - cap] = v; } }
    __device__ __forceinline__ HnEnt get_l(int i) const { return i < cap ? l[i] : g[i - cap]; }   // per-lane index
    __device__ __forceinline__ void set_l(int i, HnEnt v) const { if (i < cap) l[i] = v; else g[i - cap] = v; }
};

struct HnswArgs {
    const float *vec;         // [n][D]
    const uint32_t *links0;   // [n][maxM0 + 1]: count, then neighbours
    const int64_t *labels;    // [n]
    const int64_t *upper_off; // [n]: first word of the element's upper-level block in `upper`, -1 if none
    const uint32_t *upper;    // per element: levels x (maxM + 1) words
    int64_t n;
    int D, maxM, maxM0, maxlevel;
    uint32_t enterpoint;
    const float *q;           // fp32 mode: queries [nq][D]
    const float *lut;         // ADC mode: per-query tables [nq][M][K] (lut_kernel), codes [n][M]
    const uint8_t *codes;
    int M, K;
    int nq, k, ef;
    float *out_d;
    int64_t *out_label;
    uint32_t *visited;        // [slots][words]
    HnEnt *cand_g;            // [slots][gcap]
    int64_t words, gcap;
    int *err;
    int raw_ids;              // 1 = out_label receives internal ids instead of labels (the re-rank pass needs the node)
    int top_lds;              // entries of the top queue kept in LDS (the candidate queue: HN_LCAP)
    int dynamic;              // 1 = queries handed out by the counter err[1] (one ticket per wave), 0 = static stride
};

// Distance evaluators: per-query state in LDS (`prepare`), one neighbour per lane (`operator()`).
template <bool IP, int LANES>
struct DistF32 {
    const HnswArgs &a;
    float *qs;
    __device__ __forceinline__ int smem_floats() const { return (a.D + 3) & ~3; }
    __device__ __forceinline__ void prepare(int qi, int lane) const
    {
        for (int i = lane; i < a.D; i += 64) qs[i] = a.q[(int64_t)qi * a.D + i];
    }
    __device__ __forceinline__ float operator()(uint32_t id) const
    {
        float o[1];
        dist_f32_row<IP, LANES, 1>(a.vec + (int64_t)id * a.D, qs, a.D, o);
        return o[0];
    }
    // level 0: nothing is fetched before the visited test (4 D bytes per neighbour is what bounds this traversal)
    struct Early {};
    __device__ __forceinline__ Early early(uint32_t) const { return Early(); }
    __device__ __forceinline__ float finish(uint32_t id, const Early &) const { return (*this)(id); }
};
// "HNSW over OPQ-compressed vectors": the node's M code bytes index the query's distance tables, summed in m
// order from +0.0f exactly as the ADC scan does (IVFOPQ.cpp:302-306) -- 16 bytes gathered per neighbour
// instead of 4 D.
// Where the fp32 tables live (round 5).  LDS_TABLES = true: copied into LDS per query (16 KB at M = 16: 8 query slots per CU, and 64 loads
// per lane before a traversal can start).  LDS_TABLES = false (default): read where lut_kernel left them -- a lane's 16 entries are 16
// independent 4-byte gathers from the query's 16 KB table (hot in L2 / Infinity Cache: a traversal touches it ~10^4 times), ONE more memory
// round trip per expanded node, the same one the fp32 traversal spends on its 4 D-byte vector gather -- and the slot shrinks to its two
// queues (4 KB): 32 traversals per CU instead of 8.  The traversal is latency-bound, so queries in flight are what it lives on.
template <bool LDS_TABLES>
struct DistADC {
    const HnswArgs &a;
    float *lds;          // [M][K] (LDS_TABLES)
    mutable const float *lut;
    __device__ __forceinline__ int smem_floats() const { return LDS_TABLES ? ((a.M * a.K + 3) & ~3) : 0; }
    __device__ __forceinline__ void prepare(int qi, int lane) const
    {
        const float *src = a.lut + (int64_t)qi * a.M * a.K;
        if (LDS_TABLES) {
            for (int i = lane; i < a.M * a.K; i += 64) lds[i] = src[i];
            lut = lds;
        } else {
            lut = src;
        }
    }
    __device__ __forceinline__ float sum16(const uint4 &v) const
    {
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
        float e[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) e[m] = lut[m * a.K + ((w[m >> 2] >> (8 * (m & 3))) & 0xffu)];   // 16 independent look-ups first
        float s = 0.0f;
#pragma unroll
        for (int m = 0; m < 16; ++m) s = __fadd_rn(s, e[m]);
        return s;
    }
    __device__ __forceinline__ float operator()(uint32_t id) const
    {
        const uint8_t *c = a.codes + (int64_t)id * a.M;
        if (a.M == 16) return sum16(*reinterpret_cast<const uint4 *>(c));  // one 16-byte gather per neighbour
        float s = 0.0f;
        for (int m = 0; m < a.M; ++m) s = __fadd_rn(s, lut[m * a.K + c[m]]);
        return s;
    }
    // level 0: a neighbour's 16 code bytes are requested together with its visited bit -- 1 KB per expanded node, and the traversal
    // is a chain of dependent round trips: this one now overlaps the test-and-set instead of following it
    struct Early { uint4 v; };
    __device__ __forceinline__ Early early(uint32_t id) const
    {
        Early e;
        e.v = make_uint4(0u, 0u, 0u, 0u);
        if (a.M == 16) e.v = *reinterpret_cast<const uint4 *>(a.codes + (int64_t)id * 16);
        return e;
    }
    __device__ __forceinline__ float finish(uint32_t id, const Early &e) const { return a.M == 16 ? sum16(e.v) : (*this)(id); }
};

template <class DIST>
__global__ __launch_bounds__(64, 8) void hnsw_search_kernel(const HnswArgs a)  // <= 64 VGPRs: 8 waves per SIMD, the traversal lives on queries in flight
{
    extern __shared__ __attribute__((aligned(16))) float hn_smem[];
    const DIST dist{ a, hn_smem };                                     // query state first (padded to 16 bytes; lut: set by prepare)
    HnEnt *top_l = reinterpret_cast<HnEnt *>(hn_smem + dist.smem_floats());
    const int ef_cap = (a.ef > a.k ? a.ef : a.k) + 1;   // the top queue never holds more than ef + 1 entries
    const int top_cap = ef_cap < a.top_lds ? ef_cap : a.top_lds;
    HnEnt *cand_l = top_l + top_cap;
    const int lane = threadIdx.x;
    const bool w = lane == 0;
    uint32_t *vis = a.visited + (int64_t)blockIdx.x * a.words;
    // per-slot HBM scratch: [top queue past HN_LCAP: ef + 1 entries][candidate queue past HN_LCAP: gcap entries]
    HnEnt *slot_g = a.cand_g + (int64_t)blockIdx.x * (a.gcap + ef_cap);
    const SplitArr top{ top_l, slot_g, w, top_cap };
    const SplitArr cand{ cand_l, slot_g + ef_cap, w, HN_LCAP };
    const int ef = a.ef > a.k ? a.ef : a.k;
    const int64_t cand_cap = HN_LCAP + a.gcap;

    // Queries are drawn from a counter (err[1], zeroed with the flag) instead of a static stride: traversals differ in length, and the last
    // of the 1.2 - 2.6 rounds used to wait for its slowest waves.  EVERY lane executes the atomic (a header that only lane 0 executes
    // lets the compiler send lane 0 and the other lanes through the loop on different paths, and the wave-level operations of the
    // body then run without lane 0 -- observed: the kernel never ended), but only lane 0 adds: its old value is a ticket that is
    // unique per wave whether or not the compiler folds the wave's atomics into one.
    for (int it = 0;; ++it) {
        int qi;
        if (a.dynamic) {
            const unsigned t = atomicAdd(reinterpret_cast<unsigned *>(a.err) + 1, lane == 0 ? 1u : 0u);
            qi = __builtin_amdgcn_readfirstlane((int)t);
        } else {
            qi = (int)blockIdx.x + it * (int)gridDim.x;
        }
        if (qi >= a.nq) break;
        dist.prepare(qi, lane);
        for (int64_t i = lane; i < a.words; i += 64) vis[i] = 0u;
        for (int i = lane; i < a.k; i += 64) { a.out_d[(int64_t)qi * a.k + i] = 0.0f; a.out_label[(int64_t)qi * a.k + i] = -1; }
        __builtin_amdgcn_s_waitcnt(0);
        __threadfence_block();
        if (a.n == 0) continue;

        // ---- upper levels: first minimum among the neighbours that beats the current distance (:692-712) ----
        uint32_t cur = a.enterpoint;
        float curdist = dist(cur);
        for (int level = a.maxlevel; level > 0; --level) {
            bool changed = true;
            while (changed) {
                changed = false;
                const uint32_t *ll = a.upper + a.upper_off[cur] + (int64_t)(level - 1) * (a.maxM + 1);
                const int size = (int)ll[0];
                for (int base = 0; base < size; base += 64) {
                    const int j = base + lane;
                    const bool act = j < size;
                    const uint32_t nb = act ? ll[1 + j] : cur;
                    float o[1];
                    o[0] = dist(nb);
                    // the scalar loop keeps the FIRST neighbour that reaches the running minimum
                    unsigned long long better = __ballot(act && o[0] < curdist);
                    while (better) {
                        const int b = __ffsll((long long)better) - 1;
                        const float db = __shfl(o[0], b);
                        const uint32_t ib = (uint32_t)__shfl((int)nb, b);
                        if (db < curdist) { curdist = db; cur = ib; changed = true; }
                        better &= better - 1;
                        better &= __ballot(act && o[0] < curdist);
                    }
                }
            }
        }

        // ---- level 0: searchBaseLayerST (:217-280) ----
        int top_n = 0, cand_n = 0;
        {
            float o[1];
            o[0] = dist(cur);
            hn_push(top, top_n, o[0], cur, lane);
            hn_push(cand, cand_n, -o[0], cur, lane);
            if (w) vis[cur >> 5] |= 1u << (cur & 31);
            __builtin_amdgcn_s_waitcnt(0);
        }
        float lower = top.get(0).d;
        bool overflow = false;
        while (cand_n > 0) {
            const HnEnt c = cand.get(0);
            if (-c.d > lower) break;
            // the expanded node's links are requested BEFORE the pop walks the queue (the walk's deeper levels live in HBM): the two
            // latencies overlap instead of adding up
            // (count and first 64 neighbours in ONE round trip: a node's row always holds maxM0 + 1 words, stale ones past the count
            //  are masked below once the count is known)
            const uint32_t *ll = a.links0 + (int64_t)c.id * (a.maxM0 + 1);
            const uint32_t nb0 = lane < a.maxM0 ? ll[1 + lane] : 0u;
            const int size = (int)ll[0];
            hn_pop(cand, cand_n, lane);
            for (int base = 0; base < size; base += 64) {
                const int j = base + lane;
                bool act = j < size;
                const uint32_t nb = base == 0 ? nb0 : (act ? ll[1 + j] : 0u);
                typename DIST::Early pre = dist.early(act ? nb : c.id);
                if (act) {
                    const uint32_t bit = 1u << (nb & 31);
                    act = (atomicOr(&vis[nb >> 5], bit) & bit) == 0;
                }
                float o[1] = { 0.0f };
                if (act) o[0] = dist.finish(nb, pre);
                // list order.  Once the top queue is full its maximum only falls, so a neighbour that fails `lower > d` now fails it
                // for the rest of this list: the rejected ones are dropped by ballot, and the walk visits accepted neighbours only
                unsigned long long m = __ballot(act && (top_n < ef || lower > o[0]));
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const float d = __shfl(o[0], b);
                    const uint32_t id = (uint32_t)__shfl((int)nb, b);
                    if (lower > d || top_n < ef) {   // lower == top.get(0).d whenever the queue is non-empty
                        if (cand_n >= cand_cap) { overflow = true; break; }
                        hn_push(cand, cand_n, -d, id, lane);
                        hn_push(top, top_n, d, id, lane);
                        if (top_n > ef) hn_pop(top, top_n, lane);
                        lower = top.get(0).d;
                        if (top_n >= ef) m &= __ballot(act && lower > o[0]);
                    }
                }
                if (overflow) break;
            }
            if (overflow) break;
        }
        if (overflow) {
            if (w) atomicExch(a.err, 1);
            continue;
        }
        while (top_n > a.k) hn_pop(top, top_n, lane);
        // pops come out in non-increasing distance: write them back to front, then order ties by label,
        // which is the (dist, label) order of the reference's result queue (:719-726)
        const int m = top_n;
        for (int i = m - 1; i >= 0; --i) {
            const HnEnt e = top.get(0);
            if (w) {
                a.out_d[(int64_t)qi * a.k + i] = e.d;
                a.out_label[(int64_t)qi * a.k + i] = a.raw_ids ? (int64_t)e.id : a.labels[e.id];
            }
            hn_pop(top, top_n, lane);
        }
        if (w) {
            float *od = a.out_d + (int64_t)qi * a.k;
            int64_t *ol = a.out_label + (int64_t)qi * a.k;
            for (int i = 1; i < m; ++i) {
                const float d = od[i];
                const int64_t l = ol[i];
                int j = i - 1;
                while (j >= 0 && od[j] == d && ol[j] > l) { ol[j + 1] = ol[j]; --j; }
                ol[j + 1] = l;
            }
        }
    }
}

static void hnsw_fill_args(HnswArgs &a, const HnswDevGraph &g, int64_t nq, int k, int ef, float *out_d, int64_t *out_label,
                           uint32_t *visited, void *cand_scratch, int64_t words, int64_t gcap, int *err)
{
    a.vec = g.vec; a.links0 = g.links0; a.labels = g.labels; a.upper_off = g.upper_off; a.upper = g.upper;
    a.n = g.n; a.D = g.D; a.maxM = g.maxM; a.maxM0 = g.maxM0; a.maxlevel = g.maxlevel; a.enterpoint = g.enterpoint;
    a.q = nullptr; a.lut = nullptr; a.codes = nullptr; a.M = 0; a.K = 0;
    a.nq = (int)nq; a.k = k; a.ef = ef; a.out_d = out_d; a.out_label = out_label;
    a.visited = visited; a.cand_g = reinterpret_cast<HnEnt *>(cand_scratch); a.words = words; a.gcap = gcap; a.err = err;
    a.raw_ids = 0;
    a.dynamic = 1;
    a.top_lds = hnsw_top_lds(ef > k ? ef : k);
}

int launch_hnsw_search(const HnswDevGraph &g, int metric, const float *q, int64_t nq, int k, int ef, float *out_d,
                       int64_t *out_label, uint32_t *visited, void *cand_scratch, int slots, int64_t words, int64_t gcap,
                       int *err, hipStream_t st)
{
    if (nq <= 0) return CVTMI_OK;
    HnswArgs a;
    hnsw_fill_args(a, g, nq, k, ef, out_d, out_label, visited, cand_scratch, words, gcap, err);
    a.q = q;
    const size_t lds = (size_t)hnsw_lds_bytes(g.D, ef > k ? ef : k);
    const bool ip = metric == CVTMI_METRIC_IP;
    const int lanes = (g.D % 4 != 0) ? 1 : (ip ? 4 : (g.D % 16 == 0 ? 8 : 4));
#define CVTMI_HN(IPV, L)                                                                                                                   \
    do {                                                                                                                                   \
        CVTMI_HIP(hipFuncSetAttribute((const void *)hnsw_search_kernel<DistF32<IPV, L> >, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((hnsw_search_kernel<DistF32<IPV, L> >), dim3((unsigned)slots), dim3(64), lds, st, a);                            \
    } while (0)
    if (ip) { if (lanes == 4) CVTMI_HN(true, 4); else CVTMI_HN(true, 1); }
    else { if (lanes == 8) CVTMI_HN(false, 8); else if (lanes == 4) CVTMI_HN(false, 4); else CVTMI_HN(false, 1); }
#undef CVTMI_HN
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// same traversal, distances = ADC over the nodes' PQ codes (codes [n][M] in internal-id order, lut [nq][M][K])
int launch_hnsw_search_adc(const HnswDevGraph &g, const float *lut, const uint8_t *codes, int M, int K, int64_t nq, int k, int ef,
                           float *out_d, int64_t *out_label, uint32_t *visited, void *cand_scratch, int slots, int64_t words,
                           int64_t gcap, int *err, hipStream_t st, int raw_ids, int state_floats)
{
    if (nq <= 0) return CVTMI_OK;
    HnswArgs a;
    hnsw_fill_args(a, g, nq, k, ef, out_d, out_label, visited, cand_scratch, words, gcap, err);
    a.lut = lut; a.codes = codes; a.M = M; a.K = K; a.raw_ids = raw_ids;
    // state_floats: what the caller sized the slots for (hnsw_adc_state_floats, read ONCE per search: M K = tables in LDS, 0 = in the scratch)
    const bool lds_tables = state_floats != 0;
    const size_t lds = (size_t)hnsw_lds_bytes(state_floats, ef > k ? ef : k);
    if (lds_tables) {
        CVTMI_HIP(hipFuncSetAttribute((const void *)hnsw_search_kernel<DistADC<true> >, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((hnsw_search_kernel<DistADC<true> >), dim3((unsigned)slots), dim3(64), lds, st, a);
    } else {
        CVTMI_HIP(hipFuncSetAttribute((const void *)hnsw_search_kernel<DistADC<false> >, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((hnsw_search_kernel<DistADC<false> >), dim3((unsigned)slots), dim3(64), lds, st, a);
    }
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// Exact re-rank of an ADC result list: ids [nq][R] internal ids (-1 = padding) -> out_d [nq][R] = the fp32 distance of the
// raw query to the node's own vector, in the summation order of the reference's distance functions (+inf for padding).
// One lane per candidate; a wave's 64 gathers of 4 D bytes are what the fp32 traversal pays per expanded node.
template <bool IP, int LANES>
__global__ __launch_bounds__(kBlock) void hnsw_rerank_kernel(const float *__restrict__ vec, int D, const float *__restrict__ q,
                                                             const int64_t *__restrict__ ids, int64_t total, int R,
                                                             float *__restrict__ out_d)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const int64_t id = ids[i];
    float o[1] = { __uint_as_float(0x7f800000u) };
    if (id >= 0) dist_f32_row<IP, LANES, 1>(vec + id * D, q + (i / R) * D, D, o);
    out_d[i] = o[0];
}

int launch_hnsw_rerank(const HnswDevGraph &g, int metric, const float *q, int64_t nq, int R, const int64_t *ids, float *out_d,
                       hipStream_t st)
{
    const int64_t total = nq * R;
    if (total <= 0) return CVTMI_OK;
    const unsigned blocks = (unsigned)((total + kBlock - 1) / kBlock);
    const bool ip = metric == CVTMI_METRIC_IP;
    const int lanes = (g.D % 4 != 0) ? 1 : (ip ? 4 : (g.D % 16 == 0 ? 8 : 4));  // as launch_hnsw_search picks them
#define CVTMI_RR(IPV, L) hipLaunchKernelGGL((hnsw_rerank_kernel<IPV, L>), dim3(blocks), dim3(kBlock), 0, st, g.vec, g.D, q, ids, total, R, out_d)
    if (ip) { if (lanes == 4) CVTMI_RR(true, 4); else CVTMI_RR(true, 1); }
    else { if (lanes == 8) CVTMI_RR(false, 8); else if (lanes == 4) CVTMI_RR(false, 4); else CVTMI_RR(false, 1); }
#undef CVTMI_RR
    CVTMI_HIP(hipGetLastError());
    return CVTMI_OK;
}

// LDS bytes of one query slot: query state (floats) + top queue (ef + 1) + the LDS part of the candidate queue
// Entries of the top queue (ef + 1) kept in LDS; the rest of it lives in HBM like the candidate queue's.  Keeping all of it in LDS (8 KB at
// ef = 1000) was measured: fp32 ef = 1000 174 K -> 151 K queries/s, ADC 101 K -> 97 K at 1 M nodes -- the query slots lost per CU cost more than
// the HBM levels of the heap walks, which the wave-wide pop already crosses in one or two round trips.  cvtmi_set_tuning("hnsw_top_lds", n),
// 0 = everything.
static std::atomic<int> g_hnsw_top_lds{ 256 };
void set_hnsw_top_lds(int v) { g_hnsw_top_lds = v < 0 ? 0 : v; }
int hnsw_top_lds(int ef)
{
    const int cap = g_hnsw_top_lds.load();
    const int want = ef + 1;
    return cap > 0 && cap < want ? (cap < 16 ? 16 : cap) : want;
}
int hnsw_lds_bytes(int state_floats, int ef)
{
    return (int)(((state_floats + 3) & ~3) * sizeof(float) + (size_t)(hnsw_top_lds(ef) + HN_LCAP) * sizeof(HnEnt));
}
// cvtmi_set_tuning("hnsw_adc_tables"): 0 = a query's fp32 tables are read from the scratch lut_kernel wrote (default), 1 = copied into LDS
static std::atomic<int> g_hnsw_adc_tables_lds{ 0 };
void set_hnsw_adc_tables(int v) { g_hnsw_adc_tables_lds = v != 0; }
int hnsw_adc_state_floats(int MK) { return g_hnsw_adc_tables_lds.load() ? MK : 0; }   // per-slot query state of the ADC traversal, in floats
int hnsw_ef_max() { return HN_EF_MAX; }
int hnsw_lcap() { return HN_LCAP; }

}  // namespace cvtmi
