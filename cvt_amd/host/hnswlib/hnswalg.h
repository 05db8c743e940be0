// HierarchicalNSW<float> -- mirror of hnsw_sifts_retrieval/hnswlib/hnswalg.h above the C ABI.
//
//  * SEARCH runs on the MI355X (cvtmi_hnsw_search: one wave per query, the reference's labels and distances
//    bit for bit).  searchKnnBatch is the call that fills the GPU.
//  * CONSTRUCTION is a host algorithm in the reference (one insertion at a time, every insertion sees the
//    graph the previous ones left) and stays one here: addPoint follows hnswalg.h:584-684 -- level drawn from
//    std::default_random_engine(100) (:139-149), greedy descent through the upper levels, a best-first search
//    with ef_construction on every level the new node lives on (:152-216), neighbour selection by the
//    "closer to the query than to any already selected neighbour" rule (:283-325), mutual links with
//    re-selection when a neighbour's list is full (:340-440).  The queues are the same std::priority_queue
//    types with the same comparators as the reference's, so ties fall the same way, and saveIndex writes the
//    reference's file layout (:491-519): a graph built here is byte-identical to one built by the reference
//    from the same rows in the same order (tests/test_oracle_golden.py).  The graph is uploaded to the GPU
//    lazily, on the first search after a change.
//  * PARALLEL CONSTRUCTION: addPoint may be called from several threads, as the reference's may (hnswalg.h:178,386,594-613: a
//    guard around the element counter, one lock per node around every read and rewrite of its link lists, a global lock
//    held by an insertion that raises the top level).  The per-node lock here is a sequence lock -- writers (back links,
//    the new node's own lists) bump a counter to odd, write, bump it to even; readers copy the list and retry if the
//    counter moved -- because with 256 host threads every insertion starts at the same few upper-level nodes and a mutex
//    there turns the build into a queue of futex calls (1 M nodes: 36 s on 256 threads against 21 s on 64 before).  A new
//    node's lists are therefore visible level by level as they are written rather than when addPoint returns; each list
//    is still read and written atomically, which is all the algorithm relies on.
//    addPoints(rows, labels, n, threads) is the batch form makeIdx-style builders want: ids and levels are handed out
//    in row order first (so the level of row i is the same draw whatever the thread count), then `threads` workers
//    insert.  One thread = the sequential algorithm, the byte-identical file; more threads = a graph that depends on
//    the interleaving, as in the reference.
//
// Host data layout is the mirror's own (separate arrays for vectors / links / labels); only the file is the
// reference's interleaved block.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <atomic>
#ifdef __linux__
#include <sched.h>
#endif
#include <fstream>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../../include/cvtmi.h"

namespace hnswlib {
typedef unsigned int tableint;

template <typename dist_t> class HierarchicalNSW : public AlgorithmInterface<dist_t> {
    typedef std::pair<dist_t, tableint> Cand;
    struct ByDist {
        bool operator()(const Cand &a, const Cand &b) const { return a.first < b.first; }
    };
    typedef std::priority_queue<Cand, std::vector<Cand>, ByDist> DistHeap;  // max-heap on the distance only

public:
    HierarchicalNSW(SpaceInterface<dist_t> *s) : dev_(NULL) { bind(s); }
    HierarchicalNSW(SpaceInterface<dist_t> *s, const std::string &location, bool nmslib = false) : dev_(NULL)
    {
        (void)nmslib;
        loadIndex(location, s);
    }
    HierarchicalNSW(SpaceInterface<dist_t> *s, size_t max_elements, size_t M = 16, size_t ef_construction = 200) : dev_(NULL)
    {
        bind(s);
        cap_ = max_elements;
        M_ = M; maxM_ = M; maxM0_ = 2 * M;
        if (maxM0_ > kMaxLinks) throw std::runtime_error("cvt_amd: more than 512 links per node");
        efc_ = std::max(ef_construction, M_);
        mult_ = 1 / log(1.0 * M_);
        // (vectors and level-0 lists are left untouched: every element is written by the thread that inserts it, so on a two-socket host the
        //  pages land on the inserting threads' memory instead of all on the constructor's -- a zero-filled 512 MB array sits on one NUMA node)
        vec_.reset(new float[cap_ * dim_]);
        label_.assign(cap_, 0);
        level_.assign(cap_, 0);
        link0_.reset(new tableint[cap_ * (maxM0_ + 1)]);
        upper_.assign(cap_, std::vector<tableint>());
        seq_.reset(new std::atomic<uint32_t>[cap_]());
    }
    ~HierarchicalNSW()
    {
        if (dev_) cvtmi_hnsw_destroy(dev_);
    }

    void setEf(size_t ef) { ef_ = ef; }
    size_t ntotal() const { return count_; }

    // ---- construction (host) ----
    // thread-safe, as the reference's (hnswalg.h:591-684)
    void addPoint(void *data_point, labeltype label)
    {
        tableint id;
        int lvl;
        {
            std::lock_guard<std::mutex> g(count_guard_);
            if (count_ >= cap_) throw std::runtime_error("The number of elements exceeds the specified limit");
            id = (tableint)count_++;
            lvl = draw_level();
            dirty_ = true;
            link0_[(size_t)id * (maxM0_ + 1)] = 0u;   // (the row is uninitialised storage until insert() fills it)
        }
        try {
            insert(id, (const float *)data_point, label, lvl);
        } catch (...) {   // count_ already covers the row: the graph must not be saved, uploaded or searched in this state
            broken_ = true;
            throw;
        }
    }

    // worker count for `threads` = 0: the hardware threads this process may actually use -- its affinity mask and, in a container, its CPU
    // quota (cgroup cpu.max: the MI355X boxes show 256 logical CPUs and a quota of 16; 64 workers there take as long as 16, 256 longer)
    static unsigned usable_threads()
    {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
#ifdef __linux__
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
        std::ifstream v2("/sys/fs/cgroup/cpu.max");
        std::string quota;
        long long period = 0;
        if (v2 >> quota >> period && quota != "max" && period > 0)
            n = std::min<unsigned>(n, (unsigned)std::max(1LL, (atoll(quota.c_str()) + period - 1) / period));
        else {
            std::ifstream q1("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), p1("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
            long long q = -1, p = 0;
            if (q1 >> q && p1 >> p && q > 0 && p > 0) n = std::min<unsigned>(n, (unsigned)std::max(1LL, (q + p - 1) / p));
        }
#endif
        return n;
    }

    // rows [n][dim], labels [n] (NULL: the row number past the current count): ids and levels in row order, insertion by `threads`
    // workers (0 = one per hardware thread).  The first row of an empty graph goes in alone (it only seeds the graph).
    void addPoints(const void *rows, const labeltype *labels, size_t n, unsigned threads = 0)
    {
        if (!n) return;
        if (!threads) threads = usable_threads();
        size_t base;
        std::vector<int> lv(n);
        {
            std::lock_guard<std::mutex> g(count_guard_);
            if (count_ + n > cap_) throw std::runtime_error("The number of elements exceeds the specified limit");
            base = count_;
            count_ += n;
            dirty_ = true;
            for (size_t i = 0; i < n; ++i) lv[i] = draw_level();
            // vec_ / link0_ are uninitialised storage (first touch belongs to the worker that inserts the row): at least the link COUNT of
            // every reserved row is zeroed here, so that a row no worker reached can never be followed anywhere
            for (size_t i = 0; i < n; ++i) link0_[(base + i) * (maxM0_ + 1)] = 0u;
        }
        const float *x = (const float *)rows;
        auto one = [&](size_t i) { insert((tableint)(base + i), x + i * dim_, labels ? labels[i] : (labeltype)(base + i), lv[i]); };
        size_t first = 0;
        if (base == 0) { try { one(first++); } catch (...) { broken_ = true; throw; } }
        if (threads == 1 || n - first < 2 * (size_t)threads) {
            try {
                for (size_t i = first; i < n; ++i) one(i);
            } catch (...) { broken_ = true; throw; }
            return;
        }
        std::atomic<size_t> next(first);
        std::mutex err_mu;
        std::string err;
        auto work = [&]() {
            try {
                for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) one(i);
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> g(err_mu);
                if (err.empty()) err = e.what();
                next.store(n);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work);
        work();
        for (size_t t = 0; t < pool.size(); ++t) pool[t].join();
        if (!err.empty()) {   // count_ covers rows that were never inserted: saveIndex / upload / search refuse from here on
            broken_ = true;
            throw std::runtime_error(err);
        }
    }

    // the reference's file (:491-519)
    void saveIndex(const std::string &location)
    {
        const std::vector<char> bytes = serialise();
        std::ofstream out(location, std::ios::binary);
        out.write(bytes.data(), (std::streamsize)bytes.size());
    }

    void loadIndex(const std::string &location, SpaceInterface<dist_t> *s)
    {
        bind(s);
        std::ifstream in(location, std::ios::binary | std::ios::ate);
        if (!in) throw std::runtime_error("Cannot open file " + location);
        const std::streamsize bytes = in.tellg();
        in.seekg(0);
        std::vector<char> buf((size_t)bytes);
        in.read(buf.data(), bytes);
        parse(buf);
        ef_ = 10;  // hnswalg.h:560
        dirty_ = true;
    }

    // ---- search (device) ----
    std::priority_queue<std::pair<dist_t, labeltype> > searchKnn(void *query_data, size_t k)
    {
        return searchKnnBatch(query_data, 1, k)[0];
    }

    // nq queries, contiguous [nq][dim] floats
    std::vector<std::priority_queue<std::pair<dist_t, labeltype> > > searchKnnBatch(const void *queries, size_t nq, size_t k)
    {
        upload();
        std::vector<float> d(nq * k);
        std::vector<int64_t> l(nq * k);
        if (cvtmi_hnsw_search(dev_, (const float *)queries, (int64_t)nq, (int)k, (int)ef_, d.data(), l.data()) != CVTMI_OK)
            throw std::runtime_error(std::string("cvt_amd: ") + cvtmi_last_error());
        std::vector<std::priority_queue<std::pair<dist_t, labeltype> > > out(nq);
        for (size_t q = 0; q < nq; ++q)
            for (size_t i = 0; i < k && l[q * k + i] >= 0; ++i) out[q].push(std::make_pair((dist_t)d[q * k + i], (labeltype)l[q * k + i]));
        return out;
    }

private:
    int draw_level()   // under count_guard_: the i-th element gets the i-th draw (hnswalg.h:139-149)
    {
        std::uniform_real_distribution<double> u01(0.0, 1.0);
        return (int)(-log(u01(rng_)) * mult_);
    }

    void insert(tableint id, const float *x, labeltype label, int lvl)
    {
        // an insertion that raises the top level keeps the global lock until it is the new entry point (:610-613, :676-679)
        std::unique_lock<std::mutex> top_lock(global_, std::defer_lock);
        uint64_t tv = top_.load(std::memory_order_acquire);
        if (lvl > (int)(tv & 0xffffffffu) - 1) {
            top_lock.lock();
            tv = top_.load(std::memory_order_acquire);
            if (lvl <= (int)(tv & 0xffffffffu) - 1) top_lock.unlock();
        }
        const int top = (int)(tv & 0xffffffffu) - 1;
        const bool seed = (tv >> 32) == 0;
        tableint cur = seed ? 0 : (tableint)((tv >> 32) - 1);
        level_[id] = lvl;
        std::copy(x, x + dim_, &vec_[(size_t)id * dim_]);
        label_[id] = label;
        std::fill(&link0_[(size_t)id * (maxM0_ + 1)], &link0_[(size_t)(id + 1) * (maxM0_ + 1)], 0u);
        upper_[id].assign((size_t)lvl * (maxM_ + 1), 0u);
        if (seed) {  // the first element only seeds the graph
            entry_ = (int)id;
            top_level_ = lvl;
            top_.store(((uint64_t)id + 1) << 32 | (uint32_t)(lvl + 1), std::memory_order_release);
            return;
        }
        if (lvl < top) {  // greedy descent to the first level the new node lives on
            dist_t cd = dist(x, row(cur));
            tableint nb[kMaxLinks];
            for (int l = top; l > lvl; --l) {
                bool moved = true;
                while (moved) {
                    moved = false;
                    const tableint cnt = snapshot(cur, l, nb);
                    for (tableint i = 0; i < cnt; ++i) {
                        const dist_t d = dist(x, row(nb[i]));
                        if (d < cd) { cd = d; cur = nb[i]; moved = true; }
                    }
                }
            }
        }
        for (int l = std::min(lvl, top); l >= 0; --l) {  // every search starts from the same entry (:660-667)
            DistHeap found = search_level(cur, x, l);
            connect(id, found, l);
        }
        if (lvl > top) {
            entry_ = (int)id; top_level_ = lvl;
            top_.store(((uint64_t)id + 1) << 32 | (uint32_t)(lvl + 1), std::memory_order_release);
        }
    }

    // a node's neighbours on level l, copied under its lock (the reference reads them in place under the same lock, :178-186)
    enum { kMaxLinks = 512 };
    tableint snapshot(tableint node, int l, tableint *out, uint32_t *seen_seq = NULL)
    {
        const tableint *ll = links(node, l);
        for (;;) {
            const uint32_t s0 = seq_[node].load(std::memory_order_acquire);
            if (s0 & 1u) { relax(); continue; }
            const tableint cnt = std::min<tableint>(__atomic_load_n(ll, __ATOMIC_RELAXED), (tableint)kMaxLinks);
            for (tableint i = 0; i < cnt; ++i) out[i] = __atomic_load_n(ll + 1 + i, __ATOMIC_RELAXED);
            std::atomic_thread_fence(std::memory_order_acquire);
            if (seq_[node].load(std::memory_order_relaxed) == s0) {
                if (seen_seq) *seen_seq = s0;
                return cnt;
            }
        }
    }
    static void relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    struct WriteLock {   // exclusive among writers of one node; readers see the counter odd and wait
        std::atomic<uint32_t> &s;
        explicit WriteLock(std::atomic<uint32_t> &seq) : s(seq)
        {
            uint32_t v = s.load(std::memory_order_relaxed);
            while ((v & 1u) || !s.compare_exchange_weak(v, v + 1, std::memory_order_acquire, std::memory_order_relaxed)) {
                relax();
                v = s.load(std::memory_order_relaxed);
            }
            std::atomic_thread_fence(std::memory_order_release);
        }
        ~WriteLock() { s.fetch_add(1, std::memory_order_release); }
    };
    static void put_link(tableint *p, tableint v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }

    void bind(SpaceInterface<dist_t> *s)
    {
        dim_ = s->get_data_size() / sizeof(float);
        metric_ = s->device_metric();
        if (metric_ != CVTMI_METRIC_IP && metric_ != CVTMI_METRIC_L2F)
            throw std::runtime_error("cvt_amd: HierarchicalNSW needs InnerProductSpace or L2Space");
    }
    const float *row(tableint id) const { return &vec_[(size_t)id * dim_]; }
    tableint *links(tableint id, int l) { return l == 0 ? &link0_[(size_t)id * (maxM0_ + 1)] : &upper_[id][(size_t)(l - 1) * (maxM_ + 1)]; }

    // Distances in the summation order of the reference's own functions for this width (space_ip.h: one 4-lane
    // accumulator for D % 4 == 0, the AVX branches are compiled out; space_l2.h: 8 lanes for D % 16 == 0, 4 for
    // D % 4 == 0), so that construction takes the decisions the reference takes.
    dist_t dist(const float *a, const float *b) const
    {
        const size_t n = dim_;
        if (n % 4 != 0) {
            float r = 0;
            if (metric_ == CVTMI_METRIC_IP) { for (size_t i = 0; i < n; ++i) r += a[i] * b[i]; return 1.0f - r; }
            for (size_t i = 0; i < n; ++i) { const float t = a[i] - b[i]; r += t * t; }
            return r;
        }
        const size_t lanes = (metric_ == CVTMI_METRIC_IP) ? 4 : (n % 16 == 0 ? 8 : 4);
        float acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (metric_ == CVTMI_METRIC_IP) {
            for (size_t i = 0; i < n; i += lanes)
                for (size_t l = 0; l < lanes; ++l) acc[l] += a[i + l] * b[i + l];
        } else {
            for (size_t i = 0; i < n; i += lanes)
                for (size_t l = 0; l < lanes; ++l) { const float t = a[i + l] - b[i + l]; acc[l] += t * t; }
        }
        float s = acc[0];
        for (size_t l = 1; l < lanes; ++l) s += acc[l];
        return metric_ == CVTMI_METRIC_IP ? 1.0f - s : s;
    }

    // best-first search on one level with ef_construction results (:152-216); visited marks are per thread
    DistHeap search_level(tableint start, const float *x, int l)
    {
        static thread_local std::vector<unsigned short> seen;
        static thread_local unsigned short stamp = 0;
        if (seen.size() < cap_) { seen.assign(cap_, 0); stamp = 0; }
        if (++stamp == 0) { std::fill(seen.begin(), seen.end(), 0); stamp = 1; }
        // (the two queues get their storage up front: growing from empty cost a dozen reallocations per level and insertion)
        std::vector<Cand> best_store, frontier_store;
        best_store.reserve(efc_ + 2); frontier_store.reserve(4 * efc_ + 64);
        DistHeap best(ByDist(), std::move(best_store)), frontier(ByDist(), std::move(frontier_store));
        const dist_t d0 = dist(x, row(start));
        best.emplace(d0, start);
        frontier.emplace(-d0, start);
        seen[start] = stamp;
        dist_t bound = d0;
        tableint nbs[kMaxLinks];
        while (!frontier.empty()) {
            const Cand c = frontier.top();
            if (-c.first > bound) break;
            frontier.pop();
            const tableint cnt = snapshot(c.second, l, nbs);
            // (the neighbours' vectors are scattered over hundreds of megabytes: ask for the next one while this one is scored -- the
            //  reference does the same with _mm_prefetch, hnswalg.h:190-205)
            if (cnt) { __builtin_prefetch(&seen[nbs[0]]); __builtin_prefetch(row(nbs[0])); }
            for (tableint j = 0; j < cnt; ++j) {
                const tableint nb = nbs[j];
                if (j + 1 < cnt) {
                    __builtin_prefetch(&seen[nbs[j + 1]]);
                    const char *nx = (const char *)row(nbs[j + 1]);
                    __builtin_prefetch(nx); __builtin_prefetch(nx + 64); __builtin_prefetch(nx + 128); __builtin_prefetch(nx + 192);
                }
                if (seen[nb] == stamp) continue;
                seen[nb] = stamp;
                const dist_t d = dist(x, row(nb));
                if (best.top().first > d || best.size() < efc_) {
                    frontier.emplace(-d, nb);
                    best.emplace(d, nb);
                    if (best.size() > efc_) best.pop();
                    bound = best.top().first;
                }
            }
        }
        return best;
    }

    // keep at most m of the candidates: nearest first, a candidate survives when it is closer to the query point
    // than to every neighbour kept so far (:283-325).  Untouched when fewer than m candidates.
    void select(DistHeap &cands, size_t m)
    {
        if (cands.size() < m) return;
        std::priority_queue<Cand> nearest;  // (negated distance, id): largest first = nearest first
        while (!cands.empty()) { nearest.emplace(-cands.top().first, cands.top().second); cands.pop(); }
        std::vector<Cand> kept;
        while (!nearest.empty() && kept.size() < m) {
            const Cand c = nearest.top();
            nearest.pop();
            const dist_t to_query = -c.first;
            bool ok = true;
            for (size_t i = 0; i < kept.size() && ok; ++i)
                if (dist(row(kept[i].second), row(c.second)) < to_query) ok = false;
            if (ok) kept.push_back(c);
        }
        for (size_t i = 0; i < kept.size(); ++i) cands.emplace(-kept[i].first, kept[i].second);
    }

    // links of the new node on level l, and the back links (:340-440)
    void connect(tableint id, DistHeap found, int l)
    {
        const size_t room = l ? maxM_ : maxM0_;
        select(found, M_);
        if (found.size() > M_) throw std::runtime_error("Should be not be more than M_ candidates returned by the heuristic");
        std::vector<tableint> chosen;
        while (!found.empty()) { chosen.push_back(found.top().second); found.pop(); }
        tableint *mine = links(id, l);
        {
            WriteLock w(seq_[id]);
            for (size_t i = 0; i < chosen.size(); ++i)
                if (l > level_[chosen[i]]) throw std::runtime_error("Trying to make a link on a non-existent level");
            // Parallel builds only: a node that met this one on the level above starts its search of level l HERE and may already have
            // linked itself to the still empty list (the reference makes it wait on this node's lock instead, hnswalg.h:602 -- two
            // nodes that are each other's entry on the same level wait for ever).  Its links are kept: the list becomes the chosen
            // neighbours plus those early links, cut back by the selection rule if that is more than the level holds.
            const size_t early = mine[0];
            std::vector<tableint> all(chosen);
            for (size_t j = 0; j < early; ++j)
                if (std::find(chosen.begin(), chosen.end(), mine[1 + j]) == chosen.end()) all.push_back(mine[1 + j]);
            if (all.size() > room) {
                DistHeap pool;
                for (size_t j = 0; j < all.size(); ++j) pool.emplace(dist(row(all[j]), row(id)), all[j]);
                select(pool, room);
                all.clear();
                while (!pool.empty()) { all.push_back(pool.top().second); pool.pop(); }
            }
            for (size_t i = 0; i < all.size(); ++i) put_link(mine + 1 + i, all[i]);
            put_link(mine, (tableint)all.size());
        }
        for (size_t i = 0; i < chosen.size(); ++i) {
            const tableint other = chosen[i];
            if (other == id) throw std::runtime_error("Trying to connect an element to itself");
            // (:386) the neighbour's list is rewritten from a consistent copy; the new list is computed outside the lock (a full list
            // costs up to room^2 / 2 distances) and stored only if the list has not changed meanwhile
            tableint *ol = links(other, l);
            for (;;) {
                tableint old_list[kMaxLinks], new_list[kMaxLinks];
                uint32_t s0;
                const size_t have = snapshot(other, l, old_list, &s0);
                if (have > room) throw std::runtime_error("Bad value of sz_link_list_other");
                if (std::find(old_list, old_list + have, id) != old_list + have) break;   // (parallel builds: it linked to this node first)
                size_t cnt = 0;
                if (have == room) {  // full: re-select among its neighbours and the new node
                    DistHeap pool;
                    pool.emplace(dist(row(id), row(other)), id);
                    for (size_t j = 0; j < have; ++j) pool.emplace(dist(row(old_list[j]), row(other)), old_list[j]);
                    select(pool, room);
                    while (!pool.empty()) { new_list[cnt++] = pool.top().second; pool.pop(); }
                }
                uint32_t expect = s0;
                if (!seq_[other].compare_exchange_strong(expect, s0 + 1, std::memory_order_acquire, std::memory_order_relaxed)) continue;
                std::atomic_thread_fence(std::memory_order_release);
                if (have < room) {
                    put_link(ol + 1 + have, id);
                    put_link(ol, (tableint)(have + 1));
                } else {
                    for (size_t j = 0; j < cnt; ++j) put_link(ol + 1 + j, new_list[j]);
                    put_link(ol, (tableint)cnt);
                }
                seq_[other].fetch_add(1, std::memory_order_release);
                break;
            }
        }
    }

    // ---- the reference's file image ----
    template <typename T> static void put(std::vector<char> &b, const T &v)
    {
        const char *p = (const char *)&v;
        b.insert(b.end(), p, p + sizeof(T));
    }
    std::vector<char> serialise() const
    {
        if (broken_) throw std::runtime_error("HierarchicalNSW: an insertion failed part-way, the graph holds rows that were never linked; rebuild it");
        const size_t link0_bytes = maxM0_ * sizeof(tableint) + sizeof(unsigned int);
        const size_t per_elem = link0_bytes + dim_ * sizeof(float) + sizeof(labeltype);
        const size_t off_level0 = 0, off_data = link0_bytes, off_label = link0_bytes + dim_ * sizeof(float);
        const size_t upper_bytes = maxM_ * sizeof(tableint) + sizeof(unsigned int);
        std::vector<char> b;
        b.reserve(96 + cap_ * per_elem + cap_ * 4);
        put(b, off_level0); put(b, (size_t)cap_); put(b, (size_t)count_); put(b, per_elem); put(b, off_label); put(b, off_data);
        put(b, (int)top_level_); put(b, (tableint)entry_);
        put(b, (size_t)maxM_); put(b, (size_t)maxM0_); put(b, (size_t)M_); put(b, (double)mult_); put(b, (size_t)efc_);
        const size_t base = b.size();
        b.resize(base + cap_ * per_elem, 0);  // unused elements stay zero (the reference leaves them as malloc returned them)
        for (size_t i = 0; i < count_; ++i) {
            char *e = &b[base + i * per_elem];
            memcpy(e, &link0_[i * (maxM0_ + 1)], link0_bytes);
            memcpy(e + off_data, &vec_[i * dim_], dim_ * sizeof(float));
            memcpy(e + off_label, &label_[i], sizeof(labeltype));
        }
        for (size_t i = 0; i < cap_; ++i) {
            const unsigned int sz = (i < count_ && level_[i] > 0) ? (unsigned int)(upper_bytes * (size_t)level_[i]) : 0u;
            put(b, sz);
            if (sz) {
                const char *p = (const char *)upper_[i].data();
                b.insert(b.end(), p, p + sz);
            }
        }
        return b;
    }
    void parse(const std::vector<char> &buf)
    {
        broken_ = false;
        if (buf.size() < 96) throw std::runtime_error("cvt_amd: not a saveIndex file");
        const char *p = buf.data();
        size_t off_level0, per_elem, off_label, off_data, a, b2, c, efc;
        int maxlevel; tableint ep; double mult;
        auto get = [&](void *dst, size_t nb) { memcpy(dst, p, nb); p += nb; };
        get(&off_level0, 8); get(&cap_, 8); get(&count_, 8); get(&per_elem, 8); get(&off_label, 8); get(&off_data, 8);
        get(&maxlevel, 4); get(&ep, 4); get(&a, 8); get(&b2, 8); get(&c, 8); get(&mult, 8); get(&efc, 8);
        maxM_ = a; maxM0_ = b2; M_ = c; mult_ = mult; efc_ = efc; top_level_ = maxlevel; entry_ = count_ ? (int)ep : -1;
        top_.store(entry_ >= 0 ? ((uint64_t)entry_ + 1) << 32 | (uint32_t)(top_level_ + 1) : 0);
        if (per_elem != 4 + 4 * maxM0_ + 4 * dim_ + 8 || buf.size() < 96 + cap_ * per_elem)
            throw std::runtime_error("cvt_amd: saveIndex file does not match the space's dimension");
        vec_.reset(new float[cap_ * dim_]); label_.assign(cap_, 0); level_.assign(cap_, 0);
        link0_.reset(new tableint[cap_ * (maxM0_ + 1)]); upper_.assign(cap_, std::vector<tableint>());
        seq_.reset(new std::atomic<uint32_t>[cap_]());
        if (maxM0_ > kMaxLinks) throw std::runtime_error("cvt_amd: more than 512 links per node");
        for (size_t i = 0; i < count_; ++i) {
            const char *e = p + i * per_elem;
            memcpy(&link0_[i * (maxM0_ + 1)], e + off_level0, 4 + 4 * maxM0_);
            memcpy(&vec_[i * dim_], e + off_data, 4 * dim_);
            memcpy(&label_[i], e + off_label, 8);
        }
        p += cap_ * per_elem;
        const size_t upper_bytes = 4 * maxM_ + 4;
        for (size_t i = 0; i < cap_; ++i) {
            if (p + 4 > buf.data() + buf.size()) throw std::runtime_error("cvt_amd: truncated saveIndex file");
            unsigned int sz; get(&sz, 4);
            if (sz) {
                if (p + sz > buf.data() + buf.size() || sz % upper_bytes) throw std::runtime_error("cvt_amd: corrupt saveIndex file");
                if (i < count_) {
                    level_[i] = (int)(sz / upper_bytes);
                    upper_[i].resize(sz / 4);
                    memcpy(upper_[i].data(), p, sz);
                }
                p += sz;
            }
        }
    }
    void upload()
    {
        if (!dirty_ && dev_) return;
        if (dev_) { cvtmi_hnsw_destroy(dev_); dev_ = NULL; }
        const std::vector<char> bytes = serialise();
        if (cvtmi_hnsw_load(bytes.data(), (int64_t)bytes.size(), metric_, (int)dim_, &dev_) != CVTMI_OK)
            throw std::runtime_error(std::string("cvt_amd: ") + cvtmi_last_error());
        dirty_ = false;
    }

    size_t dim_ = 0, cap_ = 0, count_ = 0, M_ = 16, maxM_ = 16, maxM0_ = 32, efc_ = 200, ef_ = 10;
    int metric_ = 0, entry_ = -1, top_level_ = -1;
    double mult_ = 0;
    std::unique_ptr<float[]> vec_;                // [cap][dim]; elements past count_ are never read
    std::vector<labeltype> label_;
    std::vector<int> level_;
    bool broken_ = false;                         // an insertion threw after count_ had been advanced
    std::unique_ptr<tableint[]> link0_;           // [cap][maxM0 + 1]: count, neighbours
    std::vector<std::vector<tableint> > upper_;   // per node: levels x (maxM + 1)
    std::unique_ptr<std::atomic<uint32_t>[]> seq_;   // one sequence lock per node (the reference: link_list_locks_, hnswalg.h:76)
    std::mutex count_guard_, global_;
    std::atomic<uint64_t> top_{0};                     // (entry + 1) << 32 | (top level + 1): what an insertion starts from, read without the global lock
    std::default_random_engine rng_ = std::default_random_engine(100);  // hnswalg.h:139
    cvtmi_hnsw_t dev_;
    bool dirty_ = true;
};
}  // namespace hnswlib
