// Harness over the reference's hnsw_sifts_retrieval/hnswlib (HierarchicalNSW, hnswalg.h), compiled in
// place -> oracle/_ref/libref_hnsw.so.  Test infra only: builds a graph exactly as siftsIndex.cpp does
// (sequential addPoint, labels = row numbers unless given), saves it with the reference's own saveIndex, and
// answers searchKnn on a saved file.  The reference prints debug text from its constructors: stdout is
// silenced around the calls.
#include <atomic>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
#include <fcntl.h>
#include "hnswlib.h"
#include "hnswalg.h"

namespace {
struct Quiet {
    int saved;
    Quiet() { fflush(stdout); saved = dup(1); int nul = open("/dev/null", O_WRONLY); dup2(nul, 1); close(nul); }
    ~Quiet() { fflush(stdout); dup2(saved, 1); close(saved); }
};
}

// metric: 0 = InnerProductSpace, 1 = L2Space
extern "C" __attribute__((visibility("default")))
int ref_hnsw_build(int metric, int D, const float *data, const int64_t *labels, int64_t n, int64_t max_elements, int M,
                   int ef_construction, const char *out_path)
{
    using namespace hnswlib;
    Quiet q;
    SpaceInterface<float> *space = metric == 0 ? (SpaceInterface<float> *)new InnerProductSpace((size_t)D)
                                               : (SpaceInterface<float> *)new L2Space((size_t)D);
    HierarchicalNSW<float> *alg = new HierarchicalNSW<float>(space, (size_t)max_elements, (size_t)M, (size_t)ef_construction);
    std::vector<float> tmp((size_t)D);
    for (int64_t i = 0; i < n; ++i) {
        memcpy(tmp.data(), data + i * D, sizeof(float) * (size_t)D);
        alg->addPoint((void *)tmp.data(), (labeltype)(labels ? labels[i] : i));
    }
    alg->saveIndex(std::string(out_path));
    delete alg;
    return 0;
}

// The same, with addPoint called from `threads` host threads (rows handed out by an atomic counter after the first one):
// HierarchicalNSW::addPoint guards the element counter, every node's link list and the entry point with its own mutexes
// (hnswalg.h:109-111, :178, :386, :446, :594-613), which is what hnswlib's own parallel builders rely on.  The graph then
// depends on the interleaving (internal ids follow arrival order), so callers compare searches on the saved FILE.
extern "C" __attribute__((visibility("default")))
int ref_hnsw_build_mt(int metric, int D, const float *data, const int64_t *labels, int64_t n, int64_t max_elements, int M,
                      int ef_construction, const char *out_path, int threads)
{
    using namespace hnswlib;
    Quiet q;
    SpaceInterface<float> *space = metric == 0 ? (SpaceInterface<float> *)new InnerProductSpace((size_t)D)
                                               : (SpaceInterface<float> *)new L2Space((size_t)D);
    HierarchicalNSW<float> *alg = new HierarchicalNSW<float>(space, (size_t)max_elements, (size_t)M, (size_t)ef_construction);
    if (n > 0) {
        std::vector<float> first(data, data + D);
        alg->addPoint((void *)first.data(), (labeltype)(labels ? labels[0] : 0));
    }
    std::atomic<int64_t> next(1);
    std::atomic<int> failed(0);
    std::vector<std::thread> pool;
    if (threads < 1) threads = 1;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&]() {
            std::vector<float> tmp((size_t)D);
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n) break;
                memcpy(tmp.data(), data + i * D, sizeof(float) * (size_t)D);
                try { alg->addPoint((void *)tmp.data(), (labeltype)(labels ? labels[i] : i)); } catch (...) { failed = 1; }
            }
        });
    for (auto &th : pool) th.join();
    alg->saveIndex(std::string(out_path));
    delete alg;
    return failed.load();
}

extern "C" __attribute__((visibility("default")))
int ref_hnsw_search(int metric, int D, const char *index_path, const float *queries, int64_t nq, int64_t k, int64_t ef,
                    float *out_d, int64_t *out_label)
{
    using namespace hnswlib;
    Quiet q;
    SpaceInterface<float> *space = metric == 0 ? (SpaceInterface<float> *)new InnerProductSpace((size_t)D)
                                               : (SpaceInterface<float> *)new L2Space((size_t)D);
    HierarchicalNSW<float> *alg = new HierarchicalNSW<float>(space, std::string(index_path), false);
    alg->setEf((size_t)ef);
    std::vector<float> tmp((size_t)D);
    for (int64_t qi = 0; qi < nq; ++qi) {
        memcpy(tmp.data(), queries + qi * D, sizeof(float) * (size_t)D);
        std::priority_queue<std::pair<float, labeltype> > res = alg->searchKnn((void *)tmp.data(), (size_t)k);
        const int64_t m = (int64_t)res.size();
        for (int64_t i = 0; i < k; ++i) { out_d[qi * k + i] = 0.0f; out_label[qi * k + i] = -1; }
        for (int64_t i = m - 1; i >= 0; --i) {  // ascending (dist, label)
            out_d[qi * k + i] = res.top().first; out_label[qi * k + i] = (int64_t)res.top().second; res.pop();
        }
    }
    return 0;  // alg is leaked on purpose: the reference's destructor frees per cur_element_count on a loaded index
}
