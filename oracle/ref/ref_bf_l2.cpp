// Harness over the reference's hnsw_sifts_retrieval/hnswlib headers (L2Space, L2SpaceI and that
// tree's BruteforceSearch), compiled in place -> oracle/_ref/libref_bf_l2.so.  Test infra only.
// Separate shared object from ref_bf_ip: both trees define namespace hnswlib classes of the same name.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "hnswlib.h"

extern "C" __attribute__((visibility("default")))
int ref_bf_l2f_search(int D, const float *data, const int64_t *labels, int64_t n, const float *queries, int64_t nq,
                      int64_t k, float *out_d, int64_t *out_label)
{
    using namespace hnswlib;
    L2Space space((size_t)D);
    BruteforceSearch<float> alg(&space, (size_t)n);
    std::vector<float> tmp(D);
    for (int64_t i = 0; i < n; ++i) {
        memcpy(tmp.data(), data + i * D, sizeof(float) * D);
        alg.addPoint((void *)tmp.data(), (labeltype)(labels ? labels[i] : i));
    }
    for (int64_t q = 0; q < nq; ++q) {
        memcpy(tmp.data(), queries + q * D, sizeof(float) * D);
        std::priority_queue<std::pair<float, labeltype> > res = alg.searchKnn((void *)tmp.data(), (size_t)k);
        int64_t m = (int64_t)res.size();
        for (int64_t i = m - 1; i >= 0; --i) {
            out_d[q * k + i] = res.top().first; out_label[q * k + i] = (int64_t)res.top().second; res.pop();
        }
    }
    return 0;
}

extern "C" __attribute__((visibility("default")))
int ref_bf_l2u8_search(int D, const uint8_t *data, const int64_t *labels, int64_t n, const uint8_t *queries,
                       int64_t nq, int64_t k, int32_t *out_d, int64_t *out_label)
{
    using namespace hnswlib;
    L2SpaceI space((size_t)D);
    BruteforceSearch<int> alg(&space, (size_t)n);
    std::vector<uint8_t> tmp(D);
    for (int64_t i = 0; i < n; ++i) {
        memcpy(tmp.data(), data + i * D, (size_t)D);
        alg.addPoint((void *)tmp.data(), (labeltype)(labels ? labels[i] : i));
    }
    for (int64_t q = 0; q < nq; ++q) {
        memcpy(tmp.data(), queries + q * D, (size_t)D);
        std::priority_queue<std::pair<int, labeltype> > res = alg.searchKnn((void *)tmp.data(), (size_t)k);
        int64_t m = (int64_t)res.size();
        for (int64_t i = m - 1; i >= 0; --i) {
            out_d[q * k + i] = res.top().first; out_label[q * k + i] = (int64_t)res.top().second; res.pop();
        }
    }
    return 0;
}
