// Harness over the reference's brute_force_search/src headers (BruteforceSearch<float> +
// InnerProductSpace), compiled in place -> oracle/_ref/libref_bf_ip.so.  Test infrastructure only.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "brutoforce.hpp"
#include "space_ip.hpp"

extern "C" __attribute__((visibility("default")))
int ref_bf_ip_search(int D, const float *data, const int64_t *labels, int64_t n, const float *queries, int64_t nq,
                     int64_t k, float *out_d, int64_t *out_label)
{
    using namespace hnswlib;
    InnerProductSpace space((size_t)D);
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
        for (int64_t i = m - 1; i >= 0; --i) {  // heap pops worst first
            out_d[q * k + i] = res.top().first; out_label[q * k + i] = (int64_t)res.top().second; res.pop();
        }
    }
    return 0;
}

// the same with the two phases timed apart (bench.py's cpu_baseline of the fp32 flat search: kind "reference")
#include <chrono>
extern "C" __attribute__((visibility("default")))
int ref_bf_ip_search_timed(int D, const float *data, int64_t n, const float *queries, int64_t nq, int64_t k, float *out_d,
                           int64_t *out_label, double *t_add, double *t_search)
{
    using namespace hnswlib;
    typedef std::chrono::steady_clock clk;
    InnerProductSpace space((size_t)D);
    BruteforceSearch<float> alg(&space, (size_t)n);
    clk::time_point t0 = clk::now();
    for (int64_t i = 0; i < n; ++i) alg.addPoint((void *)(data + i * D), (labeltype)i);
    clk::time_point t1 = clk::now();
    for (int64_t q = 0; q < nq; ++q) {
        std::priority_queue<std::pair<float, labeltype> > res = alg.searchKnn((void *)(queries + q * D), (size_t)k);
        int64_t m = (int64_t)res.size();
        for (int64_t i = m - 1; i >= 0; --i) {
            out_d[q * k + i] = res.top().first; out_label[q * k + i] = (int64_t)res.top().second; res.pop();
        }
    }
    clk::time_point t2 = clk::now();
    *t_add = std::chrono::duration<double>(t1 - t0).count();
    *t_search = std::chrono::duration<double>(t2 - t1).count();
    return 0;
}

extern "C" __attribute__((visibility("default")))
float ref_ip_dist(int which, const float *a, const float *b, int D)
{
    size_t d = (size_t)D;
    if (which == 0) return hnswlib::InnerProduct(a, b, &d);
    if (which == 4) return hnswlib::InnerProductSIMD4Ext(a, b, &d);
    return hnswlib::InnerProductSIMD16Ext(a, b, &d);
}
