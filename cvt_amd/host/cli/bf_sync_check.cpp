// bf_sync_check -- self-check of hnswlib::BruteforceSearch's device synchronisation (cvt_amd/host/hnswlib/bruteforce.h): searches
// interleaved with addPoint (ascending labels: appended to the device copy; a smaller label: full re-sort) and removePoint must
// answer exactly like an index built from scratch over the same rows.  Prints "OK" and returns 0, or the first difference.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../hnswlib/hnswlib.h"

using namespace hnswlib;

static bool same(BruteforceSearch<float> &a, const std::vector<float> &rows, const std::vector<labeltype> &labels, size_t dim,
                 SpaceInterface<float> *space, const std::vector<float> &q, size_t nq, size_t k, const char *what)
{
    BruteforceSearch<float> fresh(space, labels.size() + 1);
    for (size_t i = 0; i < labels.size(); ++i) fresh.addPoint((void *)&rows[i * dim], labels[i]);
    std::vector<float> d1(nq * k), d2(nq * k);
    std::vector<int64_t> l1(nq * k), l2(nq * k);
    a.searchKnnBatch(q.data(), nq, k, d1.data(), l1.data());
    fresh.searchKnnBatch(q.data(), nq, k, d2.data(), l2.data());
    for (size_t i = 0; i < nq * k; ++i)
        if (l1[i] != l2[i] || d1[i] != d2[i]) {
            printf("MISMATCH after %s: entry %zu: (%g, %lld) vs (%g, %lld)\n", what, i, d1[i], (long long)l1[i], d2[i], (long long)l2[i]);
            return false;
        }
    return true;
}

int main()
{
    const size_t dim = 64, nq = 7, k = 10;
    std::mt19937 rng(7);
    std::normal_distribution<float> g(0.f, 1.f);
    L2Space space(dim);
    BruteforceSearch<float> idx(&space, 5000);
    std::vector<float> rows;
    std::vector<labeltype> labels;
    std::vector<float> q(nq * dim);
    for (auto &v : q) v = g(rng);
    auto add = [&](labeltype lab) {
        std::vector<float> r(dim);
        for (auto &v : r) v = g(rng);
        if (lab % 7 == 0 && !rows.empty()) r.assign(rows.begin(), rows.begin() + dim);  // duplicates: (distance, label) ties
        idx.addPoint(r.data(), lab);
        rows.insert(rows.end(), r.begin(), r.end());
        labels.push_back(lab);
    };
    for (labeltype l = 0; l < 1000; ++l) add(l);
    if (!same(idx, rows, labels, dim, &space, q, nq, k, "first 1000 rows")) return 1;
    for (labeltype l = 1000; l < 1500; ++l) add(l * 3);                       // ascending: appended
    if (!same(idx, rows, labels, dim, &space, q, nq, k, "ascending append")) return 1;
    add(100017); add(5000001);                                                // still ascending
    add(1234567);                                                             // below the largest uploaded label: full re-sort
    if (!same(idx, rows, labels, dim, &space, q, nq, k, "out-of-order label")) return 1;
    for (labeltype l = 6000000; l < 6000100; ++l) add(l);
    if (!same(idx, rows, labels, dim, &space, q, nq, k, "append after re-sort")) return 1;
    // removePoint moves the last row into the hole (brutoforce.hpp:58-70): mirror that in the expectation
    for (int t = 0; t < 5; ++t) {
        const size_t victim = (size_t)(rng() % labels.size());
        idx.removePoint(labels[victim]);
        const size_t last = labels.size() - 1;
        labels[victim] = labels[last];
        for (size_t e = 0; e < dim; ++e) rows[victim * dim + e] = rows[last * dim + e];
        labels.pop_back();
        rows.resize(rows.size() - dim);
    }
    if (!same(idx, rows, labels, dim, &space, q, nq, k, "removePoint")) return 1;
    add(7000000);
    if (!same(idx, rows, labels, dim, &space, q, nq, k, "append after removePoint")) return 1;
    printf("OK\n");
    return 0;
}
