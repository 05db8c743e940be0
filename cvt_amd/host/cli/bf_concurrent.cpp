// bf_concurrent -- T host threads calling BruteforceSearch::searchKnn on ONE index at the same time (the reference's searchKnn is
// a pure read and de-facto concurrent, brutoforce.hpp:73-93):
//   bf_concurrent [rows] [dim] [threads] [queries per thread] [k]
// checks that every thread gets what a single thread gets, and prints the one-thread and the T-thread rate.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../hnswlib/hnswlib.h"

static uint32_t rs = 7u;
static float rnd() { rs = rs * 1664525u + 1013904223u; return ((int)((rs >> 8) % 20001) - 10000) * 1e-4f; }

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? atol(argv[1]) : 1000000, dim = argc > 2 ? atol(argv[2]) : 128;
    const int T = argc > 3 ? atoi(argv[3]) : 4, per = argc > 4 ? atoi(argv[4]) : 64, k = argc > 5 ? atoi(argv[5]) : 100;
    hnswlib::InnerProductSpace space(dim);
    hnswlib::BruteforceSearch<float> alg(&space, n);
    std::vector<float> row(dim);
    for (size_t i = 0; i < n; ++i) {
        for (size_t d = 0; d < dim; ++d) row[d] = rnd();
        alg.addPoint(row.data(), i);
    }
    std::vector<float> q((size_t)T * per * dim);
    for (size_t i = 0; i < q.size(); ++i) q[i] = rnd();
    typedef std::priority_queue<std::pair<float, hnswlib::labeltype> > heap_t;
    std::vector<heap_t> ref((size_t)T * per), got((size_t)T * per);
    alg.searchKnn(q.data(), k);   // uploads the rows
    typedef std::chrono::steady_clock clk;
    clk::time_point t0 = clk::now();
    for (int i = 0; i < T * per; ++i) ref[i] = alg.searchKnn(&q[(size_t)i * dim], k);
    const double t_one = std::chrono::duration<double>(clk::now() - t0).count();
    t0 = clk::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            for (int i = t * per; i < (t + 1) * per; ++i) got[i] = alg.searchKnn(&q[(size_t)i * dim], k);
        });
    for (auto &x : th) x.join();
    const double t_par = std::chrono::duration<double>(clk::now() - t0).count();
    for (int i = 0; i < T * per; ++i) {
        heap_t a = ref[i], b = got[i];
        if (a.size() != b.size()) { printf("MISMATCH (size) at query %d\n", i); return 1; }
        while (!a.empty()) {
            if (a.top() != b.top()) { printf("MISMATCH at query %d\n", i); return 1; }
            a.pop(); b.pop();
        }
    }
    printf("rows %zu dim %zu k %d: 1 thread %.1f queries/s, %d threads %.1f queries/s (x%.2f) OK\n", n, dim, k, T * per / t_one, T,
           T * per / t_par, t_one / t_par);
    return 0;
}
