// opq_concurrent -- T host threads calling IVFOPQ::SearchTopK on ONE index at the same time.  The reference's QueryThrehold only
// reads the index (opq/src/IVFOPQ.cpp:322-422), so concurrent queries are de-facto legal there; here every search leases its own
// scratch set and stream from the handle (csrc/api.hip: OpqLease), only add / reset and the lazily built row copy are exclusive:
//   opq_concurrent [rows] [threads] [calls per thread] [queries per call] [k]
// checks that every thread gets what a single thread gets, and prints the one-thread and the T-thread rate.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../IVFOPQ.h"

static uint32_t rs = 11u;
static float rnd() { rs = rs * 1664525u + 1013904223u; return ((int)((rs >> 8) % 20001) - 10000) * 1e-4f; }

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int T = argc > 2 ? atoi(argv[2]) : 4, calls = argc > 3 ? atoi(argv[3]) : 64, nq = argc > 4 ? atoi(argv[4]) : 8, k = argc > 5 ? atoi(argv[5]) : 100;
    const int D = 128, M = 16, K = 256;
    {   // a model file in the layout IVFOPQ::LoadModel reads (IVFOPQ.cpp:75-95): zero coarse centroid, random codebooks, identity order
        FILE *f = fopen("opq_concurrent_model.bin", "wb");
        if (!f) { printf("cannot write the model file\n"); return 1; }
        const int hdr[4] = { D, 1, M, K };
        fwrite(hdr, sizeof(int), 4, f);
        std::vector<float> coarse(D, 0.0f), books((size_t)M * K * (D / M));
        for (size_t i = 0; i < books.size(); ++i) books[i] = rnd();
        std::vector<int> order(D);
        for (int i = 0; i < D; ++i) order[i] = i;
        fwrite(coarse.data(), sizeof(float), coarse.size(), f);
        fwrite(books.data(), sizeof(float), books.size(), f);
        fwrite(order.data(), sizeof(int), order.size(), f);
        fclose(f);
    }
    IVFOPQ index((int)n + 16);
    if (index.LoadModel("opq_concurrent_model.bin") != 1) { printf("LoadModel failed\n"); return 1; }
    {
        const int chunk = 65536;
        std::vector<float> rows((size_t)chunk * D);
        for (long long a = 0; a < n; a += chunk) {
            const int m = (int)std::min<long long>(chunk, n - a);
            for (size_t i = 0; i < (size_t)m * D; ++i) rows[i] = rnd();
            if (index.AddRows(rows.data(), m) != 1) { printf("AddRows failed: %s\n", index.lastError().c_str()); return 1; }
        }
    }
    const int total = T * calls;
    std::vector<float> q((size_t)total * nq * D);
    for (size_t i = 0; i < q.size(); ++i) q[i] = rnd();
    std::vector<float> rd((size_t)total * nq * k), gd(rd.size());
    std::vector<long long> ri(rd.size()), gi(rd.size());
    if (index.SearchTopK(q.data(), nq, k, rd.data(), ri.data()) != 1) { printf("SearchTopK failed: %s\n", index.lastError().c_str()); return 1; }
    typedef std::chrono::steady_clock clk;
    clk::time_point t0 = clk::now();
    for (int c = 0; c < total; ++c)
        if (index.SearchTopK(&q[(size_t)c * nq * D], nq, k, &rd[(size_t)c * nq * k], &ri[(size_t)c * nq * k]) != 1) { printf("SearchTopK failed\n"); return 1; }
    const double t_one = std::chrono::duration<double>(clk::now() - t0).count();
    int failed = 0;
    t0 = clk::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            for (int c = t * calls; c < (t + 1) * calls; ++c)
                if (index.SearchTopK(&q[(size_t)c * nq * D], nq, k, &gd[(size_t)c * nq * k], &gi[(size_t)c * nq * k]) != 1) failed = 1;
        });
    for (auto &x : th) x.join();
    const double t_par = std::chrono::duration<double>(clk::now() - t0).count();
    if (failed) { printf("SearchTopK failed in a thread: %s\n", index.lastError().c_str()); return 1; }
    if (memcmp(rd.data(), gd.data(), rd.size() * sizeof(float)) != 0 || ri != gi) { printf("MISMATCH between the one-thread and the %d-thread results\n", T); return 1; }
    printf("rows %lld, %d queries per call, k %d: 1 thread %.1f calls/s, %d threads %.1f calls/s (x%.2f) OK\n", n, nq, k, total / t_one, T, total / t_par,
           t_one / t_par);
    return 0;
}
