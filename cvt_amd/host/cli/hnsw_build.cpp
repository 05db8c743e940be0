// hnsw_build -- build and save a HierarchicalNSW graph the way makeIdx.cpp:325-396 does: one addPoint per row,
// in row order, then saveIndex.  The file is byte-identical to the one the reference writes for the same rows.
//   hnsw_build <rows.bin> <dim> <M> <efConstruction> <out index> [ip|l2] [labels.bin|-] [threads]
// rows.bin: raw fp32 [n][dim]; labels.bin: raw uint64 [n] (default / "-": the row number, makeIdx.cpp:364).
// threads (absent: one addPoint per row): given = addPoints, the reference's locked parallel insertion (hnswalg.h:594-608; 0 = all hardware
// threads) -- ids and levels stay those of the row order, the links depend on the interleaving.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "../hnswlib/hnswlib.h"

template <typename T> static bool slurp(const char *path, std::vector<T> &v)
{
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) return false;
    const size_t bytes = (size_t)in.tellg();
    in.seekg(0);
    v.resize(bytes / sizeof(T));
    in.read((char *)v.data(), (std::streamsize)(v.size() * sizeof(T)));
    return true;
}

int main(int argc, char *argv[])
{
    if (argc < 6) {
        std::cout << "usage: hnsw_build <rows.bin> <dim> <M> <efConstruction> <out index> [ip|l2] [labels.bin|-] [threads]\n";
        return -1;
    }
    const size_t dim = (size_t)atoi(argv[2]), M = (size_t)atoi(argv[3]), efc = (size_t)atoi(argv[4]);
    const bool l2 = argc > 6 && !strcmp(argv[6], "l2");
    std::vector<float> rows;
    std::vector<uint64_t> labels;
    if (!slurp(argv[1], rows)) { std::cout << "cannot open " << argv[1] << "\n"; return 1; }
    const int threads = argc > 8 ? atoi(argv[8]) : -1;   // -1: the addPoint loop
    if (argc > 7 && strcmp(argv[7], "-") && !slurp(argv[7], labels)) { std::cout << "cannot open " << argv[7] << "\n"; return 1; }
    const size_t n = rows.size() / dim;
    if (!labels.empty() && labels.size() != n) { std::cout << "labels.bin does not hold one label per row\n"; return 1; }
    try {
        hnswlib::InnerProductSpace ip(dim);
        hnswlib::L2Space l2s(dim);
        hnswlib::SpaceInterface<float> *space = l2 ? (hnswlib::SpaceInterface<float> *)&l2s : (hnswlib::SpaceInterface<float> *)&ip;
        const auto t0 = std::chrono::steady_clock::now();
        hnswlib::HierarchicalNSW<float> alg(space, n, M, efc);
        if (threads < 0) {
            for (size_t i = 0; i < n; ++i) alg.addPoint(&rows[i * dim], labels.empty() ? (hnswlib::labeltype)i : (hnswlib::labeltype)labels[i]);
        } else {
            std::vector<hnswlib::labeltype> lab(labels.begin(), labels.end());
            alg.addPoints(rows.data(), lab.empty() ? NULL : lab.data(), n, (unsigned)threads);
        }
        const auto t1 = std::chrono::steady_clock::now();
        alg.saveIndex(argv[5]);
        const auto t2 = std::chrono::steady_clock::now();
        std::cout << n << " rows indexed (insert " << std::chrono::duration<double>(t1 - t0).count() << " s, save "
                  << std::chrono::duration<double>(t2 - t1).count() << " s)" << std::endl;
    } catch (const std::exception &e) {
        std::cout << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
