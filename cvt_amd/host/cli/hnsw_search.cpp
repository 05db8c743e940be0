// hnsw_search -- batched search over a graph saved by the reference's HierarchicalNSW::saveIndex.
//   hnsw_search <index file> <queries.bin> <dim> <k> <ef> <out.txt> [ip|l2]
// queries.bin: raw fp32 [nq][dim].  out.txt: one line per query, "label:dist" pairs in ascending (dist, label)
// order -- what makeSearch.cpp:49-60 reads off searchKnn, for a whole batch at once.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "../hnswlib/hnswlib.h"

int main(int argc, char *argv[])
{
    if (argc < 7) {
        std::cout << "usage: hnsw_search <index> <queries.bin> <dim> <k> <ef> <out.txt> [ip|l2]\n";
        return -1;
    }
    const int dim = atoi(argv[3]), k = atoi(argv[4]), ef = atoi(argv[5]);
    const bool l2 = argc > 7 && !strcmp(argv[7], "l2");
    std::ifstream in(argv[2], std::ios::binary | std::ios::ate);
    if (!in) { std::cout << "cannot open " << argv[2] << "\n"; return 1; }
    const size_t bytes = (size_t)in.tellg();
    in.seekg(0);
    const size_t nq = bytes / (sizeof(float) * (size_t)dim);
    std::vector<float> q(nq * (size_t)dim);
    in.read((char *)q.data(), (std::streamsize)(nq * dim * sizeof(float)));
    try {
        hnswlib::InnerProductSpace ip((size_t)dim);
        hnswlib::L2Space l2s((size_t)dim);
        hnswlib::SpaceInterface<float> *space = l2 ? (hnswlib::SpaceInterface<float> *)&l2s : (hnswlib::SpaceInterface<float> *)&ip;
        hnswlib::HierarchicalNSW<float> alg(space, std::string(argv[1]));
        alg.setEf((size_t)ef);
        std::vector<std::priority_queue<std::pair<float, hnswlib::labeltype> > > res = alg.searchKnnBatch(q.data(), nq, (size_t)k);
        FILE *f = fopen(argv[6], "w");
        if (!f) { std::cout << "cannot write " << argv[6] << "\n"; return 1; }
        for (size_t i = 0; i < nq; ++i) {
            std::vector<std::pair<float, hnswlib::labeltype> > v;
            while (!res[i].empty()) { v.push_back(res[i].top()); res[i].pop(); }
            for (size_t j = v.size(); j-- > 0;) fprintf(f, "%zu:%.9g ", (size_t)v[j].second, (double)v[j].first);
            fprintf(f, "\n");
        }
        fclose(f);
        std::cout << nq << " queries over " << alg.ntotal() << " nodes" << std::endl;
    } catch (const std::exception &e) {
        std::cout << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
