// brute_force -- the ground-truth CLI of the reference (brute_force_search/src/brute_force.cpp:12-112):
// db.bin / querys.bin (int32 count; per record: int32 idLen, id bytes, int32 dim, fp32[dim]) -> index.bin, gt.txt
//   brute_force [db.bin] [querys.bin] [index.bin] [gt.txt] [dim=128] [topK=100]
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>
#include "../hnswlib/hnswlib.h"
using namespace std;
using namespace hnswlib;

static bool read_record(std::ifstream &fp, int ddim, std::string &id, float *feat)
{
    int idSize = 0;
    fp.read((char *)&idSize, sizeof(int));
    if (!fp || idSize < 0 || idSize >= 1024) return false;
    char idName[1024] = { "" };
    fp.read(idName, idSize);
    id = std::string(idName);
    int dim_feat = 0;
    fp.read((char *)&dim_feat, sizeof(int));
    if (dim_feat != ddim) {
        cout << "file error";
        exit(1);
    }
    fp.read(reinterpret_cast<char *>(feat), dim_feat * sizeof(float));
    return (bool)fp;
}

int main(int argc, const char *argv[])
{
    std::string db_path = argc > 1 ? argv[1] : "db.bin";
    std::string querys_path = argc > 2 ? argv[2] : "querys.bin";
    std::string index_path = argc > 3 ? argv[3] : "index.bin";
    std::string gtfile_path = argc > 4 ? argv[4] : "gt.txt";
    int ddim = argc > 5 ? atoi(argv[5]) : 128;
    int topK = argc > 6 ? atoi(argv[6]) : 100;

    ofstream gtFile(gtfile_path.c_str());
    std::ifstream fp(db_path.c_str(), std::ios::in | std::ios::binary);
    if (!fp.is_open()) { cerr << "cannot open " << db_path << endl; return 1; }
    int num_db = 0;
    fp.read((char *)&num_db, sizeof(int));
    InnerProductSpace ipspace(ddim);
    BruteforceSearch<float> *bruteAlg = new BruteforceSearch<float>(&ipspace, num_db);
    std::vector<float> feat(ddim);
    std::vector<std::string> dbIds;
    for (int i = 0; i < num_db; i++) {
        std::string idStr;
        if (!read_record(fp, ddim, idStr, feat.data())) { cerr << "db truncated" << endl; return 1; }
        bruteAlg->addPoint((void *)feat.data(), (size_t)i);
        dbIds.push_back(idStr);
        if ((i + 1) % 100000 == 0) printf("indexed %d points\n", (i + 1));
    }
    fp.close();
    bruteAlg->saveIndex(index_path);
    printf("brute force index finished building\n");

    fp.open(querys_path.c_str(), std::ios::in | std::ios::binary);
    if (!fp.is_open()) { cerr << "cannot open " << querys_path << endl; return 1; }
    int num_querys = 0;
    fp.read((char *)&num_querys, sizeof(int));
    // all queries in one device batch (the reference loops searchKnn per query, brute_force.cpp:86)
    std::vector<float> qs((size_t)num_querys * ddim);
    std::vector<std::string> qIds(num_querys);
    for (int i = 0; i < num_querys; i++)
        if (!read_record(fp, ddim, qIds[i], &qs[(size_t)i * ddim])) { cerr << "queries truncated" << endl; return 1; }
    std::vector<float> d((size_t)num_querys * topK);
    std::vector<int64_t> l((size_t)num_querys * topK);
    if (num_querys) bruteAlg->searchKnnBatch(qs.data(), num_querys, topK, d.data(), l.data());
    for (int i = 0; i < num_querys; i++) {
        std::stringstream ids, dists;
        for (int j = 0; j < topK; ++j) {
            if (l[(size_t)i * topK + j] < 0) break;
            ids << dbIds[l[(size_t)i * topK + j]].c_str() << " ";
            dists << (float)(1.0 - d[(size_t)i * topK + j]) << " ";  // the CLI reports the inner product (brute_force.cpp:92)
        }
        gtFile << qIds[i].c_str() << " " << "topK: " << ids.str().c_str() << "dists: " << dists.str().c_str() << "\n";
        if ((i + 1) % 1000 == 0) printf("searched %d points\n", (i + 1));
    }
    gtFile.close();
    delete bruteAlg;
    return 0;
}
