// sq_train -- scalar_quantization/train/src/sq_train.cpp:40-103 above the C ABI:
//   sq_train <feats.bin> <model.bin> [dim=64] [--faiss]  (feats.bin: int32 count; per record int32 idLen, id, int32 dim, fp32[dim])
// Rows are L2-normalised, per-dimension min / max-min are computed on the GPU, the model is written as
// int32 d; float vmin[d]; float vdiff[d] -- or, with --faiss, as the empty faiss "IxSQ" container sq_train.cpp:103 writes.
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>
#include "../int8_quan.h"
int main(int argc, char *argv[])
{
    bool faiss_form = false;
    if (argc > 3 && !strcmp(argv[argc - 1], "--faiss")) { faiss_form = true; --argc; }
    if (argc < 3) { std::cerr << "usage: sq_train <feats.bin> <model.bin> [dim] [--faiss]" << std::endl; return 2; }
    size_t d = argc > 3 ? (size_t)atoi(argv[3]) : 64;
    std::ifstream fp(argv[1], std::ios::in | std::ios::binary);
    if (!fp.is_open()) { std::cerr << "cannot open " << argv[1] << std::endl; return 1; }
    int num_db = 0;
    fp.read((char *)&num_db, sizeof(int));
    std::cout << "db num: " << num_db << std::endl;
    std::vector<float> xb((size_t)num_db * d);
    for (size_t i = 0; i < (size_t)num_db; ++i) {
        int idSize = 0;
        char idName[1024] = { "" };
        fp.read((char *)&idSize, sizeof(int));
        if (idSize < 0 || idSize >= 1024) { std::cerr << "bad record" << std::endl; return 1; }
        fp.read(idName, idSize);
        int dim_feat = 0;
        fp.read((char *)&dim_feat, sizeof(int));
        if ((size_t)dim_feat != d) { std::cout << "file error: " << dim_feat << std::endl; return 1; }
        fp.read(reinterpret_cast<char *>(&xb[i * d]), dim_feat * sizeof(float));
    }
    cvtk::quant::Sq8Model m;
    if (!cvtk::quant::train_sq8_model(xb.data(), (size_t)num_db, (int)d, true, m)) return 1;
    if (!(faiss_form ? cvtk::quant::write_ixsq_model(argv[2], m) : cvtk::quant::write_sq8_model(argv[2], m))) { std::cerr << "cannot write " << argv[2] << std::endl; return 1; }
    std::cout << "vmin (from model data): " << std::endl;
    for (size_t i = 0; i < d; ++i) std::cout << m.vmin[i] << " ";
    std::cout << std::endl << "vdiff (from model data): " << std::endl;
    for (size_t i = 0; i < d; ++i) std::cout << m.vdiff[i] << " ";
    std::cout << std::endl;
    return 0;
}
