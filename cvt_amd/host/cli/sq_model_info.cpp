// sq_model_info -- what Int8Quan(model_path) sees in a model file (either the faiss "IxSQ" container of sq_train.cpp:103 /
// int8_quan.cc:14, or this library's plain form): the dimension and the bit patterns of vmin / vdiff.  Host only, no device.
//   sq_model_info <model.bin>
#include <cstdio>
#include <cstring>
#include <iostream>
#include "../int8_quan.h"
int main(int argc, char *argv[])
{
    if (argc < 2) { std::cerr << "usage: sq_model_info <model.bin>" << std::endl; return 2; }
    cvtk::quant::Int8Quan q(argv[1]);
    if (!q.status()) return 1;
    cvtk::quant::Sq8Model m;
    if (!cvtk::quant::read_sq8_model(argv[1], m)) return 1;
    std::string why;
    cvtk::quant::Sq8Model probe;
    std::cout << "format: " << (cvtk::quant::read_ixsq_model(argv[1], probe, &why) ? "faiss IxSQ" : "plain") << std::endl;
    std::cout << "d: " << m.d << std::endl << "vmin_bits:";
    for (float v : m.vmin) { unsigned u; memcpy(&u, &v, 4); char b[16]; snprintf(b, sizeof b, " %08x", u); std::cout << b; }
    std::cout << std::endl << "vdiff_bits:";
    for (float v : m.vdiff) { unsigned u; memcpy(&u, &v, 4); char b[16]; snprintf(b, sizeof b, " %08x", u); std::cout << b; }
    std::cout << std::endl;
    return 0;
}
