// int8_quan_demo -- scalar_quantization/scalar_quantization/int8_quan_test.cpp:10-65: encode / decode one vector.
//   int8_quan_demo <model.bin> [raw fp32 vector file]      (default input: the 64-d vector of the reference demo)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>
#include "../int8_quan.h"
int main(int argc, char *argv[])
{
    if (argc < 2) { std::cerr << "usage: int8_quan_demo <model.bin> [vector.f32]" << std::endl; return 2; }
    cvtk::quant::Int8Quan q(argv[1]);
    if (!q.status()) return 1;
    std::vector<float> x = { 0.7678224, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 2.6331244, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.583638, 0.76271933, 0.0, 0.0,
        0.0, 0.0, 0.0, 0.0, 0.0, 0.21529453, 0.0, 0.0, 1.2015152, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.88310665, 0.0, 0.0,
        0.19277531, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 2.5779805, 0.0, 0.0, 0.7728174, 0.0, 2.21898, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    if (argc > 2) {
        std::ifstream f(argv[2], std::ios::binary);
        f.seekg(0, std::ios::end); size_t n = (size_t)f.tellg() / 4; f.seekg(0);
        x.resize(n); f.read((char *)x.data(), n * 4);
    }
    std::vector<uint8_t> bytes(x.size());
    std::vector<float> dec(x.size());
    if (!q.Int8Encode(x.data(), bytes.data(), x.size(), false, 0)) { std::cerr << "encode failed" << std::endl; return 1; }
    std::string s(bytes.begin(), bytes.end());
    if (!q.Int8Decode(s, dec.data())) { std::cerr << "decode failed" << std::endl; return 1; }
    float ip = 0;
    for (size_t i = 0; i < x.size(); ++i) ip += x[i] * dec[i];
    std::cout << "normalised: "; for (float v : x) std::cout << v << " ";
    std::cout << std::endl << "int8: "; for (uint8_t b : bytes) std::cout << unsigned(b) << " ";
    std::cout << std::endl << "decoded: "; for (float v : dec) std::cout << v << " ";
    std::cout << std::endl << "inner_product: " << ip << std::endl;
    // the other decode entry: Int8Decode(uint8_t*) = faiss' own fp32 codec in the reference (int8_quan.cc:96-104); bit patterns, so a
    // test can compare them exactly (they differ from `decoded` by one ulp in about a third of the places)
    std::vector<float> dec2(x.size());
    if (!q.Int8Decode(bytes.data(), dec2.data(), x.size(), 0)) { std::cerr << "faiss-path decode failed" << std::endl; return 1; }
    std::cout << "decoded_faiss_bits:";
    for (float v : dec2) { unsigned u; memcpy(&u, &v, 4); char b[16]; snprintf(b, sizeof b, " %08x", u); std::cout << b; }
    std::cout << std::endl;
    return 0;
}
