// int8_quan.h -- mirror of cvtk::quant::Int8Quan (scalar_quantization/scalar_quantization/int8_quan.h:17-37)
// above the C ABI.  Same methods, same 1 / 0 return convention, same in-place normalisation of the caller's
// buffer.  faiss is not a dependency: the model is the two arrays faiss keeps in sq.trained.  Two files are read:
//   * the faiss "IxSQ" container itself -- what the reference's trainer writes (sq_train.cpp:103 write_index of an
//     IndexScalarQuantizer(d, QT_8bit)) and its Int8Quan(model_path) loads (int8_quan.cc:14 faiss::read_index): a user's
//     existing model file drops in.  The layout is faiss 1.5.3's published index_io one (read_ixsq_model below); only what
//     Int8Quan uses (d, qtype, trained = vmin | vdiff) is kept, stored codes are skipped.
//   * this library's own plain form   int32 d; float vmin[d]; float vdiff[d]   (write_sq8_model).
// read_sq8_model tells them apart by the fourcc.  write_ixsq_model writes an empty (ntotal = 0) IxSQ container, the file
// sq_train.cpp:103 produces (`sq_train ... --faiss`).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace cvtk {
namespace quant {
struct Sq8Model {
    int d = 0;
    std::vector<float> vmin, vdiff;
};
bool read_sq8_model(const std::string &path, Sq8Model &m);      // either format
bool write_sq8_model(const std::string &path, const Sq8Model &m);
bool read_ixsq_model(const std::string &path, Sq8Model &m, std::string *why = nullptr);   // faiss IxSQ container only
bool write_ixsq_model(const std::string &path, const Sq8Model &m);
// sq_train.cpp:84-103 on the GPU: per-dimension min / max-min over L2-normalised rows
bool train_sq8_model(const float *x, size_t n, int d, bool l2norm, Sq8Model &m);

class Int8Quan {
public:
    explicit Int8Quan(const std::string &model_path);                       // single model
    explicit Int8Quan(const std::string &model_conf_path, int num_source);  // {"0":{"model_path":"..."}, "1":{...}}
    ~Int8Quan();

    // n_dims = number of floats in x; a multiple of the model dimension
    int Int8EncodeFaiss(float *x, uint8_t *bytes, size_t n_dims, bool turn_off_l2norm = false, int source = 0);
    int Int8Encode(float *x, uint8_t *bytes, size_t n_dims, bool turn_off_l2norm = false, int source = 0);

    int Int8DecodeFaiss(std::string &embeddding, float *x, int source = 0);
    int Int8Decode(uint8_t *bytes, float *x, size_t n_dims, int source = 0);
    int Int8Decode(std::string &embeddding, float *x, int source = 0);

    bool status();

private:
    bool load_model_ok = true;
    std::vector<Sq8Model> models_;
};
}  // namespace quant
}  // namespace cvtk
