// int8_quan.cc -- see int8_quan.h.  Reference: scalar_quantization/scalar_quantization/int8_quan.cc.
#include "int8_quan.h"

#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/cvtmi.h"

namespace cvtk {
namespace quant {

bool read_sq8_model(const std::string &path, Sq8Model &m)
{
    std::ifstream fin(path.c_str(), std::ios::binary);
    if (!fin.good()) return false;
    int32_t d = 0;
    fin.read((char *)&d, sizeof d);
    if (!fin || d <= 0 || d > (1 << 20)) return false;
    m.d = d; m.vmin.resize(d); m.vdiff.resize(d);
    fin.read((char *)m.vmin.data(), sizeof(float) * d);
    fin.read((char *)m.vdiff.data(), sizeof(float) * d);
    return (bool)fin;
}

bool write_sq8_model(const std::string &path, const Sq8Model &m)
{
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f.good()) return false;
    int32_t d = m.d;
    f.write((const char *)&d, sizeof d);
    f.write((const char *)m.vmin.data(), sizeof(float) * d);
    f.write((const char *)m.vdiff.data(), sizeof(float) * d);
    return (bool)f;
}

bool train_sq8_model(const float *x, size_t n, int d, bool l2norm, Sq8Model &m)
{
    m.d = d; m.vmin.resize(d); m.vdiff.resize(d);
    if (cvtmi_sq8_train(x, (int64_t)n, d, l2norm ? 1 : 0, m.vmin.data(), m.vdiff.data()) != CVTMI_OK) {
        std::cout << "cvtmi_sq8_train: " << cvtmi_last_error() << std::endl;
        return false;
    }
    return true;
}

Int8Quan::Int8Quan(const std::string &model_path)
{
    Sq8Model m;
    if (!read_sq8_model(model_path, m)) {
        std::cout << "model file is not exists" << std::endl;
        load_model_ok = false;
        return;
    }
    load_model_ok = true;
    models_.push_back(m);
}

// minimal reader for the reference's JSON conf (int8_quan.cc:20-39): {"0": {"model_path": "..."}, "1": {...}}
Int8Quan::Int8Quan(const std::string &model_conf_path, int num_source)
{
    std::ifstream fin(model_conf_path.c_str());
    if (!fin.good()) {
        std::cout << "model file is not exists" << std::endl;
        load_model_ok = false;
        return;
    }
    std::stringstream ss;
    ss << fin.rdbuf();
    const std::string txt = ss.str();
    load_model_ok = true;
    for (int i = 0;; ++i) {
        const std::string key = "\"" + std::to_string(i) + "\"";
        size_t p = txt.find(key);
        if (p == std::string::npos) break;
        p = txt.find("\"model_path\"", p);
        if (p == std::string::npos) break;
        p = txt.find(':', p);
        size_t a = txt.find('"', p), b = txt.find('"', a + 1);
        if (a == std::string::npos || b == std::string::npos) break;
        const std::string path = txt.substr(a + 1, b - a - 1);
        std::cout << "load model: " << path << std::endl;
        Sq8Model m;
        if (!read_sq8_model(path, m)) { load_model_ok = false; return; }
        models_.push_back(m);
    }
    if (models_.empty() || (num_source > 0 && (int)models_.size() < num_source)) load_model_ok = false;
}

Int8Quan::~Int8Quan() {}
bool Int8Quan::status() { return load_model_ok; }

// all n = n_dims / d vectors (what sq.compute_codes does, int8_quan.cc:58-70)
int Int8Quan::Int8EncodeFaiss(float *x, uint8_t *bytes, size_t n_dims, bool turn_off_l2norm, int source)
{
    if (source < 0 || source >= (int)models_.size()) return 0;
    const Sq8Model &m = models_[source];
    if (n_dims % m.d != 0) return 0;
    return cvtmi_sq8_encode(m.vmin.data(), m.vdiff.data(), m.d, x, (int64_t)(n_dims / m.d), turn_off_l2norm ? 0 : 1, bytes) == CVTMI_OK;
}

// only the FIRST vector is normalised and encoded, whatever n_dims says (int8_quan.cc:72-94)
int Int8Quan::Int8Encode(float *x, uint8_t *bytes, size_t n_dims, bool turn_off_l2norm, int source)
{
    if (source < 0 || source >= (int)models_.size()) return 0;
    const Sq8Model &m = models_[source];
    if (n_dims % m.d != 0) return 0;
    return cvtmi_sq8_encode(m.vmin.data(), m.vdiff.data(), m.d, x, 1, turn_off_l2norm ? 0 : 1, bytes) == CVTMI_OK;
}

int Int8Quan::Int8Decode(uint8_t *bytes, float *x, size_t n_dims, int source)
{
    if (source < 0 || source >= (int)models_.size()) return 0;
    const Sq8Model &m = models_[source];
    if (n_dims % m.d != 0) return 0;
    // all n_dims / d vectors, in the arithmetic of faiss' own 8-bit codec (int8_quan.cc:96-104 -> sq.decode): fp32, not the double
    // formula of Int8Decode(std::string&)
    return cvtmi_sq8_decode_faiss(m.vmin.data(), m.vdiff.data(), m.d, bytes, (int64_t)(n_dims / m.d), x) == CVTMI_OK;
}

int Int8Quan::Int8DecodeFaiss(std::string &embedding, float *x, int source)
{
    if (embedding.empty()) return 0;
    return Int8Decode(reinterpret_cast<uint8_t *>(&embedding[0]), x, embedding.size(), source);
}

// first vector only (int8_quan.cc:117-132)
int Int8Quan::Int8Decode(std::string &embedding, float *x, int source)
{
    if (embedding.empty()) return 0;
    if (source < 0 || source >= (int)models_.size()) return 0;
    const Sq8Model &m = models_[source];
    if (embedding.size() % m.d != 0) return 0;
    return cvtmi_sq8_decode(m.vmin.data(), m.vdiff.data(), m.d, reinterpret_cast<uint8_t *>(&embedding[0]), 1, x) == CVTMI_OK;
}

}  // namespace quant
}  // namespace cvtk
