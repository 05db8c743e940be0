// int8_quan.cc -- see int8_quan.h.  Reference: scalar_quantization/scalar_quantization/int8_quan.cc.
#include "int8_quan.h"

#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/cvtmi.h"

namespace cvtk {
namespace quant {

// The faiss 1.5.3 container of an IndexScalarQuantizer (index_io: write_index -> "IxSQ"), little endian, no padding:
//   uint32 fourcc "IxSQ"
//   index header: int32 d; int64 ntotal; int64 dummy (1 << 20) x 2; uint8 is_trained; int32 metric_type (0 IP, 1 L2)
//   scalar quantiser: int32 qtype (0 = QT_8bit); int32 rangestat (0 = RS_minmax); float rangestat_arg;
//                     uint64 d; uint64 code_size; uint64 n_trained; float trained[n_trained]   (QT_8bit: vmin[d] | vdiff[d])
//   uint64 n_codes; uint8 codes[n_codes]                                                       (ntotal x code_size; ignored)
// Call sites in the reference: faiss::read_index (int8_quan.cc:14, :35-36), sq.trained / sq.code_size (:59-60, :81-83,
// :126-129), faiss::write_index (sq_train.cpp:103).
namespace {
const uint32_t kIxSQ = 'I' | ('x' << 8) | ('S' << 16) | ((uint32_t)'Q' << 24);
template <class T> bool rd(std::istream &f, T &v) { f.read((char *)&v, sizeof v); return (bool)f; }
template <class T> void wr(std::ostream &f, const T &v) { f.write((const char *)&v, sizeof v); }
}

bool read_ixsq_model(const std::string &path, Sq8Model &m, std::string *why)
{
    auto fail = [&](const char *w) { if (why) *why = w; return false; };
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f.good()) return fail("cannot open");
    uint32_t h = 0; int32_t d = 0, metric = 0, qtype = 0, rangestat = 0; int64_t ntotal = 0, dummy = 0; uint8_t trained_flag = 0;
    float rs_arg = 0; uint64_t sq_d = 0, code_size = 0, n_tr = 0;
    if (!rd(f, h) || h != kIxSQ) return fail("not an IxSQ container");
    if (!rd(f, d) || !rd(f, ntotal) || !rd(f, dummy) || !rd(f, dummy) || !rd(f, trained_flag) || !rd(f, metric)) return fail("truncated index header");
    if (!rd(f, qtype) || !rd(f, rangestat) || !rd(f, rs_arg) || !rd(f, sq_d) || !rd(f, code_size) || !rd(f, n_tr)) return fail("truncated quantiser header");
    if (d <= 0 || d > (1 << 20) || sq_d != (uint64_t)d) return fail("bad dimension");
    if (qtype != 0) return fail("quantiser type is not QT_8bit (the only one Int8Quan's in-tree formulas cover, int8_quan.cc:79-92)");
    if (code_size != (uint64_t)d || n_tr != 2 * (uint64_t)d) return fail("code_size / trained size do not fit QT_8bit");
    if (!trained_flag) return fail("index is not trained");
    m.d = d; m.vmin.resize(d); m.vdiff.resize(d);
    f.read((char *)m.vmin.data(), sizeof(float) * d);
    f.read((char *)m.vdiff.data(), sizeof(float) * d);
    if (!f) return fail("truncated trained vector");
    return true;
}

bool write_ixsq_model(const std::string &path, const Sq8Model &m)
{
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f.good()) return false;
    const int64_t dummy = 1 << 20;
    wr(f, kIxSQ); wr(f, (int32_t)m.d); wr(f, (int64_t)0); wr(f, dummy); wr(f, dummy); wr(f, (uint8_t)1); wr(f, (int32_t)1 /* METRIC_L2, sq_train.cpp:100 */);
    wr(f, (int32_t)0); wr(f, (int32_t)0); wr(f, 0.0f); wr(f, (uint64_t)m.d); wr(f, (uint64_t)m.d); wr(f, (uint64_t)(2 * m.d));
    f.write((const char *)m.vmin.data(), sizeof(float) * m.d);
    f.write((const char *)m.vdiff.data(), sizeof(float) * m.d);
    wr(f, (uint64_t)0);
    return (bool)f;
}

bool read_sq8_model(const std::string &path, Sq8Model &m)
{
    std::ifstream fin(path.c_str(), std::ios::binary);
    if (!fin.good()) return false;
    uint32_t h = 0;
    fin.read((char *)&h, sizeof h);
    if (fin && h == kIxSQ) {
        std::string why;
        if (read_ixsq_model(path, m, &why)) return true;
        std::cout << "faiss IxSQ model " << path << ": " << why << std::endl;
        return false;
    }
    fin.clear(); fin.seekg(0);
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
