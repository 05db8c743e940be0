// pca_utils.cc -- see pca_utils.h.  Reference: pca_train_project/pca_online/pca_utils.cc:16-35.
#include "pca_utils.h"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "../../include/cvtmi.h"

namespace cvtk {

static size_t find_key(const std::string &t, const std::string &key, size_t from, size_t until)
{
    const size_t p = t.find(key, from);
    if (p == std::string::npos || p >= until) throw std::runtime_error("PCAUtils: malformed matrix node (no '" + key + "')");
    return p + key.size();
}

Mat32f read_opencv_matrix(const std::string &t, const std::string &name)
{
    // top-level key at the start of a line
    size_t p = 0;
    const std::string key = name + ":";
    for (;;) {
        p = t.find(key, p);
        if (p == std::string::npos) throw std::runtime_error("PCAUtils: no '" + name + "' node in the model file");
        if (p == 0 || t[p - 1] == '\n') break;
        p += key.size();
    }
    const size_t close = t.find(']', p);
    if (close == std::string::npos) throw std::runtime_error("PCAUtils: unterminated data of '" + name + "'");
    const long rows = strtol(t.c_str() + find_key(t, "rows:", p, close), NULL, 10);
    const long cols = strtol(t.c_str() + find_key(t, "cols:", p, close), NULL, 10);
    size_t dt = find_key(t, "dt:", p, close);
    while (t[dt] == ' ') ++dt;
    if (t[dt] != 'f' && t[dt] != 'd') throw std::runtime_error("PCAUtils: '" + name + "' is not a float matrix");
    if (rows < 1 || cols < 1 || rows * cols > (1L << 28)) throw std::runtime_error("PCAUtils: bad shape of '" + name + "'");
    Mat32f m;
    m.create((int)rows, (int)cols);
    const char *c = t.c_str() + find_key(t, "[", find_key(t, "data:", p, close), close + 1);
    const char *end = t.c_str() + close;
    size_t i = 0;
    while (c < end) {
        while (c < end && (*c == ' ' || *c == ',' || *c == '\n' || *c == '\r' || *c == '\t')) ++c;
        if (c >= end) break;
        char *next = NULL;
        const double v = strtod(c, &next);  // FileStorage parses decimal text to double, then narrows
        if (next == c) throw std::runtime_error("PCAUtils: bad number in '" + name + "'");
        if (i >= m.data.size()) throw std::runtime_error("PCAUtils: too many values in '" + name + "'");
        m.data[i++] = (float)v;
        c = next;
    }
    if (i != m.data.size()) throw std::runtime_error("PCAUtils: too few values in '" + name + "'");
    return m;
}

void PCAUtils::loadModel(const std::string &filename)
{
    std::ifstream in(filename, std::ios::binary);
    if (!in) throw std::runtime_error("PCAUtils: cannot open " + filename);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    eigenvectors = read_opencv_matrix(text, "vectors");
    eigenvalues = read_opencv_matrix(text, "values");
    mean = read_opencv_matrix(text, "mean");
    if (mean.rows != 1 || mean.cols != eigenvectors.cols)
        throw std::runtime_error("PCAUtils: mean must be 1 x " + std::to_string(eigenvectors.cols));
}

void PCAUtils::reduceDim(const float *data, int num, int dim, Mat32f &reduceMat)
{
    if (eigenvectors.empty()) throw std::runtime_error("PCAUtils: no model loaded");
    if (dim != eigenvectors.cols) throw std::runtime_error("PCAUtils: the model projects " + std::to_string(eigenvectors.cols) + "-d rows");
    reduceMat.create(num, eigenvectors.rows);
    if (cvtmi_pca_project(mean.data.data(), eigenvectors.data.data(), dim, eigenvectors.rows, data, num, 1, reduceMat.data.data()) != CVTMI_OK)
        throw std::runtime_error(std::string("cvt_amd: ") + cvtmi_last_error());
}

void PCAUtils::reduceDim(const Mat32f &mat, Mat32f &reduceMat) { reduceDim(mat.data.data(), mat.rows, mat.cols, reduceMat); }

Mat32f PCAUtils::reduceDim(const Mat32f &mat)
{
    Mat32f out;
    reduceDim(mat, out);
    return out;
}

}  // namespace cvtk
