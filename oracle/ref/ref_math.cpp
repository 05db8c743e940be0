// Harness over the reference's utils/math_util.h (cvtk::MathUtil, header-only standard C++), compiled in place
// -> oracle/_ref/libref_math.so.  Test infra only.  MathUtil::L2NormArray (:29-39) is the same arithmetic as
// Int8Quan::L2NormalizeVector (scalar_quantization/scalar_quantization/int8_quan.cc:46-56) and sq_train.cpp's
// L2NomalizeVector -- the one piece of the scalar-quantisation path that can be run here without faiss.
#include <cstdint>
#include <cstring>
#include <vector>
#include "math_util.h"

// rows [n][d] normalised IN PLACE, row by row, by the reference's own function
extern "C" __attribute__((visibility("default")))
int ref_l2norm_array(float *x, int64_t n, int d)
{
    for (int64_t r = 0; r < n; ++r) cvtk::MathUtil::L2NormArray(x + r * (int64_t)d, d);
    return 0;
}

// the std::vector twin (:18-27): norm rounded to float BEFORE the clamp, out-of-place
extern "C" __attribute__((visibility("default")))
int ref_l2norm_vec(const float *x, int64_t n, int d, float *out)
{
    for (int64_t r = 0; r < n; ++r) {
        std::vector<float> v(x + r * (int64_t)d, x + (r + 1) * (int64_t)d);
        std::vector<float> o = cvtk::MathUtil::L2NormVec(v);
        if ((int)o.size() != d) return 1;
        memcpy(out + r * (int64_t)d, o.data(), sizeof(float) * (size_t)d);
    }
    return 0;
}
