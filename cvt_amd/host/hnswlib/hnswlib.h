// hnswlib.h -- the plugin interfaces of the reference (brute_force_search/src/hnswlib.hpp:22-58), unchanged in
// shape: SpaceInterface<MTYPE>, DISTFUNC<MTYPE>, AlgorithmInterface<dist_t>, labeltype.  A space additionally
// says which device metric it stands for; searches evaluate distances in the HIP kernels, never in a host loop
// (the one host algorithm is HNSW graph construction, hnswalg.h, sequential in the reference as well);
// get_dist_func() hands out a host function for single distances, as the reference's spaces do.
#pragma once
#include <queue>
#include <stdexcept>
#include <string>
#include <string.h>
#include <iostream>

namespace hnswlib {
typedef size_t labeltype;

template <typename T> static void writeBinaryPOD(std::ostream &out, const T &podRef) { out.write((char *)&podRef, sizeof(T)); }
template <typename T> static void readBinaryPOD(std::istream &in, T &podRef) { in.read((char *)&podRef, sizeof(T)); }

template <typename MTYPE> using DISTFUNC = MTYPE (*)(const void *, const void *, const void *);

template <typename MTYPE> class SpaceInterface {
public:
    virtual size_t get_data_size() = 0;
    virtual DISTFUNC<MTYPE> get_dist_func() = 0;
    virtual void *get_dist_func_param() = 0;
    // MI355X build: which cvtmi_metric the space stands for (CVTMI_METRIC_IP / _L2F / _L2U8), -1 = none
    virtual int device_metric() { return -1; }
    virtual ~SpaceInterface() {}
};

template <typename dist_t> class AlgorithmInterface {
public:
    virtual void addPoint(void *datapoint, labeltype label) = 0;
    virtual std::priority_queue<std::pair<dist_t, labeltype> > searchKnn(void *, size_t) = 0;
    virtual void saveIndex(const std::string &location) = 0;
    virtual ~AlgorithmInterface() {}
};

// get_dist_func() of the three built-in spaces: host functions with the summation order of the reference's own builds (the
// order the device kernels reproduce, csrc/dist_f32.h), for callers that evaluate single distances through the plugin
// seam -- recall harnesses like hnsw_sifts_retrieval/makeIdx.cpp:231-285.  Searches never come here: searchKnn runs on the
// MI355X.  Separate multiply and add (the file is built with -ffp-contract=off), so the bits match the device's.
//   inner product  dim % 4 == 0: four lane accumulators, 1 - (((a0 + a1) + a2) + a3)   (space_ip.hpp:84-131, :168-206: the SSE
//                  branches its CMake flags build); otherwise the scalar loop (:25-34)
//   squared L2     dim % 16 == 0: eight lanes summed left to right (space_l2.h:46-73, USE_AVX is hard-defined at :12);
//                  dim % 4 == 0: four lanes (:123-151); otherwise the scalar loop (:26-37)
//   uint8 L2       groups of four bytes, a dim % 4 tail is dropped (:198-215); exact integers
static float host_dist_ip(const void *pa, const void *pb, const void *pd)
{
    const float *a = (const float *)pa, *b = (const float *)pb;
    const size_t d = *(const size_t *)pd;
    if (d % 4 == 0) {
        float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (size_t i = 0; i < d; i += 4)
            for (int l = 0; l < 4; ++l) { const float p = a[i + l] * b[i + l]; acc[l] = acc[l] + p; }
        float s = acc[0] + acc[1];
        s = s + acc[2];
        s = s + acc[3];
        return 1.0f - s;
    }
    float s = 0.0f;
    for (size_t i = 0; i < d; ++i) { const float p = a[i] * b[i]; s = s + p; }
    return 1.0f - s;
}
static float host_dist_l2(const void *pa, const void *pb, const void *pd)
{
    const float *a = (const float *)pa, *b = (const float *)pb;
    const size_t d = *(const size_t *)pd;
    const int lanes = d % 16 == 0 ? 8 : (d % 4 == 0 ? 4 : 1);
    float acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (size_t i = 0; i < d; i += lanes)
        for (int l = 0; l < lanes; ++l) { const float t = a[i + l] - b[i + l]; const float p = t * t; acc[l] = acc[l] + p; }
    float s = acc[0];
    for (int l = 1; l < lanes; ++l) s = s + acc[l];
    return s;
}
static int host_dist_l2u8(const void *pa, const void *pb, const void *pd)
{
    const unsigned char *a = (const unsigned char *)pa, *b = (const unsigned char *)pb;
    const size_t d = *(const size_t *)pd & ~(size_t)3;
    int s = 0;
    for (size_t i = 0; i < d; ++i) { const int t = (int)a[i] - (int)b[i]; s += t * t; }
    return s;
}
}  // namespace hnswlib

#include "space_ip.h"
#include "space_l2.h"
#include "bruteforce.h"
#include "hnswalg.h"
