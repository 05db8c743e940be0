// hnswlib.h -- the plugin interfaces of the reference (brute_force_search/src/hnswlib.hpp:22-58), unchanged in
// shape: SpaceInterface<MTYPE>, DISTFUNC<MTYPE>, AlgorithmInterface<dist_t>, labeltype.  A space additionally
// says which device metric it stands for; searches evaluate distances in the HIP kernels, never in a host loop
// (the one host algorithm is HNSW graph construction, hnswalg.h, sequential in the reference as well).
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

// The reference hands out CPU distance functions through get_dist_func(); this build has no CPU path, so
// the function a space returns only reports that fact if something calls it.
template <typename MTYPE> static MTYPE device_only_dist(const void *, const void *, const void *)
{
    throw std::runtime_error("cvt_amd: distances are evaluated on the MI355X (BruteforceSearch::searchKnn); "
                             "there is no host distance function");
}
}  // namespace hnswlib

#include "space_ip.h"
#include "space_l2.h"
#include "bruteforce.h"
#include "hnswalg.h"
