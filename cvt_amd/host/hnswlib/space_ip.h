// InnerProductSpace (brute_force_search/src/space_ip.hpp:211-239): distance = 1 - <q, x>, fp32.
#pragma once
namespace hnswlib {
class InnerProductSpace : public SpaceInterface<float> {
    size_t data_size_, dim_;
public:
    InnerProductSpace(size_t dim) : data_size_(dim * sizeof(float)), dim_(dim) {}
    size_t get_data_size() { return data_size_; }
    DISTFUNC<float> get_dist_func() { return host_dist_ip; }
    void *get_dist_func_param() { return &dim_; }
    int device_metric() { return 0; /* CVTMI_METRIC_IP */ }
};
}  // namespace hnswlib
