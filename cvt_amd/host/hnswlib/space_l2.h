// L2Space / L2SpaceI (hnsw_sifts_retrieval/hnswlib/space_l2.h:153-184, :221-245): squared L2 over fp32 rows,
// and over uint8 rows with an int distance (groups of four bytes, a dim % 4 tail is dropped as in :198).
#pragma once
namespace hnswlib {
class L2Space : public SpaceInterface<float> {
    size_t data_size_, dim_;
public:
    L2Space(size_t dim) : data_size_(dim * sizeof(float)), dim_(dim) {}
    size_t get_data_size() { return data_size_; }
    DISTFUNC<float> get_dist_func() { return host_dist_l2; }
    void *get_dist_func_param() { return &dim_; }
    int device_metric() { return 1; /* CVTMI_METRIC_L2F */ }
};
class L2SpaceI : public SpaceInterface<int> {
    size_t data_size_, dim_;
public:
    L2SpaceI(size_t dim) : data_size_(dim * sizeof(unsigned char)), dim_(dim) {}
    size_t get_data_size() { return data_size_; }
    DISTFUNC<int> get_dist_func() { return host_dist_l2u8; }
    void *get_dist_func_param() { return &dim_; }
    int device_metric() { return 2; /* CVTMI_METRIC_L2U8 */ }
};
}  // namespace hnswlib
