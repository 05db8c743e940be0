// HierarchicalNSW<float> -- search-side mirror of hnsw_sifts_retrieval/hnswlib/hnswalg.h above the C ABI.
// Same constructor for a saved graph (space, location), loadIndex, setEf, searchKnn with the reference's
// return type; the traversal runs on the MI355X (cvtmi_hnsw_search) and returns the reference's labels and
// distances bit for bit.  searchKnnBatch is the call that fills the GPU: one wave per query.
// Graph CONSTRUCTION (addPoint on an empty index, saveIndex) is not offered in this build: graphs are
// built and saved with the reference's tools; the members say so loudly instead of falling back to a host
// implementation.
#pragma once
#include <fstream>
#include <vector>

#include "../../../include/cvtmi.h"

namespace hnswlib {
template <typename dist_t> class HierarchicalNSW : public AlgorithmInterface<dist_t> {
public:
    HierarchicalNSW(SpaceInterface<dist_t> *s) : h_(NULL), ef_(10), dim_(0) { (void)s; }
    HierarchicalNSW(SpaceInterface<dist_t> *s, const std::string &location, bool nmslib = false) : h_(NULL), ef_(10), dim_(0)
    {
        (void)nmslib;
        loadIndex(location, s);
    }
    HierarchicalNSW(SpaceInterface<dist_t> *, size_t, size_t = 16, size_t = 200) : h_(NULL), ef_(10), dim_(0)
    {
        throw std::runtime_error("cvt_amd: HierarchicalNSW graph construction is not offered on the MI355X build; "
                                 "build and save the graph with the reference's tools, then load it here");
    }
    ~HierarchicalNSW()
    {
        if (h_) cvtmi_hnsw_destroy(h_);
    }

    void setEf(size_t ef) { ef_ = ef; }

    void loadIndex(const std::string &location, SpaceInterface<dist_t> *s)
    {
        std::ifstream in(location, std::ios::binary | std::ios::ate);
        if (!in) throw std::runtime_error("Cannot open file " + location);
        const std::streamsize bytes = in.tellg();
        in.seekg(0);
        std::vector<char> buf((size_t)bytes);
        in.read(buf.data(), bytes);
        dim_ = s->get_data_size() / sizeof(float);
        if (h_) { cvtmi_hnsw_destroy(h_); h_ = NULL; }
        if (cvtmi_hnsw_load(buf.data(), (int64_t)bytes, s->device_metric(), (int)dim_, &h_) != CVTMI_OK)
            throw std::runtime_error(std::string("cvt_amd: ") + cvtmi_last_error());
        ef_ = 10;  // hnswalg.h:560
    }

    void addPoint(void *, labeltype)
    {
        throw std::runtime_error("cvt_amd: HierarchicalNSW::addPoint is not offered on the MI355X build");
    }
    void saveIndex(const std::string &)
    {
        throw std::runtime_error("cvt_amd: HierarchicalNSW::saveIndex is not offered on the MI355X build");
    }

    // hnswalg.h:688-729
    std::priority_queue<std::pair<dist_t, labeltype> > searchKnn(void *query_data, size_t k)
    {
        std::vector<std::priority_queue<std::pair<dist_t, labeltype> > > r = searchKnnBatch(query_data, 1, k);
        return r[0];
    }

    // nq queries, contiguous [nq][dim] floats: what keeps the GPU busy
    std::vector<std::priority_queue<std::pair<dist_t, labeltype> > > searchKnnBatch(const void *queries, size_t nq, size_t k)
    {
        if (!h_) throw std::runtime_error("cvt_amd: HierarchicalNSW: no graph loaded");
        std::vector<float> d(nq * k);
        std::vector<int64_t> l(nq * k);
        if (cvtmi_hnsw_search(h_, (const float *)queries, (int64_t)nq, (int)k, (int)ef_, d.data(), l.data()) != CVTMI_OK)
            throw std::runtime_error(std::string("cvt_amd: ") + cvtmi_last_error());
        std::vector<std::priority_queue<std::pair<dist_t, labeltype> > > out(nq);
        for (size_t q = 0; q < nq; ++q)
            for (size_t i = 0; i < k && l[q * k + i] >= 0; ++i) out[q].push(std::make_pair((dist_t)d[q * k + i], (labeltype)l[q * k + i]));
        return out;
    }

    size_t ntotal() const { return h_ ? (size_t)cvtmi_hnsw_ntotal(h_) : 0; }

private:
    cvtmi_hnsw_t h_;
    size_t ef_, dim_;
};
}  // namespace hnswlib
