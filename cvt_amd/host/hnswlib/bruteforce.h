// BruteforceSearch<dist_t> -- mirror of brute_force_search/src/brutoforce.hpp:9-136 above the C ABI.
// Same members the reference exposes (data_, maxelements_, cur_element_count, size_per_element_, ...), same
// row layout [vector bytes][size_t label] in data_ and in index.bin, same exceptions on duplicate label /
// capacity.  searchKnn runs cvtmi_flat_search on the MI355X.
//
// Tie rule: the reference returns the k smallest (dist, label) pairs.  The device breaks ties by row, so
// rows are uploaded in ascending LABEL order (a permutation only the device copy sees; data_ keeps the
// reference's insertion order for saveIndex).
#pragma once
#include <algorithm>
#include <atomic>
#include <fstream>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <stdlib.h>

#include "../../../include/cvtmi.h"

namespace hnswlib {
template <typename dist_t> class BruteforceSearch : public AlgorithmInterface<dist_t> {
public:
    BruteforceSearch(SpaceInterface<dist_t> *s) : data_(NULL), maxelements_(0), cur_element_count(0), h_(NULL), dirty_(true)
    {
        bind(s);
    }
    BruteforceSearch(SpaceInterface<dist_t> *s, const std::string &location) : data_(NULL), h_(NULL), dirty_(true)
    {
        loadIndex(location, s);
    }
    BruteforceSearch(SpaceInterface<dist_t> *s, size_t maxElements) : h_(NULL), dirty_(true)
    {
        maxelements_ = maxElements;
        bind(s);
        data_ = (char *)malloc(maxElements * size_per_element_ + 1);
        cur_element_count = 0;
    }
    ~BruteforceSearch()
    {
        free(data_);
        if (h_) cvtmi_flat_destroy(h_);
    }

    char *data_;
    size_t maxelements_;
    size_t cur_element_count;
    size_t size_per_element_;
    size_t data_size_;
    DISTFUNC<dist_t> fstdistfunc_;
    void *dist_func_param_;
    std::unordered_map<labeltype, size_t> dict_external_to_internal;

    void addPoint(void *datapoint, labeltype label)
    {
        if (dict_external_to_internal.count(label)) throw std::runtime_error("Ids have to be unique");
        if (cur_element_count >= maxelements_) throw std::runtime_error("The number of elements exceeds the specified limit\n");
        memcpy(data_ + size_per_element_ * cur_element_count + data_size_, &label, sizeof(labeltype));
        memcpy(data_ + size_per_element_ * cur_element_count, datapoint, data_size_);
        dict_external_to_internal[label] = cur_element_count;
        cur_element_count++;
        dirty_ = true;
    }

    void removePoint(labeltype cur_external)
    {
        size_t cur_c = dict_external_to_internal[cur_external];
        dict_external_to_internal.erase(cur_external);
        labeltype label = *((labeltype *)(data_ + size_per_element_ * (cur_element_count - 1) + data_size_));
        dict_external_to_internal[label] = cur_c;
        memcpy(data_ + size_per_element_ * cur_c, data_ + size_per_element_ * (cur_element_count - 1), data_size_ + sizeof(labeltype));
        cur_element_count--;
        dirty_ = true;
        rebuild_ = true;  // a row moved: the device copy is re-ordered from scratch on the next search
    }

    std::priority_queue<std::pair<dist_t, labeltype> > searchKnn(void *query_data, size_t k)
    {
        std::priority_queue<std::pair<dist_t, labeltype> > top;
        if (k == 0) return top;
        std::vector<dist_t> d(k);
        std::vector<int64_t> l(k);
        searchKnnBatch(query_data, 1, k, d.data(), l.data());
        for (size_t i = 0; i < k; ++i)
            if (l[i] >= 0) top.push(std::pair<dist_t, labeltype>(d[i], (labeltype)l[i]));
        return top;
    }

    // nq queries at once (not in the reference): dist[nq][k], labels[nq][k] ascending, -1 = fewer than k rows
    void searchKnnBatch(const void *queries, size_t nq, size_t k, dist_t *dist, int64_t *labels)
    {
        sync_device();
        static_assert(sizeof(dist_t) == 4, "distances are 32-bit (float or int)");
        if (cvtmi_flat_search(h_, queries, (int64_t)nq, (int)k, dist, labels) != CVTMI_OK)
            throw std::runtime_error(std::string("cvtmi_flat_search: ") + cvtmi_last_error());
    }

    // index.bin: size_t max, size_t per_elem, size_t count, raw rows (brutoforce.hpp:95-106)
    void saveIndex(const std::string &location)
    {
        std::ofstream output(location, std::ios::binary);
        writeBinaryPOD(output, maxelements_);
        writeBinaryPOD(output, size_per_element_);
        writeBinaryPOD(output, cur_element_count);
        output.write(data_, maxelements_ * size_per_element_);
        output.close();
    }

    void loadIndex(const std::string &location, SpaceInterface<dist_t> *s)
    {
        std::ifstream input(location, std::ios::binary);
        if (!input.is_open()) throw std::runtime_error("cannot open " + location);
        readBinaryPOD(input, maxelements_);
        readBinaryPOD(input, size_per_element_);
        readBinaryPOD(input, cur_element_count);
        bind(s);
        free(data_);
        data_ = (char *)malloc(maxelements_ * size_per_element_ + 1);
        input.read(data_, maxelements_ * size_per_element_);
        input.close();
        dict_external_to_internal.clear();  // the reference leaves the map empty after a load; rebuilt here
        for (size_t i = 0; i < cur_element_count; ++i)
            dict_external_to_internal[*((labeltype *)(data_ + size_per_element_ * i + data_size_))] = i;
        dirty_ = true;
        rebuild_ = true;
    }

private:
    cvtmi_flat_s *h_;
    std::atomic<bool> dirty_;
    std::mutex sync_mu_;
    bool rebuild_ = true;        // the device copy must be rebuilt (rows removed / loaded / labels not ascending)
    size_t synced_ = 0;          // rows [0, synced_) of data_ are on the device, in this order
    labeltype synced_max_ = 0;   // the largest label among them
    int metric_;

    void bind(SpaceInterface<dist_t> *s)
    {
        data_size_ = s->get_data_size();
        fstdistfunc_ = s->get_dist_func();
        dist_func_param_ = s->get_dist_func_param();
        size_per_element_ = data_size_ + sizeof(labeltype);
        metric_ = s->device_metric();
        if (metric_ < 0) throw std::runtime_error("cvt_amd BruteforceSearch: the space has no MI355X metric");
    }

    // concurrent searchKnn calls (a pure read in the reference, brutoforce.hpp:73-93) may arrive together after a change: the
    // first one brings the device copy up to date, the others wait for it; afterwards nobody takes the lock
    void sync_device()
    {
        if (!dirty_.load(std::memory_order_acquire) && h_) return;
        std::lock_guard<std::mutex> g(sync_mu_);
        if (!dirty_.load(std::memory_order_acquire) && h_) return;
        const size_t dim = *((size_t *)dist_func_param_);
        if (!h_ && cvtmi_flat_create(metric_, (int)dim, &h_) != CVTMI_OK)
            throw std::runtime_error(std::string("cvtmi_flat_create: ") + cvtmi_last_error());
        // The device keeps rows in ascending label order (the (distance, label) tie rule of searchKnn's heap).  The common
        // case -- addPoint with ever larger labels, as brute_force.cpp does (labels = row numbers) -- only appends the new
        // rows; anything else (removePoint, loadIndex, a label below one already uploaded) re-sorts and re-uploads everything.
        if (!rebuild_ && synced_ <= cur_element_count) {
            bool ascending = true;
            labeltype prev = synced_max_;
            for (size_t i = synced_; i < cur_element_count && ascending; ++i) {
                const labeltype lab = *((const labeltype *)(data_ + size_per_element_ * i + data_size_));
                ascending = (i == 0 && synced_ == 0) || lab > prev;
                prev = lab;
            }
            if (ascending) {
                const size_t m = cur_element_count - synced_;
                if (m) {
                    std::vector<char> rows(m * data_size_ + 1);
                    std::vector<int64_t> labels(m);
                    for (size_t i = 0; i < m; ++i) {
                        memcpy(&rows[i * data_size_], data_ + size_per_element_ * (synced_ + i), data_size_);
                        labels[i] = (int64_t) * ((const labeltype *)(data_ + size_per_element_ * (synced_ + i) + data_size_));
                    }
                    if (cvtmi_flat_add(h_, rows.data(), labels.data(), (int64_t)m) != CVTMI_OK)
                        throw std::runtime_error(std::string("cvtmi_flat_add: ") + cvtmi_last_error());
                    synced_max_ = (labeltype)labels[m - 1];
                    synced_ = cur_element_count;
                }
                dirty_.store(false, std::memory_order_release);
                return;
            }
        }
        cvtmi_flat_reset(h_);
        const size_t n = cur_element_count;
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = i;
        const char *base = data_;
        const size_t spe = size_per_element_, ds = data_size_;
        std::sort(order.begin(), order.end(), [=](size_t a, size_t b) {
            return *((const labeltype *)(base + spe * a + ds)) < *((const labeltype *)(base + spe * b + ds));
        });
        std::vector<char> rows(n * ds + 1);
        std::vector<int64_t> labels(n);
        for (size_t i = 0; i < n; ++i) {
            memcpy(&rows[i * ds], data_ + spe * order[i], ds);
            labels[i] = (int64_t) * ((const labeltype *)(data_ + spe * order[i] + ds));
        }
        if (n && cvtmi_flat_add(h_, rows.data(), labels.data(), (int64_t)n) != CVTMI_OK)
            throw std::runtime_error(std::string("cvtmi_flat_add: ") + cvtmi_last_error());
        rebuild_ = false;
        synced_ = n;
        synced_max_ = n ? (labeltype)labels[n - 1] : 0;
        dirty_.store(false, std::memory_order_release);
    }
};
}  // namespace hnswlib
