// vdb_hnsw.hpp — the reference's C++ index interface on top of the C-ABI (include/vdb_index.h).
//
// A user of the reference's src/hnsw/hnsw_scratch keeps their code: the class and method names,
// argument meaning, return types and error behaviour (std::runtime_error) are those of
//   AlgorithmInterface<dist_t>        hnswlib.hpp:71-99
//   SpaceInterface / L2Space / InnerProductSpace   hnswlib.hpp:54-61, space_l2.hpp:16-37, space_ip.hpp:20-41
//   HierarchicalNSW<dist_t>           hnsw_core.hpp:21 (ctor :118-123, load ctor :110-116,
//                                     setefSearch :207, searchKnn :1107, markDelete :709, ...)
//   BruteforceSearch<dist_t>          bruteforce.hpp:14
// with dist_t = float.  Building runs on the host (reference algorithm); every search runs the
// gfx950 kernels.  The index is uploaded lazily on the first search after a change; a batch
// entry point (searchKnnBatch) is added because one query per call cannot fill a GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <limits>
#include <queue>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vdb_index.h"

namespace vdb_amd {

using labeltype = size_t;

class BaseFilterFunctor {  // hnswlib.hpp:64-68
public:
    virtual bool operator()(labeltype) { return true; }
    virtual ~BaseFilterFunctor() {}
};

// hnswlib.hpp:10-23.  The device search cannot call back into the host per candidate, so the conditions it runs are a
// closed set known to the library; the one there is: EpsilonSearchStopCondition (the reference defines none; this
// is hnswlib 0.8.0's).  Any other subclass is refused by searchStopConditionClosest.
template <typename dist_t>
class BaseSearchStopCondition {
public:
    virtual void add_point_to_result(labeltype label, const void* datapoint, dist_t dist) = 0;
    virtual void remove_point_from_result(labeltype label, const void* datapoint, dist_t dist) = 0;
    virtual bool should_stop_search(dist_t candidate_dist, dist_t lowerBound) const = 0;
    virtual bool should_consider_candidate(dist_t candidate_dist, dist_t lowerBound) const = 0;
    virtual bool should_remove_extra() const = 0;
    virtual void filter_results(std::vector<std::pair<dist_t, labeltype>>& candidates) = 0;
    virtual ~BaseSearchStopCondition() = default;
};

class EpsilonSearchStopCondition : public BaseSearchStopCondition<float> {
public:
    float epsilon_;
    size_t min_num_candidates_, max_num_candidates_, curr_num_items_ = 0;
    EpsilonSearchStopCondition(float epsilon, size_t min_num_candidates, size_t max_num_candidates)
        : epsilon_(epsilon), min_num_candidates_(min_num_candidates), max_num_candidates_(max_num_candidates) {}
    void add_point_to_result(labeltype, const void*, float) override { curr_num_items_ += 1; }
    void remove_point_from_result(labeltype, const void*, float) override { curr_num_items_ -= 1; }
    bool should_stop_search(float candidate_dist, float lowerBound) const override {
        if (candidate_dist > lowerBound && curr_num_items_ == max_num_candidates_) return true;
        return candidate_dist > epsilon_ && curr_num_items_ >= min_num_candidates_;
    }
    bool should_consider_candidate(float candidate_dist, float lowerBound) const override {
        return curr_num_items_ < max_num_candidates_ || lowerBound > candidate_dist;
    }
    bool should_remove_extra() const override { return curr_num_items_ > max_num_candidates_; }
    void filter_results(std::vector<std::pair<float, labeltype>>& candidates) override {
        while (!candidates.empty() && candidates.back().first > epsilon_) candidates.pop_back();
        while (candidates.size() > max_num_candidates_) candidates.pop_back();
    }
};

template <typename dist_t>
class AlgorithmInterface {  // hnswlib.hpp:71-99
public:
    virtual void addPoint(const void* datapoint, labeltype label, bool replace_deleted = false) = 0;
    virtual std::priority_queue<std::pair<dist_t, labeltype>> SearchKNN(const void* query, size_t k,
                                                                        BaseFilterFunctor* filter = nullptr) const = 0;
    virtual std::vector<std::pair<dist_t, labeltype>> SearchKNNCloserFirst(const void* query, size_t k,
                                                                           BaseFilterFunctor* filter = nullptr) const {
        auto heap = SearchKNN(query, k, filter);
        std::vector<std::pair<dist_t, labeltype>> out(heap.size());
        for (size_t i = heap.size(); i > 0; --i) {
            out[i - 1] = heap.top();
            heap.pop();
        }
        return out;
    }
    virtual void saveIndex(const std::string& location) = 0;
    virtual ~AlgorithmInterface() = default;
};

struct SpaceInterface {  // hnswlib.hpp:54-61; the distance itself runs on the device
    size_t dim;
    int metric;
    size_t get_data_size() const { return dim * sizeof(float); }
    size_t get_dim() const { return dim; }
};
struct L2Space : SpaceInterface {
    explicit L2Space(size_t d) : SpaceInterface{d, VDB_L2} {}
};
struct InnerProductSpace : SpaceInterface {
    explicit InnerProductSpace(size_t d) : SpaceInterface{d, VDB_IP} {}
};

inline void vdb_check(int rc) {
    if (rc != VDB_OK) throw std::runtime_error(vdb_last_error());
}

class HierarchicalNSW : public AlgorithmInterface<float> {
public:
    HierarchicalNSW(const SpaceInterface* s, size_t capacity, size_t M = 16, size_t efConstruction = 200,
                    size_t random_seed = 100, bool reuse_deleted = false, int device = 0)
        : device_(device) {
        if (reuse_deleted) throw std::runtime_error("reuse_deleted is not implemented");
        vdb_check(vdb_index_create(&idx_, s->metric, s->dim, capacity, M, efConstruction, random_seed));
    }
    HierarchicalNSW(const SpaceInterface* s, const std::string& location, bool /*nmslib*/ = false, size_t capacity = 0,
                    int device = 0)
        : device_(device) {
        vdb_check(vdb_index_load(&idx_, location.c_str(), s->metric, s->dim, capacity));
    }
    // The load constructor without the host copy of the level-0 block (vdb_index_load_to_device): for indexes that are
    // only searched.  `HierarchicalNSW big(&space, path, HierarchicalNSW::StreamToDevice{0});`
    struct StreamToDevice {
        int device = 0;
    };
    HierarchicalNSW(const SpaceInterface* s, const std::string& location, StreamToDevice to) : device_(to.device) {
        vdb_check(vdb_index_load_to_device(&idx_, location.c_str(), s->metric, s->dim, to.device));
        dirty_ = false;
    }
    ~HierarchicalNSW() override { vdb_index_free(idx_); }
    HierarchicalNSW(const HierarchicalNSW&) = delete;
    HierarchicalNSW& operator=(const HierarchicalNSW&) = delete;

    void addPoint(const void* datapoint, labeltype label, bool replace_deleted = false) override {
        if (replace_deleted) throw std::runtime_error("Replacement of deleted elements is disabled in constructor");
        vdb_check(vdb_index_add(idx_, static_cast<const float*>(datapoint), label, -1));
        dirty_ = true;
    }
    void addPoints(const float* vecs, const uint64_t* labels, size_t n, int n_threads) {
        vdb_check(vdb_index_add_batch(idx_, vecs, labels, n, n_threads));
        dirty_ = true;
    }
    // the same insertion with layer 0 built on the GPU (include/vdb_index.h); leaves the index uploaded
    void addPointsDevice(const float* vecs, const uint64_t* labels, size_t n, int n_threads) {
        vdb_check(vdb_index_add_batch_device(idx_, vecs, labels, n, n_threads, device_));
        dirty_ = false;
    }
    void saveIndex(const std::string& location) override { vdb_check(vdb_index_save(idx_, location.c_str())); }
    void setefSearch(size_t ef) { vdb_check(vdb_index_set_ef(idx_, ef)); }
    size_t getCapacity() const { return vdb_index_capacity(idx_); }
    size_t getElementCount() const { return vdb_index_count(idx_); }
    size_t getDeletedCount() const { return vdb_index_deleted_count(idx_); }
    void markDelete(labeltype label) {
        vdb_check(vdb_index_mark_delete(idx_, label));
        dirty_ = true;
    }
    void unmarkDelete(labeltype label) {
        vdb_check(vdb_index_unmark_delete(idx_, label));
        dirty_ = true;
    }
    void resizeIndex(size_t new_capacity) {
        vdb_check(vdb_index_resize(idx_, new_capacity));
        dirty_ = true;
    }
    void checkIntegrity() const {
        if (vdb_index_check_integrity(idx_, nullptr, nullptr) != 0) throw std::runtime_error("integrity check failed");
    }

    // nq queries at once: out_dist/out_label [nq*k], closest first, padded with +inf / SIZE_MAX.
    void searchKnnBatch(const float* queries, size_t nq, size_t k, float* out_dist, uint64_t* out_label) const {
        sync_device();
        vdb_check(vdb_search_batch(idx_, queries, nq, k, out_dist, out_label, nullptr));
    }
    // A functor's verdicts (hnswlib.hpp:64-68, called once per element) as the bitmap over internal
    // ids that the C-ABI takes; evaluate once and reuse it for every batch under the same filter.
    std::vector<uint32_t> filterBitmap(BaseFilterFunctor& filter) const {
        const size_t n = getElementCount();
        std::vector<uint64_t> labels(n);
        if (n) vdb_check(vdb_index_labels(idx_, labels.data(), n));
        std::vector<uint32_t> bits((n + 31) / 32 + 1, 0u);
        for (size_t i = 0; i < n; ++i)
            if (filter((labeltype)labels[i])) bits[i >> 5] |= 1u << (i & 31);
        return bits;
    }
    // searchKnn(query, k, filter) for nq queries: the non-bare path, hnsw_core.hpp:1141-1148
    void searchKnnBatchFiltered(const float* queries, size_t nq, size_t k, const uint32_t* allow, float* out_dist,
                                uint64_t* out_label) const {
        sync_device();
        vdb_check(vdb_search_batch_filtered(idx_, queries, nq, k, allow, out_dist, out_label, nullptr));
    }
    // hnsw_core.hpp:1107-1159: max-heap of (dist, label)
    std::priority_queue<std::pair<float, labeltype>> searchKnn(const void* query, size_t k,
                                                               BaseFilterFunctor* filter = nullptr) const {
        std::vector<float> d(k);
        std::vector<uint64_t> l(k);
        if (filter) {
            const std::vector<uint32_t> allow = filterBitmap(*filter);
            searchKnnBatchFiltered(static_cast<const float*>(query), 1, k, allow.data(), d.data(), l.data());
        } else {
            searchKnnBatch(static_cast<const float*>(query), 1, k, d.data(), l.data());
        }
        std::priority_queue<std::pair<float, labeltype>> out;
        for (size_t i = 0; i < k; ++i)
            if (l[i] != std::numeric_limits<uint64_t>::max()) out.emplace(d[i], (labeltype)l[i]);
        return out;
    }
    std::priority_queue<std::pair<float, labeltype>> SearchKNN(const void* query, size_t k,
                                                               BaseFilterFunctor* filter = nullptr) const override {
        return searchKnn(query, k, filter);
    }
    // hnsw_core.hpp:1162-1213: (distance, label) closest first, at most the condition's max_num_candidates of them.
    // (the reference's :1205 hands back internal ids; labels here, as from searchKnn)
    std::vector<std::pair<float, labeltype>> searchStopConditionClosest(const void* query_data,
                                                                        BaseSearchStopCondition<float>& stop_condition,
                                                                        BaseFilterFunctor* isIdAllowed = nullptr) const {
        auto* eps = dynamic_cast<EpsilonSearchStopCondition*>(&stop_condition);
        if (!eps) throw std::runtime_error("searchStopConditionClosest: the device path runs EpsilonSearchStopCondition only");
        sync_device();
        const size_t k = eps->max_num_candidates_;
        std::vector<float> d(k);
        std::vector<uint64_t> l(k);
        std::vector<uint32_t> allow;
        if (isIdAllowed) allow = filterBitmap(*isIdAllowed);
        vdb_check(vdb_search_epsilon_batch(idx_, static_cast<const float*>(query_data), 1, eps->epsilon_, eps->min_num_candidates_, k,
                                           isIdAllowed ? allow.data() : nullptr, d.data(), l.data(), nullptr));
        std::vector<std::pair<float, labeltype>> result;
        for (size_t i = 0; i < k; ++i)
            if (l[i] != std::numeric_limits<uint64_t>::max()) result.emplace_back(d[i], (labeltype)l[i]);
        return result;
    }
    vdb_index* handle() const { return idx_; }

private:
    void sync_device() const {
        if (dirty_ || vdb_index_device_bytes(idx_) == 0) {
            vdb_check(vdb_index_to_device(idx_, device_));
            dirty_ = false;
        }
    }
    vdb_index* idx_ = nullptr;
    int device_ = 0;
    mutable bool dirty_ = true;
};

class BruteforceSearch : public AlgorithmInterface<float> {
public:
    BruteforceSearch(const SpaceInterface* s, size_t maxElements, int device = 0) : device_(device) {
        vdb_check(vdb_bf_create(&bf_, s->metric, s->dim, maxElements));
    }
    BruteforceSearch(const SpaceInterface* s, const std::string& location, int device = 0) : device_(device) {
        vdb_check(vdb_bf_load(&bf_, location.c_str(), s->metric, s->dim));
    }
    ~BruteforceSearch() override { vdb_bf_free(bf_); }
    BruteforceSearch(const BruteforceSearch&) = delete;
    BruteforceSearch& operator=(const BruteforceSearch&) = delete;

    void addPoint(const void* datapoint, labeltype label, bool = false) override {
        uint64_t l = label;
        vdb_check(vdb_bf_add_batch(bf_, static_cast<const float*>(datapoint), &l, 1));
        dirty_ = true;
    }
    void removePoint(labeltype label) {
        vdb_check(vdb_bf_remove(bf_, label));
        dirty_ = true;
    }
    void saveIndex(const std::string& location) override { vdb_check(vdb_bf_save(bf_, location.c_str())); }
    void searchKnnBatch(const float* queries, size_t nq, size_t k, float* out_dist, uint64_t* out_label,
                        int precision = VDB_BF_FP32) const {
        if (dirty_) {
            vdb_check(vdb_bf_to_device(bf_, device_));
            dirty_ = false;
        }
        vdb_check(vdb_bf_search_batch(bf_, queries, nq, k, precision, out_dist, out_label));
    }
    std::priority_queue<std::pair<float, labeltype>> SearchKNN(const void* query, size_t k,
                                                               BaseFilterFunctor* filter = nullptr) const override {
        std::vector<float> d(k);
        std::vector<uint64_t> l(k);
        if (filter) {  // bruteforce.hpp:158-161, :170-173: the functor's verdicts cross the ABI as a bitmap over rows
            const size_t n = vdb_bf_count(bf_);
            std::vector<uint64_t> labels(n);
            vdb_check(vdb_bf_labels(bf_, labels.data(), n));
            std::vector<uint32_t> allow((n + 31) / 32 + 1, 0u);
            for (size_t i = 0; i < n; ++i)
                if ((*filter)((labeltype)labels[i])) allow[i >> 5] |= 1u << (i & 31);
            if (dirty_) {
                vdb_check(vdb_bf_to_device(bf_, device_));
                dirty_ = false;
            }
            vdb_check(vdb_bf_search_batch_filtered(bf_, static_cast<const float*>(query), 1, k, allow.data(), d.data(), l.data()));
        } else {
            searchKnnBatch(static_cast<const float*>(query), 1, k, d.data(), l.data());
        }
        std::priority_queue<std::pair<float, labeltype>> out;
        for (size_t i = 0; i < k; ++i)
            if (l[i] != std::numeric_limits<uint64_t>::max()) out.emplace(d[i], (labeltype)l[i]);
        return out;
    }

private:
    vdb_bruteforce* bf_ = nullptr;
    int device_ = 0;
    mutable bool dirty_ = true;
};

}  // namespace vdb_amd
