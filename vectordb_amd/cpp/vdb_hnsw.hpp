// vdb_hnsw.hpp — the reference's C++ index interface on top of the C-ABI (include/vdb_index.h).
//
// A user of the reference's src/hnsw/hnsw_scratch keeps their code: the class and method names,
// argument meaning, return types and error behaviour (std::runtime_error) are those of
//   AlgorithmInterface<dist_t>        hnswlib.hpp:71-99
//   SpaceInterface / L2Space / InnerProductSpace   hnswlib.hpp:54-61, space_l2.hpp:16-37, space_ip.hpp:20-41
//   HierarchicalNSW<dist_t>           hnsw_core.hpp:21 (ctor :118-123, load ctor :110-116,
//                                     setefSearch :207, searchKnn :1107, markDelete :709, ...)
//   BruteforceSearch<dist_t>          bruteforce.hpp:14
// in the reference's own spellings - class templates over dist_t / <Scalar, DistanceFunction> - with dist_t = float
// the one instantiation there is (static_assert).  A caller written for the reference compiles unchanged:
//     L2Space<float, L2Sqr<float>> space(dim);
//     HierarchicalNSW<float> idx(&space, n, 16, 200);
//     idx.addPoint(v, label);  idx.setefSearch(64);  auto r = idx.SearchKNNCloserFirst(q, 10);
// (every template argument has a default, so `L2Space space(dim); HierarchicalNSW idx(&space, n);` works too).
// Building runs on the host (reference algorithm); every search runs the
// gfx950 kernels.  The index is uploaded lazily on the first search after a change (under a lock: searchKnn is
// const and concurrency-safe as in the reference); a batch
// entry point (searchKnnBatch) is added because one query per call cannot fill a GPU.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <string>
#include <type_traits>
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

// The host-side distance functors, space_l2.hpp:1-14 and space_ip.hpp:3-18: what get_distance_function_() hands out.
// Sequential fp32 sums in the reference's order (note the reference's signatures: L2Sqr takes the dimension by
// pointer, the inner-product functors by value).  The searches do not call them - they run the kernels.
template <typename Scalar>
struct L2Sqr {
    [[nodiscard]] inline Scalar operator()(const Scalar* vec1, const Scalar* vec2, const size_t* dim) const {
        Scalar result = 0;
        for (size_t i = 0; i < *dim; i++) {
            const Scalar diff = vec1[i] - vec2[i];
            result += diff * diff;
        }
        return result;
    }
};
template <typename Scalar>
struct InnerProduct {
    [[nodiscard]] inline Scalar operator()(const Scalar* vec1, const Scalar* vec2, size_t dim) const {
        Scalar dot_product = 0;
        for (size_t i = 0; i < dim; ++i) dot_product += vec1[i] * vec2[i];
        return dot_product;
    }
};
template <typename Scalar>
struct InnerProductDistance {
    [[nodiscard]] inline Scalar operator()(const Scalar* vec1, const Scalar* vec2, size_t dim) const {
        return 1.0f - InnerProduct<Scalar>{}(vec1, vec2, dim);
    }
};

// which device metric a distance functor stands for
template <typename DistanceFunction>
struct MetricOf;
template <>
struct MetricOf<L2Sqr<float>> {
    static constexpr int value = VDB_L2;
};
template <>
struct MetricOf<InnerProductDistance<float>> {
    static constexpr int value = VDB_IP;
};

// hnswlib.hpp:53-61.  The reference declares SpaceInterface<Scalar, DistanceFunction> but names it with ONE argument
// where an index takes it (hnsw_core.hpp:108-118 `SpaceInterface<dist_t>* space`): both spellings exist here - the
// one-argument form is the base every space derives from.
template <typename Scalar, typename DistanceFunction = void>
class SpaceInterface;
template <typename Scalar>
class SpaceInterface<Scalar, void> {
public:
    virtual size_t get_data_size() const = 0;
    virtual size_t get_dim() const = 0;
    virtual int vdb_metric() const = 0;  // VDB_L2 / VDB_IP: what crosses the C-ABI in place of the functor
    virtual ~SpaceInterface() = default;
};
template <typename Scalar, typename DistanceFunction>
class SpaceInterface : public SpaceInterface<Scalar, void> {
public:
    virtual DistanceFunction get_distance_function_() const = 0;
    int vdb_metric() const override { return MetricOf<DistanceFunction>::value; }
};

template <typename Scalar = float, typename DistanceFunction = L2Sqr<Scalar>>
class L2Space : public SpaceInterface<Scalar, DistanceFunction> {  // space_l2.hpp:16-37
    DistanceFunction distance_function_;
    size_t dim_;
    size_t data_size_;

public:
    L2Space(size_t dim) : distance_function_(), dim_(dim), data_size_(dim * sizeof(Scalar)) {}
    size_t get_data_size() const override { return data_size_; }
    DistanceFunction get_distance_function_() const override { return distance_function_; }
    size_t get_dim() const override { return dim_; }
};

template <typename Scalar = float, typename DistanceFunction = InnerProductDistance<Scalar>>
class InnerProductSpace : public SpaceInterface<Scalar, DistanceFunction> {  // space_ip.hpp:20-41
    DistanceFunction distance_function_;
    size_t dim_;
    size_t data_size_;

public:
    InnerProductSpace(size_t dim) : distance_function_(), dim_(dim), data_size_(dim * sizeof(Scalar)) {}
    size_t get_data_size() const override { return data_size_; }
    DistanceFunction get_distance_function_() const override { return distance_function_; }
    size_t get_dim() const override { return dim_; }
};

inline void vdb_check(int rc) {
    if (rc != VDB_OK) throw std::runtime_error(vdb_last_error());
}

template <typename dist_t = float>
class HierarchicalNSW : public AlgorithmInterface<dist_t> {  // hnsw_core.hpp:20-21
    static_assert(std::is_same<dist_t, float>::value, "the device path computes float distances (the reference's one working instantiation)");

public:
    // hnsw_core.hpp:118-123 (+ the device ordinal)
    HierarchicalNSW(const SpaceInterface<dist_t>* s, size_t capacity, size_t M = 16, size_t efConstruction = 200,
                    size_t random_seed = 100, bool reuse_deleted = false, int device = 0)
        : device_(device) {
        vdb_check(vdb_index_create(&idx_, s->vdb_metric(), s->get_dim(), capacity, M, efConstruction, random_seed));
        init_reuse(reuse_deleted);
    }
    // hnsw_core.hpp:110-116
    HierarchicalNSW(const SpaceInterface<dist_t>* s, const std::string& location, bool /*nmslib*/ = false, size_t capacity = 0,
                    bool reuse_deleted = false, int device = 0)
        : device_(device) {
        vdb_check(vdb_index_load(&idx_, location.c_str(), s->vdb_metric(), s->get_dim(), capacity));
        init_reuse(reuse_deleted);
    }
    // The load constructor without the host copy of the level-0 block (vdb_index_load_to_device): for indexes that are
    // only searched.  `HierarchicalNSW<float> big(&space, path, HierarchicalNSW<float>::StreamToDevice{0});`
    struct StreamToDevice {
        int device = 0;
    };
    HierarchicalNSW(const SpaceInterface<dist_t>* s, const std::string& location, StreamToDevice to) : device_(to.device) {
        vdb_check(vdb_index_load_to_device(&idx_, location.c_str(), s->vdb_metric(), s->get_dim(), to.device));
        dirty_ = false;
    }
    ~HierarchicalNSW() override { vdb_index_free(idx_); }
    HierarchicalNSW(const HierarchicalNSW&) = delete;
    HierarchicalNSW& operator=(const HierarchicalNSW&) = delete;

    // hnsw_core.hpp:804-842: inserts; updates the element when the label exists (:1001-1016); with replace_deleted
    // takes the place of a deleted element (needs reuse_deleted in the constructor, :805-807)
    void addPoint(const void* datapoint, labeltype label, bool replace_deleted = false) override {
        vdb_check(vdb_index_add_point(idx_, static_cast<const float*>(datapoint), label, replace_deleted ? 1 : 0));
        touch();
    }
    void addPoints(const float* vecs, const uint64_t* labels, size_t n, int n_threads) {
        vdb_check(vdb_index_add_batch(idx_, vecs, labels, n, n_threads));
        touch();
    }
    // the same insertion with layer 0 built on the GPU (include/vdb_index.h); leaves the index uploaded
    void addPointsDevice(const float* vecs, const uint64_t* labels, size_t n, int n_threads) {
        std::unique_lock<std::mutex> g(upload_lock_);
        vdb_check(vdb_index_add_batch_device(idx_, vecs, labels, n, n_threads, device_));
        version_ += 1;
        dirty_ = false;
    }
    void saveIndex(const std::string& location) override { vdb_check(vdb_index_save(idx_, location.c_str())); }
    void setefSearch(size_t ef) { vdb_check(vdb_index_set_ef(idx_, ef)); }
    size_t getCapacity() const { return vdb_index_capacity(idx_); }
    size_t getElementCount() const { return vdb_index_count(idx_); }
    size_t getDeletedCount() const { return vdb_index_deleted_count(idx_); }
    void markDelete(labeltype label) {
        vdb_check(vdb_index_mark_delete(idx_, label));
        touch();
    }
    void unmarkDelete(labeltype label) {
        vdb_check(vdb_index_unmark_delete(idx_, label));
        touch();
    }
    void resizeIndex(size_t new_capacity) {
        vdb_check(vdb_index_resize(idx_, new_capacity));
        touch();
    }
    // hnsw_core.hpp:683-703
    template <typename data_t>
    std::vector<data_t> getDataByLabel(labeltype label) const {
        static_assert(std::is_same<data_t, float>::value, "the index stores float vectors");
        std::vector<data_t> data(vdb_index_dim(idx_));
        vdb_check(vdb_index_get_data_by_label(idx_, label, data.data()));
        return data;
    }
    void checkIntegrity() const {
        if (vdb_index_check_integrity(idx_, nullptr, nullptr) != 0) throw std::runtime_error("integrity check failed");
    }

    // nq queries at once: out_dist/out_label [nq*k], closest first, padded with +inf / SIZE_MAX.
    void searchKnnBatch(const float* queries, size_t nq, size_t k, float* out_dist, uint64_t* out_label) const {
        sync_device();
        vdb_check(vdb_search_batch(idx_, queries, nq, k, out_dist, out_label, nullptr));
    }
    // A functor's verdicts (hnswlib.hpp:64-68) as the bitmap over internal ids that the C-ABI takes: the functor is
    // called once per ELEMENT here, where the reference calls it once per candidate the search admits
    // (hnsw_core.hpp:340-343 shape) - the same answers for a functor that is a pure function of the label, and a
    // documented deviation for one that counts its calls.  Evaluate once and reuse it for every
    // batch under the same filter.
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
    // the batch call that takes the functor itself: evaluated once for the whole batch
    void searchKnnBatch(const float* queries, size_t nq, size_t k, float* out_dist, uint64_t* out_label,
                        BaseFilterFunctor* filter) const {
        if (!filter) return searchKnnBatch(queries, nq, k, out_dist, out_label);
        const std::vector<uint32_t> allow = filterBitmap(*filter);
        searchKnnBatchFiltered(queries, nq, k, allow.data(), out_dist, out_label);
    }
    // Single-query calls with a functor evaluate it over all n labels per call (O(n) host work).  A caller whose
    // functor is a pure function of the label can switch this on: the bitmap is then kept per (functor object, index
    // version) and re-made only when the index changes or another functor comes.  Off by default: a functor whose
    // state changes between calls (the reference's "store state for filtering, e.g. from a std::set") would otherwise
    // be answered from stale verdicts.
    void setFilterCaching(bool on) {
        std::unique_lock<std::mutex> g(filter_lock_);
        filter_caching_ = on;
        cached_filter_ = nullptr;
        cached_bits_.clear();
    }
    // hnsw_core.hpp:1107-1159: max-heap of (dist, label)
    std::priority_queue<std::pair<dist_t, labeltype>> searchKnn(const void* query, size_t k,
                                                                BaseFilterFunctor* filter = nullptr) const {
        std::vector<float> d(k);
        std::vector<uint64_t> l(k);
        if (filter) {
            std::unique_lock<std::mutex> g(filter_lock_);
            if (filter_caching_) {
                if (cached_filter_ != filter || cached_version_ != version_.load()) {
                    cached_version_ = version_.load();
                    cached_bits_ = filterBitmap(*filter);
                    cached_filter_ = filter;
                }
                searchKnnBatchFiltered(static_cast<const float*>(query), 1, k, cached_bits_.data(), d.data(), l.data());
            } else {
                g.unlock();
                const std::vector<uint32_t> allow = filterBitmap(*filter);
                searchKnnBatchFiltered(static_cast<const float*>(query), 1, k, allow.data(), d.data(), l.data());
            }
        } else {
            searchKnnBatch(static_cast<const float*>(query), 1, k, d.data(), l.data());
        }
        std::priority_queue<std::pair<dist_t, labeltype>> out;
        for (size_t i = 0; i < k; ++i)
            if (l[i] != std::numeric_limits<uint64_t>::max()) out.emplace(d[i], (labeltype)l[i]);
        return out;
    }
    std::priority_queue<std::pair<dist_t, labeltype>> SearchKNN(const void* query, size_t k,
                                                                BaseFilterFunctor* filter = nullptr) const override {
        return searchKnn(query, k, filter);
    }
    // hnsw_core.hpp:1162-1213: (distance, label) closest first, at most the condition's max_num_candidates of them.
    // (the reference's :1205 hands back internal ids; labels here, as from searchKnn)
    std::vector<std::pair<dist_t, labeltype>> searchStopConditionClosest(const void* query_data,
                                                                         BaseSearchStopCondition<dist_t>& stop_condition,
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
        std::vector<std::pair<dist_t, labeltype>> result;
        for (size_t i = 0; i < k; ++i)
            if (l[i] != std::numeric_limits<uint64_t>::max()) result.emplace_back(d[i], (labeltype)l[i]);
        return result;
    }
    vdb_index* handle() const { return idx_; }

private:
    void init_reuse(bool reuse_deleted) {
        if (!reuse_deleted) return;
        const int rc = vdb_index_set_reuse_deleted(idx_, 1);
        if (rc != VDB_OK) {
            vdb_index_free(idx_);
            idx_ = nullptr;
            vdb_check(rc);
        }
    }
    void touch() {  // the host index changed: the device copy is stale, cached filter verdicts too
        version_ += 1;
        dirty_ = true;
    }
    // searchKnn is const and may be called from many threads: the lazy upload runs once, under a lock
    void sync_device() const {
        if (!dirty_.load(std::memory_order_acquire)) return;
        std::unique_lock<std::mutex> g(upload_lock_);
        if (dirty_.load(std::memory_order_relaxed)) {
            vdb_check(vdb_index_to_device(idx_, device_));
            dirty_.store(false, std::memory_order_release);
        }
    }
    vdb_index* idx_ = nullptr;
    int device_ = 0;
    mutable std::mutex upload_lock_, filter_lock_;
    mutable std::atomic<bool> dirty_{true};
    std::atomic<uint64_t> version_{0};
    bool filter_caching_ = false;
    mutable BaseFilterFunctor* cached_filter_ = nullptr;
    mutable uint64_t cached_version_ = 0;
    mutable std::vector<uint32_t> cached_bits_;
};

template <typename dist_t = float>
class BruteforceSearch : public AlgorithmInterface<dist_t> {  // bruteforce.hpp:13-14
    static_assert(std::is_same<dist_t, float>::value, "the device path computes float distances");

public:
    BruteforceSearch(const SpaceInterface<dist_t>* s, size_t maxElements, int device = 0) : device_(device) {  // bruteforce.hpp:54-65
        vdb_check(vdb_bf_create(&bf_, s->vdb_metric(), s->get_dim(), maxElements));
    }
    BruteforceSearch(const SpaceInterface<dist_t>* s, const std::string& location, int device = 0) : device_(device) {  // :49-52
        vdb_check(vdb_bf_load(&bf_, location.c_str(), s->vdb_metric(), s->get_dim()));
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
        sync_device();
        vdb_check(vdb_bf_search_batch(bf_, queries, nq, k, precision, out_dist, out_label));
    }
    std::priority_queue<std::pair<dist_t, labeltype>> SearchKNN(const void* query, size_t k,
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
            sync_device();
            vdb_check(vdb_bf_search_batch_filtered(bf_, static_cast<const float*>(query), 1, k, allow.data(), d.data(), l.data()));
        } else {
            searchKnnBatch(static_cast<const float*>(query), 1, k, d.data(), l.data());
        }
        std::priority_queue<std::pair<dist_t, labeltype>> out;
        for (size_t i = 0; i < k; ++i)
            if (l[i] != std::numeric_limits<uint64_t>::max()) out.emplace(d[i], (labeltype)l[i]);
        return out;
    }

private:
    void sync_device() const {
        if (!dirty_.load(std::memory_order_acquire)) return;
        std::unique_lock<std::mutex> g(upload_lock_);
        if (dirty_.load(std::memory_order_relaxed)) {
            vdb_check(vdb_bf_to_device(bf_, device_));
            dirty_.store(false, std::memory_order_release);
        }
    }
    vdb_bruteforce* bf_ = nullptr;
    int device_ = 0;
    mutable std::mutex upload_lock_;
    mutable std::atomic<bool> dirty_{true};
};

}  // namespace vdb_amd
