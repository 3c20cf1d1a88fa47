// oracle_capi.cpp — plain-C entry points over vdb_oracle.hpp so that tests/ and
// bench.py's cpu_baseline leg can drive the CPU restatement through ctypes.
// TEST INFRASTRUCTURE ONLY (see the header of vdb_oracle.hpp).
#include <algorithm>
#include <chrono>
#include <thread>

#include "vdb_oracle.hpp"

using namespace vdb_oracle;

namespace {
thread_local std::string g_err;
int fail(const std::exception& e) {
    g_err = e.what();
    return -1;
}
Space make_space(int metric, int mode, size_t dim) {
    Space s;
    s.metric = metric == 0 ? kL2 : kIP;
    s.mode = mode == 0 ? kSeq : kTree64;
    s.dim = dim;
    return s;
}
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

float orc_distance(int metric, int mode, const float* a, const float* b, size_t dim) {
    return make_space(metric, mode, dim)(a, b);
}

// visited_list_pool.hpp:42-53, :78-91 — tags handed out by n consecutive get/release pairs
void orc_visited_tags(int n_elements, int n_rounds, unsigned short* tags_out, int* was_cleared_out) {
    VisitedListPool pool(1, n_elements);
    for (int r = 0; r < n_rounds; r++) {
        VisitedList* vl = pool.getFreeVisitedList();
        tags_out[r] = vl->currentVisited;
        int dirty = 0;
        for (int i = 0; i < n_elements; i++) dirty |= (vl->visitedAt[i] == vl->currentVisited);
        was_cleared_out[r] = !dirty;
        vl->visitedAt[r % n_elements] = vl->currentVisited;
        pool.releaseVisitedList(vl);
    }
}

// ---- HNSW ------------------------------------------------------------------
void* orc_hnsw_create(int metric, int mode, size_t dim, size_t capacity, size_t M, size_t efc, size_t seed) {
    try {
        return new HierarchicalNSW(make_space(metric, mode, dim), capacity, M, efc, seed);
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
}
void* orc_hnsw_load(const char* path, int metric, int mode, size_t dim) {
    try {
        return new HierarchicalNSW(make_space(metric, mode, dim), std::string(path));
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
}
// the constructors' reuse_deleted flag (hnsw_core.hpp:114, :123)
void* orc_hnsw_create_reuse(int metric, int mode, size_t dim, size_t capacity, size_t M, size_t efc, size_t seed, int reuse_deleted) {
    try {
        return new HierarchicalNSW(make_space(metric, mode, dim), capacity, M, efc, seed, reuse_deleted != 0);
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
}
void* orc_hnsw_load_reuse(const char* path, int metric, int mode, size_t dim, int reuse_deleted) {
    try {
        return new HierarchicalNSW(make_space(metric, mode, dim), std::string(path), 0, reuse_deleted != 0);
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
}
// addPoint(datapoint, label, replace_deleted), hnsw_core.hpp:804-842
int orc_hnsw_add_replace(void* h, const float* vec, size_t label, int replace_deleted) {
    try {
        static_cast<AlgorithmInterface*>(static_cast<HierarchicalNSW*>(h))->addPoint(vec, label, replace_deleted != 0);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
void orc_hnsw_free(void* h) { delete static_cast<HierarchicalNSW*>(h); }
int orc_hnsw_save(void* h, const char* path) {
    try {
        static_cast<HierarchicalNSW*>(h)->saveIndex(path);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_hnsw_add(void* h, const float* vec, size_t label, int level) {
    try {
        static_cast<HierarchicalNSW*>(h)->addPoint(vec, label, level);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_hnsw_add_batch(void* h, const float* vecs, const size_t* labels, size_t n) {
    try {
        auto* idx = static_cast<HierarchicalNSW*>(h);
        size_t dim = idx->space_.dim;
        for (size_t i = 0; i < n; i++) idx->addPoint(vecs + i * dim, labels ? labels[i] : i, -1);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
void orc_hnsw_set_ef(void* h, size_t ef) { static_cast<HierarchicalNSW*>(h)->setefSearch(ef); }
void orc_hnsw_set_mode(void* h, int mode) { static_cast<HierarchicalNSW*>(h)->space_.mode = mode == 0 ? kSeq : kTree64; }
size_t orc_hnsw_count(void* h) { return static_cast<HierarchicalNSW*>(h)->getElementCount(); }
size_t orc_hnsw_capacity(void* h) { return static_cast<HierarchicalNSW*>(h)->getCapacity(); }
size_t orc_hnsw_deleted(void* h) { return static_cast<HierarchicalNSW*>(h)->getDeletedCount(); }
int orc_hnsw_max_level(void* h) { return static_cast<HierarchicalNSW*>(h)->max_level_; }
unsigned orc_hnsw_entry(void* h) { return static_cast<HierarchicalNSW*>(h)->entry_id_; }
int orc_hnsw_level_of(void* h, unsigned id) { return static_cast<HierarchicalNSW*>(h)->element_levels_[id]; }
int orc_hnsw_check_integrity(void* h, int* min_in, int* max_in) {
    return static_cast<HierarchicalNSW*>(h)->checkIntegrity(min_in, max_in);
}
int orc_hnsw_mark_delete(void* h, size_t label, int undo) {
    try {
        if (undo)
            static_cast<HierarchicalNSW*>(h)->unmarkDelete(label);
        else
            static_cast<HierarchicalNSW*>(h)->markDelete(label);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
// neighbours of one node on one level, in list order; returns the count
int orc_hnsw_links(void* h, unsigned id, int level, unsigned* out, int cap) {
    auto* idx = static_cast<HierarchicalNSW*>(h);
    if (level > idx->element_levels_[id]) return -1;
    idx_t* list = idx->get_neighbors_at_level(id, level);
    int n = idx->getListCount(list);
    for (int i = 0; i < n && i < cap; i++) out[i] = list[1 + i];
    return n;
}

namespace {
struct ModFilter : BaseFilterFunctor {
    size_t mod, keep;
    bool operator()(label_t l) override { return l % mod == keep; }
};
}  // namespace

// SearchKNNCloserFirst for nq queries (hnswlib.hpp:87-99 over hnsw_core.hpp:1107-1159).
// Outputs are [nq*k], closest first, rows short of k padded with (+inf, SIZE_MAX).
// filter_mod > 0 applies the filter `label % filter_mod == filter_keep`.
// Returns wall seconds of the search loop (< 0 on error).
double orc_hnsw_search(void* h, const float* queries, size_t nq, size_t k, float* out_dist, size_t* out_label,
                       unsigned* out_count, uint64_t* out_ndis, uint64_t* out_nhops0, uint64_t* out_nhops_upper,
                       int n_threads, size_t filter_mod, size_t filter_keep) {
    auto* idx = static_cast<HierarchicalNSW*>(h);
    size_t dim = idx->space_.dim;
    if (n_threads < 1) n_threads = 1;
    std::atomic<int> bad{0};
    std::string err;
    std::mutex err_lock;
    auto work = [&](size_t lo, size_t hi) {
        ModFilter mf;
        mf.mod = filter_mod;
        mf.keep = filter_keep;
        BaseFilterFunctor* filter = filter_mod ? &mf : nullptr;
        try {
            for (size_t q = lo; q < hi; q++) {
                SearchStats st;
                auto heap = idx->searchKnn(queries + q * dim, k, filter, &st);
                std::vector<std::pair<float, label_t>> res(heap.size());
                for (size_t i = heap.size(); i > 0; --i) {
                    res[i - 1] = heap.top();
                    heap.pop();
                }
                for (size_t i = 0; i < k; i++) {
                    out_dist[q * k + i] = i < res.size() ? res[i].first : std::numeric_limits<float>::infinity();
                    out_label[q * k + i] = i < res.size() ? res[i].second : (size_t)-1;
                }
                if (out_count) out_count[q] = (unsigned)res.size();
                if (out_ndis) out_ndis[q] = st.ndis;
                if (out_nhops0) out_nhops0[q] = st.nhops0;
                if (out_nhops_upper) out_nhops_upper[q] = st.nhops_upper;
            }
        } catch (const std::exception& e) {
            std::unique_lock<std::mutex> l(err_lock);
            err = e.what();
            bad = 1;
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (n_threads == 1) {
        work(0, nq);
    } else {
        std::vector<std::thread> pool;
        size_t per = (nq + n_threads - 1) / n_threads;
        for (int t = 0; t < n_threads; t++) {
            size_t lo = std::min(nq, t * per), hi = std::min(nq, lo + per);
            if (lo < hi) pool.emplace_back(work, lo, hi);
        }
        for (auto& th : pool) th.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (bad) {
        g_err = err;
        return -1.0;
    }
    return std::chrono::duration<double>(t1 - t0).count();
}

// searchStopConditionClosest (hnsw_core.hpp:1162-1213) with the epsilon condition for nq queries.  Outputs are
// [nq * max_cand], closest first (ties by label, as the GPU epilogue orders them), padded with (+inf, SIZE_MAX);
// out_count[q] = results of query q.  Returns 0, < 0 on error.
int orc_hnsw_search_epsilon(void* h, const float* queries, size_t nq, float epsilon, size_t min_cand, size_t max_cand,
                            float* out_dist, size_t* out_label, unsigned* out_count, uint64_t* out_ndis, uint64_t* out_nhops0,
                            size_t filter_mod, size_t filter_keep) {
    auto* idx = static_cast<HierarchicalNSW*>(h);
    const size_t dim = idx->space_.dim;
    ModFilter mf;
    mf.mod = filter_mod;
    mf.keep = filter_keep;
    try {
        for (size_t q = 0; q < nq; q++) {
            EpsilonSearchStopCondition stop(epsilon, min_cand, max_cand);
            SearchStats st;
            auto res = idx->searchStopConditionClosest(queries + q * dim, stop, filter_mod ? &mf : nullptr, &st);
            std::stable_sort(res.begin(), res.end());  // (dist, label): the tie order of DESIGN.md
            for (size_t i = 0; i < max_cand; i++) {
                out_dist[q * max_cand + i] = i < res.size() ? res[i].first : std::numeric_limits<float>::infinity();
                out_label[q * max_cand + i] = i < res.size() ? res[i].second : (size_t)-1;
            }
            if (out_count) out_count[q] = (unsigned)res.size();
            if (out_ndis) out_ndis[q] = st.ndis;
            if (out_nhops0) out_nhops0[q] = st.nhops0;
        }
    } catch (const std::exception& e) {
        return fail(e);
    }
    return 0;
}

// ---- brute force -------------------------------------------------------------
void* orc_bf_create(int metric, int mode, size_t dim, size_t capacity) {
    try {
        return new BruteforceSearch(make_space(metric, mode, dim), capacity);
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
}
void* orc_bf_load(const char* path, int metric, int mode, size_t dim) {
    try {
        return new BruteforceSearch(make_space(metric, mode, dim), std::string(path));
    } catch (const std::exception& e) {
        fail(e);
        return nullptr;
    }
}
void orc_bf_free(void* h) { delete static_cast<BruteforceSearch*>(h); }
int orc_bf_save(void* h, const char* path) {
    try {
        static_cast<BruteforceSearch*>(h)->saveIndex(path);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
int orc_bf_add_batch(void* h, const float* vecs, const size_t* labels, size_t n) {
    try {
        auto* idx = static_cast<BruteforceSearch*>(h);
        size_t dim = idx->space_.dim;
        for (size_t i = 0; i < n; i++) idx->addPoint(vecs + i * dim, labels ? labels[i] : i);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}
void orc_bf_remove(void* h, size_t label) { static_cast<BruteforceSearch*>(h)->removePoint(label); }
size_t orc_bf_count(void* h) { return static_cast<BruteforceSearch*>(h)->element_count_; }

double orc_bf_search(void* h, const float* queries, size_t nq, size_t k, float* out_dist, size_t* out_label,
                     unsigned* out_count, int n_threads, size_t filter_mod, size_t filter_keep) {
    auto* idx = static_cast<BruteforceSearch*>(h);
    size_t dim = idx->space_.dim;
    if (n_threads < 1) n_threads = 1;
    auto work = [&](size_t lo, size_t hi) {
        ModFilter mf;
        mf.mod = filter_mod;
        mf.keep = filter_keep;
        BaseFilterFunctor* filter = filter_mod ? &mf : nullptr;
        for (size_t q = lo; q < hi; q++) {
            auto res = idx->SearchKNNCloserFirst(queries + q * dim, k, filter);
            for (size_t i = 0; i < k; i++) {
                out_dist[q * k + i] = i < res.size() ? res[i].first : std::numeric_limits<float>::infinity();
                out_label[q * k + i] = i < res.size() ? res[i].second : (size_t)-1;
            }
            if (out_count) out_count[q] = (unsigned)res.size();
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (n_threads == 1) {
        work(0, nq);
    } else {
        std::vector<std::thread> pool;
        size_t per = (nq + n_threads - 1) / n_threads;
        for (int t = 0; t < n_threads; t++) {
            size_t lo = std::min(nq, t * per), hi = std::min(nq, lo + per);
            if (lo < hi) pool.emplace_back(work, lo, hi);
        }
        for (auto& th : pool) th.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
