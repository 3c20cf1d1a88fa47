// host_index.hpp — host-resident HNSW index in the reference's byte layout, plus the
// graph builder.  This is the "index interface and graph layout stay drop-in" half of the
// path: it owns what the reference's HierarchicalNSW owns (hnsw_core.hpp:28-105), reads
// and writes the reference's saveIndex format (:551-682), and inserts points with the
// reference's algorithm (:994-1104, :293-498) so that the graphs are the same graphs.
// Searching is NOT done here: queries go to the gfx950 kernels (hnsw_search.hip).
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_distance.hpp"

namespace vdb {

// One byte per node; the reference uses a std::mutex per node (hnsw_core.hpp:80).
struct SpinLock {
    std::atomic<uint8_t> f{0};
    void lock() {
        while (f.exchange(1, std::memory_order_acquire)) {
            while (f.load(std::memory_order_relaxed)) __builtin_ia32_pause();
        }
    }
    void unlock() { f.store(0, std::memory_order_release); }
};

struct DistId {
    float d;
    uint32_t id;
};

// Per-thread scratch for the construction-time beam search.
struct BuildScratch {
    std::vector<uint32_t> seen;  // epoch per node
    uint32_t epoch = 0;
    std::vector<DistId> top, cand, work, keep;
    void ensure(size_t n) {
        if (seen.size() < n) {
            seen.assign(n, 0);
            epoch = 0;
        }
    }
    uint32_t next_epoch() {
        if (++epoch == 0) {
            std::fill(seen.begin(), seen.end(), 0);
            epoch = 1;
        }
        return epoch;
    }
};

class HostIndex {
public:
    // reference fields, hnsw_core.hpp:28-75
    int metric = 0;
    size_t dim = 0;
    size_t capacity = 0;
    std::atomic<size_t> count{0};
    std::atomic<size_t> deleted{0};
    int max_level = -1;
    uint32_t entry = 0xFFFFFFFFu;
    std::vector<int> levels;
    size_t M = 0, maxM = 0, maxM0 = 0, ef_construction = 0, ef_search = 10;
    double level_lambda = 0.0;
    size_t stride = 0, link0_off = 0, link0_stride = 0, data_off = 0, data_size = 0, label_off = 0;
    size_t link_stride = 0;
    char* level0 = nullptr;
    char** upper = nullptr;
    int sum_order = 1;  // vdb_sum_order
    uint64_t version = 0;  // bumped by every mutation; the device copy remembers the one it mirrors
    // An index opened for streaming to the device (vdb_index_load_to_device): everything but the level-0 block
    // (level0 == nullptr) is here; the block - rows [list | vector | label], all but a few percent of the file -
    // is read chunk by chunk by the uploader from rows_path at rows_offset and never kept on the host.
    bool rows_resident = true;
    std::string rows_path;
    long long rows_offset = 0;
    void require_rows(const char* what) const {
        if (!rows_resident)
            throw std::runtime_error(std::string(what) + ": this index was loaded straight to the device and keeps no rows on the "
                                     "host - load it with vdb_index_load to edit, inspect or save it");
    }

    // layer0_lists_as_upper: layer 0 gets the capacity of an upper layer (maxM0 = M).  Only the device
    // builder uses it, for the HNSW over the elements of level >= l that IS layer l of a larger index.
    HostIndex(int metric, size_t dim, size_t capacity, size_t M, size_t efc, size_t seed, bool layer0_lists_as_upper = false);
    // stream_rows: see rows_resident
    HostIndex(const std::string& path, int metric, size_t dim, size_t capacity_hint, bool stream_rows = false);
    ~HostIndex();
    HostIndex(const HostIndex&) = delete;
    HostIndex& operator=(const HostIndex&) = delete;

    void add(const float* vec, uint64_t label, int level);
    // addPoint(datapoint, label, replace_deleted), hnsw_core.hpp:804-842: a label that exists is UPDATED in place
    // (updatePoint :844-913, repairConnectionsForUpdate :924-980); with replace_deleted a deleted element's place is reused.
    void add_point(const float* vec, uint64_t label, bool replace_deleted);
    // the constructors' reuse_deleted flag (:114, :123): deleted elements are remembered as vacant places
    void set_reuse_deleted(bool on);
    bool reuse_deleted() const { return reuse_deleted_; }
    void add_batch(const float* vecs, const uint64_t* labels, size_t n, int n_threads, int forced_level = -1);
    // the two halves of add_batch, for the device builder (hnsw_build.hip): store, then link a subset on the host
    size_t append_unlinked(const float* vecs, const uint64_t* labels, size_t n, int forced_level = -1,
                           const int* given_levels = nullptr);
    // upper-layer lists of `id` (levels[id] of them), zeroed; the element must not have any yet
    void alloc_upper_lists(uint32_t id);
    void set_entry(uint32_t id, int level) {
        entry = id;
        max_level = level;
        ++version;
    }
    void link_elements(const uint32_t* ids, size_t n, int n_threads);
    // Undo for the device builder: drop every element (rows, labels, levels, lists), keep the capacity.
    void truncate_to_empty();
    uint64_t rng_state() const { return rng_.s; }
    void set_rng_state(uint64_t s) { rng_.s = s; }
    void save(const std::string& path) const;
    void mark_delete(uint64_t label, bool undo);
    void get_data_by_label(uint64_t label, float* out);  // getDataByLabel, :683-703
    void resize(size_t new_capacity);
    int check_integrity(int* min_in, int* max_in) const;

    // accessors, hnsw_core.hpp:219-290
    char* row(uint32_t id) const { return level0 + (size_t)id * stride; }
    const float* vec(uint32_t id) const { return (const float*)(row(id) + data_off); }
    uint32_t* list0(uint32_t id) const { return (uint32_t*)(row(id) + link0_off); }
    uint32_t* list_at(uint32_t id, int level) const {
        return level == 0 ? list0(id) : (uint32_t*)(upper[id] + (size_t)(level - 1) * link_stride);
    }
    uint64_t label_of(uint32_t id) const {
        uint64_t l;
        __builtin_memcpy(&l, row(id) + label_off, 8);
        return l;
    }
    bool is_deleted(uint32_t id) const { return (((const unsigned char*)list0(id))[2] & 0x01) != 0; }
    static unsigned list_count(const uint32_t* p) { return *(const unsigned short*)p; }
    static void set_list_count(uint32_t* p, unsigned n) { *(unsigned short*)p = (unsigned short)n; }

private:
    struct LevelRng {
        uint64_t s = 1;
        void seed(uint64_t v) { s = (v % 2147483647ull == 0) ? 1 : v % 2147483647ull; }
        uint64_t next() { return s = (s * 16807ull) % 2147483647ull; }
        double uniform();
    } rng_;
    std::mutex global_lock_;
    std::mutex label_lock_;
    std::unique_ptr<SpinLock[]> node_locks_;
    std::unordered_map<uint64_t, uint32_t> label_map_;
    dist_fn dist_ = nullptr;

    bool layer0_as_upper_ = false;
    bool reuse_deleted_ = false;             // :103
    std::mutex deleted_lock_;                // :104
    std::set<uint32_t> deleted_elements_;    // :105 (ordered: the vacant place taken first is the smallest id - the
                                             // reference takes begin() of an unordered_set, i.e. an unspecified one)
    void update_point(uint32_t id, const float* vec, BuildScratch& sc);
    void repair_connections(uint32_t id, const float* vec, uint32_t entry_copy, int elem_level, int max_level_copy, BuildScratch& sc);
    void set_deleted_flag(uint32_t id, bool deleted);
    void init_layout();
    void alloc_storage(size_t cap, bool with_rows = true);
    void release();
    int draw_level();
    void insert(uint32_t id, const float* vec, int level, BuildScratch& sc);
    void beam_search_layer(uint32_t ep, const float* q, int layer, BuildScratch& sc);
    void prune_heuristic(std::vector<DistId>& pool, size_t m, BuildScratch& sc);
    uint32_t connect(uint32_t id, int level, BuildScratch& sc, bool update = false);
    void refresh_dist() { dist_ = pick_distance(metric, sum_order); }

public:
    void set_sum_order(int o) {
        sum_order = o;
        refresh_dist();
    }
};

// Host-resident brute-force index, rows [f32 x dim | u64 label] (bruteforce.hpp:17-29, :54-65).
class HostBruteforce {
public:
    int metric = 0;
    size_t dim = 0, capacity = 0, count = 0, data_size = 0, stride = 0;
    char* data = nullptr;
    uint64_t version = 0;
    std::mutex lock;
    std::unordered_map<uint64_t, size_t> label_to_index;

    HostBruteforce(int metric, size_t dim, size_t capacity);
    HostBruteforce(const std::string& path, int metric, size_t dim);
    ~HostBruteforce();
    void add(const float* vec, uint64_t label);
    void remove(uint64_t label);
    void save(const std::string& path) const;
};

}  // namespace vdb
