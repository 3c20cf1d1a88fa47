// launch.hpp — host-callable launchers of the gfx950 kernels (defined in the .hip files).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "device_index.hpp"

namespace vdb {

// Limits of the traversal kernels (checked by the launcher, reported as VDB_ERR_UNSUPPORTED).
constexpr uint32_t kMaxEf = 256;           // sorted key array: 4 registers per lane, 8 in the large configuration
constexpr uint32_t kMaxVecStride = 1024;   // floats per row (16 chunks of 64)
constexpr uint32_t kMaxL0Stride = 64;      // layer-0 list slots (maxM0 <= 64, i.e. M <= 32)
constexpr uint32_t kLargeBlocks = 128;     // resident waves of the large configuration (one bitmap each)

struct HnswSearchWork {
    uint32_t* redo = nullptr;     // [1 + nq] u32; [0] is zeroed by the launcher
    uint32_t* bitmaps = nullptr;  // [kLargeBlocks][bitmap_words]
    uint32_t bitmap_words = 0;
};

// Enqueues the search of nq queries on `stream` (two launches: fast configuration, then the
// large configuration over the queries the first one gave up on).  Returns 0, or a
// vdb_status with `err` set.
int launch_hnsw_search(const DevGraph& g, const float* d_queries, uint32_t nq, uint32_t k, uint32_t ef,
                       float* d_out_dist, uint64_t* d_out_label, uint32_t* d_out_stats, const HnswSearchWork& work,
                       hipStream_t stream, std::string* err);

// ---- brute force (bf_search.hip) ---------------------------------------------------------
constexpr uint32_t kBfCap = 4096;   // candidate entries per query between two compactions
constexpr uint32_t kBfMaxK = 1024;

// Bytes of scratch the exact scan needs for nq queries.
size_t bf_exact_workspace_bytes(uint32_t nq);

// Exact k-NN of nq queries over base[n][vstride] (rows zero-padded beyond dim): enqueues the
// slab sweep on `stream`.  Outputs ascending (dist, label), padded with (+inf, UINT64_MAX).
int launch_bf_exact(const float* d_base, const uint64_t* d_labels, uint32_t n, uint32_t dim, uint32_t vstride,
                    int metric, const float* d_queries, uint32_t nq, uint32_t k, float* d_out_dist,
                    uint64_t* d_out_label, void* workspace, hipStream_t stream, std::string* err);

// G ascending per-shard lists [G][nq][k] -> [nq][k], the k smallest (dist, label).
int launch_merge_topk(const float* d_dist, const uint64_t* d_label, uint32_t n_lists, uint32_t nq, uint32_t k,
                      float* d_out_dist, uint64_t* d_out_label, hipStream_t stream, std::string* err);

}  // namespace vdb
