// launch.hpp — host-callable launchers of the gfx950 kernels (defined in the .hip files).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "device_index.hpp"

namespace vdb {

// Limits of the traversal kernels (checked by the launcher, reported as VDB_ERR_UNSUPPORTED).
constexpr uint32_t kMaxEf = 508;           // sorted key array: up to 8 registers per lane (512 keys), 16 in the large configuration
constexpr uint32_t kMaxVecStride = 1024;   // floats per row (16 chunks of 64)
constexpr uint32_t kMaxL0Stride = 64;      // layer-0 list slots (maxM0 <= 64, i.e. M <= 32)
constexpr uint32_t kGlobalHashBits = 15;     // 32 768 entries (128 KB) per resident wave: 24 576 visited nodes
constexpr uint32_t kGlobalHashBlocks = 4096;  // resident waves of the global-hash configuration (512 MB of tables)
constexpr uint32_t kLargeBlocks = 128;     // resident waves of the large configuration (one n-bit bitmap each), at least; as many
constexpr uint32_t kLargeBlocksMax = 2048; // more, up to 2048 (8 per CU by its registers), as fit kLargeBitmapBytes
constexpr size_t kLargeBitmapBytes = (size_t)256 << 20;
constexpr uint32_t kSpillBits = 14;        // 16 384 entries (64 KB) per spill table: 12 288 more visited nodes
constexpr uint32_t kSpillTablesMin = 512;  // spill tables per launch: nq within [512, 4096] (32-256 MB).  A table goes back when its
constexpr uint32_t kSpillTablesMax = 4096; // query ends, and a GPU holds at most 4096 of these searches at once (16 per CU by
                                           // their LDS): with nq >= 4096 a query always finds one
constexpr uint32_t kFiltLevels = 5;        // filtered searches: the first level's candidate buffer in LDS holds 256 << level keys
constexpr uint32_t kRedoKeyOverflow = 4;   // header word: first-level queries whose KEY ARRAY overflowed (filtered searches)
constexpr uint32_t kRedoVisitedSum = 5;    // header word: sum over the first-level queries of (visited nodes + 8) / 16, launches with a report
constexpr uint32_t kRedoSpillMap = 8;      // word offset in the work list's header of the bitmap of spill tables in use
constexpr uint32_t kRedoHeader = kRedoSpillMap + kSpillTablesMax / 32;  // words ahead of the re-run list

// The no-allocation entry point (vdb_search_batch_device_noalloc): limits and the layout of its static
// scratch (noalloc_scratch.hip): [work list: kRedoHeader + kNoAllocMaxQueries words | pad to 256 B |
// kNoAllocLargeBlocks visited bitmaps of kNoAllocMaxElements bits].
constexpr uint32_t kNoAllocMaxQueries = 32768;
constexpr uint32_t kNoAllocMaxElements = 1u << 23;
constexpr uint32_t kNoAllocLargeBlocks = 4;
constexpr size_t kNoAllocBitmapOffset = (((size_t)kRedoHeader + kNoAllocMaxQueries) * 4 + 255) / 256 * 256;
constexpr size_t kNoAllocScratchBytes = kNoAllocBitmapOffset + (size_t)kNoAllocLargeBlocks * (kNoAllocMaxElements / 32) * 4;
// Address of that scratch on `device` (which must be current); asked from the runtime once per device.
int noalloc_scratch(int device, void** out, std::string* err);

// The caller's rule for HnswSearchWork::hbits_bias, from a launch's report (`outgrew` of `nq` queries filled their LDS
// table and went on in a spill table or in the large configuration).  The first step, 2^11 -> 2^12 entries, costs a
// third of the residency and is taken when more than 2 % outgrew the table; the later ones halve it again (or move
// the visited set to HBM, hnsw_search.hip) and are taken only when more than a quarter still do - the rest is what
// the spill tables are for (20 M x 128, ef 100, 10 000 queries: 2.2 ms at 2^12 with the long searches spilling, 3.1 ms
// at 2^13 with none).
inline uint32_t next_hbits_bias(uint32_t bias, uint32_t outgrew, uint32_t nq) {
    if (bias == 0) return (uint64_t)outgrew * 50u > nq ? 1u : 0u;
    return (uint64_t)outgrew * 4u > nq && bias < 3u ? bias + 1u : bias;
}

struct HnswSearchWork {
    uint32_t* ghash = nullptr;    // nullable: [kGlobalHashBlocks << kGlobalHashBits] u32, visited tables in HBM (large ef)
    uint32_t* report = nullptr;   // nullable, host-mapped [4]: {queries that outgrew the LDS visited table, nq, queries whose key
                                  // array overflowed at the first level, sum of the queries' visited nodes / 16} of this launch
    uint32_t filt_level = 0;      // filtered searches: the first level's candidate buffer holds 256 << filt_level keys (< kFiltLevels)
    uint32_t hbits_bias = 0;      // added to log2 of the visited-set size the launcher picks
    uint32_t vis_mean = 0;        // nodes a search at this ef visits on average, from earlier launches' reports (0 = not known):
                                  // sizes the LDS table of the launches that have spill tables behind it (launch_cfg)
    uint32_t* redo = nullptr;     // [kRedoHeader + nq] u32; the header ([0] re-run count, [1] unanswerable filtered queries,
                                  // [2] spill tables taken, [3] queries beyond heap_cap, [kRedoKeyOverflow], [kRedoVisitedSum], [kRedoSpillMap..]
                                  // spill tables in use) is zeroed by the launcher
    uint32_t spill_tables = 0;    // a multiple of 32, at most kSpillTablesMax
    uint32_t* spill = nullptr;    // nullable: [spill_tables << kSpillBits] u32, second-level visited tables in HBM for the
                                  // queries that fill their LDS table (any contents: a table is cleared when taken)
    uint32_t* bitmaps = nullptr;  // [large_blocks][bitmap_words]
    uint32_t bitmap_words = 0;
    uint32_t large_blocks = kLargeBlocks;  // resident waves of the large configuration (one bitmap each)
    // searchStopConditionClosest (hnsw_core.hpp:1162-1213) with the epsilon-radius condition: stop once the best open
    // candidate lies beyond eps and at least eps_min results are held; only results within eps are returned.  Filtered
    // launches only (d_allow given; k = ef = the condition's max_num_candidates).  eps_ord = ordered u32 of epsilon,
    // 0xFFFFFFFF = no stop condition.
    uint32_t eps_ord = 0xFFFFFFFFu;
    uint32_t eps_min = 0;
    // Filtered searches whose live candidates outgrow even the large configuration's key array (a filter that rejects
    // almost everything) are run once more by hnsw_search_unbounded_kernel: candidates in a heap in HBM, no limit but
    // heap_cap.  fail_list [nq] u32 (query ids, counted in redo[1]); heaps [heap_blocks][heap_cap] u64; heap_blocks <=
    // large_blocks (it reuses their bitmaps).  Without them such queries are reported (redo[1], NaN rows); with them
    // redo[3] counts the queries that outgrew heap_cap as well.
    uint32_t* fail_list = nullptr;
    uint64_t* heaps = nullptr;
    uint32_t heap_cap = 0, heap_blocks = 0;
};

// Enqueues the search of nq queries on `stream` (two launches: fast configuration, then the
// large configuration over the queries the first one gave up on).  Returns 0, or a
// vdb_status with `err` set.
// d_allow: nullptr = bare search; else bit i of word i/32 says node i may enter the results (not
// deleted and accepted by the filter): the non-bare path of hnsw_core.hpp:1141-1148.
int launch_hnsw_search(const DevGraph& g, const float* d_queries, uint32_t nq, uint32_t k, uint32_t ef,
                       const uint32_t* d_allow, float* d_out_dist, uint64_t* d_out_label, uint32_t* d_out_stats,
                       const HnswSearchWork& work, hipStream_t stream, std::string* err);
// out[i] = a[i] & b[i] over `words` u32 (delete flags AND a caller's filter)
int launch_and_bitmaps(const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t words, hipStream_t stream);

// ---- layer-0 graph construction (hnsw_build.hip) --------------------------------------------
struct BuildWork {
    uint32_t *key_in = nullptr, *val_in = nullptr, *key_out = nullptr, *val_out = nullptr;  // [max batch * M] each
    void* sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
};
size_t build_sort_temp_bytes(uint32_t max_pairs);
// out[i][0..dim) = vecs[ids[i]][0..dim)
int launch_gather_rows(const float* d_vecs, uint32_t vstride, uint32_t dim, const uint32_t* d_ids, uint32_t n, float* d_out,
                       hipStream_t stream);
// One insertion round after the search: neighbour selection for the nb new points (candidates =
// the search's output with identity labels), their lists, and the reverse links.  See hnsw_build.hip.
int launch_build_round(const DevGraph& g, uint32_t* d_links0, const uint64_t* d_cand_label, const float* d_cand_dist,
                       const uint32_t* d_new_ids, uint32_t nb, uint32_t efc, uint32_t M, uint32_t maxM0, const BuildWork& w,
                       hipStream_t stream, std::string* err);

// After the last round: every element without an inbound link on layer 0 gets one from the nearest element of its own
// list that can take it (the reference's checkIntegrity demands one, hnsw_core.hpp:1235-1241).  Synchronises `stream`.
// found / fixed: elements without an inbound link before / given one.
int build_repair_orphans(uint32_t* d_links0, uint32_t n, uint32_t l0stride, uint32_t maxM0, hipStream_t stream,
                         uint32_t* found, uint32_t* fixed, std::string* err);

// ---- brute force (bf_search.hip) ---------------------------------------------------------
constexpr uint32_t kBfCap = 4096;   // candidate entries per query between two compactions
constexpr uint32_t kBfMaxK = 1024;

// Bytes of scratch the exact scan needs for nq queries.
size_t bf_exact_workspace_bytes(uint32_t nq);

// d_allow: nullable bitmap over rows (a BaseFilterFunctor's verdicts): the reference's filtered scan, bruteforce.hpp:154-181.
// Exact k-NN of nq queries over base[n][vstride] (rows zero-padded beyond dim): enqueues the
// slab sweep on `stream`.  Outputs ascending (dist, label), padded with (+inf, UINT64_MAX).
int launch_bf_exact(const float* d_base, const uint64_t* d_labels, uint32_t n, uint32_t dim, uint32_t vstride,
                    int metric, const float* d_queries, uint32_t nq, uint32_t k, float* d_out_dist,
                    uint64_t* d_out_label, void* workspace, hipStream_t stream, std::string* err,
                    const uint32_t* d_qmap = nullptr, const uint32_t* d_nq_live = nullptr, const uint32_t* d_allow = nullptr);

// ---- MFMA candidate pass (bf_mfma.hip) ----------------------------------------------------
constexpr uint32_t kBfMfmaCap = 2048;     // candidate entries per query in the MFMA path
constexpr uint32_t kBfMaxSlab = 131072;   // base rows per GEMM launch at most

struct BfDeviceData {  // device pointers of an uploaded brute-force index
    const float* raw = nullptr;     // [n][vstride] original rows, zero-padded
    const float* xc = nullptr;      // [n][vstride] centred rows x - mu
    const void* xb = nullptr;       // [n][kp64]    centred rows in bf16
    const float* nx = nullptr;      // [n]          ||x - mu||^2
    const float* mu = nullptr;      // [kp64]       mean vector
    const float* stats = nullptr;   // [0] max ||x - mu||, [1] ||mu||
    const uint64_t* labels = nullptr;
    uint32_t n = 0, dim = 0, vstride = 0, kp64 = 0;
    int metric = 0;
};

size_t bf_prepare_scratch_bytes(uint32_t vstride);
int launch_bf_prepare(const float* d_raw, uint32_t n, uint32_t dim, uint32_t vstride, uint32_t kp64, float* d_xc,
                      void* d_xb, float* d_nx, float* d_mu, float* d_stats, void* d_scratch, hipStream_t stream,
                      std::string* err);
size_t bf_mfma_workspace_bytes(uint32_t nq, uint32_t vstride, uint32_t kp64);
// precision: 0 = f32 MFMA + exact re-rank + certificate (+ exact re-run of unproven queries),
//            1 = bf16 MFMA only, 2 = bf16 MFMA + exact re-rank (no certificate).
int launch_bf_mfma(const BfDeviceData& d, int precision, const float* d_queries, uint32_t nq, uint32_t k, float* d_out_dist,
                   uint64_t* d_out_label, void* workspace, hipStream_t stream, std::string* err);

// G ascending per-shard lists [G][nq][k] -> [nq][k], the k smallest (dist, label).
int launch_merge_topk(const float* d_dist, const uint64_t* d_label, uint32_t n_lists, uint32_t nq, uint32_t k,
                      float* d_out_dist, uint64_t* d_out_label, hipStream_t stream, std::string* err);

}  // namespace vdb
