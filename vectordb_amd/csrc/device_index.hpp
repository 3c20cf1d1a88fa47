// device_index.hpp — the HBM-resident form of an HNSW index (DESIGN.md "Data layout").
// Shared by the host-side uploader (capi.cpp) and the gfx950 kernels (hnsw_search.hip).
#pragma once
#include <cstdint>

namespace vdb {

constexpr uint32_t kNoId = 0xFFFFFFFFu;  // empty link slot / empty hash slot / "no upper block"

// All pointers are device pointers.  Node ids are the reference's internal ids.
struct DevGraph {
    const float* vecs;            // [n][vstride]   row-major f32, rows 128-B aligned, tail zero
    const uint32_t* links0;       // [n][l0stride]  layer-0 neighbour ids in list order, unused = kNoId
    const uint32_t* upper_row;    // [n]            first row of this node's upper lists in upper_links, or kNoId
    const uint32_t* upper_links;  // [rows][ustride] one row per (node, level>=1), unused = kNoId
    const uint64_t* labels;       // [n]            external labels
    uint32_t n;
    uint32_t dim;
    uint32_t vstride;   // floats per vector row  (multiple of 32)
    uint32_t l0stride;  // u32 per layer-0 row    (maxM0 rounded up to 32)
    uint32_t ustride;   // u32 per upper row      (maxM rounded up to 16)
    uint32_t entry;     // entry_id_
    int32_t max_level;  // max_level_
    int32_t metric;     // 0 L2, 1 IP
};

// out_stats layout per query (u32 x 4)
enum { kStatNdis = 0, kStatHops0 = 1, kStatHopsUpper = 2, kStatFound = 3 };
constexpr uint32_t kFoundOverflow = 0xFFFFFFFFu;  // visited table overflowed: query re-run by the fallback

}  // namespace vdb
