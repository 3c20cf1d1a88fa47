// bf_internal.hpp — shared between bf_search.hip (exact scan, compaction) and bf_mfma.hip
// (MFMA candidate pass).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace vdb {

struct CompactArgs {
    uint32_t nq, k, cap;
    uint32_t* count;
    uint32_t* tau;
    uint64_t* entries;
    const uint64_t* labels;  // [n] label of a row
    int32_t final_pass;
    float* out_dist;         // [nq][k]   (final pass)
    uint64_t* out_label;     // [nq][k]
    const uint32_t* qmap;    // nullable: slot -> output row
    const uint32_t* nq_dev;  // nullable: live slots
    const float* cert;       // nullable (final pass): the result of slot q is proven exact iff its
                             // k-th distance is < cert[q]; otherwise q is appended to redo
    uint32_t* redo;          // [0] = count, [1 + i] = query ids to re-run exactly
    const uint32_t* tau_floor;  // nullable [nq]: the threshold a query keeps while it has fewer than k results (filtered scan)
    uint32_t* overflow;      // nullable [nq]: set when a buffer received more than `cap` candidates
                             // (some were lost): such a slot is never proven exact
};


// One wave per query slot: sort the candidate buffer, keep the k smallest (dist, label);
// intermediate passes publish tau, the final pass writes the output rows.
hipError_t enqueue_bf_compact(const CompactArgs& c, hipStream_t stream);

}  // namespace vdb
