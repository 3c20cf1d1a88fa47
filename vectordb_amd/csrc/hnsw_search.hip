// hnsw_search.hip — gfx950 kernels for HierarchicalNSW::searchKnn.
//
// Reference path restated here (paths relative to the reference's src/hnsw/hnsw_scratch/):
//   searchKnn                      hnsw_core.hpp:1107-1159
//     upper-layer greedy descent   :1111-1138
//     searchBaseLayerST<bare=true> called :1143-1147, defined nowhere; restated from the
//                                  construction-time loop :293-356
//     epilogue (trim to k, labels) :1150-1158 + SearchKNNCloserFirst hnswlib.hpp:87-99
//   distances                      space_l2.hpp:1-14, space_ip.hpp:3-18 (summation order
//                                  "tree64", DESIGN.md "Numerics")
//   visited set                    visited_list_pool.hpp:36-58 -> exact hash set in LDS
//
// One wavefront (64 lanes) runs one query from entry point to output.
//
//  * Distances.  A 16-lane group scores one neighbour: lane t of the group owns the
//    virtual lanes 4t..4t+3 of the tree64 order, i.e. elements 64j+4t..64j+4t+3 of the
//    vector for every 64-element chunk j, fetched as one 16-byte load per chunk (16 lanes
//    x 16 B = 256 contiguous bytes = two full 128-B lines).  A wave scores four
//    neighbours per instruction stream and keeps the loads of up to 32 neighbours in
//    flight before the first FMA.  The 64 partial sums are folded by the pairwise tree:
//    two levels inside the lane, four DPP levels (xor 1, xor 2, half-mirror, mirror).
//  * Result set and candidate set are ONE sorted array of 64-bit keys
//    (ordered(dist) << 32 | id << 1 | expanded) striped over the lanes' registers, R
//    registers per lane (capacity 64 R >= ef).  Inserting is a wave-wide shift by one lane
//    (DPP wave_shr) - no heap.  The next candidate is the first key whose `expanded` bit is
//    clear.  Why this equals the reference's two heaps, ties included: DESIGN.md
//    "Traversal state".
//  * Visited set: open-addressing hash of node ids in LDS, inserted with ds_cmpst
//    (atomicCAS) so that "test and mark" of all neighbours of a node is one parallel step.
//
// Queries whose state outgrows the fast configuration (hash fill limit, or equal-distance
// keys pushed off the end of the array) are appended to a work list and re-run by the
// same kernel body in its exact large configuration (2R registers, visited bitmap in HBM).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "device_index.hpp"
#include "launch.hpp"
#include "traversal_common.hpp"

// Compiled in four parts (Makefile: -DVDB_PART=0..3, one object each, in parallel - as one translation unit the
// instantiations take seven minutes): part 0 = bare searches of rows up to 128 floats + the entry point, 1 = bare, longer
// rows, 2 / 3 = the same two for filtered searches.  Without VDB_PART the file is one translation unit.
#ifndef VDB_PART
#define VDB_PART (-1)
#endif

namespace vdb {

namespace detail {
using namespace dev;
struct SearchArgs {
    DevGraph g;
    const float* queries;  // [nq][dim]
    uint32_t nq, k, ef, hbits;
    uint32_t lds_entries;  // LdsVisited::size of this launch
    uint32_t lds_mod;      // LdsVisited::mod
    uint32_t stage_off;    // word offset in the block's LDS of the 64 R keys that KeyArray::merge stages (behind the LDS
                           // table; 0 in the configurations without one, where it shares the epilogue's space)
    float* out_dist;       // [nq][k]
    uint64_t* out_label;   // [nq][k]
    uint32_t* out_stats;   // nullable [nq][4]
    const uint32_t* allow; // filtered search: bit i of word i/32 = node i may enter the results
    uint32_t* redo;        // [0] = count, [1] = queries that could not be answered, [2] = spill tables taken,
                           // [kRedoHeader + i] = query ids to re-run
    uint32_t* spill;       // nullable: [spill_tables << kSpillBits], see LdsVisited
    uint32_t spill_tables;
    uint32_t* ghash;       // global-hash configuration: [gridDim.x][1 << kGlobalHashBits]
    uint32_t* bitmaps;     // large configuration: [gridDim.x][bitmap_words]
    uint32_t* report;      // nullable, host-mapped: [0] = queries the fast configuration gave up on, [1] = nq
    uint32_t bitmap_words;
    uint32_t large_blocks; // grid of the large configuration (host side only)
    uint32_t cb_cap;       // first-level filtered launches: keys the candidate buffer holds (at stage_off; see SPLIT)
    uint32_t eps_ord;      // FILT only: the epsilon stop condition (HnswSearchWork), kEmptyWord = none
    uint32_t eps_min;
    uint32_t* fail_list;   // nullable: queries the large configuration could not answer, for hnsw_search_unbounded_kernel
    uint32_t filt_level;   // host side only: see HnswSearchWork
    uint32_t vis_mean;     // HnswSearchWork::vis_mean
    uint64_t* heaps;       // [heap_blocks][heap_cap]
    uint32_t heap_cap, heap_blocks;
};

// one part's dispatch over row length and ef (the parts it does not hold answer 5)
int launch_search_part0(const SearchArgs& a, uint32_t nch, hipStream_t stream, std::string* err);
int launch_search_part1(const SearchArgs& a, uint32_t nch, hipStream_t stream, std::string* err);
int launch_search_part2(const SearchArgs& a, uint32_t nch, hipStream_t stream, std::string* err);
int launch_search_part3(const SearchArgs& a, uint32_t nch, hipStream_t stream, std::string* err);
}  // namespace detail

namespace {

using namespace dev;
using detail::SearchArgs;

// ------------------------------------------------------------------ visited set

// SPILL = a second level in HBM may be taken; without it the code is the plain LDS hash
template <bool SPILL>
struct LdsVisited {
    uint32_t* tab;
    uint32_t size;  // entries, a multiple of 4 (not a power of two: the launcher gives a query the share of the
                    // CU's LDS that its residency leaves, see launch_cfg)
    uint32_t mod;   // the largest prime <= size: the slots in use.  Double hashing (any step is coprime to a prime): at the
                    // 75 % fill at which a table is frozen a lookup of an id that is not there walks 4 slots on average,
                    // where linear probing walks 8.5 - and every probe of a spilled search is such a lookup
    // Second level, in HBM, taken when the LDS table reaches its fill limit (take_spill): the LDS
    // table is then frozen (looked up, never inserted into) and new ids go to the spill table, so the
    // set stays exact and the search carries on instead of starting again in the large configuration.
    uint32_t* spill;
    __device__ __forceinline__ void clear(int lane) {
        for (uint32_t i = lane * 4; i < size; i += 256) *reinterpret_cast<uint4*>(tab + i) = uint4{~0u, ~0u, ~0u, ~0u};
    }
    // true when `id` was not in the set (and inserts it)
    __device__ __forceinline__ bool test_and_set(uint32_t id) {
        uint32_t s = __umulhi(id * 0x9E3779B1u, mod);  // multiplicative hash onto [0, mod)
        const uint32_t step = 1u + __umulhi(id * 0x85EBCA6Bu, mod - 1u);  // in [1, mod)
        if (!SPILL || spill == nullptr) {  // (wave-uniform)
            uint32_t old = atomicCAS(&tab[s], kEmptyWord, id);  // almost always decides
            while (old != kEmptyWord && old != id) {
                s += step;
                if (s >= mod) s -= mod;
                old = atomicCAS(&tab[s], kEmptyWord, id);
            }
            return old == kEmptyWord;
        }
        uint32_t v = tab[s];  // frozen at 75 % fill or less: the probe ends
        while (v != kEmptyWord && v != id) {
            s += step;
            if (s >= mod) s -= mod;
            v = tab[s];
        }
        if (v == id) return false;
        constexpr uint32_t gmask = (1u << kSpillBits) - 1u;
        s = (id * 0x9E3779B1u) >> (32u - kSpillBits);
        uint32_t old = atomicCAS(&spill[s], kEmptyWord, id);
        while (old != kEmptyWord && old != id) {
            s = (s + 1u) & gmask;
            old = atomicCAS(&spill[s], kEmptyWord, id);
        }
        return old == kEmptyWord;
    }
    __device__ __forceinline__ uint32_t limit() const { return mod - (mod >> 2); }  // 75 % fill
    // Called by the whole wave when the next hop could pass the limit; `map` = the launch's bitmap of tables in use
    // (`tables` bits, a multiple of 32), `taken` = its counter of queries that took one.  Returns the new limit, or 0
    // when there is no table to take (none configured, all in use, or this query has one already).  The table goes
    // back with release_spill when the query ends: the tables a launch needs are the queries resident at once, not
    // the queries that spill (round 1 handed each table out once: a batch whose every query spilled - 10 000 queries
    // at ef 100 on a 20 M-row graph, 625 tables - ran 30 ms instead of 3, nine tenths of it re-run by the large
    // configuration).
    __device__ __forceinline__ uint32_t take_spill(uint32_t* pool, uint32_t tables, uint32_t* map, uint32_t* taken, int lane) {
        if (!SPILL || pool == nullptr || spill != nullptr) return 0u;
        uint32_t t = kEmptyWord;
        if (lane == 0) {
            const uint32_t words = tables >> 5;
            uint32_t w = __umulhi(blockIdx.x * 0x9E3779B1u, words);  // (spread the first probes over the words)
            for (uint32_t i = 0; i < words && t == kEmptyWord; ++i) {
                uint32_t cur = __hip_atomic_load(&map[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (cur != 0xFFFFFFFFu) {
                    const uint32_t bit = 1u << __builtin_ctz(~cur);
                    cur = atomicOr(&map[w], bit);
                    if ((cur & bit) == 0u) {
                        t = (w << 5) + (uint32_t)__builtin_ctz(bit);
                        break;
                    }
                }
                w = w + 1u == words ? 0u : w + 1u;
            }
            if (t != kEmptyWord) atomicAdd(taken, 1u);
        }
        t = readlane(t, 0);
        if (t == kEmptyWord) return 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // behind the last owner's release
        spill = pool + ((size_t)t << kSpillBits);
        for (uint32_t i = lane * 4; i < (1u << kSpillBits); i += 256)
            *reinterpret_cast<uint4*>(spill + i) = uint4{~0u, ~0u, ~0u, ~0u};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the stores reach L2 ahead of the other lanes' atomics
        return limit() + ((3u << kSpillBits) >> 2);
    }
    __device__ __forceinline__ void release_spill(uint32_t* pool, uint32_t* map, int lane) {
        if (!SPILL || spill == nullptr) return;
        const uint32_t t = (uint32_t)((size_t)(spill - pool) >> kSpillBits);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this query's probes are done before the table changes hands
        if (lane == 0) atomicAnd(&map[t >> 5], ~(1u << (t & 31u)));
        spill = nullptr;
    }
};

// The same hash in HBM, one table per resident wave: for searches that visit thousands of nodes (large
// ef on large graphs, the builder at ef_construction), where an LDS table of 32-64 KB would leave two
// to five queries resident per CU.  A probe costs an L2 round trip instead of an LDS one; residency is
// then bounded by registers only.
struct GlobalHashVisited {
    uint32_t* tab;
    uint32_t hbits;
    __device__ __forceinline__ void clear(int lane) {
        const uint32_t n = 1u << hbits;
        for (uint32_t i = lane * 4; i < n; i += 256) *reinterpret_cast<uint4*>(tab + i) = uint4{~0u, ~0u, ~0u, ~0u};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the stores reach L2 ahead of the other lanes' atomics
    }
    __device__ __forceinline__ bool test_and_set(uint32_t id) {
        const uint32_t mask = (1u << hbits) - 1u;
        uint32_t s = (id * 0x9E3779B1u) >> (32u - hbits);
        uint32_t old = atomicCAS(&tab[s], kEmptyWord, id);
        while (old != kEmptyWord && old != id) {
            s = (s + 1u) & mask;
            old = atomicCAS(&tab[s], kEmptyWord, id);
        }
        return old == kEmptyWord;
    }
    __device__ __forceinline__ uint32_t limit() const { return (3u << hbits) >> 2; }
};

struct BitmapVisited {  // exact for any number of visited nodes; one n-bit map in HBM per resident wave
    uint32_t* bits;
    uint32_t words;
    __device__ __forceinline__ void clear(int lane) {
        for (uint32_t i = lane; i < words; i += 64) bits[i] = 0u;
    }
    __device__ __forceinline__ bool test_and_set(uint32_t id) {
        const uint32_t bit = 1u << (id & 31u);
        return (atomicOr(&bits[id >> 5], bit) & bit) == 0u;
    }
    __device__ __forceinline__ uint32_t limit() const { return 0xFFFFFFFFu; }
};

// ------------------------------------------------------------------ sorted key array

// Entry i of the array lives in register i/64 of lane i%64.  EMPTY = all ones.
// Low word of a key: id << S | flags.  S = 1: bit 0 = expanded.  S = 2 (filtered search): bit 1 =
// the node is not allowed into the results (deleted, or rejected by the filter), bit 0 = expanded.
template <int R, int S>
struct KeyArray {
    uint32_t hi[R], lo[R];

    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int r = 0; r < R; ++r) hi[r] = lo[r] = kEmptyWord;
    }
    __device__ __forceinline__ uint32_t hi_at(uint32_t idx) const {  // idx wave-uniform
        uint32_t v = kEmptyWord;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if ((idx >> 6) == (uint32_t)r) v = readlane(hi[r], (int)(idx & 63u));
        return v;
    }
    // Insert (khi, klo); keys are distinct.  (prev_hi, prev_lo) = this lane's last-register key
    // before the shift: in lane 63 that is the key that fell off the end.
    __device__ __forceinline__ void insert(uint32_t khi, uint32_t klo, int lane, uint32_t& prev_hi, uint32_t& prev_lo) {
        prev_hi = hi[R - 1];
        prev_lo = lo[R - 1];
        uint32_t uh[R], ul[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uh[r] = dpp_u32<kDppWaveShr1>(0u, hi[r]);
            ul[r] = dpp_u32<kDppWaveShr1>(0u, lo[r]);
            if (r > 0) {
                const uint32_t ph = readlane(hi[r - 1], 63), pl = readlane(lo[r - 1], 63);
                if (lane == 0) {
                    uh[r] = ph;
                    ul[r] = pl;
                }
            }
        }
        const uint64_t key = ((uint64_t)khi << 32) | klo;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t up = ((uint64_t)uh[r] << 32) | ul[r];
            const uint64_t my = ((uint64_t)hi[r] << 32) | lo[r];
            const uint64_t nv = up > key ? up : (my > key ? key : my);
            hi[r] = (uint32_t)(nv >> 32);
            lo[r] = (uint32_t)nv;
        }
    }
    __device__ __forceinline__ uint64_t key(int r) const { return ((uint64_t)hi[r] << 32) | lo[r]; }
    // Entries at index >= ef survive only while their distance equals the bound.
    __device__ __forceinline__ void trim(uint32_t ef, uint32_t bound_hi, int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t idx = (uint32_t)(r * 64 + lane);
            if (idx >= ef && hi[r] != bound_hi) hi[r] = lo[r] = kEmptyWord;
        }
    }
    // First key with the expanded bit clear, not marked: (id, ordered distance).
    // (no early exit from the register loops below: a loop left in its middle is not unrolled,
    // and lo[] indexed by a run-time r moves to scratch memory)
    __device__ __forceinline__ bool peek_candidate(uint32_t& id, uint32_t& dhi) const {
        bool found = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t open = ballot((lo[r] & 1u) == 0u);
            if (!found && open) {
                const int p = __builtin_ctzll(open);
                id = readlane(lo[r], p) >> S;
                dhi = readlane(hi[r], p);
                found = true;
            }
        }
        return found;
    }
    // First key with the expanded bit clear: returns its node id and sets the bit.
    __device__ __forceinline__ bool pop_candidate(int lane, uint32_t& id) {
        bool found = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t open = ballot((lo[r] & 1u) == 0u);
            if (!found && open) {
                const int p = __builtin_ctzll(open);
                id = readlane(lo[r], p) >> S;
                if (lane == p) lo[r] |= 1u;
                found = true;
            }
        }
        return found;
    }

    __device__ __forceinline__ uint32_t count_valid() const {  // entries held
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) c += (uint32_t)__builtin_popcountll(ballot(!(hi[r] == kEmptyWord && lo[r] == kEmptyWord)));
        return c;
    }

    // ---- filtered search (S == 2) only
    // Entry `p` (wave-uniform) leaves the array; everything behind it moves one place forward.
    __device__ __forceinline__ void erase(uint32_t p, int lane) {
        uint32_t dh[R], dl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dh[r] = dpp_u32<kDppWaveShl1>(kEmptyWord, hi[r]);
            dl[r] = dpp_u32<kDppWaveShl1>(kEmptyWord, lo[r]);
            const uint32_t nh = r + 1 < R ? readlane(hi[r + 1 < R ? r + 1 : r], 0) : kEmptyWord;
            const uint32_t nl = r + 1 < R ? readlane(lo[r + 1 < R ? r + 1 : r], 0) : kEmptyWord;
            if (lane == 63) {
                dh[r] = nh;
                dl[r] = nl;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((uint32_t)(r * 64 + lane) >= p) {
                hi[r] = dh[r];
                lo[r] = dl[r];
            }
        }
    }
    // First key with the expanded bit clear.  An allowed node stays (it is a result) with the bit
    // set; a node that is not allowed was only ever a candidate and leaves the array.
    __device__ __forceinline__ bool pop_candidate_filtered(int lane, uint32_t& id) {
        bool found = false;
        uint32_t gone = kEmptyWord;  // index of the entry to erase
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t open = ballot((lo[r] & 1u) == 0u);
            if (!found && open) {
                const int p = __builtin_ctzll(open);
                const uint32_t w = readlane(lo[r], p);
                id = w >> S;
                if (w & 2u) gone = (uint32_t)(r * 64 + p);
                else if (lane == p) lo[r] |= 1u;
                found = true;
            }
        }
        if (gone != kEmptyWord) erase(gone, lane);
        return found;
    }
    // The result set is the first ef ALLOWED entries.  Full: (bhi, bpos) = distance and index of the
    // ef-th of them (lowerBound of hnsw_core.hpp:348-349); else false.
    __device__ __forceinline__ bool bound_filtered(uint32_t ef, uint32_t& bhi, uint32_t& bpos) const {
        uint32_t base = 0;
        bool full = false;
        bhi = kEmptyWord;
        bpos = kEmptyWord;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = hi[r] != kEmptyWord && (lo[r] & 2u) == 0u;
            const uint64_t m = ballot(ok);
            const uint32_t rank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint64_t hit = ballot(ok && rank == ef - 1u);
            if (hit) {
                const int p = __builtin_ctzll(hit);
                bhi = readlane(hi[r], p);
                bpos = (uint32_t)(r * 64 + p);
                full = true;
            }
            base += (uint32_t)__builtin_popcountll(m);
        }
        return full;
    }
    // Number of allowed entries (the result set holds the first ef of them).
    __device__ __forceinline__ uint32_t count_allowed() const {
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) c += (uint32_t)__builtin_popcountll(ballot(hi[r] != kEmptyWord && (lo[r] & 2u) == 0u));
        return c;
    }
    // Behind the bound entry only keys AT the bound's distance can still be expanded (:316).
    __device__ __forceinline__ void trim_filtered(uint32_t bpos, uint32_t bound_hi, int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t idx = (uint32_t)(r * 64 + lane);
            if (idx > bpos && hi[r] != bound_hi) hi[r] = lo[r] = kEmptyWord;
        }
    }
};

// ------------------------------------------------------------------ the search


// FILT = the non-bare searchBaseLayerST (hnsw_core.hpp:1141-1148 with deleted elements or a filter):
// every admitted node is a candidate, only allowed ones are results (:340-343), the search stops when
// the best candidate is beyond the ef-th RESULT (:316).  Restated on the one sorted array: see
// KeyArray (S == 2) and DESIGN.md "Filtered search".
// VIS: where the visited set lives - 0 hash in LDS (one query per block), 1 hash in HBM (persistent
// blocks walking the batch), 2 exact bitmap in HBM = the large configuration over the re-run list
// (exact unless the key array itself overflows, which is reported per query: found = UINT32_MAX),
// 3 = 0 with a second-level table in HBM for the queries that fill the LDS one (LdsVisited).
// Registers: R <= 2 (ef <= 124, bare) fits 128 VGPRs = four waves per SIMD; left alone the R == 2 body took 129.
// Four waves per SIMD (128 registers) where the body fits them without scratch: one or two key registers, and the larger
// arrays with 8 row chunks of 16 B per lane in flight instead of 16 (launch_r); up to dim 256 (R = 8 at dim 129-256 is 6
// registers short; from dim 257 on the query and one row in flight alone take 64-128 registers: bounding those to 128
// put 13-130 registers into scratch, which round 1 did for R <= 2).
constexpr bool four_waves(int nch, int gpb, int r, int vis, bool filt) {
    if (filt) return vis != 2 && nch * gpb <= 8 && nch <= 2 && r <= 4;  // (first-level filtered launches: 8 row chunks in flight)
    return vis != 2 && nch <= 4 && (r <= 2 || (nch * gpb <= 8 && (nch <= 2 || r <= 5)));
}

// FULL: rows of exactly NCH chunks (score_nodes) - the kernels for 64- and 128-float rows are compiled both ways.
template <int NCH, int GPB, int R, int VIS, bool IS_L2, bool FILT, bool FULL = false>
__global__ __launch_bounds__(64, four_waves(NCH, GPB, R, VIS, FILT) ? 4 : 2) void hnsw_search_kernel(SearchArgs a) {
    // SPLIT (round 3): the first-level filtered launches keep two structures, as the reference keeps two heaps.  The key
    // array holds the ALLOWED nodes only - it is then exactly the bare search's array: results = entries 0..ef-1, bound =
    // entry ef-1, behind it only keys at the bound's distance - and the admitted nodes that are not allowed, which can
    // only ever be candidates, wait unsorted in a buffer in LDS (`cb`: 64-bit keys ordered-distance << 32 | id, with the
    // smallest one known).  The next candidate is the smaller of the array's first open entry and the buffer's minimum;
    // one taken from the buffer leaves it.  Before, such nodes stood in the array too, which then had to hold about
    // ef / selectivity keys: a filter that keeps a tenth at ef 64 needs 640 - beyond the 512 keys of eight registers, so
    // every query went through the large configuration (26 ms per 10 000 queries on the C2 graph, 81 ms at ef 128).
    // The large configuration (VIS == 2) is the one-array design still (S == 2, 1024 keys), and behind it the heap kernel.
    constexpr bool LARGE = VIS == 2;
    constexpr bool SPLIT = FILT && !LARGE;
    constexpr int S = (FILT && !SPLIT) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = lane_id();
    const DevGraph& g = a.g;

    typename std::conditional<LARGE, BitmapVisited, typename std::conditional<VIS == 1, GlobalHashVisited, LdsVisited<VIS == 3>>::type>::type vis;
    if constexpr (LARGE) {
        vis.bits = a.bitmaps + (size_t)blockIdx.x * a.bitmap_words;
        vis.words = a.bitmap_words;
    } else if constexpr (VIS == 1) {
        vis.tab = a.ghash + ((size_t)blockIdx.x << kGlobalHashBits);
        vis.hbits = kGlobalHashBits;
    } else {
        vis.tab = lds;
        vis.size = a.lds_entries;
        vis.mod = a.lds_mod;
        vis.spill = nullptr;
    }

    const uint32_t n_work = LARGE ? a.redo[0] : a.nq;
    if constexpr (LARGE) {  // tells the host how the visited-set size chosen for this launch fared
        if (blockIdx.x == 0 && lane == 0 && a.report) {
            const uint32_t spilled = a.redo[2];
            a.report[0] = n_work + spilled;  // the queries that outgrew the LDS table, whichever way they went on
            a.report[1] = a.nq;
            a.report[2] = a.redo[kRedoKeyOverflow];
            a.report[3] = a.redo[kRedoVisitedSum];
        }
    }
#pragma unroll 1
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const uint32_t qi = LARGE ? a.redo[kRedoHeader + w] : w;
        const uint32_t k = a.k, ef = a.ef;
        float* od = a.out_dist + (size_t)qi * k;
        uint64_t* ol = a.out_label + (size_t)qi * k;

        uint32_t ndis = 0, hops0 = 0, hops_up = 0;
        if (g.n == 0) {  // hnsw_core.hpp:1109
            for (uint32_t i = lane; i < k; i += 64) {
                od[i] = __builtin_inff();
                ol[i] = ~0ull;
            }
            if (a.out_stats && lane == 0) {
                uint32_t* st = a.out_stats + (size_t)qi * 4;
                st[0] = st[1] = st[2] = st[3] = 0;
            }
            continue;
        }

        // the query, in the register layout of score_nodes (zero beyond dim)
        f32x4 q[NCH];
        {
            const float* qp = a.queries + (size_t)qi * g.dim;
            const int t = lane & 15;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                float e[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t pos = (uint32_t)(64 * j + 4 * t + c);
                    e[c] = pos < g.dim ? qp[pos] : 0.f;
                }
                q[j] = f32x4{e[0], e[1], e[2], e[3]};
            }
        }

        vis.clear(lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- entry point and greedy descent, hnsw_core.hpp:1111-1138
        uint32_t cur = g.entry;
        uint32_t cur_d = readlane(score_nodes<NCH, GPB, IS_L2, FULL>(g, q, cur, 1u, lane), 0);
        ndis += 1;
        uint32_t cur_row = g.max_level > 0 ? g.upper_row[cur] : 0u;  // first upper list of `cur`
        for (int level = g.max_level; level > 0; --level) {
            bool changed = true;
            while (changed) {
                changed = false;
                const uint32_t row = cur_row + (uint32_t)(level - 1);
                const uint32_t nb = (uint32_t)lane < g.ustride ? g.upper_links[(size_t)row * g.ustride + lane] : kNoId;
                // every neighbour's own list position, fetched beside its vector: no dependent load after the move
                const uint32_t nb_row = nb != kNoId ? g.upper_row[nb] : 0u;
                const uint32_t n = (uint32_t)__builtin_popcountll(ballot(nb != kNoId));
                hops_up += 1;
                ndis += n;
                if (n == 0) break;
                // all n distances are evaluated by the reference too; the scan keeps the
                // first strict minimum below cur_d
                uint32_t d = score_nodes<NCH, GPB, IS_L2, FULL>(g, q, nb, n, lane);
                if ((uint32_t)lane >= n) d = kEmptyWord;
                const uint32_t bd = wave_min_u32(d);
                if (bd < cur_d) {  // first lane holding the minimum == first strict improvement of the scan
                    cur_d = bd;
                    const int bl = __builtin_ctzll(ballot(d == bd));
                    cur = readlane(nb, bl);
                    cur_row = readlane(nb_row, bl);
                    changed = true;
                }
            }
        }

        // ---- layer 0: searchBaseLayerST<bare> (restated from hnsw_core.hpp:293-356)
        KeyArray<R, S> arr;
        arr.clear();
        auto not_allowed = [&](uint32_t id) { return ((a.allow[id >> 5] >> (id & 31u)) & 1u) ^ 1u; };
        uint32_t bound;  // lowerBound once the result set is full, else "no bound"
        bool ep_only_candidate = false;  // (:310 seeds it at "distance" -FLT_MAX: no stop condition fires on it)
        // SPLIT: the candidates that are not allowed (see the head of the kernel), in two parts.  NEAR: at most 64 keys, one
        // per lane in registers (nr_hi = ordered distance, nr_lo = id; all ones = free lane), unsorted - its minimum is two
        // wave-wide reductions away.  FAR: the rest, unsorted in LDS.  Every key in FAR has distance >= tx, so a NEAR
        // minimum below tx is the minimum of both; when NEAR has nothing below tx the smallest keys of FAR are moved over
        // (far_refill), which also drops FAR's dead keys (beyond the bound).  A key enters NEAR when it lies below tx and
        // a lane is free, else FAR (and tx comes down to it if it lay below).  Which keys sit where is a matter of speed
        // only: the candidate taken is always the exact minimum by (distance, id).
        uint64_t* far = reinterpret_cast<uint64_t*>(lds + a.stage_off);
        uint32_t nr_hi = kEmptyWord, nr_lo = kEmptyWord;
        uint32_t far_n = 0, near_n = 0, tx = kEmptyWord, tx_step = 1u << 18;  // (near_n: keys in NEAR)
        auto lds_sync = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        };
        auto lane_rank = [&](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
        // the lanes with `f` hand their key to free lanes of NEAR (the caller has made sure there are enough)
        auto near_put = [&](bool f, uint32_t khi, uint32_t klo) {
            const uint64_t fm = ballot(f);
            const uint32_t take = (uint32_t)__builtin_popcountll(fm);
            const uint32_t r = lane_rank(fm);
            const uint32_t dest = f ? r : take + ((uint32_t)lane - r);  // a full permutation: no two lanes write one destination
            const uint32_t ch = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)(f ? khi : kEmptyWord));
            const uint32_t cl = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)(f ? klo : kEmptyWord));
            const bool is_free = nr_hi == kEmptyWord;
            const uint32_t fr = lane_rank(ballot(is_free));
            const uint32_t gh = bpermute(ch, (int)(fr & 63u)), gl = bpermute(cl, (int)(fr & 63u));
            if (is_free && fr < take) {
                nr_hi = gh;
                nr_lo = gl;
            }
            near_n += take;
        };
        // One pass over FAR: keys below `ntx` go to NEAR while it has free lanes, dead keys (beyond the bound of a full
        // result set) go away, the others move up.  Leaves tx = the exclusive lower bound of what stayed.
        auto far_sweep = [&](uint32_t ntx) {
            uint32_t w = 0, stay_min = kEmptyWord, moved = 0;
            for (uint32_t i0 = 0; i0 < far_n; i0 += 64) {
                const uint32_t i = i0 + (uint32_t)lane;
                const uint64_t kk = i < far_n ? far[i] : ~0ull;
                const uint32_t kh = (uint32_t)(kk >> 32);
                const bool live = i < far_n && !(bound != kEmptyWord && kh > bound);
                const bool elig = live && kh < ntx;
                const uint64_t em = ballot(elig);
                const uint32_t nfree = (uint32_t)__builtin_popcountll(ballot(nr_hi == kEmptyWord));
                const bool go = elig && lane_rank(em) < nfree;
                if (em && nfree) near_put(go, kh, (uint32_t)kk);
                moved += (uint32_t)__builtin_popcountll(ballot(go));
                const bool keep = live && !go;
                const uint64_t km = ballot(keep);
                if (keep && kh < stay_min) stay_min = kh;
                lds_sync();
                if (keep) far[w + lane_rank(km)] = kk;
                w += (uint32_t)__builtin_popcountll(km);
                lds_sync();
            }
            far_n = w;
            tx = wave_min_u32(stay_min);  // (all ones when nothing stayed)
            return moved;
        };
        // NEAR has nothing below tx and FAR is not empty: bring FAR's smallest keys over
        auto far_refill = [&]() {
            if (ballot(nr_hi == kEmptyWord) == 0ull) {  // NEAR is full (of keys at or above tx): back into FAR with it
                if (far_n + 64u > a.cb_cap) return false;
                lds_sync();
                far[far_n + (uint32_t)lane] = ((uint64_t)nr_hi << 32) | nr_lo;
                lds_sync();
                far_n += 64;
                nr_hi = nr_lo = kEmptyWord;
                near_n = 0;
            }
            // FAR's smallest key by (distance, id) goes over first, whatever else does: after a refill the minimum of NEAR
            // is the minimum of both parts (ties in distance included)
            uint32_t bh = kEmptyWord, bl = kEmptyWord, bp = 0;
            for (uint32_t i = (uint32_t)lane; i < far_n; i += 64) {
                const uint64_t kk = far[i];
                const uint32_t kh = (uint32_t)(kk >> 32), kl = (uint32_t)kk;
                if (kh < bh || (kh == bh && kl < bl)) {
                    bh = kh;
                    bl = kl;
                    bp = i;
                }
            }
            const uint32_t m = wave_min_u32(bh);
            const uint32_t ml = wave_min_u32(bh == m ? bl : kEmptyWord);
            const uint32_t pos = readlane(bp, __builtin_ctzll(ballot(bh == m && bl == ml)));
            lds_sync();
            if (lane == 0) far[pos] = far[far_n - 1u];
            lds_sync();
            far_n -= 1;
            if (lane == __builtin_ctzll(ballot(nr_hi == kEmptyWord))) {
                nr_hi = m;
                nr_lo = ml;
            }
            near_n += 1;
            const uint32_t want = m + 1u + tx_step;
            const uint32_t moved = far_sweep(want > m ? want : kEmptyWord);  // (no wrap-around)
            // the window follows what it caught: a quarter to three quarters of NEAR is what a sweep should bring
            if (moved < 16u && tx_step < (1u << 28)) tx_step <<= 1;
            else if (moved > 48u && tx_step > 1u) tx_step >>= 1;
            return true;
        };
        // the minimum of NEAR: (distance, id, lane); distance all ones = NEAR is empty
        auto near_min = [&](uint32_t& mh, uint32_t& ml, int& ml_lane) {
            if (near_n == 0) {  // (an index with a few deleted elements mostly has no such candidate: nothing to reduce)
                mh = ml = kEmptyWord;
                ml_lane = 63;
                return;
            }
            mh = wave_min_u32(nr_hi);
            ml = wave_min_u32(nr_hi == mh ? nr_lo : kEmptyWord);
            ml_lane = __builtin_ctzll(ballot(nr_hi == mh && nr_lo == ml) | (1ull << 63));
        };
        if constexpr (SPLIT) {
            const uint32_t dis = not_allowed(cur);  // :302-312: an entry point that is not allowed is a candidate only
            ep_only_candidate = dis != 0;
            if (dis) {
                if (lane == 0) {
                    nr_hi = cur_d;
                    nr_lo = cur;
                }
                near_n = 1;
                bound = kEmptyWord;
            } else {
                if (lane == 0) {
                    arr.hi[0] = cur_d;
                    arr.lo[0] = cur << 1;
                }
                bound = ef > 1 ? kEmptyWord : cur_d;
            }
        } else if constexpr (FILT) {
            const uint32_t dis = not_allowed(cur);  // :302-312: an entry point that is not allowed is a candidate only
            ep_only_candidate = dis != 0;
            if (lane == 0) {
                arr.hi[0] = cur_d;
                arr.lo[0] = (cur << 2) | (dis << 1);
            }
            bound = (ef == 1 && dis == 0) ? cur_d : kEmptyWord;
        } else {
            if (lane == 0) {
                arr.hi[0] = cur_d;
                arr.lo[0] = cur << 1;
            }
            bound = ef > 1 ? kEmptyWord : cur_d;  // hi of entry ef-1
        }
        uint32_t nvis = 1, vis_limit = vis.limit();
        bool overflow = false, key_overflow = false;
        if (lane == 0) (void)vis.test_and_set(cur);

        // Link rows fetched ahead of need (hints only: a wrong guess costs one 128-B load).
        //  A: the best unexpanded entry at the time a node's neighbours are being scored;
        //  B: the best of those neighbours when it beats A - the two nodes the next pop can be.
        uint32_t pfa_id = kNoId, pfb_id = kNoId, pfa_row = kNoId, pfb_row = kNoId;
        uint32_t cand;
#pragma unroll 1
        while (true) {
            if constexpr (SPLIT) {
                // the next candidate: the array's first open entry or the smallest key that is not allowed, by (distance, id)
                uint32_t nmh, nml;
                int nm_lane;
                near_min(nmh, nml, nm_lane);
                if (far_n != 0 && nmh >= tx) {
                    if (!far_refill()) {
                        overflow = key_overflow = true;
                        break;
                    }
                    near_min(nmh, nml, nm_lane);
                }
                uint32_t aid = kNoId, ad = kEmptyWord;
                const bool has_a = arr.peek_candidate(aid, ad);
                if (!has_a && nmh == kEmptyWord) break;
                const bool from_cb = nmh < ad || (nmh == ad && nml < aid);  // (an empty NEAR compares above everything)
                const uint32_t cd = from_cb ? nmh : ad;
                // (the epsilon condition's should_stop_search, as below)
                if (a.eps_ord != kEmptyWord && !(hops0 == 0 && ep_only_candidate) && cd > a.eps_ord) {
                    const uint32_t held = arr.count_valid();
                    if ((held < ef ? held : ef) >= a.eps_min) break;
                }
                if (bound != kEmptyWord && cd > bound) break;  // :316, `cand > lowerBound && size == ef` (an open entry of the array never is)
                if (from_cb) {
                    cand = nml;
                    if (lane == nm_lane) nr_hi = nr_lo = kEmptyWord;
                    near_n -= 1;
                } else {
                    (void)arr.pop_candidate(lane, cand);
                }
            } else {
            if constexpr (FILT) {
                // BaseSearchStopCondition::should_stop_search of the epsilon condition, asked before the candidate is
                // taken (searchBaseLayerST with a stop condition): the best open candidate lies beyond
                // epsilon and min_num_candidates results are held.  Its other clause (beyond the bound with the
                // result set full) is what an empty candidate set means on this array.
                if (a.eps_ord != kEmptyWord && !(hops0 == 0 && ep_only_candidate)) {
                    uint32_t pid, pd;
                    if (arr.peek_candidate(pid, pd) && pd > a.eps_ord) {
                        const uint32_t held = arr.count_allowed();
                        if ((held < ef ? held : ef) >= a.eps_min) break;
                    }
                }
            }
            if (!(FILT ? arr.pop_candidate_filtered(lane, cand) : arr.pop_candidate(lane, cand))) break;
            }
            hops0 += 1;
            if (nvis + g.l0stride > vis_limit) {
                uint32_t more = 0;
                if constexpr (VIS == 3) more = vis.take_spill(a.spill, a.spill_tables, a.redo + kRedoSpillMap, a.redo + 2, lane);
                if (more == 0) {
                    overflow = true;
                    break;
                }
                vis_limit = more;
            }
            uint32_t nb;
            if (cand == pfa_id) nb = pfa_row;
            else if (cand == pfb_id) nb = pfb_row;
            else nb = (uint32_t)lane < g.l0stride ? g.links0[(size_t)cand * g.l0stride + lane] : kNoId;
            const bool fresh = nb != kNoId && vis.test_and_set(nb);
            const uint64_t fm = ballot(fresh);
            const uint32_t m_all = (uint32_t)__builtin_popcountll(fm);
            if (m_all == 0) continue;
            nvis += m_all;
            ndis += m_all;
            // compact the unvisited neighbours to lanes 0..m-1, list order kept: a full
            // permutation (fresh lanes to their rank, the others behind them) so that no
            // two lanes of the forward permute write the same destination
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
            const uint32_t dest = fresh ? rank : m_all + ((uint32_t)lane - rank);
            const uint32_t ids_c = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)(fresh ? nb : kNoId));
            uint32_t peek_d = kEmptyWord;
            bool have_a = arr.peek_candidate(pfa_id, peek_d);
            if constexpr (SPLIT) {  // (a hint: FAR is not looked at)
                uint32_t nmh, nml;
                int nm_lane;
                near_min(nmh, nml, nm_lane);
                if (nmh < peek_d || (nmh == peek_d && nmh != kEmptyWord && nml < pfa_id) || (!have_a && nmh != kEmptyWord)) {
                    pfa_id = nml;
                    peek_d = nmh;
                    have_a = true;
                }
            }
            if (have_a)
                pfa_row = (uint32_t)lane < g.l0stride ? g.links0[(size_t)pfa_id * g.l0stride + lane] : kNoId;
            else
                pfa_id = kNoId;

#pragma unroll 1
            for (uint32_t c0 = 0; c0 < m_all; c0 += 32) {
                const uint32_t m = m_all - c0 < 32u ? m_all - c0 : 32u;
                const uint32_t ids_part = c0 == 0 ? ids_c : bpermute(ids_c, (lane + 32) & 63);
                uint32_t dis = 0;  // requested ahead of the row gathers, used after them
                if constexpr (FILT) dis = (uint32_t)lane < m ? not_allowed(ids_part) : 0u;
                uint32_t dord = score_nodes<NCH, GPB, IS_L2, FULL>(g, q, ids_part, m, lane);
                if ((uint32_t)lane >= m) dord = kEmptyWord;
                {
                    const uint32_t best = wave_min_u32(dord);
                    if (best < bound && best < peek_d) {
                        pfb_id = readlane(ids_part, __builtin_ctzll(ballot(dord == best)));
                        pfb_row = (uint32_t)lane < g.l0stride ? g.links0[(size_t)pfb_id * g.l0stride + lane] : kNoId;
                    }
                }
                // sequential admission in list order with the evolving bound, :339-349
                uint64_t pend = ballot(dord < bound);
                bool lost = false;  // lane 63: an unexpanded key at the bound's distance fell off the end
                bool trim_due = false;
                if constexpr (SPLIT) {
                    // List order, evolving bound (:339-349).  Only an ALLOWED neighbour's admission moves the bound, so the
                    // neighbours that are not allowed and stand ahead of the next allowed one go into the buffer together.
                    const uint64_t nal = ballot(dis != 0u);
                    while (pend) {
                        const uint64_t pa = pend & ~nal;
                        const int al = pa ? __builtin_ctzll(pa) : 64;
                        const uint64_t seg = pend & nal & (al == 64 ? ~0ull : ((1ull << al) - 1ull));
                        if (seg) {
                            const bool in_seg = (seg >> lane) & 1u;
                            // below tx and a lane free: NEAR; else FAR
                            const uint64_t em = ballot(in_seg && dord < tx);
                            const uint32_t nfree = (uint32_t)__builtin_popcountll(ballot(nr_hi == kEmptyWord));
                            const bool go = in_seg && dord < tx && lane_rank(em) < nfree;
                            if (em && nfree) near_put(go, dord, ids_part);
                            const bool to_far = in_seg && !go;
                            const uint64_t fm = ballot(to_far);
                            if (fm) {
                                const uint32_t cnt = (uint32_t)__builtin_popcountll(fm);
                                if (far_n + cnt > a.cb_cap) {
                                    (void)far_sweep(tx);  // (drops the dead keys)
                                    // still more than three quarters full: the next pushes would sweep again and again -
                                    // the query goes to the large configuration and the host gives the next launches
                                    // at this ef a larger buffer
                                    if (far_n + cnt > a.cb_cap - (a.cb_cap >> 2)) {
                                        overflow = key_overflow = true;
                                        break;
                                    }
                                }
                                lds_sync();
                                if (to_far) far[far_n + lane_rank(fm)] = ((uint64_t)dord << 32) | ids_part;
                                lds_sync();
                                far_n += cnt;
                                if (ballot(to_far && dord < tx)) {
                                    const uint32_t lowest = wave_min_u32(to_far ? dord : kEmptyWord);
                                    tx = lowest < tx ? lowest : tx;
                                }
                            }
                            pend &= ~seg;
                        }
                        if (al < 64) {
                            pend &= ~(1ull << al);
                            const uint32_t d = readlane(dord, al);
                            if (d < bound) {
                                uint32_t prev_hi, prev_lo;
                                arr.insert(d, readlane(ids_part, al) << 1, lane, prev_hi, prev_lo);
                                bound = arr.hi_at(ef - 1);
                                lost |= lane == 63 && prev_hi != kEmptyWord && prev_hi == bound && (prev_lo & 1u) == 0u;
                                trim_due = true;
                                pend &= ballot(dord < bound);
                            }
                        }
                    }
                    if (trim_due) arr.trim(ef, bound, lane);
                    pend = 0;
                }
                if constexpr (!FILT) {
                    // Two or more neighbours below the bound: one merge instead of one wave-wide shift each.  Every key
                    // learns its place in the merged order by counting (an array entry moves back by the number of new
                    // keys below it, a new key stands behind the array entries and the new keys below it), the key
                    // that lands on index ef-1 gives the bound that sequential admission would end with, and the keys
                    // move through LDS.  The two orders of work end in the same array unless a new key TIES with the
                    // final bound and is not alone at that distance (:339 admits on strict <, so which of the equal
                    // keys stay depends on the list order): then nothing has been touched yet and the sequential loop
                    // below runs as before.  (Otherwise: a new key beyond the final bound is either never admitted or
                    // admitted, pushed behind index ef-1 and trimmed; one at or below it is admitted in both.)
                    if (pend & (pend - 1)) {
                        const bool in_p = (pend >> lane) & 1u;
                        const uint32_t my_lo = ids_part << 1;
                        const uint64_t mk = ((uint64_t)dord << 32) | my_lo;
                        uint32_t sh[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) sh[r] = 0;
                        uint32_t mypos = 0;
                        for (uint64_t it = pend; it; it &= it - 1) {
                            const int c = __builtin_ctzll(it);
                            const uint64_t ck = ((uint64_t)readlane(dord, c) << 32) | readlane(my_lo, c);
                            uint32_t below = 0;
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                const bool lt = arr.key(r) < ck;  // (keys are distinct: a new key is an unvisited node)
                                sh[r] += lt ? 0u : 1u;
                                below += (uint32_t)__builtin_popcountll(ballot(lt));
                            }
                            mypos += lane == c ? below : (mk > ck ? 1u : 0u);
                        }
                        uint32_t nbound = kEmptyWord;  // an EMPTY entry on index ef-1: the result set is not full yet
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const uint64_t h = ballot((uint32_t)(r * 64 + lane) + sh[r] == ef - 1u);
                            if (h) nbound = readlane(arr.hi[r], __builtin_ctzll(h));
                        }
                        {
                            const uint64_t h = ballot(in_p && mypos == ef - 1u);
                            if (h) nbound = readlane(dord, __builtin_ctzll(h));
                        }
                        bool tie = false;
                        const uint64_t at = ballot(in_p && dord == nbound);
                        if (at) {
                            uint32_t cnt = (uint32_t)__builtin_popcountll(at);
#pragma unroll
                            for (int r = 0; r < R; ++r) cnt += (uint32_t)__builtin_popcountll(ballot(arr.hi[r] == nbound));
                            tie = cnt >= 2;
                        }
                        if (!tie) {
                            uint64_t* st = reinterpret_cast<uint64_t*>(lds + a.stage_off);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                const uint32_t pos = (uint32_t)(r * 64 + lane) + sh[r];
                                if (pos < 64u * R) st[pos] = arr.key(r);
                                else lost |= arr.hi[r] != kEmptyWord && arr.hi[r] == nbound && (arr.lo[r] & 1u) == 0u;
                            }
                            if (in_p && mypos < 64u * R) st[mypos] = mk;  // (one beyond the end is beyond the bound: ties went the other way)
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                const uint64_t v = st[r * 64 + lane];
                                arr.hi[r] = (uint32_t)(v >> 32);
                                arr.lo[r] = (uint32_t)v;
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                            __builtin_amdgcn_wave_barrier();
                            bound = nbound;
                            trim_due = true;
                            pend = 0;
                        }
                    }
                }
                while (!SPLIT && pend) {
                    const int r = __builtin_ctzll(pend);
                    pend &= pend - 1;
                    const uint32_t d = readlane(dord, r);
                    if (d < bound) {
                        const uint32_t id = readlane(ids_part, r);
                        uint32_t prev_hi, prev_lo;
                        if constexpr (FILT) {
                            arr.insert(d, (id << 2) | (readlane(dis, r) << 1), lane, prev_hi, prev_lo);
                            uint32_t bhi, bpos;
                            const bool full = arr.bound_filtered(ef, bhi, bpos);
                            bound = full ? bhi : kEmptyWord;
                            // a key that fell off the end is missed unless it was dead: with the set full
                            // it lay behind the bound entry, where only unexpanded keys AT the bound live
                            lost |= lane == 63 && prev_hi != kEmptyWord && (!full || (prev_hi == bound && (prev_lo & 1u) == 0u));
                            if (full) arr.trim_filtered(bpos, bound, lane);
                        } else {
                            arr.insert(d, id << 1, lane, prev_hi, prev_lo);
                            bound = arr.hi_at(ef - 1);
                            lost |= lane == 63 && prev_hi != kEmptyWord && prev_hi == bound && (prev_lo & 1u) == 0u;
                            trim_due = true;
                        }
                        pend &= ballot(dord < bound);
                    }
                }
                // Entries behind index ef-1 whose distance is no longer the bound's are dropped once per chunk, not per
                // insert: until then they sit behind the live ones (they are larger), fall off the end before them, and
                // nothing between two pops looks at them.
                if constexpr (!FILT) {
                    if (trim_due) arr.trim(ef, bound, lane);
                }
                if (ballot(lost)) overflow = key_overflow = true;  // (any lane: the merge loses keys from every lane that moves past the end)
            }
            if (overflow) break;
        }
        if constexpr (VIS == 3) vis.release_spill(a.spill, a.redo + kRedoSpillMap, lane);
        if constexpr (!LARGE) {  // what a search at this ef visits: the host sizes later launches' LDS tables by it (launch_cfg)
            if (a.report && lane == 0) atomicAdd(&a.redo[kRedoVisitedSum], (nvis + 8u) >> 4);
        }

        if (overflow) {
            if constexpr (!LARGE) {
                if (lane == 0) {
                    const uint32_t slot = atomicAdd(&a.redo[0], 1u);
                    a.redo[kRedoHeader + slot] = qi;
                    if (FILT && key_overflow) atomicAdd(&a.redo[kRedoKeyOverflow], 1u);  // (the host sizes the next launches' arrays)
                }
                continue;
            } else {
                // more live candidates outside the result set than the largest array holds (a filter that
                // rejects almost everything; without a filter: more equal-distance keys at the bound than
                // fit behind it, i.e. hundreds of duplicate vectors): reported, never answered approximately
                for (uint32_t i = lane; i < k; i += 64) {
                    od[i] = __builtin_nanf("");
                    ol[i] = ~0ull;
                }
                if (lane == 0) {
                    const uint32_t slot = atomicAdd(&a.redo[1], 1u);
                    if (a.fail_list) a.fail_list[slot] = qi;  // answered by hnsw_search_unbounded_kernel after this launch
                    if (a.out_stats) {
                        uint32_t* st = a.out_stats + (size_t)qi * 4;
                        st[kStatNdis] = ndis;
                        st[kStatHops0] = hops0;
                        st[kStatHopsUpper] = hops_up;
                        st[kStatFound] = 0xFFFFFFFFu;
                    }
                }
                continue;
            }
        }

        // ---- epilogue, hnsw_core.hpp:1150-1158 + hnswlib.hpp:87-99
        // results = entries 0..min(size, ef)-1 (filtered: the first ef allowed entries); keep the k
        // smallest, order by (dist, label).  pos[r] = place of this lane's entry among the results.
        // place_of(r, base): this lane's entry of register r among the results (kEmptyWord: not one);
        // `base` runs over the registers in order
        auto place_of = [&](int r, uint32_t& base) -> uint32_t {
            if constexpr (FILT && !SPLIT) {
                // (filter_results of the epsilon condition: results beyond epsilon are dropped; they are the tail of the order)
                const bool ok = arr.hi[r] != kEmptyWord && (arr.lo[r] & 2u) == 0u && arr.hi[r] <= a.eps_ord;
                const uint64_t m = ballot(ok);
                const uint32_t p = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                base += (uint32_t)__builtin_popcountll(m);
                return ok ? p : kEmptyWord;
            } else {
                const uint32_t idx = (uint32_t)(r * 64 + lane);
                bool ok = idx < ef && !(arr.hi[r] == kEmptyWord && arr.lo[r] == kEmptyWord);
                if constexpr (SPLIT) ok = ok && arr.hi[r] <= a.eps_ord;  // (filter_results of the epsilon condition)
                base += (uint32_t)__builtin_popcountll(ballot(ok));
                return ok ? idx : kEmptyWord;
            }
        };
        uint32_t size = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) (void)place_of(r, size);
        if (size > ef) size = ef;
        const uint32_t kout = size < k ? size : k;
        // stage (ord dist, label) in LDS, then every entry finds its rank among ties
        __builtin_amdgcn_wave_barrier();
        uint32_t* s_hi = lds;                                        // [64 R]
        uint64_t* s_lab = reinterpret_cast<uint64_t*>(lds + 64 * R);  // [64 R]
        bool tie = false;
        uint32_t base = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t pl = place_of(r, base);
            if (pl < kout) {
                s_hi[pl] = arr.hi[r];
                s_lab[pl] = g.labels[arr.lo[r] >> S];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        base = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t pl = place_of(r, base);
            if (pl > 0 && pl < kout && s_hi[pl - 1] == arr.hi[r]) tie = true;
        }
        const bool any_tie = ballot(tie) != 0ull;
        base = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t pl = place_of(r, base);
            if (pl < kout) {
                uint32_t pos = pl;
                const uint64_t lab = s_lab[pl];
                if (any_tie) {
                    pos = 0;
                    for (uint32_t j = 0; j < kout; ++j) {
                        const uint32_t oh = s_hi[j];
                        const uint64_t olab = s_lab[j];
                        pos += (oh < arr.hi[r] || (oh == arr.hi[r] && olab < lab)) ? 1u : 0u;
                    }
                }
                od[pos] = ord2f(arr.hi[r]);
                ol[pos] = lab;
            }
        }
        for (uint32_t i = kout + lane; i < k; i += 64) {
            od[i] = __builtin_inff();
            ol[i] = ~0ull;
        }
        if (a.out_stats && lane == 0) {
            uint32_t* st = a.out_stats + (size_t)qi * 4;
            st[kStatNdis] = ndis;
            st[kStatHops0] = hops0;
            st[kStatHopsUpper] = hops_up;
            st[kStatFound] = kout;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ the search without a capacity limit
//
// The queries the large configuration gave up on (a filter that rejects almost every node: the candidates that are not
// results outnumber any register array; the reference's two std::priority_queues have no limit, hnsw_core.hpp:1141-1148).
// A literal restatement of the non-bare searchBaseLayerST instead of the one-array design: the candidate set is a
// 64-ary min-heap of (ordered distance << 32 | id) keys in HBM (one wave: a sift-down compares the 64 children of a
// node in one step), the result set a sorted array of at most ef keys in LDS, the visited set the large
// configuration's bitmap.  Same distances (score_nodes), same order of admission, same counters: bit-identical to the
// oracle, just slow (every heap step is an L2 round trip) - it exists so that no query goes unanswered.
struct HbmHeap {
    uint64_t* h;
    uint32_t size, cap;
    // (agent-scope accesses: L2, not this CU's L1, which another lane's store does not refresh)
    static __device__ __forceinline__ uint64_t ld(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void st(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ bool push(uint64_t key, int lane) {  // key wave-uniform
        if (size >= cap) return false;
        uint32_t pos = size++;
        while (pos > 0) {
            const uint32_t par = (pos - 1u) >> 6;
            const uint64_t pk = ld(h + par);
            if (pk <= key) break;
            if (lane == 0) st(h + pos, pk);
            pos = par;
        }
        if (lane == 0) st(h + pos, key);
        return true;
    }
    __device__ __forceinline__ uint64_t top() const { return ld(h); }
    __device__ __forceinline__ void pop(int lane) {  // size > 0
        const uint64_t last = ld(h + --size);
        uint32_t pos = 0;
        while (true) {
            const uint32_t c = 64u * pos + 1u;
            if (c >= size) break;
            const uint64_t ck = c + (uint32_t)lane < size ? ld(h + c + lane) : ~0ull;
            const uint32_t mh = wave_min_u32((uint32_t)(ck >> 32));
            const uint32_t ml = wave_min_u32((uint32_t)(ck >> 32) == mh ? (uint32_t)ck : kEmptyWord);
            const uint64_t mk = ((uint64_t)mh << 32) | ml;
            if (mk >= last) break;
            const int ml_lane = __builtin_ctzll(ballot(ck == mk));
            if (lane == 0) st(h + pos, mk);
            pos = c + (uint32_t)ml_lane;
        }
        if (lane == 0) st(h + pos, last);
    }
};

constexpr uint32_t kUnbResults = 512;  // >= kMaxEf + 1

template <int NCH, int GPB, bool IS_L2>
__global__ __launch_bounds__(64) void hnsw_search_unbounded_kernel(SearchArgs a) {
    __shared__ uint64_t s_res[kUnbResults];  // the result set, ascending (ordered dist << 32 | id)
    __shared__ uint64_t s_lab[kUnbResults];
    const int lane = lane_id();
    const DevGraph& g = a.g;
    BitmapVisited vis;
    vis.bits = a.bitmaps + (size_t)blockIdx.x * a.bitmap_words;
    vis.words = a.bitmap_words;
    HbmHeap heap;
    heap.h = a.heaps + (size_t)blockIdx.x * a.heap_cap;
    heap.cap = a.heap_cap;
    const uint32_t n_work = a.redo[1];
    auto not_allowed = [&](uint32_t id) { return a.allow ? ((a.allow[id >> 5] >> (id & 31u)) & 1u) ^ 1u : 0u; };
#pragma unroll 1
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const uint32_t qi = a.fail_list[w];
        const uint32_t k = a.k, ef = a.ef;
        float* od = a.out_dist + (size_t)qi * k;
        uint64_t* ol = a.out_label + (size_t)qi * k;
        uint32_t ndis = 0, hops0 = 0, hops_up = 0;
        f32x4 q[NCH];
        {
            const float* qp = a.queries + (size_t)qi * g.dim;
            const int t = lane & 15;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                float e[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t pos = (uint32_t)(64 * j + 4 * t + c);
                    e[c] = pos < g.dim ? qp[pos] : 0.f;
                }
                q[j] = f32x4{e[0], e[1], e[2], e[3]};
            }
        }
        vis.clear(lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();

        // entry point and greedy descent, hnsw_core.hpp:1111-1138 (as in hnsw_search_kernel)
        uint32_t cur = g.entry;
        uint32_t cur_d = readlane(score_nodes<NCH, GPB, IS_L2>(g, q, cur, 1u, lane), 0);
        ndis += 1;
        uint32_t cur_row = g.max_level > 0 ? g.upper_row[cur] : 0u;
        for (int level = g.max_level; level > 0; --level) {
            bool changed = true;
            while (changed) {
                changed = false;
                const uint32_t row = cur_row + (uint32_t)(level - 1);
                const uint32_t nb = (uint32_t)lane < g.ustride ? g.upper_links[(size_t)row * g.ustride + lane] : kNoId;
                const uint32_t nb_row = nb != kNoId ? g.upper_row[nb] : 0u;
                const uint32_t n = (uint32_t)__builtin_popcountll(ballot(nb != kNoId));
                hops_up += 1;
                ndis += n;
                if (n == 0) break;
                uint32_t d = score_nodes<NCH, GPB, IS_L2>(g, q, nb, n, lane);
                if ((uint32_t)lane >= n) d = kEmptyWord;
                const uint32_t bd = wave_min_u32(d);
                if (bd < cur_d) {
                    cur_d = bd;
                    const int bl = __builtin_ctzll(ballot(d == bd));
                    cur = readlane(nb, bl);
                    cur_row = readlane(nb_row, bl);
                    changed = true;
                }
            }
        }

        // layer 0: the oracle's searchBaseLayerST<false>, line by line
        uint32_t nres = 0;
        heap.size = 0;
        auto wave_lds_sync = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        };
        auto res_insert = [&](uint64_t key) {  // top.emplace(d, id); if (top.size() > ef) top.pop();
            uint32_t pos = 0;
            for (uint32_t base = 0; base < nres; base += 64) {
                const uint32_t j = base + (uint32_t)lane;
                pos += (uint32_t)__builtin_popcountll(ballot(j < nres && s_res[j] < key));
            }
            for (int base = (int)((nres + 63u) & ~63u) - 64; base >= 0; base -= 64) {
                const uint32_t j = (uint32_t)base + (uint32_t)lane;
                const bool mv = j < nres && j >= pos && j + 1u < kUnbResults;
                const uint64_t v = mv ? s_res[j] : 0ull;
                wave_lds_sync();
                if (mv) s_res[j + 1u] = v;
                wave_lds_sync();
            }
            if (lane == 0) s_res[pos] = key;
            wave_lds_sync();
            nres += 1;
            if (nres > ef) nres = ef;
        };
        bool failed = false;
        const bool ep_only_candidate = not_allowed(cur) != 0;  // (:310 seeds it at "distance" -FLT_MAX: no stop condition fires on it)
        {
            const uint64_t key = ((uint64_t)cur_d << 32) | cur;
            (void)heap.push(key, lane);  // (cap >= 1)
            if (!ep_only_candidate) res_insert(key);
            if (lane == 0) (void)vis.test_and_set(cur);
        }
#pragma unroll 1
        while (heap.size > 0) {
            const uint64_t ck = heap.top();
            const uint32_t cd = (uint32_t)(ck >> 32), cand = (uint32_t)ck;
            const uint32_t lb = nres ? (uint32_t)(s_res[nres - 1u] >> 32) : kEmptyWord;
            if (cd > lb && nres == ef) break;                                        // :316
            if (a.eps_ord != kEmptyWord && cd > a.eps_ord && nres >= a.eps_min && !(hops0 == 0 && ep_only_candidate)) break;  // epsilon condition
            heap.pop(lane);
            hops0 += 1;
            const uint32_t nb = (uint32_t)lane < g.l0stride ? g.links0[(size_t)cand * g.l0stride + lane] : kNoId;
            const bool fresh = nb != kNoId && vis.test_and_set(nb);
            const uint64_t fm = ballot(fresh);
            const uint32_t m_all = (uint32_t)__builtin_popcountll(fm);
            if (m_all == 0) continue;
            ndis += m_all;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
            const uint32_t dest = fresh ? rank : m_all + ((uint32_t)lane - rank);
            const uint32_t ids_c = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)(fresh ? nb : kNoId));
#pragma unroll 1
            for (uint32_t c0 = 0; c0 < m_all && !failed; c0 += 32) {
                const uint32_t m = m_all - c0 < 32u ? m_all - c0 : 32u;
                const uint32_t ids_part = c0 == 0 ? ids_c : bpermute(ids_c, (lane + 32) & 63);
                const uint32_t dis = (uint32_t)lane < m ? not_allowed(ids_part) : 0u;
                uint32_t dord = score_nodes<NCH, GPB, IS_L2>(g, q, ids_part, m, lane);
                if ((uint32_t)lane >= m) dord = kEmptyWord;
                // sequential admission in list order with the evolving bound, :339-349
                for (uint32_t r = 0; r < m; ++r) {
                    const uint32_t d = readlane(dord, (int)r);
                    const uint32_t bound = nres ? (uint32_t)(s_res[nres - 1u] >> 32) : kEmptyWord;
                    if (nres < ef || bound > d) {
                        const uint64_t key = ((uint64_t)d << 32) | readlane(ids_part, (int)r);
                        if (!heap.push(key, lane)) {
                            failed = true;
                            break;
                        }
                        if (!readlane(dis, (int)r)) res_insert(key);
                    }
                }
            }
            if (failed) break;
        }
        if (failed) {  // more candidates than heap_cap: the row stays as the large configuration reported it
            if (lane == 0) atomicAdd(&a.redo[3], 1u);
            continue;
        }

        // epilogue, hnsw_core.hpp:1150-1158: the k smallest by (dist, id), ordered by (dist, label); with the stop
        // condition only those within epsilon
        uint32_t kout = nres < k ? nres : k;
        {
            uint32_t within = 0;
            for (uint32_t base = 0; base < kout; base += 64) {
                const uint32_t j = base + (uint32_t)lane;
                within += (uint32_t)__builtin_popcountll(ballot(j < kout && (uint32_t)(s_res[j] >> 32) <= a.eps_ord));
            }
            kout = within;
        }
        for (uint32_t j = lane; j < kout; j += 64) s_lab[j] = g.labels[(uint32_t)s_res[j]];
        wave_lds_sync();
        for (uint32_t j = lane; j < kout; j += 64) {
            const uint32_t hj = (uint32_t)(s_res[j] >> 32);
            const uint64_t lj = s_lab[j];
            uint32_t pos = 0;
            for (uint32_t i = 0; i < kout; ++i) {
                const uint32_t hi_i = (uint32_t)(s_res[i] >> 32);
                pos += (hi_i < hj || (hi_i == hj && s_lab[i] < lj)) ? 1u : 0u;
            }
            od[pos] = ord2f(hj);
            ol[pos] = lj;
        }
        for (uint32_t i = kout + lane; i < k; i += 64) {
            od[i] = __builtin_inff();
            ol[i] = ~0ull;
        }
        if (a.out_stats && lane == 0) {
            uint32_t* st = a.out_stats + (size_t)qi * 4;
            st[kStatNdis] = ndis;
            st[kStatHops0] = hops0;
            st[kStatHopsUpper] = hops_up;
            st[kStatFound] = kout;
        }
        wave_lds_sync();
    }
}

__global__ __launch_bounds__(256) void and_bitmaps_kernel(const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t words) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < words) out[i] = a[i] & b[i];
}

// ------------------------------------------------------------------ launch

// GPBL: row chunks in flight in the large configuration's kernel; SPILL_ONLY: only ever called with spill tables
// (the instantiations without them are left out)
template <int NCH, int GPB, int R, int RL, bool FILT, int GPBL = GPB, bool SPILL_ONLY = false, bool FULL = false>
int launch_cfg(const SearchArgs& a, hipStream_t stream, std::string* err) {
    const size_t epilogue_bytes = (size_t)64 * R * 12;
    size_t lds_small = (size_t)4 << a.hbits;
    if (lds_small < epilogue_bytes) lds_small = epilogue_bytes;
    // FAR of a filtered launch (hnsw_search_kernel, SPLIT): 256 keys, twice as many with each step of filt_level (the host:
    // what it knows of the filter's selectivity, and launches at this ef that reported overflows)
    const size_t far_bytes = FILT ? (size_t)2048 << (a.filt_level < kFiltLevels ? a.filt_level : kFiltLevels - 1u) : 0;
    if constexpr (R >= 8 || FILT) {
        // Searches that visit far more than a 16 KB LDS table holds: the whole visited set in HBM, one table per resident
        // wave.  From ef 317 on only: below, the LDS table with a spill table behind it is as fast or faster now that a
        // launch cannot run out of spill tables (20 M x 128, 10 000 queries, LDS + spill against this: ef 200 6.4 / 7.0 ms,
        // ef 300 9.5 / 9.6, ef 400 14.2 / 12.3, ef 508 23.2 / 15.7; double hashing then took another 10 % off the
        // left-hand numbers).  Filtered searches at any ef once launches reported that the 8 KB table is outgrown: a
        // filter that keeps a tenth makes a search visit ten times the nodes, and a 16-32 KB table beside the candidate
        // buffer leaves 3-6 queries per CU.
        if (a.ghash && a.hbits >= (FILT ? 12u : 13u)) {
            // persistent blocks: as many as are resident at once, each with its own table
            const size_t pers_lds = epilogue_bytes > far_bytes ? epilogue_bytes : far_bytes;
            const uint32_t resident = [&] {
                int per_cu = 0, dev = 0;
                hipDeviceProp_t prop;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hnsw_search_kernel<NCH, GPB, R, 1, true, FILT, FULL>, 64,
                                                                 pers_lds) != hipSuccess ||
                    hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu <= 0)
                    return 2048u;
                return (uint32_t)per_cu * (uint32_t)prop.multiProcessorCount;
            }();
            uint32_t blocks = resident < kGlobalHashBlocks ? resident : kGlobalHashBlocks;
            if (a.nq < blocks) blocks = a.nq;
            SearchArgs c = a;
            c.stage_off = 0;  // (the candidate buffer shares the epilogue's space: it is done with when the epilogue starts)
            c.cb_cap = (uint32_t)(far_bytes / 8);
            if (a.g.metric == 0)
                hipLaunchKernelGGL((hnsw_search_kernel<NCH, GPB, R, 1, true, FILT, FULL>), dim3(blocks), dim3(64), pers_lds, stream, c);
            else
                hipLaunchKernelGGL((hnsw_search_kernel<NCH, GPB, R, 1, false, FILT, FULL>), dim3(blocks), dim3(64), pers_lds, stream, c);
            lds_small = 0;
        }
    }
    SearchArgs b = a;
    if (lds_small) {
        // The table gets the share of the CU's LDS that the kernel's register-limited residency leaves to a
        // query: LDS, not registers, should never be what bounds the chains in flight (a 16 KB table at ef 100
        // left 10 queries resident per CU where the registers allow 16: 1.27 ms per C2 batch against 0.92).
        // Residency is given up for a larger table only when `hbits` asks for one, i.e. after launches at this ef
        // in which more than 2 % of the queries outgrew the table (a 20M x 128 graph at ef 100 visits 2 000 nodes a
        // query where C2 visits 900: sending most of a batch through the spill tables cost 5.9 ms against 2.6).
        static const uint32_t per_cu = [&] {  // blocks of this instantiation a CU holds by registers alone
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hnsw_search_kernel<NCH, GPB, R, SPILL_ONLY ? 3 : 0, true, FILT, FULL>, 64, 1024) != hipSuccess || n <= 0)
                return 8u;
            // (a body under 100 registers could have 20: measured in round 3, profiles/r3/residency_and_batch_size.txt -
            // no gain at ef <= 100, where a loaded GPU runs at the HBM rate, and the 6 KB table that is left is outgrown
            // by half the searches from ef 128 on)
            return (uint32_t)(n > 16 ? 16 : n);
        }();
        size_t share = ((size_t)160 * 1024 / per_cu) / 1024 * 1024;
        if (lds_small < share) lds_small = share;
        // with spill tables to carry on in, never the 32 KB that `hbits` may ask for: that cost more than the spilling of
        // the long searches did (3.1 against 2.2 ms at ef 100 on the 20 M-row graph; round 2 capped at 16 KB, 10 per CU).
        // Round 3, on the same graph at ef 100-300: 20 KB (8 per CU) is never slower than 16 KB (10 per CU) - these launches
        // wait for rows whatever the residency - and the table that just holds an average search is the fastest one: a
        // search that has spilled pays an L2 round trip per hop, so the later it spills the better, until the residency
        // falls to 6 per CU (ms at 16 / 20 / 22 / 26 KB - ef 160, 2 840 nodes visited: 3.35 / 3.11 / 3.33 / 3.65; ef 200,
        // 3 330 nodes: 5.49 / 4.36 / 4.35 / 4.97; ef 252, 3 910 nodes: 6.54 / 6.23 / 5.65 / 6.04; ef 300, 4 400 nodes: 8.56 /
        // 8.21 / 8.47 / 8.23).  So: 20 KB, and 22 or 26 KB when that is the table whose fill limit reaches the average that the
        // earlier launches at this ef reported (vis_mean) where 20 KB's does not.
        if (FILT && a.spill && lds_small > (size_t)16 * 1024) lds_small = (size_t)16 * 1024;  // (filtered: the candidate buffer is added below)
        if (!FILT && a.spill && lds_small > (size_t)16 * 1024) {
            auto holds = [&](size_t kb) {
                const size_t entries = (kb * 1024 - (size_t)64 * R * 8) / 4, limit = entries - (entries >> 2);
                return limit * 100 >= (size_t)a.vis_mean * 97;
            };
            size_t kb = 20;
            if (a.vis_mean != 0 && !holds(20)) kb = holds(22) ? 22 : holds(26) ? 26 : 20;  // (26 KB: 6 per CU, the last size that pays)
            lds_small = kb * 1024;
        }
        if (lds_small > (size_t)64 * 1024) lds_small = (size_t)64 * 1024;
        // (behind the table: the 64 R keys of KeyArray::merge)
        // (behind the table: the 64 R keys of KeyArray::merge; filtered: the buffer of the candidates that are not allowed)
        size_t stage_bytes = (size_t)64 * R * 8;
        if constexpr (FILT) {
            stage_bytes = far_bytes;
            if (lds_small < ((size_t)4 << a.hbits) + stage_bytes) lds_small = ((size_t)4 << a.hbits) + stage_bytes;
            if (lds_small > (size_t)64 * 1024) lds_small = (size_t)64 * 1024;
            b.cb_cap = (uint32_t)(stage_bytes / 8);
        }
        b.lds_entries = (uint32_t)((lds_small - stage_bytes) / 4);
        b.stage_off = b.lds_entries;
        auto is_prime = [](uint32_t v) {  // (a few trial divisions below 16 384)
            for (uint32_t d = 2; d * d <= v; ++d)
                if (v % d == 0u) return false;
            return v >= 2u;
        };
        b.lds_mod = b.lds_entries;
        while (!is_prime(b.lds_mod)) --b.lds_mod;
        // the spill variant costs the searches that never spill 1-5 % (a uniform branch per hop in the
        // hash probe): the caller passes the tables only when earlier launches at this ef needed them
        if (a.spill) {
            if (a.g.metric == 0)
                hipLaunchKernelGGL((hnsw_search_kernel<NCH, GPB, R, 3, true, FILT, FULL>), dim3(a.nq), dim3(64), lds_small, stream, b);
            else
                hipLaunchKernelGGL((hnsw_search_kernel<NCH, GPB, R, 3, false, FILT, FULL>), dim3(a.nq), dim3(64), lds_small, stream, b);
        } else if constexpr (!SPILL_ONLY) {
            if (a.g.metric == 0)
                hipLaunchKernelGGL((hnsw_search_kernel<NCH, GPB, R, 0, true, FILT, FULL>), dim3(a.nq), dim3(64), lds_small, stream, b);
            else
                hipLaunchKernelGGL((hnsw_search_kernel<NCH, GPB, R, 0, false, FILT, FULL>), dim3(a.nq), dim3(64), lds_small, stream, b);
        }
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) {
        const uint32_t blocks = a.nq < a.large_blocks ? a.nq : a.large_blocks;
        if (a.g.metric == 0)
            hipLaunchKernelGGL((hnsw_search_kernel<NCH, GPBL, RL, 2, true, FILT, FULL>), dim3(blocks), dim3(64),
                               (size_t)64 * RL * 12, stream, a);
        else
            hipLaunchKernelGGL((hnsw_search_kernel<NCH, GPBL, RL, 2, false, FILT, FULL>), dim3(blocks), dim3(64),
                               (size_t)64 * RL * 12, stream, a);
        e = hipGetLastError();
    }
    if constexpr (FILT) {  // (the launcher hands the heaps to filtered searches only)
    if (e == hipSuccess && a.fail_list && a.heaps && a.heap_blocks) {
        uint32_t blocks = a.heap_blocks < a.large_blocks ? a.heap_blocks : a.large_blocks;
        if (a.nq < blocks) blocks = a.nq;
        if (a.g.metric == 0)
            hipLaunchKernelGGL((hnsw_search_unbounded_kernel<NCH, GPB, true>), dim3(blocks), dim3(64), 0, stream, a);
        else
            hipLaunchKernelGGL((hnsw_search_unbounded_kernel<NCH, GPB, false>), dim3(blocks), dim3(64), 0, stream, a);
        e = hipGetLastError();
    }
    }
    if (e != hipSuccess) {
        if (err) *err = std::string("hnsw_search launch failed: ") + hipGetErrorString(e);
        return 3;
    }
    return 0;
}

template <int NCH, int GPB, bool FILT, bool FULL = false>
int launch_r(const SearchArgs& a, hipStream_t stream, std::string* err) {
    if constexpr (FILT) {
        // filtered: the first level's key array holds the allowed nodes only, so it is sized like the bare search's; the
        // candidates that are not allowed wait in a buffer in LDS (hnsw_search_kernel, SPLIT; its size: launch_cfg).  The
        // large configuration behind it is the one-array design with 1024 keys.
        // (half the rows in flight per scoring pass, as the bare search has from three key registers on: a filtered
        // hop brings 2-12 unvisited neighbours, and the registers that frees make four waves per SIMD)
        constexpr int GF = GPB >= 2 ? GPB / 2 : GPB;
        if (a.ef <= 60) return launch_cfg<NCH, GF, 1, 16, true, GF, false, FULL>(a, stream, err);
        if (a.ef <= 124) return launch_cfg<NCH, GF, 2, 16, true, GF, false, FULL>(a, stream, err);
        if (a.ef <= 188) return launch_cfg<NCH, GF, 3, 16, true, GF, false, FULL>(a, stream, err);
        if (a.ef <= 252) return launch_cfg<NCH, GF, 4, 16, true, GF, false, FULL>(a, stream, err);
        if (a.ef <= 316) return launch_cfg<NCH, GF, 5, 16, true, GF, false, FULL>(a, stream, err);
        return launch_cfg<NCH, GF, 8, 16, true, GF, false, FULL>(a, stream, err);
    } else {
    // capacity 64 R > ef: with capacity == ef every tie at the bound that is pushed off the end sends
    // its query to the large configuration (measured on the C4 data: ef 64 took 13.4 ms against 7.4 ms
    // at ef 62 and at ef 65)
    if (a.ef <= 60) return launch_cfg<NCH, GPB, 1, 2, false, GPB, false, FULL>(a, stream, err);
    if (a.ef <= 124) return launch_cfg<NCH, GPB, 2, 4, false, GPB, false, FULL>(a, stream, err);
    // From three key registers on, half the rows in flight per pass of score_nodes: at these ef a hop brings 4-8 unvisited
    // neighbours, not 16-32, and the 32 registers that frees take the body from 134-150 to 92-127: four waves per SIMD
    // instead of three (C2, 100 000 queries: ef 128 10.8 -> 8.5 ms, ef 200 15.9 -> 12.6, ef 252 19.6 -> 14.9).
    constexpr int GH = GPB >= 2 ? GPB / 2 : GPB;
    // ... unless the launch carries spill tables AND a table of 16 KB or more, i.e. earlier launches at this ef outgrew
    // the smaller ones: that is a large graph (a hop there brings 16-20 unvisited neighbours: two passes of 16), and its
    // tables bound the residency at 7-10 queries per CU whatever the registers (20 M x 128: ef 128 2.89 -> 2.69 ms, ef 160
    // 3.58 -> 3.35, ef 252 6.78 -> 6.54, ef 300 8.96 -> 8.56; 145-155 registers)
    if (GH != GPB && a.spill && a.hbits >= 12u) {
        if (a.ef <= 188) return launch_cfg<NCH, GPB, 3, 6, false, GH, true, FULL>(a, stream, err);
        if (a.ef <= 252) return launch_cfg<NCH, GPB, 4, 8, false, GH, true, FULL>(a, stream, err);
        if (a.ef <= 316) return launch_cfg<NCH, GPB, 5, 10, false, GH, true, FULL>(a, stream, err);
    }
    if (a.ef <= 188) return launch_cfg<NCH, GH, 3, 6, false, GH, false, FULL>(a, stream, err);
    if (a.ef <= 252) return launch_cfg<NCH, GH, 4, 8, false, GH, false, FULL>(a, stream, err);
    if (a.ef <= 316) return launch_cfg<NCH, GH, 5, 10, false, GH, false, FULL>(a, stream, err);
    return launch_cfg<NCH, GH, 8, 16, false, GH, false, FULL>(a, stream, err);
    }
}

// this part's share of the dispatch (SHORT: rows up to 128 floats)
template <bool FILT, bool SHORT>
int launch_part(const SearchArgs& a, uint32_t nch, hipStream_t stream, std::string* err) {
#ifdef VDB_ONLY_DIM128  // (make FAST=1: a quick build for kernel experiments on 65-128-dimensional data; never shipped)
    if (nch != 2 || !SHORT) {
        if (err) *err = "this is a FAST=1 build: dim 65-128 only";
        return 4;
    }
    if (a.g.vstride == 128u) return launch_r<2, 8, FILT, true>(a, stream, err);
    return launch_r<2, 8, FILT>(a, stream, err);
#else
    if constexpr (SHORT) {
        // (rows of exactly 64 or 128 floats: the kernels without the per-chunk tests)
        if (a.g.vstride == 64u * nch) return nch <= 1 ? launch_r<1, 8, FILT, true>(a, stream, err) : launch_r<2, 8, FILT, true>(a, stream, err);
        if (nch <= 1) return launch_r<1, 8, FILT>(a, stream, err);
        return launch_r<2, 8, FILT>(a, stream, err);
    } else {
        if (nch <= 4) return launch_r<4, 4, FILT>(a, stream, err);
        if (nch <= 8) return launch_r<8, 2, FILT>(a, stream, err);
        return launch_r<16, 1, FILT>(a, stream, err);
    }
#endif
}

}  // namespace

namespace detail {
#if VDB_PART == 0 || VDB_PART == -1
int launch_search_part0(const SearchArgs& a, uint32_t nch, hipStream_t stream, std::string* err) { return launch_part<false, true>(a, nch, stream, err); }
#endif
#if VDB_PART == 1 || VDB_PART == -1
int launch_search_part1(const SearchArgs& a, uint32_t nch, hipStream_t stream, std::string* err) { return launch_part<false, false>(a, nch, stream, err); }
#endif
#if VDB_PART == 2 || VDB_PART == -1
int launch_search_part2(const SearchArgs& a, uint32_t nch, hipStream_t stream, std::string* err) { return launch_part<true, true>(a, nch, stream, err); }
#endif
#if VDB_PART == 3 || VDB_PART == -1
int launch_search_part3(const SearchArgs& a, uint32_t nch, hipStream_t stream, std::string* err) { return launch_part<true, false>(a, nch, stream, err); }
#endif
}  // namespace detail

#if VDB_PART == 0 || VDB_PART == -1
int launch_and_bitmaps(const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t words, hipStream_t stream) {
    if (words == 0) return 0;
    hipLaunchKernelGGL(and_bitmaps_kernel, dim3((words + 255) / 256), dim3(256), 0, stream, a, b, out, words);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int launch_hnsw_search(const DevGraph& g, const float* d_queries, uint32_t nq, uint32_t k, uint32_t ef,
                       const uint32_t* d_allow, float* d_out_dist, uint64_t* d_out_label, uint32_t* d_out_stats,
                       const HnswSearchWork& work, hipStream_t stream, std::string* err) {
    if (nq == 0) return 0;
    auto unsupported = [&](const char* what) {
        if (err) *err = what;
        return 4;
    };
    if (ef == 0 || ef > kMaxEf) return unsupported("ef (= max(efSearch, k)) above 508 is not supported by the device path");
    if (g.vstride > kMaxVecStride) return unsupported("dim above 1024 is not supported by the device path");
    if (g.l0stride > kMaxL0Stride || g.ustride > kMaxL0Stride) return unsupported("M above 32 is not supported by the device path");
    if (d_allow && g.n >= (1u << 30)) return unsupported("filtered search needs fewer than 2^30 elements per index (shard it)");
    if (!work.redo || (g.n && (!work.bitmaps || work.bitmap_words < (g.n + 31u) / 32u))) {
        if (err) *err = "hnsw_search: workspace missing";
        return 1;
    }
    if (g.n && (g.entry >= g.n || g.max_level < 0)) {  // the descent would start at a row that does not exist
        if (err) *err = "hnsw_search: the graph has elements but no entry point";
        return 2;
    }
    SearchArgs a{};
    a.g = g;
    a.queries = d_queries;
    a.nq = nq;
    a.k = k;
    a.ef = ef;
    // Visited-set size.  The LDS table is what the kernel's residency leaves to a query (launch_cfg: 10-13 KB, 1 900 to
    // 2 500 nodes at the 75 % fill limit), whatever ef: a search that visits more carries on in a spill table, or is
    // re-run exactly by the large configuration.  `hbits` starts at 2^11 entries and is raised by the caller when more
    // than 2 % of a launch at this ef outgrew the table (next_hbits_bias): 2^12 entries take LDS from the residency, and
    // from 2^13 on the searches with eight key registers (ef > 316) move their visited set to HBM altogether (launch_cfg).
    // (round 1 sized the table by ef - 2^12 from ef 65 on, more by ef * maxM0 / 2: ef 100 then ran with 10 queries per CU)
    uint32_t hb = 11;
    hb += work.hbits_bias;
    if (hb > 14) hb = 14;
    a.hbits = hb;
    a.allow = d_allow;
    a.out_dist = d_out_dist;
    a.out_label = d_out_label;
    a.out_stats = d_out_stats;
    a.redo = work.redo;
    a.spill = work.spill_tables ? work.spill : nullptr;
    a.spill_tables = work.spill ? work.spill_tables : 0u;
    a.bitmaps = work.bitmaps;
    a.bitmap_words = work.bitmap_words;
    a.large_blocks = work.large_blocks ? work.large_blocks : 1u;
    a.filt_level = work.filt_level;
    a.vis_mean = work.vis_mean;
    a.ghash = work.ghash;
    a.report = work.report;
    a.eps_ord = work.eps_ord;
    a.eps_min = work.eps_min;
    const bool unbounded = d_allow && work.fail_list && work.heaps && work.heap_cap && work.heap_blocks;
    a.fail_list = unbounded ? work.fail_list : nullptr;
    a.heaps = unbounded ? work.heaps : nullptr;
    a.heap_cap = unbounded ? work.heap_cap : 0u;
    a.heap_blocks = unbounded ? work.heap_blocks : 0u;
    if (a.eps_ord != kEmptyWord && (!d_allow || k != ef)) {
        if (err) *err = "hnsw_search: the stop condition needs the filtered path and k == ef";
        return 1;
    }
    hipError_t e = hipMemsetAsync(work.redo, 0, kRedoHeader * sizeof(uint32_t), stream);
    if (e != hipSuccess) {
        if (err) *err = std::string("hipMemsetAsync: ") + hipGetErrorString(e);
        return 3;
    }
    const uint32_t nch = (g.vstride + 63u) / 64u;
    if (d_allow) return nch <= 2 ? detail::launch_search_part2(a, nch, stream, err) : detail::launch_search_part3(a, nch, stream, err);
    return nch <= 2 ? detail::launch_search_part0(a, nch, stream, err) : detail::launch_search_part1(a, nch, stream, err);
}
#endif  // VDB_PART == 0 || VDB_PART == -1

}  // namespace vdb
