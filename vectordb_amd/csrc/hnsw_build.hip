// hnsw_build.hip — layer-0 graph construction on the device.
//
// Reference: addPoint hnsw_core.hpp:994-1104 = searchBaseLayer (:293-356) at ef_construction,
// getNeighborsByHeuristic2 (:357-390), mutuallyConnectNewElement (:392-498).  The reference inserts
// one point at a time under per-node locks.  Here a BATCH of points is inserted per round with no
// locks, every step a data-parallel kernel:
//
//   1. search      the traversal kernel of hnsw_search.hip (ef = k = ef_construction) over the graph
//                  as it stands, one wave per new point: its ef_construction nearest candidates
//   2. select      one wave per new point: the heuristic picks <= M of them (a candidate is kept when
//                  it is closer to the new point than to every neighbour kept so far); the new
//                  point's own list is written, and one (neighbour, new point) pair per link emitted
//   3. sort        the pairs by neighbour (rocPRIM radix sort)
//   4. reverse     one wave per neighbour that gained links: append while the list has room, else the
//                  same heuristic over (old list + newcomers) as seen from that neighbour (:428-468)
//
// Points of one batch do not see each other in step 1 (they meet through step 4), so a batch is kept
// small against the graph it is added to (the host grows the batch with the graph).  The result is a
// valid HNSW layer 0 but not the reference's graph bit for bit: insertion order is part of the
// reference's result and a lock-free batch has none.  It is therefore gated on recall and integrity
// (tests/test_build_gpu.py), as SURVEY.md §8f prescribes.  Upper layers (1/16 of the points) are
// built by the host builder first, with the reference algorithm.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>  // rocPRIM's headers call memset without including it
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "device_index.hpp"
#include "launch.hpp"
#include "traversal_common.hpp"

namespace vdb {

namespace {

using namespace dev;

// out[i][0..dim) = vecs[ids[i]][0..dim): the new points as a query batch
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* vecs, uint32_t vstride, uint32_t dim, const uint32_t* ids,
                                                          uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* src = vecs + (size_t)ids[i] * vstride;
    for (uint32_t c = threadIdx.x & 63; c < dim; c += 64) out[(size_t)i * dim + c] = src[c];
}

// A base row in the register layout score_nodes wants for its query (every 16-lane group a copy).
template <int NCH>
__device__ __forceinline__ void load_row_as_query(const DevGraph& g, uint32_t id, int lane, f32x4 (&q)[NCH]) {
    const int t = lane & 15;
    const float* row = g.vecs + (size_t)id * g.vstride + 4 * t;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        q[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((uint32_t)(64 * j + 4 * t) < g.vstride) q[j] = *reinterpret_cast<const f32x4*>(row + 64 * j);
    }
}

// getNeighborsByHeuristic2 (:357-390) over candidates sorted closest first (c_id / c_d in LDS, c_d =
// ordered distance to the centre).  Returns the number kept; lane j < kept holds the j-th kept id.
template <int NCH, int GPB, bool IS_L2>
__device__ __forceinline__ uint32_t select_heuristic(const DevGraph& g, const uint32_t* c_id, const uint32_t* c_d, uint32_t n_c,
                                                     uint32_t m, int lane, uint32_t& keep_id) {
    keep_id = kNoId;
    uint32_t nk = 0;
#pragma unroll 1
    for (uint32_t i = 0; i < n_c && nk < m; ++i) {
        const uint32_t c = c_id[i], cd = c_d[i];
        bool good = true;
        if (nk) {
            f32x4 qc[NCH];
            load_row_as_query<NCH>(g, c, lane, qc);
            uint32_t d = score_nodes<NCH, GPB, IS_L2>(g, qc, keep_id, nk, lane);
            good = ballot((uint32_t)lane < nk && d < cd) == 0ull;  // :375-381: closer to a kept neighbour than to the centre
        }
        if (good) {
            if ((uint32_t)lane == nk) keep_id = c;
            nk += 1;
        }
    }
    return nk;
}

struct SelectArgs {
    DevGraph g;
    uint32_t* links0;            // the graph's layer-0 lists, written here
    const uint64_t* cand_label;  // [nb][efc] internal ids (the search ran with identity labels), closest first
    const float* cand_dist;      // [nb][efc]
    const uint32_t* new_ids;     // [nb]
    uint32_t nb, efc, M;
    uint32_t* pair_key;          // [nb * M] neighbour that gains a link (all ones: none)
    uint32_t* pair_val;          // [nb * M] the new point
};

template <int NCH, int GPB, bool IS_L2>
__global__ __launch_bounds__(64) void build_select_kernel(SelectArgs a) {
    __shared__ uint32_t c_id[kMaxEf], c_d[kMaxEf];
    const int lane = lane_id();
    const uint32_t b = blockIdx.x;
    const uint32_t me = a.new_ids[b];
    uint32_t n_c = 0;
    for (uint32_t i0 = 0; i0 < a.efc; i0 += 64) {
        const uint32_t i = i0 + lane;
        const uint64_t lab = i < a.efc ? a.cand_label[(size_t)b * a.efc + i] : ~0ull;
        const bool ok = lab != ~0ull;
        if (ok) {
            c_id[i] = (uint32_t)lab;
            c_d[i] = f2ord(a.cand_dist[(size_t)b * a.efc + i]);
        }
        n_c += (uint32_t)__builtin_popcountll(ballot(ok));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t keep_id;
    const uint32_t nk = select_heuristic<NCH, GPB, IS_L2>(a.g, c_id, c_d, n_c, a.M, lane, keep_id);
    // own list, farthest first / closest last as the reference drains its heap (:404-420)
    const uint32_t src = nk ? (nk - 1u - (uint32_t)lane) & 63u : 0u;
    const uint32_t v = bpermute(keep_id, (int)src);
    if ((uint32_t)lane < a.g.l0stride) a.links0[(size_t)me * a.g.l0stride + lane] = (uint32_t)lane < nk ? v : kNoId;
    if ((uint32_t)lane < a.M) {
        a.pair_key[(size_t)b * a.M + lane] = (uint32_t)lane < nk ? keep_id : kEmptyWord;
        a.pair_val[(size_t)b * a.M + lane] = me;
    }
}

struct ReverseArgs {
    DevGraph g;
    uint32_t* links0;
    const uint32_t* key;  // sorted: neighbour ids, all-ones at the end
    const uint32_t* val;  // the new points, in step with key
    uint32_t n_pairs, maxM0;
};

// One wave per run of equal keys (the block at the head of the run does the work).
// The pool of a full list is the old list (up to maxM0 <= 64 ids) plus up to 32 newcomers: at most
// 96 entries, kept two per lane (entry i in lane i % 64, register i / 64).
template <int NCH, int GPB, bool IS_L2>
__global__ __launch_bounds__(64) void build_reverse_kernel(ReverseArgs a) {
    constexpr uint32_t kPoolCap = 128;
    __shared__ uint32_t c_id[kPoolCap], c_d[kPoolCap];
    __shared__ uint64_t s_key[kPoolCap];
    const int lane = lane_id();
    const uint32_t i = blockIdx.x;
    const uint32_t nb = a.key[i];
    if (nb == kEmptyWord || (i > 0 && a.key[i - 1] == nb)) return;
    // newcomers of this neighbour: at most 32 are taken (a hub that gains more in one batch keeps the
    // first 32 in sort order; the others still link TO it)
    const bool mine = (uint32_t)lane < 32u && i + lane < a.n_pairs && a.key[i + lane] == nb;
    const uint64_t mm = ballot(mine);
    const uint32_t t = (uint32_t)__builtin_ctzll(~mm);  // length of the run of ones from lane 0
    const uint32_t add = (uint32_t)lane < t ? a.val[i + lane] : kNoId;
    const uint32_t ls = a.g.l0stride;
    const uint32_t cur = (uint32_t)lane < ls ? a.links0[(size_t)nb * ls + lane] : kNoId;
    const uint32_t c = (uint32_t)__builtin_popcountll(ballot(cur != kNoId));  // lists are packed to the front
    if (c + t <= a.maxM0) {  // room: append (:424-427)
        if ((uint32_t)lane < t) a.links0[(size_t)nb * ls + c + lane] = add;
        return;
    }
    // full: the heuristic over old list + newcomers, seen from nb (:428-468)
    const uint32_t n_pool = c + t;  // c <= 64, t <= 32
    const uint32_t sh0 = bpermute(add, (lane - (int)c) & 63);       // entry lane      = add[lane - c]
    const uint32_t sh1 = bpermute(add, (lane + 64 - (int)c) & 63);  // entry 64 + lane = add[64 + lane - c]
    const uint32_t pool0 = (uint32_t)lane < c ? cur : ((uint32_t)lane < n_pool ? sh0 : kNoId);
    const uint32_t pool1 = 64u + (uint32_t)lane < n_pool ? sh1 : kNoId;
    f32x4 qn[NCH];
    load_row_as_query<NCH>(a.g, nb, lane, qn);
    uint32_t d0 = kEmptyWord, d1 = kEmptyWord;  // ordered distances of pool0 / pool1 to nb
    {
        const uint32_t m0 = n_pool < 32u ? n_pool : 32u;
        const uint32_t r0 = score_nodes<NCH, GPB, IS_L2>(a.g, qn, pool0, m0, lane);
        if ((uint32_t)lane < m0) d0 = r0;
        if (n_pool > 32u) {
            const uint32_t m1 = n_pool - 32u < 32u ? n_pool - 32u : 32u;
            const uint32_t ids_hi = bpermute(pool0, (lane + 32) & 63);
            const uint32_t r1 = score_nodes<NCH, GPB, IS_L2>(a.g, qn, ids_hi, m1, lane);
            const uint32_t r1s = bpermute(r1, (lane - 32) & 63);
            if ((uint32_t)lane >= 32u && (uint32_t)lane < 32u + m1) d0 = r1s;
        }
        if (n_pool > 64u) {
            const uint32_t m2 = n_pool - 64u;
            const uint32_t r2 = score_nodes<NCH, GPB, IS_L2>(a.g, qn, pool1, m2, lane);
            if ((uint32_t)lane < m2) d1 = r2;
        }
    }
    // sort the pool closest first (rank by counting; equal distances: larger id first, :363-366)
    const uint64_t k0 = ((uint64_t)d0 << 32) | (uint32_t)~pool0;
    const uint64_t k1 = ((uint64_t)d1 << 32) | (uint32_t)~pool1;
    s_key[lane] = (uint32_t)lane < n_pool ? k0 : ~0ull;
    s_key[64 + lane] = 64u + (uint32_t)lane < n_pool ? k1 : ~0ull;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t rank0 = 0, rank1 = 0;
    for (uint32_t j = 0; j < n_pool; ++j) {
        const uint64_t o = s_key[j];
        rank0 += o < k0 ? 1u : 0u;
        rank1 += o < k1 ? 1u : 0u;
    }
    if ((uint32_t)lane < n_pool) {
        c_id[rank0] = pool0;
        c_d[rank0] = d0;
    }
    if (64u + (uint32_t)lane < n_pool) {
        c_id[rank1] = pool1;
        c_d[rank1] = d1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t keep_id;
    const uint32_t nk = select_heuristic<NCH, GPB, IS_L2>(a.g, c_id, c_d, n_pool, a.maxM0, lane, keep_id);
    const uint32_t src = nk ? (nk - 1u - (uint32_t)lane) & 63u : 0u;
    const uint32_t v = bpermute(keep_id, (int)src);
    if ((uint32_t)lane < ls) a.links0[(size_t)nb * ls + lane] = (uint32_t)lane < nk ? v : kNoId;
}

// ---- orphans.  The reference's checkIntegrity (:1235-1241) demands an inbound link for every element.  The heuristic
// can prune the last one away - in the reference too (33 of 2 000 elements of a clustered set at M = 4, host build),
// and a batch makes it likelier (up to 32 newcomers compete for one full list in a single pass).  After the last
// round an element nobody links to on layer 0 is given one link FROM an element near it.  Tried in this order, each
// closest first: (1) an element of its own list with room - the link is appended; (2) a neighbour of one of those, with
// room; (3) an element of its own list that is full - the link replaces the entry most others link to as well (two at
// least, so nothing is orphaned in turn); (4) the same among the neighbours' neighbours.  A list gives up at most a
// quarter of its entries this way (one at least): many orphans share one nearest hub, and without the limit they took
// its whole list - the hub, a popular landing point of the upper layers, then led nowhere but to them (M = 4, 2 000
// rows: one build in twenty lost 0.09 of recall@10, tests/fuzz_build.py).  Whoever finds no taker stays as the
// heuristic left it, as in the reference.
__global__ __launch_bounds__(256) void count_inbound_kernel(const uint32_t* links0, size_t words, uint32_t n, uint32_t* cnt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
        const uint32_t v = links0[i];
        if (v < n) atomicAdd(&cnt[v], 1u);
    }
}

__global__ __launch_bounds__(256) void list_orphans_kernel(const uint32_t* cnt, uint32_t n, uint32_t* list, uint32_t cap, uint32_t* n_found) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (cnt[i] == 0u) {
            const uint32_t slot = atomicAdd(n_found, 1u);
            if (slot < cap) list[slot] = (uint32_t)i;
        }
    }
}

// One wave, the orphans one after the other in ascending id order (the host sorts the list): a repair sees the
// lists and counts the repairs before it left, and the outcome does not depend on scheduling.
// `given[q]` = entries q's list has lost to repairs so far (zeroed by the caller).
__global__ __launch_bounds__(64) void repair_orphans_kernel(uint32_t* links0, uint32_t ls, uint32_t maxM0, uint32_t* cnt, uint32_t* given,
                                                            const uint32_t* list, uint32_t n_list, uint32_t* n_fixed) {
    const int lane = lane_id();
    auto ld = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const uint32_t give_max = maxM0 / 4u > 1u ? maxM0 / 4u : 1u;
    uint32_t fixed = 0;
#pragma unroll 1
    for (uint32_t w = 0; w < n_list; ++w) {
        const uint32_t p = list[w];
        if (ld(cnt + p) != 0u) continue;
        // q takes the link to p if it can (true when it did): by appending, or (replace) in place of an entry
        auto offer = [&](uint32_t q, bool replace) {
            const uint32_t theirs = (uint32_t)lane < ls ? ld(links0 + (size_t)q * ls + lane) : kNoId;
            const uint32_t c = (uint32_t)__builtin_popcountll(ballot(theirs != kNoId));  // lists are packed to the front
            if (!replace) {
                if (c >= maxM0) return false;
                if (lane == 0) st(links0 + (size_t)q * ls + c, p);
                return true;
            }
            if (c < maxM0 || ld(given + q) >= give_max) return false;
            const uint32_t their_cnt = theirs != kNoId ? ld(cnt + theirs) : 0u;
            // the entry with the most inbound links (>= 2), the earlier one among equals
            const uint32_t key = their_cnt >= 2u ? ~((their_cnt < 0x3FFFFFFu ? their_cnt : 0x3FFFFFFu) << 6 | (uint32_t)(63 - lane)) : kEmptyWord;
            const uint32_t best = wave_min_u32(key);
            if (best == kEmptyWord) return false;
            if (key == best) {
                st(cnt + theirs, their_cnt - 1u);
                st(links0 + (size_t)q * ls + lane, p);
                st(given + q, ld(given + q) + 1u);
            }
            return true;
        };
        const uint32_t mine = (uint32_t)lane < ls ? ld(links0 + (size_t)p * ls + lane) : kNoId;
        const uint32_t np = (uint32_t)__builtin_popcountll(ballot(mine != kNoId));
        bool done = false;
#pragma unroll 1
        for (int pass = 0; pass < 4 && !done; ++pass) {
            const bool replace = pass >= 2, two_hops = pass & 1;
#pragma unroll 1
            for (uint32_t j = np; j-- > 0 && !done;) {  // closest last
                const uint32_t q = readlane(mine, (int)j);
                if (!two_hops) {
                    done = offer(q, replace);
                    continue;
                }
                const uint32_t theirs = (uint32_t)lane < ls ? ld(links0 + (size_t)q * ls + lane) : kNoId;
                const uint32_t c = (uint32_t)__builtin_popcountll(ballot(theirs != kNoId));
#pragma unroll 1
                for (uint32_t i = c; i-- > 0 && !done;) {
                    const uint32_t r = bpermute(theirs, (int)i);
                    if (r != p) done = offer(r, replace);
                }
            }
        }
        if (done) {
            if (lane == 0) st(cnt + p, 1u);
            fixed += 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    }
    if (lane == 0) *n_fixed = fixed;
}

template <int NCH, int GPB>
void launch_select(const SelectArgs& a, hipStream_t s) {
    if (a.g.metric == 0) hipLaunchKernelGGL((build_select_kernel<NCH, GPB, true>), dim3(a.nb), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((build_select_kernel<NCH, GPB, false>), dim3(a.nb), dim3(64), 0, s, a);
}
template <int NCH, int GPB>
void launch_reverse(const ReverseArgs& a, hipStream_t s) {
    if (a.g.metric == 0) hipLaunchKernelGGL((build_reverse_kernel<NCH, GPB, true>), dim3(a.n_pairs), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((build_reverse_kernel<NCH, GPB, false>), dim3(a.n_pairs), dim3(64), 0, s, a);
}

}  // namespace

size_t build_sort_temp_bytes(uint32_t max_pairs) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, (size_t)max_pairs, 0, 32, nullptr);
    return bytes;
}

int launch_gather_rows(const float* d_vecs, uint32_t vstride, uint32_t dim, const uint32_t* d_ids, uint32_t n, float* d_out,
                       hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, d_vecs, vstride, dim, d_ids, n, d_out);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int build_repair_orphans(uint32_t* d_links0, uint32_t n, uint32_t l0stride, uint32_t maxM0, hipStream_t stream,
                         uint32_t* found, uint32_t* fixed, std::string* err) {
    if (found) *found = 0;
    if (fixed) *fixed = 0;
    if (n < 2) return 0;
    constexpr uint32_t kCap = 1u << 20;
    uint32_t *cnt = nullptr, *given = nullptr, *list = nullptr, *hdr = nullptr;
    auto fail = [&](hipError_t e) {
        if (err) *err = std::string("device graph build (orphan repair): ") + hipGetErrorString(e);
        (void)hipFree(cnt);
        (void)hipFree(given);
        (void)hipFree(list);
        (void)hipFree(hdr);
        return 3;
    };
    hipError_t e = hipMalloc((void**)&cnt, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&given, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&list, (size_t)kCap * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&hdr, 8);
    if (e == hipSuccess) e = hipMemsetAsync(cnt, 0, (size_t)n * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(given, 0, (size_t)n * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(hdr, 0, 8, stream);
    if (e != hipSuccess) return fail(e);
    const size_t words = (size_t)n * l0stride;
    const uint32_t blocks = (uint32_t)std::min<size_t>((words + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(count_inbound_kernel, dim3(blocks), dim3(256), 0, stream, d_links0, words, n, cnt);
    hipLaunchKernelGGL(list_orphans_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 256 * 64)), dim3(256), 0, stream, cnt, n, list, kCap, hdr);
    uint32_t h[2] = {0, 0};
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, hdr, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(e);
    const uint32_t n_list = h[0] < kCap ? h[0] : kCap;
    if (found) *found = h[0];
    if (n_list) {
        std::vector<uint32_t> ids(n_list);
        e = hipMemcpy(ids.data(), list, (size_t)n_list * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(e);
        std::sort(ids.begin(), ids.end());  // (the list was filled through a counter: its order is the waves' arrival)
        e = hipMemcpy(list, ids.data(), (size_t)n_list * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(e);
        hipLaunchKernelGGL(repair_orphans_kernel, dim3(1), dim3(64), 0, stream, d_links0, l0stride, maxM0, cnt, given, list, n_list, hdr + 1);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h, hdr, 8, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return fail(e);
        if (fixed) *fixed = h[1];
    }
    (void)hipFree(cnt);
    (void)hipFree(given);
    (void)hipFree(list);
    (void)hipFree(hdr);
    return 0;
}

int launch_build_round(const DevGraph& g, uint32_t* d_links0, const uint64_t* d_cand_label, const float* d_cand_dist,
                       const uint32_t* d_new_ids, uint32_t nb, uint32_t efc, uint32_t M, uint32_t maxM0, const BuildWork& w,
                       hipStream_t stream, std::string* err) {
    if (nb == 0) return 0;
    if (efc > kMaxEf || M > 32 || maxM0 > 64 || g.l0stride > kMaxL0Stride || maxM0 > g.l0stride) {
        if (err) *err = "device graph build: ef_construction above 508 or M above 32 is not supported";
        return 4;
    }
    const uint32_t n_pairs = nb * M;
    SelectArgs sa{g, d_links0, d_cand_label, d_cand_dist, d_new_ids, nb, efc, M, w.key_in, w.val_in};
    const uint32_t nch = (g.vstride + 63u) / 64u;
    if (nch <= 1) launch_select<1, 8>(sa, stream);
    else if (nch <= 2) launch_select<2, 8>(sa, stream);
    else if (nch <= 4) launch_select<4, 4>(sa, stream);
    else if (nch <= 8) launch_select<8, 2>(sa, stream);
    else launch_select<16, 1>(sa, stream);
    hipError_t e = hipGetLastError();
    size_t temp = w.sort_temp_bytes;
    if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(w.sort_temp, temp, w.key_in, w.key_out, w.val_in, w.val_out, (size_t)n_pairs, 0, 32, stream);
    if (e == hipSuccess) {
        ReverseArgs ra{g, d_links0, w.key_out, w.val_out, n_pairs, maxM0};
        if (nch <= 1) launch_reverse<1, 8>(ra, stream);
        else if (nch <= 2) launch_reverse<2, 8>(ra, stream);
        else if (nch <= 4) launch_reverse<4, 4>(ra, stream);
        else if (nch <= 8) launch_reverse<8, 2>(ra, stream);
        else launch_reverse<16, 1>(ra, stream);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        if (err) *err = std::string("device graph build failed: ") + hipGetErrorString(e);
        return 3;
    }
    return 0;
}

}  // namespace vdb
