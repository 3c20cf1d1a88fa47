// bf_search.hip — gfx950 kernels for BruteforceSearch::SearchKNN and the shard merge.
//
// Reference path (src/hnsw/hnsw_scratch/bruteforce.hpp:147-184): linear scan, admit on
// `dist <= lastdist`, max-heap of (dist, label) under the default pair order -> the answer is
// the k smallest (dist, label) pairs; SearchKNNCloserFirst (hnswlib.hpp:87-99) returns them
// ascending.  Distances: space_l2.hpp:1-14 / space_ip.hpp:3-18 in the tree64 summation order
// (DESIGN.md "Numerics"), bit-identical to the traversal kernels and to the oracle.
//
// Exact scan (VDB_BF_FP32):
//   bf_scan_exact   a block stages a tile of queries in LDS (tree64 register layout), every
//                   16-lane group holds one base row in registers and scores it against each
//                   staged query; pairs with dist <= tau[q] are appended to the query's
//                   candidate buffer in HBM.
//   bf_compact      one wave per query: bitonic-sorts the buffer in LDS, keeps the k smallest
//                   (dist, label), publishes tau[q] = k-th distance for the next slab.
// The base set is swept in slabs of (capacity - k) rows, so a buffer can never overflow: the
// result is exact for any data order, duplicates included.
//
//   merge_topk      shard merge: G ascending lists per query -> the k smallest
//                   (dist, label); one wave per query, rank by counting.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "bf_internal.hpp"
#include "device_index.hpp"
#include "launch.hpp"

namespace vdb {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __builtin_bit_cast(float, (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ exact scan

struct ScanArgs {
    const float* base;     // [n][vstride], rows zero-padded
    const float* queries;  // [nq][dim]
    uint32_t dim, vstride, nq, qt;  // qt = queries per block
    uint32_t row_lo, row_hi;        // slab
    int32_t metric;
    uint32_t cap;                   // entries per query buffer
    uint32_t* count;                // [nq]
    const uint32_t* tau;            // [nq] ordered-u32 threshold
    uint64_t* entries;              // [nq][cap]  ord(dist) << 32 | row
    const uint32_t* qmap;           // nullable: slot i scans query qmap[i] (re-run of flagged queries)
    const uint32_t* nq_dev;         // nullable: number of live slots, read on the device
    const uint32_t* allow;          // nullable: bit (row % 32) of word row / 32 = the filter accepts the row's label
                                    // (BaseFilterFunctor, bruteforce.hpp:158-161, :170-173)
    const uint32_t* first_allowed;  // nullable, filtered scans: the first row the filter accepts (bf_first_allowed)
    uint32_t k_first;               // the first K rows (bruteforce.hpp:154)
    uint32_t only_first_allowed;    // 1: scan that one row alone, and only when it lies beyond the first K;
                                    // 0: the sweep - skip it there, it has been taken already
};

constexpr int kScanRowsPerBlock = 256;  // 4 waves x 16 steps x 4 rows

template <int NCH>
__global__ __launch_bounds__(256) void bf_scan_exact(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_q[];  // [qt][vstride] then tau[qt]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 4, t = lane & 15;
    const uint32_t q0 = blockIdx.y * a.qt;
    const uint32_t nq_live = a.nq_dev ? *a.nq_dev : a.nq;
    if (q0 >= nq_live) return;  // block-uniform
    const uint32_t nqt = nq_live - q0 < a.qt ? nq_live - q0 : a.qt;
    uint32_t* s_tau = reinterpret_cast<uint32_t*>(s_q + (size_t)a.qt * a.vstride);

    for (uint32_t i = tid; i < nqt * a.vstride; i += 256) {
        const uint32_t qi = i / a.vstride, e = i - qi * a.vstride;
        const uint32_t src = a.qmap ? a.qmap[q0 + qi] : q0 + qi;
        s_q[i] = e < a.dim ? a.queries[(size_t)src * a.dim + e] : 0.f;
    }
    for (uint32_t i = tid; i < nqt; i += 256) s_tau[i] = a.tau[q0 + i];
    __syncthreads();

    const bool is_l2 = a.metric == 0;
    uint32_t row_lo = a.row_lo, row_hi = a.row_hi, taken = 0xFFFFFFFFu;
    if (a.first_allowed) {
        const uint32_t r0 = *a.first_allowed;
        if (a.only_first_allowed) {
            if (r0 < a.k_first || r0 >= a.row_hi) return;  // among the first K, or no row is accepted at all
            row_lo = r0;
            row_hi = r0 + 1;
        } else if (r0 >= a.k_first) {
            taken = r0;
        }
    }
    const uint32_t block_row = row_lo + blockIdx.x * kScanRowsPerBlock;
#pragma unroll 1
    for (int step = 0; step < 16; ++step) {
        const uint32_t row = block_row + (uint32_t)((wave * 16 + step) * 4 + grp);
        if (block_row + (uint32_t)((wave * 16 + step) * 4) >= row_hi) break;  // wave-uniform
        const bool live = row < row_hi && row != taken;
        f32x4 x[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            x[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live && (uint32_t)(64 * j + 4 * t) < a.vstride)
                x[j] = *reinterpret_cast<const f32x4*>(a.base + (size_t)row * a.vstride + 64 * j + 4 * t);
        }
#pragma unroll 1
        for (uint32_t qi = 0; qi < nqt; ++qi) {
            const float* qrow = s_q + (size_t)qi * a.vstride + 4 * t;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((uint32_t)(64 * j + 4 * t) < a.vstride) q = *reinterpret_cast<const f32x4*>(qrow + 64 * j);
                if (is_l2) {
                    const float d0 = q.x - x[j].x, d1 = q.y - x[j].y, d2 = q.z - x[j].z, d3 = q.w - x[j].w;
                    a0 = __builtin_fmaf(d0, d0, a0);
                    a1 = __builtin_fmaf(d1, d1, a1);
                    a2 = __builtin_fmaf(d2, d2, a2);
                    a3 = __builtin_fmaf(d3, d3, a3);
                } else {
                    a0 = __builtin_fmaf(q.x, x[j].x, a0);
                    a1 = __builtin_fmaf(q.y, x[j].y, a1);
                    a2 = __builtin_fmaf(q.z, x[j].z, a2);
                    a3 = __builtin_fmaf(q.w, x[j].w, a3);
                }
            }
            float s = (a0 + a1) + (a2 + a3);
            s = s + dpp_f32<kDppXor1>(s);
            s = s + dpp_f32<kDppXor2>(s);
            s = s + dpp_f32<kDppHalfMirror>(s);
            s = s + dpp_f32<kDppMirror>(s);
            if (!is_l2) s = 1.0f - s;
            const uint32_t od = f2ord(s);
            if (t == 0 && live && od <= s_tau[qi] &&  // bruteforce.hpp:167 `dist <= lastdist`, then the filter (:170)
                (a.allow == nullptr || ((a.allow[row >> 5] >> (row & 31u)) & 1u) != 0u)) {
                const uint32_t pos = atomicAdd(&a.count[q0 + qi], 1u);
                if (pos < a.cap) a.entries[(size_t)(q0 + qi) * a.cap + pos] = ((uint64_t)od << 32) | row;
            }
        }
    }
}

// ------------------------------------------------------------------ compaction / output

// ascending bitonic sort of p (power of two, >= 64) keys in LDS by one wave
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* s, uint32_t p, int lane) {
    for (uint32_t size = 2; size <= p; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = lane; i < (p >> 1); i += 64) {
                const uint32_t lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t x = s[lo], y = s[hi];
                if ((x > y) == up) {
                    s[lo] = y;
                    s[hi] = x;
                }
            }
            wave_sync();
        }
    }
}

__global__ __launch_bounds__(64) void bf_compact(CompactArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_key[];  // [cap]
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    if (q >= (a.nq_dev ? *a.nq_dev : a.nq)) return;
    const uint32_t k = a.k;
    uint32_t cnt = a.count[q];
    if (cnt > a.cap) {  // exact scan: cannot happen (slab + k <= cap); candidate pass: remembered
        cnt = a.cap;
        if (a.overflow && lane == 0) a.overflow[q] = 1u;
    }
    uint64_t* ent = a.entries + (size_t)q * a.cap;

    uint32_t p = 64;
    while (p < cnt) p <<= 1;
    for (uint32_t i = lane; i < p; i += 64) s_key[i] = i < cnt ? ent[i] : ~0ull;
    wave_sync();
    bitonic_sort_lds(s_key, p, lane);

    // the k smallest by (dist, row) are s_key[0..k); the reference orders ties by label
    uint32_t keep = cnt < k ? cnt : k;
    if (cnt > k) {
        const uint32_t dk = (uint32_t)(s_key[k - 1] >> 32);
        if ((uint32_t)(s_key[k] >> 32) == dk) {
            // a group of equal distances straddles the cut: keep its smallest labels
            uint32_t ga = k - 1, gb = k;
            while (ga > 0 && (uint32_t)(s_key[ga - 1] >> 32) == dk) --ga;
            while (gb < cnt && (uint32_t)(s_key[gb] >> 32) == dk) ++gb;
            const uint32_t want = k - ga;
            // rank by label inside the group; winners are rewritten in place behind ga
            // (two phases so that no lane overwrites a key another lane still reads)
            uint64_t* scratch = ent;  // the HBM buffer is free to use: its content now lives in LDS
            for (uint32_t i = ga + lane; i < gb; i += 64) {
                const uint64_t li = a.labels[(uint32_t)s_key[i]];
                uint32_t rank = 0;
                for (uint32_t j = ga; j < gb; ++j) rank += a.labels[(uint32_t)s_key[j]] < li ? 1u : 0u;
                if (rank < want) scratch[ga + rank] = s_key[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            wave_sync();
            for (uint32_t i = ga + lane; i < k; i += 64) s_key[i] = scratch[i];
            wave_sync();
        }
    }

    if (!a.final_pass) {
        for (uint32_t i = lane; i < keep; i += 64) ent[i] = s_key[i];
        if (lane == 0) {
            a.count[q] = keep;
            // lastdist = the worst result once there are k of them (bruteforce.hpp:174-180); before that it is
            // "no bound" without a filter, and with one what the first K rows left behind (tau_floor, bf_first_rows_bound)
            a.tau[q] = keep == k ? (uint32_t)(s_key[k - 1] >> 32) : (a.tau_floor ? a.tau_floor[q] : 0xFFFFFFFFu);
        }
        return;
    }
    // final: ascending (dist, label)
    const uint32_t orow = a.qmap ? a.qmap[q] : q;
    if (a.cert) {
        // a candidate pass fed this buffer: everything it dropped is at distance >= cert[q]
        bool proven = keep < k ? a.cert[q] == __builtin_inff() : ord2f((uint32_t)(s_key[k - 1] >> 32)) < a.cert[q];
        if (a.overflow && __builtin_amdgcn_readfirstlane((int)a.overflow[q]) != 0) proven = false;
        if (!proven) {
            if (lane == 0) a.redo[1 + atomicAdd(&a.redo[0], 1u)] = orow;
            return;
        }
    }
    float* od = a.out_dist + (size_t)orow * k;
    uint64_t* ol = a.out_label + (size_t)orow * k;
    for (uint32_t i = lane; i < keep; i += 64) {
        const uint32_t di = (uint32_t)(s_key[i] >> 32);
        const uint64_t li = a.labels[(uint32_t)s_key[i]];
        uint32_t pos = i;
        const bool tie = (i > 0 && (uint32_t)(s_key[i - 1] >> 32) == di) || (i + 1 < keep && (uint32_t)(s_key[i + 1] >> 32) == di);
        if (tie) {
            uint32_t lo = i;
            while (lo > 0 && (uint32_t)(s_key[lo - 1] >> 32) == di) --lo;
            pos = lo;
            for (uint32_t j = lo; j < keep && (uint32_t)(s_key[j] >> 32) == di; ++j)
                pos += a.labels[(uint32_t)s_key[j]] < li ? 1u : 0u;
        }
        od[pos] = ord2f(di);
        ol[pos] = li;
    }
    for (uint32_t i = keep + lane; i < k; i += 64) {
        od[i] = __builtin_inff();
        ol[i] = ~0ull;
    }
}

// With a filter the reference's `lastdist` starts as the worst distance among the ACCEPTED rows of the first K
// (bruteforce.hpp:154-163: `topResults.top().first`) and from then on is `topResults.top().first` after every row
// at or below it (:177-180) - also while fewer than K results exist.  When none of the first K was accepted it
// starts as the largest float, and the first accepted row of the sweep, whatever its distance, becomes the top
// and with it the bound.  Either way a later row can only ever be a result when its distance is <= that first
// bound: the answer is the K smallest (dist, label) among the accepted rows at or below it, not among all
// accepted rows.  Restated: the first K rows - and the first accepted row, when it lies beyond them - are scanned
// with an open threshold, this kernel turns what they appended into the bound, the sweep goes on.

// The first row the filter accepts (0xFFFFFFFF when none): *out must start as 0xFFFFFFFF.
__global__ __launch_bounds__(256) void bf_first_allowed(const uint32_t* allow, uint32_t n, uint32_t* out) {
    const uint32_t words = (n + 31u) / 32u;
    for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < words; w += gridDim.x * 256) {
        uint32_t v = allow[w];
        if (w == words - 1 && (n & 31u)) v &= (1u << (n & 31u)) - 1u;  // bits beyond the last row are not rows
        if (v) {
            atomicMin(out, w * 32u + (uint32_t)__builtin_ctz(v));
            return;  // this thread's later words hold later rows
        }
    }
}

__global__ __launch_bounds__(256) void bf_first_rows_bound(const uint32_t* count, const uint64_t* entries, uint32_t cap, uint32_t nq,
                                                           uint32_t* tau, uint32_t* tau_floor) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const uint32_t c = count[q] < cap ? count[q] : cap;
    uint32_t worst = 0xFFFFFFFFu;  // nothing accepted: std::numeric_limits<dist_t>::max() - everything passes
    if (c) {
        worst = 0;
        for (uint32_t i = 0; i < c; ++i) {
            const uint32_t d = (uint32_t)(entries[(size_t)q * cap + i] >> 32);
            worst = d > worst ? d : worst;
        }
    }
    tau[q] = worst;
    tau_floor[q] = worst;
}

// ------------------------------------------------------------------ shard merge

struct MergeArgs {
    const float* dist;      // [G][nq][k]
    const uint64_t* label;  // [G][nq][k]
    uint32_t G, nq, k;
    float* out_dist;        // [nq][k]
    uint64_t* out_label;
};

__global__ __launch_bounds__(64) void merge_topk_kernel(MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_m[];  // [G k] labels, then [G k] u32 ord dist
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x, total = a.G * a.k;
    uint64_t* s_lab = s_m;
    uint32_t* s_d = reinterpret_cast<uint32_t*>(s_m + total);
    for (uint32_t i = lane; i < total; i += 64) {
        const uint32_t gi = i / a.k, j = i - gi * a.k;
        const size_t src = ((size_t)gi * a.nq + q) * a.k + j;
        s_d[i] = f2ord(a.dist[src]);
        s_lab[i] = a.label[src];
    }
    wave_sync();
    for (uint32_t i = lane; i < total; i += 64) {
        const uint32_t di = s_d[i];
        const uint64_t li = s_lab[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < total; ++j) {
            const uint32_t dj = s_d[j];
            const uint64_t lj = s_lab[j];
            const bool less = dj < di || (dj == di && (lj < li || (lj == li && j < i)));
            rank += less ? 1u : 0u;
        }
        if (rank < a.k) {
            a.out_dist[(size_t)q * a.k + rank] = ord2f(di);
            a.out_label[(size_t)q * a.k + rank] = li;
        }
    }
}

template <int NCH>
void launch_scan(const ScanArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL((bf_scan_exact<NCH>), grid, dim3(256), lds, stream, a);
}

}  // namespace

hipError_t enqueue_bf_compact(const CompactArgs& c, hipStream_t stream) {
    hipLaunchKernelGGL(bf_compact, dim3(c.nq), dim3(64), (size_t)c.cap * 8, stream, c);
    return hipGetLastError();
}

size_t bf_exact_workspace_bytes(uint32_t nq) { return (size_t)nq * (8 + (size_t)kBfCap * 8) + (size_t)nq * 4 + 256; }

int launch_bf_exact(const float* d_base, const uint64_t* d_labels, uint32_t n, uint32_t dim, uint32_t vstride,
                    int metric, const float* d_queries, uint32_t nq, uint32_t k, float* d_out_dist,
                    uint64_t* d_out_label, void* workspace, hipStream_t stream, std::string* err,
                    const uint32_t* d_qmap, const uint32_t* d_nq_live, const uint32_t* d_allow) {
    if (nq == 0) return 0;
    if (k == 0 || k > kBfMaxK) {
        if (err) *err = "brute force: k must be in 1..1024";
        return 4;
    }
    if (vstride > kMaxVecStride) {
        if (err) *err = "dim above 1024 is not supported by the device path";
        return 4;
    }
    uint32_t* count = static_cast<uint32_t*>(workspace);
    uint32_t* tau = count + nq;
    uint64_t* entries = reinterpret_cast<uint64_t*>(tau + nq);  // nq*8 bytes in: 8-byte aligned
    hipError_t e = hipMemsetAsync(count, 0, (size_t)nq * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(tau, 0xFF, (size_t)nq * 4, stream);

    ScanArgs s{};
    s.base = d_base;
    s.queries = d_queries;
    s.dim = dim;
    s.vstride = vstride;
    s.nq = nq;
    uint32_t qt = (uint32_t)((48u * 1024u) / ((size_t)vstride * 4));
    if (qt > 64) qt = 64;
    if (qt < 1) qt = 1;
    s.qt = qt;
    s.metric = metric;
    s.cap = kBfCap;
    s.count = count;
    s.tau = tau;
    s.entries = entries;
    s.qmap = d_qmap;
    s.nq_dev = d_nq_live;
    s.allow = d_allow;
    uint32_t* tau_floor = reinterpret_cast<uint32_t*>(entries + (size_t)nq * kBfCap);
    uint32_t* first_allowed = tau_floor + nq;
    if (d_allow) {
        if (e == hipSuccess) e = hipMemsetAsync(first_allowed, 0xFF, 4, stream);
        if (e == hipSuccess) {
            const uint32_t words = (n + 31u) / 32u;
            const uint32_t blocks = words / 256u + 1u < 1024u ? words / 256u + 1u : 1024u;
            hipLaunchKernelGGL(bf_first_allowed, dim3(blocks), dim3(256), 0, stream, d_allow, n, first_allowed);
            e = hipGetLastError();
        }
        s.first_allowed = first_allowed;
        s.k_first = k;
    }
    const size_t scan_lds = (size_t)qt * vstride * 4 + (size_t)qt * 4;

    CompactArgs c{};
    c.nq = nq;
    c.k = k;
    c.cap = kBfCap;
    c.count = count;
    c.tau = tau;
    c.entries = entries;
    c.labels = d_labels;
    c.out_dist = d_out_dist;
    c.out_label = d_out_label;
    c.qmap = d_qmap;
    c.nq_dev = d_nq_live;
    c.tau_floor = d_allow ? tau_floor : nullptr;

    const uint32_t slab = kBfCap - k;
    const uint32_t nch = (vstride + 63u) / 64u;
    auto scan = [&](uint32_t lo, uint32_t hi) {
        s.row_lo = lo;
        s.row_hi = hi;
        const dim3 grid(s.only_first_allowed ? 1u : (hi - lo + kScanRowsPerBlock - 1) / kScanRowsPerBlock, (nq + qt - 1) / qt);
        if (nch <= 1) launch_scan<1>(s, grid, scan_lds, stream);
        else if (nch <= 2) launch_scan<2>(s, grid, scan_lds, stream);
        else if (nch <= 4) launch_scan<4>(s, grid, scan_lds, stream);
        else if (nch <= 8) launch_scan<8>(s, grid, scan_lds, stream);
        else launch_scan<16>(s, grid, scan_lds, stream);
        return hipGetLastError();
    };
    uint32_t lo = 0;
    bool first_rows = d_allow != nullptr;  // filtered: the first K rows alone, then their bound (bf_first_rows_bound)
    do {
        uint32_t hi = n - lo > slab ? lo + slab : n;
        if (first_rows) hi = n < k ? n : k;
        if (hi > lo && e == hipSuccess) e = scan(lo, hi);
        lo = hi;
        c.final_pass = lo >= n ? 1 : 0;
        if (first_rows && !c.final_pass) {
            first_rows = false;
            if (e == hipSuccess) {  // the first accepted row, when the first K hold none (one block of rows, row read on the device)
                s.only_first_allowed = 1;
                e = scan(0, n);
                s.only_first_allowed = 0;
            }
            if (e == hipSuccess) {
                hipLaunchKernelGGL(bf_first_rows_bound, dim3((nq + 255) / 256), dim3(256), 0, stream, count, entries, (uint32_t)kBfCap, nq,
                                   tau, tau_floor);
                e = hipGetLastError();
            }
            continue;  // nothing to compact: at most K entries so far
        }
        first_rows = false;
        if (e == hipSuccess) {
            e = enqueue_bf_compact(c, stream);
        }
    } while (lo < n && e == hipSuccess);
    if (e != hipSuccess) {
        if (err) *err = std::string("brute-force launch failed: ") + hipGetErrorString(e);
        return 3;
    }
    return 0;
}

int launch_merge_topk(const float* d_dist, const uint64_t* d_label, uint32_t n_lists, uint32_t nq, uint32_t k,
                      float* d_out_dist, uint64_t* d_out_label, hipStream_t stream, std::string* err) {
    if (nq == 0 || k == 0) return 0;
    const size_t lds = (size_t)n_lists * k * 12;
    if (n_lists == 0 || lds > 64 * 1024) {
        if (err) *err = "merge: n_lists * k must be in 1..5461";
        return 1;
    }
    MergeArgs a{d_dist, d_label, n_lists, nq, k, d_out_dist, d_out_label};
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(64), lds, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("merge launch failed: ") + hipGetErrorString(e);
        return 3;
    }
    return 0;
}

}  // namespace vdb
