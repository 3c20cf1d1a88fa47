// traversal_common.hpp — device code shared by the search kernels (hnsw_search.hip) and the graph
// builder kernels (hnsw_build.hip): lane primitives, the ordered-float encoding, and the tree64
// distance engine (space_l2.hpp:1-14 / space_ip.hpp:3-18 in the summation order of DESIGN.md
// "Numerics").  Everything here is `static`/inline device code: include it inside namespace vdb.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_index.hpp"

namespace vdb {
namespace dev {

constexpr uint32_t kEmptyWord = 0xFFFFFFFFu;

// ------------------------------------------------------------------ lane primitives

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // lane i <- lane 7-i inside each 8
constexpr int kDppMirror = 0x140;      // lane i <- lane 15-i inside each 16
constexpr int kDppWaveShr1 = 0x138;    // lane i <- lane i-1 over the whole wave
constexpr int kDppWaveShl1 = 0x130;    // lane i <- lane i+1 over the whole wave

__device__ __forceinline__ uint32_t readlane(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint32_t bpermute(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// float bits -> unsigned with the same order (and back).  -0.0 never occurs (sums of
// squares start from +0; 1 - x rounds to +0), NaN inputs are outside the contract.
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __builtin_bit_cast(float, (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// minimum of v over the 64 lanes, returned wave-uniform: xor butterfly inside each row of 16
// (DPP), then the four rows through SGPRs
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    uint32_t o;
    o = dpp_u32<kDppXor1>(v, v);
    v = o < v ? o : v;
    o = dpp_u32<kDppXor2>(v, v);
    v = o < v ? o : v;
    o = dpp_u32<kDppHalfMirror>(v, v);
    v = o < v ? o : v;
    o = dpp_u32<kDppMirror>(v, v);
    v = o < v ? o : v;
    const uint32_t a = readlane(v, 0), b = readlane(v, 16), c = readlane(v, 32), d = readlane(v, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// ------------------------------------------------------------------ distance engine

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Scores up to 64 nodes (the search kernels pass at most 32 at a time, the builder's heuristic up to
// maxM0 <= 64 kept neighbours).  In: lane r (< m) holds node id ids[r].  Out: lane r (< m) holds
// the ordered-u32 distance of node r; the other lanes hold garbage.
// NCH = number of 64-float chunks per row that are touched; GPB = 16-lane groups whose
// loads are issued before the first FMA (4 nodes each).
// (returns the tree sum itself: `1 - s` of the inner product and the ordered-integer form are applied once per pass by the
// caller, to the value each lane ends up with - the same function of the same float, 4-5 instructions per group fewer)
using f32x2 = __attribute__((ext_vector_type(2))) float;
// a - b on two floats at once (the compiler scalarises a vector subtraction; it finds this instruction only when it
// vectorises scalar code itself, and then pairs elements 0, 2 and 1, 3: six v_mov per group of rows to build the pairs)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Packed arithmetic on the register pairs the row loads deliver ((x, y) and (z, w) of every 16-B piece): per 64-float chunk
// two v_pk_add_f32 and two v_pk_fma_f32 (L2), then three scalar adds - 11 vector instructions per group of four rows at
// 128 dimensions where the compiler's own vectorisation of the scalar form took 16.  Same operations, same order: a_i
// accumulates element i of every piece by fma, the lane's sum is (a0 + a1) + (a2 + a3).
template <int NCH, bool IS_L2>
__device__ __forceinline__ float group_distance(const f32x4 (&q)[NCH], const f32x4 (&x)[NCH]) {
    f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const f32x2 ql = __builtin_shufflevector(q[j], q[j], 0, 1), qh = __builtin_shufflevector(q[j], q[j], 2, 3);
        const f32x2 xl = __builtin_shufflevector(x[j], x[j], 0, 1), xh = __builtin_shufflevector(x[j], x[j], 2, 3);
        if constexpr (IS_L2) {
            const f32x2 d01 = pk_sub(ql, xl), d23 = pk_sub(qh, xh);
            a01 = __builtin_elementwise_fma(d01, d01, a01);
            a23 = __builtin_elementwise_fma(d23, d23, a23);
        } else {
            a01 = __builtin_elementwise_fma(ql, xl, a01);
            a23 = __builtin_elementwise_fma(qh, xh, a23);
        }
    }
    // pairwise tree over the 64 virtual lanes: 2 levels in-lane, 4 across lanes
    float s01 = a01.x + a01.y;
    const float s23 = a23.x + a23.y;
    asm("" : "+v"(s01));  // (keeps the two adds scalar: packed, they cost three v_mov to pair a0 with a2 and a1 with a3)
    float s = s01 + s23;
    s = s + dpp_f32<kDppXor1>(s);
    s = s + dpp_f32<kDppXor2>(s);
    s = s + dpp_f32<kDppHalfMirror>(s);
    s = s + dpp_f32<kDppMirror>(s);
    return s;
}

// FULL: the rows are exactly NCH chunks long (vstride == 64 NCH: 64, 128, 256, ... floats) - every lane loads every chunk and
// the per-chunk tests of the general form (14 of its 25 instructions per group of four rows) are not compiled.
template <int NCH, int GPB, bool IS_L2, bool FULL = false>
__device__ __forceinline__ uint32_t score_nodes(const DevGraph& g, const f32x4 (&q)[NCH], uint32_t ids, uint32_t m,
                                                int lane) {
    const int grp = lane >> 4, t = lane & 15;
    const uint32_t vstride = g.vstride;
    uint32_t out = kEmptyWord;
    const uint32_t ngroups = (m + 3u) >> 2;
#pragma unroll 1
    for (uint32_t base = 0; base < ngroups; base += GPB) {
        // node ids of every group first (independent cross-lane reads, one wait), then all the
        // row loads back to back, then the arithmetic
        uint32_t nid[GPB];
#pragma unroll
        for (int b = 0; b < GPB; ++b) nid[b] = bpermute(ids, (int)((4u * (base + b) + (uint32_t)grp) & 63u));
        f32x4 x[GPB][NCH];
        if constexpr (FULL) {
#pragma unroll
            for (int b = 0; b < GPB; ++b) {
#pragma unroll
                for (int j = 0; j < NCH; ++j) asm volatile("" : "=v"(x[b][j]));  // (as below: rows beyond m may hold anything)
                if (4u * (base + b) + (uint32_t)grp < m) {
                    const float* row = g.vecs + (size_t)nid[b] * (64u * NCH) + 4 * t;
#pragma unroll
                    for (int j = 0; j < NCH; ++j) x[b][j] = *reinterpret_cast<const f32x4*>(row + 64 * j);
                }
            }
        } else {
#pragma unroll
        for (int b = 0; b < GPB; ++b) {
            // Rows beyond m are never loaded and their lanes' results never used (every caller masks lanes >= m), so
            // their registers may hold anything: the empty asm makes them "written" without the 4 NCH v_mov per group
            // that zero-filling costs on every call (64 of the ~430 vector instructions of a hop at d = 128).
            // (chunks that reach past the row length - NCH is a power of two, the row a multiple of 32 floats - are
            // not, or not fully, loaded in real rows either: those start from zero; wave-uniform per chunk)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                asm volatile("" : "=v"(x[b][j]));
                if ((uint32_t)(64 * (j + 1)) > vstride) x[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (4u * (base + b) + (uint32_t)grp < m) {
                const float* row = g.vecs + (size_t)nid[b] * vstride + 4 * t;
#pragma unroll
                for (int j = 0; j < NCH; ++j)
                    if ((uint32_t)(64 * j + 4 * t) < vstride) x[b][j] = *reinterpret_cast<const f32x4*>(row + 64 * j);
            }
        }
        }
        // Group b's result (the same in its 16 lanes) is kept by lane t == b of each group ...
        float keep = 0.f;
#pragma unroll
        for (int b = 0; b < GPB; ++b) {
            if (base + b < ngroups) {  // wave-uniform
                const float s0 = group_distance<NCH, IS_L2>(q, x[b]);
                if (t == b) keep = s0;
            }
        }
        // ... and node r = 4 (base + b) + g is handed to lane r in one cross-lane read
        const uint32_t rl = (uint32_t)lane - 4u * base;  // r relative to this batch
        float v = __builtin_bit_cast(float, bpermute(__builtin_bit_cast(uint32_t, keep), (int)((16u * (rl & 3u) + (rl >> 2)) & 63u)));
        if constexpr (!IS_L2) v = 1.0f - v;
        if (rl < 4u * GPB) out = f2ord(v);
    }
    return out;
}

}  // namespace dev
}  // namespace vdb
