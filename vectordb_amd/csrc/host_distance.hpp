// host_distance.hpp — host-side distance functions used while BUILDING a graph.
//
// Two summation orders (include/vdb_index.h, enum vdb_sum_order):
//   seq     the reference's loop: one sequential fp32 sum, multiply and add separate
//           (reference space_l2.hpp:5-11, space_ip.hpp:6-9).  Built with -ffp-contract=off.
//   tree64  the order the gfx950 kernels sum in (DESIGN.md "Numerics"): 64 virtual lanes,
//           lane v folds elements v, v+64, … with fused multiply-adds, then a pairwise
//           neighbour tree over the 64 partial sums.  The AVX2 form below produces the
//           same bits as the scalar form and as the device code.
#pragma once
#include <immintrin.h>

#include <cstddef>

namespace vdb {

inline float l2_seq(const float* a, const float* b, size_t dim) {
    float s = 0.f;
    for (size_t i = 0; i < dim; ++i) {
        const float t = a[i] - b[i];
        s += t * t;
    }
    return s;
}

inline float ip_seq(const float* a, const float* b, size_t dim) {
    float s = 0.f;
    for (size_t i = 0; i < dim; ++i) s += a[i] * b[i];
    return 1.0f - s;
}

template <bool IS_L2>
inline float tree64_scalar(const float* a, const float* b, size_t dim) {
    float part[64];
    for (int v = 0; v < 64; ++v) {
        float s = 0.f;
        for (size_t e = v; e < dim; e += 64) {
            if (IS_L2) {
                const float t = a[e] - b[e];
                s = __builtin_fmaf(t, t, s);
            } else {
                s = __builtin_fmaf(a[e], b[e], s);
            }
        }
        part[v] = s;
    }
    for (int w = 32; w >= 1; w >>= 1)
        for (int i = 0; i < w; ++i) part[i] = part[2 * i] + part[2 * i + 1];
    return IS_L2 ? part[0] : 1.0f - part[0];
}

// AVX2+FMA form.  Register r (0..7) holds virtual lanes 8r..8r+7.
template <bool IS_L2>
__attribute__((target("avx2,fma"))) inline float tree64_avx2(const float* a, const float* b, size_t dim) {
    __m256 acc[8];
    for (int r = 0; r < 8; ++r) acc[r] = _mm256_setzero_ps();
    size_t e = 0;
    for (; e + 64 <= dim; e += 64) {
        for (int r = 0; r < 8; ++r) {
            const __m256 x = _mm256_loadu_ps(a + e + 8 * r);
            const __m256 y = _mm256_loadu_ps(b + e + 8 * r);
            if (IS_L2) {
                const __m256 t = _mm256_sub_ps(x, y);
                acc[r] = _mm256_fmadd_ps(t, t, acc[r]);
            } else {
                acc[r] = _mm256_fmadd_ps(x, y, acc[r]);
            }
        }
    }
    if (e < dim) {  // ragged tail: lanes past the end keep their partial sum untouched
        const size_t rem = dim - e;
        for (int r = 0; r < 8 && (size_t)(8 * r) < rem; ++r) {
            const int n = (int)((rem - 8 * r) < 8 ? (rem - 8 * r) : 8);
            alignas(32) int m[8];
            for (int i = 0; i < 8; ++i) m[i] = i < n ? -1 : 0;
            const __m256i mask = _mm256_load_si256((const __m256i*)m);
            const __m256 x = _mm256_maskload_ps(a + e + 8 * r, mask);
            const __m256 y = _mm256_maskload_ps(b + e + 8 * r, mask);
            __m256 upd;
            if (IS_L2) {
                const __m256 t = _mm256_sub_ps(x, y);
                upd = _mm256_fmadd_ps(t, t, acc[r]);
            } else {
                upd = _mm256_fmadd_ps(x, y, acc[r]);
            }
            acc[r] = _mm256_blendv_ps(acc[r], upd, _mm256_castsi256_ps(mask));
        }
    }
    // pairwise tree: bits 0,1,2 of the lane index inside each register, then bits 3,4,5
    const __m256 t0 = _mm256_hadd_ps(acc[0], acc[1]);
    const __m256 t1 = _mm256_hadd_ps(acc[2], acc[3]);
    const __m256 t2 = _mm256_hadd_ps(acc[4], acc[5]);
    const __m256 t3 = _mm256_hadd_ps(acc[6], acc[7]);
    const __m256 u0 = _mm256_hadd_ps(t0, t1);  // [r0 r1 r2 r3 | r0' r1' r2' r3'] quad sums
    const __m256 u1 = _mm256_hadd_ps(t2, t3);
    const __m128 v0 = _mm_add_ps(_mm256_castps256_ps128(u0), _mm256_extractf128_ps(u0, 1));  // R0..R3
    const __m128 v1 = _mm_add_ps(_mm256_castps256_ps128(u1), _mm256_extractf128_ps(u1, 1));  // R4..R7
    const __m128 w = _mm_hadd_ps(v0, v1);                                                    // R01 R23 R45 R67
    const __m128 x = _mm_hadd_ps(w, w);                                                      // R0123 R4567 ..
    const float total = _mm_cvtss_f32(x) + _mm_cvtss_f32(_mm_shuffle_ps(x, x, 1));
    return IS_L2 ? total : 1.0f - total;
}

inline bool cpu_has_avx2_fma() {
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return ok;
}

using dist_fn = float (*)(const float*, const float*, size_t);

inline dist_fn pick_distance(int metric /*0 L2, 1 IP*/, int sum_order /*0 seq, 1 tree64*/) {
    if (sum_order == 0) return metric == 0 ? &l2_seq : &ip_seq;
    if (cpu_has_avx2_fma()) return metric == 0 ? &tree64_avx2<true> : &tree64_avx2<false>;
    return metric == 0 ? &tree64_scalar<true> : &tree64_scalar<false>;
}

}  // namespace vdb
