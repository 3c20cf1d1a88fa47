// gather_ceiling.hip - what the memory system gives a kernel that does nothing but the traversal's row gathers.
// (development tool; hipcc --offload-arch=gfx950 -O3 -o gather_ceiling gather_ceiling.hip; run on the GPU box)
//
// One wave per block, as hnsw_search_kernel.  Each step a wave reads ROWS pseudo-random rows of 512 B (128 floats) the
// way score_nodes does - a 16-lane group per row, two 16-B loads per lane, all loads of a step issued before the first
// use - and adds them up; the next step's row numbers do not depend on the data (a counter hashed), so nothing but the
// registers and the residency bounds the bytes in flight.  Printed: GB/s by table size, rows per step and waves per CU.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// DEP: the next step's rows depend on this step's data (one dependent gather per step, like a hop of the walk)
template <int GPB, bool DEP>
__global__ __launch_bounds__(64) void gather_kernel(const float* __restrict__ rows, uint32_t n_rows, uint32_t steps, float* out) {
    const int lane = threadIdx.x, grp = lane >> 4, t = lane & 15;
    uint32_t ctr = blockIdx.x * 0x9E3779B1u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    uint32_t carry = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        f32x4 x[GPB][2];
#pragma unroll
        for (int b = 0; b < GPB; ++b) {
            const uint32_t id = (uint32_t)(((uint64_t)mix(ctr + (uint32_t)(4 * b + grp) + carry) * n_rows) >> 32);
            const float* r = rows + (size_t)id * 128 + 4 * t;
            x[b][0] = *reinterpret_cast<const f32x4*>(r);
            x[b][1] = *reinterpret_cast<const f32x4*>(r + 64);
        }
#pragma unroll
        for (int b = 0; b < GPB; ++b) acc += x[b][0] + x[b][1];
        ctr += 4 * GPB;
        if (DEP) carry = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, acc.x)) & 0xFFu;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;  // (keeps the loads alive)
}

template <int GPB, bool DEP>
double run(const float* d_rows, uint32_t n_rows, uint32_t blocks, uint32_t steps, float* d_out) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<GPB, DEP>), dim3(blocks), dim3(64), 0, 0, d_rows, n_rows, steps / 4, d_out);  // warm-up
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<GPB, DEP>), dim3(blocks), dim3(64), 0, 0, d_rows, n_rows, steps, d_out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return (double)blocks * steps * (4.0 * GPB) * 512.0 / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const uint32_t steps = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 2000;
    float* d_out;
    CHECK(hipMalloc(&d_out, 1 << 20));
    for (uint64_t n_rows : {(uint64_t)1000000, (uint64_t)20000000, (uint64_t)100000000}) {
        float* d_rows;
        if (hipMalloc(&d_rows, n_rows * 512) != hipSuccess) {
            std::printf("%llu rows: allocation failed\n", (unsigned long long)n_rows);
            continue;
        }
        CHECK(hipMemset(d_rows, 0, n_rows * 512));
        for (uint32_t per_cu : {8u, 10u, 16u}) {
            const uint32_t blocks = 256 * per_cu * 4;  // four rounds of the resident waves
            std::printf("%9llu rows (%5.1f GB), %2u waves per CU:  independent steps  16 rows %6.0f  32 rows %6.0f GB/s   |  one dependent gather per step  16 rows %6.0f  32 rows %6.0f GB/s\n",
                        (unsigned long long)n_rows, n_rows * 512 / 1e9, per_cu, run<4, false>(d_rows, (uint32_t)n_rows, blocks, steps, d_out),
                        run<8, false>(d_rows, (uint32_t)n_rows, blocks, steps, d_out), run<4, true>(d_rows, (uint32_t)n_rows, blocks, steps, d_out),
                        run<8, true>(d_rows, (uint32_t)n_rows, blocks, steps, d_out));
            std::fflush(stdout);
        }
        CHECK(hipFree(d_rows));
    }
    return 0;
}
