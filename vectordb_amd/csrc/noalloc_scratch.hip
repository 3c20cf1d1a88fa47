// noalloc_scratch.hip — the device memory behind vdb_search_batch_device_noalloc (include/vdb_index.h).
//
// That entry point is for callers that cannot call hipMalloc (code under HIP graph capture, tight
// serving loops, harnesses that forbid it).  The traversal needs two scratch buffers beyond its inputs
// and outputs (hnsw_search.hip): the work list of queries the fast configuration hands to the large
// configuration (header + one word per query), and one exact visited bitmap per resident wave of the
// large configuration.  Every other entry point takes them from hipMalloc (capi.cpp, SearchScratch);
// this one takes them from the array below, which every GPU the library runs on holds until the process
// ends (kNoAllocScratchBytes, about 4.1 MB).
//
// In host code the array's NAME is a host-side stand-in, not a device address.  Its address on a
// device comes from hipGetSymbolAddress, asked once per device with that device current and kept; host
// code uses only that pointer (kernel arguments, hipMemsetAsync).  Each GPU has its own copy of the
// array at its own address.
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "launch.hpp"

namespace vdb {

namespace {

static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_lock;
void* g_ptr[kMaxDevices] = {};

}  // namespace

// The address of `scratch` on `device`, which the caller has made current.  0, or a vdb_status with *err set.
int noalloc_scratch(int device, void** out, std::string* err) {
    if (device < 0 || device >= kMaxDevices) {
        if (err) *err = "no-allocation scratch: device ordinal out of range";
        return 1;
    }
    std::unique_lock<std::mutex> guard(g_lock);
    if (!g_ptr[device]) {
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e == hipSuccess && cur != device) {
            if (err) *err = "no-allocation scratch: the device is not current";
            return 1;
        }
        void* p = nullptr;
        if (e == hipSuccess) e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            if (err) *err = std::string("hipGetSymbolAddress(scratch): ") + (e != hipSuccess ? hipGetErrorString(e) : "null address");
            return 3;
        }
        g_ptr[device] = p;
    }
    *out = g_ptr[device];
    return 0;
}

}  // namespace vdb
