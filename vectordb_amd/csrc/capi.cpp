// capi.cpp — the C-ABI of include/vdb_index.h: host index objects, the HBM copy of an
// index, and the enqueueing of the gfx950 kernels.  No CPU search lives here: every search
// entry point runs the kernels or fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is bound at run time (rccl_api below)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vdb_index.h"
#include "device_index.hpp"
#include "host_index.hpp"
#include "launch.hpp"

using namespace vdb;

namespace {

thread_local std::string g_error;

int set_error(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

#define VDB_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) throw DeviceError(std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

struct DeviceError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct Unsupported : std::runtime_error {
    using std::runtime_error::runtime_error;
};

template <typename F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const std::invalid_argument& e) {
        return set_error(VDB_ERR_INVALID, e.what());
    } catch (const DeviceError& e) {
        return set_error(VDB_ERR_NO_DEVICE, e.what());
    } catch (const Unsupported& e) {
        return set_error(VDB_ERR_UNSUPPORTED, e.what());
    } catch (const std::bad_alloc&) {
        return set_error(VDB_ERR_RUNTIME, "Not enough memory");
    } catch (const std::exception& e) {
        return set_error(VDB_ERR_RUNTIME, e.what());
    }
}

struct DeviceGuard {  // makes `dev` current for the scope, then restores the caller's device
    int prev = -1;
    explicit DeviceGuard(int dev) {
        VDB_HIP(hipGetDevice(&prev));
        if (prev != dev) VDB_HIP(hipSetDevice(dev));
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void alloc(size_t n) {
        release();
        if (n == 0) n = 256;
        VDB_HIP(hipMalloc(&p, n));
        bytes = n;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    void reserve(size_t n) {  // grow only (the contents are not kept)
        if (bytes < n || !p) alloc(n);
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// Per-stream scratch of the traversal (work list of re-run queries + visited bitmaps).
struct SearchScratch {
    DevBuf redo, bitmaps, allow;  // allow: delete flags AND the caller's filter, when both exist
    DevBuf ghash;                 // visited tables in HBM for the large-ef configuration (allocated on first need)
    DevBuf spill;                 // second-level visited tables for the queries that fill their LDS table
    DevBuf heaps;                 // candidate heaps of hnsw_search_unbounded_kernel (filtered searches; allocated on first need)
    uint32_t heap_cap = 0, heap_blocks = 0;
    uint32_t large_blocks = kLargeBlocks;  // bitmaps held (scratch_for)
    uint32_t spill_tables = 0;
    // host-mapped {re-run queries, nq} written by the last launch on this stream, and the ef it ran at:
    // read (without synchronising) before the next launch to size the visited set (see search_device)
    uint32_t* report = nullptr;
    uint32_t report_ef = 0;
    bool report_filtered = false;
    uint32_t report_level = 0;  // the filt_level that launch ran with
    // recorded behind the launch that writes `report`; the words are read (and the next observation
    // armed) only once it has completed, so what a launch adapts to never depends on timing
    hipEvent_t report_done = nullptr;
    bool report_pending = false;
    ~SearchScratch() {
        if (report) (void)hipHostFree(report);
        if (report_done) (void)hipEventDestroy(report_done);
    }
    size_t redo_cap = 0;  // queries per call that `redo` holds: kRedoHeader + 2 * redo_cap words
    uint32_t bitmap_words = 0;
};

// Device staging of the host-pointer entry points, kept per index (grow only) instead of a
// hipMalloc/hipFree of four buffers per call; `lock` serialises those (synchronous) calls.
struct HostStage {
    std::mutex lock;
    DevBuf q, dist, label, stats, filter;
    void release() {
        q.release();
        dist.release();
        label.release();
        stats.release();
        filter.release();
    }
};

int spill_mode_from_env() {  // VDB_SPILL=0 / 1 (tests): never / always hand out spill tables; unset: adaptive
    const char* e = std::getenv("VDB_SPILL");
    return e ? (e[0] == '1' ? 1 : 0) : -1;
}

}  // namespace

struct vdb_index {
    std::unique_ptr<HostIndex> host;
    // device copy
    int device = -1;
    uint64_t device_version = ~0ull;
    DevGraph graph{};
    DevBuf vecs, links0, upper_row, upper_links, labels;
    DevBuf live;  // bit i = element i is not deleted; allocated only when the upload saw deleted elements
    std::map<uint32_t, uint32_t> hbits_bias;  // per ef: extra log2 of the visited-set size (guarded by scratch_lock)
    std::map<uint32_t, uint32_t> vis_mean;    // per ef: nodes a bare search visits on average, as the launches reported (same lock)
    std::set<uint32_t> spill_ef;              // the ef values at which a launch had queries outgrow the LDS table (same lock)
    std::map<uint32_t, uint32_t> filt_level;  // per ef, filtered searches: doublings of the first level's candidate buffer (same lock)
    int spill_mode = spill_mode_from_env();   // -1 adaptive, 0 never, 1 always (read once, at creation)
    std::mutex scratch_lock;
    std::map<hipStream_t, std::unique_ptr<SearchScratch>> scratch;
    HostStage stage;

    size_t device_bytes() const {
        return device < 0 ? 0 : vecs.bytes + links0.bytes + upper_row.bytes + upper_links.bytes + labels.bytes + live.bytes;
    }
    void drop_device() {
        if (device < 0) return;
        try {
            DeviceGuard g(device);
            (void)hipDeviceSynchronize();
            scratch.clear();
            stage.release();
            vecs.release();
            links0.release();
            upper_row.release();
            upper_links.release();
            labels.release();
            live.release();
        } catch (...) {
        }
        device = -1;
    }
    ~vdb_index() { drop_device(); }
};

struct vdb_bruteforce {
    std::unique_ptr<HostBruteforce> host;
    int device = -1;
    uint64_t device_version = ~0ull;
    DevBuf vecs, labels, xc, xb, nx, mu, stats;
    uint32_t n = 0, vstride = 0, kp64 = 0;
    std::mutex scratch_lock;
    std::map<hipStream_t, std::unique_ptr<DevBuf>> scratch;
    HostStage stage;

    void drop_device() {
        if (device < 0) return;
        try {
            DeviceGuard g(device);
            (void)hipDeviceSynchronize();
            scratch.clear();
            stage.release();
            vecs.release();
            labels.release();
            xc.release();
            xb.release();
            nx.release();
            mu.release();
            stats.release();
        } catch (...) {
        }
        device = -1;
    }
    ~vdb_bruteforce() { drop_device(); }
};

namespace {

void upload_bf(vdb_bruteforce* bf, int device) {
    HostBruteforce& h = *bf->host;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw DeviceError("no HIP device visible");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device ordinal out of range");
    bf->drop_device();
    DeviceGuard guard(device);
    std::unique_lock<std::mutex> g(h.lock);
    const size_t n = h.count;
    if (n >= 0xFFFFFFFFull) throw Unsupported("more than 2^32-1 rows per brute-force index (shard it)");
    const uint32_t vstride = (uint32_t)round_up(h.dim, 32);
    if (vstride > kMaxVecStride) throw Unsupported("dim above 1024 is not supported by the device path");
    bf->vecs.alloc(n * (size_t)vstride * 4);
    bf->labels.alloc(n * 8);
    const size_t chunk = 1 << 16;
    std::vector<float> sv(chunk * vstride, 0.f);
    std::vector<uint64_t> sl(chunk);
    for (size_t lo = 0; lo < n; lo += chunk) {
        const size_t m = std::min(chunk, n - lo);
        for (size_t i = 0; i < m; ++i) {
            const char* r = h.data + (lo + i) * h.stride;
            std::memcpy(sv.data() + i * vstride, r, h.data_size);
            std::memcpy(&sl[i], r + h.data_size, 8);
        }
        VDB_HIP(hipMemcpy((char*)bf->vecs.p + lo * vstride * 4, sv.data(), m * vstride * 4, hipMemcpyHostToDevice));
        VDB_HIP(hipMemcpy((char*)bf->labels.p + lo * 8, sl.data(), m * 8, hipMemcpyHostToDevice));
    }
    // centred f32 / bf16 copies, norms and the mean for the MFMA candidate pass (bf_mfma.hip)
    const uint32_t kp64 = (uint32_t)round_up(h.dim, 64);
    const size_t npad = round_up(n, 256);  // the GEMM operands are stored in 256-row panels (bf_mfma.hip)
    bf->xc.alloc(npad * (size_t)vstride * 4);
    bf->xb.alloc(npad * (size_t)kp64 * 2);
    bf->nx.alloc(npad * 4);
    if (npad) VDB_HIP(hipMemset(bf->nx.p, 0, npad * 4));
    bf->mu.alloc((size_t)kp64 * 4);
    bf->stats.alloc(16);
    {
        DevBuf tmp;
        tmp.alloc(bf_prepare_scratch_bytes(vstride));
        VDB_HIP(hipMemset(bf->mu.p, 0, (size_t)kp64 * 4));
        std::string err;
        if (launch_bf_prepare((const float*)bf->vecs.p, (uint32_t)n, (uint32_t)h.dim, vstride, kp64, (float*)bf->xc.p,
                              bf->xb.p, (float*)bf->nx.p, (float*)bf->mu.p, (float*)bf->stats.p, tmp.p, nullptr, &err) != 0)
            throw DeviceError(err);
        VDB_HIP(hipStreamSynchronize(nullptr));
    }
    bf->n = (uint32_t)n;
    bf->vstride = vstride;
    bf->kp64 = kp64;
    bf->device = device;
    bf->device_version = h.version;
}

// Host layout (reference byte layout) -> HBM layout of device_index.hpp, streamed in
// chunks so that a 100 M-row index never needs a second full host copy.
void upload(vdb_index* idx, int device) {
    HostIndex& h = *idx->host;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw DeviceError("no HIP device visible");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device ordinal out of range");
    idx->drop_device();
    DeviceGuard guard(device);

    const size_t n = h.count;
    if (n >= 0x7FFFFFFFull) throw Unsupported("more than 2^31-1 elements per index (shard it)");
    const uint32_t vstride = (uint32_t)round_up(h.dim, 32);
    const uint32_t l0stride = (uint32_t)round_up(h.maxM0, 32);
    const uint32_t ustride = (uint32_t)round_up(h.maxM, 16);
    if (vstride > kMaxVecStride) throw Unsupported("dim above 1024 is not supported by the device path");
    if (l0stride > kMaxL0Stride) throw Unsupported("M above 32 is not supported by the device path");

    // upper rows: one per (node, level >= 1)
    std::vector<uint32_t> up_row(std::max<size_t>(n, 1), kNoId);
    size_t rows = 0;
    for (size_t i = 0; i < n; ++i) {
        if (h.levels[i] > 0) {
            up_row[i] = (uint32_t)rows;
            rows += (size_t)h.levels[i];
        }
    }
    if (rows >= 0xFFFFFFFFull) throw Unsupported("too many upper-layer lists");

    idx->vecs.alloc(n * (size_t)vstride * 4);
    idx->links0.alloc(n * (size_t)l0stride * 4);
    idx->upper_row.alloc(n * 4);
    idx->upper_links.alloc(rows * (size_t)ustride * 4);
    idx->labels.alloc(n * 8);

    // What the kernels assume of a graph, checked here so that a corrupt or foreign file is an error
    // (the reference's "Index seems to be corrupted", hnsw_core.hpp:1129) and never an out-of-bounds read:
    // the entry point exists and has max_level levels; every link of layer lv names an existing element
    // that HAS a layer lv; no list names an element twice.
    if (n != 0 && (h.entry >= n || h.max_level < 0 || h.levels[h.entry] != h.max_level))
        throw std::runtime_error("Index seems to be corrupted: bad entry point or maximum level");
    auto check_list = [&](const uint32_t* ids, unsigned cnt, size_t cap, int lv) {
        if (cnt > cap) throw std::runtime_error("Index seems to be corrupted: list count above its capacity");
        for (unsigned a = 0; a < cnt; ++a) {
            if (ids[a] >= n) throw std::runtime_error("Index seems to be corrupted: link to a non-existent element");
            if (h.levels[ids[a]] < lv) throw std::runtime_error("Index seems to be corrupted: link on a level the target does not have");
            for (unsigned b = 0; b < a; ++b)
                if (ids[a] == ids[b]) throw std::runtime_error("Index seems to be corrupted: duplicate link in a list");
        }
    };

    // Rows come from the host copy, or - an index opened by vdb_index_load_to_device - straight from the file, one
    // chunk at a time (hnsw_core.hpp:581-682 reads the whole block into memory first; a 65 GB index need not).
    const size_t chunk = 1 << 16;
    std::ifstream rows_in;
    std::vector<char> rows_buf;
    if (!h.rows_resident) {
        rows_in.open(h.rows_path, std::ios::binary);
        if (!rows_in.is_open()) throw std::runtime_error("Cannot open file");
        rows_in.seekg((std::streamoff)h.rows_offset, rows_in.beg);
        rows_buf.resize(chunk * h.stride);
    }
    std::vector<uint32_t> live((n + 31) / 32, 0u);
    size_t n_deleted = 0;
    std::vector<float> sv(chunk * vstride);
    std::vector<uint32_t> sl(chunk * l0stride);
    std::vector<uint64_t> slab(chunk);
    std::vector<uint32_t> su;
    const int n_threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    for (size_t lo = 0; lo < n; lo += chunk) {
        const size_t m = std::min(chunk, n - lo);
        const size_t urow_lo = [&] {
            for (size_t i = lo; i < lo + m; ++i)
                if (up_row[i] != kNoId) return (size_t)up_row[i];
            return rows;
        }();
        size_t urow_hi = urow_lo;
        for (size_t i = lo; i < lo + m; ++i) urow_hi += (size_t)std::max(h.levels[i], 0);
        su.assign((urow_hi - urow_lo) * ustride, kNoId);
        const char* rows = h.rows_resident ? h.level0 + lo * h.stride : rows_buf.data();
        if (!h.rows_resident) {
            rows_in.read(rows_buf.data(), (std::streamsize)(m * h.stride));
            if (!rows_in.good()) throw std::runtime_error("Index seems to be corrupted or unsupported");
        }
        for (size_t i = 0; i < m; ++i) {  // delete flag: byte +2 of the count word, hnsw_core.hpp:784-787
            if (((const unsigned char*)(rows + i * h.stride + h.link0_off))[2] & 0x01) n_deleted += 1;
            else live[(lo + i) >> 5] |= 1u << ((lo + i) & 31);
        }

        std::atomic<bool> bad{false};
        std::string bad_msg;
        std::mutex bad_lock;
        auto work = [&](size_t a, size_t b) {
            try {
                for (size_t i = a; i < b; ++i) {
                    const uint32_t id = (uint32_t)(lo + i);
                    float* dv = sv.data() + i * vstride;
                    const char* r = rows + i * h.stride;
                    std::memcpy(dv, r + h.data_off, h.dim * 4);
                    std::fill(dv + h.dim, dv + vstride, 0.f);
                    const uint32_t* l0 = (const uint32_t*)(r + h.link0_off);
                    const unsigned c0 = HostIndex::list_count(l0);
                    check_list(l0 + 1, c0, h.maxM0, 0);
                    uint32_t* dl = sl.data() + i * l0stride;
                    std::copy(l0 + 1, l0 + 1 + c0, dl);
                    std::fill(dl + c0, dl + l0stride, kNoId);
                    std::memcpy(&slab[i], r + h.label_off, 8);
                    for (int lv = 1; lv <= h.levels[id]; ++lv) {
                        const uint32_t* ul = h.list_at(id, lv);
                        const unsigned cu = HostIndex::list_count(ul);
                        check_list(ul + 1, cu, h.maxM, lv);
                        uint32_t* du = su.data() + ((size_t)up_row[id] + (size_t)(lv - 1) - urow_lo) * ustride;
                        std::copy(ul + 1, ul + 1 + cu, du);
                    }
                }
            } catch (const std::exception& e) {
                std::unique_lock<std::mutex> g(bad_lock);
                bad_msg = e.what();
                bad = true;
            }
        };
        std::vector<std::thread> pool;
        const size_t per = (m + n_threads - 1) / n_threads;
        for (int t = 0; t < n_threads; ++t) {
            const size_t a = std::min(m, t * per), b = std::min(m, a + per);
            if (a < b) pool.emplace_back(work, a, b);
        }
        for (auto& th : pool) th.join();
        if (bad) throw std::runtime_error(bad_msg);

        VDB_HIP(hipMemcpy((char*)idx->vecs.p + lo * vstride * 4, sv.data(), m * vstride * 4, hipMemcpyHostToDevice));
        VDB_HIP(hipMemcpy((char*)idx->links0.p + lo * l0stride * 4, sl.data(), m * l0stride * 4, hipMemcpyHostToDevice));
        VDB_HIP(hipMemcpy((char*)idx->labels.p + lo * 8, slab.data(), m * 8, hipMemcpyHostToDevice));
        if (!su.empty())
            VDB_HIP(hipMemcpy((char*)idx->upper_links.p + urow_lo * ustride * 4, su.data(), su.size() * 4,
                              hipMemcpyHostToDevice));
    }
    if (n) VDB_HIP(hipMemcpy(idx->upper_row.p, up_row.data(), n * 4, hipMemcpyHostToDevice));
    if (!h.rows_resident) idx->host->deleted = n_deleted;
    if (n_deleted != 0) {  // the delete flags as a bitmap of LIVE elements
        idx->live.alloc(live.size() * 4);
        VDB_HIP(hipMemcpy(idx->live.p, live.data(), live.size() * 4, hipMemcpyHostToDevice));
    }

    DevGraph& g = idx->graph;
    g.vecs = (const float*)idx->vecs.p;
    g.links0 = (const uint32_t*)idx->links0.p;
    g.upper_row = (const uint32_t*)idx->upper_row.p;
    g.upper_links = (const uint32_t*)idx->upper_links.p;
    g.labels = (const uint64_t*)idx->labels.p;
    g.n = (uint32_t)n;
    g.dim = (uint32_t)h.dim;
    g.vstride = vstride;
    g.l0stride = l0stride;
    g.ustride = ustride;
    g.entry = h.entry;
    g.max_level = h.max_level;
    g.metric = h.metric;
    {  // set-up of the no-allocation entry point: this device's address of the static scratch array
        void* p = nullptr;
        std::string err;
        if (noalloc_scratch(device, &p, &err) != 0) throw DeviceError(err);
    }
    idx->device = device;
    idx->device_version = h.version;
}

SearchScratch& scratch_for(vdb_index* idx, hipStream_t stream, size_t nq) {
    auto& slot = idx->scratch[stream];
    if (!slot) slot.reset(new SearchScratch);
    SearchScratch& s = *slot;
    // [header | re-run list: nq words | fail list: nq words] - the fail list (search_device) starts at kRedoHeader + nq
    // of the CURRENT call, so a call needs kRedoHeader + 2 nq words whatever an earlier, smaller call allocated
    if (s.redo_cap < nq) {
        // a previous call on this stream may still be using the old buffer
        if (s.redo.p) VDB_HIP(hipStreamSynchronize(stream));
        const size_t cap = std::max<size_t>(nq + nq / 2, 64);  // grow by more than asked: fewer reallocations
        s.redo.alloc((kRedoHeader + 2 * cap) * 4);
        s.redo_cap = cap;
    }
    const uint32_t tables = (uint32_t)std::min<size_t>(std::max<size_t>((nq + 31) / 32 * 32, kSpillTablesMin), kSpillTablesMax);
    if (s.spill_tables < tables) {
        if (s.spill.p) VDB_HIP(hipStreamSynchronize(stream));
        s.spill.alloc(((size_t)tables << kSpillBits) * 4);
        s.spill_tables = tables;
    }
    const uint32_t words = (idx->graph.n + 31u) / 32u;
    if (!s.bitmaps.p || s.bitmap_words < words) {
        if (s.bitmaps.p) VDB_HIP(hipStreamSynchronize(stream));
        // The large configuration re-runs what the first level gives up on, one resident wave per bitmap.  With 128 of
        // them a batch that overflows wholesale (a filter that keeps half the elements at ef 64 fills the 128 keys of
        // the first level: all 10 000 queries) ran 52 ms on a 1 M-element graph; as many as fit 256 MB, up to the 2 048
        // that can be resident, take it to a few ms.  Above 16 M elements it is 128 again.
        const size_t one = (size_t)std::max(words, 1u) * 4;
        s.large_blocks = (uint32_t)std::min<size_t>(kLargeBlocksMax, std::max<size_t>(kLargeBlocks, kLargeBitmapBytes / one));
        s.bitmaps.alloc((size_t)s.large_blocks * one);
        s.bitmap_words = std::max(words, 1u);
    }
    return s;
}

void require_search_ready(const vdb_index* idx) {
    if (!idx) throw std::invalid_argument("null index");
    if (idx->device < 0) throw DeviceError("index is not on a device: call vdb_index_to_device first");
    if (idx->device_version != idx->host->version)
        throw DeviceError("index changed since vdb_index_to_device: upload it again");
}

// The epsilon stop condition of searchStopConditionClosest (hnsw_core.hpp:1162-1213; the one condition that crosses the ABI).
struct EpsilonStop {
    float epsilon;
    size_t min_num_candidates, max_num_candidates;
};

// searchKnn over a batch; d_filter = nullable device bitmap over internal ids (1 = passes the filter).
// stop != nullptr: searchStopConditionClosest instead - the non-bare layer-0 search with ef = k = max_num_candidates.
// filter_accepts: how many elements the caller's filter accepts, when the caller knows (host bitmaps); SIZE_MAX = unknown
int search_device(vdb_index* idx, const float* d_queries, size_t nq, size_t k, const uint32_t* d_filter, float* d_out_dist,
                  uint64_t* d_out_label, uint32_t* d_out_stats, hipStream_t stream, uint32_t** fail_word,
                  const EpsilonStop* stop = nullptr, size_t filter_accepts = SIZE_MAX) {
    require_search_ready(idx);
    if (stop) {
        if (!(stop->epsilon == stop->epsilon)) throw std::invalid_argument("epsilon is NaN");
        if (stop->max_num_candidates == 0 || stop->min_num_candidates > stop->max_num_candidates)
            throw std::invalid_argument("stop condition: need 0 < max_num_candidates and min_num_candidates <= max_num_candidates");
        if (k != stop->max_num_candidates) throw std::invalid_argument("stop condition: the output rows hold max_num_candidates entries");
    }
    if (nq == 0) return (int)VDB_OK;
    if (!d_queries || !d_out_dist || !d_out_label) throw std::invalid_argument("null buffer");
    if (k == 0) throw std::invalid_argument("k must be > 0");
    if (nq > 0x7FFFFFFFull || k > 0x7FFFFFFFull) throw std::invalid_argument("batch too large");
    DeviceGuard guard(idx->device);
    const size_t ef = stop ? k : std::max(idx->host->ef_search, k);  // hnsw_core.hpp:1144; :1201 passes no ef, the condition bounds the set
    if (ef > kMaxEf) throw Unsupported("max(efSearch, k) above 508 is not supported by the device path");
    std::unique_lock<std::mutex> lock(idx->scratch_lock);
    SearchScratch& s = scratch_for(idx, stream, nq);
    // bare = no deleted element and no filter (hnsw_core.hpp:1141); else one bitmap of allowed nodes
    const uint32_t* allow = d_filter;
    const uint32_t words = (idx->graph.n + 31u) / 32u;
    if (idx->live.p) {
        allow = (const uint32_t*)idx->live.p;
        if (d_filter) {
            if (s.allow.bytes < (size_t)words * 4) {
                if (s.allow.p) VDB_HIP(hipStreamSynchronize(stream));
                s.allow.alloc((size_t)std::max(words, 1u) * 4);
            }
            if (launch_and_bitmaps((const uint32_t*)idx->live.p, d_filter, (uint32_t*)s.allow.p, words, stream) != 0)
                throw DeviceError("bitmap kernel launch failed");
            allow = (const uint32_t*)s.allow.p;
        }
    }
    if (stop && !allow) {  // a stop condition always runs the non-bare search (:1201): every node allowed
        if (s.allow.bytes < (size_t)std::max(words, 1u) * 4) {
            if (s.allow.p) VDB_HIP(hipStreamSynchronize(stream));
            s.allow.alloc((size_t)std::max(words, 1u) * 4);
        }
        VDB_HIP(hipMemsetAsync(s.allow.p, 0xFF, (size_t)std::max(words, 1u) * 4, stream));
        allow = (const uint32_t*)s.allow.p;
    }
    // The LDS visited set is sized for a typical search (hnsw_search.hip); what "typical" is depends on
    // the data.  Each launch reports how many queries outgrew it, and next_hbits_bias (launch.hpp)
    // decides whether the next launches at this ef get a table twice the size (sticky, at most 2^14 entries).
    if (!s.report) {
        VDB_HIP(hipHostMalloc((void**)&s.report, 16, hipHostMallocMapped));
        s.report[0] = s.report[1] = s.report[2] = s.report[3] = 0;
        VDB_HIP(hipEventCreateWithFlags(&s.report_done, hipEventDisableTiming));
    }
    // One observation per launch, taken only when the launch that wrote the words has completed
    // (its event); while it is still running the words are left alone and this launch does not
    // ask for a report, so nothing is read or zeroed under a kernel that may be writing it.
    bool arm_report = true;
    if (s.report_pending) {
        if (hipEventQuery(s.report_done) == hipSuccess) {
            // (a query whose key array overflowed says nothing about the visited table)
            const uint32_t keys = s.report[2], redone = s.report[0] - std::min(s.report[0], keys), of = s.report[1];
            if (std::getenv("VDB_TRACE_SEARCH"))
                std::fprintf(stderr, "[vdb search] report of the launch at ef %u: %u of %u queries outgrew the visited table, %u the key array / candidate buffer\n",
                             s.report_ef, redone, of, keys);
            if (of != 0 && s.report_ef != 0) {
                if (redone != 0) idx->spill_ef.insert(s.report_ef);
                uint32_t& b = idx->hbits_bias[s.report_ef];
                b = next_hbits_bias(b, redone, of);
                if (!s.report_filtered && s.report[3] != 0)  // (latest launch: the figure moves with the data the index holds)
                    idx->vis_mean[s.report_ef] = (uint32_t)std::min<uint64_t>((uint64_t)s.report[3] * 16u / of, 0x7FFFFFFFu);
                // filtered searches: the first level keeps the candidates that are not allowed in a buffer in LDS (about
                // ef (1 - s) / s live keys at selectivity s, plus the ones the shrinking bound has left behind); when more
                // than 0.2 % of a launch outgrew it the next filtered launches at this ef get twice the buffer (sticky;
                // 256 -> ... -> 4096 keys beside the 64 in registers)
                // (0.2 %, not the 2 % of the visited table: an overflow here is a re-run by the large configuration's few waves)
                // (ANY overflow up to 2 048 keys: however few they are, the queries that overflow are re-run behind the
                // launch by the large configuration and take a whole search's latency there)
                if (s.report_filtered && keys != 0 && (s.report_level + 2u < kFiltLevels || (uint64_t)keys * 500u > of)) {
                    uint32_t& lv = idx->filt_level[s.report_ef];
                    lv = std::min(std::max(lv, s.report_level) + 1u, kFiltLevels - 1u);
                }
            }
            s.report[0] = s.report[1] = s.report[2] = s.report[3] = 0;
            s.report_pending = false;
        } else {
            (void)hipGetLastError();  // hipErrorNotReady is not an error here
            arm_report = false;
        }
    }
    // A first bare launch at this ef: start from what the launches at the nearest ef values that have reported taught, not
    // from the smallest table - on a large graph the first launches at a new ef otherwise send most of a batch through the
    // spill tables or the large configuration.  The nodes a search visits grow like ef^x with x = 0.36 on C2 (1 M rows:
    // 560 / 890 / 1 090 at ef 32 / 100 / 200) and 0.72 on 20 M rows (2 020 / 3 330 / 4 400 at ef 100 / 200 / 300): x comes
    // from the two nearest ef that are known, or is taken low (0.35) when only one is - the table size only ever grows
    // with the reports, so the starting point must not be beyond where they would settle.
    if (!allow && idx->hbits_bias.find((uint32_t)ef) == idx->hbits_bias.end() && !idx->vis_mean.empty()) {
        auto hi = idx->vis_mean.lower_bound((uint32_t)ef);
        auto near = hi == idx->vis_mean.end() ? std::prev(hi) : hi;
        if (hi != idx->vis_mean.begin() && hi != idx->vis_mean.end() && ef - std::prev(hi)->first < hi->first - ef) near = std::prev(hi);
        double x = 0.35;
        if (idx->vis_mean.size() >= 2) {
            auto other = near == idx->vis_mean.begin() ? std::next(near) : std::prev(near);
            if (near->second > 0 && other->second > 0)
                x = std::log((double)near->second / (double)other->second) / std::log((double)near->first / (double)other->first);
            x = std::min(0.8, std::max(0.3, x));
        }
        const double guess = (double)near->second * std::pow((double)ef / (double)near->first, x);
        // (the tables' fill limits: the 10 KB share holds about 1 700 nodes, 16 KB about 2 800 - launch_cfg)
        const uint32_t bias = guess > 2800.0 ? 2u : guess > 1700.0 ? 1u : 0u;
        idx->hbits_bias[(uint32_t)ef] = bias;
        idx->vis_mean[(uint32_t)ef] = (uint32_t)guess;
        if (bias) idx->spill_ef.insert((uint32_t)ef);
    }
    HnswSearchWork work;
    work.redo = (uint32_t*)s.redo.p;
    // Spill tables (hnsw_search.hip, LdsVisited) go to the launches at an ef at which an earlier launch
    // had queries outgrow the LDS table: the kernel variant that can spill is 1-5 % slower for the
    // searches that never do.  vdb_index_set_spill_mode / VDB_SPILL at creation (tests) = never / always.
    {
        // (from 4 M elements on and ef >= 64 also on the first launch at an ef: searches of such graphs visit more than
        // the LDS table holds as a rule, and the large configuration has 128 waves for them there - 10 000 queries on a
        // 100 M-element graph would wait 80 ms for their first answer)
        const bool on = idx->spill_mode >= 0 ? idx->spill_mode == 1
                                             : idx->spill_ef.count((uint32_t)ef) != 0 || (idx->graph.n >= (1u << 22) && ef >= 64);
        work.spill = on ? (uint32_t*)s.spill.p : nullptr;
        work.spill_tables = s.spill_tables;
    }
    work.bitmaps = (uint32_t*)s.bitmaps.p;
    work.bitmap_words = s.bitmap_words;
    work.large_blocks = s.large_blocks;
    {
        auto it = idx->hbits_bias.find((uint32_t)ef);
        work.hbits_bias = it == idx->hbits_bias.end() ? 0u : it->second;
        auto iv = idx->vis_mean.find((uint32_t)ef);
        work.vis_mean = (iv == idx->vis_mean.end() || allow) ? 0u : iv->second;
    }
    // visited sets in HBM: bare searches at large ef, filtered ones once launches at this ef reported the LDS table outgrown
    if ((ef > 124 && !allow) || (allow && work.hbits_bias >= 1)) {
        if (!s.ghash.p) s.ghash.alloc(((size_t)kGlobalHashBlocks << kGlobalHashBits) * 4);
        work.ghash = (uint32_t*)s.ghash.p;
    }
    if (allow && idx->graph.n) {
        // filtered searches: the queries that outgrow every key array are answered from a heap in HBM.  One heap holds
        // every node of the graph at worst (a filter that accepts nothing walks the whole graph); at most 256 MB in all.
        const uint32_t cap = (uint32_t)std::min<size_t>(idx->graph.n, (size_t)1 << 25);
        if (s.heap_cap < cap) {
            if (s.heaps.p) VDB_HIP(hipStreamSynchronize(stream));
            const uint32_t blocks = (uint32_t)std::max<size_t>(1, std::min<size_t>(64, ((size_t)256 << 20) / ((size_t)cap * 8)));
            s.heaps.alloc((size_t)blocks * cap * 8);
            s.heap_cap = cap;
            s.heap_blocks = blocks;
        }
        work.heaps = (uint64_t*)s.heaps.p;
        work.heap_cap = s.heap_cap;
        work.heap_blocks = s.heap_blocks;
        work.fail_list = work.redo + kRedoHeader + nq;  // the second half of the work list's buffer (scratch_for)
    }
    work.report = arm_report ? s.report : nullptr;
    if (allow) {
        auto it = idx->filt_level.find((uint32_t)ef);
        work.filt_level = it == idx->filt_level.end() ? 0u : it->second;
        // What the host knows ahead of any report: when a fraction r of the elements is rejected (deleted, or refused by a
        // filter whose verdicts the caller counted) the buffer holds about ef r / (1 - r) live keys; half as much again
        // leaves room for the ones the shrinking bound has left behind (a sweep drops them when the buffer is full).
        const size_t n = idx->graph.n, dead = std::min<size_t>(idx->host->deleted, n);
        size_t rejected = dead;
        if (filter_accepts != SIZE_MAX && d_filter) {
            // (deleted elements the filter accepts are rejected too: assume the two are independent)
            const size_t acc = std::min(filter_accepts, n);
            rejected = n - (size_t)((double)acc * (double)(n - dead) / (double)std::max<size_t>(n, 1));
        }
        if (n != 0 && rejected != 0 && rejected < n) {
            const double need = 1.5 * (double)ef * (double)rejected / (double)(n - rejected);
            uint32_t cap = 256u + 64u, lv = 0;
            while (lv + 1 < kFiltLevels && (double)cap < need) cap = (cap - 64u) * 2u + 64u, lv += 1;
            work.filt_level = std::max(work.filt_level, lv);
        }
    }
    if (stop) {
        uint32_t u;
        std::memcpy(&u, &stop->epsilon, 4);
        if (u == 0x80000000u) u = 0;  // -0.0 == +0.0
        work.eps_ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        if (work.eps_ord == 0xFFFFFFFFu) work.eps_ord = 0xFFFFFFFEu;  // (unreachable for a non-NaN epsilon: +inf is 0xFF800000)
        work.eps_min = (uint32_t)stop->min_num_candidates;
    }
    if (fail_word) *fail_word = work.heaps ? work.redo + 3 : work.redo + 1;  // queries left unanswered
    std::string err;
    const int rc = launch_hnsw_search(idx->graph, d_queries, (uint32_t)nq, (uint32_t)k, (uint32_t)ef, allow, d_out_dist,
                                      d_out_label, d_out_stats, work, stream, &err);
    if (rc != 0) return set_error(rc, err);
    static const bool trace = std::getenv("VDB_TRACE_SEARCH") != nullptr;
    if (trace)
        std::fprintf(stderr, "[vdb search] nq %zu ef %zu k %zu %s: visited-set bias %u (a search visits %u nodes), spill tables %s, candidate-buffer level %u, report %s\n",
                     nq, ef, k, allow ? "filtered" : "bare", work.hbits_bias, work.vis_mean, work.spill ? "on" : "off", work.filt_level,
                     arm_report ? "armed" : "not armed");
    if (arm_report) {
        s.report_ef = (uint32_t)ef;
        s.report_filtered = allow != nullptr;
        s.report_level = work.filt_level;
        VDB_HIP(hipEventRecord(s.report_done, stream));
        s.report_pending = true;
    }
    return (int)VDB_OK;
}

int search_host(vdb_index* idx, const float* queries, size_t nq, size_t k, const uint32_t* filter, float* out_dist,
                uint64_t* out_label, uint32_t* out_stats, const EpsilonStop* stop = nullptr) {
    require_search_ready(idx);
    if (nq == 0) return (int)VDB_OK;
    if (!queries || !out_dist || !out_label) throw std::invalid_argument("null buffer");
    if (k == 0) throw std::invalid_argument("k must be > 0");
    DeviceGuard guard(idx->device);
    const size_t dim = idx->host->dim;
    HostStage& st = idx->stage;
    std::unique_lock<std::mutex> stage_guard(st.lock);
    st.q.reserve(nq * dim * 4);
    st.dist.reserve(nq * k * 4);
    st.label.reserve(nq * k * 8);
    if (out_stats) st.stats.reserve(nq * 16);
    VDB_HIP(hipMemcpy(st.q.p, queries, nq * dim * 4, hipMemcpyHostToDevice));
    if (filter) {
        const size_t words = ((size_t)idx->graph.n + 31) / 32;
        st.filter.reserve(std::max<size_t>(words, 1) * 4);
        VDB_HIP(hipMemcpy(st.filter.p, filter, words * 4, hipMemcpyHostToDevice));
    }
    uint32_t* fail_word = nullptr;
    size_t accepts = SIZE_MAX;
    if (filter) {  // (the verdicts are here anyway: count them, so that the first launch sizes its candidate buffer right)
        const size_t n = idx->graph.n, words = (n + 31) / 32;
        accepts = 0;
        for (size_t w = 0; w + 1 < words; ++w) accepts += (size_t)__builtin_popcount(filter[w]);
        if (words) accepts += (size_t)__builtin_popcount(filter[words - 1] & (n % 32 ? (1u << (n % 32)) - 1u : 0xFFFFFFFFu));
    }
    const int rc = search_device(idx, (const float*)st.q.p, nq, k, filter ? (const uint32_t*)st.filter.p : nullptr,
                                 (float*)st.dist.p, (uint64_t*)st.label.p, out_stats ? (uint32_t*)st.stats.p : nullptr, nullptr,
                                 &fail_word, stop, accepts);
    if (rc != VDB_OK) return rc;
    VDB_HIP(hipStreamSynchronize(nullptr));
    uint32_t failed = 0;
    if (fail_word) VDB_HIP(hipMemcpy(&failed, fail_word, 4, hipMemcpyDeviceToHost));
    VDB_HIP(hipMemcpy(out_dist, st.dist.p, nq * k * 4, hipMemcpyDeviceToHost));
    VDB_HIP(hipMemcpy(out_label, st.label.p, nq * k * 8, hipMemcpyDeviceToHost));
    if (out_stats) VDB_HIP(hipMemcpy(out_stats, st.stats.p, nq * 16, hipMemcpyDeviceToHost));
    if (failed)
        throw Unsupported(std::to_string(failed) +
                          " queries hold more live candidates outside the result set than the device search keeps "
                          "(a filter that rejects almost every element, or hundreds of base vectors at exactly the "
                          "same distance): their rows are NaN / UINT64_MAX");
    return (int)VDB_OK;
}

}  // namespace

extern "C" {

const char* vdb_last_error(void) { return g_error.c_str(); }

int vdb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vdb_index_create(vdb_index** out, int metric, size_t dim, size_t capacity, size_t M, size_t efc, size_t seed) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("null output pointer");
        if (metric != VDB_L2 && metric != VDB_IP) throw std::invalid_argument("unknown metric");
        std::unique_ptr<vdb_index> idx(new vdb_index);
        idx->host.reset(new HostIndex(metric, dim, capacity, M, efc, seed));
        *out = idx.release();
        return (int)VDB_OK;
    });
}

int vdb_index_load(vdb_index** out, const char* path, int metric, size_t dim, size_t capacity) {
    return guarded([&] {
        if (!out || !path) throw std::invalid_argument("null argument");
        if (metric != VDB_L2 && metric != VDB_IP) throw std::invalid_argument("unknown metric");
        std::unique_ptr<vdb_index> idx(new vdb_index);
        idx->host.reset(new HostIndex(std::string(path), metric, dim, capacity));
        *out = idx.release();
        return (int)VDB_OK;
    });
}

int vdb_index_save(const vdb_index* idx, const char* path) {
    return guarded([&] {
        if (!idx || !path) throw std::invalid_argument("null argument");
        idx->host->save(path);
        return (int)VDB_OK;
    });
}

void vdb_index_free(vdb_index* idx) { delete idx; }

int vdb_index_add(vdb_index* idx, const float* vec, uint64_t label, int level) {
    return guarded([&] {
        if (!idx || !vec) throw std::invalid_argument("null argument");
        idx->host->add(vec, label, level);
        return (int)VDB_OK;
    });
}

int vdb_index_add_point(vdb_index* idx, const float* vec, uint64_t label, int replace_deleted) {
    return guarded([&] {
        if (!idx || !vec) throw std::invalid_argument("null argument");
        idx->host->add_point(vec, label, replace_deleted != 0);
        return (int)VDB_OK;
    });
}

int vdb_index_set_reuse_deleted(vdb_index* idx, int on) {
    return guarded([&] {
        if (!idx) throw std::invalid_argument("null index");
        idx->host->set_reuse_deleted(on != 0);
        return (int)VDB_OK;
    });
}

int vdb_index_add_batch(vdb_index* idx, const float* vecs, const uint64_t* labels, size_t n, int n_threads) {
    return guarded([&] {
        if (!idx || (!vecs && n)) throw std::invalid_argument("null argument");
        idx->host->add_batch(vecs, labels, n, n_threads);
        return (int)VDB_OK;
    });
}

}  // extern "C"

namespace {

struct BuildTrace {  // VDB_TRACE=1: stage timings on stderr (a 100M-element build runs for minutes)
    bool on = std::getenv("VDB_TRACE") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void lap(const char* what, size_t n) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[vdb build] %-44s %9zu elements  %.1fs\n", what, n, std::chrono::duration<double>(now - last).count());
        last = now;
    }
};

// Upload, then insert `dev_ids` (stored, unlinked, level 0) into layer 0 in rounds of
// search -> select -> sort -> reverse links (hnsw_build.hip), then write the lists back to the host.
// `linked` = elements layer 0 already holds.
void device_rounds(vdb_index* idx, const std::vector<uint32_t>& dev_ids, size_t linked, int device, BuildTrace& tr) {
    HostIndex& h = *idx->host;
    upload(idx, device);  // vectors of all elements, the links made so far, the finished upper layers
    tr.lap("upload", h.count);
    DeviceGuard guard(device);
    const DevGraph& g0 = idx->graph;
    const uint32_t efc = (uint32_t)h.ef_construction, M = (uint32_t)h.M, maxM0 = (uint32_t)h.maxM0;
    const size_t total = h.count;
    const uint32_t max_batch = 32768;
    DevBuf ident, ids, q, cd, cl, k_in, k_out, v_in, v_out, temp;
    {
        std::vector<uint64_t> id64(total);
        for (size_t i = 0; i < total; ++i) id64[i] = i;
        ident.alloc(total * 8);
        VDB_HIP(hipMemcpy(ident.p, id64.data(), total * 8, hipMemcpyHostToDevice));
    }
    ids.alloc(dev_ids.size() * 4);
    VDB_HIP(hipMemcpy(ids.p, dev_ids.data(), dev_ids.size() * 4, hipMemcpyHostToDevice));
    q.alloc((size_t)max_batch * h.dim * 4);
    cd.alloc((size_t)max_batch * efc * 4);
    cl.alloc((size_t)max_batch * efc * 8);
    BuildWork w;
    const size_t pair_bytes = (size_t)max_batch * M * 4;
    k_in.alloc(pair_bytes);
    k_out.alloc(pair_bytes);
    v_in.alloc(pair_bytes);
    v_out.alloc(pair_bytes);
    w.key_in = (uint32_t*)k_in.p;
    w.key_out = (uint32_t*)k_out.p;
    w.val_in = (uint32_t*)v_in.p;
    w.val_out = (uint32_t*)v_out.p;
    w.sort_temp_bytes = build_sort_temp_bytes(max_batch * M);
    temp.alloc(w.sort_temp_bytes);
    w.sort_temp = temp.p;

    DevGraph gb = g0;  // the search returns internal ids
    gb.labels = (const uint64_t*)ident.p;
    uint32_t* links0 = (uint32_t*)idx->links0.p;
    hipStream_t stream = nullptr;
    SearchScratch* sc;
    {
        std::unique_lock<std::mutex> lock(idx->scratch_lock);
        sc = &scratch_for(idx, stream, max_batch);
    }
    HnswSearchWork work;
    work.redo = (uint32_t*)sc->redo.p;
    work.bitmaps = (uint32_t*)sc->bitmaps.p;
    work.bitmap_words = sc->bitmap_words;
    work.large_blocks = sc->large_blocks;
    // the visited-set size follows what the rounds report, as in search_device: the searches of a
    // large graph at ef_construction visit more nodes than the default table holds, and the large
    // configuration (an n-bit map per re-run query) is no place for a whole batch
    if (!sc->report) {
        VDB_HIP(hipHostMalloc((void**)&sc->report, 16, hipHostMallocMapped));
        sc->report[0] = sc->report[1] = sc->report[2] = sc->report[3] = 0;
        VDB_HIP(hipEventCreateWithFlags(&sc->report_done, hipEventDisableTiming));
    }
    if (sc->report_pending) {  // a query launch's observation still in flight: let it land, then drop it
        VDB_HIP(hipEventSynchronize(sc->report_done));
        sc->report[0] = sc->report[1] = sc->report[3] = 0;
        sc->report_pending = false;
    }
    sc->report_ef = 0;  // (not an observation of a query launch)
    if (efc > 124) {
        if (!sc->ghash.p) sc->ghash.alloc(((size_t)kGlobalHashBlocks << kGlobalHashBits) * 4);
        work.ghash = (uint32_t*)sc->ghash.p;
    }
    std::string err;
    size_t done = 0, round = 0;
    size_t batch_div = M < 8 ? 16 : 4;
#ifdef VDB_DIAG  // (make DIAG=1: tools/exp_build_small_m.py; the shipped library reads no such switch)
    if (const char* e = std::getenv("VDB_BUILD_BATCH_DIV")) batch_div = (size_t)std::max(1l, std::atol(e));
#endif
    while (done < dev_ids.size()) {
        // as in search_device: an observation is read only behind the event of the round that wrote it
        bool arm_report = true;
        if (sc->report_pending) {
            if (hipEventQuery(sc->report_done) == hipSuccess) {
                const uint32_t redone = sc->report[0], of = sc->report[1];
                if (of != 0) {
                    if (redone != 0) {
                        work.spill = (uint32_t*)sc->spill.p;
                        work.spill_tables = sc->spill_tables;
                    }
                    work.hbits_bias = next_hbits_bias(work.hbits_bias, redone, of);
                    if (sc->report[3] != 0) work.vis_mean = (uint32_t)std::min<uint64_t>((uint64_t)sc->report[3] * 16u / of, 0x7FFFFFFFu);
                }
                sc->report[0] = sc->report[1] = sc->report[3] = 0;
                sc->report_pending = false;
            } else {
                (void)hipGetLastError();
                arm_report = false;
            }
        }
        work.report = arm_report ? sc->report : nullptr;
        // a batch is searched against the graph WITHOUT its own members: keep it a fraction of it
        // (a quarter at M >= 8; a sixteenth below: with four links per element a batch that large cost 0.02-0.03 of
        // recall@10 on 2 000-row sets, profiles/r2/fuzz_build.log)
        const size_t nb = std::min<size_t>(std::min<size_t>(max_batch, std::max<size_t>(linked / batch_div, 8)), dev_ids.size() - done);
        const uint32_t* bid = (const uint32_t*)ids.p + done;
        if (launch_gather_rows(g0.vecs, g0.vstride, g0.dim, bid, (uint32_t)nb, (float*)q.p, stream) != 0)
            throw DeviceError("gather kernel launch failed");
        int rc = launch_hnsw_search(gb, (const float*)q.p, (uint32_t)nb, efc, efc, nullptr, (float*)cd.p, (uint64_t*)cl.p, nullptr,
                                    work, stream, &err);
        if (rc == 0 && arm_report) {
            VDB_HIP(hipEventRecord(sc->report_done, stream));
            sc->report_pending = true;
        }
        if (rc == 0)
            rc = launch_build_round(gb, links0, (const uint64_t*)cl.p, (const float*)cd.p, bid, (uint32_t)nb, efc, M, maxM0, w,
                                    stream, &err);
        if (rc != 0) throw DeviceError(err);
        // A round depends on the one before it, so waiting for it here costs the launch latency only (20 us of a
        // 30 ms round) - and without the wait the host enqueues thousands of rounds before the first one's report
        // exists: the whole build then ran with the visited-set size and the spill setting of its first round
        // (100 M x 128: 161 s against 103 s).
        if (sc->report_pending) VDB_HIP(hipStreamSynchronize(stream));
        done += nb;
        linked += nb;
        if (tr.on && ++round % 256 == 0) {
            VDB_HIP(hipStreamSynchronize(stream));
            std::fprintf(stderr, "[vdb build]   round %zu: %zu of %zu elements, visited-set bias %u, %.1fs\n", round, done,
                         dev_ids.size(), work.hbits_bias,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - tr.last).count());
        }
    }
    VDB_HIP(hipStreamSynchronize(stream));
    sc->report[0] = sc->report[1] = sc->report[3] = 0;
    sc->report_pending = false;
    tr.lap("device rounds (layer 0)", dev_ids.size());
    {
        uint32_t found = 0, fixed = 0;
        bool repair = true;
#ifdef VDB_DIAG  // (make DIAG=1: tools/exp_build_small_m.py)
        repair = std::getenv("VDB_BUILD_NO_REPAIR") == nullptr;
#endif
        if (repair && build_repair_orphans(links0, (uint32_t)total, g0.l0stride, maxM0, stream, &found, &fixed, &err) != 0)
            throw DeviceError(err);
        if (tr.on) std::fprintf(stderr, "[vdb build] elements without an inbound link on layer 0: %u, given one: %u\n", found, fixed);
        tr.lap("orphan repair", found);
    }
    // the device lists are the graph now: write them back in the reference's layout (:259-261, :283-285)
    const uint32_t ls = g0.l0stride;
    const size_t chunk = 1 << 16;
    std::vector<uint32_t> rows(chunk * ls);
    for (size_t lo = 0; lo < total; lo += chunk) {
        const size_t m = std::min(chunk, total - lo);
        VDB_HIP(hipMemcpy(rows.data(), links0 + lo * ls, m * ls * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < m; ++i) {
            uint32_t* list = h.list0((uint32_t)(lo + i));
            const uint32_t* r = rows.data() + i * ls;
            unsigned c = 0;
            while (c < maxM0 && r[c] != kNoId) ++c;
            const unsigned char flags = ((const unsigned char*)list)[2];
            std::memset(list, 0, h.link0_stride);
            HostIndex::set_list_count(list, c);
            ((unsigned char*)list)[2] = flags;
            std::memcpy(list + 1, r, (size_t)c * 4);
        }
    }
    h.version += 1;
    idx->device_version = h.version;  // the device copy IS this graph
    tr.lap("write-back", total);
}

// Links every element of an index whose rows are stored and whose graph is EMPTY.
//
// Layer l >= 1 of an HNSW index is itself the layer 0 of an HNSW over the elements of level >= l
// (same vectors, lists of capacity maxM instead of maxM0, levels one lower).  So: build that smaller
// index first (recursively; small ones by the host builder, the reference algorithm), copy its lists
// in as the upper layers, seed layer 0 of the elements with upper levels with their layer-1 lists
// (a connected graph over them), and insert everything else into layer 0 by the device rounds; the
// reverse links of those rounds fill and re-prune the seeded lists like any other.
void build_empty_graph(vdb_index* idx, int n_threads, int device, size_t host_base, BuildTrace& tr) {
    HostIndex& h = *idx->host;
    const size_t n = h.count;
    if (n <= host_base) {
        std::vector<uint32_t> all(n);
        for (size_t i = 0; i < n; ++i) all[i] = (uint32_t)i;
        h.link_elements(all.data(), n, n_threads);
        tr.lap("host-built graph (reference algorithm)", n);
        return;
    }
    std::vector<uint32_t> s1, dev_ids;
    for (size_t i = 0; i < n; ++i) (h.levels[i] > 0 ? s1 : dev_ids).push_back((uint32_t)i);
    if (s1.empty()) {  // (n > host_base makes this practically impossible) the first element anchors the graph
        s1.push_back(dev_ids.front());
        dev_ids.erase(dev_ids.begin());
    }
    {
        vdb_index sub;
        sub.host.reset(new HostIndex(h.metric, h.dim, s1.size(), h.M, h.ef_construction, 100, /*layer0_lists_as_upper=*/true));
        HostIndex& hs = *sub.host;
        hs.set_sum_order(h.sum_order);
        {
            std::vector<float> v(std::min<size_t>(s1.size(), 1 << 16) * h.dim);
            std::vector<int> lv(std::min<size_t>(s1.size(), 1 << 16));
            for (size_t lo = 0; lo < s1.size(); lo += 1 << 16) {
                const size_t m = std::min<size_t>(1 << 16, s1.size() - lo);
                for (size_t i = 0; i < m; ++i) {
                    std::memcpy(v.data() + i * h.dim, h.vec(s1[lo + i]), h.dim * 4);
                    lv[i] = std::max(h.levels[s1[lo + i]] - 1, 0);
                }
                hs.append_unlinked(v.data(), nullptr, m, -1, lv.data());
            }
        }
        build_empty_graph(&sub, n_threads, device, host_base, tr);
        // its lists are this index's upper layers (and the seed of layer 0), ids mapped back
        for (size_t j = 0; j < s1.size(); ++j) {
            const uint32_t gid = s1[j];
            h.alloc_upper_lists(gid);
            for (int l = 0; l <= hs.levels[j]; ++l) {
                const uint32_t* src = hs.list_at((uint32_t)j, l);
                const unsigned c = HostIndex::list_count(src);
                if (l + 1 <= h.levels[gid]) {
                    uint32_t* dst = h.list_at(gid, l + 1);
                    HostIndex::set_list_count(dst, c);
                    for (unsigned i = 0; i < c; ++i) dst[1 + i] = s1[src[1 + i]];
                }
                if (l == 0) {
                    uint32_t* dst = h.list0(gid);
                    HostIndex::set_list_count(dst, c);
                    for (unsigned i = 0; i < c; ++i) dst[1 + i] = s1[src[1 + i]];
                }
            }
        }
        h.set_entry(s1[hs.entry], hs.max_level + 1);
        tr.lap("upper layers taken from the sub-index", s1.size());
    }  // the sub-index and its device copy go away here
    device_rounds(idx, dev_ids, s1.size(), device, tr);
}

}  // namespace

extern "C" {

// addPoint for a batch with the graph built on the device (hnsw_build.hip).
int vdb_index_add_batch_device(vdb_index* idx, const float* vecs, const uint64_t* labels, size_t n, int n_threads, int device) {
    return guarded([&] {
        if (!idx || (!vecs && n)) throw std::invalid_argument("null argument");
        if (n == 0) return (int)VDB_OK;
        HostIndex& h = *idx->host;
        if (h.deleted != 0) throw Unsupported("device graph build with deleted elements in the index");
        if (h.ef_construction > kMaxEf) throw Unsupported("device graph build needs ef_construction <= 508");
        if (round_up(h.maxM0, 32) > kMaxL0Stride) throw Unsupported("M above 32 is not supported by the device path");
        if (round_up(h.dim, 32) > kMaxVecStride) throw Unsupported("dim above 1024 is not supported by the device path");
        {  // nothing is stored unless the device is there
            int ndev = 0;
            if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw DeviceError("no HIP device visible");
            if (device < 0 || device >= ndev) throw std::invalid_argument("device ordinal out of range");
            if (h.count + n > h.capacity) throw std::runtime_error("The number of elements exceeds the specified limit");
        }
        BuildTrace tr;
        const uint64_t rng0 = h.rng_state();
        const size_t base = h.append_unlinked(vecs, labels, n);
        tr.lap("rows stored", n);
        // A failure from here on must not leave stored elements that no search can reach.  A graph that
        // was empty goes back to empty (rows, labels, levels and the level generator as before the call);
        // for an append the host graph is intact until the rounds' write-back, so the elements the device
        // did not link are linked by the host builder (the reference algorithm) and the error is returned.
        auto undo_new_graph = [&] {
            idx->drop_device();
            h.truncate_to_empty();
            h.set_rng_state(rng0);
        };
        // A new graph: every layer through the device rounds (small sub-graphs on the host).  Not for
        // M < 8: there a quarter or more of the elements have upper levels, and their layer-0 lists,
        // seeded from layer 1 and filled by reverse links only, cost recall (0.935 against 0.981 for
        // the host build at M = 4) - those go the way of an append below.
        if (base == 0 && h.M >= 8) {
            size_t host_base = 8192;
            if (const char* e = std::getenv("VDB_BUILD_HOST_BASE")) host_base = (size_t)std::max(64l, std::atol(e));
            try {
                build_empty_graph(idx, n_threads, device, host_base, tr);
                if (idx->device < 0 || idx->device_version != h.version) upload(idx, device);
            } catch (...) {
                undo_new_graph();
                throw;
            }
            return (int)VDB_OK;
        }
        // appending to an existing graph: the host builder (reference algorithm) inserts the elements
        // that have upper levels, the others go into layer 0 by the device rounds
        std::vector<uint32_t> host_ids, dev_ids;
        for (size_t i = 0; i < n; ++i) (h.levels[base + i] > 0 ? host_ids : dev_ids).push_back((uint32_t)(base + i));
        if (h.entry == kNoId && host_ids.empty()) {  // an empty graph and no element drew an upper level:
            host_ids.push_back(dev_ids.front());     // the first one anchors it (sets the entry point, :1099-1102)
            dev_ids.erase(dev_ids.begin());
        }
        try {
            h.link_elements(host_ids.data(), host_ids.size(), n_threads);
            tr.lap("host-linked elements with upper levels", host_ids.size());
            if (dev_ids.empty()) {
                upload(idx, device);
                return (int)VDB_OK;
            }
        } catch (...) {
            if (base == 0) undo_new_graph();
            throw;
        }
        try {
            device_rounds(idx, dev_ids, base + host_ids.size(), device, tr);
        } catch (const std::exception& e) {
            if (base == 0) {
                undo_new_graph();
                throw;
            }
            idx->drop_device();
            const std::string why = e.what();
            h.link_elements(dev_ids.data(), dev_ids.size(), n_threads);
            throw DeviceError(why + " (the device rounds failed; the " + std::to_string(dev_ids.size()) +
                              " elements they were to link were linked by the host builder instead)");
        }
        return (int)VDB_OK;
    });
}

int vdb_index_set_build_sum_order(vdb_index* idx, int order) {
    return guarded([&] {
        if (!idx) throw std::invalid_argument("null index");
        if (order != VDB_SUM_SEQ && order != VDB_SUM_TREE64) throw std::invalid_argument("unknown summation order");
        idx->host->set_sum_order(order);
        return (int)VDB_OK;
    });
}

int vdb_index_set_ef(vdb_index* idx, size_t ef) {
    return guarded([&] {
        if (!idx) throw std::invalid_argument("null index");
        idx->host->ef_search = ef;
        return (int)VDB_OK;
    });
}

int vdb_index_set_spill_mode(vdb_index* idx, int mode) {
    return guarded([&] {
        if (!idx) throw std::invalid_argument("null index");
        if (mode < -1 || mode > 1) throw std::invalid_argument("spill mode must be -1 (adaptive), 0 (never) or 1 (always)");
        std::unique_lock<std::mutex> lock(idx->scratch_lock);
        idx->spill_mode = mode;
        return (int)VDB_OK;
    });
}

int vdb_index_mark_delete(vdb_index* idx, uint64_t label) {
    return guarded([&] {
        if (!idx) throw std::invalid_argument("null index");
        idx->host->mark_delete(label, false);
        return (int)VDB_OK;
    });
}

int vdb_index_unmark_delete(vdb_index* idx, uint64_t label) {
    return guarded([&] {
        if (!idx) throw std::invalid_argument("null index");
        idx->host->mark_delete(label, true);
        return (int)VDB_OK;
    });
}

int vdb_index_get_data_by_label(const vdb_index* idx, uint64_t label, float* out) {
    return guarded([&] {
        if (!idx || !out) throw std::invalid_argument("null argument");
        idx->host->get_data_by_label(label, out);
        return (int)VDB_OK;
    });
}

int vdb_index_resize(vdb_index* idx, size_t new_capacity) {
    return guarded([&] {
        if (!idx) throw std::invalid_argument("null index");
        idx->host->resize(new_capacity);
        return (int)VDB_OK;
    });
}

int vdb_index_check_integrity(const vdb_index* idx, int* min_inbound, int* max_inbound) {
    if (!idx) return -1;
    int verdict = -1;
    const int rc = guarded([&] {  // (throws for an index whose rows are not on the host: no exception crosses the C ABI)
        verdict = idx->host->check_integrity(min_inbound, max_inbound);
        return (int)VDB_OK;
    });
    return rc == VDB_OK ? verdict : -1;
}

size_t vdb_index_capacity(const vdb_index* idx) { return idx ? idx->host->capacity : 0; }
size_t vdb_index_count(const vdb_index* idx) { return idx ? (size_t)idx->host->count : 0; }
size_t vdb_index_deleted_count(const vdb_index* idx) { return idx ? (size_t)idx->host->deleted : 0; }
size_t vdb_index_dim(const vdb_index* idx) { return idx ? idx->host->dim : 0; }
size_t vdb_index_M(const vdb_index* idx) { return idx ? idx->host->M : 0; }
size_t vdb_index_ef(const vdb_index* idx) { return idx ? idx->host->ef_search : 0; }
int vdb_index_max_level(const vdb_index* idx) { return idx ? idx->host->max_level : -1; }
uint32_t vdb_index_entry(const vdb_index* idx) { return idx ? idx->host->entry : 0xFFFFFFFFu; }
int vdb_index_level_of(const vdb_index* idx, uint32_t id) {
    if (!idx || id >= idx->host->count) return -1;
    return idx->host->levels[id];
}
int vdb_index_links(const vdb_index* idx, uint32_t id, int level, uint32_t* out, int cap) {
    if (!idx || id >= idx->host->count || level < 0 || level > idx->host->levels[id]) return -1;
    if (cap > 0 && !out) return -1;
    if (level == 0 && !idx->host->rows_resident) {  // opened by vdb_index_load_to_device: layer 0 lives on the device only
        set_error(VDB_ERR_RUNTIME, "get_linklist0: the level-0 rows of this index are not held on the host (vdb_index_load_to_device)");
        return -1;
    }
    int n = -1;
    const int rc = guarded([&] {  // no exception crosses the C ABI
        const uint32_t* list = idx->host->list_at(id, level);
        n = (int)HostIndex::list_count(list);
        for (int i = 0; i < n && i < cap; ++i) out[i] = list[1 + i];
        return (int)VDB_OK;
    });
    return rc == VDB_OK ? n : -1;
}

int vdb_index_to_device(vdb_index* idx, int device) {
    return guarded([&] {
        if (!idx) throw std::invalid_argument("null index");
        upload(idx, device);
        return (int)VDB_OK;
    });
}

int vdb_index_load_to_device(vdb_index** out, const char* path, int metric, size_t dim, int device) {
    return guarded([&] {
        if (!out || !path) throw std::invalid_argument("null argument");
        if (metric != VDB_L2 && metric != VDB_IP) throw std::invalid_argument("unknown metric");
        std::unique_ptr<vdb_index> idx(new vdb_index);
        idx->host.reset(new HostIndex(std::string(path), metric, dim, 0, /*stream_rows=*/true));
        upload(idx.get(), device);
        *out = idx.release();
        return (int)VDB_OK;
    });
}

size_t vdb_index_device_bytes(const vdb_index* idx) { return idx ? idx->device_bytes() : 0; }

int vdb_search_batch_device(const vdb_index* cidx, const float* d_queries, size_t nq, size_t k, float* d_out_dist,
                            uint64_t* d_out_label, uint32_t* d_out_stats, void* stream_) {
    return guarded([&] {
        return search_device(const_cast<vdb_index*>(cidx), d_queries, nq, k, nullptr, d_out_dist, d_out_label, d_out_stats,
                             (hipStream_t)stream_, nullptr);
    });
}

int vdb_search_batch_device_filtered(const vdb_index* cidx, const float* d_queries, size_t nq, size_t k,
                                     const uint32_t* d_allow, float* d_out_dist, uint64_t* d_out_label,
                                     uint32_t* d_out_stats, void* stream_) {
    return guarded([&] {
        return search_device(const_cast<vdb_index*>(cidx), d_queries, nq, k, d_allow, d_out_dist, d_out_label, d_out_stats,
                             (hipStream_t)stream_, nullptr);
    });
}

// vdb_search_batch_device for callers that cannot allocate: no hipMalloc / hipFree / hipHostMalloc here or
// below (launch_hnsw_search enqueues one hipMemsetAsync and the kernels); the work list and the visited
// bitmaps of the large configuration live in the library's static device array (noalloc_scratch.hip).
int vdb_search_batch_device_noalloc(const vdb_index* cidx, const float* d_queries, size_t nq, size_t k, float* d_out_dist,
                                    uint64_t* d_out_label, uint32_t* d_out_stats, void* stream_) {
    return guarded([&] {
        vdb_index* idx = const_cast<vdb_index*>(cidx);
        require_search_ready(idx);
        if (nq == 0) return (int)VDB_OK;
        if (!d_queries || !d_out_dist || !d_out_label) throw std::invalid_argument("null buffer");
        if (k == 0) throw std::invalid_argument("k must be > 0");
        if (nq > kNoAllocMaxQueries)
            throw std::invalid_argument("vdb_search_batch_device_noalloc: at most " + std::to_string(kNoAllocMaxQueries) +
                                        " queries per call (split the batch, or use vdb_search_batch_device)");
        if (idx->graph.n > kNoAllocMaxElements)
            throw std::invalid_argument("vdb_search_batch_device_noalloc: at most " + std::to_string(kNoAllocMaxElements) +
                                        " elements per index (use vdb_search_batch_device)");
        const size_t ef = std::max(idx->host->ef_search, k);  // hnsw_core.hpp:1144
        if (ef > kMaxEf) throw Unsupported("max(efSearch, k) above 508 is not supported by the device path");
        DeviceGuard guard(idx->device);
        void* base = nullptr;
        std::string err;
        int rc = noalloc_scratch(idx->device, &base, &err);
        if (rc != 0) return set_error(rc, err);
        HnswSearchWork work;
        work.redo = (uint32_t*)base;
        work.bitmaps = (uint32_t*)((char*)base + kNoAllocBitmapOffset);
        work.bitmap_words = std::max((idx->graph.n + 31u) / 32u, 1u);
        work.large_blocks = kNoAllocLargeBlocks;
        {
            std::unique_lock<std::mutex> lock(idx->scratch_lock);
            auto it = idx->hbits_bias.find((uint32_t)ef);
            work.hbits_bias = it == idx->hbits_bias.end() ? 0u : it->second;
            auto iv = idx->vis_mean.find((uint32_t)ef);
            work.vis_mean = (iv == idx->vis_mean.end() || idx->live.p) ? 0u : iv->second;
        }
        rc = launch_hnsw_search(idx->graph, d_queries, (uint32_t)nq, (uint32_t)k, (uint32_t)ef, (const uint32_t*)idx->live.p,
                                d_out_dist, d_out_label, d_out_stats, work, (hipStream_t)stream_, &err);
        if (rc != 0) return set_error(rc, err);
        return (int)VDB_OK;
    });
}

int vdb_search_batch(const vdb_index* idx, const float* queries, size_t nq, size_t k, float* out_dist,
                     uint64_t* out_label, uint32_t* out_stats) {
    return guarded([&] {
        return search_host(const_cast<vdb_index*>(idx), queries, nq, k, nullptr, out_dist, out_label, out_stats);
    });
}

int vdb_search_batch_filtered(const vdb_index* idx, const float* queries, size_t nq, size_t k, const uint32_t* allow,
                              float* out_dist, uint64_t* out_label, uint32_t* out_stats) {
    return guarded([&] {
        return search_host(const_cast<vdb_index*>(idx), queries, nq, k, allow, out_dist, out_label, out_stats);
    });
}

int vdb_search_epsilon_batch_device(const vdb_index* cidx, const float* d_queries, size_t nq, float epsilon,
                                    size_t min_num_candidates, size_t max_num_candidates, const uint32_t* d_allow, float* d_out_dist,
                                    uint64_t* d_out_label, uint32_t* d_out_stats, void* stream_) {
    return guarded([&] {
        const EpsilonStop stop{epsilon, min_num_candidates, max_num_candidates};
        return search_device(const_cast<vdb_index*>(cidx), d_queries, nq, max_num_candidates, d_allow, d_out_dist, d_out_label,
                             d_out_stats, (hipStream_t)stream_, nullptr, &stop);
    });
}

int vdb_search_epsilon_batch(const vdb_index* idx, const float* queries, size_t nq, float epsilon, size_t min_num_candidates,
                             size_t max_num_candidates, const uint32_t* allow, float* out_dist, uint64_t* out_label,
                             uint32_t* out_stats) {
    return guarded([&] {
        const EpsilonStop stop{epsilon, min_num_candidates, max_num_candidates};
        return search_host(const_cast<vdb_index*>(idx), queries, nq, max_num_candidates, allow, out_dist, out_label, out_stats, &stop);
    });
}

int vdb_index_labels(const vdb_index* idx, uint64_t* out, size_t cap) {
    return guarded([&] {
        if (!idx || !out) throw std::invalid_argument("null argument");
        const size_t n = idx->host->count;
        idx->host->require_rows("getExternalLabel");
        if (cap < n) throw std::invalid_argument("label buffer smaller than the element count");
        for (size_t i = 0; i < n; ++i) out[i] = idx->host->label_of((uint32_t)i);
        return (int)VDB_OK;
    });
}

// ---- BruteforceSearch<float> ----------------------------------------------------------

int vdb_bf_create(vdb_bruteforce** out, int metric, size_t dim, size_t capacity) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("null output pointer");
        if (metric != VDB_L2 && metric != VDB_IP) throw std::invalid_argument("unknown metric");
        std::unique_ptr<vdb_bruteforce> bf(new vdb_bruteforce);
        bf->host.reset(new HostBruteforce(metric, dim, capacity));
        *out = bf.release();
        return (int)VDB_OK;
    });
}

int vdb_bf_load(vdb_bruteforce** out, const char* path, int metric, size_t dim) {
    return guarded([&] {
        if (!out || !path) throw std::invalid_argument("null argument");
        if (metric != VDB_L2 && metric != VDB_IP) throw std::invalid_argument("unknown metric");
        std::unique_ptr<vdb_bruteforce> bf(new vdb_bruteforce);
        bf->host.reset(new HostBruteforce(std::string(path), metric, dim));
        *out = bf.release();
        return (int)VDB_OK;
    });
}

int vdb_bf_save(const vdb_bruteforce* bf, const char* path) {
    return guarded([&] {
        if (!bf || !path) throw std::invalid_argument("null argument");
        bf->host->save(path);
        return (int)VDB_OK;
    });
}

void vdb_bf_free(vdb_bruteforce* bf) { delete bf; }

int vdb_bf_add_batch(vdb_bruteforce* bf, const float* vecs, const uint64_t* labels, size_t n) {
    return guarded([&] {
        if (!bf || (!vecs && n)) throw std::invalid_argument("null argument");
        const size_t dim = bf->host->dim;
        const size_t base = bf->host->count;
        for (size_t i = 0; i < n; ++i) bf->host->add(vecs + i * dim, labels ? labels[i] : (uint64_t)(base + i));
        return (int)VDB_OK;
    });
}

int vdb_bf_remove(vdb_bruteforce* bf, uint64_t label) {
    return guarded([&] {
        if (!bf) throw std::invalid_argument("null index");
        bf->host->remove(label);
        return (int)VDB_OK;
    });
}

size_t vdb_bf_count(const vdb_bruteforce* bf) { return bf ? bf->host->count : 0; }

int vdb_bf_to_device(vdb_bruteforce* bf, int device) {
    return guarded([&] {
        if (!bf) throw std::invalid_argument("null index");
        upload_bf(bf, device);
        return (int)VDB_OK;
    });
}

}  // extern "C"

namespace {

// SearchKNN over a batch; d_allow = nullable device bitmap over ROWS (a BaseFilterFunctor's verdicts).
int bf_search_device(vdb_bruteforce* bf, const float* d_queries, size_t nq, size_t k, int precision, const uint32_t* d_allow,
                     float* d_out_dist, uint64_t* d_out_label, hipStream_t stream) {
    if (!bf) throw std::invalid_argument("null index");
    if (bf->device < 0) throw DeviceError("index is not on a device: call vdb_bf_to_device first");
    if (bf->device_version != bf->host->version) throw DeviceError("index changed since vdb_bf_to_device: upload it again");
    if (nq == 0) return (int)VDB_OK;
    if (!d_queries || !d_out_dist || !d_out_label) throw std::invalid_argument("null buffer");
    if (k == 0) throw std::invalid_argument("k must be > 0");
    if (nq > 0x7FFFFFFFull) throw std::invalid_argument("batch too large");
    if (precision != VDB_BF_FP32 && precision != VDB_BF_BF16 && precision != VDB_BF_BF16_RERANK)
        throw std::invalid_argument("unknown brute-force precision");
    if (k > kBfMaxK) throw Unsupported("brute force: k above 1024 is not supported by the device path");
    // small problems and very large k go straight to the exact scan (same results, no setup cost); so does a
    // filtered search: the reference's filtered scan depends on the row order (bf_search.hip, bf_first_rows_bound)
    const bool exact_only = d_allow != nullptr || (precision == VDB_BF_FP32 && (bf->n < 4 * kBfMfmaCap || k + 32 > kBfMfmaCap / 2));
    DeviceGuard guard(bf->device);
    std::unique_lock<std::mutex> lock(bf->scratch_lock);
    auto& ws = bf->scratch[stream];
    if (!ws) ws.reset(new DevBuf);
    const size_t need = exact_only ? bf_exact_workspace_bytes((uint32_t)nq) : bf_mfma_workspace_bytes((uint32_t)nq, bf->vstride, bf->kp64);
    if (ws->bytes < need) {
        if (ws->p) VDB_HIP(hipStreamSynchronize(stream));
        ws->alloc(need);
    }
    std::string err;
    int rc;
    if (exact_only) {
        rc = launch_bf_exact((const float*)bf->vecs.p, (const uint64_t*)bf->labels.p, bf->n, (uint32_t)bf->host->dim, bf->vstride,
                             bf->host->metric, d_queries, (uint32_t)nq, (uint32_t)k, d_out_dist, d_out_label, ws->p, stream, &err,
                             nullptr, nullptr, d_allow);
    } else {
        BfDeviceData d;
        d.raw = (const float*)bf->vecs.p;
        d.xc = (const float*)bf->xc.p;
        d.xb = bf->xb.p;
        d.nx = (const float*)bf->nx.p;
        d.mu = (const float*)bf->mu.p;
        d.stats = (const float*)bf->stats.p;
        d.labels = (const uint64_t*)bf->labels.p;
        d.n = bf->n;
        d.dim = (uint32_t)bf->host->dim;
        d.vstride = bf->vstride;
        d.kp64 = bf->kp64;
        d.metric = bf->host->metric;
        rc = launch_bf_mfma(d, precision, d_queries, (uint32_t)nq, (uint32_t)k, d_out_dist, d_out_label, ws->p, stream, &err);
    }
    if (rc != 0) return set_error(rc, err);
    return (int)VDB_OK;
}

int bf_search_host(vdb_bruteforce* bf, const float* queries, size_t nq, size_t k, int precision, const uint32_t* allow,
                   float* out_dist, uint64_t* out_label) {
    if (!bf) throw std::invalid_argument("null index");
    if (bf->device < 0) throw DeviceError("index is not on a device: call vdb_bf_to_device first");
    if (nq == 0) return (int)VDB_OK;
    if (!queries || !out_dist || !out_label) throw std::invalid_argument("null buffer");
    if (k == 0) throw std::invalid_argument("k must be > 0");
    DeviceGuard guard(bf->device);
    const size_t dim = bf->host->dim;
    HostStage& st = bf->stage;
    std::unique_lock<std::mutex> stage_guard(st.lock);
    st.q.reserve(nq * dim * 4);
    st.dist.reserve(nq * k * 4);
    st.label.reserve(nq * k * 8);
    VDB_HIP(hipMemcpy(st.q.p, queries, nq * dim * 4, hipMemcpyHostToDevice));
    if (allow) {
        const size_t words = ((size_t)bf->n + 31) / 32;
        st.filter.reserve(std::max<size_t>(words, 1) * 4);
        VDB_HIP(hipMemcpy(st.filter.p, allow, words * 4, hipMemcpyHostToDevice));
    }
    const int rc = bf_search_device(bf, (const float*)st.q.p, nq, k, precision, allow ? (const uint32_t*)st.filter.p : nullptr,
                                    (float*)st.dist.p, (uint64_t*)st.label.p, nullptr);
    if (rc != VDB_OK) return rc;
    VDB_HIP(hipStreamSynchronize(nullptr));
    VDB_HIP(hipMemcpy(out_dist, st.dist.p, nq * k * 4, hipMemcpyDeviceToHost));
    VDB_HIP(hipMemcpy(out_label, st.label.p, nq * k * 8, hipMemcpyDeviceToHost));
    return (int)VDB_OK;
}

}  // namespace

extern "C" {

int vdb_bf_search_batch_device(const vdb_bruteforce* bf, const float* d_queries, size_t nq, size_t k, int precision,
                               float* d_out_dist, uint64_t* d_out_label, void* stream) {
    return guarded([&] {
        return bf_search_device(const_cast<vdb_bruteforce*>(bf), d_queries, nq, k, precision, nullptr, d_out_dist, d_out_label,
                                (hipStream_t)stream);
    });
}

int vdb_bf_search_batch_device_filtered(const vdb_bruteforce* bf, const float* d_queries, size_t nq, size_t k, const uint32_t* d_allow,
                                        float* d_out_dist, uint64_t* d_out_label, void* stream) {
    return guarded([&] {
        return bf_search_device(const_cast<vdb_bruteforce*>(bf), d_queries, nq, k, VDB_BF_FP32, d_allow, d_out_dist, d_out_label,
                                (hipStream_t)stream);
    });
}

int vdb_bf_search_batch_filtered(const vdb_bruteforce* bf, const float* queries, size_t nq, size_t k, const uint32_t* allow,
                                 float* out_dist, uint64_t* out_label) {
    return guarded([&] {
        return bf_search_host(const_cast<vdb_bruteforce*>(bf), queries, nq, k, VDB_BF_FP32, allow, out_dist, out_label);
    });
}

int vdb_bf_labels(const vdb_bruteforce* bf, uint64_t* out, size_t cap) {
    return guarded([&] {
        if (!bf || !out) throw std::invalid_argument("null argument");
        HostBruteforce& h = *bf->host;
        std::unique_lock<std::mutex> g(h.lock);
        if (cap < h.count) throw std::invalid_argument("label buffer smaller than the element count");
        for (size_t i = 0; i < h.count; ++i) std::memcpy(&out[i], h.data + i * h.stride + h.data_size, 8);
        return (int)VDB_OK;
    });
}

int vdb_bf_search_batch(const vdb_bruteforce* bf, const float* queries, size_t nq, size_t k, int precision,
                        float* out_dist, uint64_t* out_label) {
    return guarded([&] {
        return bf_search_host(const_cast<vdb_bruteforce*>(bf), queries, nq, k, precision, nullptr, out_dist, out_label);
    });
}

// ---- base-set sharding over RCCL ------------------------------------------------------------

}  // extern "C"

namespace {

// RCCL is bound with dlopen at the first use: a process that never shards needs no librccl, and one that has
// PyTorch loaded gets PyTorch's copy (same SONAME) instead of a second one.
struct RcclApi {
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string error;
};

const RcclApi& rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            const char* why = dlerror();  // (a second call returns NULL: the first one clears the message)
            a.error = std::string("RCCL is not loadable (librccl.so.1): ") + (why ? why : "?");
            return a;
        }
        auto bind = [&](auto& fn, const char* sym) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, sym));
            if (!fn && a.error.empty()) a.error = std::string("RCCL symbol missing: ") + sym;
        };
        bind(a.get_unique_id, "ncclGetUniqueId");
        bind(a.comm_init_rank, "ncclCommInitRank");
        bind(a.comm_destroy, "ncclCommDestroy");
        bind(a.comm_count, "ncclCommCount");
        bind(a.comm_user_rank, "ncclCommUserRank");
        bind(a.all_gather, "ncclAllGather");
        bind(a.group_start, "ncclGroupStart");
        bind(a.group_end, "ncclGroupEnd");
        bind(a.error_string, "ncclGetErrorString");
        return a;
    }();
    if (!api.error.empty()) throw DeviceError(api.error);
    return api;
}

#define VDB_RCCL(call)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) throw DeviceError(std::string(#call) + ": " + rccl_api().error_string(r_));       \
    } while (0)

}  // namespace

struct vdb_comm {
    ncclComm_t comm = nullptr;
    bool owned = false;
    int n_ranks = 0, rank = 0, device = -1;
    std::mutex lock;
    DevBuf my_dist, my_label, all_dist, all_label;  // [nq][k] and [n_ranks][nq][k], grow only
    ~vdb_comm() {
        if (device >= 0) {
            try {
                DeviceGuard g(device);
                (void)hipDeviceSynchronize();
                my_dist.release();
                my_label.release();
                all_dist.release();
                all_label.release();
                if (owned && comm) (void)rccl_api().comm_destroy(comm);
            } catch (...) {
            }
        }
    }
};

extern "C" {

int vdb_comm_unique_id(void* id128) {
    return guarded([&] {
        if (!id128) throw std::invalid_argument("null buffer");
        ncclUniqueId id;
        VDB_RCCL(rccl_api().get_unique_id(&id));
        static_assert(sizeof(id) == VDB_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
        std::memcpy(id128, &id, sizeof(id));
        return (int)VDB_OK;
    });
}

int vdb_comm_create(vdb_comm** out, const void* id128, int n_ranks, int rank, int device) {
    return guarded([&] {
        if (!out || !id128) throw std::invalid_argument("null argument");
        if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("rank outside 0 .. n_ranks-1");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw DeviceError("no HIP device visible");
        if (device < 0 || device >= ndev) throw std::invalid_argument("device ordinal out of range");
        std::unique_ptr<vdb_comm> c(new vdb_comm);
        DeviceGuard guard(device);
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        VDB_RCCL(rccl_api().comm_init_rank(&c->comm, n_ranks, id, rank));
        c->owned = true;
        c->n_ranks = n_ranks;
        c->rank = rank;
        c->device = device;
        *out = c.release();
        return (int)VDB_OK;
    });
}

int vdb_comm_wrap(vdb_comm** out, void* nccl_comm, int device) {
    return guarded([&] {
        if (!out || !nccl_comm) throw std::invalid_argument("null argument");
        std::unique_ptr<vdb_comm> c(new vdb_comm);
        c->comm = (ncclComm_t)nccl_comm;
        VDB_RCCL(rccl_api().comm_count(c->comm, &c->n_ranks));
        VDB_RCCL(rccl_api().comm_user_rank(c->comm, &c->rank));
        c->device = device;
        *out = c.release();
        return (int)VDB_OK;
    });
}

void vdb_comm_free(vdb_comm* c) { delete c; }
int vdb_comm_size(const vdb_comm* c) { return c ? c->n_ranks : 0; }
int vdb_comm_rank(const vdb_comm* c) { return c ? c->rank : -1; }

int vdb_sharded_search_batch_device(const vdb_index* shard, vdb_comm* comm, const float* d_queries, size_t nq, size_t k,
                                    float* d_out_dist, uint64_t* d_out_label, void* stream_) {
    return guarded([&] {
        if (!shard || !comm) throw std::invalid_argument("null argument");
        if (nq == 0) return (int)VDB_OK;
        if (!d_queries || !d_out_dist || !d_out_label) throw std::invalid_argument("null buffer");
        if (k == 0) throw std::invalid_argument("k must be > 0");
        if ((size_t)comm->n_ranks * k > 5461) throw std::invalid_argument("sharded search: n_ranks * k must be at most 5461");
        if (shard->device != comm->device) throw std::invalid_argument("the shard and the communicator are on different devices");
        hipStream_t stream = (hipStream_t)stream_;
        DeviceGuard guard(comm->device);
        std::unique_lock<std::mutex> lock(comm->lock);
        const size_t per = nq * k, G = (size_t)comm->n_ranks;
        if (comm->my_dist.bytes < per * 4 || comm->all_dist.bytes < G * per * 4) {
            VDB_HIP(hipStreamSynchronize(stream));  // an earlier call may still be using the old buffers
            comm->my_dist.reserve(per * 4);
            comm->my_label.reserve(per * 8);
            comm->all_dist.reserve(G * per * 4);
            comm->all_label.reserve(G * per * 8);
        }
        // 1. this rank's shard (hnsw_core.hpp:1107-1159 on the sub-graph)
        int rc = search_device(const_cast<vdb_index*>(shard), d_queries, nq, k, nullptr, (float*)comm->my_dist.p,
                               (uint64_t*)comm->my_label.p, nullptr, stream, nullptr);
        if (rc != VDB_OK) return rc;
        // 2. the one exchange step: every rank's [nq][k] lists to every rank, both tensors in one group
        const RcclApi& api = rccl_api();
        VDB_RCCL(api.group_start());
        VDB_RCCL(api.all_gather(comm->my_dist.p, comm->all_dist.p, per, ncclFloat32, comm->comm, stream));
        VDB_RCCL(api.all_gather(comm->my_label.p, comm->all_label.p, per, ncclUint64, comm->comm, stream));
        VDB_RCCL(api.group_end());
        // 3. G ascending lists -> the k smallest (dist, label), on every rank
        std::string err;
        rc = launch_merge_topk((const float*)comm->all_dist.p, (const uint64_t*)comm->all_label.p, (uint32_t)G, (uint32_t)nq,
                               (uint32_t)k, d_out_dist, d_out_label, stream, &err);
        if (rc != 0) return set_error(rc, err);
        return (int)VDB_OK;
    });
}

// ---- shard merge ------------------------------------------------------------------------

int vdb_merge_topk_device(const float* d_dist, const uint64_t* d_label, size_t n_lists, size_t nq, size_t k,
                          float* d_out_dist, uint64_t* d_out_label, void* stream) {
    return guarded([&] {
        if (nq == 0 || k == 0) return (int)VDB_OK;
        if (!d_dist || !d_label || !d_out_dist || !d_out_label) throw std::invalid_argument("null buffer");
        if (n_lists == 0 || n_lists * k > 5461) throw std::invalid_argument("merge: n_lists * k must be in 1..5461");
        std::string err;
        const int rc = launch_merge_topk(d_dist, d_label, (uint32_t)n_lists, (uint32_t)nq, (uint32_t)k, d_out_dist,
                                         d_out_label, (hipStream_t)stream, &err);
        if (rc != 0) return set_error(rc, err);
        return (int)VDB_OK;
    });
}

}  // extern "C"
