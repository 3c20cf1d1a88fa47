// host_index.cpp — see host_index.hpp.  Reference line numbers are those of
// src/hnsw/hnsw_scratch/hnsw_core.hpp unless another file is named.
#include "host_index.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <set>
#include <stdexcept>
#include <thread>
#include <unordered_set>

namespace vdb {

namespace {
// total orders (DESIGN.md "Tie order"): results evict the largest (d, id); candidates pop
// the smallest (d, id).
inline bool result_less(const DistId& a, const DistId& b) { return a.d < b.d || (a.d == b.d && a.id < b.id); }
inline bool cand_less(const DistId& a, const DistId& b) { return a.d > b.d || (a.d == b.d && a.id > b.id); }

template <typename T>
void put(std::ostream& o, const T& v) {
    o.write(reinterpret_cast<const char*>(&v), sizeof(T));
}
template <typename T>
void get(std::istream& i, T& v) {
    i.read(reinterpret_cast<char*>(&v), sizeof(T));
}
}  // namespace

// libstdc++'s std::uniform_real_distribution<double>(0,1) over std::default_random_engine
// (:92, :145, :241), written out: two draws of minstd_rand0, low word first.
double HostIndex::LevelRng::uniform() {
    const double range = 2147483646.0;
    const double lo = double(next() - 1);
    const double hi = double(next() - 1);
    double r = (lo + hi * range) / (range * range);
    if (r >= 1.0) r = std::nextafter(1.0, 0.0);
    return r;
}

void HostIndex::init_layout() {  // :140-152, :176-178
    maxM = M;
    maxM0 = layer0_as_upper_ ? M : 2 * M;
    data_size = dim * sizeof(float);
    link0_stride = maxM0 * sizeof(uint32_t) + sizeof(uint32_t);
    stride = link0_stride + data_size + sizeof(uint64_t);
    link0_off = 0;
    data_off = link0_off + link0_stride;
    label_off = data_off + data_size;
    link_stride = maxM * sizeof(uint32_t) + sizeof(uint32_t);
    level_lambda = 1.0 / std::log(1.0 * (double)M);
}

void HostIndex::alloc_storage(size_t cap, bool with_rows) {
    level0 = with_rows ? (char*)std::malloc(std::max<size_t>(cap * stride, 1)) : nullptr;
    upper = (char**)std::calloc(std::max<size_t>(cap, 1), sizeof(char*));
    if ((with_rows && !level0) || !upper) {
        std::free(level0);
        std::free(upper);
        level0 = nullptr;
        upper = nullptr;
        throw std::runtime_error("Not enough memory! Requested " + std::to_string(cap * stride) + " bytes.");
    }
    levels.assign(cap, 0);
    node_locks_.reset(new SpinLock[std::max<size_t>(cap, 1)]);
    capacity = cap;
}

HostIndex::HostIndex(int metric_, size_t dim_, size_t cap, size_t M_, size_t efc, size_t seed, bool layer0_lists_as_upper) {
    layer0_as_upper_ = layer0_lists_as_upper;
    if (dim_ == 0) throw std::invalid_argument("dim must be > 0");
    if (M_ < 2) throw std::invalid_argument("M must be >= 2");
    metric = metric_;
    dim = dim_;
    M = M_ <= 10000 ? M_ : 10000;  // :133-139
    if (2 * M > 0xFFFF) throw std::invalid_argument("2*M must fit the 16-bit list count");
    init_layout();
    ef_construction = std::max(efc, M);  // :142
    ef_search = 10;                      // :143
    rng_.seed(seed);                     // :145
    alloc_storage(cap);
    refresh_dist();
}

void HostIndex::release() {  // clear(), :187-199
    if (upper) {
        const size_t n = count;
        for (size_t i = 0; i < n; ++i)
            if (levels[i] > 0) std::free(upper[i]);
    }
    std::free(level0);
    std::free(upper);
    level0 = nullptr;
    upper = nullptr;
    count = 0;
}

HostIndex::~HostIndex() { release(); }

int HostIndex::draw_level() {  // :240-244 as intended: floor(-ln(U) / ln(M))
    return (int)(-std::log(rng_.uniform()) * level_lambda);
}

// ------------------------------------------------------------------------------ building

// Construction-time beam search on one layer, :293-356.  Result left in sc.top (a max-heap
// under result_less).
void HostIndex::beam_search_layer(uint32_t ep, const float* q, int layer, BuildScratch& sc) {
    const uint32_t tag = sc.next_epoch();
    auto& top = sc.top;
    auto& cand = sc.cand;
    top.clear();
    cand.clear();
    const size_t efc = ef_construction;
    float bound;
    if (!is_deleted(ep)) {
        const float d = dist_(q, vec(ep), dim);
        top.push_back({d, ep});
        cand.push_back({d, ep});
        bound = d;
    } else {
        bound = std::numeric_limits<float>::max();
        cand.push_back({-bound, ep});
    }
    sc.seen[ep] = tag;

    while (!cand.empty()) {
        const DistId c = cand.front();
        if (c.d > bound && top.size() == efc) break;
        std::pop_heap(cand.begin(), cand.end(), cand_less);
        cand.pop_back();

        SpinLock& lk = node_locks_[c.id];
        lk.lock();
        const uint32_t* list = list_at(c.id, layer);
        const unsigned n = list_count(list);
        const uint32_t* ids = list + 1;
        for (unsigned j = 0; j < n; ++j) __builtin_prefetch(vec(ids[j]), 0, 1);
        for (unsigned j = 0; j < n; ++j) {
            const uint32_t nb = ids[j];
            if (sc.seen[nb] == tag) continue;
            sc.seen[nb] = tag;
            const float d = dist_(q, vec(nb), dim);
            if (top.size() < efc || bound > d) {
                cand.push_back({d, nb});
                std::push_heap(cand.begin(), cand.end(), cand_less);
                if (!is_deleted(nb)) {
                    top.push_back({d, nb});
                    std::push_heap(top.begin(), top.end(), result_less);
                }
                if (top.size() > efc) {
                    std::pop_heap(top.begin(), top.end(), result_less);
                    top.pop_back();
                }
                if (!top.empty()) bound = top.front().d;
            }
        }
        lk.unlock();
    }
}

// getNeighborsByHeuristic2, :357-390.  `pool` in: candidates (any order); out: survivors.
void HostIndex::prune_heuristic(std::vector<DistId>& pool, size_t m, BuildScratch& sc) {
    if (pool.size() < m) return;
    // closest first; equal distances: larger id first (the reference's default pair order
    // on (-dist, id), :363-366)
    std::sort(pool.begin(), pool.end(),
              [](const DistId& a, const DistId& b) { return a.d < b.d || (a.d == b.d && a.id > b.id); });
    auto& keep = sc.keep;
    keep.clear();
    for (const DistId& c : pool) {
        if (keep.size() >= m) break;
        bool good = true;
        for (const DistId& k : keep) {
            if (dist_(vec(k.id), vec(c.id), dim) < c.d) {
                good = false;
                break;
            }
        }
        if (good) keep.push_back(c);
    }
    pool.assign(keep.begin(), keep.end());
}

// mutuallyConnectNewElement, :392-498; `update` = its updateFlag (an insert holds the element's own lock
// already, an update takes it for the rewrite of the element's list, :411-414).  Consumes sc.top.
uint32_t HostIndex::connect(uint32_t id, int level, BuildScratch& sc, bool update) {
    const size_t max_m = level ? maxM : maxM0;
    auto& sel = sc.work;
    sel.assign(sc.top.begin(), sc.top.end());
    prune_heuristic(sel, M, sc);
    if (sel.size() > M) throw std::runtime_error("Should not be more than M_ candidates returned by the heuristic");
    // the reference drains a max-heap: farthest first, closest last (:404-409)
    std::sort(sel.begin(), sel.end(), [](const DistId& a, const DistId& b) { return result_less(b, a); });
    if (sel.empty()) throw std::runtime_error("no neighbour candidates for a new element");
    const uint32_t next_closest = sel.back().id;
    {
        SpinLock& own = node_locks_[id];
        if (update) own.lock();
        struct Unlock {
            SpinLock* l;
            ~Unlock() {
                if (l) l->unlock();
            }
        } unlock_own{update ? &own : nullptr};
        uint32_t* mine = list_at(id, level);
        if (*mine && !update) throw std::runtime_error("The newly inserted element should have a blank neighbor list");
        set_list_count(mine, (unsigned)sel.size());
        uint32_t* data = mine + 1;
        for (size_t i = 0; i < sel.size(); ++i) {
            if (data[i] && !update) throw std::runtime_error("Possible memory corruption");
            if (level > levels[sel[i].id]) throw std::runtime_error("Trying to make a link on a non-existent level");
            data[i] = sel[i].id;
        }
    }
    std::vector<DistId> pool;
    // sel is sc.work; prune_heuristic below reuses sc.keep only, so sel stays intact
    for (size_t i = 0; i < sel.size(); ++i) {
        const uint32_t other_id = sel[i].id;
        SpinLock& lk = node_locks_[other_id];
        lk.lock();
        uint32_t* other = list_at(other_id, level);
        const size_t n_other = list_count(other);
        if (n_other > max_m || other_id == id || level > levels[other_id]) {
            lk.unlock();
            if (n_other > max_m) throw std::runtime_error("Bad value of sizeof_link_other");
            if (other_id == id) throw std::runtime_error("Trying to connect an element to itself");
            throw std::runtime_error("Trying to make a link on a non-existent level");
        }
        uint32_t* data = other + 1;
        bool present = false;  // :455-463: an updated element may be linked from there already
        if (update)
            for (size_t j = 0; j < n_other && !present; ++j) present = data[j] == id;
        if (present) {
        } else if (n_other < max_m) {
            data[n_other] = id;
            set_list_count(other, (unsigned)(n_other + 1));
        } else {
            pool.clear();
            pool.push_back({dist_(vec(id), vec(other_id), dim), id});
            for (size_t j = 0; j < n_other; ++j) pool.push_back({dist_(vec(data[j]), vec(other_id), dim), data[j]});
            prune_heuristic(pool, max_m, sc);
            std::sort(pool.begin(), pool.end(), [](const DistId& a, const DistId& b) { return result_less(b, a); });
            for (size_t j = 0; j < pool.size(); ++j) data[j] = pool[j].id;
            set_list_count(other, (unsigned)pool.size());
        }
        lk.unlock();
    }
    return next_closest;
}

// addPoint(data, label, level) after the id has been assigned, :1027-1103.
void HostIndex::insert(uint32_t id, const float* v, int level, BuildScratch& sc) {
    sc.ensure(capacity);
    SpinLock& self = node_locks_[id];
    self.lock();
    struct Unlock {
        SpinLock& l;
        ~Unlock() { l.unlock(); }
    } unlock_self{self};

    const int cur_level = level;
    levels[id] = cur_level;

    std::unique_lock<std::mutex> top_guard(global_lock_);
    const int top_level = max_level;
    if (cur_level <= top_level) top_guard.unlock();
    uint32_t cur = entry;
    const uint32_t entry_copy = entry;

    if (cur_level) {
        upper[id] = (char*)std::calloc(link_stride * (size_t)cur_level + 1, 1);
        if (!upper[id]) throw std::runtime_error("Not enough memory: addPoint failed to allocate linklist");
    }

    if (cur != 0xFFFFFFFFu) {
        if (cur_level < top_level) {
            float cur_d = dist_(v, vec(cur), dim);
            for (int lv = top_level; lv > cur_level; --lv) {
                bool moved = true;
                while (moved) {
                    moved = false;
                    SpinLock& lk = node_locks_[cur];
                    lk.lock();
                    const uint32_t* list = list_at(cur, lv);
                    const unsigned n = list_count(list);
                    const uint32_t* ids = list + 1;
                    uint32_t best = cur;
                    for (unsigned i = 0; i < n; ++i) {
                        const uint32_t c = ids[i];
                        if (c > capacity) {
                            lk.unlock();
                            throw std::runtime_error("candidate error!");
                        }
                        const float d = dist_(v, vec(c), dim);
                        if (d < cur_d) {
                            cur_d = d;
                            best = c;
                            moved = true;
                        }
                    }
                    lk.unlock();
                    cur = best;
                }
            }
        }
        const bool entry_deleted = is_deleted(entry_copy);
        for (int lv = std::min(cur_level, top_level); lv >= 0; --lv) {
            beam_search_layer(cur, v, lv, sc);
            if (entry_deleted) {  // :1085-1089
                sc.top.push_back({dist_(v, vec(entry_copy), dim), entry_copy});
                std::push_heap(sc.top.begin(), sc.top.end(), result_less);
                if (sc.top.size() > ef_construction) {
                    std::pop_heap(sc.top.begin(), sc.top.end(), result_less);
                    sc.top.pop_back();
                }
            }
            cur = connect(id, lv, sc);
        }
    } else {
        entry = id;
        max_level = cur_level;
    }
    if (cur_level > top_level) {
        entry = id;
        max_level = cur_level;
    }
}

void HostIndex::add(const float* v, uint64_t label, int level) {  // addPoint(data, label, level), :994-1104
    require_rows("addPoint");
    uint32_t existing = 0xFFFFFFFFu;
    {
        std::unique_lock<std::mutex> g(label_lock_);
        auto it = label_map_.find(label);
        if (it != label_map_.end()) existing = it->second;
    }
    if (existing != 0xFFFFFFFFu) {  // :1001-1016: the element is updated instead of created
        if (reuse_deleted_ && is_deleted(existing))
            throw std::runtime_error("Can't use addPoint to update deleted elements if replacement of deleted elements is enabled.");
        if (is_deleted(existing)) set_deleted_flag(existing, false);
        BuildScratch sc;
        update_point(existing, v, sc);
        ++version;
        return;
    }
    uint64_t l = label;
    add_batch(v, &l, 1, 1, level);
}

void HostIndex::set_reuse_deleted(bool on) {
    require_rows("reuse_deleted");
    std::unique_lock<std::mutex> g(deleted_lock_);
    reuse_deleted_ = on;
    deleted_elements_.clear();
    if (on) {  // as the load constructor does, :672-677
        const size_t n = count;
        for (size_t i = 0; i < n; ++i)
            if (is_deleted((uint32_t)i)) deleted_elements_.insert((uint32_t)i);
    }
}

void HostIndex::set_deleted_flag(uint32_t id, bool del) {  // markDeletedInternal / unmarkDeletedInternal, :728-782
    unsigned char* flag = ((unsigned char*)list0(id)) + 2;
    if (del) {
        if (*flag & 0x01) throw std::runtime_error("The requested to delete element is already deleted");
        *flag |= 0x01;
        deleted += 1;
    } else {
        if (!(*flag & 0x01)) throw std::runtime_error("The requested to undelete element is not deleted");
        *flag &= (unsigned char)~0x01;
        deleted -= 1;
    }
    if (reuse_deleted_) {
        std::unique_lock<std::mutex> g(deleted_lock_);
        if (del) deleted_elements_.insert(id);
        else deleted_elements_.erase(id);
    }
}

void HostIndex::add_point(const float* v, uint64_t label, bool replace_deleted) {  // :804-842
    require_rows("addPoint");
    if (!reuse_deleted_ && replace_deleted) throw std::runtime_error("Replacement of deleted elements is disabled in constructor");
    if (!replace_deleted) return add(v, label, -1);
    uint32_t vacant = 0xFFFFFFFFu;
    {
        std::unique_lock<std::mutex> g(deleted_lock_);
        if (!deleted_elements_.empty()) {
            vacant = *deleted_elements_.begin();
            deleted_elements_.erase(deleted_elements_.begin());
        }
    }
    if (vacant == 0xFFFFFFFFu) return add(v, label, -1);
    const uint64_t old_label = label_of(vacant);
    std::memcpy(row(vacant) + label_off, &label, 8);  // setExternalLabel
    {
        std::unique_lock<std::mutex> g(label_lock_);
        label_map_.erase(old_label);
        label_map_[label] = vacant;
    }
    set_deleted_flag(vacant, false);
    BuildScratch sc;
    update_point(vacant, v, sc);
    ++version;
}

// updatePoint(dataPoint, internalId, updateNeighborProbability = 1.0), :844-913.  The reference walks two
// unordered_sets; here the candidates are met in ascending id order, which together with the (dist, id) heap order
// decides which of several candidates at exactly the same distance survive `distance < top` (:888) - the order the
// neighbours are taken in does not matter, each list is rebuilt from distances alone.  (update_rng_, :865: with
// probability 1.0 no draw can skip a neighbour, so none is made.)
void HostIndex::update_point(uint32_t id, const float* v, BuildScratch& sc) {
    sc.ensure(capacity);
    std::memcpy(row(id) + data_off, v, data_size);
    const int max_level_copy = max_level;
    const uint32_t entry_copy = entry;
    if (entry_copy == id && count == 1) return;
    const int elem_level = levels[id];
    auto connections = [&](uint32_t node, int layer) {  // getConnectionsWithLock, :983-991
        SpinLock& lk = node_locks_[node];
        lk.lock();
        const uint32_t* list = list_at(node, layer);
        std::vector<uint32_t> out(list + 1, list + 1 + list_count(list));
        lk.unlock();
        return out;
    };
    std::vector<DistId> pool;
    for (int layer = 0; layer <= elem_level; ++layer) {
        const std::vector<uint32_t> one_hop = connections(id, layer);
        if (one_hop.empty()) continue;
        std::set<uint32_t> s_cand, s_neigh;
        s_cand.insert(id);
        for (uint32_t a : one_hop) {
            s_cand.insert(a);
            s_neigh.insert(a);
            for (uint32_t b : connections(a, layer)) s_cand.insert(b);
        }
        for (uint32_t neigh : s_neigh) {
            const size_t size = s_cand.count(neigh) ? s_cand.size() - 1 : s_cand.size();
            const size_t keep = std::min(ef_construction, size);
            pool.clear();  // max-heap under result_less: the `keep` closest candidates of `neigh`
            for (uint32_t c : s_cand) {
                if (c == neigh) continue;
                const float d = dist_(vec(neigh), vec(c), dim);
                if (pool.size() < keep) {
                    pool.push_back({d, c});
                    std::push_heap(pool.begin(), pool.end(), result_less);
                } else if (keep != 0 && d < pool.front().d) {
                    std::pop_heap(pool.begin(), pool.end(), result_less);
                    pool.back() = {d, c};
                    std::push_heap(pool.begin(), pool.end(), result_less);
                }
            }
            prune_heuristic(pool, layer == 0 ? maxM0 : maxM, sc);
            std::sort(pool.begin(), pool.end(), [](const DistId& a, const DistId& b) { return result_less(b, a); });
            SpinLock& lk = node_locks_[neigh];
            lk.lock();
            uint32_t* list = list_at(neigh, layer);
            set_list_count(list, (unsigned)pool.size());  // (entries behind the new count keep their old bytes, as :902-907)
            for (size_t i = 0; i < pool.size(); ++i) list[1 + i] = pool[i].id;
            lk.unlock();
        }
    }
    repair_connections(id, v, entry_copy, elem_level, max_level_copy, sc);
}

// repairConnectionsForUpdate, :924-980
void HostIndex::repair_connections(uint32_t id, const float* v, uint32_t entry_copy, int elem_level, int max_level_copy,
                                   BuildScratch& sc) {
    uint32_t cur = entry_copy;
    if (elem_level < max_level_copy) {
        float cur_d = dist_(v, vec(cur), dim);
        for (int lv = max_level_copy; lv > elem_level; --lv) {
            bool moved = true;
            while (moved) {
                moved = false;
                SpinLock& lk = node_locks_[cur];
                lk.lock();
                const uint32_t* list = list_at(cur, lv);
                const unsigned n = list_count(list);
                uint32_t best = cur;
                for (unsigned i = 0; i < n; ++i) {
                    const float d = dist_(v, vec(list[1 + i]), dim);
                    if (d < cur_d) {
                        cur_d = d;
                        best = list[1 + i];
                        moved = true;
                    }
                }
                lk.unlock();
                cur = best;
            }
        }
    }
    if (elem_level > max_level_copy) throw std::runtime_error("Level of item to be updated cannot be bigger than max level");
    for (int lv = elem_level; lv >= 0; --lv) {
        beam_search_layer(cur, v, lv, sc);
        auto& top = sc.top;  // drop the element itself (:962-967)
        top.erase(std::remove_if(top.begin(), top.end(), [&](const DistId& e) { return e.id == id; }), top.end());
        std::make_heap(top.begin(), top.end(), result_less);
        if (top.empty()) continue;
        if (is_deleted(entry_copy)) {  // :970-975
            top.push_back({dist_(v, vec(entry_copy), dim), entry_copy});
            std::push_heap(top.begin(), top.end(), result_less);
            if (top.size() > ef_construction) {
                std::pop_heap(top.begin(), top.end(), result_less);
                top.pop_back();
            }
        }
        cur = connect(id, lv, sc, /*update=*/true);
    }
}

// Ids, levels and rows of n new elements, assigned in input order exactly as n sequential addPoint
// calls would (:1018-1043); nothing is linked yet.  Returns the first new id.
size_t HostIndex::append_unlinked(const float* vecs, const uint64_t* labels, size_t n, int forced_level, const int* given_levels) {
    require_rows("addPoint");
    const size_t base = count;
    if (base + n > capacity) throw std::runtime_error("The number of elements exceeds the specified limit");  // :1018-1020
    {
        std::unique_lock<std::mutex> g(label_lock_);
        std::unordered_set<uint64_t> fresh;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t l = labels ? labels[i] : (uint64_t)(base + i);
            if (label_map_.count(l) || !fresh.insert(l).second)
                throw std::runtime_error("label already present in a bulk insert (add the point alone to update it: hnsw_core.hpp:1001-1016)");
        }
        for (size_t i = 0; i < n; ++i) label_map_[labels ? labels[i] : (uint64_t)(base + i)] = (uint32_t)(base + i);
    }
    for (size_t i = 0; i < n; ++i) {
        int lv = draw_level();  // drawn even when overridden, as :1028-1029
        if (forced_level > 0) lv = forced_level;
        if (given_levels) lv = given_levels[i];
        const uint32_t id = (uint32_t)(base + i);
        char* r = row(id);
        std::memset(r, 0, stride);  // :1039
        const uint64_t l = labels ? labels[i] : (uint64_t)(base + i);
        std::memcpy(r + label_off, &l, 8);
        std::memcpy(r + data_off, vecs + i * dim, data_size);
        levels[id] = lv;
    }
    count = base + n;
    ++version;
    return base;
}

void HostIndex::alloc_upper_lists(uint32_t id) {
    if (levels[id] <= 0) return;
    if (upper[id]) throw std::runtime_error("element already has upper-layer lists");
    upper[id] = (char*)std::calloc(link_stride * (size_t)levels[id] + 1, 1);  // :1046-1049
    if (!upper[id]) throw std::runtime_error("Not enough memory: addPoint failed to allocate linklist");
}

// addPoint's linking (:1044-1103) for elements that append_unlinked stored, in the order given.
void HostIndex::link_elements(const uint32_t* ids, size_t n, int n_threads) {
    require_rows("addPoint");
    if (n == 0) return;
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::string err;
    std::mutex err_lock;
    auto worker = [&]() {
        BuildScratch sc;
        try {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n || failed.load()) break;
                insert(ids[i], vec(ids[i]), levels[ids[i]], sc);
            }
        } catch (const std::exception& e) {
            std::unique_lock<std::mutex> g(err_lock);
            err = e.what();
            failed = true;
        }
    };
    if (n_threads <= 1) {
        worker();
    } else {
        // the very first element of an empty index goes in alone (it only sets the entry)
        if (entry == 0xFFFFFFFFu) {
            BuildScratch sc;
            insert(ids[0], vec(ids[0]), levels[ids[0]], sc);
            next = 1;
        }
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
    }
    ++version;
    if (failed) throw std::runtime_error(err);
}

void HostIndex::truncate_to_empty() {
    require_rows("addPoint");
    const size_t n = count;
    for (size_t i = 0; i < n; ++i) {
        if (upper[i]) std::free(upper[i]);
        upper[i] = nullptr;
        levels[i] = 0;
    }
    {
        std::unique_lock<std::mutex> g(label_lock_);
        label_map_.clear();
    }
    count = 0;
    deleted = 0;
    {
        std::unique_lock<std::mutex> g(deleted_lock_);
        deleted_elements_.clear();
    }
    entry = 0xFFFFFFFFu;
    max_level = -1;
    ++version;
}

void HostIndex::add_batch(const float* vecs, const uint64_t* labels, size_t n, int n_threads, int forced_level) {
    if (n == 0) return;
    if (labels && n > 1) {
        // a label that exists (or comes twice) means an update (:1001-1016): such a batch is n sequential addPoint calls
        bool updates = false;
        {
            std::unique_lock<std::mutex> g(label_lock_);
            std::unordered_set<uint64_t> fresh;
            for (size_t i = 0; i < n && !updates; ++i) updates = label_map_.count(labels[i]) != 0 || !fresh.insert(labels[i]).second;
        }
        if (updates) {
            for (size_t i = 0; i < n; ++i) add(vecs + i * dim, labels[i], forced_level);
            return;
        }
    }
    const size_t base = append_unlinked(vecs, labels, n, forced_level);
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i] = (uint32_t)(base + i);
    link_elements(ids.data(), n, n_threads);
}

// ------------------------------------------------------------------------------ persistence

void HostIndex::save(const std::string& path) const {  // :551-579
    require_rows("saveIndex");
    std::ofstream out(path, std::ios::binary);
    if (!out.is_open()) throw std::runtime_error("Cannot open file");
    const size_t n = count;
    put(out, link0_off);
    put(out, capacity);
    put(out, n);
    put(out, stride);
    put(out, label_off);
    put(out, data_off);
    put(out, max_level);
    put(out, entry);
    put(out, maxM);
    put(out, maxM0);
    put(out, M);
    put(out, ef_search);
    put(out, ef_construction);
    out.write(level0, (std::streamsize)(n * stride));
    for (size_t i = 0; i < n; ++i) {
        const unsigned bytes = levels[i] > 0 ? (unsigned)(link_stride * (size_t)levels[i]) : 0u;
        put(out, bytes);
        if (bytes) out.write(upper[i], bytes);
    }
    if (!out.good()) throw std::runtime_error("write failed");
}

HostIndex::HostIndex(const std::string& path, int metric_, size_t dim_, size_t capacity_hint, bool stream_rows) {  // :581-682
    metric = metric_;
    dim = dim_;
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) throw std::runtime_error("Cannot open file");
    in.seekg(0, in.end);
    const std::streamoff total = in.tellg();
    in.seekg(0, in.beg);
    size_t n = 0, file_cap = 0;
    get(in, link0_off);
    get(in, file_cap);
    get(in, n);
    get(in, stride);
    get(in, label_off);
    get(in, data_off);
    get(in, max_level);
    get(in, entry);
    get(in, maxM);
    get(in, maxM0);
    get(in, M);
    get(in, ef_search);
    get(in, ef_construction);
    if (!in.good() || M < 1 || maxM0 > 0xFFFF) throw std::runtime_error("Index seems to be corrupted or unsupported");
    data_size = dim * sizeof(float);
    link_stride = maxM * sizeof(uint32_t) + sizeof(uint32_t);
    link0_stride = maxM0 * sizeof(uint32_t) + sizeof(uint32_t);
    if (stride != link0_stride + data_size + sizeof(uint64_t) || data_off != link0_off + link0_stride ||
        label_off != data_off + data_size)
        throw std::runtime_error("Index layout does not match the requested space (dim/metric)");
    // walk the file once to validate its length (:617-633)
    std::streamoff pos = in.tellg();
    std::streamoff cursor = pos + (std::streamoff)(n * stride);
    for (size_t i = 0; i < n; ++i) {
        if (cursor < 0 || cursor >= total) throw std::runtime_error("Index seems to be corrupted or unsupported");
        in.seekg(cursor, in.beg);
        unsigned bytes = 0;
        get(in, bytes);
        cursor += 4 + (std::streamoff)bytes;
    }
    if (cursor != total) throw std::runtime_error("Index seems to be corrupted or unsupported");
    in.clear();
    in.seekg(pos, in.beg);

    size_t cap = capacity_hint;
    if (cap < n) cap = file_cap;
    if (cap < n) cap = n;
    if (stream_rows) {  // the level-0 block stays in the file: the uploader reads it (capi.cpp upload)
        rows_resident = false;
        rows_path = path;
        rows_offset = (long long)pos;
        alloc_storage(n, false);
        in.seekg(pos + (std::streamoff)(n * stride), in.beg);
    } else {
        alloc_storage(cap);
        in.read(level0, (std::streamsize)(n * stride));
    }
    level_lambda = 1.0 / std::log(1.0 * (double)M);  // the reference forgets this (:654)
    ef_search = 10;                                  // :655
    rng_.seed(100);
    count = n;
    for (size_t i = 0; i < n; ++i) {
        if (rows_resident) label_map_[label_of((uint32_t)i)] = (uint32_t)i;
        unsigned bytes = 0;
        get(in, bytes);
        if (bytes == 0) {
            levels[i] = 0;
            upper[i] = nullptr;
        } else {
            levels[i] = (int)(bytes / link_stride);  // :664
            upper[i] = (char*)std::malloc(bytes);
            if (!upper[i]) throw std::runtime_error("Not enough memory: loadIndex failed to allocate linklist");
            in.read(upper[i], bytes);
        }
    }
    size_t del = 0;
    if (rows_resident)
        for (size_t i = 0; i < n; ++i) del += is_deleted((uint32_t)i);
    deleted = del;  // (streamed rows: counted by the uploader)
    // searchKnn starts at entry_id_ on max_level_ (:1111-1114) without looking: a file that names an entry
    // point outside the index, or one without that many levels, is corrupt (link targets are validated per
    // level when the graph is repacked for the device, capi.cpp upload)
    if (n != 0 && (entry >= n || max_level < 0 || levels[entry] != max_level)) {
        release();
        throw std::runtime_error("Index seems to be corrupted or unsupported");
    }
    refresh_dist();
}

// ------------------------------------------------------------------------------ maintenance

void HostIndex::mark_delete(uint64_t label, bool undo) {  // :709-782
    require_rows("markDelete");
    uint32_t id;
    {
        std::unique_lock<std::mutex> g(label_lock_);
        auto it = label_map_.find(label);
        if (it == label_map_.end()) throw std::runtime_error("Label not found");
        id = it->second;
    }
    set_deleted_flag(id, !undo);
    ++version;
}

void HostIndex::get_data_by_label(uint64_t label, float* out) {  // :683-703
    require_rows("getDataByLabel");
    std::unique_lock<std::mutex> g(label_lock_);
    auto it = label_map_.find(label);
    if (it == label_map_.end() || is_deleted(it->second)) throw std::runtime_error("Label not found");
    std::memcpy(out, vec(it->second), data_size);
}

void HostIndex::resize(size_t new_cap) {  // :500-522
    require_rows("resizeIndex");
    if (new_cap < count) throw std::runtime_error("Cannot resize, max element is less than the current number of elements");
    char* l0 = (char*)std::realloc(level0, std::max<size_t>(new_cap * stride, 1));
    if (!l0) throw std::runtime_error("Not enough memory: resizeIndex failed to allocate base layer");
    level0 = l0;
    char** up = (char**)std::realloc(upper, std::max<size_t>(new_cap, 1) * sizeof(char*));
    if (!up) throw std::runtime_error("Not enough memory: resizeIndex failed to allocate other layers");
    upper = up;
    for (size_t i = capacity; i < new_cap; ++i) upper[i] = nullptr;
    levels.resize(new_cap);
    node_locks_.reset(new SpinLock[std::max<size_t>(new_cap, 1)]);
    capacity = new_cap;
    ++version;
}

int HostIndex::check_integrity(int* min_in, int* max_in) const {  // :1216-1245
    require_rows("checkIntegrity");
    const size_t n = count;
    std::vector<int> inbound(n, 0);
    std::vector<uint32_t> sorted;
    for (size_t i = 0; i < n; ++i) {
        for (int l = 0; l <= levels[i]; ++l) {
            const uint32_t* list = list_at((uint32_t)i, l);
            const unsigned size = list_count(list);
            sorted.assign(list + 1, list + 1 + size);
            for (uint32_t v : sorted) {
                if (!(v < n)) return 1;
                if (v == i) return 2;
                inbound[v]++;
            }
            std::sort(sorted.begin(), sorted.end());
            if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return 3;
        }
    }
    if (n > 1) {
        int lo = inbound[0], hi = inbound[0];
        for (size_t i = 0; i < n; ++i) {
            if (inbound[i] <= 0) return 4;
            lo = std::min(lo, inbound[i]);
            hi = std::max(hi, inbound[i]);
        }
        if (min_in) *min_in = lo;
        if (max_in) *max_in = hi;
    }
    return 0;
}

// ------------------------------------------------------------------------------ brute force

HostBruteforce::HostBruteforce(int metric_, size_t dim_, size_t cap) {  // bruteforce.hpp:54-65
    if (dim_ == 0) throw std::invalid_argument("dim must be > 0");
    metric = metric_;
    dim = dim_;
    capacity = cap;
    data_size = dim * sizeof(float);
    stride = data_size + sizeof(uint64_t);
    data = (char*)std::malloc(std::max<size_t>(cap * stride, 1));
    if (!data) throw std::runtime_error("Not enough memory: BruteforceSearch failed to allocate data");
}

HostBruteforce::HostBruteforce(const std::string& path, int metric_, size_t dim_) {  // bruteforce.hpp:208-231
    metric = metric_;
    dim = dim_;
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) throw std::runtime_error("Cannot open file");
    size_t file_stride = 0;
    get(in, capacity);
    get(in, file_stride);
    get(in, count);
    data_size = dim * sizeof(float);
    stride = data_size + sizeof(uint64_t);
    if (!in.good() || file_stride != stride || count > capacity)
        throw std::runtime_error("Index layout does not match the requested space (dim)");
    data = (char*)std::malloc(std::max<size_t>(capacity * stride, 1));
    if (!data) throw std::runtime_error("Not enough memory: loadIndex failed to allocate data");
    in.read(data, (std::streamsize)(capacity * stride));
    for (size_t i = 0; i < count; ++i) {
        uint64_t l;
        std::memcpy(&l, data + i * stride + data_size, 8);
        label_to_index[l] = i;
    }
}

HostBruteforce::~HostBruteforce() { std::free(data); }

void HostBruteforce::add(const float* v, uint64_t label) {  // bruteforce.hpp:84-108
    size_t idx;
    {
        std::unique_lock<std::mutex> g(lock);
        auto it = label_to_index.find(label);
        if (it != label_to_index.end()) {
            idx = it->second;
        } else {
            if (count >= capacity) throw std::runtime_error("The number of elements exceeds the specified limit\n");
            idx = count;
            label_to_index[label] = idx;
            count++;
        }
        ++version;
    }
    std::memcpy(data + stride * idx + data_size, &label, 8);
    std::memcpy(data + stride * idx, v, data_size);
}

void HostBruteforce::remove(uint64_t label) {  // bruteforce.hpp:119-138
    std::unique_lock<std::mutex> g(lock);
    auto it = label_to_index.find(label);
    if (it == label_to_index.end()) return;
    const size_t slot = it->second;
    label_to_index.erase(it);
    const size_t last = count - 1;
    if (slot != last) {
        uint64_t moved;
        std::memcpy(&moved, data + stride * last + data_size, 8);
        label_to_index[moved] = slot;
        std::memcpy(data + stride * slot, data + stride * last, stride);
    }
    count--;
    ++version;
}

void HostBruteforce::save(const std::string& path) const {  // bruteforce.hpp:190-202
    std::ofstream out(path, std::ios::binary);
    if (!out.is_open()) throw std::runtime_error("Cannot open file");
    put(out, capacity);
    put(out, stride);
    put(out, count);
    out.write(data, (std::streamsize)(capacity * stride));
    if (!out.good()) throw std::runtime_error("write failed");
}

}  // namespace vdb
