// ref_fragments.cpp — C entry points over the parts of the REFERENCE that compile untouched.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ when
// /root/reference is present; nothing of the reference is copied into this repository:
// the headers are included from where they lie, and the two distance functors are cut
// out of their (otherwise non-compiling) files into a temporary directory at build
// time (space_l2.hpp lines 1-14, space_ip.hpp lines 3-18) and removed afterwards.
//
// The reference's index classes themselves (hnsw_core.hpp, bruteforce.hpp) do not
// compile and are NOT built here; see DESIGN.md.
#include <cstddef>
#include <cstring>

#include "hnswlib.hpp"            // reference: AlgorithmInterface, SearchKNNCloserFirst
#include "visited_list_pool.hpp"  // reference: VisitedList, VisitedListPool
#include "l2_fragment.inc"        // reference: template <typename Scalar> struct L2Sqr
#include "ip_fragment.inc"        // reference: InnerProduct, InnerProductDistance

namespace {
// A stand-in index whose SearchKNN hands back a prepared heap, to drive the reference's
// own SearchKNNCloserFirst (hnswlib.hpp:87-99).
struct CannedIndex : AlgorithmInterface<float> {
    std::priority_queue<std::pair<float, size_t>> canned;
    void addPoint(const void*, size_t, bool) override {}
    std::priority_queue<std::pair<float, size_t>> SearchKNN(const void*, size_t, BaseFilterFunctor*) const override {
        return canned;
    }
    void saveIndex(const std::string&) override {}
};
}  // namespace

extern "C" {

float ref_l2sqr(const float* a, const float* b, size_t dim) { return L2Sqr<float>{}(a, b, &dim); }

float ref_ipdist(const float* a, const float* b, size_t dim) { return InnerProductDistance<float>{}(a, b, dim); }

void ref_visited_tags(int n_elements, int n_rounds, unsigned short* tags_out, int* was_cleared_out) {
    VisitedListPool pool(1, n_elements);
    for (int r = 0; r < n_rounds; r++) {
        VisitedList* vl = pool.getFreeVisitedList();
        tags_out[r] = vl->currentVisited;
        int dirty = 0;
        for (int i = 0; i < n_elements; i++) dirty |= (vl->visitedAt[i] == vl->currentVisited);
        was_cleared_out[r] = !dirty;
        vl->visitedAt[r % n_elements] = vl->currentVisited;
        pool.releaseVisitedList(vl);
    }
}

// Push n (dist, label) pairs into the reference's result heap type and drain it with the
// reference's SearchKNNCloserFirst.
void ref_closer_first(const float* dist, const size_t* label, size_t n, float* out_dist, size_t* out_label) {
    CannedIndex idx;
    for (size_t i = 0; i < n; i++) idx.canned.emplace(dist[i], label[i]);
    auto v = idx.SearchKNNCloserFirst(nullptr, n, nullptr);
    for (size_t i = 0; i < v.size(); i++) {
        out_dist[i] = v[i].first;
        out_label[i] = v[i].second;
    }
}

}  // extern "C"
