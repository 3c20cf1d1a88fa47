// A caller written in the REFERENCE's own spellings (hnswlib.hpp:53-99, space_l2.hpp:16-37, space_ip.hpp:20-41,
// hnsw_core.hpp:20-21, :110-123, bruteforce.hpp:13-14), compiled against vdb_hnsw.hpp: class templates over dist_t and
// <Scalar, DistanceFunction>, get_distance_function_(), addPoint on an existing label (update), replace_deleted,
// getDataByLabel, SearchKNNCloserFirst - and searchKnn from several threads at once (const, concurrency-safe).
//   argv[1] = "host" | "gpu", argv[2] = where the L2 index is saved, argv[3] = where the inner-product one is.
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>

#include "../vectordb_amd/cpp/vdb_hnsw.hpp"

using namespace vdb_amd;

template <typename Space>
static int fill(AlgorithmInterface<float>* alg, const Space& space, const std::vector<float>& x, size_t n, size_t dim) {
    if (space.get_dim() != dim || space.get_data_size() != dim * sizeof(float)) return 10;
    for (size_t i = 0; i < n; ++i) alg->addPoint(x.data() + i * dim, 7000 + i);
    return 0;
}

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && !std::strcmp(argv[1], "gpu");
    const std::string path_l2 = argc > 2 ? argv[2] : "/tmp/vdb_ref_style_l2.bin";
    const std::string path_ip = argc > 3 ? argv[3] : "/tmp/vdb_ref_style_ip.bin";
    const size_t n = 600, dim = 20;
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::vector<float> x(n * dim), y(n * dim);
    for (auto& v : x) v = u(rng);
    for (auto& v : y) v = u(rng);

    // the host functors a space hands out (space_l2.hpp:1-14: dimension BY POINTER; space_ip.hpp:13-18: by value)
    L2Space<float, L2Sqr<float>> space(dim);
    InnerProductSpace<float, InnerProductDistance<float>> ip_space(dim);
    {
        const float a[5] = {1, 2, 3, 4, 0.1f}, b[5] = {0, 0, 0, 0, 0};
        const size_t five = 5;
        if (L2Space<float, L2Sqr<float>>(5).get_distance_function_()(a, b, &five) != 30.0100002f) return 11;
        const float c[3] = {1, 2, 3}, d[3] = {.5f, .25f, .125f};
        if (ip_space.get_distance_function_()(c, d, 3) != -0.375f) return 12;
    }

    HierarchicalNSW<float> idx(&space, n, 16, 200, 100, /*reuse_deleted=*/true);
    if (int rc = fill(&idx, space, x, n, dim)) return rc;
    // addPoint on a label the index holds: updated in place (hnsw_core.hpp:1001-1016)
    for (size_t i = 0; i < n; i += 9) idx.addPoint(y.data() + i * dim, 7000 + i);
    if (idx.getElementCount() != n) return 13;
    if (idx.getDataByLabel<float>(7009) != std::vector<float>(y.begin() + 9 * dim, y.begin() + 10 * dim)) return 14;
    // markDelete + addPoint(..., replace_deleted = true): the vacant place is reused (:815-841)
    for (size_t i = 2; i < n; i += 50) idx.markDelete(7000 + i);
    for (size_t j = 0; j < 5; ++j) idx.addPoint(y.data() + (n - 1 - j) * dim, 9000 + j, true);
    if (idx.getElementCount() != n || idx.getDeletedCount() != 12 - 5) return 15;
    idx.saveIndex(path_l2);

    HierarchicalNSW<float> ip(&ip_space, n, 8, 100);
    if (int rc = fill(&ip, ip_space, x, n, dim)) return rc;
    bool threw = false;
    try {
        ip.addPoint(x.data(), 1, true);  // hnsw_core.hpp:805-807
    } catch (const std::runtime_error& e) {
        threw = std::strstr(e.what(), "disabled in constructor") != nullptr;
    }
    if (!threw) return 16;
    ip.saveIndex(path_ip);

    BruteforceSearch<float> bf(&space, n);
    for (size_t i = 0; i < n; ++i) bf.addPoint(x.data() + i * dim, i);

    if (!gpu) {
        try {
            (void)idx.searchKnn(x.data(), 5);
            return 17;  // no CPU fallback
        } catch (const std::runtime_error&) {
        }
        std::puts("host ok");
        return 0;
    }
    idx.setefSearch(64);
    ip.setefSearch(30);
    const size_t nq = 16;
    // searchKnn is const: eight threads share the two indexes, each query searched twice (the first search of an index
    // also uploads it - once, under the lock)
    std::vector<std::vector<std::pair<float, size_t>>> res_l2(nq), res_ip(nq);
    std::vector<std::thread> pool;
    for (int t = 0; t < 8; ++t)
        pool.emplace_back([&, t] {
            for (size_t q = t % 4; q < nq; q += 4) {
                auto a = idx.SearchKNNCloserFirst(x.data() + q * dim, 10);
                auto b = ip.SearchKNNCloserFirst(x.data() + q * dim, 7);
                if (t < 4) {
                    res_l2[q] = a;
                    res_ip[q] = b;
                }
            }
        });
    for (auto& th : pool) th.join();
    for (size_t q = 0; q < nq; ++q)
        for (auto& p : res_l2[q]) std::printf("%zu %a\n", p.second, p.first);
    std::puts("ip");
    for (size_t q = 0; q < nq; ++q)
        for (auto& p : res_ip[q]) std::printf("%zu %a\n", p.second, p.first);
    std::puts("bruteforce");
    for (size_t q = 0; q < nq; ++q) {
        std::priority_queue<std::pair<float, size_t>> heap = bf.SearchKNN(x.data() + q * dim, 3);
        while (!heap.empty()) {  // farthest first: std::priority_queue<pair<dist, label>>, hnsw_core.hpp:1153-1158
            std::printf("%zu %a\n", heap.top().second, heap.top().first);
            heap.pop();
        }
    }
    return 0;
}
