"""GPU parity of the exact brute-force scan and of the shard merge against the oracle /
numpy: labels and distances bit for bit, duplicates and ragged shapes included."""
import numpy as np
import pytest

from vectordb_amd import datasets
from vectordb_amd import index as vi

pytestmark = pytest.mark.gpu


def oracle_bf(orc, x, labels, metric, q, k):
    b = orc.OracleBruteforce(metric, x.shape[1], max(len(x), 1), mode=orc.TREE64)
    if len(x):
        b.add(x, labels)
    return b.search(q, k)


@pytest.mark.parametrize("n,dim,metric,k", [
    (5000, 128, vi.L2, 10),
    (9000, 96, vi.IP, 10),      # more than two slabs of the sweep (4096 - k rows each)
    (3000, 40, vi.L2, 1),
    (700, 960, vi.L2, 100),
    (1000, 200, vi.IP, 33),
    (50, 16, vi.L2, 64),        # k > n: padded rows
])
def test_bruteforce_matches_oracle(orc, n, dim, metric, k):
    x = datasets.uniform(n, dim, 21)
    q = datasets.uniform(37, dim, 22)
    labels = (np.arange(n, dtype=np.uint64) * 7 + 3)
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    b = vi.BruteforceSearch(space, n)
    b.add_items(x, labels)
    b.to_device(0)
    d, l = b.search_batch(q, k)
    r = oracle_bf(orc, x, labels, metric, q, k)
    np.testing.assert_array_equal(l, r["label"].astype(np.uint64))
    np.testing.assert_array_equal(d.view(np.uint32), r["dist"].view(np.uint32))


def test_bruteforce_ties_by_label(orc):
    """Every vector stored 9 times with labels running against row order: the cut through a
    group of equal distances must keep the smallest labels (bruteforce.hpp:167 `<=` + pair order)."""
    rng = np.random.default_rng(3)
    base = rng.random((700, 12), dtype=np.float32)
    x = np.concatenate([base] * 9)[rng.permutation(6300)]
    labels = (20_000_000_000 - np.arange(6300)).astype(np.uint64)
    q = np.concatenate([base[:30], rng.random((10, 12), dtype=np.float32)])
    b = vi.BruteforceSearch(vi.L2Space(12), 6300)
    b.add_items(x, labels)
    b.to_device(0)
    for k in (4, 9, 10, 25):
        d, l = b.search_batch(q, k)
        r = oracle_bf(orc, x, labels, vi.L2, q, k)
        np.testing.assert_array_equal(l, r["label"].astype(np.uint64))
        np.testing.assert_array_equal(d.view(np.uint32), r["dist"].view(np.uint32))


def test_bruteforce_empty_and_errors():
    b = vi.BruteforceSearch(vi.L2Space(8), 4)
    b.to_device(0)
    d, l = b.search_batch(np.zeros((3, 8), np.float32), 2)
    assert np.all(np.isinf(d)) and np.all(l == np.uint64(2**64 - 1))
    b.add_items(np.ones((2, 8), np.float32))
    with pytest.raises(vi.VdbError):          # stale device copy
        b.search_batch(np.zeros((3, 8), np.float32), 2)
    b.to_device(0)
    d, l = b.search_batch(np.zeros((3, 8), np.float32), 2, precision=vi.BF_BF16)   # tiny index through the MFMA path
    assert sorted(l[0].tolist()) == [0, 1] and np.allclose(d, 8.0, rtol=1e-2)
    with pytest.raises(vi.VdbError) as e:
        b.search_batch(np.zeros((3, 8), np.float32), 2, precision=7)
    assert e.value.code == 1


def test_hnsw_full_ef_equals_bruteforce_on_device():
    """Size-independent property: ef >= n makes the traversal exhaustive on a connected graph."""
    x = datasets.uniform(200, 32, 5)
    q = datasets.uniform(50, 32, 6)
    h = vi.HierarchicalNSW(vi.L2Space(32), 200, 16, 100)
    h.add_items(x)
    h.to_device(0)
    h.setefSearch(256)
    b = vi.BruteforceSearch(vi.L2Space(32), 200)
    b.add_items(x)
    b.to_device(0)
    dh, lh = h.search_batch(q, 10)
    db, lb = b.search_batch(q, 10)
    np.testing.assert_array_equal(lh, lb)
    np.testing.assert_array_equal(dh.view(np.uint32), db.view(np.uint32))


@pytest.mark.parametrize("G,nq,k", [(2, 100, 10), (8, 333, 10), (8, 50, 100), (3, 7, 1)])
def test_merge_topk_kernel(G, nq, k):
    import torch
    rng = np.random.default_rng(G * 1000 + k)
    d = np.sort(rng.integers(0, 40, (G, nq, k)).astype(np.float32) / 8, axis=2)   # many equal distances
    l = rng.permutation(G * nq * k).reshape(G, nq, k).astype(np.int64) + (1 << 40)
    # each list ascending under (dist, label)
    order = np.lexsort((l, d), axis=2)
    d, l = np.take_along_axis(d, order, 2), np.take_along_axis(l, order, 2)
    d[0, 0, k // 2:] = np.inf                                                     # a short list, padded
    l[0, 0, k // 2:] = -1
    md, ml = vi.merge_topk(torch.from_numpy(d).cuda(), torch.from_numpy(l).cuda())
    torch.cuda.synchronize()
    alld = d.transpose(1, 0, 2).reshape(nq, -1)
    alll = l.transpose(1, 0, 2).reshape(nq, -1).view(np.uint64)
    o = np.lexsort((alll, alld), axis=1)[:, :k]
    np.testing.assert_array_equal(md.cpu().numpy(), np.take_along_axis(alld, o, 1))
    np.testing.assert_array_equal(ml.cpu().numpy().view(np.uint64), np.take_along_axis(alll, o, 1))


# ---- MFMA candidate pass (bf_mfma.hip): n >= 8192 routes VDB_BF_FP32 through it ------------

@pytest.mark.parametrize("n,dim,metric,k", [
    (20000, 128, vi.L2, 10),
    (12000, 96, vi.IP, 10),
    (9000, 40, vi.L2, 1),
    (8500, 960, vi.L2, 100),
    (10000, 200, vi.IP, 33),
])
def test_mfma_fp32_path_is_exact(orc, n, dim, metric, k):
    x = datasets.clustered(n, dim, 31, normalize=(metric == vi.IP))
    q = datasets.clustered(150, dim, 32, normalize=(metric == vi.IP))
    labels = (np.arange(n, dtype=np.uint64) * 3 + 11)
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    b = vi.BruteforceSearch(space, n)
    b.add_items(x, labels)
    b.to_device(0)
    d, l = b.search_batch(q, k, precision=vi.BF_FP32)
    r = oracle_bf(orc, x, labels, metric, q, k)
    np.testing.assert_array_equal(l, r["label"].astype(np.uint64))
    np.testing.assert_array_equal(d.view(np.uint32), r["dist"].view(np.uint32))


@pytest.mark.parametrize("n,nq,dim,metric", [(9100, 2500, 64, vi.L2), (16500, 700, 24, vi.IP)])
def test_mfma_many_query_tiles(orc, n, nq, dim, metric):
    """More than one 8-tile super-tile of queries, ragged in both directions (the persistent GEMM's
    tile walk, the padded panels and the DMA-staged thresholds all see partial tiles)."""
    x = datasets.clustered(n, dim, 51, normalize=(metric == vi.IP))
    q = datasets.clustered(nq, dim, 52, normalize=(metric == vi.IP))
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    b = vi.BruteforceSearch(space, n)
    b.add_items(x)
    b.to_device(0)
    d, l = b.search_batch(q, 10, precision=vi.BF_FP32)
    r = oracle_bf(orc, x, None, metric, q, 10)
    np.testing.assert_array_equal(l, r["label"].astype(np.uint64))
    np.testing.assert_array_equal(d.view(np.uint32), r["dist"].view(np.uint32))
    d2, l2 = b.search_batch(q, 10, precision=vi.BF_BF16_RERANK)
    assert datasets.recall_at_k(l2, l) >= 0.99


def test_mfma_fp32_ties_fall_back_to_exact(orc):
    """Duplicated rows: the k-th exact distance equals distances the candidate pass dropped, the
    certificate cannot hold, and the exact scan re-runs those queries."""
    rng = np.random.default_rng(9)
    base = rng.random((300, 48), dtype=np.float32)
    x = np.concatenate([base] * 40)[rng.permutation(12000)]
    labels = (5_000_000_000 - np.arange(12000)).astype(np.uint64)
    q = np.concatenate([base[:40], rng.random((24, 48), dtype=np.float32)])
    b = vi.BruteforceSearch(vi.L2Space(48), 12000)
    b.add_items(x, labels)
    b.to_device(0)
    for k in (10, 45, 90):
        d, l = b.search_batch(q, k)
        r = oracle_bf(orc, x, labels, vi.L2, q, k)
        np.testing.assert_array_equal(l, r["label"].astype(np.uint64))
        np.testing.assert_array_equal(d.view(np.uint32), r["dist"].view(np.uint32))


@pytest.mark.parametrize("dim,metric", [(128, vi.L2), (960, vi.L2), (96, vi.IP)])
def test_mfma_bf16_paths(orc, dim, metric):
    n, k = 30000, 10
    x = datasets.clustered(n, dim, 41, normalize=(metric == vi.IP))
    q = datasets.clustered(200, dim, 42, normalize=(metric == vi.IP))
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    b = vi.BruteforceSearch(space, n)
    b.add_items(x)
    b.to_device(0)
    r = oracle_bf(orc, x, None, metric, q, k)
    truth = r["label"].astype(np.uint64)
    # bf16 candidates + exact re-rank: every returned distance is the exact one of its label
    d2, l2 = b.search_batch(q, k, precision=vi.BF_BF16_RERANK)
    assert datasets.recall_at_k(l2, truth) >= 0.99
    allq = orc.OracleBruteforce(metric, dim, n, mode=orc.TREE64)
    allq.add(x)
    for i in range(0, 200, 37):
        for j in range(k):
            assert d2[i, j] == np.float32(orc.distance(metric, orc.TREE64, q[i], x[int(l2[i, j])]))
    assert np.all(np.diff(d2, axis=1) >= 0)
    # bf16 only: distances are bf16-grade, recall reported separately
    d1, l1 = b.search_batch(q, k, precision=vi.BF_BF16)
    rec = datasets.recall_at_k(l1, truth)
    assert rec >= 0.85, rec
    exact = np.array([[orc.distance(metric, orc.TREE64, q[i], x[int(l1[i, j])]) for j in range(k)] for i in range(0, 200, 20)])
    scale = np.abs(exact).max()
    np.testing.assert_allclose(d1[0:200:20], exact, atol=0.02 * scale)


@pytest.mark.parametrize("n,dim,metric,k,mod,keep", [(5000, 64, vi.L2, 10, 3, 1), (9000, 48, vi.IP, 25, 7, 0), (300, 16, vi.L2, 40, 2, 1),
                                                      (6000, 24, vi.L2, 10, 1000, 999), (20, 8, vi.L2, 32, 2, 0)])
def test_bruteforce_with_a_filter(orc, n, dim, metric, k, mod, keep):
    """bruteforce.hpp:154-181 with a BaseFilterFunctor.  The reference's filtered scan is order dependent (a row
    beyond the first K counts only at or below the worst accepted row of the first K), and so is the oracle's
    restatement; the device answer must equal it bit for bit - also when the filter accepts almost nothing
    (mod 1000), when K > n, and through the one-query spelling."""
    x = datasets.uniform(n, dim, 121)
    q = datasets.uniform(33, dim, 122)
    labels = np.random.default_rng(7).permutation(n).astype(np.uint64) + 17
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    b = vi.BruteforceSearch(space, n)
    b.add_items(x, labels)
    b.to_device(0)
    d, l = b.search_batch(q, k, filter=lambda lab: lab % mod == keep)
    ob = orc.OracleBruteforce(metric, dim, n, mode=orc.TREE64)
    ob.add(x, labels)
    r = ob.search(q, k, filter_mod=mod, filter_keep=keep)
    np.testing.assert_array_equal(l, r["label"].astype(np.uint64))
    np.testing.assert_array_equal(d.view(np.uint32), r["dist"].view(np.uint32))
    found = l[l != np.uint64(2**64 - 1)]
    assert (found % mod == keep).all()
    one = b.SearchKNNCloserFirst(q[0], k, filter=lambda lab: lab % mod == keep)
    assert [lab for _, lab in one] == [int(v) for v in l[0] if v != np.uint64(2**64 - 1)]


def test_mfma_bf16_duplicate_flood(orc):
    """Thousands of base rows at exactly the k-th distance: every one of them passes the running threshold, the waves'
    record lists of bf_gemm8_filter fill up (512 per wave and launch) and its in-place path and the buffers' overflow
    handling run.  With ties the labels are not determined by the bf16 pass, the DISTANCES are: they must be the exact
    top-k distances."""
    rng = np.random.default_rng(9)
    dim, n, k = 64, 40000, 10
    v0 = rng.standard_normal(dim).astype(np.float32)
    x = rng.standard_normal((n, dim)).astype(np.float32) * 3
    dup = rng.permutation(n)[:30000]
    x[dup] = v0
    q = (v0 + 0.01 * rng.standard_normal((300, dim))).astype(np.float32)
    b = vi.BruteforceSearch(vi.L2Space(dim), n)
    b.add_items(x)
    b.to_device(0)
    r = oracle_bf(orc, x, None, vi.L2, q[:40], k)
    d2, l2 = b.search_batch(q, k, precision=vi.BF_BF16_RERANK)
    np.testing.assert_array_equal(d2[:40].view(np.uint32), r["dist"].view(np.uint32))
    assert np.isin(l2, dup).all()
    d0, l0 = b.search_batch(q[:40], k)    # the f32 path stays bit-exact, labels included
    np.testing.assert_array_equal(l0, r["label"].astype(np.uint64))
    np.testing.assert_array_equal(d0.view(np.uint32), r["dist"].view(np.uint32))
