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
    with pytest.raises(vi.VdbError) as e:
        b.search_batch(np.zeros((3, 8), np.float32), 2, precision=vi.BF_BF16)
    assert e.value.code == 4


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
