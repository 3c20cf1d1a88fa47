"""addPoint on a label that exists (updatePoint / repairConnectionsForUpdate, hnsw_core.hpp:844-980, :1001-1016) and
the replace-deleted insert (:804-842, ctor flag :123): the product's host builder and the oracle's restatement write
the same bytes after a mixed add / update / delete / replace sequence, in both summation orders; the device search
over the resulting graph equals the oracle's bit for bit (GPU part)."""
import numpy as np
import pytest

from tests.helpers import assert_same_results
from vectordb_amd import index as vi


def rand(n, d, seed):
    return np.random.default_rng(seed).random((n, d), dtype=np.float32)


def mixed_sequence(h, o, x, y, reuse):
    """The same calls on the product (h) and on the oracle (o)."""
    n = x.shape[0]
    for i in range(n):  # plain inserts
        h.addPoint(x[i], 1000 + i)
        o.add_point(x[i], 1000 + i)
    for i in range(0, n, 7):  # updates in place: same label, new vector
        h.addPoint(y[i], 1000 + i)
        o.add_point(y[i], 1000 + i)
    for i in range(3, n, 5):
        h.markDelete(1000 + i)
        o.mark_delete(1000 + i)
    if reuse:
        for j in range(12):  # take vacant places (smallest internal id first), new labels
            h.addPoint(y[n - 1 - j], 5000 + j, replace_deleted=True)
            o.add_point(y[n - 1 - j], 5000 + j, replace_deleted=True)
    else:
        for i in range(3, n, 10):  # addPoint on a deleted label un-deletes and updates it (:1010-1013)
            h.addPoint(y[i], 1000 + i)
            o.add_point(y[i], 1000 + i)
    h.addPoint(x[0] * 0.5, 1000)  # the first element (often the entry point's neighbourhood) once more
    o.add_point(x[0] * 0.5, 1000)


@pytest.mark.parametrize("metric,dim,M,order", [(vi.L2, 16, 4, "seq"), (vi.L2, 24, 8, "tree64"), (vi.IP, 33, 6, "seq"), (vi.IP, 20, 16, "tree64")])
@pytest.mark.parametrize("reuse", [False, True])
def test_update_and_replace_write_the_oracles_bytes(orc, tmp_path, metric, dim, M, order, reuse):
    n = 300
    x, y = rand(n, dim, 1), rand(n, dim, 2)
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    h = vi.HierarchicalNSW(space, n, M, 40, 100, reuse_deleted=reuse)
    h.set_build_sum_order(vi.SUM_SEQ if order == "seq" else vi.SUM_TREE64)
    o = orc.OracleHNSW(metric, dim, n, M, 40, 100, mode=orc.SEQ if order == "seq" else orc.TREE64, reuse_deleted=reuse)
    mixed_sequence(h, o, x, y, reuse)
    assert h.getElementCount() == o.count == n
    assert h.getDeletedCount() == o.deleted
    h.saveIndex(str(tmp_path / "h.bin"))
    o.save(str(tmp_path / "o.bin"))
    assert open(tmp_path / "h.bin", "rb").read() == open(tmp_path / "o.bin", "rb").read()
    rc = h.checkIntegrity()[0]
    assert rc in (0, 4)  # (4 = a node without inbound links: allowed, updates can orphan one)


def test_update_error_behaviour(orc):
    x = rand(20, 8, 3)
    h = vi.HierarchicalNSW(vi.L2Space(8), 20, 4, 16)
    for i in range(10):
        h.addPoint(x[i], i)
    with pytest.raises(vi.VdbError, match="disabled in constructor"):  # hnsw_core.hpp:805-807
        h.addPoint(x[10], 99, replace_deleted=True)
    h.addPoint(x[11], 3)  # update: the count stays
    assert h.getElementCount() == 10
    h.markDelete(4)
    h.addPoint(x[12], 4)  # un-deletes (:1010-1012)
    assert h.getDeletedCount() == 0
    r = vi.HierarchicalNSW(vi.L2Space(8), 20, 4, 16, reuse_deleted=True)
    for i in range(10):
        r.addPoint(x[i], i)
    r.markDelete(5)
    with pytest.raises(vi.VdbError, match="Can't use addPoint to update deleted elements"):  # :1003-1007
        r.addPoint(x[13], 5)
    r.addPoint(x[13], 77, replace_deleted=True)  # takes the place of label 5
    assert r.getElementCount() == 10 and r.getDeletedCount() == 0
    with pytest.raises(vi.VdbError, match="Label not found"):
        r.markDelete(5)
    r.markDelete(77)
    r.addPoint(x[14], 78, replace_deleted=False)  # no replacement asked: appended
    assert r.getElementCount() == 11
    # a bulk insert whose labels exist is n sequential addPoint calls
    h.add_items(x[:4] * 2.0, np.array([0, 1, 50, 1], np.uint64))
    assert h.getElementCount() == 11


@pytest.mark.gpu
@pytest.mark.parametrize("reuse", [False, True])
def test_device_search_over_an_updated_graph(orc, tmp_path, reuse):
    n, dim = 2000, 32
    x, y = rand(n, dim, 5), rand(n, dim, 6)
    h = vi.HierarchicalNSW(vi.L2Space(dim), n, 8, 60, 100, reuse_deleted=reuse)
    o = orc.OracleHNSW(orc.L2, dim, n, 8, 60, 100, mode=orc.TREE64, reuse_deleted=reuse)
    mixed_sequence(h, o, x, y, reuse)
    q = rand(200, dim, 7)
    for ef in (10, 48):
        h.setefSearch(ef)
        o.set_ef(ef)
        h.to_device(0)
        dist, label, st = h.search_batch(q, 10, stats=True)
        assert_same_results((dist, label), o.search(q, 10), st)
    # an updated label is found at its NEW vector (label 1000 was updated last to x[0] * 0.5)
    h.setefSearch(64)
    d, l = h.search_batch(x[:1] * 0.5, 1)
    assert l[0, 0] == 1000 and d[0, 0] == 0.0
