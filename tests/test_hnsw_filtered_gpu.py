"""The non-bare path of searchKnn (hnsw_core.hpp:1141-1148): deleted elements and filter functors.

GPU == oracle(tree64) bit for bit, counters included.  The oracle's searchBaseLayerST<false> is the
restatement of SURVEY.md §8a row A5 (candidates always, results only when not deleted and
filter(label), stop test `cand > lowerBound && size == ef`)."""
import numpy as np
import pytest

from vectordb_amd import _lib, datasets
from vectordb_amd import index as vi
from tests.helpers import assert_same_results, build_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    o.build()
    return o


def _delete_both(h, o, labels):
    for l in labels:
        h.markDelete(int(l))
        o.mark_delete(int(l))


@pytest.mark.parametrize("frac,ef,k", [(0.1, 32, 10), (0.5, 64, 10), (0.9, 48, 5), (0.3, 200, 100), (0.02, 10, 10)])
def test_deleted_elements(orc, frac, ef, k):
    n, dim = 6000, 32
    x = datasets.clustered(n, dim, 61)
    q = datasets.clustered(300, dim, 62)
    labels = np.arange(n, dtype=np.uint64) * 7 + 3
    h, o = build_pair(orc, x, labels=labels, n_threads=4)
    rng = np.random.default_rng(5)
    dead = labels[rng.permutation(n)[: int(frac * n)]]
    _delete_both(h, o, dead)
    assert h.getDeletedCount() == len(dead) == o.deleted
    h.to_device(0)
    h.setefSearch(ef)
    o.set_ef(ef)
    d, l, st = h.search_batch(q, k, stats=True)
    r = o.search(q, k)
    assert_same_results((d, l), r, st)
    assert not np.isin(l, dead).any()


def test_deleted_entry_point(orc):
    """:302-312: an entry point that is not allowed seeds the candidates only."""
    n, dim = 3000, 16
    x = datasets.uniform(n, dim, 63)
    q = datasets.uniform(100, dim, 64)
    h, o = build_pair(orc, x, n_threads=2)
    ep_label = int(h.labels()[h.entry])
    _delete_both(h, o, [ep_label])
    h.to_device(0)
    for ef, k in ((1, 1), (16, 4), (64, 10)):
        h.setefSearch(ef)
        o.set_ef(ef)
        d, l, st = h.search_batch(q, k, stats=True)
        assert_same_results((d, l), o.search(q, k), st)
        assert not (l == ep_label).any()


@pytest.mark.parametrize("mod,keep,ef", [(2, 0, 40), (5, 3, 64), (16, 7, 40)])
def test_filter_functor(orc, mod, keep, ef):
    n, dim = 8000, 48
    x = datasets.clustered(n, dim, 65, normalize=True)
    q = datasets.clustered(200, dim, 66, normalize=True)
    labels = np.random.default_rng(3).permutation(n).astype(np.uint64) + 1000
    h, o = build_pair(orc, x, metric=vi.IP, labels=labels, n_threads=4)
    h.to_device(0)
    h.setefSearch(ef)
    o.set_ef(ef)
    d, l, st = h.search_batch(q, 10, stats=True, filter=lambda lab: lab % mod == keep)
    r = o.search(q, 10, filter_mod=mod, filter_keep=keep)
    assert_same_results((d, l), r, st)
    found = l[l != np.uint64(0xFFFFFFFFFFFFFFFF)]
    assert (found % mod == keep).all()
    # the reference's one-query spelling with a functor
    one = h.SearchKNNCloserFirst(q[0], 10, filter=lambda lab: lab % mod == keep)
    assert [b for _, b in one] == [int(v) for v in l[0] if v != np.uint64(0xFFFFFFFFFFFFFFFF)]


@pytest.mark.parametrize("mod,ef", [(2, 64), (4, 40), (10, 32), (3, 128)])
def test_repeated_filtered_launches_adapt_and_stay_exact(orc, mod, ef):
    """The first level of a filtered search keeps the candidates that are not allowed in a buffer (64 keys in registers,
    256 in LDS to begin with): a filter that keeps 1/mod of the elements needs about ef * (mod - 1) live ones.  Host
    arrays: the verdicts are counted and the first launch sizes the buffer from them; whatever still overflows is re-run
    in the large configuration and its report doubles the buffer of the next launches at this ef: every launch equals
    the oracle, whichever kernels answered it."""
    n, dim = 20000, 24
    x = datasets.clustered(n, dim, 71)
    q = datasets.clustered(600, dim, 72)
    h, o = build_pair(orc, x, n_threads=4)
    h.to_device(0)
    h.setefSearch(ef)
    o.set_ef(ef)
    r = o.search(q, 10, filter_mod=mod, filter_keep=1)
    for _ in range(4):   # (search_batch on host arrays returns after the launch has completed: each report is read by the next)
        d, l, st = h.search_batch(q, 10, stats=True, filter=lambda lab: lab % mod == 1)
        assert_same_results((d, l), r, st)
    # an unfiltered search at the same ef is not disturbed by what the filtered ones reported
    assert_same_results(h.search_batch(q, 10, stats=True)[:2], o.search(q, 10), h.search_batch(q, 10, stats=True)[2])


def test_filter_and_deleted_on_device_tensors(orc):
    import torch
    n, dim = 5000, 24
    x = datasets.clustered(n, dim, 67)
    q = datasets.clustered(128, dim, 68)
    h, o = build_pair(orc, x, n_threads=4)
    dead = np.arange(0, n, 3, dtype=np.uint64)
    _delete_both(h, o, dead)
    h.to_device(0)
    h.setefSearch(50)
    o.set_ef(50)
    bitmap = torch.frombuffer(bytearray(h.filter_bitmap(lambda lab: lab % 4 == 1)), dtype=torch.int32).cuda()
    dq = torch.from_numpy(q).cuda()
    d, l, st = h.search_batch(dq, 10, stats=True, filter=bitmap)
    r = o.search(q, 10, filter_mod=4, filter_keep=1)
    assert_same_results((d.cpu().numpy(), l.cpu().numpy().view(np.uint64)), r, st.cpu().numpy().view(np.uint32))


def test_everything_filtered_out(orc):
    """A filter that accepts nothing: every node is a candidate that is never a result, the search walks the whole
    graph and returns nothing - as the reference's unbounded heaps do (hnsw_core.hpp:1141-1148).  On the device that is
    the heap-in-HBM kernel behind the register configurations; counters equal the oracle's."""
    n, dim = 2000, 16
    x = datasets.uniform(n, dim, 69)
    q = datasets.uniform(20, dim, 70)
    h, o = build_pair(orc, x)
    h.to_device(0)
    h.setefSearch(20)
    o.set_ef(20)
    d, l, st = h.search_batch(q, 5, stats=True, filter=np.zeros(n, bool))
    assert (l == np.uint64(0xFFFFFFFFFFFFFFFF)).all() and np.isinf(d).all()
    r = o.search(q, 5, filter_mod=1, filter_keep=5)  # label % 1 == 5: never
    assert (r["count"] == 0).all()
    assert_same_results((d, l), r, st)
    assert (st[:, 0] >= n).all()  # every element was scored on layer 0 (plus the descent)


@pytest.mark.parametrize("mod,ef,k", [(400, 10, 5), (1500, 64, 10), (97, 200, 100)])
def test_extreme_selectivity_is_answered(orc, mod, ef, k):
    """Filters that keep one element in `mod`: far more live candidates outside the result set than any register array
    holds (the queries end up in hnsw_search_unbounded_kernel).  Answers and counters equal the oracle's bit for bit."""
    n, dim = 12000, 24
    x = datasets.clustered(n, dim, 81)
    q = datasets.clustered(64, dim, 82)
    h, o = build_pair(orc, x, n_threads=4)
    for lab in range(0, n, 7):  # with deleted elements as well
        h.markDelete(lab)
        o.mark_delete(lab)
    h.to_device(0)
    h.setefSearch(ef)
    o.set_ef(ef)
    d, l, st = h.search_batch(q, k, stats=True, filter=lambda lab: lab % mod == 3)
    r = o.search(q, k, filter_mod=mod, filter_keep=3)
    assert_same_results((d, l), r, st)
    found = l[l != np.uint64(0xFFFFFFFFFFFFFFFF)]
    assert len(found) and (found % mod == 3).all() and (found % 7 != 0).all()
    # and with the epsilon stop condition on top
    eps = float(np.median(r["dist"][:, 0]) * 1.5)
    de, le, cnt, ste = h.search_stop_condition_closest_batch(q, eps, 2, ef, filter=lambda lab: lab % mod == 3)
    re = o.search_epsilon(q, eps, 2, ef, filter_mod=mod, filter_keep=3)
    np.testing.assert_array_equal(cnt, re["count"])
    np.testing.assert_array_equal(le, re["label"].astype(np.uint64))
    np.testing.assert_array_equal(de.view(np.uint32), re["dist"].view(np.uint32))
    np.testing.assert_array_equal(ste[:, 0], re["ndis"].astype(np.uint32))


def test_device_against_committed_golden():
    """The committed fixtures (tests/golden, tree64 order) straight against the device path: the saved
    256 x 16 index file, bare searches, then every 5th element deleted with and without the functor."""
    import os
    gold_dir = os.path.join(os.path.dirname(__file__), "golden")
    gold = np.load(os.path.join(gold_dir, "hnsw_golden.npz"))
    h = vi.HierarchicalNSW(vi.L2Space(16), location=os.path.join(gold_dir, "hnsw_256x16.bin"))
    q = gold["q"]

    def check(tag, ef, k, flt):
        h.setefSearch(ef)
        d, l, st = h.search_batch(q, k, stats=True, filter=flt)
        np.testing.assert_array_equal(l, gold[f"{tag}_ef{ef}_k{k}_label"].astype(np.uint64))
        np.testing.assert_array_equal(d.view(np.uint32), gold[f"{tag}_ef{ef}_k{k}_dist"].view(np.uint32))
        np.testing.assert_array_equal(st[:, 0], gold[f"{tag}_ef{ef}_k{k}_ndis"])
        np.testing.assert_array_equal(st[:, 1], gold[f"{tag}_ef{ef}_k{k}_nhops0"])
        np.testing.assert_array_equal(st[:, 2], gold[f"{tag}_ef{ef}_k{k}_nhops_upper"])

    h.to_device(0)
    for ef in (8, 16, 64):
        for k in (1, 10):
            check("tree64", ef, k, None)
    for lab in range(0, 256, 5):
        h.markDelete(lab)
    h.to_device(0)
    for ef, k in ((16, 5), (64, 10)):
        check("nonbare_del", ef, k, None)
        check("nonbare_delmod3", ef, k, lambda lab: lab % 3 == 1)


@pytest.mark.parametrize("mod,ef", [(8, 64), (12, 100), (25, 40)])
def test_filter_as_device_bitmap_adapts_from_reports(orc, mod, ef):
    """The same through a DEVICE bitmap (vdb_search_batch_device_filtered): the library cannot count the verdicts, so the
    first launch runs with the smallest candidate buffer, most queries overflow it and are answered by the large
    configuration (and the heap kernel behind it); the reports then grow the buffer launch by launch, and the visited
    set moves to HBM when the LDS table is outgrown.  Every launch equals the oracle bit for bit."""
    import torch
    n, dim = 30000, 24
    x = datasets.clustered(n, dim, 91)
    q = datasets.clustered(512, dim, 92)
    h, o = build_pair(orc, x, n_threads=4)
    h.to_device(0)
    h.setefSearch(ef)
    o.set_ef(ef)
    r = o.search(q, 10, filter_mod=mod, filter_keep=2)
    bitmap = torch.frombuffer(bytearray(h.filter_bitmap(lambda lab: lab % mod == 2)), dtype=torch.int32).cuda()
    dq = torch.from_numpy(q).cuda()
    for _ in range(6):
        d, l, st = h.search_batch(dq, 10, stats=True, filter=bitmap)
        torch.cuda.synchronize()
        assert_same_results((d.cpu().numpy(), l.cpu().numpy().view(np.uint64)), r, st.cpu().numpy().view(np.uint32))


def test_many_equal_distances_among_rejected_candidates(orc):
    """Blocks of identical rows that the filter rejects: hundreds of candidates at exactly the same distance go through
    the candidate buffer (refills move the smallest (distance, id) first, ties included) - ids and counters equal the
    oracle's."""
    rng = np.random.default_rng(17)
    base = datasets.clustered(40, 16, 93)
    x = np.repeat(base, 150, axis=0)               # 6 000 rows, 150 copies of each of 40 vectors
    x = x[rng.permutation(len(x))].copy()
    q = datasets.clustered(100, 16, 94)
    h, o = build_pair(orc, x, M=8, efc=60, n_threads=4)
    h.to_device(0)
    for ef, mod in ((20, 7), (64, 31), (128, 5)):
        h.setefSearch(ef)
        o.set_ef(ef)
        r = o.search(q, 10, filter_mod=mod, filter_keep=1)
        for _ in range(3):
            d, l, st = h.search_batch(q, 10, stats=True, filter=lambda lab: lab % mod == 1)
            assert_same_results((d, l), r, st)
