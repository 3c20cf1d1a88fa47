"""GPU parity of the traversal: device search == oracle search on the same graph, neighbour
ids AND distances bit for bit, plus the reference's work counters (ndis / hops)."""
import numpy as np
import pytest

from tests.helpers import assert_same_results, build_pair
from vectordb_amd import datasets
from vectordb_amd import index as vi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,dim,M,efc,metric", [
    (2000, 128, 16, 100, vi.L2),
    (1500, 16, 4, 32, vi.L2),
    (1200, 96, 16, 100, vi.IP),
    (800, 40, 8, 64, vi.L2),      # ragged dim: not a multiple of 16 / 32 / 64
    (600, 200, 12, 64, vi.IP),    # 4 chunks, maxM0 = 24
    (300, 960, 16, 64, vi.L2),    # GIST-like rows: 15 chunks
])
def test_search_matches_oracle(orc, n, dim, M, efc, metric):
    x = datasets.uniform(n, dim, 1)
    q = datasets.uniform(64, dim, 2)
    if metric == vi.IP:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    h, o = build_pair(orc, x, metric, M, efc)
    h.to_device(0)
    for ef in (8, 16, 64, 100, 200):
        for k in (1, 10):
            h.setefSearch(ef)
            o.set_ef(ef)
            dist, label, st = h.search_batch(q, k, stats=True)
            assert_same_results((dist, label), o.search(q, k), st)


def test_clustered_c1_like(orc):
    x = datasets.clustered(4000, 128, 1)
    q = datasets.clustered(200, 128, 2)
    h, o = build_pair(orc, x, vi.L2, 16, 200)
    h.to_device(0)
    h.setefSearch(64)
    o.set_ef(64)
    dist, label, st = h.search_batch(q, 10, stats=True)
    assert_same_results((dist, label), o.search(q, 10), st)


def test_labels_not_ids_and_ties(orc):
    """Duplicate vectors force exact distance ties: eviction order (dist, id), output order
    (dist, label) with labels running AGAINST ids."""
    rng = np.random.default_rng(5)
    base = rng.random((40, 24), dtype=np.float32)
    x = np.concatenate([base] * 12, axis=0)           # every vector 12 times
    perm = rng.permutation(x.shape[0])
    x = x[perm]
    labels = (10_000_000_000 - np.arange(x.shape[0])).astype(np.uint64)  # descending, > 2^32
    q = np.concatenate([base[:20], rng.random((20, 24), dtype=np.float32)])
    h, o = build_pair(orc, x, vi.L2, 8, 64, labels=labels)
    h.to_device(0)
    for ef, k in ((10, 10), (16, 5), (64, 10), (64, 30), (100, 100)):
        h.setefSearch(ef)
        o.set_ef(ef)
        dist, label, st = h.search_batch(q, k, stats=True)
        assert_same_results((dist, label), o.search(q, k), st)


def test_small_and_empty_indexes(orc):
    dim = 32
    q = datasets.uniform(5, dim, 3)
    # empty index -> every row padded (hnsw_core.hpp:1109)
    h, o = build_pair(orc, np.zeros((0, dim), np.float32))
    h.to_device(0)
    dist, label, st = h.search_batch(q, 3, stats=True)
    assert np.all(np.isinf(dist)) and np.all(label == np.uint64(2**64 - 1)) and np.all(st == 0)
    # fewer elements than k
    x = datasets.uniform(7, dim, 4)
    h, o = build_pair(orc, x, M=4, efc=16)
    h.to_device(0)
    h.setefSearch(10)
    o.set_ef(10)
    dist, label, st = h.search_batch(q, 10, stats=True)
    assert_same_results((dist, label), o.search(q, 10), st)
    assert np.all(st[:, 3] == 7)


@pytest.mark.parametrize("nq,spill", [(32, "1"), (32, "0"), (700, "1"), (32, None)])
def test_visited_overflow_is_exact(orc, monkeypatch, nq, spill):
    """ef=256 on a graph of 6000 nodes visits most of it: every query fills its LDS table.  With spill
    tables ("1") it carries on in a second-level table in HBM (nq=700: more spilling queries than the
    pool's 704 tables can be short of when they are resident at once); without ("0") it is re-run by the
    large configuration (bitmap).  Unset, the library
    hands out the tables from the second launch on.  Same answers every way."""
    if spill is None:
        monkeypatch.delenv("VDB_SPILL", raising=False)
    else:
        monkeypatch.setenv("VDB_SPILL", spill)
    x = datasets.uniform(6000, 16, 11)
    q = datasets.uniform(nq, 16, 12)
    h, o = build_pair(orc, x, vi.L2, 16, 64)
    h.to_device(0)
    h.setefSearch(256)
    o.set_ef(256)
    for _ in range(3):  # later launches adapt (spill tables, larger LDS table) and reuse dirty spill tables
        dist, label, st = h.search_batch(q, 100, stats=True)
        assert_same_results((dist, label), o.search(q, 100), st)


@pytest.mark.parametrize("dim,ef", [(100, 200), (40, 300), (200, 160)])
def test_table_size_and_rows_in_flight_follow_the_reports(orc, monkeypatch, dim, ef):
    """Round 3: launches with spill tables behind a table of 16 KB or more take the kernels with twice the rows in flight
    at ef > 124, and from the third launch on a 20 / 22 KB table chosen from the reported average of visited nodes
    (launch_cfg).  Five launches walk through those configurations; every one equals the oracle."""
    monkeypatch.delenv("VDB_SPILL", raising=False)
    x = datasets.uniform(7000, dim, 41)
    q = datasets.uniform(200, dim, 42)
    h, o = build_pair(orc, x, vi.L2, 16, 64)
    h.to_device(0)
    h.setefSearch(ef)
    o.set_ef(ef)
    ref = o.search(q, 100)
    for _ in range(5):
        dist, label, st = h.search_batch(q, 100, stats=True)
        assert_same_results((dist, label), ref, st)


def test_batches_on_two_streams_in_turn(orc):
    """a serving loop enqueues consecutive batches on two streams in turn (the next launch fills the last
    one's tail).  Scratch is per (index, stream), the adaptive state per index: eight launches of two different batches
    overlapping on one index, every one equal to the oracle."""
    import torch
    x = datasets.clustered(30000, 64, 51)
    qs = [datasets.clustered(3000, 64, 52 + i) for i in range(2)]
    h, o = build_pair(orc, x, vi.L2, 16, 100)
    h.to_device(0)
    h.setefSearch(150)
    o.set_ef(150)
    refs = [o.search(q, 10) for q in qs]
    dq = [torch.from_numpy(q).cuda() for q in qs]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    outs = []
    for i in range(8):
        with torch.cuda.stream(streams[i & 1]):
            outs.append((i & 1, h.search_batch(dq[i & 1], 10, stats=True)))
    torch.cuda.synchronize()
    for which, (dist, label, st) in outs:
        assert_same_results((dist.cpu().numpy(), label.cpu().numpy().view(np.uint64), ), refs[which], st.cpu().numpy().view(np.uint32))


def test_spill_pool_grows_with_the_batch(orc, monkeypatch):
    """The pool holds one spill table per query of the largest batch (512 to 4096) and a table goes back
    to it when its query ends: a 20 000-query batch after a small one re-allocates the pool, and its
    20 000 spilling queries share the 4 096 tables (each is taken about five times in the launch).  The
    answers equal those of a run without spill tables (bitmap re-runs), and the oracle's on a sample."""
    x = datasets.uniform(6000, 16, 11)
    q = datasets.uniform(20000, 16, 31)
    h, o = build_pair(orc, x, vi.L2, 16, 64)
    h.to_device(0)
    h.setefSearch(256)
    o.set_ef(256)
    h.set_spill_mode(1)
    h.search_batch(q[:32], 100)
    d1, l1, st1 = h.search_batch(q, 100, stats=True)
    h.set_spill_mode(0)
    d0, l0, st0 = h.search_batch(q, 100, stats=True)
    np.testing.assert_array_equal(l1, l0)
    np.testing.assert_array_equal(d1.view(np.uint32), d0.view(np.uint32))
    np.testing.assert_array_equal(st1, st0)
    sample = np.arange(0, 20000, 100)
    assert_same_results((d1[sample], l1[sample]), o.search(q[sample], 100), st1[sample])


@pytest.mark.parametrize("metric", [vi.L2, vi.IP])
def test_spill_tables_at_small_ef(orc, monkeypatch, metric):
    """ef=100 (two key registers per lane, 2^12-entry LDS table) on a dense small graph: some queries
    spill, most do not - both in one launch."""
    monkeypatch.setenv("VDB_SPILL", "1")
    x = datasets.uniform(9000, 8, 21)
    q = datasets.uniform(300, 8, 22)
    h, o = build_pair(orc, x, metric, 32, 64)
    h.to_device(0)
    for ef in (100, 124):
        h.setefSearch(ef)
        o.set_ef(ef)
        dist, label, st = h.search_batch(q, 10, stats=True)
        assert_same_results((dist, label), o.search(q, 10), st)


@pytest.mark.parametrize("ef,k", [(300, 300), (316, 100), (317, 10), (508, 508), (124, 124), (125, 100), (60, 60), (61, 50),
                                  (188, 188), (189, 100), (252, 252), (253, 10)])
def test_large_ef_and_key_array_boundaries(orc, ef, k):
    """Every key-array size (1, 2, 3, 4, 5, 8 registers per lane) at its boundaries, k up to ef = 508."""
    x = datasets.clustered(20000, 24, 13)
    q = datasets.clustered(64, 24, 14)
    h, o = build_pair(orc, x, vi.L2, 16, 100, n_threads=4)
    h.to_device(0)
    h.setefSearch(ef)
    o.set_ef(ef)
    dist, label, st = h.search_batch(q, k, stats=True)
    assert_same_results((dist, label), o.search(q, k), st)


def test_ef_above_the_device_limit_is_reported(orc):
    x = datasets.uniform(500, 8, 15)
    h, _ = build_pair(orc, x, vi.L2, 8, 32)
    h.to_device(0)
    h.setefSearch(509)
    with pytest.raises(vi.VdbError) as e:
        h.search_batch(x[:4], 5)
    assert e.value.code == 4


def test_device_pointer_api_and_errors(orc):
    import torch
    x = datasets.uniform(1000, 64, 1)
    q = datasets.uniform(100, 64, 2)
    h, o = build_pair(orc, x, vi.L2, 16, 64)
    with pytest.raises(vi.VdbError) as e:
        h.search_batch(q, 10)                      # not uploaded yet
    assert e.value.code == 3
    h.to_device(0)
    h.setefSearch(32)
    o.set_ef(32)
    tq = torch.from_numpy(q).cuda()
    dist, label = h.search_batch(tq, 10)
    torch.cuda.synchronize()
    ref = o.search(q, 10)
    np.testing.assert_array_equal(label.cpu().numpy().view(np.uint64), ref["label"].astype(np.uint64))
    np.testing.assert_array_equal(dist.cpu().numpy().view(np.uint32), ref["dist"].view(np.uint32))
    h.setefSearch(600)
    with pytest.raises(vi.VdbError) as e:
        h.search_batch(q, 10)
    assert e.value.code == 4
    h.setefSearch(32)
    h.addPoint(x[0] + 1.0, 5000) if h.getCapacity() > 1000 else None
    h.markDelete(3)
    with pytest.raises(vi.VdbError):
        h.search_batch(q, 10)                      # stale device copy / deleted elements


def test_hundreds_of_identical_vectors(orc):
    """300 copies of one vector beside 2000 others: a query near the copies holds hundreds of keys at
    exactly the bound's distance.  The reference expands every one of them.  Rows the device answers
    equal the oracle's bit for bit; a query whose tie keys outgrow even the large configuration's array
    is flagged (found = UINT32_MAX, NaN row), never answered from a truncated search."""
    import torch
    rng = np.random.default_rng(91)
    x = datasets.uniform(2300, 16, 92)
    x[500:800] = x[500]
    q = np.concatenate([x[500:501] + 1e-3 * rng.standard_normal((40, 16)).astype(np.float32), datasets.uniform(60, 16, 93)])
    h, o = build_pair(orc, x, vi.L2, 16, 64)
    h.to_device(0)
    for ef in (32, 100):
        h.setefSearch(ef)
        o.set_ef(ef)
        d, l, st = (t.cpu().numpy() for t in h.search_batch(torch.from_numpy(q).cuda(), 10, stats=True))
        ref = o.search(q, 10)
        flagged = st.view(np.uint32)[:, 3] == 0xFFFFFFFF
        ok = ~flagged
        assert ok[40:].all()                                   # queries away from the copies are always answered
        assert_same_results((d[ok], l[ok].view(np.uint64)), {key: v[ok] for key, v in ref.items() if key != "seconds"},
                            st.view(np.uint32)[ok])
        assert np.isnan(d[flagged]).all() and (l[flagged] == -1).all()
        if flagged.any():
            with pytest.raises(vi.VdbError) as e:
                h.search_batch(q, 10)
            assert e.value.code == 4


def test_corrupt_graphs_are_refused(orc, tmp_path):
    """A file whose entry point or links do not fit its own level structure must be an error at load /
    upload (the reference: "Index seems to be corrupted"), not an out-of-bounds read in a kernel."""
    import struct
    x = datasets.uniform(3000, 8, 94)
    h, _ = build_pair(orc, x, vi.L2, 8, 32)
    path = str(tmp_path / "g.bin")
    h.saveIndex(path)
    raw = bytearray(open(path, "rb").read())
    # header: 6 x size_t, int max_level, u32 entry, 5 x size_t (hnsw_core.hpp:555-568)
    off_entry = 6 * 8 + 4
    bad = bytearray(raw)
    bad[off_entry:off_entry + 4] = struct.pack("<I", 3000)          # entry point outside the index
    open(path, "wb").write(bad)
    with pytest.raises(vi.VdbError):
        vi.HierarchicalNSW(vi.L2Space(8), location=path)
    # an upper-layer link to an element that has no upper layers
    lvl0 = next(i for i in range(3000) if h.level_of(i) == 0)
    hdr = 6 * 8 + 4 + 4 + 5 * 8
    stride = struct.unpack_from("<Q", raw, 3 * 8)[0]
    pos = hdr + 3000 * stride
    patched = False
    for i in range(3000):
        nbytes = struct.unpack_from("<I", raw, pos)[0]
        pos += 4
        if nbytes and not patched and struct.unpack_from("<H", raw, pos)[0] > 0:
            struct.pack_into("<I", raw, pos + 4, lvl0)              # first link of this node's layer-1 list
            patched = True
        pos += nbytes
    assert patched
    open(path, "wb").write(raw)
    g = vi.HierarchicalNSW(vi.L2Space(8), location=path)
    with pytest.raises(vi.VdbError) as e:
        g.to_device(0)
    assert "corrupted" in str(e.value)
