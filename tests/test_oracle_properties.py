"""Self-consistency of the oracle that needs no fixture."""
import numpy as np
import pytest

from vectordb_amd import datasets


def test_bruteforce_equals_argsort(orc):
    rng = np.random.default_rng(3)
    x = rng.random((500, 32), dtype=np.float32)
    q = rng.random((20, 32), dtype=np.float32)
    for metric in (orc.L2, orc.IP):
        b = orc.OracleBruteforce(metric, 32, 500, mode=orc.SEQ)
        b.add(x)
        r = b.search(q, 10)
        x64, q64 = x.astype(np.float64), q.astype(np.float64)
        if metric == orc.L2:
            d = ((q64[:, None, :] - x64[None]) ** 2).sum(-1)
        else:
            d = 1.0 - q64 @ x64.T
        want = np.argsort(d, axis=1, kind="stable")[:, :10]
        np.testing.assert_array_equal(r["label"], want.astype(np.uint64))
        np.testing.assert_allclose(r["dist"], np.take_along_axis(d, want, 1), rtol=1e-5, atol=1e-6)


def test_hnsw_with_full_ef_is_exact(orc):
    """ef >= N on a connected graph visits everything: HNSW == brute force."""
    rng = np.random.default_rng(4)
    x = rng.random((300, 24), dtype=np.float32)
    q = rng.random((25, 24), dtype=np.float32)
    o = orc.OracleHNSW(orc.L2, 24, 300, 8, 64, 100, mode=orc.TREE64)
    o.add(x)
    assert o.check_integrity()[0] == 0
    o.set_ef(300)
    b = orc.OracleBruteforce(orc.L2, 24, 300, mode=orc.TREE64)
    b.add(x)
    r, rb = o.search(q, 10), b.search(q, 10)
    np.testing.assert_array_equal(r["label"], rb["label"])
    np.testing.assert_array_equal(r["dist"], rb["dist"])
    assert np.all(r["ndis"] >= 299)


def test_save_load_search_identical(orc, tmp_path):
    rng = np.random.default_rng(6)
    x = rng.random((400, 20), dtype=np.float32)
    q = rng.random((10, 20), dtype=np.float32)
    o = orc.OracleHNSW(orc.L2, 20, 400, 6, 40, 100, mode=orc.TREE64)
    o.add(x)
    o.set_ef(30)
    r = o.search(q, 5)
    o.save(str(tmp_path / "i.bin"))
    o2 = orc.OracleHNSW.load(str(tmp_path / "i.bin"), orc.L2, 20, mode=orc.TREE64)
    o2.set_ef(30)
    r2 = o2.search(q, 5)
    for name in ("dist", "label", "ndis", "nhops0", "nhops_upper"):
        np.testing.assert_array_equal(r[name], r2[name])


def test_filter_and_delete_paths(orc):
    """Non-bare path of the oracle (hnsw_core.hpp:1141-1148): deleted elements and a label
    filter never appear in results; the device path does not implement it yet (N3)."""
    rng = np.random.default_rng(8)
    x = rng.random((300, 16), dtype=np.float32)
    o = orc.OracleHNSW(orc.L2, 16, 300, 8, 64, 100, mode=orc.TREE64)
    o.add(x)
    o.set_ef(50)
    for lab in range(0, 300, 3):
        o.mark_delete(lab)
    assert o.deleted == 100
    r = o.search(x[:20], 5)
    assert np.all(r["label"] % 3 != 0)
    with pytest.raises(RuntimeError):
        o.mark_delete(0)
    r = o.search(x[:20], 5, filter_mod=2, filter_keep=1)
    assert np.all(r["label"] % 2 == 1) and np.all(r["label"] % 3 != 0)


def test_epsilon_stop_condition_properties(orc):
    """searchStopConditionClosest with the epsilon condition (restated from hnswlib 0.8.0: parity unpinned, see
    oracle/vdb_oracle.hpp): with epsilon = +inf it is the non-bare search at ef = max_num_candidates; with a finite
    epsilon the results are a prefix of that order cut at epsilon, never more than max, and the search does less work."""
    x = datasets.clustered(4000, 24, 5)
    q = datasets.clustered(60, 24, 6)
    o = orc.OracleHNSW(orc.L2, 24, 4000, M=12, ef_construction=100, mode=orc.TREE64)
    o.add(x)
    mx = 40
    o.set_ef(mx)
    full = o.search(q, mx, filter_mod=1, filter_keep=0)          # a filter that accepts everything: the non-bare path
    e = o.search_epsilon(q, float("inf"), 3, mx)
    np.testing.assert_array_equal(e["label"], full["label"])
    np.testing.assert_array_equal(e["dist"].view(np.uint32), full["dist"].view(np.uint32))
    np.testing.assert_array_equal(e["ndis"], full["ndis"])
    eps = float(np.median(full["dist"][:, 8]))
    e = o.search_epsilon(q, eps, mx, mx)                         # min = max: stops early only once max results are held
    assert (e["ndis"] <= full["ndis"]).all()
    for i in range(len(q)):
        c = int(e["count"][i])
        assert c <= mx and (e["dist"][i, :c] <= eps).all() and np.isinf(e["dist"][i, c:]).all()
    with pytest.raises(RuntimeError):
        o.search_epsilon(q, 1.0, 5, 4)
