"""searchStopConditionClosest (hnsw_core.hpp:1162-1213) with the epsilon-radius stop condition on the device.

GPU == oracle(tree64) bit for bit: labels, distances, result counts, distance computations and layer-0 hops.
PARITY UNPINNED by the reference for this entry point: it holds the abstract BaseSearchStopCondition only, no concrete
condition, and the layer-0 search that would call it is undefined; the oracle restates both from hnswlib 0.8.0
(oracle/vdb_oracle.hpp)."""
import numpy as np
import pytest

from vectordb_amd import datasets
from tests.helpers import build_pair

pytestmark = pytest.mark.gpu


def _check(h, o, q, eps, mn, mx, fmod=0, fkeep=0):
    filt = (lambda l: l % fmod == fkeep) if fmod else None
    d, l, cnt, st = h.search_stop_condition_closest_batch(q, eps, mn, mx, filter=filt)
    r = o.search_epsilon(q, eps, mn, mx, filter_mod=fmod, filter_keep=fkeep)
    np.testing.assert_array_equal(cnt, r["count"])
    np.testing.assert_array_equal(l, r["label"].astype(np.uint64))
    np.testing.assert_array_equal(d.view(np.uint32), r["dist"].view(np.uint32))
    np.testing.assert_array_equal(st[:, 0], r["ndis"].astype(np.uint32))
    np.testing.assert_array_equal(st[:, 1], r["nhops0"].astype(np.uint32))
    for i in range(len(q)):  # what the condition promises: within epsilon, at most max, padded behind
        assert cnt[i] <= mx and (d[i, :cnt[i]] <= eps).all() and np.isinf(d[i, cnt[i]:]).all()
    return cnt


def test_epsilon_search_matches_oracle(orc):
    n, dim = 8000, 32
    x = datasets.clustered(n, dim, 71)
    q = datasets.clustered(200, dim, 72)
    labels = np.arange(n, dtype=np.uint64) * 5 + 1
    h, o = build_pair(orc, x, labels=labels, n_threads=4)
    h.to_device(0)
    o.set_ef(64)
    ref = o.search(q, 64)["dist"]
    for quant, mn, mx in [(5, 1, 40), (20, 10, 64), (40, 64, 64), (10, 5, 200), (0, 3, 16)]:
        eps = float(np.median(ref[:, quant]))
        cnt = _check(h, o, q, eps, mn, mx)
        assert cnt.max() > 0
    _check(h, o, q, 0.0, 1, 10)              # nothing lies within 0 (queries are not base vectors)
    _check(h, o, q, float("inf"), 7, 50)     # never stops early: the non-bare search at ef = 50
    _check(h, o, q[:50], float(np.median(ref[:, 12])), 4, 48, fmod=3, fkeep=1)
    # deleted elements are traversed, never returned
    dead = labels[np.random.default_rng(3).permutation(n)[: n // 4]]
    for lab in dead:
        h.markDelete(int(lab))
        o.mark_delete(int(lab))
    h.to_device(0)
    eps = float(np.median(ref[:, 15]))
    d, l, cnt, st = h.search_stop_condition_closest_batch(q, eps, 5, 60)
    assert not np.isin(l[l != np.uint64(2**64 - 1)], dead).any()
    _check(h, o, q, eps, 5, 60)
    _check(h, o, q, eps, 5, 60, fmod=2, fkeep=0)


def test_epsilon_search_argument_errors(orc):
    from vectordb_amd.index import VdbError
    x = datasets.uniform(300, 16, 1)
    h, o = build_pair(orc, x, n_threads=2)
    h.to_device(0)
    q = datasets.uniform(4, 16, 2)
    with pytest.raises(VdbError):
        h.search_stop_condition_closest_batch(q, 1.0, 5, 4)       # min > max
    with pytest.raises(VdbError):
        h.search_stop_condition_closest_batch(q, float("nan"), 1, 4)
    with pytest.raises(VdbError):
        h.search_stop_condition_closest_batch(q, 1.0, 1, 600)     # beyond the device limit
