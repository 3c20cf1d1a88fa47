"""The oracle against the parts of the REFERENCE that compile untouched (oracle/_ref, built
from /root/reference where it is mounted): L2Sqr, InnerProductDistance, VisitedListPool,
SearchKNNCloserFirst — plus the two known answers recorded in SURVEY.md §8c."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(orc):
    r = orc.ref_fragments()
    if r is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return r


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_known_answers(orc):
    a = np.array([1, 2, 3, 4, 0.1], np.float32)
    assert orc.distance(orc.L2, orc.SEQ, a, np.zeros(5, np.float32)) == np.float32(30.0100002)
    assert orc.distance(orc.IP, orc.SEQ, np.array([1, 2, 3], np.float32),
                        np.array([.5, .25, .125], np.float32)) == np.float32(-0.375)


@pytest.mark.parametrize("dim", [1, 3, 16, 96, 128, 131, 960])
def test_seq_distances_equal_reference_bitwise(orc, ref, dim):
    rng = np.random.default_rng(dim)
    for _ in range(50):
        a = rng.standard_normal(dim).astype(np.float32)
        b = rng.standard_normal(dim).astype(np.float32)
        assert np.float32(orc.distance(orc.L2, orc.SEQ, a, b)).view(np.uint32) == \
            np.float32(ref.ref_l2sqr(_p(a), _p(b), dim)).view(np.uint32)
        assert np.float32(orc.distance(orc.IP, orc.SEQ, a, b)).view(np.uint32) == \
            np.float32(ref.ref_ipdist(_p(a), _p(b), dim)).view(np.uint32)


@pytest.mark.parametrize("dim", [16, 96, 128, 200, 960])
def test_tree64_close_to_seq(orc, dim):
    rng = np.random.default_rng(dim + 1)
    for _ in range(50):
        a = rng.random(dim).astype(np.float32)
        b = rng.random(dim).astype(np.float32)
        s, t = orc.distance(orc.L2, orc.SEQ, a, b), orc.distance(orc.L2, orc.TREE64, a, b)
        assert abs(s - t) <= 1e-5 * abs(s)           # north star: 1e-5 relative for fp32
        exact = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum())
        assert abs(t - exact) <= 1e-5 * exact


def test_visited_pool_tags(orc, ref):
    n, rounds = 37, 70000          # crosses the u16 wrap (visited_list_pool.hpp:47-53)
    ours = (C.c_ushort * rounds)()
    ours_clear = (C.c_int * rounds)()
    theirs = (C.c_ushort * rounds)()
    theirs_clear = (C.c_int * rounds)()
    orc.lib().orc_visited_tags(n, rounds, ours, ours_clear)
    ref.ref_visited_tags(n, rounds, theirs, theirs_clear)
    assert list(ours) == list(theirs)
    assert list(ours_clear) == list(theirs_clear)
    assert ours[0] == 1 and all(ours_clear)


def test_closer_first_order(orc, ref):
    """The reference's result heap orders by (dist, label): ties come out by label."""
    d = np.array([0.5, 0.25, 0.5, 0.125, 0.5], np.float32)
    l = np.array([9, 3, 2, 7, 5], np.uint64)
    od = np.empty(5, np.float32)
    ol = np.empty(5, np.uint64)
    ref.ref_closer_first(_p(d), l.ctypes.data_as(C.POINTER(C.c_size_t)), 5, _p(od),
                         ol.ctypes.data_as(C.POINTER(C.c_size_t)))
    assert od.tolist() == [0.125, 0.25, 0.5, 0.5, 0.5]
    assert ol.tolist() == [7, 3, 2, 5, 9]
