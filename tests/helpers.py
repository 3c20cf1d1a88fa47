"""Shared helpers of the parity tests: build one graph with the product's host builder, hand
the saved file to the oracle, and compare device search results with the oracle's."""
from __future__ import annotations

import os
import tempfile

import numpy as np

from vectordb_amd import index as vi


def build_pair(orc, x, metric=vi.L2, M=16, efc=200, seed=100, labels=None, n_threads=1):
    """-> (product index, oracle index over the SAME graph bytes, tree64 distances)."""
    n, dim = x.shape
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    h = vi.HierarchicalNSW(space, max(n, 1), M, efc, seed)
    if n:
        h.add_items(x, labels, n_threads=n_threads)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "g.bin")
        h.saveIndex(path)
        o = orc.OracleHNSW.load(path, metric, dim, mode=orc.TREE64)
    return h, o


def assert_same_results(dev, ref, stats=None):
    d_dist, d_label = dev
    np.testing.assert_array_equal(d_label, ref["label"].astype(np.uint64))
    # distances bit for bit (both sides sum in the tree64 order)
    np.testing.assert_array_equal(d_dist.view(np.uint32), ref["dist"].view(np.uint32))
    if stats is not None:
        np.testing.assert_array_equal(stats[:, 0].astype(np.uint64), ref["ndis"])
        np.testing.assert_array_equal(stats[:, 1].astype(np.uint64), ref["nhops0"])
        np.testing.assert_array_equal(stats[:, 2].astype(np.uint64), ref["nhops_upper"])
        np.testing.assert_array_equal(stats[:, 3], ref["count"])
