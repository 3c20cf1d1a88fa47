#!/usr/bin/env python3
"""Parity sweep of the filtered search against the oracle at dim 96 (what a `make FAST=1` build can run): filters of
selectivity 1/2..1/300, deleted elements, a deleted entry point, the epsilon stop condition; four launches each so that the
adaptive state (candidate-buffer size, spill tables) moves under the comparison.  Development tool, not collected by pytest."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from tests.helpers import assert_same_results, build_pair
from vectordb_amd import datasets, index as vi
orc.build()
n, dim = int(os.environ.get("CHK_N", 20000)), 96
x = datasets.clustered(n, dim, 71); q = datasets.clustered(400, dim, 72)
for metric in (vi.L2, vi.IP):
    h, o = build_pair(orc, x, metric=metric, n_threads=8)
    for lab in range(0, n, 9):
        h.markDelete(lab); o.mark_delete(lab)
    ep_label = int(h.labels()[h.entry])
    if ep_label % 9:
        h.markDelete(ep_label); o.mark_delete(ep_label)
    h.to_device(0)
    for mod in (1, 2, 4, 10, 30, 100, 300):
        for ef, k in ((10, 10), (32, 10), (64, 10), (128, 10), (200, 100), (400, 10)):
            h.setefSearch(ef); o.set_ef(ef)
            r = o.search(q, k, filter_mod=mod, filter_keep=mod - 1, n_threads=8)
            for rep in range(3):
                d, l, st = h.search_batch(q, k, stats=True, filter=(lambda lab: lab % mod == mod - 1))
                assert_same_results((d, l), r, st)
            print(f"metric {metric} mod {mod} ef {ef} k {k}: equal (mean found {r['count'].mean():.1f}, ndis {r['ndis'].mean():.0f})", flush=True)
        eps = float(np.median(r["dist"][:, 0]) * 1.5) if np.isfinite(r["dist"][:, 0]).any() else 1.0
        for mn in (2, 20):
            de, le, cnt, ste = h.search_stop_condition_closest_batch(q, eps, mn, 64, filter=(lambda lab: lab % mod == mod - 1))
            re = o.search_epsilon(q, eps, mn, 64, filter_mod=mod, filter_keep=mod - 1)
            np.testing.assert_array_equal(cnt, re["count"])
            np.testing.assert_array_equal(le, re["label"].astype(np.uint64))
            np.testing.assert_array_equal(de.view(np.uint32), re["dist"].view(np.uint32))
            np.testing.assert_array_equal(ste[:, 0], re["ndis"].astype(np.uint32))
        print(f"metric {metric} mod {mod}: epsilon searches equal", flush=True)
print("ALL EQUAL")
