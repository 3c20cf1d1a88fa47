#!/usr/bin/env python3
"""Randomised parity sweep (not collected by pytest; run by hand on a GPU box):
    python tests/fuzz_parity.py [seconds] [seed]
Random graphs (size, dim, M, metric, quantised coordinates for exact distance ties, duplicated rows), random ef / k,
bare and filtered, one epsilon stop-condition search per graph, several launches per index so that the adaptive state (visited-table size, spill tables, key-array
level) moves under the comparison.  Every launch must equal the oracle bit for bit: ids, distances, counters."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc          # the checker
from tests.helpers import assert_same_results, build_pair
from vectordb_amd import index as vi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
orc.build()
t0, cases, launches = time.time(), 0, 0
while time.time() - t0 < budget:
    n = int(rng.choice([300, 1500, 6000, 20000]))
    dim = int(rng.choice([8, 16, 24, 32, 48, 96, 128, 200]))
    M = int(rng.choice([4, 8, 16, 24]))
    metric = vi.L2 if rng.random() < 0.6 else vi.IP
    x = rng.standard_normal((n, dim)).astype(np.float32)
    style = rng.integers(0, 3)
    if style == 1:   # few distinct coordinate values: many exact ties
        x = np.round(x * 2) / 2
    elif style == 2:  # blocks of identical rows
        reps = int(rng.choice([2, 5, 40]))
        x = np.repeat(x[: max(n // reps, 1)], reps, axis=0)[:n]
        x = x[rng.permutation(x.shape[0])]
    if metric == vi.IP:
        x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-6)
    n = x.shape[0]
    q = x[rng.integers(0, n, 80)] + (0.05 * rng.standard_normal((80, dim))).astype(np.float32) * (rng.random() < 0.7)
    q = np.ascontiguousarray(q, np.float32)
    h, o = build_pair(orc, x, metric, M, int(rng.choice([32, 100, 200])), n_threads=4)
    h.to_device(0)
    h.set_spill_mode(int(rng.choice([-1, -1, 0, 1])))   # adaptive / never / always second-level visited tables
    for _ in range(4):
        ef = int(rng.choice([1, 7, 32, 60, 61, 100, 124, 125, 160, 188, 189, 252, 253, 300, 400]))
        k = int(min(ef, rng.choice([1, 5, 10, 100])))
        mod = int(rng.choice([0, 0, 2, 3, 7]))
        h.setefSearch(ef); o.set_ef(ef)
        ref = o.search(q, k, filter_mod=mod, filter_keep=1 % max(mod, 1)) if mod else o.search(q, k)
        flt = (lambda lab, m=mod: lab % m == 1 % m) if mod else None
        for _ in range(3):
            d, l, st = h.search_batch(q, k, stats=True, filter=flt)
            ok = st[:, 3] != 0xFFFFFFFF   # (hundreds of keys AT the bound's distance: flagged, never truncated; see the tests)
            assert_same_results((d[ok], l[ok]), {key: v[ok] for key, v in ref.items() if key != "seconds"}, st[ok])
            launches += 1
    # the stop-condition search (epsilon radius): a radius drawn from the distances just seen
    if np.isfinite(ref["dist"]).any():
        eps = float(rng.choice(ref["dist"][np.isfinite(ref["dist"])]))
        mx = int(rng.choice([1, 16, 64, 200])); mn = int(rng.integers(1, mx + 1))
        mod = int(rng.choice([0, 0, 3]))
        r = o.search_epsilon(q, eps, mn, mx, filter_mod=mod, filter_keep=1 % max(mod, 1)) if mod else o.search_epsilon(q, eps, mn, mx)
        d, l, cnt, st = h.search_stop_condition_closest_batch(q, eps, mn, mx, filter=(lambda lab, m=mod: lab % m == 1 % m) if mod else None)
        np.testing.assert_array_equal(cnt, r["count"])
        np.testing.assert_array_equal(l, r["label"].astype(np.uint64))
        np.testing.assert_array_equal(d.view(np.uint32), r["dist"].view(np.uint32))
        np.testing.assert_array_equal(st[:, 0], r["ndis"].astype(np.uint32))
        np.testing.assert_array_equal(st[:, 1], r["nhops0"].astype(np.uint32))
        launches += 1
    cases += 1
    if cases % 5 == 0:
        print(f"{cases} graphs, {launches} launches equal to the oracle, {time.time() - t0:.0f}s", flush=True)
print(f"done: {cases} graphs, {launches} launches, all equal to the oracle")
