#!/usr/bin/env python3
"""Randomised parity sweep of the brute-force paths (not collected by pytest; run by hand on a GPU box):
    python tests/fuzz_parity_bf.py [seconds] [seed]
Random base sets (ragged sizes, eight dimensions up to 960, exact ties, duplicated rows), random k and batch sizes.
The fp32 path (exact scan or MFMA candidate pass + exact re-rank, whichever the library picks) must equal the oracle bit
for bit; the bf16 candidate pass + exact re-rank must return the same distances wherever it returns the same labels,
with recall >= 0.99 against them."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc          # the checker
from vectordb_amd import datasets, index as vi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
orc.build()
t0, cases, filtered = time.time(), 0, 0
while time.time() - t0 < budget:
    dim = int(rng.choice([8, 40, 96, 128, 200, 512, 960]))
    nq = int(rng.choice([5, 37, 150, 700]))
    n = int(min(rng.choice([50, 700, 4097, 9000, 40000]), max(50, 2_000_000_000 // (dim * nq))))
    k = int(rng.choice([1, 10, 33, 100]))
    metric = vi.L2 if rng.random() < 0.6 else vi.IP
    x = rng.standard_normal((n, dim)).astype(np.float32)
    style = rng.integers(0, 3)
    if style == 1:
        x = np.round(x * 2) / 2
    elif style == 2:
        reps = int(rng.choice([2, 9, 60]))
        x = np.repeat(x[: max(n // reps, 1)], reps, axis=0)[:n]
        x = x[rng.permutation(x.shape[0])]
    if metric == vi.IP:
        x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-6)
    n = x.shape[0]
    q = np.ascontiguousarray(x[rng.integers(0, n, nq)] + (0.05 * rng.standard_normal((nq, dim))).astype(np.float32) * (rng.random() < 0.7), np.float32)
    labels = (30_000_000_000 - np.arange(n) * 3).astype(np.uint64)
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    b = vi.BruteforceSearch(space, n)
    b.add_items(x, labels)
    b.to_device(0)
    ob = orc.OracleBruteforce(metric, dim, n, mode=orc.TREE64)
    ob.add(x, labels)
    r = ob.search(q, k)
    d, l = b.search_batch(q, k, precision=vi.BF_FP32)
    np.testing.assert_array_equal(l, r["label"].astype(np.uint64))
    np.testing.assert_array_equal(d.view(np.uint32), r["dist"].view(np.uint32))
    d2, l2 = b.search_batch(q, k, precision=vi.BF_BF16_RERANK)
    same = l2 == l
    np.testing.assert_array_equal(d2.view(np.uint32)[same], d.view(np.uint32)[same])
    pad = np.uint64(0xFFFFFFFFFFFFFFFF)   # (k > n: padded rows; they are no neighbours)
    hit = sum(len(set(a[a != pad]) & set(c[c != pad])) for a, c in zip(l2, l))
    assert hit >= 0.99 * int((l != pad).sum()), (n, dim, k, metric, style)
    np.testing.assert_array_equal(l2 == pad, l == pad)
    if rng.random() < 0.5:   # with a filter functor (bruteforce.hpp:158-173): label % 7 == 2
        rf = ob.search(q, k, filter_mod=7, filter_keep=2)
        df, lf = b.search_batch(q, k, precision=vi.BF_FP32, filter=lambda lab: lab % 7 == 2)
        np.testing.assert_array_equal(lf, rf["label"].astype(np.uint64))
        np.testing.assert_array_equal(df.view(np.uint32), rf["dist"].view(np.uint32))
        filtered += 1
    cases += 1
    if cases % 10 == 0:
        print(f"{cases} base sets equal to the oracle, {time.time() - t0:.0f}s", flush=True)
print(f"done: {cases} base sets ({filtered} also with a filter), fp32 equal to the oracle bit for bit, bf16 + re-rank recall >= 0.99")
