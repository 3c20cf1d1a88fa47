#!/usr/bin/env python3
"""Diagnostic for indexes beyond 2^32 floats (33.5M rows x 128): brute force vs numpy, HNSW recall by label range."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import datasets, index as vi
n = int(os.environ.get("EXP_N", 40_000_000)); dim = 128; cut = (1 << 32) // dim
law = datasets.ClusteredLaw(dim, latent=32)
base = np.empty((n, dim), np.float32)
for lo in range(0, n, 10_000_000):
    m = min(10_000_000, n - lo); base[lo:lo + m] = law.sample(m, seed=1000 + lo // 10_000_000); print("data", lo + m, flush=True)
q = law.sample(200, seed=2); dq = torch.from_numpy(q).cuda()
bf = vi.BruteforceSearch(vi.L2Space(dim), n); bf.add_items(base); bf.to_device(0)
d, truth = bf.search_batch(dq, 10); torch.cuda.synchronize(); truth = truth.cpu().numpy().view(np.uint64); del bf
for i in range(int(os.environ.get('EXP_NUMPY_Q', 3))):
    best = np.full(10, np.inf); bl = np.zeros(10, np.int64)
    for lo in range(0, n, 4_000_000):
        dd = ((base[lo:lo + 4_000_000] - q[i]) ** 2).sum(1)
        idx = np.argpartition(dd, 10)[:10]
        alld = np.concatenate([best, dd[idx]]); alll = np.concatenate([bl, idx + lo]); o = np.argsort(alld)[:10]
        best, bl = alld[o], alll[o]
    print("query", i, "numpy top10", sorted(bl.tolist()), "bf top10", sorted(truth[i].tolist()), "same", set(bl.tolist()) == set(truth[i].tolist()), flush=True)
print("truth labels >= cut:", float((truth >= cut).mean()), "cut", cut, flush=True)
h = vi.HierarchicalNSW(vi.L2Space(dim), n, 16, 200, 100)
t0 = time.time(); done = threading.Event()
threading.Thread(target=lambda: [print("  building", flush=True) for _ in iter(lambda: done.wait(30), True)], daemon=True).start()
h.add_items_device(base, n_threads=16); done.set(); print("built", time.time() - t0, flush=True)
h.setefSearch(100)
_, l = h.search_batch(dq, 10); l = l.cpu().numpy().view(np.uint64)
hit_lo = hit_hi = tot_lo = tot_hi = 0
for f, t in zip(l, truth):
    fs = set(f.tolist())
    for x in t.tolist():
        if x >= cut: tot_hi += 1; hit_hi += x in fs
        else: tot_lo += 1; hit_lo += x in fs
print(f"recall on truth labels < cut: {hit_lo}/{tot_lo}; >= cut: {hit_hi}/{tot_hi}", flush=True)
print("hnsw q0", sorted(l[0].tolist()), flush=True)
print("truth label histogram by tenth of n:", np.histogram(truth.astype(np.int64), bins=10, range=(0, n))[0].tolist(), flush=True)
print("found label histogram by tenth of n:", np.histogram(l.astype(np.int64), bins=10, range=(0, n))[0].tolist(), flush=True)
print("found labels >= cut:", float((l >= cut).mean()), flush=True)
for i in (cut - 5, cut + 5, n - 1):
    print("links of", i, h.links(i, 0)[:8], "level", h.level_of(i), flush=True)
