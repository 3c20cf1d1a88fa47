#!/usr/bin/env python3
"""One filtered configuration on the C2 graph for counter passes (development tool): EXP_SEL (default 0.1), EXP_EF (128),
EXP_REPS (6) launches of 10 000 queries after the adaptive state has settled; prints the median time."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import datasets, index as vi
n, dim, nq, k = 1_000_000, 128, 10_000, 10
sel, ef, reps = float(os.environ.get("EXP_SEL", 0.1)), int(os.environ.get("EXP_EF", 128)), int(os.environ.get("EXP_REPS", 6))
law = datasets.ClusteredLaw(dim, latent=32)
base = law.sample(n, seed=1); q = law.sample(nq, seed=2); dq = torch.from_numpy(q).cuda()
h = vi.HierarchicalNSW(vi.L2Space(dim), n, 16, 200, 100); h.add_items_device(base, n_threads=16)
allow = np.random.default_rng(7).random(n) < sel
bitmap = torch.frombuffer(bytearray(h.filter_bitmap(allow)), dtype=torch.int32).cuda()
h.setefSearch(ef)
for _ in range(6):
    h.search_batch(dq, k, filter=bitmap); torch.cuda.synchronize()
print("SETTLED", flush=True)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record(); h.search_batch(dq, k, filter=bitmap); b.record()
torch.cuda.synchronize()
print(f"filter keeps {sel} ef={ef}: {float(np.median([a.elapsed_time(b) for a, b in ev])):.3f} ms per batch", flush=True)
