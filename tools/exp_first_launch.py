#!/usr/bin/env python3
"""How long the FIRST launches at a new ef take on a large graph (development tool): 20 M x 128 (EXP_N), k = 100, launches 1, 2, 3
and 6 at each ef of a rising and of a scattered sequence, each synchronised.  The library starts a new ef from what the nearest
known ef values taught (capi.cpp, search_device)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import datasets, index as vi
n = int(os.environ.get("EXP_N", 20_000_000)); dim, nq, k = 128, 10_000, 100
law = datasets.ClusteredLaw(dim, latent=32)
base = np.concatenate([law.sample(min(10_000_000, n - lo), seed=1000 + lo // 10_000_000) for lo in range(0, n, 10_000_000)])
dq = torch.from_numpy(law.sample(nq, seed=2)).cuda()
h = vi.HierarchicalNSW(vi.L2Space(dim), n, 16, 200, 100)
h.add_items_device(base, n_threads=16)
print("built", flush=True)
for ef in [int(v) for v in os.environ.get("EXP_EFS", "100,128,200,300,160,252").split(",")]:
    h.setefSearch(ef)
    ms = []
    for i in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); h.search_batch(dq, k); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    print(f"ef {ef}: launch 1 / 2 / 3 / 6: {ms[0]:.2f} / {ms[1]:.2f} / {ms[2]:.2f} / {ms[5]:.2f} ms", flush=True)
