#!/usr/bin/env python3
"""Tuning harness (not part of the product or of bench.py): one 1M x 128 graph, several batch
sizes / ef values, HIP-event time per launch and algorithmic GB/s."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import datasets, index as vi

n = int(os.environ.get("EXP_N", 1_000_000)); dim = 128
law = datasets.ClusteredLaw(dim, latent=32)
base = law.sample(n, seed=1)
h = vi.HierarchicalNSW(vi.L2Space(dim), n, 16, 200, 100)
t0 = time.time()
if os.environ.get("EXP_BUILD", "host") == "device":
    h.add_items_device(base, n_threads=16)
else:
    h.add_items(base, n_threads=16)
print("build", round(time.time() - t0, 1), flush=True)
h.to_device(0)
if os.environ.get("EXP_SPILL"):
    h.set_spill_mode(int(os.environ["EXP_SPILL"]))
for nq in [int(x) for x in os.environ.get("EXP_NQ", "10000,20000,40000,100000").split(",")]:
    q = torch.from_numpy(law.sample(nq, seed=2)).cuda()
    for ef in [int(x) for x in os.environ.get("EXP_EF", "32,64").split(",")]:
        h.setefSearch(ef)
        d, l, st = h.search_batch(q, 10, stats=True)
        torch.cuda.synchronize()
        st = st.cpu().numpy().view(np.uint32).astype(np.int64)
        alg = (st[:, 0] * 512 + st[:, 1] * 132 + st[:, 2] * 68).sum() + nq * (512 + 120)
        for _ in range(3): h.search_batch(q, 10)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(int(os.environ.get("EXP_REPS", 10)))]
        for a, b in ev:
            a.record(); h.search_batch(q, 10); b.record()
        torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in ev])
        print(f"nq={nq} ef={ef} ms={ms:.3f} qps={nq / ms * 1e3:.3e} GB/s={alg / ms / 1e6:.0f} ndis={st[:, 0].mean():.0f} hops={st[:, 1].mean():.1f}", flush=True)
