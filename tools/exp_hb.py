#!/usr/bin/env python3
"""Experiment: visited-table size (VDB_HB) x ef on an EXP_N-row graph."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import datasets, index as vi
n = int(os.environ.get("EXP_N", 20_000_000)); dim, nq, k = 128, 10_000, 100
law = datasets.ClusteredLaw(dim, latent=32)
base = np.empty((n, dim), np.float32)
for lo in range(0, n, 10_000_000):
    m = min(10_000_000, n - lo); base[lo:lo + m] = law.sample(m, seed=1000 + lo // 10_000_000); print("data", lo + m, flush=True)
dq = torch.from_numpy(law.sample(nq, seed=2)).cuda()
h = vi.HierarchicalNSW(vi.L2Space(dim), n, 16, 200, 100)
import threading
t0 = time.time(); stop = threading.Event()
def beat():
    while not stop.wait(30): print(f"  building ... {time.time() - t0:.0f}s", flush=True)
threading.Thread(target=beat, daemon=True).start()
h.add_items_device(base, n_threads=16); stop.set(); print(f"built {time.time()-t0:.0f}s", flush=True)
h.set_spill_mode(1)
for ef in [int(v) for v in os.environ.get("EXP_EFS", "100,128,200").split(",")]:
    h.setefSearch(ef)
    for hb in [int(v) for v in os.environ.get("EXP_HBS", "11,12,13").split(",")]:
        os.environ["VDB_HB"] = str(hb)
        for _ in range(3): h.search_batch(dq, k); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record(); h.search_batch(dq, k); b.record()
        torch.cuda.synchronize()
        print(f"ef={ef} hb={hb} ms={np.median([a.elapsed_time(b) for a, b in ev]):.2f}", flush=True)
