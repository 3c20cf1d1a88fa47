#!/usr/bin/env python3
"""Check: searches through spill tables (recycled) == searches re-run by the large configuration, on a big graph."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import datasets, index as vi
n = int(os.environ.get("EXP_N", 5_000_000)); dim, nq, k = 128, 10_000, 100
law = datasets.ClusteredLaw(dim, latent=32)
base = law.sample(n, seed=1000)
dq = torch.from_numpy(law.sample(nq, seed=2)).cuda()
h = vi.HierarchicalNSW(vi.L2Space(dim), n, 16, 200, 100)
t0 = time.time(); h.add_items_device(base, n_threads=16); print(f"built {time.time()-t0:.0f}s", flush=True)
for ef in (100, 128, 200, 300, 508):
    h.setefSearch(ef)
    res = {}
    for mode in (1, 0, -1, -1, -1):
        h.set_spill_mode(mode)
        d, l, st = h.search_batch(dq, k, stats=True); torch.cuda.synchronize()
        cur = (d.cpu().numpy().view(np.uint32), l.cpu().numpy(), st.cpu().numpy())
        if not res: res = cur
        same = all(np.array_equal(a, b) for a, b in zip(res, cur))
        print(f"ef={ef} mode={mode} same_as_first={same} ndis={cur[2].view(np.uint32)[:,0].mean():.0f}", flush=True)
