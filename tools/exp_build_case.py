#!/usr/bin/env python3
"""One case of tests/fuzz_build.py again, several builds of each kind (development tool):
python tools/exp_build_case.py n dim M efc metric host_base data_seed query_seed [builds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import datasets, index as vi
n, dim, M, efc, metric, hb, s1, s2 = (int(v) for v in sys.argv[1:9]); builds = int(sys.argv[9]) if len(sys.argv) > 9 else 6
os.environ["VDB_BUILD_HOST_BASE"] = str(hb)
x = datasets.clustered(n, dim, s1, normalize=(metric == vi.IP)); q = datasets.clustered(300, dim, s2, normalize=(metric == vi.IP))
space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
b = vi.BruteforceSearch(space, n); b.add_items(x); b.to_device(0)
_, truth = b.search_batch(q, 10)
def rec(h):
    out = []
    for ef in (64, 200):
        h.setefSearch(ef); _, l = h.search_batch(q, 10); out.append(round(datasets.recall_at_k(l, truth), 4))
    return out
for kind in ("host t16", "host t1", "device t16", "device t1"):
    res = []
    for _ in range(builds):
        h = vi.HierarchicalNSW(space, n, M, efc, 100)
        t = 16 if kind.endswith("16") else 1
        if kind.startswith("host"): h.add_items(x, n_threads=t); h.to_device(0)
        else: h.add_items_device(x, n_threads=t, device=0)
        res.append(rec(h) + [h.checkIntegrity()[0]])
    print(kind, res, flush=True)
