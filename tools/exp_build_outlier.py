#!/usr/bin/env python3
"""Many device builds of one case, looking for a rare bad graph (development tool): any build whose recall@10 at
ef 64 falls 0.04 below the median is saved under gpurun_out/ with the data seeds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import datasets, index as vi
n, dim, M, efc, metric, hb, s1, s2 = (int(v) for v in sys.argv[1:9]); builds = int(sys.argv[9]) if len(sys.argv) > 9 else 100
os.environ["VDB_BUILD_HOST_BASE"] = str(hb)
x = datasets.clustered(n, dim, s1, normalize=(metric == vi.IP)); q = datasets.clustered(300, dim, s2, normalize=(metric == vi.IP))
space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
b = vi.BruteforceSearch(space, n); b.add_items(x); b.to_device(0)
_, truth = b.search_batch(q, 10)
recs, saved = [], 0
for i in range(builds):
    h = vi.HierarchicalNSW(space, n, M, efc, 100)
    h.add_items_device(x, n_threads=int(os.environ.get("EXP_THREADS", 16)), device=0)
    h.setefSearch(64); _, l = h.search_batch(q, 10); r = datasets.recall_at_k(l, truth)
    recs.append(r)
    if i < 2: h.saveIndex(f"gpurun_out/normal_{i}.bin")
    if r < 0.88 and saved < 4:
        saved += 1
        h.saveIndex(f"gpurun_out/outlier_{i}.bin"); print(f"build {i}: recall {r:.4f}, max_level {h.max_level}, integrity {h.checkIntegrity()}", flush=True)
recs = np.array(recs)
print(f"{builds} builds: recall min {recs.min():.4f} median {np.median(recs):.4f} max {recs.max():.4f}; sorted lowest five {np.sort(recs)[:5]}")
