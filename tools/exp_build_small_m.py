#!/usr/bin/env python3
"""Recall of the device build against the host build at small M on small sets, many data seeds (development tool):
python tools/exp_build_small_m.py [seeds]   - prints the deficit distribution; VDB_BUILD_BATCH_DIV / VDB_BUILD_NO_REPAIR
switch the builder's batch fraction and orphan pass in a diagnostic build of the library (make DIAG=1)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import datasets, index as vi
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
def recall(h, x, q, metric):
    space = vi.L2Space(x.shape[1]) if metric == vi.L2 else vi.InnerProductSpace(x.shape[1])
    b = vi.BruteforceSearch(space, len(x)); b.add_items(x); b.to_device(0)
    _, truth = b.search_batch(q, 10)
    h.setefSearch(64)
    _, l = h.search_batch(q, 10)
    return datasets.recall_at_k(l, truth)
shapes = ((2000, 128, 4, 40, vi.L2), (2000, 96, 4, 100, vi.IP), (9000, 128, 4, 40, vi.L2), (2000, 32, 6, 40, vi.L2))
if os.environ.get('EXP_BIG'): shapes = ((80000, 96, 4, 100, vi.L2), (30000, 128, 4, 40, vi.L2), (80000, 32, 6, 100, vi.IP))
for n, dim, M, efc, metric in shapes:
    defs, orph = [], []
    for s in range(seeds):
        x = datasets.clustered(n, dim, 5000 + s, normalize=(metric == vi.IP)); q = datasets.clustered(300, dim, 6000 + s, normalize=(metric == vi.IP))
        space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
        host = vi.HierarchicalNSW(space, n, M, efc, 100); host.add_items(x, n_threads=16); host.to_device(0)
        dev = vi.HierarchicalNSW(space, n, M, efc, 100); dev.add_items_device(x, n_threads=16, device=0)
        defs.append(recall(host, x, q, metric) - recall(dev, x, q, metric)); orph.append(dev.checkIntegrity()[0])
    defs = np.array(defs)
    print(f"n={n} dim={dim} M={M} efc={efc} metric={metric}: deficit mean {defs.mean():+.4f} max {defs.max():+.4f} p90 {np.quantile(defs, 0.9):+.4f}, "
          f"beyond 0.02: {(defs > 0.02).sum()}/{seeds}, integrity codes {sorted(set(orph))}", flush=True)
