#!/usr/bin/env python3
"""BASELINE.json configs[3] (C4) at full size on ONE GPU: 10M x 96, rows L2-normalised, metric IP,
8 shards x 1.25M contiguous rows (each an independent graph, seed 100+s, labels = global row ids),
10 000 queries, k=10.  The shards are built on the device, searched one after the other with the same
query batch, and merged by the merge_topk kernel: the multi-GPU pipeline of sharding.py with the
all-gather replaced by "the lists are already here".  Recall is against exact brute force over all
10M rows.  Not part of bench.py (its contract is the C2 line); results go to profiles/."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import datasets, sharding, index as vi

n = int(os.environ.get("EXP_N", 10_000_000)); dim, G, nq, k = 96, 8, 10_000, 10
law = datasets.ClusteredLaw(dim, latent=32, normalize=True)
t0 = time.time(); base = law.sample(n, seed=1); q = law.sample(nq, seed=2)
print(f"data {base.shape} in {time.time() - t0:.1f}s", flush=True)
dq = torch.from_numpy(q).cuda()

t0 = time.time()
bf = vi.BruteforceSearch(vi.InnerProductSpace(dim), n); bf.add_items(base); bf.to_device(0)
_, truth = bf.search_batch(dq, k); torch.cuda.synchronize(); truth = truth.cpu().numpy().view(np.uint64); del bf
print(f"exact ground truth over {n} rows in {time.time() - t0:.1f}s", flush=True)

shards, t0 = [], time.time()
for s in range(G):
    shards.append(sharding.build_shard(base, G, s, vi.IP, 16, 200, n_threads=16, device_build=0))
build_s = time.time() - t0
print(f"{G} shards built on the device in {build_s:.1f}s ({n / build_s:.0f} points/s)", flush=True)

out = {"config": f"C4: {n}x{dim} normalised, IP, {G} shards, M=16, efC=200, {nq} queries, k={k}", "build_seconds_device": round(build_s, 1), "ef": {}}
for ef in [int(v) for v in os.environ.get("EXP_EFS", "32,48,64,96").split(",")]:
    for h in shards: h.setefSearch(ef)
    def step():
        ds, ls = zip(*(h.search_batch(dq, k) for h in shards))
        return vi.merge_topk(torch.stack(ds), torch.stack(ls))
    md, ml = step(); torch.cuda.synchronize()
    rec = datasets.recall_at_k(ml.cpu().numpy().view(np.uint64), truth)
    for _ in range(2): step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    out["ef"][ef] = {"recall_at_10": round(rec, 4), "ms_all_shards_plus_merge": round(ms, 3), "queries_per_s_one_gpu_serial_shards": round(nq / ms * 1e3)}
    print(f"ef={ef}: recall@10 {rec:.4f}, {ms:.2f} ms for {G} shard searches + merge ({nq / ms * 1e3:.3e} q/s on one GPU)", flush=True)
print(json.dumps(out))
