#!/usr/bin/env python3
"""Row N3 measured (not part of bench.py): the C2 graph (1M x 128, M=16, device build), 10 000-query batches, k=10,
searched with a filter bitmap of the given selectivity (random ids), and with a tenth of the elements deleted.
HIP-event time per launch, algorithmic GB/s from the kernel's own counters, recall against exact brute force over the
allowed rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import datasets, index as vi

n, dim, nq, k = 1_000_000, 128, 10_000, 10
law = datasets.ClusteredLaw(dim, latent=32)
base = law.sample(n, seed=1)
q = law.sample(nq, seed=2)
dq = torch.from_numpy(q).cuda()
h = vi.HierarchicalNSW(vi.L2Space(dim), n, 16, 200, 100)
h.add_items_device(base, n_threads=16)
bf = vi.BruteforceSearch(vi.L2Space(dim), n); bf.add_items(base); bf.to_device(0)
rng = np.random.default_rng(7)

def run(tag, allow_bool, efs):
    bitmap = torch.frombuffer(bytearray(h.filter_bitmap(allow_bool)), dtype=torch.int32).cuda() if allow_bool is not None else None
    _, truth = bf.search_batch(dq[:1000], k, filter=allow_bool) if allow_bool is not None else bf.search_batch(dq[:1000], k)
    truth = truth.cpu().numpy().view(np.uint64)
    for ef in efs:
        h.setefSearch(ef)
        d, l, st = h.search_batch(dq, k, stats=True, filter=bitmap); torch.cuda.synchronize()
        rec = datasets.recall_at_k(l[:1000].cpu().numpy().view(np.uint64), truth)
        st = st.cpu().numpy().view(np.uint32).astype(np.int64)
        alg = (st[:, 0] * 512 + st[:, 1] * 132 + st[:, 2] * 68).sum() + nq * (512 + 120)
        for _ in range(3):
            h.search_batch(dq, k, filter=bitmap); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); h.search_batch(dq, k, filter=bitmap); b.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        print(f"{tag} ef={ef}: recall@10 {rec:.4f}, {ms:.3f} ms per batch ({nq / ms * 1e3:.3e} q/s, {alg / ms / 1e6:.0f} GB/s algorithmic), "
              f"ndis {st[:, 0].mean():.0f}, hops {st[:, 1].mean():.1f}", flush=True)

run("no filter          ", None, (32, 64))
for sel in (0.5, 0.1):
    run(f"filter keeps {sel:<5}", rng.random(n) < sel, (32, 64, 128))
