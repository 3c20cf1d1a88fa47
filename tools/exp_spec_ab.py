#!/usr/bin/env python3
"""A/B harness for a kernel switch read from the environment once per process (e.g. VDB_SPEC): builds the C2-shaped graph
deterministically (device build, one host thread), searches at several ef, prints time per batch and writes ids /
distances / counters to gpurun_out/ab_<tag>.npz; `python tools/exp_spec_ab.py compare a b` checks two such files
for bit equality.  Not part of the product or of bench.py."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "compare":
    a, b = (np.load(f"gpurun_out/ab_{t}.npz") for t in sys.argv[2:4])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("equal" if not bad else f"DIFFERENT: {bad}")
    sys.exit(1 if bad else 0)

import torch
from vectordb_amd import datasets, index as vi
tag = sys.argv[1]
n = int(os.environ.get("EXP_N", 1_000_000)); dim = 128
law = datasets.ClusteredLaw(dim, latent=32)
base = law.sample(n, seed=1)
h = vi.HierarchicalNSW(vi.L2Space(dim), n, 16, 200, 100)
t0 = time.time()
h.add_items_device(base, n_threads=1)
print("build", round(time.time() - t0, 1), flush=True)
out = {}
k = int(os.environ.get("EXP_K", 10))
for nq in [int(x) for x in os.environ.get("EXP_NQ", "10000").split(",")]:
    q = torch.from_numpy(law.sample(nq, seed=2)).cuda()
    for ef in [int(x) for x in os.environ.get("EXP_EF", "32,100,128,200,256,400").split(",")]:
        h.setefSearch(ef)
        for _ in range(3): d, l, st = h.search_batch(q, k, stats=True)
        torch.cuda.synchronize()
        stn = st.cpu().numpy().view(np.uint32).astype(np.int64)
        alg = (stn[:, 0] * 512 + stn[:, 1] * 132 + stn[:, 2] * 68).sum() + nq * (512 + 12 * k)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(int(os.environ.get("EXP_REPS", 10)))]
        for a, b in ev:
            a.record(); h.search_batch(q, k); b.record()
        torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in ev])
        print(f"{tag}: nq={nq} ef={ef} ms={ms:.3f} qps={nq / ms * 1e3:.3e} GB/s={alg / ms / 1e6:.0f} frac={alg / ms / 1e6 / 8000:.3f} ndis={stn[:, 0].mean():.0f} hops={stn[:, 1].mean():.1f}", flush=True)
        if nq <= 10000:
            out[f"l_{nq}_{ef}"] = l.cpu().numpy(); out[f"d_{nq}_{ef}"] = d.cpu().numpy().view(np.uint32); out[f"s_{nq}_{ef}"] = stn
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/ab_{tag}.npz", **out)
