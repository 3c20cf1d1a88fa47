#!/usr/bin/env python3
"""A/B of bf_gemm_filter variants (VDB_GEMM_VARIANT_LIVE, bf_mfma.hip) on the C3 workload in ONE process,
interleaved rounds (cdna_hip_programming.md §5.4 rule 24): ms per 10 000-query batch, bf16_rerank and fp32,
and that every variant returns the same answers."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectordb_amd import datasets, index as vi

n = int(os.environ.get("EXP_N", 1_000_000)); dim, nq, k = 960, 10_000, 10
variants = [int(v) for v in os.environ.get("EXP_VARIANTS", "0,1,2,3").split(",")]
rounds = int(os.environ.get("EXP_ROUNDS", 3)); steps = int(os.environ.get("EXP_STEPS", 4))
law = datasets.ClusteredLaw(dim, latent=64)
t0 = time.time(); base = law.sample(n, seed=1); q = law.sample(nq, seed=2)
bf = vi.BruteforceSearch(vi.L2Space(dim), n); bf.add_items(base); bf.to_device(0)
dq = torch.from_numpy(q).cuda()
print(f"setup {time.time() - t0:.1f}s", flush=True)

def run(prec, steps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); out = bf.search_batch(dq, k, precision=prec); b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev], out

res = {v: {"bf16": [], "fp32": []} for v in variants}
ref = None
for r in range(rounds):
    for v in variants:
        pass
        run(vi.BF_BF16_RERANK, 1)
        ms, (d16, l16) = run(vi.BF_BF16_RERANK, steps)
        res[v]["bf16"] += ms
        if r == 0:
            ms32, (d32, l32) = run(vi.BF_FP32, 2)
            res[v]["fp32"] += ms32[1:]
            l32 = l32.cpu().numpy(); d32 = d32.cpu().numpy()
            if ref is None: ref = (l32, d32)
            same = bool(np.array_equal(l32, ref[0]) and np.array_equal(d32.view(np.uint32), ref[1].view(np.uint32)))
            rec = datasets.recall_at_k(l16.cpu().numpy().view(np.uint64), ref[0].view(np.uint64))
            print(f"variant {v}: fp32 answers identical to variant {variants[0]}: {same}; bf16_rerank recall {rec:.4f}", flush=True)
flops = 2.0 * nq * n * dim
out = {}
for v in variants:
    b = float(np.median(res[v]["bf16"])); f = float(np.median(res[v]["fp32"]))
    out[v] = {"bf16_ms_median": round(b, 3), "bf16_ms_min": round(min(res[v]["bf16"]), 3), "bf16_tflops": round(flops / b / 1e9, 1),
              "fp32_ms": round(f, 2), "fp32_tflops": round(flops / f / 1e9, 1)}
    print(v, out[v], flush=True)
print(json.dumps(out))
