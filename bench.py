#!/usr/bin/env python3
"""bench.py — queries/sec of the HNSW k-NN query path on MI355X (BASELINE.json metric).

Workload at any N: BASELINE.json configs[1] — 1M x 128 fp32 "SIFT-like" synthetic base set,
HNSW M=16 efConstruction=200, batches of 10 000 queries, k=10, ef = the smallest value of
the sweep {32,64,96,128,192,256} whose recall@10 against exact brute force is >= 0.95.

A "step" is one pass of the hot path (vdb_search_batch_device: greedy descent + layer-0 beam
search + top-k output) over one batch of 10 000 queries already resident in HBM.

N > 1 (one process per GPU, torch.distributed / RCCL): the independent units of the path
are the queries, so every rank holds a replica of the index and searches its own batch of
10 000 queries ("weak": per-GPU work fixed); no collective sits on the data path.  The
sharded variant of the north star (base set split over the ranks, per-rank top-k lists
all-gathered over RCCL and merged on the GPU) is timed in the same run and reported under
"sharded" — its queries/sec is by construction not additive over ranks.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EF_SWEEP = (32, 64, 96, 128, 192, 256)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main() -> None:
    if os.environ.get("VDB_BENCH_WATCHDOG"):  # diagnostics: dump every thread's stack after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VDB_BENCH_WATCHDOG"]), exit=False)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--base-rows", dest="n", type=int, default=1_000_000, help="base vectors (default = the named workload)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--queries", dest="nq", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--build-threads", type=int, default=0)
    ap.add_argument("--cpu-queries", type=int, default=2000, help="bounded sample of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true")
    ap.add_argument("--workload", choices=("hnsw", "bruteforce"), default="hnsw",
                    help="hnsw = BASELINE.json configs[1] (the headline metric); bruteforce = configs[2] "
                         "(exact k-NN 1M x 960 on the MFMA path, its own JSON line)")
    ap.add_argument("--precision", choices=("fp32", "bf16", "bf16_rerank"), default="bf16_rerank")
    ap.add_argument("--build", choices=("host", "device"), default="host",
                    help="graph builder: host = the reference algorithm on the host cores (default: the graph the headline\n"
                         "number is quoted on), device = vdb_index_add_batch_device (DESIGN.md 4.3)")
    args = ap.parse_args()
    if args.workload == "bruteforce":
        return run_bruteforce(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # rehearsal on a one-GPU box only: VDB_BENCH_REHEARSAL=1 puts every rank on device 0 over gloo
    rehearsal = os.environ.get("VDB_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from vectordb_amd import datasets
    from vectordb_amd import index as vi

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    threads = args.build_threads or int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    threads = max(1, min(threads, 16))

    # ---- synthetic data -------------------------------------------
    t0 = time.time()
    law = datasets.ClusteredLaw(args.dim, latent=32)
    base = law.sample(args.n, seed=1)
    queries = law.sample(args.nq, seed=2 + rank)          # every rank its own batch
    if rank == 0:
        log(f"data {base.shape} + {queries.shape} in {time.time() - t0:.1f}s")

    # ---- graph build on the host cores (reference algorithm), upload ------------------
    t0 = time.time()
    h = vi.HierarchicalNSW(vi.L2Space(args.dim), args.n, args.M, args.efc, 100)
    if args.build == "device":
        h.add_items_device(base, n_threads=threads, device=local_rank)
    else:
        h.add_items(base, n_threads=threads)
    build_s = time.time() - t0
    t0 = time.time()
    h.to_device(local_rank)
    if rank == 0:
        log(f"graph built in {build_s:.1f}s ({args.build} builder, {threads} host threads), uploaded {h.device_bytes() / 1e6:.0f} MB in "
            f"{time.time() - t0:.1f}s, max_level {h.max_level}")

    dq = torch.from_numpy(queries).to(dev)

    # ---- ground truth by the exact brute-force kernels --------------------------------
    t0 = time.time()
    bf = vi.BruteforceSearch(vi.L2Space(args.dim), args.n)
    bf.add_items(base)
    bf.to_device(local_rank)
    gt_d, gt_l = bf.search_batch(dq, args.k)
    torch.cuda.synchronize()
    gt = gt_l.cpu().numpy().view(np.uint64)
    del bf
    if rank == 0:
        log(f"exact ground truth in {time.time() - t0:.1f}s")

    # ---- ef sweep: smallest ef with recall@k >= 0.95 ----------------------------------
    sweep = {}
    chosen = None
    for ef in EF_SWEEP:
        if ef < args.k:
            continue
        h.setefSearch(ef)
        _, lab = h.search_batch(dq, args.k)
        torch.cuda.synchronize()
        rec = datasets.recall_at_k(lab.cpu().numpy().view(np.uint64), gt)
        sweep[ef] = round(rec, 4)
        if chosen is None and rec >= 0.95:
            chosen = ef
            break
    if chosen is None:
        chosen = max(sweep)
    if world > 1:  # all ranks time the same ef: the largest any rank needs
        t = torch.tensor([chosen], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        chosen = int(t.item())
    h.setefSearch(chosen)
    dist_out, lab, st = h.search_batch(dq, args.k, stats=True)
    torch.cuda.synchronize()
    recall = datasets.recall_at_k(lab.cpu().numpy().view(np.uint64), gt)
    st = st.cpu().numpy().view(np.uint32).astype(np.int64)
    # algorithmic bytes of one launch: vectors scored, link rows read, query, output
    maxM0, maxM = 2 * args.M, args.M
    alg_bytes = int((st[:, 0] * 4 * args.dim + st[:, 1] * 4 * (maxM0 + 1) + st[:, 2] * 4 * (maxM + 1)).sum()
                    + args.nq * (4 * args.dim + args.k * 12))
    if rank == 0:
        log(f"ef sweep {sweep} -> ef={chosen}, recall@{args.k}={recall:.4f}, mean ndis {st[:, 0].mean():.0f}, "
            f"mean hops {st[:, 1].mean():.1f}, {alg_bytes / 1e9:.2f} GB algorithmic per launch")

    # ---- timed region: W warm-up + K steps, barrier + synchronize on both sides --------
    for _ in range(args.warmup):
        h.search_batch(dq, args.k)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()          # same stream the kernels are enqueued on (torch's current stream)
        h.search_batch(dq, args.k)
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    qps = world * args.nq * args.steps / elapsed

    # ---- sharded variant (north star): base split over ranks + RCCL all-gather + merge --
    sharded = None
    if world > 1 and not args.no_sharded:
        from vectordb_amd import sharding
        per = (args.n + world - 1) // world
        log(f"replicas: {qps:.3e} queries/s over {world} ranks; building the {per}-row shards for the sharded variant")
        t0 = time.time()
        hs = sharding.build_shard(base, world, rank, vi.L2, args.M, args.efc, n_threads=threads)
        hs.to_device(local_rank)
        log(f"shard graph built in {time.time() - t0:.1f}s")
        hs.setefSearch(chosen)
        shared_q = torch.from_numpy(law.sample(args.nq, seed=2)).to(dev)   # the SAME batch on every rank

        def sharded_step():
            return sharding.sharded_search(hs, shared_q, args.k)

        md, ml = sharded_step()
        torch.cuda.synchronize()
        # rank 0's ground truth is for its own batch == seed 2 == the shared batch
        srec = datasets.recall_at_k(ml.cpu().numpy().view(np.uint64), gt) if rank == 0 else 0.0
        log(f"sharded search + all-gather + merge: recall@{args.k} {srec:.4f}")
        for _ in range(args.warmup):
            sharded_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sharded_step()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sharded = {"queries_per_s": round(args.nq * args.steps / float(t.item()), 1), "recall_at_k": round(srec, 4),
                   "shards": world, "rows_per_shard": per, "ef": chosen,
                   "exchange": "RCCL all_gather of [nq,k] (f32 dist, u64 label) + merge_topk kernel"}

    # ---- CPU baseline: the oracle (a port of the reference algorithm) on the host cores --
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc  # the checker/baseline, never the product path
        orc.build()
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "bench_index.bin")
            h.saveIndex(path)
            o = orc.OracleHNSW.load(path, orc.L2, args.dim, mode=orc.SEQ, o3=True)
        o.set_ef(chosen)
        ncpu = min(args.cpu_queries, args.nq)
        r = o.search(queries[:ncpu], args.k, n_threads=threads)
        # cheap end-to-end sanity at full size: the GPU's answers for the sample against the
        # oracle in the GPU's summation order (bit-exact parity lives in tests/)
        o.set_mode(orc.TREE64)
        r2 = o.search(queries[:min(256, ncpu)], args.k, n_threads=threads)
        same = bool(np.array_equal(r2["label"].astype(np.uint64), lab.cpu().numpy().view(np.uint64)[:len(r2["label"])]))
        cpu = {"value": round(ncpu / r["seconds"], 1), "unit": "queries/s", "cores": threads, "kind": "port",
               "sample": f"first {ncpu} queries of the batch, same graph, ef={chosen}, k={args.k}, "
                         f"oracle -O3 in the reference's sequential-sum arithmetic, {threads} threads",
               "recall_at_k": round(datasets.recall_at_k(r["label"].astype(np.uint64), gt[:ncpu]), 4),
               "gpu_ids_equal_oracle_on_256": same}

    if rank == 0:
        # HBM-side traffic per launch comes from a separate rocprofv3 --pmc FETCH_SIZE pass of this same
        # command (counters cannot be read from inside the process); the committed summary is quoted
        # when the workload is the one it was collected on.
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r1", "hnsw_c2_pmc_fetch_final.json")
        if os.path.exists(pmc) and args.build == "host" and (args.n, args.dim, args.nq, args.k, args.M, chosen) == (1_000_000, 128, 10_000, 10, 16, 32):
            with open(pmc) as f:
                traffic = int(json.load(f)["traffic_bytes_per_launch"])
            traffic_src = "profiles/r1/hnsw_c2_pmc_fetch_final.json (FETCH_SIZE KB x 1024 x 2, gfx950 correction)"
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "queries/sec @ recall@10>=0.95 (1M x 128, k=10)",
            "value": round(qps, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"HNSW {args.n}x{args.dim} fp32 SIFT-like synthetic, M={args.M}, "
                                   f"efConstruction={args.efc}, {args.nq}-query batch per GPU, k={args.k}",
                       "ef": chosen, "recall_at_10": round(recall, 4), "ef_sweep": sweep,
                       "parallelism": "query-parallel replicas" if world > 1 else "single GPU",
                       "mean_ndis": round(float(st[:, 0].mean()), 1), "mean_hops": round(float(st[:, 1].mean()), 1),
                       "build_seconds_host" if args.build == "host" else "build_seconds_device": round(build_s, 1),
                       "build_threads": threads},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src, "kernel": "hnsw_search_kernel", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes},
            "cpu_baseline": cpu,
        }
        if sharded is not None:
            out["sharded"] = sharded
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_bruteforce(args) -> None:
    """BASELINE.json configs[2]: exact brute-force k-NN, 1M x 960 (GIST-like synthetic), 10 000-query
    batches, k=10, on the MFMA candidate pass + exact re-rank.  One GPU (replicas for N > 1)."""
    import torch

    from vectordb_amd import datasets
    from vectordb_amd import index as vi

    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--workload bruteforce is a single-GPU line")
    dim = 960 if args.dim == 128 else args.dim
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    threads = max(1, min(args.build_threads or int(os.environ.get("OMP_NUM_THREADS", "0")) or 16, 16))
    t0 = time.time()
    law = datasets.ClusteredLaw(dim, latent=64)
    base = law.sample(args.n, seed=1)
    queries = law.sample(args.nq, seed=2)
    log(f"data {base.shape} + {queries.shape} in {time.time() - t0:.1f}s")
    t0 = time.time()
    bf = vi.BruteforceSearch(vi.L2Space(dim), args.n)
    bf.add_items(base)
    bf.to_device(0)
    log(f"uploaded + prepared (mean, centred f32/bf16 copies, norms) in {time.time() - t0:.1f}s")
    dq = torch.from_numpy(queries).to(dev)
    prec = {"fp32": vi.BF_FP32, "bf16": vi.BF_BF16, "bf16_rerank": vi.BF_BF16_RERANK}[args.precision]

    def timed(precision, steps, warmup):
        for _ in range(warmup):
            bf.search_batch(dq, args.k, precision=precision)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in ev:
            a.record()
            out = bf.search_batch(dq, args.k, precision=precision)
            b.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, float(np.mean([a.elapsed_time(b) for a, b in ev])), out

    # exact answer (f32 MFMA + exact re-rank + certificate) = ground truth for the bf16 recall
    el32, ms32, (d32, l32) = timed(vi.BF_FP32, max(1, args.steps // 4), 1)
    truth = l32.cpu().numpy().view(np.uint64)
    elapsed, ms, (dd, ll) = timed(prec, args.steps, args.warmup)
    recall = datasets.recall_at_k(ll.cpu().numpy().view(np.uint64), truth)
    flops = 2.0 * args.nq * args.n * dim
    peak = 157.3 if args.precision == "fp32" else 2500.0   # MI355X_MICROARCH.md dense peaks, TFLOP/s
    achieved = flops / (ms * 1e-3) / 1e12
    log(f"fp32-exact {ms32:.1f} ms/batch ({flops / (ms32 * 1e-3) / 1e12:.1f} TFLOP/s of 157.3); "
        f"{args.precision} {ms:.1f} ms/batch, recall@{args.k} vs exact {recall:.4f}")

    cpu = None
    if not args.no_cpu_baseline:
        from oracle import oracle as orc
        orc.build()
        ncpu = min(2 * threads, args.nq)
        ob = orc.OracleBruteforce(orc.L2, dim, args.n, mode=orc.SEQ, o3=True)
        ob.add(base)
        r = ob.search(queries[:ncpu], args.k, n_threads=threads)
        ob.set_mode = None
        ot = orc.OracleBruteforce(orc.L2, dim, args.n, mode=orc.TREE64, o3=True)
        ot.add(base)
        r2 = ot.search(queries[:4], args.k, n_threads=4)
        same = bool(np.array_equal(r2["label"].astype(np.uint64), truth[:4]) and
                    np.array_equal(r2["dist"].view(np.uint32), d32.cpu().numpy()[:4].view(np.uint32)))
        cpu = {"value": round(ncpu / r["seconds"], 2), "unit": "queries/s", "cores": threads, "kind": "port",
               "sample": f"first {ncpu} queries, full {args.n}x{dim} scan, oracle -O3 sequential-sum arithmetic",
               "gpu_fp32_bitexact_vs_oracle_on_4": same}
    out = {
        "metric": "queries/sec exact brute-force k-NN (1M x 960, k=10)",
        "value": round(args.nq * args.steps / elapsed, 1), "unit": "queries/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.precision != "fp32" else "f32",
        "data": "synthetic",
        "config": {"workload": f"brute-force k-NN {args.n}x{dim} GIST-like synthetic, {args.nq}-query batch, k={args.k}, "
                               f"precision {args.precision}", "recall_at_10_vs_exact": round(recall, 4),
                   "fp32_exact_ms_per_batch": round(ms32, 2),
                   "fp32_exact_tflops": round(flops / (ms32 * 1e-3) / 1e12, 1)},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": None, "kernel": "bf_gemm_filter (whole step timed: "
                     "GEMM+filter, compaction, re-rank)", "kernel_ms": round(ms, 3), "algorithmic_flops_per_launch": flops},
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
