#!/usr/bin/env python3
"""bench.py — queries/sec of the HNSW k-NN query path on MI355X (BASELINE.json metric).

Default workload (any N): BASELINE.json configs[1] — 1M x 128 fp32 "SIFT-like" synthetic base set,
HNSW M=16 efConstruction=200, batches of 10 000 queries, k=10, ef = the smallest value of the sweep
{32,64,96,128,192,256} whose recall@10 against exact brute force is >= 0.95.

A "step" is one pass of the hot path (vdb_search_batch_device: greedy descent + layer-0 beam search
+ top-k output) over one batch of 10 000 queries already resident in HBM.

At N = 1 the same JSON line also carries a "bruteforce" block: BASELINE.json configs[2] (exact k-NN
1M x 960, 10 000 queries, k=10) on the MFMA path in f32 and bf16, with its own roofline and CPU
baseline (the other half of the metric: "% HBM/MFMA roofline").

N > 1 (one process per GPU, torch.distributed / RCCL).  configs[1] fits one GPU, so `value` is
what a deployment would run for it: a replica of the index per rank, every rank its own batch of
10 000 queries ("weak": per-GPU work fixed), no collective on the data path.  The base-set sharding
of the north star (each rank its own sub-graph, per-rank top-k lists all-gathered over RCCL, merged
on the GPU) is timed in the same run under "sharded" with the step split into search / all-gather /
merge.  `--workload c4` / `c5` are the configurations that only exist sharded (10M x 96 inner
product, 100M x 128 k=100): there the sharded step IS the step, `value` is its queries/sec, and
scaling is "strong" (the base set is fixed, each rank builds 1/N of it on its GPU).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EF_SWEEP = (32, 64, 96, 128, 192, 256)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
# What a kernel that does nothing but this path's row gathers gets (tools/micro/gather_ceiling.hip, measured on the box:
# profiles/r3/pipelined_walk_experiment.txt): random 512-B rows, one dependent gather per step, 8-16 waves per CU
GATHER_PEAK_GBS_1M = 7600.0     # 1 M rows (0.5 GB: partly in the 256 MB Infinity Cache)
GATHER_PEAK_GBS_LARGE = 6200.0  # 20 M - 100 M rows (10 - 51 GB)
GATHER_PEAK_SOURCE = "profiles/r3/pipelined_walk_experiment.txt (tools/micro/gather_ceiling.hip: the row gathers alone)"
MFMA_PEAK_TF = {"fp32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md dense peaks, TFLOP/s


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def host_cores() -> dict:
    """The host cores this process may use.  The GPU box is a shared machine: os.cpu_count() shows
    every CPU of the host, the job's share is what its affinity mask, its cgroup CPU quota and the
    thread-count variable the box exports (OMP_NUM_THREADS) allow, whichever is smallest."""
    visible = os.cpu_count() or 1
    limits = {"cpu_count": visible}
    try:
        limits["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                limits["cgroup_quota"] = max(1, math.ceil(int(quota) / int(period)))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            limits["cgroup_quota"] = max(1, math.ceil(q / p))
    except (OSError, ValueError):
        pass
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if omp > 0:
        limits["OMP_NUM_THREADS"] = omp
    share = min(limits.values())
    return {"cores": share, "host_cpus_visible": visible, "limits": limits}


class Dist:
    """torch.distributed, or nothing at N = 1.  In rehearsal mode (several ranks on ONE GPU over gloo,
    VDB_BENCH_REHEARSAL=1) every collective runs on CPU tensors."""

    def __init__(self, world: int, rehearsal: bool, dev):
        self.world, self.rehearsal, self.dev = world, rehearsal, dev
        if world > 1:
            import torch.distributed as dist
            self.d = dist
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)

    def barrier(self):
        import torch
        torch.cuda.synchronize()
        if self.world > 1:
            self.d.barrier()
        torch.cuda.synchronize()

    def reduce(self, value: float, op: str = "max") -> float:
        if self.world == 1:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64, device="cpu" if self.rehearsal else self.dev)
        self.d.all_reduce(t, op={"max": self.d.ReduceOp.MAX, "sum": self.d.ReduceOp.SUM, "min": self.d.ReduceOp.MIN}[op])
        return float(t.item())

    def gather_objects(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.d.all_gather_object(out, obj)
        return out

    def close(self):
        if self.world > 1:
            self.d.destroy_process_group()


def launch_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process and
    print the JSON line its rank 0 prints (everything else the children write to stdout goes to stderr)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching " + " ".join(cmd))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        try:
            if out.startswith("{") and "metric" in json.loads(out):
                line = out
                continue
        except ValueError:
            pass
        print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        log("the ranks exited without printing a result line")
        rc = 1
    return rc


def timed_sharded_steps(sharding, hs, shared_q, k, steps, warmup, dd, comm=None):
    """W + K sharded steps.  The timed step is ONE call of the C-ABI entry point
    (vdb_sharded_search_batch_device: shard search + grouped ncclAllGather + merge on one stream) when an
    RCCL communicator exists, else the torch.distributed spelling of it (gloo rehearsal).  The split into
    search / all-gather / merge comes from a second loop of the torch spelling with HIP events between
    the stages (mean over the K steps, max over ranks)."""
    import torch
    step = (lambda: comm.sharded_search(hs, shared_q, k)) if comm is not None else (lambda: sharding.sharded_search(hs, shared_q, k))
    for _ in range(warmup):
        step()
    dd.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    dd.barrier()
    wall = dd.reduce(time.perf_counter() - t0)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    for e in ev:
        sharding.sharded_search(hs, shared_q, k, events=e)
    dd.barrier()
    parts = [float(np.mean([e[i].elapsed_time(e[i + 1]) for e in ev])) for i in range(3)]
    parts = [dd.reduce(p) for p in parts]
    return wall, {"search": round(parts[0], 4), "all_gather": round(parts[1], 4), "merge": round(parts[2], 4),
                  "timed_step": "vdb_sharded_search_batch_device (C-ABI, RCCL)" if comm is not None else "torch.distributed spelling"}, out


def main() -> None:
    if os.environ.get("VDB_BENCH_WATCHDOG"):  # diagnostics: dump every thread's stack after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VDB_BENCH_WATCHDOG"]), exit=False)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--base-rows", dest="n", type=int, default=0, help="base vectors (default = the named workload's)")
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--queries", dest="nq", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--ef", type=int, default=0, help="c4/c5: the efSearch to time (default 48 / 200)")
    ap.add_argument("--build-threads", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=1.0, help="wall seconds the cpu_baseline leg aims at (all host cores busy)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true")
    ap.add_argument("--no-bruteforce", action="store_true", help="skip the configs[2] block of the default line")
    ap.add_argument("--no-ef-block", action="store_true", help="skip the larger-ef block (traversal_by_ef) of the default line")
    ap.add_argument("--workload", choices=("hnsw", "bruteforce", "c4", "c5"), default="hnsw",
                    help="hnsw = BASELINE.json configs[1] (the headline metric, + the configs[2] block at N=1); "
                         "bruteforce = configs[2] alone; c4 / c5 = the sharded configurations configs[3] / configs[4]")
    ap.add_argument("--precision", choices=("fp32", "bf16", "bf16_rerank"), default="bf16_rerank")
    ap.add_argument("--build", choices=("host", "device"), default="host",
                    help="graph builder: host = the reference algorithm on the host cores (default: the graph the headline\n"
                         "number is quoted on), device = vdb_index_add_batch_device (DESIGN.md 4.3)")
    ap.add_argument("--no-c4-block", action="store_true", help="skip the configs[3] block (10M x 96, 8 shards on one GPU) of the default line")
    ap.add_argument("--no-c5-block", action="store_true", help="skip the configs[4] block (one 12.5M-row shard of the 8-way split, k=100) of the default line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # A plain `python bench.py --gpus N`: this process becomes the launcher.  It starts the N ranks as CHILD
        # processes (torch.distributed.run, one per GPU) before anything here has touched the GPU - no torch import,
        # no HIP call - relays rank 0's JSON line and exits with the children's status.
        raise SystemExit(launch_ranks(args.gpus))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: pass --gpus {world}, or start bench.py without a launcher "
                         f"(python bench.py --gpus N starts its own ranks)")
    # rehearsal on a one-GPU box only: VDB_BENCH_REHEARSAL=1 puts every rank on device 0 over gloo
    rehearsal = os.environ.get("VDB_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    elif torch.cuda.device_count() < world:   # (counting devices does not initialise the GPU)
        raise SystemExit(f"--gpus {world} but {torch.cuda.device_count()} GPU(s) visible "
                         f"(VDB_BENCH_REHEARSAL=1 rehearses the ranks on one GPU over gloo)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dd = Dist(world, rehearsal, dev)
    cores = host_cores()
    threads = args.build_threads or cores["cores"]
    if world > 1 and not args.build_threads:
        # the ranks of one node share its host cores (torch.distributed.run, --nnodes=1): a rank builds with its part of
        # them, four at least where the share allows (a share quoted per process is not divided down to one thread)
        threads = max(min(threads, 4), threads // world)
    try:
        if args.workload == "bruteforce":
            if world != 1:
                raise SystemExit("--workload bruteforce is a single-GPU line")
            out = run_bruteforce(args, dev, cores, threads, standalone=True)
        elif args.workload in ("c4", "c5"):
            out = run_sharded_workload(args, dd, rank, local_rank, world, dev, cores, threads)
        else:
            out = run_hnsw(args, dd, rank, local_rank, world, dev, cores, threads)
        if rank == 0 and out is not None:
            print(json.dumps(out), flush=True)
    finally:
        dd.close()


# ------------------------------------------------------------------------------------------------
# configs[1]: the headline
# ------------------------------------------------------------------------------------------------

def run_hnsw(args, dd, rank, local_rank, world, dev, cores, threads):
    import torch

    from vectordb_amd import datasets
    from vectordb_amd import index as vi

    n, dim, k = args.n or 1_000_000, args.dim or 128, args.k or 10

    # ---- synthetic data -------------------------------------------
    t0 = time.time()
    law = datasets.ClusteredLaw(dim, latent=32)
    base = law.sample(n, seed=1)
    queries = law.sample(args.nq, seed=2 + rank)          # every rank its own batch
    if rank == 0:
        log(f"data {base.shape} + {queries.shape} in {time.time() - t0:.1f}s; host cores {cores}")

    # ---- graph build on the host cores (reference algorithm), upload ------------------
    t0 = time.time()
    h = vi.HierarchicalNSW(vi.L2Space(dim), n, args.M, args.efc, 100)
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
    bf = vi.BruteforceSearch(vi.L2Space(dim), n)
    bf.add_items(base)
    bf.to_device(local_rank)
    gt_d, gt_l = bf.search_batch(dq, k)
    torch.cuda.synchronize()
    gt = gt_l.cpu().numpy().view(np.uint64)
    del bf
    if rank == 0:
        log(f"exact ground truth in {time.time() - t0:.1f}s")

    # ---- ef sweep: smallest ef with recall@k >= 0.95 ----------------------------------
    sweep = {}
    chosen = None
    for ef in EF_SWEEP:
        if ef < k:
            continue
        h.setefSearch(ef)
        _, lab = h.search_batch(dq, k)
        torch.cuda.synchronize()
        rec = datasets.recall_at_k(lab.cpu().numpy().view(np.uint64), gt)
        sweep[ef] = round(rec, 4)
        if chosen is None and rec >= 0.95:
            chosen = ef
            break
    if chosen is None:
        chosen = max(sweep)
    chosen = int(dd.reduce(float(chosen)))  # all ranks time the same ef: the largest any rank needs
    h.setefSearch(chosen)
    dist_out, lab, st = h.search_batch(dq, k, stats=True)
    torch.cuda.synchronize()
    recall = datasets.recall_at_k(lab.cpu().numpy().view(np.uint64), gt)
    st = st.cpu().numpy().view(np.uint32).astype(np.int64)
    # algorithmic bytes of one launch: vectors scored, link rows read, query, output
    maxM0, maxM = 2 * args.M, args.M
    alg_bytes = int((st[:, 0] * 4 * dim + st[:, 1] * 4 * (maxM0 + 1) + st[:, 2] * 4 * (maxM + 1)).sum()
                    + args.nq * (4 * dim + k * 12))
    if rank == 0:
        log(f"ef sweep {sweep} -> ef={chosen}, recall@{k}={recall:.4f}, mean ndis {st[:, 0].mean():.0f}, "
            f"mean hops {st[:, 1].mean():.1f}, {alg_bytes / 1e9:.2f} GB algorithmic per launch")

    # ---- timed region: W warm-up + K steps, barrier + synchronize on both sides --------
    for _ in range(args.warmup):
        h.search_batch(dq, k)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    dd.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()          # same stream the kernels are enqueued on (torch's current stream)
        h.search_batch(dq, k)
        ev[i][1].record()
    dd.barrier()
    elapsed = dd.reduce(time.perf_counter() - t0)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    qps = world * args.nq * args.steps / elapsed
    if rank == 0:
        log(f"timed: {elapsed / args.steps * 1e3:.4f} ms per step, {qps:.4e} queries/s over {world} rank(s)")

    # ---- the same K steps spread over two HIP streams (outside the timed region; not `value`): a launch ends in a tail - its
    # last queries finish on a half-empty GPU - that the next batch's launch on the other stream fills
    two_streams = None
    if rank == 0 and world == 1:
        sts2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for i in range(4):
            with torch.cuda.stream(sts2[i & 1]):
                h.search_batch(dq, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            with torch.cuda.stream(sts2[i & 1]):
                h.search_batch(dq, k)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        two_streams = {"queries_per_s": round(args.nq * args.steps / el2, 1), "ms_per_step": round(el2 / args.steps * 1e3, 4),
                       "algorithmic_GBps": round(alg_bytes * args.steps / el2 / 1e9, 1),
                       "note": "independent batches on alternating streams; `value` above is one stream, one launch after the other"}
        log(f"two streams: {two_streams['ms_per_step']} ms per step, {two_streams['queries_per_s']:.4e} queries/s")

    # ---- the same batch at larger ef (outside the timed region; not `value`): the traversal beyond the headline's ef is
    # where C4 at high recall and C5 live, so its rate belongs in the driver's record too
    by_ef = None
    if rank == 0 and world == 1 and not args.no_ef_block:
        by_ef = {}
        for ef in (64, 100, 200):
            h.setefSearch(ef)
            _, lab_e, st_e = h.search_batch(dq, k, stats=True)
            torch.cuda.synchronize()
            st_e = st_e.cpu().numpy().view(np.uint32).astype(np.int64)
            bytes_e = int((st_e[:, 0] * 4 * dim + st_e[:, 1] * 4 * (maxM0 + 1) + st_e[:, 2] * 4 * (maxM + 1)).sum()
                          + args.nq * (4 * dim + k * 12))
            for _ in range(3):   # (the library sizes its visited set from the reports of completed launches)
                h.search_batch(dq, k)
                torch.cuda.synchronize()
            ev_e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in ev_e:
                a.record()
                h.search_batch(dq, k)
                b.record()
            torch.cuda.synchronize()
            ms_e = float(np.mean([a.elapsed_time(b) for a, b in ev_e]))
            # (and on two streams in turn, as `headline_on_two_streams`: what the kernel sustains when a launch's tail is filled)
            sts_e = [torch.cuda.Stream(device=dev) for _ in range(2)]
            for i in range(6):
                with torch.cuda.stream(sts_e[i & 1]):
                    h.search_batch(dq, k)
                torch.cuda.synchronize()
            t0e = time.perf_counter()
            for i in range(10):
                with torch.cuda.stream(sts_e[i & 1]):
                    h.search_batch(dq, k)
            torch.cuda.synchronize()
            ms_e2 = (time.perf_counter() - t0e) / 10 * 1e3
            by_ef[str(ef)] = {"ms_per_batch": round(ms_e, 4), "queries_per_s": round(args.nq / ms_e * 1e3, 1),
                              "recall_at_10": round(datasets.recall_at_k(lab_e.cpu().numpy().view(np.uint64), gt), 4),
                              "mean_ndis": round(float(st_e[:, 0].mean()), 1), "mean_hops": round(float(st_e[:, 1].mean()), 1),
                              "batch_ns_per_expanded_node": round(ms_e * 1e6 / float(st_e[:, 1].sum()), 3),
                              "on_two_streams": {"ms_per_batch": round(ms_e2, 4), "frac_of_hbm_peak": round(bytes_e / ms_e2 / 1e6 / HBM_PEAK_GBS, 4)},
                              "roofline": {"bound": "hbm", "achieved": round(bytes_e / ms_e / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(bytes_e / ms_e / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": bytes_e}}
        h.setefSearch(chosen)
        log("larger ef, same batch: " + ", ".join(f"ef {e}: {v['ms_per_batch']} ms ({v['roofline']['frac']:.2f} of HBM peak)" for e, v in by_ef.items()))

    # ---- the non-bare path (hnsw_core.hpp:1141-1148) on the same graph and batch: a random filter that keeps a half / a tenth
    # of the elements, ef 64, against exact filtered brute force (outside the timed region; not `value`)
    filtered = None
    if rank == 0 and world == 1 and not args.no_ef_block:
        filtered = {}
        h.setefSearch(64)
        for keep in (0.5, 0.1):
            ok = np.random.default_rng(17).random(n) < keep
            bits = torch.frombuffer(bytearray(h.filter_bitmap(ok)), dtype=torch.int32).to(dev)
            _, lab_f, st_f = h.search_batch(dq, k, stats=True, filter=bits)
            bf2 = vi.BruteforceSearch(vi.L2Space(dim), int(ok.sum()))   # exact: brute force over the allowed rows only
            bf2.add_items(base[ok], np.nonzero(ok)[0].astype(np.uint64))
            bf2.to_device(local_rank)
            _, truth_f = bf2.search_batch(dq[:1000].contiguous(), k)
            torch.cuda.synchronize()
            del bf2
            rec_f = datasets.recall_at_k(lab_f[:1000].cpu().numpy().view(np.uint64), truth_f.cpu().numpy().view(np.uint64))
            for _ in range(6):   # (the launches' reports size the candidate buffer and the visited set: up to five steps)
                h.search_batch(dq, k, filter=bits)
                torch.cuda.synchronize()
            ev_f = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in ev_f:
                a.record()
                h.search_batch(dq, k, filter=bits)
                b.record()
            torch.cuda.synchronize()
            st_f = st_f.cpu().numpy().view(np.uint32).astype(np.int64)
            ms_f = float(np.mean([a.elapsed_time(b) for a, b in ev_f]))
            filtered[f"keeps_{keep}"] = {"ef": 64, "ms_per_batch": round(ms_f, 3), "queries_per_s": round(args.nq / ms_f * 1e3, 1),
                                         "recall_at_10_vs_exact_filtered": round(rec_f, 4), "mean_ndis": round(float(st_f[:, 0].mean()), 1),
                                         "mean_hops": round(float(st_f[:, 1].mean()), 1),
                                         # what a filter costs beyond the hops it adds (a filter that keeps a tenth makes the
                                         # search expand ten times the nodes: that part is the algorithm's, hnsw_core.hpp:316)
                                         "batch_ns_per_expanded_node": round(ms_f * 1e6 / float(st_f[:, 1].sum()), 3)}
        h.setefSearch(chosen)
        log("filtered, ef 64: " + ", ".join(f"{name}: {v['ms_per_batch']} ms (recall {v['recall_at_10_vs_exact_filtered']})" for name, v in filtered.items()))

    # ---- sharded variant (north star): base split over ranks + RCCL all-gather + merge --
    sharded = None
    if world > 1 and not args.no_sharded:
        # (a failure here is reported in the block's place and never costs the headline: `value` is measured above)
        try:
            from vectordb_amd import sharding
            per = (n + world - 1) // world
            log(f"rank {rank}: building its {per}-row shard ({args.build} builder)")
            t0 = time.time()
            hs = sharding.build_shard(base, world, rank, vi.L2, args.M, args.efc, n_threads=threads,
                                      device_build=local_rank if args.build == "device" else None)
            hs.to_device(local_rank)
            shard_build_s = dd.reduce(time.time() - t0)
            hs.setefSearch(chosen)
            shared_q = torch.from_numpy(law.sample(args.nq, seed=2)).to(dev)   # the SAME batch on every rank
            md, ml = sharding.sharded_search(hs, shared_q, k)
            torch.cuda.synchronize()
            # rank 0's ground truth is for its own batch == seed 2 == the shared batch
            srec = datasets.recall_at_k(ml.cpu().numpy().view(np.uint64), gt) if rank == 0 else 0.0
            log(f"rank {rank}: sharded search + all-gather + merge ok" + (f", recall@{k} {srec:.4f}" if rank == 0 else ""))
            comm = None if dd.rehearsal else sharding.Communicator.from_torch(local_rank)
            if comm is not None:   # the C entry point returns what the torch spelling returned
                cd, cl = comm.sharded_search(hs, shared_q, k)
                torch.cuda.synchronize()
                assert torch.equal(cl, ml) and torch.equal(cd, md), "vdb_sharded_search_batch_device differs from the torch.distributed path"
            wall, split, _ = timed_sharded_steps(sharding, hs, shared_q, k, args.steps, args.warmup, dd, comm)
            ranks = dd.gather_objects({"rank": rank, "device": torch.cuda.current_device(), "host": os.uname().nodename,
                                       "rccl_rank": comm.rank if comm else None, "rccl_size": comm.n_ranks if comm else None})
            sharded = {"queries_per_s": round(args.nq * args.steps / wall, 1), "ms_per_step": round(wall / args.steps * 1e3, 4),
                       "ms_split_max_over_ranks": split, "recall_at_k": round(srec, 4),
                       "shards": world, "rows_per_shard": per, "ef": chosen, "shard_build_seconds": round(shard_build_s, 1),
                       "world_size": dd.d.get_world_size(), "backend": dd.d.get_backend(), "ranks": ranks,
                       "exchange": "all_gather_into_tensor of [nq,k] f32 dist + [nq,k] u64 label, then merge_topk kernel"}
        except Exception as e:  # noqa: BLE001
            log(f"rank {rank}: the sharded step failed: {e!r}")
            sharded = {"error": repr(e)}

    # ---- CPU baseline: the oracle (a port of the reference algorithm) on the host cores --
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc  # the checker/baseline, never the product path
        orc.build()
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "bench_index.bin")
            h.saveIndex(path)
            o = orc.OracleHNSW.load(path, orc.L2, dim, mode=orc.SEQ, o3=True)
        o.set_ef(chosen)
        ct = cores["cores"]
        r = o.search(queries, k, n_threads=ct)              # one pass sizes the sample
        passes = int(min(40, max(1, math.ceil(args.cpu_seconds / max(r["seconds"], 1e-6)))))
        secs = [o.search(queries, k, n_threads=ct)["seconds"] for _ in range(passes)]
        # cheap end-to-end sanity at full size: the GPU's answers against the oracle in the GPU's
        # summation order (bit-exact parity lives in tests/)
        o.set_mode(orc.TREE64)
        r2 = o.search(queries, k, n_threads=ct)
        same = bool(np.array_equal(r2["label"].astype(np.uint64), lab.cpu().numpy().view(np.uint64)) and
                    np.array_equal(r2["dist"].view(np.uint32), dist_out.cpu().numpy().view(np.uint32)))
        cpu = {"value": round(args.nq * passes / sum(secs), 1), "unit": "queries/s", "cores": ct, "kind": "port",
               "sample": f"{passes} passes over the whole {args.nq}-query batch ({sum(secs):.2f} s wall, {sum(secs) * ct:.0f} core-seconds), "
                         f"same graph, ef={chosen}, k={k}, oracle -O3 in the reference's sequential-sum arithmetic, {ct} threads = "
                         f"the job's share of the host ({cores['host_cpus_visible']} CPUs visible; limits {cores['limits']})",
               "host_cpus_visible": cores["host_cpus_visible"],
               "recall_at_k": round(datasets.recall_at_k(r["label"].astype(np.uint64), gt), 4),
               "gpu_ids_and_distances_equal_oracle_on_whole_batch": same}
        del o

    bruteforce = c4_block = c5_block = None
    if rank == 0 and world == 1:
        del h, base
        # further blocks of the one line; a failure in one of them is reported in its place and never costs the headline
        def guarded_block(fn, *a):
            try:
                return fn(*a)
            except Exception as e:  # noqa: BLE001
                log(f"{fn.__name__} failed: {e!r}")
                return {"error": repr(e)}
        if not args.no_bruteforce:
            bruteforce = guarded_block(run_bruteforce, args, dev, cores, threads, False)
        if not args.no_c4_block:
            c4_block = guarded_block(run_c4_one_gpu, args, dev, threads)
        if not args.no_c5_block:
            c5_block = guarded_block(run_c5_shard_one_gpu, args, dev, threads)

    if rank != 0:
        return None
    # HBM-side traffic per launch comes from a separate rocprofv3 --pmc FETCH_SIZE pass of this same
    # command (counters cannot be read from inside the process); the committed summary is quoted
    # when the workload is the one it was collected on.
    traffic, traffic_src = None, None
    for rel in ("profiles/r3/hnsw_c2_pmc_fetch.json", "profiles/r2/hnsw_c2_pmc_fetch.json", "profiles/r1/hnsw_c2_pmc_fetch_final.json"):
        pmc = os.path.join(ROOT, rel)
        if os.path.exists(pmc) and args.build == "host" and (n, dim, args.nq, k, args.M, chosen) == (1_000_000, 128, 10_000, 10, 16, 32):
            with open(pmc) as f:
                traffic = int(json.load(f)["traffic_bytes_per_launch"])
            traffic_src = rel + " (FETCH_SIZE KB x 1024 x 2, gfx950 correction)"
            break
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
        "config": {"workload": f"HNSW {n}x{dim} fp32 SIFT-like synthetic, M={args.M}, "
                               f"efConstruction={args.efc}, {args.nq}-query batch per GPU, k={k}",
                   "ef": chosen, "recall_at_10": round(recall, 4), "ef_sweep": sweep,
                   "parallelism": "query-parallel replicas (no data-path collective); base-set sharding under 'sharded'"
                   if world > 1 else "single GPU",
                   "mean_ndis": round(float(st[:, 0].mean()), 1), "mean_hops": round(float(st[:, 1].mean()), 1),
                   "build_seconds_host" if args.build == "host" else "build_seconds_device": round(build_s, 1),
                   "build_threads": threads},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_src, "kernel": "hnsw_search_kernel", "kernel_ms": round(kern_ms, 4),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "peak_gather": GATHER_PEAK_GBS_1M if n <= 2_000_000 else GATHER_PEAK_GBS_LARGE,
                     "frac_of_peak_gather": round(achieved / (GATHER_PEAK_GBS_1M if n <= 2_000_000 else GATHER_PEAK_GBS_LARGE), 4),
                     "peak_gather_source": GATHER_PEAK_SOURCE},
        "cpu_baseline": cpu,
    }
    if two_streams is not None:
        out["headline_on_two_streams"] = two_streams
    if by_ef is not None:
        out["traversal_by_ef"] = by_ef
    if filtered is not None:
        out["filtered_search"] = filtered
    if sharded is not None:
        out["sharded"] = sharded
    if bruteforce is not None:
        out["bruteforce"] = bruteforce
    if c4_block is not None:
        out["c4_one_gpu"] = c4_block
    if c5_block is not None:
        out["c5_shard_one_gpu"] = c5_block
    return out


# ------------------------------------------------------------------------------------------------
# configs[3] / configs[4] on ONE GPU, as blocks of the default line
# ------------------------------------------------------------------------------------------------

def _traversal_roofline(st, dim, M, nq, k, ms):
    st = st.cpu().numpy().view(np.uint32).astype(np.int64)
    alg = int((st[:, 0] * 4 * dim + st[:, 1] * 4 * (2 * M + 1) + st[:, 2] * 4 * (M + 1)).sum() + nq * (4 * dim + k * 12))
    return alg, st


def run_c4_one_gpu(args, dev, threads):
    """BASELINE.json configs[3] at FULL size on one GPU: 10M x 96, rows L2-normalised, metric 1 - <a,b>, 8 shards x 1.25M
    contiguous rows, each an independent graph (seed 100 + s, labels = global row ids) built on the device; the 10 000
    queries are searched on the eight shards one after the other and merged by merge_topk - the multi-GPU pipeline
    of sharding.py with the all-gather replaced by "the lists are already here" (on eight GPUs each rank runs one of the
    eight searches).  ef = the smallest of (32, 48, 64) with recall@10 >= 0.95 against exact brute force over all rows."""
    import torch

    from vectordb_amd import datasets, sharding
    from vectordb_amd import index as vi
    n, dim, G, nq, k, M = 10_000_000, 96, 8, args.nq, 10, 16
    law = datasets.ClusteredLaw(dim, latent=32, normalize=True)
    t0 = time.time()
    base = law.sample(n, seed=1)
    q = law.sample(nq, seed=2)
    dq = torch.from_numpy(q).to(dev)
    t_data = time.time() - t0
    t0 = time.time()
    bf = vi.BruteforceSearch(vi.InnerProductSpace(dim), n)
    bf.add_items(base)
    bf.to_device(dev.index)
    _, truth = bf.search_batch(dq, k)
    torch.cuda.synchronize()
    truth = truth.cpu().numpy().view(np.uint64)
    del bf
    t_truth = time.time() - t0
    t0 = time.time()
    shards = [sharding.build_shard(base, G, s, vi.IP, M, 200, n_threads=threads, device_build=dev.index) for s in range(G)]
    build_s = time.time() - t0
    del base
    log(f"c4: data {t_data:.1f}s, exact ground truth {t_truth:.1f}s, {G} shards on the device in {build_s:.1f}s")

    def step(stats=False):
        res = [h.search_batch(dq, k, stats=stats) for h in shards]
        md, ml = vi.merge_topk(torch.stack([r[0] for r in res]), torch.stack([r[1] for r in res]))
        return md, ml, ([r[2] for r in res] if stats else None)

    sweep, ef = {}, None
    for e in (32, 48, 64):
        for h in shards:
            h.setefSearch(e)
        _, ml, _ = step()
        torch.cuda.synchronize()
        sweep[e] = round(datasets.recall_at_k(ml.cpu().numpy().view(np.uint64), truth), 4)
        ef = e
        if sweep[e] >= 0.95:
            break
    _, ml, sts = step(stats=True)
    torch.cuda.synchronize()
    alg = sum(_traversal_roofline(st, dim, M, nq, k, 0)[0] for st in sts)
    for _ in range(3):
        step()
        torch.cuda.synchronize()
    steps = 10
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    for e3 in ev:
        e3[0].record()
        res = [h.search_batch(dq, k) for h in shards]
        e3[1].record()
        vi.merge_topk(torch.stack([r[0] for r in res]), torch.stack([r[1] for r in res]))
        e3[2].record()
    torch.cuda.synchronize()
    ms_search = float(np.mean([a.elapsed_time(b) for a, b, _ in ev]))
    ms_merge = float(np.mean([b.elapsed_time(c) for _, b, c in ev]))
    ms = ms_search + ms_merge
    gbs = alg / ms_search / 1e6
    # The same step with each shard's search on a HIP stream of its own (one GPU holding several shards: a launch's tail -
    # the last queries, finishing on a half-empty GPU - overlaps the next shards' searches), joined by events before the merge.
    streams = [torch.cuda.Stream(device=dev) for _ in shards]
    main = torch.cuda.current_stream(dev)

    def step_streams():
        start = torch.cuda.Event()
        start.record(main)
        res = []
        for h, st in zip(shards, streams):
            st.wait_event(start)
            with torch.cuda.stream(st):
                res.append(h.search_batch(dq, k))
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
        return vi.merge_topk(torch.stack([r[0] for r in res]), torch.stack([r[1] for r in res]))

    _, ml_s = step_streams()
    torch.cuda.synchronize()
    same_streams = bool(torch.equal(ml_s, ml))
    for _ in range(3):
        step_streams()
        torch.cuda.synchronize()
    ev_s = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev_s:
        a.record()
        step_streams()
        b.record()
    torch.cuda.synchronize()
    ms_streams = float(np.mean([a.elapsed_time(b) for a, b in ev_s]))
    log(f"c4: ef sweep {sweep} -> ef={ef}; {ms:.2f} ms per batch ({ms_search:.2f} search of {G} shards + {ms_merge:.2f} merge), "
        f"{gbs:.0f} GB/s algorithmic")
    return {"workload": f"BASELINE.json configs[3]: HNSW {n}x{dim} rows L2-normalised, metric 1 - <a,b>, {G} shards x {n // G} rows, "
                        f"M={M}, efConstruction=200, {nq}-query batch, k={k}; all {G} shards on ONE GPU, searched one after the other",
            "ef": ef, "ef_sweep_recall_at_10": sweep, "recall_at_10": sweep[ef], "ms_per_batch": round(ms, 3),
            "ms_split": {"search_all_shards": round(ms_search, 3), "merge_topk": round(ms_merge, 3)},
            "queries_per_s_one_gpu": round(nq / ms * 1e3, 1),
            "shards_on_streams_of_their_own": {"ms_per_batch": round(ms_streams, 3), "queries_per_s_one_gpu": round(nq / ms_streams * 1e3, 1),
                                               "algorithmic_GBps_whole_step": round(alg / ms_streams / 1e6, 1),
                                               "same_labels_as_the_sequential_step": same_streams},
            "shard_build_seconds_device": round(build_s, 1),
            "shard_build_points_per_s": round(n / build_s, 0),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "traffic": None, "kernel": f"hnsw_search_kernel ({G} launches per batch)", "kernel_ms": round(ms_search, 3),
                         "algorithmic_bytes_per_launch": alg}}


def run_c5_shard_one_gpu(args, dev, threads):
    """BASELINE.json configs[4], the work of ONE of its eight ranks at full size: rows [0, 12.5M) of the 100M x 128 set
    as one device-built sub-graph (8.2 GB image), the 10 000-query batch, k = 100, ef = the smallest of (100, 128, 200,
    300) whose recall@100 against exact brute force over the shard is >= 0.95, then merge_topk over the one list (the
    rank's share of the exchange step).  The whole set on one GPU (65.6 GB image, 92 s build) is tools/exp_c5.py."""
    import torch

    from vectordb_amd import datasets
    from vectordb_amd import index as vi
    n, dim, nq, k, M = 12_500_000, 128, args.nq, 100, 16
    law = datasets.ClusteredLaw(dim, latent=32)
    t0 = time.time()
    base = law.sample(n, seed=1000)
    q = law.sample(nq, seed=2)
    dq = torch.from_numpy(q).to(dev)
    t_data = time.time() - t0
    t0 = time.time()
    bf = vi.BruteforceSearch(vi.L2Space(dim), n)
    bf.add_items(base)
    bf.to_device(dev.index)
    _, truth = bf.search_batch(dq[:2000].contiguous(), k)
    torch.cuda.synchronize()
    truth = truth.cpu().numpy().view(np.uint64)
    del bf
    t_truth = time.time() - t0
    t0 = time.time()
    h = vi.HierarchicalNSW(vi.L2Space(dim), n, M, 200, 100)
    h.add_items_device(base, n_threads=threads, device=dev.index)
    build_s = time.time() - t0
    del base
    log(f"c5 shard: data {t_data:.1f}s, exact ground truth (2000 queries) {t_truth:.1f}s, graph on the device in {build_s:.1f}s "
        f"({h.device_bytes() / 1e9:.2f} GB)")
    by_ef, chosen = {}, None
    for ef in (100, 128, 200, 300):
        h.setefSearch(ef)
        _, lab, st = h.search_batch(dq, k, stats=True)
        torch.cuda.synchronize()
        rec = datasets.recall_at_k(lab[:2000].cpu().numpy().view(np.uint64), truth)
        alg, stn = _traversal_roofline(st, dim, M, nq, k, 0)
        for _ in range(3):
            h.search_batch(dq, k)
            torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record()
            h.search_batch(dq, k)
            b.record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        by_ef[str(ef)] = {"recall_at_100": round(rec, 4), "ms_per_batch": round(ms, 3), "queries_per_s": round(nq / ms * 1e3, 1),
                          "mean_ndis": round(float(stn[:, 0].mean()), 1), "mean_hops": round(float(stn[:, 1].mean()), 1),
                          "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": alg,
                                       "peak_gather": GATHER_PEAK_GBS_LARGE, "frac_of_peak_gather": round(alg / ms / 1e6 / GATHER_PEAK_GBS_LARGE, 4),
                                       "peak_gather_source": GATHER_PEAK_SOURCE}}
        if rec >= 0.95:
            chosen = ef
            break
    d, l = h.search_batch(dq, k)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record()
        vi.merge_topk(d.unsqueeze(0), l.unsqueeze(0))
        b.record()
    torch.cuda.synchronize()
    log("c5 shard: " + ", ".join(f"ef {e}: recall@100 {v['recall_at_100']}, {v['ms_per_batch']} ms ({v['roofline']['frac']:.2f} of HBM peak)" for e, v in by_ef.items()))
    return {"workload": f"BASELINE.json configs[4], one rank's share at 8 GPUs: HNSW {n}x{dim} L2 (rows [0, {n}) of 100M), M={M}, "
                        f"efConstruction=200, graph built on the device, {nq}-query batch, k={k}",
            "ef_with_recall_at_100_ge_0.95": chosen, "by_ef": by_ef, "build_seconds_device": round(build_s, 1),
            "build_points_per_s": round(n / build_s, 0), "device_image_gb": round(h.device_bytes() / 1e9, 2),
            "merge_topk_ms_one_list": round(float(np.mean([a.elapsed_time(b) for a, b in ev])), 3)}


# ------------------------------------------------------------------------------------------------
# configs[2]: exact brute force on the MFMA path
# ------------------------------------------------------------------------------------------------

def run_bruteforce(args, dev, cores, threads, standalone: bool):
    """BASELINE.json configs[2]: exact brute-force k-NN, 1M x 960 (GIST-like synthetic), 10 000-query
    batches, k=10: f32 MFMA candidate pass + exact re-rank + certificate (bit-identical to the CPU
    path) and bf16 MFMA candidate pass + exact re-rank.  standalone: its own JSON line (timing
    `--precision`); else the "bruteforce" block of the default line (both precisions)."""
    import torch

    from vectordb_amd import datasets
    from vectordb_amd import index as vi

    n = (args.n or 1_000_000) if standalone else 1_000_000
    dim = (args.dim or 960) if standalone else 960
    nq, k = args.nq, (args.k or 10)
    t0 = time.time()
    law = datasets.ClusteredLaw(dim, latent=64)
    base = law.sample(n, seed=1)
    queries = law.sample(nq, seed=2)
    log(f"brute force: data {base.shape} + {queries.shape} in {time.time() - t0:.1f}s")
    t0 = time.time()
    bf = vi.BruteforceSearch(vi.L2Space(dim), n)
    bf.add_items(base)
    bf.to_device(dev.index)
    log(f"brute force: uploaded + prepared (mean, centred f32/bf16 copies, norms) in {time.time() - t0:.1f}s")
    dq = torch.from_numpy(queries).to(dev)

    def timed(precision, steps, warmup):
        for _ in range(warmup):
            bf.search_batch(dq, k, precision=precision)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in ev:
            a.record()
            out = bf.search_batch(dq, k, precision=precision)
            b.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, float(np.mean([a.elapsed_time(b) for a, b in ev])), out

    flops = 2.0 * nq * n * dim
    # exact answer (f32 MFMA + exact re-rank + certificate) = ground truth for the bf16 recall
    el32, ms32, (d32, l32) = timed(vi.BF_FP32, max(2, args.steps // 4), 1)
    truth = l32.cpu().numpy().view(np.uint64)
    prec_name = args.precision if standalone and args.precision != "fp32" else "bf16_rerank"
    prec = {"bf16": vi.BF_BF16, "bf16_rerank": vi.BF_BF16_RERANK}[prec_name]
    el16, ms16, (d16, l16) = timed(prec, args.steps, args.warmup)
    recall = datasets.recall_at_k(l16.cpu().numpy().view(np.uint64), truth)
    tf32, tf16 = flops / (ms32 * 1e-3) / 1e12, flops / (ms16 * 1e-3) / 1e12
    log(f"brute force: fp32-exact {ms32:.1f} ms/batch ({tf32:.1f} TFLOP/s of {MFMA_PEAK_TF['fp32']}); "
        f"{prec_name} {ms16:.2f} ms/batch ({tf16:.0f} TFLOP/s of {MFMA_PEAK_TF['bf16']}), recall@{k} vs exact {recall:.4f}")

    cpu, bitexact = None, None
    if not args.no_cpu_baseline:
        from oracle import oracle as orc
        orc.build()
        ct = cores["cores"]
        ncpu = min(2 * ct, nq)
        ob = orc.OracleBruteforce(orc.L2, dim, n, mode=orc.SEQ, o3=True)
        ob.add(base)
        r = ob.search(queries[:ncpu], k, n_threads=ct)
        del ob
        ot = orc.OracleBruteforce(orc.L2, dim, n, mode=orc.TREE64, o3=True)
        ot.add(base)
        nchk = min(ct, nq)
        r2 = ot.search(queries[:nchk], k, n_threads=ct)
        del ot
        bitexact = bool(np.array_equal(r2["label"].astype(np.uint64), truth[:nchk]) and
                        np.array_equal(r2["dist"].view(np.uint32), d32.cpu().numpy()[:nchk].view(np.uint32)))
        cpu = {"value": round(ncpu / r["seconds"], 2), "unit": "queries/s", "cores": ct, "kind": "port",
               "sample": f"first {ncpu} queries, full {n}x{dim} scan ({r['seconds']:.1f} s wall, {r['seconds'] * ct:.0f} core-seconds), "
                         f"oracle -O3 sequential-sum arithmetic, {ct} threads", "host_cpus_visible": cores["host_cpus_visible"]}

    def roof(name, ms, tf):
        peak = MFMA_PEAK_TF["fp32" if name == "fp32" else "bf16"]
        extra = {}
        if name != "fp32" and (n, dim, nq, k) == (1_000_000, 960, 10_000, 10):
            # what this step runs at with the MFMAs alone (no operand stream, no fragment reads, no epilogue) at the clock
            # the chip holds under the load: measured once, profiles/r3/bf16_gemm_four_wave_experiment.txt (12.15 ms)
            extra = {"peak_sustained": 1580.0, "frac_of_peak_sustained": round(tf / 1580.0, 4),
                     "peak_sustained_source": "profiles/r3/bf16_gemm_four_wave_experiment.txt (the same step with MFMAs only)"}
        return {"ms_per_batch": round(ms, 3), "queries_per_s": round(nq / ms * 1e3, 1),
                "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), **extra,
                             "traffic": None, "kernel": "bf_gemm_filter (f32)" if name == "fp32" else "bf_gemm8_filter (bf16)",
                             "kernel_ms": round(ms, 3),
                             "kernel_ms_note": "HIP events around the whole step on its stream: prepare-queries, the slab launches of "
                                               "GEMM+filter, compactions, exact re-rank (the GEMM's own time: profiles/)",
                             "algorithmic_flops_per_launch": flops}}

    if standalone:
        sel = roof("fp32" if args.precision == "fp32" else prec_name, ms32 if args.precision == "fp32" else ms16,
                   tf32 if args.precision == "fp32" else tf16)
        steps = max(2, args.steps // 4) if args.precision == "fp32" else args.steps
        el = el32 if args.precision == "fp32" else el16
        return {"metric": "queries/sec exact brute-force k-NN (1M x 960, k=10)", "value": round(nq * steps / el, 1), "unit": "queries/s",
                "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": round(el / steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
                "config": {"workload": f"brute-force k-NN {n}x{dim} GIST-like synthetic, {nq}-query batch, k={k}, precision {args.precision}",
                           "recall_at_10_vs_exact": round(recall, 4), "fp32_exact_ms_per_batch": round(ms32, 2),
                           "fp32_exact_tflops": round(tf32, 1)},
                "roofline": sel["roofline"], "cpu_baseline": cpu, "gpu_fp32_bitexact_vs_oracle": bitexact}
    return {"workload": f"BASELINE.json configs[2]: exact brute-force k-NN {n}x{dim} GIST-like synthetic, {nq}-query batch, k={k}",
            "fp32": roof("fp32", ms32, tf32), "bf16_rerank": dict(roof("bf16_rerank", ms16, tf16), recall_at_10_vs_exact=round(recall, 4)),
            "gpu_fp32_bitexact_vs_oracle": bitexact, "cpu_baseline": cpu}


# ------------------------------------------------------------------------------------------------
# configs[3] / configs[4]: the configurations that exist only sharded
# ------------------------------------------------------------------------------------------------

def run_sharded_workload(args, dd, rank, local_rank, world, dev, cores, threads):
    """c4: 10M x 96, rows L2-normalised, inner product, k=10.  c5: 100M x 128, L2, k=100.  The base set
    is fixed and split over the ranks (contiguous global row ids; shard s's rows are drawn from the law
    with seed 1000+s so that no rank materialises another rank's rows); every rank builds its shard's
    graph on its GPU (vdb_index_add_batch_device), all ranks search the SAME 10 000-query batch, the
    [nq,k] lists are all-gathered and merged on every rank.  Ground truth: exact brute force per shard
    by the MFMA kernels, all-gathered and merged the same way.  One torchrun command per N gives the
    1/2/4/8 curve: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N --workload c4`."""
    import torch

    from vectordb_amd import datasets, sharding
    from vectordb_amd import index as vi

    c4 = args.workload == "c4"
    n = args.n or (10_000_000 if c4 else 100_000_000)
    dim = args.dim or (96 if c4 else 128)
    k = args.k or (10 if c4 else 100)
    ef = args.ef or (48 if c4 else 200)
    metric = vi.IP if c4 else vi.L2
    law = datasets.ClusteredLaw(dim, latent=32, normalize=c4)
    lo, hi = sharding.shard_range(n, world, rank)
    t0 = time.time()
    mine = law.sample(hi - lo, seed=1000 + rank)
    labels = np.arange(lo, hi, dtype=np.uint64)
    q = law.sample(args.nq, seed=2)
    dq = torch.from_numpy(q).to(dev)
    log(f"rank {rank}: rows [{lo}, {hi}) in {time.time() - t0:.1f}s")

    space = vi.InnerProductSpace(dim) if c4 else vi.L2Space(dim)
    t0 = time.time()
    bf = vi.BruteforceSearch(space, hi - lo)
    bf.add_items(mine, labels)
    bf.to_device(local_rank)
    td, tl = bf.search_batch(dq, k)
    gd, gl = sharding.all_gather_results(td, tl)
    _, truth = vi.merge_topk(gd, gl)
    torch.cuda.synchronize()
    truth = truth.cpu().numpy().view(np.uint64)
    del bf
    log(f"rank {rank}: exact ground truth over its shard + all-gather + merge in {time.time() - t0:.1f}s")

    t0 = time.time()
    hs = vi.HierarchicalNSW(space, hi - lo, args.M, args.efc, 100 + rank)
    hs.add_items_device(mine, labels, n_threads=threads, device=local_rank)
    build_s = dd.reduce(time.time() - t0)
    del mine
    hs.setefSearch(ef)
    md, ml = sharding.sharded_search(hs, dq, k)
    torch.cuda.synchronize()
    recall = datasets.recall_at_k(ml.cpu().numpy().view(np.uint64), truth)
    log(f"rank {rank}: shard graph on the device in {build_s:.1f}s ({hs.device_bytes() / 1e9:.2f} GB); ef={ef} recall@{k} {recall:.4f}")
    comm = None
    if world > 1 and not dd.rehearsal:
        comm = sharding.Communicator.from_torch(local_rank)
    elif world == 1:
        comm = sharding.Communicator(sharding.Communicator.unique_id(), 1, 0, local_rank)
    wall, split, _ = timed_sharded_steps(sharding, hs, dq, k, args.steps, args.warmup, dd, comm)
    _, _, st = hs.search_batch(dq, k, stats=True)
    st = st.cpu().numpy().view(np.uint32).astype(np.int64)
    alg_bytes = int((st[:, 0] * 4 * dim + st[:, 1] * 4 * (2 * args.M + 1) + st[:, 2] * 4 * (args.M + 1)).sum()
                    + args.nq * (4 * dim + k * 12))
    achieved = alg_bytes / (dd.reduce(split["search"], "max") * 1e-3) / 1e9
    ranks = dd.gather_objects({"rank": rank, "device": torch.cuda.current_device(), "rows": hi - lo})
    if rank != 0:
        return None
    return {
        "metric": f"queries/sec, sharded HNSW ({'10M x 96 inner product, k=10' if c4 else '100M x 128, k=100'})",
        "value": round(args.nq * args.steps / wall, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[{3 if c4 else 4}]: HNSW {n}x{dim} {'normalised, 1 - <a,b>' if c4 else 'L2'}, "
                               f"{world} shard(s), M={args.M}, efConstruction={args.efc}, {args.nq}-query batch, k={k}",
                   "ef": ef, f"recall_at_{k}": round(recall, 4), "parallelism": f"base-set sharding over {world} rank(s), "
                   "all-gather of the per-rank top-k lists + merge_topk on every rank", "ms_split_max_over_ranks": split,
                   "shard_build_seconds_device": round(build_s, 1), "world_size": world,
                   "backend": dd.d.get_backend() if world > 1 else None, "ranks": ranks},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "hnsw_search_kernel (rank 0's shard)",
                     "kernel_ms": split["search"], "algorithmic_bytes_per_launch": alg_bytes},
        "cpu_baseline": None,
    }


if __name__ == "__main__":
    main()
