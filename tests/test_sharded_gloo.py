"""The N > 1 path on CPU: two processes over gloo run the product's partition + all-gather
(vectordb_amd/sharding.py).  There is no GPU here, so each rank's per-shard result lists come
from the oracle and the merge is checked in numpy; the merge kernel itself is a GPU test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vectordb_amd import sharding


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 1000, 10_000_001):
        for world in (1, 2, 3, 8):
            r = [sharding.shard_range(n, world, i) for i in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(hi - lo for lo, hi in r) == (n + world - 1) // world or n == 0
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import oracle as orc
        from vectordb_amd import index as vi
        rng = np.random.default_rng(1)
        x = rng.random((1201, 16), dtype=np.float32)
        q = rng.random((40, 16), dtype=np.float32)
        k = 7
        h = sharding.build_shard(x, world, rank, vi.L2, M=8, ef_construction=64)
        lo, hi = sharding.shard_range(len(x), world, rank)
        assert h.getElementCount() == hi - lo
        path = os.path.join(tmp, f"shard{rank}.bin")
        h.saveIndex(path)
        o = orc.OracleHNSW.load(path, orc.L2, 16, mode=orc.TREE64)
        o.set_ef(50)
        r = o.search(q, k)
        assert r["label"].min() >= lo and r["label"].max() < hi          # labels are global row ids
        d = torch.from_numpy(r["dist"])
        l = torch.from_numpy(r["label"].astype(np.int64))
        gd, gl = sharding.all_gather_results(d, l)
        assert gd.shape == (world, 40, k) and torch.equal(gd[rank], d) and torch.equal(gl[rank], l)
        # merge under (dist, label), as the merge_topk kernel defines it
        alld = gd.permute(1, 0, 2).reshape(40, -1).numpy()
        alll = gl.permute(1, 0, 2).reshape(40, -1).numpy()
        order = np.lexsort((alll, alld), axis=1)[:, :k]
        md, ml = np.take_along_axis(alld, order, 1), np.take_along_axis(alll, order, 1)
        # against exact global brute force: recall of the sharded search
        b = orc.OracleBruteforce(orc.L2, 16, len(x), mode=orc.TREE64)
        b.add(x)
        rb = b.search(q, k)
        hits = sum(len(set(a.tolist()) & set(t.tolist())) for a, t in zip(ml, rb["label"].astype(np.int64)))
        assert hits / (40 * k) > 0.97
        assert np.all(np.diff(md, axis=1) >= 0)
        # every rank computed the same merged lists
        t = torch.from_numpy(ml.copy())
        ref = t.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(t, ref)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_search_over_gloo(orc, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
