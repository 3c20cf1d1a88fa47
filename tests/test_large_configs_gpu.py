"""The sharded k = 100 shape of BASELINE.json configs[4] (C5) composed against the oracle doing the same partition, and
indexes beyond 2^22 and 2^23 elements, where the library changes behaviour: spill tables handed out from the first
launch at ef >= 64 (capi.cpp search_device) and the element limit of the no-allocation entry point (launch.hpp)."""
import os

import numpy as np
import pytest

from tests.helpers import assert_same_results
from vectordb_amd import _lib, datasets
from vectordb_amd import index as vi

pytestmark = pytest.mark.gpu


def _oracle_of(orc, h, tmp_path, metric, dim, name="g.bin"):
    path = str(tmp_path / name)
    h.saveIndex(path)
    o = orc.OracleHNSW.load(path, metric, dim, mode=orc.TREE64)
    os.remove(path)
    return o


def test_c5_shape_sharded_k100(orc, tmp_path):
    """configs[4] in shape: 128-d L2, contiguous shards with their own device-built sub-graphs (graph seed 100 + s,
    labels = global row ids), k = 100 at ef = 200, per-shard device search, merge_topk on the device.  Ids and distances
    of a sample of queries equal the oracle doing the same partition + per-shard search + (dist, label) merge; the
    whole batch's recall@100 against exact brute force over all rows is reported and gated."""
    import torch
    from vectordb_amd import sharding
    n, dim, G, k, ef, nq, sample = 600_000, 128, 3, 100, 200, 2000, 48
    law = datasets.ClusteredLaw(dim, latent=32)
    base = law.sample(n, seed=1)
    q = law.sample(nq, seed=2)
    dq = torch.from_numpy(q).cuda()
    dl, ll, od, ol = [], [], [], []
    for s in range(G):
        hs = sharding.build_shard(base, G, s, vi.L2, 16, 200, n_threads=16, device_build=0)
        assert hs.checkIntegrity()[0] == 0
        hs.setefSearch(ef)
        d, l, st = hs.search_batch(dq, k, stats=True)
        dl.append(d)
        ll.append(l)
        o = _oracle_of(orc, hs, tmp_path, orc.L2, dim)
        o.set_ef(ef)
        r = o.search(q[:sample], k, n_threads=16)
        assert_same_results((d[:sample].cpu().numpy(), l[:sample].cpu().numpy().view(np.uint64)), r, st[:sample].cpu().numpy().view(np.uint32))
        lo, hi = sharding.shard_range(n, G, s)
        assert lo <= int(r["label"].min()) and int(r["label"].max()) < hi   # labels are global row ids of this shard
        od.append(r["dist"])
        ol.append(r["label"].astype(np.uint64))
        del hs, o
    md, ml = vi.merge_topk(torch.stack(dl), torch.stack(ll))
    torch.cuda.synchronize()
    alld, alll = np.concatenate(od, axis=1), np.concatenate(ol, axis=1)
    order = np.lexsort((alll, alld), axis=1)[:, :k]
    np.testing.assert_array_equal(ml[:sample].cpu().numpy().view(np.uint64), np.take_along_axis(alll, order, 1))
    np.testing.assert_array_equal(md[:sample].cpu().numpy().view(np.uint32), np.take_along_axis(alld, order, 1).view(np.uint32))
    bf = vi.BruteforceSearch(vi.L2Space(dim), n)
    bf.add_items(base)
    bf.to_device(0)
    _, truth = bf.search_batch(dq, k)
    rec = datasets.recall_at_k(ml.cpu().numpy().view(np.uint64), truth.cpu().numpy().view(np.uint64))
    print(f"C5 shape ({G} shards x {n // G} rows, k={k}, ef={ef}): recall@100 = {rec:.4f}")
    assert rec >= 0.95


def _big_index(n, dim, M=8, efc=64):
    x = datasets.clustered(n, dim, 5, latent=8)
    h = vi.HierarchicalNSW(vi.L2Space(dim), n, M, efc, 100)
    h.add_items_device(x, n_threads=16, device=0)
    return h, x


def test_index_beyond_2p22_elements(orc, tmp_path):
    """2^22 + 4096 elements (device-built, low dimension): from 2^22 on, launches at ef >= 64 get spill tables from the
    FIRST launch (no report exists yet).  Device == oracle on a sample at ef 48 (no tables), 64 and 200 (tables, k = 100),
    first launch and repeat; the no-allocation entry point (limit 2^23) gives the same answers."""
    import torch
    n, dim = (1 << 22) + 4096, 16
    h, x = _big_index(n, dim)
    assert h.getElementCount() == n
    o = _oracle_of(orc, h, tmp_path, orc.L2, dim)
    q = datasets.clustered(300, dim, 6, latent=8)
    dq = torch.from_numpy(q).cuda()
    for ef, k in ((48, 10), (64, 10), (200, 100)):
        h.setefSearch(ef)
        o.set_ef(ef)
        ref = o.search(q, k, n_threads=16)
        for _ in range(2):   # the first launch at this ef, and one that has seen its report
            d, l, st = h.search_batch(dq, k, stats=True)
            torch.cuda.synchronize()
            assert_same_results((d.cpu().numpy(), l.cpu().numpy().view(np.uint64)), ref, st.cpu().numpy().view(np.uint32))
        d1 = torch.empty_like(d)
        l1 = torch.empty_like(l)
        h.search_batch_noalloc(dq, d1, l1)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(l1.cpu().numpy(), l.cpu().numpy())
        np.testing.assert_array_equal(d1.cpu().numpy().view(np.uint32), d.cpu().numpy().view(np.uint32))
    # self-search: a stored vector finds itself (the graph is connected enough at this size)
    ids = np.arange(0, n, n // 200)[:200]
    d, l = h.search_batch(x[ids], 1)
    assert (l[:, 0] == ids.astype(np.uint64)).mean() > 0.99 and (d[l[:, 0] == ids.astype(np.uint64), 0] == 0).all()


def test_index_beyond_2p23_elements(orc, tmp_path):
    """2^23 + 64 elements: the no-allocation entry point refuses the index with a clear error (its static scratch holds
    bitmaps of 2^23 bits); the allocating entry point answers as the oracle does."""
    import torch
    n, dim = (1 << 23) + 64, 16
    h, x = _big_index(n, dim)
    q = datasets.clustered(64, dim, 7, latent=8)
    dq = torch.from_numpy(q).cuda()
    h.setefSearch(100)
    d, l, st = h.search_batch(dq, 10, stats=True)
    torch.cuda.synchronize()
    lib = _lib.lib()
    rc = lib.vdb_search_batch_device_noalloc(h._h, dq.data_ptr(), 64, 10, d.data_ptr(), l.data_ptr(), None, None)
    assert rc == _lib.VDB_ERR_INVALID and str(1 << 23).encode() in lib.vdb_last_error()
    o = _oracle_of(orc, h, tmp_path, orc.L2, dim)
    o.set_ef(100)
    assert_same_results((d.cpu().numpy(), l.cpu().numpy().view(np.uint64)), o.search(q, 10, n_threads=16), st.cpu().numpy().view(np.uint32))


def test_filtered_batch_that_grows_on_one_stream(orc):
    """ADVICE r2 (capi.cpp work.fail_list): same index and stream, a filter that accepts NOTHING (every query goes
    through the fail list to the unbounded kernel), nq = 100 then nq = 300: the second call's fail list must fit the
    buffer the first call sized.  Both equal the oracle (count 0); then a filter that accepts a few."""
    x = datasets.clustered(4000, 24, 31)
    q = datasets.clustered(300, 24, 32)
    from tests.helpers import build_pair
    h, o = build_pair(orc, x, vi.L2, 8, 64)
    h.to_device(0)
    h.setefSearch(32)
    o.set_ef(32)
    none = np.zeros(4000, bool)
    for nq in (100, 300, 37):
        d, l, st = h.search_batch(q[:nq], 5, stats=True, filter=none)
        ref = o.search(q[:nq], 5, filter_mod=10**9, filter_keep=10**9 - 1)   # label % 1e9 == 1e9 - 1: nothing
        assert (st[:, 3] == 0).all() and (ref["count"] == 0).all()
        np.testing.assert_array_equal(st[:, 0].astype(np.uint64), ref["ndis"])
    few = np.arange(4000) % 400 == 7
    for nq in (50, 300):
        d, l, st = h.search_batch(q[:nq], 5, stats=True, filter=few)
        assert_same_results((d, l), o.search(q[:nq], 5, filter_mod=400, filter_keep=7), st)


def test_links_of_a_streamed_index_is_an_error_not_an_exception(tmp_path):
    """vdb_index_links(level 0) on an index opened by vdb_index_load_to_device returns -1 with an error
    message (the rows are not on the host) - no exception crosses the C ABI; upper levels still answer."""
    x = datasets.clustered(3000, 16, 41)
    h = vi.HierarchicalNSW(vi.L2Space(16), 3000, 6, 40)
    h.add_items(x)
    path = str(tmp_path / "s.bin")
    h.saveIndex(path)
    s = vi.HierarchicalNSW(vi.L2Space(16), location=path, stream_to_device=0)
    import ctypes as C
    lib = _lib.lib()
    buf = (C.c_uint32 * 64)()
    assert lib.vdb_index_links(s._h, 0, 0, buf, 64) == -1 and b"not held on the host" in lib.vdb_last_error()
    top = h.entry
    assert h.level_of(top) >= 1 and s.links(top, 1) == h.links(top, 1)
    assert s.checkIntegrity()[0] == -1
