"""Layer-0 graph construction on the device (vectordb_amd/csrc/hnsw_build.hip).

A lock-free batch builder cannot reproduce the sequential builder's graph (insertion order is part of
it), so it is gated on what SURVEY.md prescribes: structural integrity (hnsw_core.hpp:1216-1245) and
recall against the reference-algorithm build of the same data.  Searching the graph it wrote back is
still bit-exact against the oracle on the same file."""
import os

import numpy as np
import pytest

from tests.helpers import assert_same_results
from vectordb_amd import datasets
from vectordb_amd import index as vi

pytestmark = pytest.mark.gpu


def _recall(h, x, q, k, ef, labels=None):
    bf = vi.BruteforceSearch(h.space, len(x))
    bf.add_items(x, labels)
    bf.to_device(0)
    _, truth = bf.search_batch(q, k)
    h.setefSearch(ef)
    _, l = h.search_batch(q, k)
    return datasets.recall_at_k(l, truth)


# host_base: sub-graphs up to this size are linked by the host builder; 64 forces the recursion over
# the layers (layer l = layer 0 of an index over the elements of level >= l) down to a handful of elements
@pytest.mark.parametrize("n,dim,metric,M,efc,host_base", [(30000, 32, vi.L2, 16, 200, 8192), (30000, 32, vi.L2, 16, 200, 64),
                                                          (20000, 96, vi.IP, 8, 64, 64), (3000, 16, vi.L2, 4, 32, 64),
                                                          (3000, 16, vi.L2, 4, 32, 8192)])
def test_device_build_recall_and_integrity(orc, tmp_path, monkeypatch, n, dim, metric, M, efc, host_base):
    monkeypatch.setenv("VDB_BUILD_HOST_BASE", str(host_base))
    x = datasets.clustered(n, dim, 71, normalize=(metric == vi.IP))
    q = datasets.clustered(500, dim, 72, normalize=(metric == vi.IP))
    labels = np.arange(n, dtype=np.uint64) + 10_000
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)

    host = vi.HierarchicalNSW(space, n, M, efc, 100)
    host.add_items(x, labels, n_threads=8)
    host.to_device(0)
    r_host = _recall(host, x, q, 10, 64, labels)

    dev = vi.HierarchicalNSW(space, n, M, efc, 100)
    dev.add_items_device(x, labels, n_threads=8, device=0)
    assert dev.getElementCount() == n
    # same level draw as the host builder: same level of every element, same upper-layer population
    assert [dev.level_of(i) for i in range(0, n, 97)] == [host.level_of(i) for i in range(0, n, 97)]
    assert dev.max_level == host.max_level and dev.level_of(dev.entry) == dev.max_level
    rc, mn, mx = dev.checkIntegrity()
    assert rc == 0 and mn >= 1, (rc, mn)   # every element has an inbound link (hnsw_core.hpp:1238; the builder's last pass sees to it)
    r_dev = _recall(dev, x, q, 10, 64, labels)
    print(f"recall@10 ef=64: host build {r_host:.4f}, device build {r_dev:.4f}")
    assert r_dev >= r_host - 0.02

    # the written-back host graph is the graph the device holds: save it, let the oracle search the
    # file, and compare with the device search bit for bit
    path = str(tmp_path / "dev_built.bin")
    dev.saveIndex(path)
    o = orc.OracleHNSW.load(path, metric, dim, mode=orc.TREE64)
    for ef in (16, 64):
        dev.setefSearch(ef)
        o.set_ef(ef)
        d, l, st = dev.search_batch(q[:200], 10, stats=True)
        assert_same_results((d, l), o.search(q[:200], 10), st)
    # and it survives a round trip through the reference's file format
    again = vi.HierarchicalNSW(space, location=path)
    again.to_device(0)
    again.setefSearch(64)
    d2, l2 = again.search_batch(q[:200], 10)
    np.testing.assert_array_equal(l2, l)


def test_device_build_appends_to_an_existing_graph(orc):
    n0, n1, dim = 8000, 12000, 24
    x = datasets.clustered(n0 + n1, dim, 73)
    q = datasets.clustered(300, dim, 74)
    h = vi.HierarchicalNSW(vi.L2Space(dim), n0 + n1, 16, 100, 100)
    h.add_items(x[:n0], n_threads=4)
    h.add_items_device(x[n0:], n_threads=4)
    assert h.getElementCount() == n0 + n1
    assert h.checkIntegrity()[0] in (0, 4)
    assert _recall(h, x, q, 10, 64) >= 0.9


@pytest.mark.parametrize("M", [24, 32])
def test_device_build_wide_lists(orc, tmp_path, monkeypatch, M):
    """maxM0 = 48 / 64: a full list plus its newcomers is up to 96 entries in build_reverse_kernel (two
    per lane), and the heuristic scores against up to 64 kept neighbours.  Integrity, recall against the
    host build, and bit-exact searches over the written-back graph."""
    monkeypatch.setenv("VDB_BUILD_HOST_BASE", "64")
    n, dim = 20000, 24
    x = datasets.clustered(n, dim, 75)
    q = datasets.clustered(300, dim, 76)
    host = vi.HierarchicalNSW(vi.L2Space(dim), n, M, 100, 100)
    host.add_items(x, n_threads=8)
    host.to_device(0)
    r_host = _recall(host, x, q, 10, 48)
    dev = vi.HierarchicalNSW(vi.L2Space(dim), n, M, 100, 100)
    dev.add_items_device(x, n_threads=8)
    assert dev.checkIntegrity()[0] == 0
    sizes = [len(dev.links(i, 0)) for i in range(0, n, 53)]
    assert max(sizes) <= 2 * M and max(sizes) > 32 and np.mean(sizes) > M / 2     # lists beyond one lane's worth exist
    r_dev = _recall(dev, x, q, 10, 48)
    print(f"M={M}: recall@10 ef=48 host {r_host:.4f} device {r_dev:.4f}, mean list {np.mean(sizes):.1f}, max {max(sizes)}")
    assert r_dev >= r_host - 0.02
    path = str(tmp_path / "wide.bin")
    dev.saveIndex(path)
    o = orc.OracleHNSW.load(path, vi.L2, dim, mode=orc.TREE64)
    dev.setefSearch(48)
    o.set_ef(48)
    d, l, st = dev.search_batch(q, 10, stats=True)
    assert_same_results((d, l), o.search(q, 10), st)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 9])
def test_device_build_of_a_handful_at_small_M(orc, tmp_path, n):
    """M = 4 goes the way of an append; with so few elements usually none draws an upper level, so the
    first one has to anchor the graph (entry point) before the device rounds run."""
    dim = 8
    x = datasets.uniform(n, dim, 77 + n)
    h = vi.HierarchicalNSW(vi.L2Space(dim), 16, 4, 16, 100)
    h.add_items_device(x, n_threads=1)
    assert h.getElementCount() == n and h.entry < n and h.max_level == h.level_of(h.entry)
    h.setefSearch(16)
    d, l = h.search_batch(x, min(n, 3))
    assert (l[:, 0] == np.arange(n)).all() and (d[:, 0] == 0).all()       # every element finds itself
    path = str(tmp_path / "tiny.bin")
    h.saveIndex(path)
    o = orc.OracleHNSW.load(path, vi.L2, dim, mode=orc.TREE64)
    o.set_ef(16)
    assert_same_results((d, l), o.search(x, min(n, 3)))


def _device_built_bytes(tmp_path, name, n=6000, dim=16, M=8, efc=48, seed=81):
    x = datasets.clustered(n, dim, seed)
    h = vi.HierarchicalNSW(vi.L2Space(dim), n, M, efc, 100)
    h.add_items_device(x, n_threads=1, device=0)     # one host thread: the host-built top of the recursion has an order
    path = str(tmp_path / name)
    h.saveIndex(path)
    with open(path, "rb") as f:
        return f.read(), h


def test_device_build_is_deterministic_and_matches_its_golden(tmp_path, monkeypatch):
    """with ONE host thread (the only unordered step is the multi-threaded host build of the small
    top sub-graph) two device builds of the same data write the same bytes - pairs are emitted at fixed positions, the
    radix sort is stable, one wave owns a neighbour's list, the orphan pass runs in id order - so a device-built graph
    can have a golden: tests/golden/device_build_golden.json holds the SHA-256 of this build's saved file (made by
    tests/golden/make_device_build_golden.py on an MI355X)."""
    import hashlib
    import json
    monkeypatch.setenv("VDB_BUILD_HOST_BASE", "64")
    a, h = _device_built_bytes(tmp_path, "a.bin")
    b, _ = _device_built_bytes(tmp_path, "b.bin")
    assert a == b
    assert h.checkIntegrity()[0] == 0
    gp = os.path.join(os.path.dirname(__file__), "golden", "device_build_golden.json")
    with open(gp) as f:
        gold = json.load(f)
    assert hashlib.sha256(a).hexdigest() == gold["sha256"], "the device builder's output changed: regenerate the golden if that is intended"


def test_device_build_leaves_no_orphans_on_clustered_shards():
    """200 000 clustered rows at M = 16 (a shard of tests/test_large_configs_gpu.py): before the orphan pass a handful of
    elements per build ended without an inbound link (checkIntegrity 4); the host build of the same rows has none."""
    law = datasets.ClusteredLaw(128, latent=32)
    x = law.sample(200_000, seed=1)
    h = vi.HierarchicalNSW(vi.L2Space(128), len(x), 16, 200, 100)
    h.add_items_device(x, n_threads=16, device=0)
    rc, mn, mx = h.checkIntegrity()
    assert rc == 0 and mn >= 1


@pytest.mark.parametrize("dim,efc,metric,seed", [(128, 100, vi.IP, 1), (96, 40, vi.L2, 2), (96, 100, vi.L2, 3), (96, 100, vi.IP, 4),
                                                 (128, 100, vi.IP, 5), (32, 40, vi.L2, 6)])
def test_device_build_small_M_recall(dim, efc, metric, seed):
    """the shapes of the six fuzz cases that fell 0.021-0.030 below the host build (all M = 4 on
    2 000 rows, profiles/r2/fuzz_build.log), three data seeds each: within 0.02 of the host build's recall@10 at ef 64."""
    n, M = 2000, 4
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    worst = 0.0
    for s in range(3):
        x = datasets.clustered(n, dim, 900 + 10 * seed + s, normalize=(metric == vi.IP))
        q = datasets.clustered(300, dim, 950 + 10 * seed + s, normalize=(metric == vi.IP))
        host = vi.HierarchicalNSW(space, n, M, efc, 100)
        host.add_items(x, n_threads=8)
        host.to_device(0)
        dev = vi.HierarchicalNSW(space, n, M, efc, 100)
        dev.add_items_device(x, n_threads=8, device=0)
        r_host, r_dev = _recall(host, x, q, 10, 64), _recall(dev, x, q, 10, 64)
        worst = max(worst, r_host - r_dev)
        print(f"dim {dim} efc {efc} metric {metric} seed {s}: host {r_host:.4f} device {r_dev:.4f}")
    assert worst <= 0.02, worst


@pytest.mark.gpu
def test_orphan_pass_does_not_empty_a_hub(monkeypatch):
    """The case that stopped tests/fuzz_build.py in round 3 (seed 33, build 452): 2 000 x 128, M = 4, sixteen host
    threads.  One build in twenty came out 0.09 of recall@10 below the others: the orphan pass had replaced every entry
    of one hub's list by orphans whose only link was that hub - a closed star that the upper layers land in.  A list now
    gives up a quarter of its entries at most (hnsw_build.hip).  Forty builds: none may fall below the median by 0.03,
    and recall at ef 500 (what a trapped query never reaches) stays above 0.97 in each."""
    monkeypatch.setenv("VDB_BUILD_HOST_BASE", "64")
    n, dim, M, efc = 2000, 128, 4, 40
    x = datasets.clustered(n, dim, 396433519)
    q = datasets.clustered(300, dim, 677343783)
    space = vi.L2Space(dim)
    b = vi.BruteforceSearch(space, n)
    b.add_items(x)
    b.to_device(0)
    _, truth = b.search_batch(q, 10)
    r64, r500 = [], []
    for _ in range(40):
        h = vi.HierarchicalNSW(space, n, M, efc, 100)
        h.add_items_device(x, n_threads=16, device=0)
        assert h.checkIntegrity()[0] == 0
        for ef, out in ((64, r64), (500, r500)):
            h.setefSearch(ef)
            _, l = h.search_batch(q, 10)
            out.append(datasets.recall_at_k(l, truth))
    assert min(r64) >= np.median(r64) - 0.03, sorted(r64)[:5]
    assert min(r500) >= 0.97, sorted(r500)[:5]
