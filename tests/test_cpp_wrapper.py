"""The C++ mirror of the reference interface (vectordb_amd/cpp/vdb_hnsw.hpp) compiled with g++
against the C-ABI library, as a maintainer of the reference would use it."""
import os
import subprocess

import numpy as np
import pytest

from vectordb_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "demo"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp_wrapper_demo.cpp"), "-o", str(out),
                           f"-L{libdir}", "-lvdb_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return str(out)


@pytest.fixture(scope="module")
def ref_style_demo(tmp_path_factory):
    """tests/cpp_reference_style_demo.cpp: a caller in the reference's template spellings (HierarchicalNSW<float>,
    L2Space<float, L2Sqr<float>>, ...), -Wall -Werror clean."""
    out = tmp_path_factory.mktemp("cppref") / "demo"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", os.path.join(ROOT, "tests", "cpp_reference_style_demo.cpp"),
                           "-o", str(out), f"-L{libdir}", "-lvdb_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return str(out)


@pytest.mark.skipif(_lib.lib().vdb_device_count() > 0, reason="box has a GPU")
def test_reference_style_caller_host_side(ref_style_demo, tmp_path):
    r = subprocess.run([ref_style_demo, "host", str(tmp_path / "a.bin"), str(tmp_path / "b.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout + r.stderr)
    assert "host ok" in r.stdout


@pytest.mark.gpu
def test_reference_style_caller_matches_oracle(ref_style_demo, orc, tmp_path):
    """HierarchicalNSW<float> idx(&space, n, 16, 200, 100, true); addPoint / update / replace_deleted;
    SearchKNNCloserFirst from eight threads: labels and distances equal the oracle's over the saved graphs."""
    pa, pb = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    r = subprocess.run([ref_style_demo, "gpu", pa, pb], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, (r.returncode, r.stdout + r.stderr)
    l2_out, rest = r.stdout.split("ip\n")
    ip_out, bf_out = rest.split("bruteforce\n")
    parse = lambda text: [(int(a), float.fromhex(b)) for a, b in (line.split() for line in text.strip().splitlines())]
    n, dim, nq = 600, 20, 16
    # the demo's base vectors, read back from the inner-product index (never updated): its first rows are the queries
    raw = np.fromfile(pb, np.uint8)
    stride = 16 * 4 + 4 + dim * 4 + 8
    x = raw[96:96 + n * stride].reshape(n, stride)[:, 68:68 + dim * 4].copy().view(np.float32)
    o = orc.OracleHNSW.load(pa, orc.L2, dim, mode=orc.TREE64)
    o.set_ef(64)
    res = o.search(x[:nq], 10)
    assert parse(l2_out) == [(int(l), float(d)) for ls, ds in zip(res["label"], res["dist"]) for l, d in zip(ls, ds)]
    o = orc.OracleHNSW.load(pb, orc.IP, dim, mode=orc.TREE64)
    o.set_ef(30)
    res = o.search(x[:nq], 7)
    assert parse(ip_out) == [(int(l), float(d)) for ls, ds in zip(res["label"], res["dist"]) for l, d in zip(ls, ds)]
    ob = orc.OracleBruteforce(orc.L2, dim, n, mode=orc.TREE64)
    ob.add(x)
    res = ob.search(x[:nq], 3)
    assert parse(bf_out) == [(int(l), float(d)) for ls, ds in zip(res["label"], res["dist"]) for l, d in zip(ls[::-1], ds[::-1])]


@pytest.mark.skipif(_lib.lib().vdb_device_count() > 0, reason="box has a GPU")
def test_cpp_wrapper_host_side(demo, tmp_path):
    r = subprocess.run([demo, "host", str(tmp_path / "i.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host ok" in r.stdout


@pytest.mark.gpu
def test_cpp_wrapper_search_matches_oracle(demo, orc, tmp_path):
    path = str(tmp_path / "i.bin")
    r = subprocess.run([demo, "gpu", path], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    hnsw_out, bf_out, bf_filt_out = r.stdout.strip().split("bruteforce")
    bare_out, filt_out = hnsw_out.split("filtered\n")
    bf_filt_out = bf_filt_out.replace(" filtered\n", "")
    parse = lambda text: [(int(a), float.fromhex(b)) for a, b in (line.split() for line in text.strip().splitlines())]
    got, got_f = parse(bare_out), parse(filt_out)
    o = orc.OracleHNSW.load(path, orc.L2, 24, mode=orc.TREE64)
    o.set_ef(40)
    # the demo's queries are its first 10 base vectors: read them back from the saved index
    raw = np.fromfile(path, np.uint8)
    stride = 16 * 4 + 4 + 24 * 4 + 8
    rows = raw[96:96 + 500 * stride].reshape(500, stride)
    x = rows[:, 68:68 + 96].copy().view(np.float32)
    res = o.search(x[:10], 5)
    want = [(int(l), float(d)) for ls, ds in zip(res["label"], res["dist"]) for l, d in zip(ls, ds)]
    assert got == want
    # second half of the demo: labels 0..9 deleted, functor label % 3 == 1
    for i in range(10):
        o.mark_delete(i)
    res = o.search(x[:10], 5, filter_mod=3, filter_keep=1)
    want_f = [(int(l), float(d)) for ls, ds, c in zip(res["label"], res["dist"], res["count"]) for l, d in zip(ls[:c], ds[:c])]
    assert got_f == want_f and all(l % 3 == 1 and l >= 10 for l, _ in got_f)
    # third part: BruteforceSearch over the same rows with labels 1000 + i, bare (k=5) and with the functor (k=7)
    ob = orc.OracleBruteforce(orc.L2, 24, 500, mode=orc.TREE64)
    ob.add(x, np.arange(500, dtype=np.uint64) + 1000)
    res = ob.search(x[:10], 5)
    assert parse(bf_out) == [(int(l), float(d)) for ls, ds in zip(res["label"], res["dist"]) for l, d in zip(ls, ds)]
    res = ob.search(x[:10], 7, filter_mod=3, filter_keep=1)
    none = np.uint64(2**64 - 1)
    want_bf = [(int(l), float(d)) for ls, ds in zip(res["label"].astype(np.uint64), res["dist"]) for l, d in zip(ls, ds) if l != none]
    assert parse(bf_filt_out) == want_bf and all(l % 3 == 1 for l, _ in want_bf)
