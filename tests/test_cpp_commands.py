"""the vector commands a maintainer adds to the reference's CommandProcessor
(vectordb_amd/cpp/vdb_commands.hpp), driven as the server would drive them."""
import os
import subprocess

import numpy as np
import pytest

from vectordb_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    out = tmp_path_factory.mktemp("cmd") / "demo"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp_commands_demo.cpp"), "-o", str(out),
                           f"-L{libdir}", "-lvdb_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return str(out)


def _replies(text):
    out, cur = [], None
    for line in text.splitlines():
        if line.startswith("> "):
            cur = [line[2:]]
            out.append(cur)
        else:
            cur.append(line)
    return out


def _check_host_part(rep):
    adds = [r for r in rep if r[0] == "vadd"]
    assert all(r[1] == "int 1" for r in adds[:300])
    assert adds[300][1].startswith("err -1 Vector has the wrong dimension")
    assert adds[301][1].startswith("err -1 Invalid label") and adds[302][1].startswith("err -1 Invalid vector component")
    cards = [r for r in rep if r[0] == "vcard"]
    assert cards[0][1] == "int 300" and cards[1][1] == "int 0"   # 300 > initial capacity 64: grew by resizeIndex
    dels = [r for r in rep if r[0] == "vdel"]
    assert dels[0][1] == "int 1" and dels[1][1] == "int 0"
    assert [r for r in rep if r[0] == "vsearch"][0][1] == "int 0"


@pytest.mark.skipif(_lib.lib().vdb_device_count() > 0, reason="box has a GPU")
def test_vector_commands_host_side(demo):
    r = subprocess.run([demo, "host"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    _check_host_part(_replies(r.stdout))


@pytest.mark.gpu
def test_vector_commands_search(demo):
    r = subprocess.run([demo, "gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = _replies(r.stdout)
    _check_host_part(rep)
    searches = [x for x in rep if x[0] == "vsearch"][1:]
    assert len(searches) == 5
    for q, s in enumerate(searches):
        assert s[1] == "int 4"
        labels = [int(v.split()[1]) for v in s[2::2]]
        dists = [float.fromhex(v.split()[1]) for v in s[3::2]]
        assert dists == sorted(dists) and len(set(labels)) == 4 and 1003 not in labels   # 1003 was VDEL'ed
        if q != 3:
            assert labels[0] == 1000 + q and dists[0] == 0.0   # the query is that stored vector
