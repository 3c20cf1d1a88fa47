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


REFERENCE = "/root/reference"


@pytest.fixture(scope="module")
def demo_reference_serializer(tmp_path_factory):
    """The same driver with the handlers instantiated over the reference's own ResponseSerializer
    (VectorCommands<ResponseSerializer>), compiled against /root/reference where it lies: build container only."""
    if not os.path.exists(os.path.join(REFERENCE, "response_serializer.hpp")):
        pytest.skip("the reference is not mounted here (it never travels to the GPU box)")
    out = tmp_path_factory.mktemp("cmdref") / "demo_ref"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-DVDB_REFERENCE_SERIALIZER", f"-I{REFERENCE}",
                           os.path.join(ROOT, "tests", "cpp_commands_demo.cpp"), "-o", str(out),
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
    assert cards[2][1] == "int 300"                                # "VCARD": matched in lower case
    assert [r for r in rep if r[0] == "vfrobnicate"][0][1].startswith("err -2 unknown command")
    assert [r for r in rep if r[0] == ""][0][1].startswith("err -1 empty command")


@pytest.mark.skipif(_lib.lib().vdb_device_count() > 0, reason="box has a GPU")
def test_vector_commands_host_side(demo):
    r = subprocess.run([demo, "host"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    _check_host_part(_replies(r.stdout))


def test_vector_commands_through_the_references_own_serializer(demo, demo_reference_serializer):
    """VectorCommands<ResponseSerializer> (response_serializer.hpp:12-58) behind a dispatch shaped as
    command_processor.hpp:29-50, :237-248: every reply, decoded by the reference's wire format, reads the same as the
    reply of the text serializer that the GPU-box test uses.  VSEARCH without a device is the no-device ERROR reply
    (there is no CPU search in the product), through that serializer too."""
    mode = "gpu" if _lib.lib().vdb_device_count() > 0 else "host"
    a = subprocess.run([demo, mode], capture_output=True, text=True, timeout=120)
    b = subprocess.run([demo_reference_serializer, mode], capture_output=True, text=True, timeout=120)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert a.stdout == b.stdout
    _check_host_part(_replies(b.stdout))
    if mode == "host":  # a VSEARCH on a box without a device: an error reply in the reference's format, not a crash
        b2 = subprocess.run([demo_reference_serializer, "gpu"], capture_output=True, text=True, timeout=120)
        assert b2.returncode == 0, b2.stderr
        searches = [x for x in _replies(b2.stdout) if x[0] == "vsearch"][1:]
        assert len(searches) == 5 and all(s[1].startswith("err -2 ") for s in searches)


@pytest.mark.gpu
def test_vector_commands_search(demo, orc, tmp_path):
    path = str(tmp_path / "docs.bin")
    r = subprocess.run([demo, "gpu", path], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = _replies(r.stdout)
    _check_host_part(rep)
    searches = [x for x in rep if x[0] == "vsearch"][1:]
    assert len(searches) == 5
    # the index behind the key, saved by the demo: the oracle runs the same searches over the same graph
    o = orc.OracleHNSW.load(path, orc.L2, 8, mode=orc.TREE64)
    o.set_ef(50)
    raw = np.fromfile(path, np.uint8)
    stride = 32 * 4 + 4 + 8 * 4 + 8
    x = raw[96:96 + 300 * stride].reshape(300, stride)[:, 132:132 + 32].copy().view(np.float32)
    ref = o.search(x[:5], 4)
    for q, s in enumerate(searches):
        assert s[1] == "int 4"
        labels = [int(v.split()[1]) for v in s[2::2]]
        dists = [float.fromhex(v.split()[1]) for v in s[3::2]]
        assert labels == [int(l) for l in ref["label"][q]]
        assert dists == [float(d) for d in ref["dist"][q]]       # (a float printed as a double: exact)
        assert 1003 not in labels                                # 1003 was VDEL'ed
        if q != 3:
            assert labels[0] == 1000 + q and dists[0] == 0.0     # the query is that stored vector
