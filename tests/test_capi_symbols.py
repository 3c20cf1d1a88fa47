"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/vdb_index.h declares; search entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vectordb_amd import _lib
from vectordb_amd import index as vi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "vdb_index.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vdb_[A-Za-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 38
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/vdb_index.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in vectordb_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_product_never_links_the_oracle():
    out = os.popen(f"readelf -d {_lib.LIB_PATH}").read()
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vectordb_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in src and "from oracle" not in src and "import oracle" not in src, f


@pytest.mark.skipif(_lib.lib().vdb_device_count() > 0, reason="box has a GPU")
def test_search_without_gpu_fails_loudly():
    h = vi.HierarchicalNSW(vi.L2Space(8), 10, 4, 8)
    h.add_items(np.random.default_rng(0).random((10, 8), dtype=np.float32))
    with pytest.raises(vi.VdbError) as e:
        h.to_device(0)
    assert e.value.code == _lib.VDB_ERR_NO_DEVICE
    with pytest.raises(vi.VdbError) as e:
        h.search_batch(np.zeros((1, 8), np.float32), 3)
    assert e.value.code == _lib.VDB_ERR_NO_DEVICE
    b = vi.BruteforceSearch(vi.L2Space(8), 10)
    b.add_items(np.zeros((3, 8), np.float32))
    with pytest.raises(vi.VdbError) as e:
        b.search_batch(np.zeros((1, 8), np.float32), 3)
    assert e.value.code == _lib.VDB_ERR_NO_DEVICE
