"""Writes tests/golden/device_build_golden.json: the SHA-256 of the file a device build of a fixed data set saves
(run on an MI355X from the repo root: python tests/golden/make_device_build_golden.py).  The data, the parameters and the
build are those of tests/test_build_gpu.py::test_device_build_is_deterministic_and_matches_its_golden."""
import hashlib
import json
import os
import pathlib
import sys
import tempfile

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
os.environ["VDB_BUILD_HOST_BASE"] = "64"

from tests.test_build_gpu import _device_built_bytes  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    a, h = _device_built_bytes(pathlib.Path(d), "g.bin")
out = {"sha256": hashlib.sha256(a).hexdigest(), "bytes": len(a), "n": 6000, "dim": 16, "M": 8, "ef_construction": 48,
       "data": "datasets.clustered(6000, 16, 81)", "host_base": 64, "n_threads": 1}
dst = os.environ.get("GOLDEN_OUT", str(pathlib.Path(__file__).with_name("device_build_golden.json")))
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print(out)
