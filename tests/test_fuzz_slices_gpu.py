"""A bounded slice of the randomised parity sweeps (tests/fuzz_parity.py, fuzz_parity_bf.py, fuzz_build.py) under
pytest, with fixed seeds and a budget of about half a minute each, so that the randomised claim of DESIGN.md - random
graphs, ef / k, filters, spill modes, an epsilon stop-condition search per graph, random brute-force base sets, random
device builds, every launch equal to the oracle bit for bit - is part of the driver's GPU run and not only of a log the
builder wrote.  Each script runs in a child process (its own HIP context) and asserts on its own."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script, seconds, seed):
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", script), str(seconds), str(seed)], capture_output=True, text=True,
                       timeout=seconds + 150, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("done:"), last
    print(last)
    return last


@pytest.mark.parametrize("seed", [20261, 20262])
def test_fuzz_traversal_slice(seed):
    last = _run("fuzz_parity.py", 25, seed)
    graphs, launches = map(int, re.match(r"done: (\d+) graphs, (\d+) launches", last).groups())
    assert graphs >= 3 and launches >= 36


def test_fuzz_bruteforce_slice():
    last = _run("fuzz_parity_bf.py", 25, 20263)
    assert int(re.match(r"done: (\d+) base sets", last).group(1)) >= 5


def test_fuzz_device_build_slice():
    last = _run("fuzz_build.py", 30, 20264)
    assert int(re.match(r"done: (\d+) builds", last).group(1)) >= 2
