"""Short seeded runs of the differential fuzzers of tools/dbg/ (their long runs are in profiles/r04_fuzz.txt): random cases, random sequences of
C-ABI calls against a host model of the handle's state, random sequences of calls on the C++ host shim.  Each exits non-zero on the first mismatch."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,args,summary", [
    ("fuzz.py", ["8", "101"], "8 cases ok"),
    ("api_fuzz.py", ["8", "101", "40"], "8 sequences ok"),
    ("host_fuzz.py", ["8", "101", "80"], "8 sequences ok"),
])
def test_seeded_differential_run(tool, args, summary):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg", tool)] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert summary in r.stdout.splitlines()[-1]
