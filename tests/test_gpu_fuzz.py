"""A fixed slice of the randomised campaigns of tools/fuzz_*.py (the full runs are recorded in profiles/r03_final/fuzz_parity.txt):
random scene shapes x model variants x arities x options against the CPU restatement — one rank, several ranks on one GPU, the
frame-windowed driver.  The seeds are fixed, so the slice is the same on every run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(name, *args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name), *map(str, args)], cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_random_problems_follow_the_oracle(built):
    out = run_tool("fuzz_parity.py", 60, 7000)
    assert "60 cases, 0 failures" in out, out[-3000:]


def test_random_problems_on_several_ranks_follow_the_oracle(built):
    out = run_tool("fuzz_multirank.py", 12, 12000)
    assert "12 multi-rank cases, 0 failures" in out, out[-3000:]


def test_random_window_schedules_follow_the_oracle(built):
    out = run_tool("fuzz_windowed.py", 60, 15000)
    assert "60 windowed cases, 0 failures" in out, out[-3000:]
