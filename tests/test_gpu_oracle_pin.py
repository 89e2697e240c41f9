"""The oracle-pin tests (tests/test_oracle_golden.py: the CPU oracle against the
reference's own known-answer data, README.md:97-122 and pattern_test.go:30-255)
need no GPU and are therefore not marked `gpu`; a GPU-only test run would skip
them.  This wrapper runs them on the GPU box as well, so the checker is pinned
on the very machine whose parity results it vouches for."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_pinned_by_the_reference_vectors_on_this_box():
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle_golden.py", "-q", "-x", "-p",
                        "no:cacheprovider"], cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert " passed" in p.stdout and "failed" not in p.stdout
