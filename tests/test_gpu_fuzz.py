"""A short run of the randomised parity sweep (tools/fuzz_parity.py; the 150-case log is profiles/r01_fuzz_parity.txt)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_sweep():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "18", "11"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert "18 of 18 cases passed" in out.stdout
