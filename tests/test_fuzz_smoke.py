"""A short run of every randomised sweep (tools/fuzz_*.py, tests/fuzz_*.py) as part of the GPU suite: a few dozen random
cases each, fixed seeds.  The long runs are done by hand (`python tools/fuzz_spectrum.py SEED N`, ...)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWEEPS = [("tools/fuzz_spectrum.py", 101, 40), ("tools/fuzz_pipeline.py", 102, 16), ("tools/fuzz_solvers.py", 103, 16),
          ("tools/fuzz_recursive_eig.py", 104, 12), ("tests/fuzz_kernels.py", 105, 8), ("tests/fuzz_knn.py", 106, 30),
          ("tests/fuzz_assembly_icp.py", 107, 12), ("tests/fuzz_tail.py", 108, 20)]


@pytest.mark.gpu
@pytest.mark.parametrize("script,seed,cases", SWEEPS, ids=[s[0].split("/")[-1][:-3] for s in SWEEPS])
def test_randomised_sweep(script, seed, cases):
    out = subprocess.run([sys.executable, os.path.join(REPO, script), str(seed), str(cases)], cwd=REPO, capture_output=True,
                         text=True, timeout=600)
    tail = (out.stdout + out.stderr)[-1500:]
    assert out.returncode == 0, tail
    assert "done: 0 failures" in out.stdout, tail
