import os
import sys

import numpy as np
import pytest
try:
    import torch  # noqa: F401  (before anything loads libpyfocusr_hip.so: torch and the library must share ONE HIP runtime -
    #               whichever copy of libamdhip64 is loaded first serves both, and torch.cuda only comes up on its own copy)
except ImportError:  # the oracle / host-logic tests need no torch; the distributed tests skip themselves without it
    torch = None

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


_cache = {}


def load_golden(name):
    if name not in _cache:
        with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
            _cache[name] = {k: z[k] for k in z.files}
    return _cache[name]


@pytest.fixture(scope="session")
def golden():
    return load_golden


def align_signs(a, b):
    """Flip columns of `a` so each has a non-negative inner product with `b`'s."""
    s = np.sign(np.sum((a - a.mean(0)) * (b - b.mean(0)), axis=0))
    s[s == 0] = 1.0
    return a * s[None, :]
