import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-gaussian-splat-attack_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def _cpu_share() -> int:
    """Worker threads for the CPU-side torch ops of the oracle: the cores this process may run on, at most 16.  A 1-GPU box
    of the pool shows all of its host's cores (256) but grants a 16-core share: torch's default of one thread per visible
    core then spends the suite in oversubscription (round 5: the GPU suite at 11-12 minutes, most of it oracle-R on the
    CPU at a fraction of its speed).  bench.py's cpu_baseline makes the same choice."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


try:
    import torch
    torch.set_num_threads(_cpu_share())
except ImportError:                       # (the C-ABI / host-math tests do not need torch)
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_twins.npz"))
