import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """Cores this process may really use: the GPU boxes show 256 hardware threads but
    grant a cgroup share of 16, and an OpenMP team of 256 on such a share makes the
    oracle's parallel loops tens of times slower than a team of 16."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


# before libgomp is loaded by the oracle
os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cores()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (reference build, this container)")
