import os
import sys

import pytest

# test_gpu_local_tiles.py runs several tiles of one decomposition in ONE process, one HIP stream each, whose kernels wait for
# each other's flags (the peer halo transport).  HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues and two
# streams that share one run their kernels in submission order -- a waiting kernel in front of the one it waits for would sit
# there until its 3 s timeout.  Ask for enough queues before the runtime initialises (one process per GPU never needs this).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle
