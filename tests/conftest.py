import os
import sys

import pytest

# The oracle's NumPy / torch-CPU code runs on OpenMP / BLAS thread pools whose workers spin while they wait.  On a box whose
# cores are shared (the 8-CPU build container next to a compile, a CI runner) spinning pools turn a 16-second CPU suite
# into 2 - 18 minutes (measured: the same 73 tests, 16 s alone, 134 s and 1,103 s beside other work).  Passive waiting
# costs nothing measurable here and must be chosen before numpy / torch load their runtimes.
for _k, _v in (("OMP_WAIT_POLICY", "PASSIVE"), ("KMP_BLOCKTIME", "0"), ("GOMP_SPINCOUNT", "0")):
    os.environ.setdefault(_k, _v)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
