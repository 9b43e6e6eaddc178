import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import ek_oracle
    ek_oracle.lib()
    return ek_oracle


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly (no skip) if it is missing."""
    from eigenkernel_amd import solver
    lib = solver.load_library()
    rc = lib.ek_hip_init(0)
    assert rc == 0, "ek_hip_init failed: %d" % rc
    return solver


@pytest.fixture(autouse=True)
def _default_tridiagonalisation_after_each_gpu_test(request):
    """Tests may force the two-stage tridiagonalisation on (or off) at small orders; the library's
    default crossover is restored whatever the outcome of the test."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from eigenkernel_amd import solver
        if solver._lib is not None:
            solver._lib.ek_hip_debug_set_two_stage(-1)
            solver._lib.ek_hip_debug_stedc_team(0, -1, 0)
            solver._lib.ek_hip_debug_potrf_team_profile(-1, 0)
