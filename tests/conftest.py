import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """TEST-ONLY CPU oracle (oracle/*.c), compiled with gcc on first use."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hip_lib():
    """The product library (built with hipcc if the in-tree .so is missing or stale);
    tests fail, not skip, when it cannot be had."""
    from madrona_rl_envs_playground_amd import _lib
    if not os.environ.get("MRL_ENVS_LIB"):
        _lib.build()
    return _lib.lib()
