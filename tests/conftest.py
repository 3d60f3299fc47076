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
    """The product library; GPU tests fail (not skip) when it is missing."""
    from madrona_rl_envs_playground_amd import _lib
    return _lib.lib()
