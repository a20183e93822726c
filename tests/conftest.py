import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: csm-1b sized CPU oracle checks (about a minute each)")


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))
    return load


@pytest.fixture(scope="session", autouse=True)
def _build_lib():
    """The C-ABI library must exist for both suites (CPU suite checks it loads and exports every symbol)."""
    from csm_hf_amd.build import build_library
    build_library()
