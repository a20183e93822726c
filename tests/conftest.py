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


@pytest.fixture(scope="session", autouse=True)
def _pin_exact_kv():
    """Round 5: the KV cache of a bf16 checkpoint is bf16 by default (`CSMModel.kv_dtype = "auto"`, the reference's own cache
    dtype).  The suites of rounds 1-4 -- bit-exact token streams against the reference's fp32-arithmetic run, bitwise invariants
    between launch shapes -- are statements about the EXACT mode (fp32 cache) and stay pinned to it; tests of the new default
    set `m.kv_dtype = "auto"` themselves (tests/test_gpu_round5.py)."""
    import torch
    from csm_hf_amd import CSMModel
    old = CSMModel.DEFAULT_KV_DTYPE
    CSMModel.DEFAULT_KV_DTYPE = torch.float32
    yield
    CSMModel.DEFAULT_KV_DTYPE = old
