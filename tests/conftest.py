import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture
def sequential_coarse():
    """forward_solve step by step (options.coarse_solve = 'sequential') for the tests of the chain kernels themselves: levels that
    qualify for the time-parallel form would otherwise never reach them. The oracle side: block_solve=False."""
    from pymgrit_amd.core.options import options
    options.coarse_solve = "sequential"
    try:
        yield
    finally:
        options.reset("coarse_solve")      # back to environment / default (an assignment would pin the value for later tests)


# Order of the suite (matters under `pytest -x`): the C ABI and the single-process oracle comparisons first, every test
# that starts other processes or rank threads last, so a fault of the multi-process harness can never hide a parity test.
_ORDER = ["test_hip_abi", "test_hip_parity", "test_hip_heat2d", "test_hip_fuzz", "test_hip_bdf", "test_advection_sc",
          "test_hip_output", "test_hip_forcing", "test_hip_plan", "test_hip_level_fusion", "test_hip_user_transfer", "test_at_mgrit", "test_local_conv"]
_LAST = ["test_hip_distributed", "test_hip_exchange_fuzz", "test_hip_bench_cli", "test_hip_full_size", "test_hip_rccl"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _ORDER:
            return (0, _ORDER.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=key)         # stable: the order inside a file is kept
