import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def oracle_lib():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope='session')
def sim_lib():
    """lock-step emulator build of the product kernel sources (CPU-only test tier)"""
    from sim import build_sim
    return build_sim.build()
