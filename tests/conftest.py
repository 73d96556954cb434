import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


# the lock-step emulator starts every workgroup with NaN bit patterns in its LDS instead of zeros (tests/sim/fake_hip): a slot read
# before anybody wrote it shows here, not on the GPU
os.environ.setdefault('CPG_SIM_LDS_POISON', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_sessionstart(session):
    # parity-pin readiness: one line that says whether the real reference imports here ("PARITY UNPINNED" in
    # this environment).  A test session only REPORTS; with CPG_CAPTURE_REFERENCE=1 (or scripts/probe_reference.py /
    # scripts/gpu_final.sh run by hand) it also captures the reference's outputs -- under a file lock -- before the tests that
    # replay them are collected
    try:
        sys.path.insert(0, os.path.join(ROOT, 'scripts'))
        import probe_reference
        probe_reference.main(capture=os.environ.get('CPG_CAPTURE_REFERENCE', '0') == '1')
    except Exception as e:       # never in the way of the test run
        print(f'reference probe failed: {e}')


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
