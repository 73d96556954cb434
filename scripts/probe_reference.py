#!/usr/bin/env python
"""Parity-pin readiness: is the REAL reference importable where this runs?  `import cvxpy, osqp, clarabel, cvxpygen` -- in the
build container and on the GPU boxes none of them is (no network, Python 3.10), so every solver-parity statement of this
repository is "versus the restatement" (DESIGN.md section 2).  Wherever they ARE importable this script runs
scripts/capture_reference.py once, which writes tests/golden/reference_outputs.npz + reference_workspace.json (DATA; commit
them) and lets the six skipped tests of tests/test_reference_outputs.py run.  Called first by scripts/gpu_final.sh and by the
CPU test tier (tests/conftest.py); prints one line either way and never fails the caller."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'reference_outputs.npz')


def probe():
    found, missing = [], []
    for name in ('cvxpy', 'osqp', 'clarabel', 'cvxpygen'):
        try:
            importlib.import_module(name)
            found.append(name)
        except Exception as e:       # ImportError, or a package that fails to initialise here
            missing.append(f'{name} ({type(e).__name__})')
    return found, missing


def main(capture: bool = True) -> int:
    """capture=False: report only (what a test session does unless CPG_CAPTURE_REFERENCE=1: a pytest run -- and every xdist worker
    of it -- must not write into the source tree or start an unbounded subprocess on its own)"""
    found, missing = probe()
    if os.path.exists(GOLD):
        print(f'reference probe: {GOLD} present (captured reference outputs): tests/test_reference_outputs.py runs')
        return 0
    if missing:
        print('reference probe: PARITY UNPINNED -- not importable here: ' + ', '.join(missing) +
              ('; importable: ' + ', '.join(found) if found else '') + ' -> no capture of the reference possible in this environment')
        return 0
    if not capture:
        print('reference probe: cvxpy / osqp / clarabel / cvxpygen import and no capture exists: run `python scripts/capture_reference.py` '
              '(or the tests with CPG_CAPTURE_REFERENCE=1) to pin the solver parity')
        return 0
    print('reference probe: cvxpy / osqp / clarabel / cvxpygen import -> capturing the reference (scripts/capture_reference.py)')
    # one writer: concurrent sessions (xdist workers, a test run next to gpu_final.sh) serialise on a lock file and re-check
    import fcntl
    with open(os.path.join(ROOT, 'tests', 'golden', '.capture.lock'), 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if os.path.exists(GOLD):
            print('reference probe: captured meanwhile by another session')
            return 0
        try:
            rc = subprocess.call([sys.executable, os.path.join(ROOT, 'scripts', 'capture_reference.py')], timeout=3600)
        except subprocess.TimeoutExpired:
            rc = -1
    print(f'reference probe: capture_reference.py exit code {rc}' + ('; commit tests/golden/reference_outputs.npz' if rc == 0 else ''))
    return 0


if __name__ == '__main__':
    sys.exit(main())
