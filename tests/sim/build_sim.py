"""
TEST INFRASTRUCTURE ONLY.  Builds the lock-step emulator variant of the product's kernel sources
(cvxpygen_amd/csrc/*.h, cpg_hip.cpp compiled with g++ -DCPG_HOST_SIM): every wavefront runs as 64
host threads that synchronise at each cross-lane primitive (cvxpygen_amd/csrc/cpg_wave.h).  It lets
the CPU-only test tier execute the real kernel logic -- executor, ADMM loop, termination and
infeasibility tests, retrieval -- through the real C-ABI.  The product never loads this library:
cvxpygen_amd.runtime only opens csrc/libcpg_hip.so unless a test passes lib_path explicitly.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, '..', '..', 'cvxpygen_amd', 'csrc')


def lib_path():
    return os.path.join(HERE, 'libcpg_sim.so')


def build(force=False):
    out = lib_path()
    deps = [os.path.join(SRC, f) for f in ('cpg_hip.cpp', 'cpg_osqp_kernel.h', 'cpg_osqp_refactor.h', 'cpg_clarabel_kernel.h', 'cpg_wave.h')]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) < os.path.getmtime(out) for d in deps):
        return out
    cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-pthread', '-DCPG_HOST_SIM', '-x', 'c++',
           os.path.join(SRC, 'cpg_hip.cpp'), '-o', out]
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    print(build(force=True))
