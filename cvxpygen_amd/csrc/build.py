"""
Builds libcpg_hip.so (the C-ABI library of include/cpg_hip.h) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repository snapshot.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['cpg_hip.cpp']
HEADERS = ['cpg_osqp_kernel.h', 'cpg_osqp_refactor.h', 'cpg_osqp_resident.h', 'cpg_wave.h', 'cpg_wave_gfx950.h', os.path.join('..', '..', 'include', 'cpg_hip.h'), 'cpg_clarabel_kernel.h']


def lib_path(tag: str = '') -> str:
    return os.path.join(HERE, f'libcpg_hip{tag}.so')


def _stale(out: str) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + ['build.py'])


def build(min_waves_per_simd: int = 3, tag: str = '', force: bool = False, verbose: bool = False,
          extra_flags=()) -> str:
    """min_waves_per_simd bounds the VGPR budget of the solve kernels (512 / waves per SIMD)."""
    out = lib_path(tag)
    if not force and not _stale(out):
        return out
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-value',
           f'-DCPG_MIN_WAVES_PER_SIMD={int(min_waves_per_simd)}', *extra_flags,
           '-o', out] + [os.path.join(HERE, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    return out


if __name__ == '__main__':
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    tag = sys.argv[2] if len(sys.argv) > 2 else ''
    print(build(w, tag, force=True, verbose=True))
