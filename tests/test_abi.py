"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/cpg_hip.h declares; the product refuses to run without it (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from cvxpygen_amd import runtime
from cvxpygen_amd.csrc import build as hipbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'cpg_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cpg_hip_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    path = hipbuild.build(min_waves_per_simd=3)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/cpg_hip.h but not exported'
    assert sorted(runtime.CpgLibrary.SYMBOLS) == names


def test_status_strings_match_osqp():
    lib = runtime.CpgLibrary(hipbuild.build(min_waves_per_simd=3))
    f = lib.L.cpg_hip_status_string
    assert f(1) == b'solved' and f(7) == b'maximum iterations reached' and f(3) == b'primal infeasible'
    for code, s in runtime.STATUS_STRINGS.items():
        assert f(code).decode() == s


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        runtime.CpgLibrary(str(tmp_path / 'libnothing.so'))


def test_product_never_references_oracle_or_emulator():
    """the oracle / emulator are test infrastructure: nothing under cvxpygen_amd/ may import them"""
    pkg = os.path.join(ROOT, 'cvxpygen_amd')
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.cpp', '.h')):
                txt = open(os.path.join(dp, fn)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, fn
                assert 'liboracle' not in txt and 'libcpg_sim' not in txt, fn
                # no second backend inside the product: the emulator lives entirely under tests/sim
                assert 'CPG_HOST_SIM' not in txt and 'pthread' not in txt and 'CPG_HIP_LIBRARY' not in txt, fn
