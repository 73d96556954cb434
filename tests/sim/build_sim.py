"""
TEST INFRASTRUCTURE ONLY.  Builds the lock-step emulator of the product's kernel SOURCES: g++ compiles
cvxpygen_amd/csrc/cpg_hip.cpp unchanged, with
  * tests/sim/cpg_wave_sim.h force-included first -- host versions of the wavefront primitives of
    csrc/cpg_wave_gfx950.h (every wavefront = 64 host threads that meet at a barrier in each
    cross-lane primitive); it defines that header's include guard, so the product header is skipped;
  * tests/sim/fake_hip/hip/hip_runtime.h standing in for the HIP runtime (malloc / memcpy, launches
    run the workgroups one after the other).
This lets the CPU-only test tier execute the real kernel logic -- executors, ADMM loop, termination and
infeasibility tests, factorisation, retrieval -- through the real C-ABI and Python runtime.  Nothing
under cvxpygen_amd/ refers to the emulator; tests hand the library to BatchSolver(lib_path=...).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cvxpygen_amd import codegen  # noqa: E402  (kernel configuration macros of the product build)

SIM_HEADERS = [os.path.join(HERE, 'cpg_wave_sim.h'), os.path.join(HERE, 'fake_hip', 'hip', 'hip_runtime.h'),
               os.path.abspath(__file__)]


def _gxx_cmd(src, defs, out):
    return ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-pthread', '-I', os.path.join(HERE, 'fake_hip'),
            '-include', os.path.join(HERE, 'cpg_wave_sim.h'), '-x', 'c++', src, *defs, '-o', out]


def lib_path():
    return os.path.join(HERE, 'libcpg_sim.so')


def build(force=False, verbose=False):
    """the generic (table-driven) library"""
    out = lib_path()
    if force and os.path.exists(out):
        os.remove(out)
    src, deps = codegen.source_files()
    return codegen.compile_if_stale(_gxx_cmd(src, [], out), out, deps + SIM_HEADERS, verbose)


def build_family(plan, out_dir, name='family', verbose=False, **kw):
    """emulator build of the family-specialised library (generated straight-line executor)"""
    hdr, defs = codegen.family_library_defs(plan, out_dir, name, **kw)
    out = os.path.join(out_dir, f'libcpg_{name}_sim.so')
    src, deps = codegen.source_files()
    return codegen.compile_if_stale(_gxx_cmd(src, defs, out), out, codegen.generated_headers(defs) + deps + SIM_HEADERS, verbose)


def build_streamed_family(plan, out_dir, name='family', verbose=False):
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f'libcpg_{name}_streamed_sim.so')
    src, deps = codegen.source_files()
    return codegen.compile_if_stale(_gxx_cmd(src, codegen.streamed_library_defs(plan), out), out,
                                    deps + SIM_HEADERS, verbose)


def build_conic_family(cplan, out_dir, name='family', verbose=False):
    """emulator build of a conic family library (generated executor of the substitution program)"""
    os.makedirs(out_dir, exist_ok=True)
    defs, hdrs = codegen.conic_library_defs(cplan, out_dir, name)
    out = os.path.join(out_dir, f'libcpg_{name}_conic_sim.so')
    src, deps = codegen.source_files()
    return codegen.compile_if_stale(_gxx_cmd(src, defs, out), out, hdrs + deps + SIM_HEADERS, verbose)


if __name__ == '__main__':
    print(build(force=True, verbose=True))
