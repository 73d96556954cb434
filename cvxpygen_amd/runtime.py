"""
Host side of the batched solver: builds the device plan of a problem family and drives
libcpg_hip.so (include/cpg_hip.h) through ctypes -- "host code stays Python calling HIP through a
thin C-ABI / ctypes layer (no PyTorch)".

This is the batched counterpart of the reference's L6/L7 layers: the Python shim
`cpg_solve` (`cvxpygen/templates/cpg_solver.py.jinja2:40-117`) + the pybind11 module
`cpg_module.solve(upd, par)` (`cvxpygen/utils.py:1194-1270`).  One `BatchSolver` replaces the
static workspace the reference emits per problem family (`cvxpygen/utils.py:470-689`).

There is no CPU fallback: if the HIP library is missing or no GPU is present the constructor
raises.  `lib_path` selects a family-specialised build of the same library (cvxpygen_amd/codegen.py).
"""

from __future__ import annotations

import ctypes as C
import os
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from . import osqp_setup as _setup
from . import solve_program as _sp
from .canon_builder import canon_lu
from .descriptor import CPG_INF, FamilyDescriptor

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u16p = C.POINTER(C.c_uint16)
_i8p = C.POINTER(C.c_int8)

STATUS_STRINGS = {1: 'solved', 2: 'solved inaccurate', 3: 'primal infeasible',
                  4: 'primal infeasible inaccurate', 5: 'dual infeasible',
                  6: 'dual infeasible inaccurate', 7: 'maximum iterations reached',
                  9: 'problem non convex', 11: 'unsolved', -2: 'needs refactorization'}
STATUS_NEEDS_REFACTOR = -2

# OSQP settings the generated shim has no setter for: the defaults of the OSQP library the solver is linked
# with, restored by every cpg_solve (osqp_set_default_settings, cvxpygen/solvers/osqp.py:100-101).  The reference
# requires osqp >= 1.0 (pyproject.toml:26; its emitted calls are the 1.0 API): rho adapted every 50 iterations,
# tolerance 5, duality-gap test.  `build_options` / generate_code(osqp_build_options=...) override.
BUILD_OPTIONS = ('adaptive_rho', 'adaptive_rho_interval', 'adaptive_rho_tolerance', 'check_dualgap',
                 # raw-handle protocol of the C-ABI (include/cpg_hip.h): a shared-factor solve WITHOUT a linked per-instance handle
                 # flags the instances whose rho would change (status -2) instead of being refused
                 'flag_rho_changes')
BUILD_OPTION_DEFAULTS = {'adaptive_rho': 1, 'adaptive_rho_interval': 50, 'adaptive_rho_tolerance': 5.0, 'check_dualgap': 1}
# the other reading of the reference's default (a solver generated against an OSQP whose codegen never adapts rho)
BUILD_OPTIONS_FIXED_RHO = {'adaptive_rho': 0, 'check_dualgap': 0}

# cvxpy-style aliases of the reference (`stgs_translation`, cvxpygen/solvers/osqp.py:110,
# cvxpygen/solvers/_interface.py:196-199)
SETTING_ALIASES = {'warm_start': 'warm_starting'}
SETTINGS_ENABLED = ['max_iter', 'eps_abs', 'eps_rel', 'eps_prim_inf', 'eps_dual_inf',
                    'scaled_termination', 'check_termination', 'warm_starting']


class _Program(C.Structure):
    _fields_ = [('n_chunks', C.c_int32), ('n_steps', C.c_int32), ('hdr', _ip), ('rows', _u16p),
                ('vals', _dp), ('cols', _u16p)]


class _Ragged(C.Structure):
    _fields_ = [('n_chunks', C.c_int32), ('nnz', C.c_int32), ('ctab', _ip),
                ('desc', C.POINTER(C.c_uint32)), ('vals', _dp), ('cols', _u16p)]


class _Csr(C.Structure):
    _fields_ = [('rows', C.c_int32), ('nnz', C.c_int32), ('ptr', _ip), ('idx', _ip), ('val', _dp)]


class _Family(C.Structure):
    _fields_ = [('n', C.c_int32), ('m', C.c_int32), ('n_eq', C.c_int32), ('is_maximization', C.c_int32),
                ('sigma', C.c_double), ('alpha', C.c_double), ('rho', C.c_double),
                ('D', _dp), ('E', _dp), ('c', C.c_double), ('ctype', _i8p),
                ('n_slots', C.c_int32), ('fpos', _u16p), ('n_vary_x', C.c_int32), ('n_vary_z', C.c_int32),
                ('kkt', _Program), ('A_rows', _Program), ('P_rows', _Program), ('At_rows', _Program),
                ('kkt_ragged', _Ragged),
                ('n_prim', C.c_int32), ('prim_idx', _ip), ('n_dual', C.c_int32), ('dual_idx', _ip),
                ('ord', _ip)]


class _Refactor(C.Structure):
    _fields_ = [('nnzP', C.c_int32), ('nnzA', C.c_int32), ('nnzL', C.c_int32), ('scaling_iters', C.c_int32),
                ('Ap', _ip), ('Ai', _ip), ('Arp', _ip), ('Aent', _ip), ('Acol', _ip),
                ('Pp', _ip), ('Pi', _ip), ('Prp', _ip), ('Pent', _ip), ('Pcol', _ip),
                ('Lcol', _ip), ('ksrc_kind', _ip), ('ksrc_idx', _ip),
                ('fac_chunks', C.c_int32), ('fac_triples', C.c_int32), ('fac_ctab', _ip),
                ('fac_task', C.POINTER(C.c_uint32)), ('fac_len', C.POINTER(C.c_uint32)),
                ('fac_a', C.POINTER(C.c_uint32)), ('fac_b', C.POINTER(C.c_uint32)), ('fac_k', C.POINTER(C.c_uint32)),
                ('sol_chunks', C.c_int32), ('sol_nnz', C.c_int32), ('sol_slots', C.c_int32),
                ('sol_ctab', _ip), ('sol_desc', C.POINTER(C.c_uint32)), ('sol_cols', _u16p),
                ('sol_kind', _ip), ('sol_idx', _ip), ('sol_fpos', _u16p),
                ('np_var', C.c_int32), ('P_base', _dp), ('A_base', _dp), ('q_base', _dp), ('u_base', _dp),
                ('d_base', C.c_double),
                ('map_P', _Csr), ('map_A', _Csr), ('map_q', _Csr), ('map_u', _Csr), ('map_d', _Csr),
                ('q_setup', _dp),
                ('shared_mats', C.c_int32), ('Ps', _dp), ('As', _dp), ('D', _dp), ('E', _dp), ('c', C.c_double)]


class _RowsProgram(C.Structure):
    _fields_ = [('n_chunks', C.c_int32), ('nnz', C.c_int32), ('ctab', _ip), ('desc', C.POINTER(C.c_uint32)), ('cols', _u16p),
                ('ent', _ip)]


class _Resident(C.Structure):
    _fields_ = [('nnzX', C.c_int32), ('fac_chunks', C.c_int32), ('fac_triples', C.c_int32), ('f_ctab', _ip),
                ('f_task', C.POINTER(C.c_uint32)), ('f_len', C.POINTER(C.c_uint32)),
                ('f_a', C.POINTER(C.c_uint32)), ('f_b', C.POINTER(C.c_uint32)), ('f_k', C.POINTER(C.c_uint32)),
                ('sol_chunks', C.c_int32), ('sol_nnz', C.c_int32), ('sol_slots', C.c_int32),
                ('sol_ctab', _ip), ('sol_desc', C.POINTER(C.c_uint32)), ('sol_cols', _u16p),
                ('sol_kind', _ip), ('sol_idx', _ip), ('sol_lcol', _ip),
                ('rows_A', _RowsProgram), ('rows_P', _RowsProgram), ('rows_At', _RowsProgram),
                ('out_ax', C.c_int32), ('out_px', C.c_int32), ('out_aty', C.c_int32)]


class _Gradient(C.Structure):
    _fields_ = [('NP', C.c_int32), ('Pcolidx', _ip), ('Acolidx', _ip), ('tptr', _ip), ('tkind', _ip),
                ('tidx', _ip), ('tcoef', _dp)]


class _Update(C.Structure):
    _fields_ = [('np_var', C.c_int32), ('q_base', _dp), ('u_base', _dp), ('d_base', C.c_double),
                ('map_q', _Csr), ('map_u', _Csr), ('map_d', _Csr)]


def default_lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libcpg_hip.so')


class CpgLibrary:
    """ctypes view of the C-ABI declared in include/cpg_hip.h."""

    SYMBOLS = ['cpg_hip_device_count', 'cpg_hip_create_osqp', 'cpg_hip_create_clarabel', 'cpg_hip_destroy', 'cpg_hip_last_error',
               'cpg_hip_status_string', 'cpg_hip_set_default_settings', 'cpg_hip_set_setting',
               'cpg_hip_get_setting', 'cpg_hip_set_build_option', 'cpg_hip_set_handover', 'cpg_hip_last_phase_ms', 'cpg_hip_set_update', 'cpg_hip_set_refactor', 'cpg_hip_set_resident', 'cpg_hip_set_gradient', 'cpg_hip_gradient_batch',
               'cpg_hip_solve_batch',
               'cpg_hip_solve_batch_device', 'cpg_hip_solve_batch_state', 'cpg_hip_solve_batch_device_state', 'cpg_hip_solve_batches_pipelined', 'cpg_hip_host_malloc',
               'cpg_hip_host_free', 'cpg_hip_synchronize', 'cpg_hip_get_stream', 'cpg_hip_last_kernel_ms',
               'cpg_hip_set_launch', 'cpg_hip_set_program_placement', 'cpg_hip_malloc', 'cpg_hip_free', 'cpg_hip_memcpy_h2d',
               'cpg_hip_memcpy_d2h']

    def __init__(self, path: Optional[str] = None):
        path = path or default_lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f'HIP extension {path} not found: build it with `python -m cvxpygen_amd.csrc.build` '
                '(there is no CPU fallback)')
        self.path = path
        L = C.CDLL(path)
        self.L = L
        L.cpg_hip_last_error.restype = C.c_char_p
        L.cpg_hip_status_string.restype = C.c_char_p
        L.cpg_hip_status_string.argtypes = [C.c_int32]
        L.cpg_hip_device_count.argtypes = [_ip]
        L.cpg_hip_create_osqp.argtypes = [C.POINTER(_Family), C.c_int, C.POINTER(C.c_void_p)]
        L.cpg_hip_destroy.argtypes = [C.c_void_p]
        L.cpg_hip_set_default_settings.argtypes = [C.c_void_p]
        L.cpg_hip_set_setting.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.cpg_hip_get_setting.argtypes = [C.c_void_p, C.c_char_p, _dp]
        L.cpg_hip_set_build_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.cpg_hip_set_handover.argtypes = [C.c_void_p, C.c_void_p]
        L.cpg_hip_last_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        L.cpg_hip_set_update.argtypes = [C.c_void_p, C.POINTER(_Update)]
        L.cpg_hip_set_refactor.argtypes = [C.c_void_p, C.POINTER(_Refactor)]
        L.cpg_hip_set_resident.argtypes = [C.c_void_p, C.POINTER(_Refactor), C.POINTER(_Resident)]
        L.cpg_hip_set_gradient.argtypes = [C.c_void_p, C.POINTER(_Gradient)]
        L.cpg_hip_gradient_batch.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp, _dp]
        L.cpg_hip_solve_batch.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp]
        L.cpg_hip_solve_batch_device.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 8
        L.cpg_hip_solve_batch_state.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp]
        L.cpg_hip_solve_batch_device_state.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 10
        L.cpg_hip_synchronize.argtypes = [C.c_void_p]
        L.cpg_hip_solve_batches_pipelined.argtypes = [C.c_void_p, C.c_int64, C.c_int32, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp]
        L.cpg_hip_host_malloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.cpg_hip_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.cpg_hip_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.cpg_hip_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.cpg_hip_set_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.cpg_hip_set_program_placement.argtypes = [C.c_void_p, C.c_int]
        L.cpg_hip_malloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.cpg_hip_free.argtypes = [C.c_void_p, C.c_void_p]
        L.cpg_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.cpg_hip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]

    def check(self, rc: int, what: str = '') -> None:
        if rc != 0:
            msg = self.L.cpg_hip_last_error().decode()
            raise RuntimeError(f'{what} failed ({rc}): {msg}')

    def device_count(self) -> int:
        n = C.c_int32(0)
        self.check(self.L.cpg_hip_device_count(C.byref(n)), 'cpg_hip_device_count')
        return n.value


def _d(a):
    return a.ctypes.data_as(_dp)


def _program_struct(p: _sp.PackedProgram, keep: list) -> _Program:
    hdr = np.ascontiguousarray(p.hdr, dtype=np.int32)
    rows = np.ascontiguousarray(p.rows, dtype=np.uint16)
    vals = np.ascontiguousarray(p.vals, dtype=np.float64)
    cols = np.ascontiguousarray(p.cols, dtype=np.uint16)
    keep += [hdr, rows, vals, cols]
    return _Program(p.n_chunks, p.steps, hdr.ctypes.data_as(_ip), rows.ctypes.data_as(_u16p),
                    _d(vals), cols.ctypes.data_as(_u16p))


def _csr_struct(M: sp.csr_matrix, keep: list) -> _Csr:
    M = sp.csr_matrix(M)
    M.sort_indices()
    ptr = np.ascontiguousarray(M.indptr, dtype=np.int32)
    idx = np.ascontiguousarray(M.indices, dtype=np.int32)
    val = np.ascontiguousarray(M.data, dtype=np.float64)
    keep += [ptr, idx, val]
    return _Csr(M.shape[0], int(M.nnz), ptr.ctypes.data_as(_ip), idx.ctypes.data_as(_ip), _d(val))


# ------------------------------------------------------------------------------------------------
@dataclass
class FamilyPlan:
    """Code-generation-time product for one OSQP problem family: scaling, factor, solve program
    and the device ordering (entries depending on user parameters first)."""
    desc: FamilyDescriptor
    osqp: _setup.OsqpPlan   # scaling, row classes and the factor on the STRUCTURAL pattern (per-instance matrix parameters)
    ordx: np.ndarray        # device position -> canonical x index
    ordz: np.ndarray        # device position -> canonical row
    posx: np.ndarray        # canonical x index -> device position
    posz: np.ndarray
    n_vary_x: int
    n_vary_z: int
    kkt: _sp.PackedProgram
    kkt_ragged: _sp.RaggedProgram
    A_rows: _sp.PackedProgram
    P_rows: _sp.PackedProgram
    At_rows: _sp.PackedProgram
    prim_idx: np.ndarray
    dual_idx: np.ndarray
    stats: Dict[str, float] = field(default_factory=dict)
    # the same workspace factored on the numerically non-zero pattern (osqp_setup.setup(prune=True)): what every
    # path uses whose matrices are the code-generation-time ones -- the shared solve program and the
    # per-instance factors of shared-matrix mode (rho adaptation, rows that changed class)
    osqp_shared: Optional[_setup.OsqpPlan] = None
    # what the solve program was packed from: its phases (logical entries = device positions before the bank-aware numbering)
    # and that numbering -- codegen packs the same phases once more for the squad executor (pack_ragged(team=W))
    phases: Optional[list] = None
    slot_perm: Optional[np.ndarray] = None
    kkt_squad: Optional[_sp.RaggedProgram] = None      # the same program packed for the squad executor (None: it does not fit / is off)


def _header_defines(path: str) -> Dict[str, str]:
    """`#define NAME value` lines of a generated header (the WHOLE file, line by line: a long comment or table in front of
    the defines must not hide them) and its `// NAME value ...` record comments"""
    out: Dict[str, str] = {}
    try:
        with open(path) as f:
            for ln in f:
                if ln.startswith('#define CPG_'):
                    parts = ln.split(None, 2)
                    if len(parts) == 3 and '(' not in parts[1]:
                        out[parts[1]] = parts[2].strip()
                elif ln.startswith('// CPG_'):
                    parts = ln[3:].split(None, 1)
                    if len(parts) == 2:
                        out['//' + parts[0]] = parts[1].strip()
    except OSError:
        pass
    return out


def _instance_fingerprints(lib_path: str, stem: str = 'cpg_instance', prefix: str = 'GENI'):
    """CPG_GENI_FINGERPRINT of the generated instance headers next to a library (what it was compiled from); with
    stem 'cpg_resident' / prefix 'GENR' the same for the resident executors (codegen.resident_header)"""
    import glob
    import re
    out = set()
    for h in glob.glob(os.path.join(os.path.dirname(os.path.abspath(lib_path)), stem + '_*.h')):
        m = re.fullmatch(r'(\d+)u', _header_defines(h).get(f'CPG_{prefix}_FINGERPRINT', ''))
        if m:
            out.add(int(m.group(1)))
    return out


def _team_widths(lib_path: str):
    """(CPG_GENT_W, CPG_GENT_MAX_GROUP_ROWS, level groups, dimensions) of the generated team headers next to a library: what their
    plans were built with; dimensions = (N, M, NNZA, NNZP, NNZL) of the family the header was generated for"""
    import glob
    out = set()
    for h in glob.glob(os.path.join(os.path.dirname(os.path.abspath(lib_path)), 'cpg_team_*.h')):
        d = _header_defines(h)
        if 'CPG_GENT_W' not in d:
            continue
        gr = d.get('//CPG_GENT_GROUPS')
        groups = tuple(tuple(int(v) for v in it.split('-')) for it in gr.split()) if gr else None
        dims = tuple(int(d[k]) if k in d else None for k in ('CPG_GENT_N', 'CPG_GENT_M', 'CPG_GENT_NNZA', 'CPG_GENT_NNZP', 'CPG_GENT_NNZL'))
        out.add((int(d['CPG_GENT_W']), int(d['CPG_GENT_MAX_GROUP_ROWS']) if 'CPG_GENT_MAX_GROUP_ROWS' in d else None, groups, dims))
    return out


def build_family_plan(desc: FamilyDescriptor, ordering: str = 'mindeg', merge: bool = True,
                      setup_settings: Optional[Dict[str, float]] = None, bank_layout: bool = True) -> FamilyPlan:
    t0 = time.time()
    n, m, n_eq = desc.n_var, desc.m, desc.n_eq
    # the shared-factor kernel relies on the two row kinds of the reference's OSQP canonical form
    Ml, Mu = sp.csr_matrix(desc.maps['l']), sp.csr_matrix(desc.maps['u'])
    if Ml.shape[0] != n_eq or (abs(Ml - Mu[:n_eq]) > 0).nnz != 0:
        raise NotImplementedError('rows with finite l != u are not produced by the OSQP canonical '
                                  'form of the reference and are not supported')
    canon0 = desc.default_canon()
    l0, u0 = canon_lu(desc, canon0)
    plan = _setup.setup(desc.P, canon0['q'], desc.A, l0, u0, settings=setup_settings,
                        ordering=ordering)
    prune = os.environ.get('CPG_PRUNE', '1') != '0'
    splan = _setup.setup(desc.P, canon0['q'], desc.A, l0, u0, settings=setup_settings, ordering=ordering, prune=True) \
        if prune and ((plan.Ax == 0.0).any() or (plan.Px == 0.0).any()) else plan
    NP = desc.NP
    vary_q = np.diff(sp.csr_matrix(desc.maps['q'])[:, :NP].tocsr().indptr) > 0
    vary_u = np.diff(Mu[:, :NP].tocsr().indptr) > 0
    ordx = np.concatenate([np.nonzero(vary_q)[0], np.nonzero(~vary_q)[0]]).astype(np.int64)
    ordz = np.concatenate([np.nonzero(vary_u)[0], np.nonzero(~vary_u)[0]]).astype(np.int64)
    posx = np.empty(n, dtype=np.int64); posx[ordx] = np.arange(n)
    posz = np.empty(m, dtype=np.int64); posz[ordz] = np.arange(m)
    devpos = np.concatenate([posx, n + posz])
    N = n + m
    phases = _sp.compile_ldl(N, splan.Lp, splan.Li, splan.Lx, splan.D, splan.perm, merge=merge,
                             devpos=devpos)
    pi = None
    bank_stats = {}
    entry_order = False
    if bank_layout:
        # bank-aware numbering of the work-vector slots (cvxpygen_amd/slot_layout.py): the entries keep their
        # region -- x / rows, parameter-dependent first, rows additionally by class so that 64-row slots of
        # one class stay uniform -- and the spill slots of merged phases are renumbered freely
        from . import slot_layout as _sl
        rg0 = _sp.pack_ragged(phases, N, balanced=True)
        pad0 = _sp.padded_offsets_fit(rg0, N)
        region = np.full(rg0.n_slots + (_sp.GEN_EXTRA_SLOTS if pad0 else 0), 99, dtype=np.int64)
        region[rg0.n_slots:] = 1000 + np.arange(len(region) - rg0.n_slots)      # dummy / zero slots stay put
        region[:n] = np.where(np.arange(n) < int(vary_q.sum()), 0, 1)
        ctz = np.asarray(plan.constr_type)[ordz]
        region[n:N] = np.where(np.arange(m) < int(vary_u.sum()), 10, 20 + (ctz + 1))
        stores = _sl.store_groups((rg0.desc & 0xFFFF).astype(np.int64), 0xFFFF)
        # ... together with the order of every row's entries among the row's (lane, step) cells (round 6: the numbering alone
        # left 0.85 extra LDS cycles per 32-lane gather group; with the entries free to choose their step, 0.14)
        entry_order = os.environ.get('CPG_ENTRY_ORDER', '1') != '0'
        sweeps = int(os.environ.get('CPG_BANK_SWEEPS', 60))
        # the squad executor (codegen.emit_squad_program: the same phases packed for a team of W wavefronts, coefficients in
        # registers; placement 3, an experiment that lost against the LDS-resident program: HISTORY.md round 6) shares the
        # numbering -- chosen for the LDS executor's gathers -- and gets the order of its own entries for it (16-byte reads of a
        # pair array: four groups of 16 lanes, slots collide modulo 16; stores in groups of 8 lanes, modulo 8)
        from . import codegen as _cg
        rq0 = None
        if entry_order and _cg.SQUAD:
            rq0 = _cg.pack_squad(phases, N)
            if not _cg.squad_fits(desc, rq0, _cg.SQUAD_WAVES):
                rq0 = None
        if rq0 is not None:
            if len(region) == rg0.n_slots:                 # (the squad's work vectors always carry the dummy slots and the zero slot)
                region = np.concatenate([region, 1000 + np.arange(_sp.GEN_EXTRA_SLOTS)])
            pi, eperm, pad_slot, c0, c1 = _sl.optimise_entries(rg0, region, pad0, seed=0, stores=stores, sweeps=sweeps)
            stores_q = _sl.store_groups((rq0.desc & 0xFFFF).astype(np.int64), 0xFFFF, group=8)
            _, eperm_q, pad_q, cq0, cq1 = _sl.optimise_entries(rq0, region, True, seed=0, stores=stores_q, sweeps=sweeps, keep_fresh=False,
                                                               lane_group=_sl.B128_LANE_GROUP, mod=16, store_mod=8, pi0=pi, slot_moves=False)
            bank_stats.update(squad_conflict_cycles_natural=int(cq0), squad_conflict_cycles=int(cq1))
        elif entry_order:
            pi, eperm, pad_slot, c0, c1 = _sl.optimise_entries(rg0, region, pad0, seed=0, stores=stores, sweeps=sweeps)
        else:
            pi, c0, c1 = _sl.optimise(_sp.gathered_slots(rg0, idle_zero=pad0), region, seed=0, stores=stores, sweeps=sweeps)
        pi_all = pi
        pi = pi[:rg0.n_slots]
        bank_stats.update(bank_conflict_cycles_natural=int(c0), bank_conflict_cycles=int(c1))   # gathers + reduce-stores
        # the device ordering follows: the entry at device position p moves to position pi[p]
        ordx2 = np.empty_like(ordx); ordx2[pi[:n]] = ordx
        ordz2 = np.empty_like(ordz); ordz2[pi[n:N] - n] = ordz
        ordx, ordz = ordx2, ordz2
        posx[ordx] = np.arange(n); posz[ordz] = np.arange(m)
    kkt = _sp.pack(phases, N=N, slot_perm=pi)
    kkt_ragged = _sp.pack_ragged(phases, N, balanced=True, slot_perm=pi)
    if pi is not None and entry_order:
        kkt_ragged = _sl.apply_entries(kkt_ragged, eperm, pad_slot, pi_all)
        gs_ = _sl.gather_groups(_sp.gathered_slots(kkt_ragged, idle_zero=pad0))
        st_ = _sl.store_groups((kkt_ragged.desc & 0xFFFF).astype(np.int64), 0xFFFF)
        ident_ = np.arange(len(region))
        bank_stats['bank_conflict_cycles_gathers'] = int(_sl.conflict_cycles(gs_, ident_))
        bank_stats['bank_conflict_cycles_stores'] = int(_sl.conflict_cycles(st_, ident_, [_sl.STORE_BANK_PAIRS] * len(st_)))
        # (the annealer's model = this recount when the offsets are stored for all 64 lanes of a step; in the ragged layout the idle
        # lanes of a partial step read the entries that follow, which the recount sees and the annealer does not)
        assert not pad0 or bank_stats['bank_conflict_cycles_gathers'] + bank_stats['bank_conflict_cycles_stores'] == bank_stats['bank_conflict_cycles']
        bank_stats['bank_conflict_cycles'] = bank_stats['bank_conflict_cycles_gathers'] + bank_stats['bank_conflict_cycles_stores']
    kkt_squad = None
    if pi is not None and entry_order and rq0 is not None:
        kkt_squad = _sl.apply_entries(_cg.pack_squad(phases, N, pi), eperm_q, pad_q, pi_all)
        gq_ = _sl.gather_groups(_sp.gathered_slots(kkt_squad, idle_zero=True), _sl.B128_LANE_GROUP)
        sq_ = _sl.store_groups((kkt_squad.desc & 0xFFFF).astype(np.int64), 0xFFFF, group=8)
        ident_ = np.arange(len(region))
        bank_stats['squad_conflict_cycles_gathers'] = int(_sl.conflict_cycles(gq_, ident_, [16] * len(gq_)))
        bank_stats['squad_conflict_cycles_stores'] = int(_sl.conflict_cycles(sq_, ident_, [8] * len(sq_)))
        assert bank_stats['squad_conflict_cycles_gathers'] + bank_stats['squad_conflict_cycles_stores'] == bank_stats['squad_conflict_cycles']
    if pi is not None:
        # final_pos is indexed by the logical entry the phases were compiled with (= old device position)
        fp = np.empty(N, dtype=np.int64); fp[pi[:N]] = kkt.final_pos
        fr = np.empty(N, dtype=np.int64); fr[pi[:N]] = kkt_ragged.final_pos
        kkt.final_pos, kkt_ragged.final_pos = fp, fr
    assert kkt_ragged.n_slots == kkt.n_slots and np.array_equal(kkt_ragged.final_pos, kkt.final_pos)
    if kkt.n_slots >= 0xFFFF:
        raise NotImplementedError('problem family too large for 16-bit LDS slot indices')
    Pu, As = splan.pruned(desc.P, desc.A)          # (the termination test's products skip the exact zeros too)
    Pf = Pu + sp.triu(Pu, 1).T
    A_dev = sp.csr_matrix(As)[ordz][:, ordx]
    P_dev = sp.csr_matrix(Pf)[ordx][:, ordx]
    A_rows = _sp.pack([_sp.spmv_phase(A_dev, 0, 'A')], natural=True)
    P_rows = _sp.pack([_sp.spmv_phase(P_dev, 0, 'P')], natural=True)
    At_rows = _sp.pack([_sp.spmv_phase(sp.csr_matrix(A_dev.T), n, 'At')], natural=True)
    prim_idx = np.concatenate([posx[v.indices] for v in desc.variables]).astype(np.int32) \
        if desc.variables else np.zeros(0, dtype=np.int32)
    dual_idx = np.concatenate([posz[d.indices] for d in desc.duals]).astype(np.int32) \
        if desc.duals else np.zeros(0, dtype=np.int32)
    stats = dict(nnzL=len(splan.Li), nnzL_structural=len(plan.Li), nnzA=int(As.nnz), nnzA_structural=int(desc.A.nnz),
                 phases=kkt.n_phases, chunks=kkt.n_chunks, steps=kkt.steps,
                 nnz_program=kkt.nnz, n_slots=kkt.n_slots, lds_program_bytes=kkt_ragged.lds_bytes(),
                 compile_s=time.time() - t0, **bank_stats)
    return FamilyPlan(desc=desc, osqp=plan, ordx=ordx, ordz=ordz, posx=posx, posz=posz,
                      n_vary_x=int(vary_q.sum()), n_vary_z=int(vary_u.sum()), kkt=kkt,
                      kkt_ragged=kkt_ragged, A_rows=A_rows,
                      P_rows=P_rows, At_rows=At_rows, prim_idx=prim_idx, dual_idx=dual_idx,
                      stats=stats, osqp_shared=splan, phases=phases, slot_perm=pi, kkt_squad=kkt_squad)


@dataclass
class BatchResult:
    prim: Dict[str, np.ndarray]
    dual: Dict[str, np.ndarray]
    obj_val: np.ndarray
    iter: np.ndarray
    status: np.ndarray
    pri_res: np.ndarray
    dua_res: np.ndarray
    solve_time: float = 0.0
    kernel_ms: float = 0.0
    prim_flat: Optional[np.ndarray] = None
    dual_flat: Optional[np.ndarray] = None
    sol_x: Optional[np.ndarray] = None      # canonical solution (full_output solvers only)
    sol_y: Optional[np.ndarray] = None
    state: Optional[np.ndarray] = None      # workspace after the solve (solve(..., return_state=True))
    ctype: Optional[np.ndarray] = None      # ... and the row classes it holds ([B, m] int8; solve(..., ctype_in=...))

    def status_str(self) -> List[str]:
        return [STATUS_STRINGS.get(int(s), 'unknown') for s in self.status]


class BatchSolver:
    """One problem family on one GPU.  `solve(params)` = the reference's
    `cpg_solve(prob, updated_params, **kwargs)` for B instances at once."""

    def __init__(self, desc: FamilyDescriptor, device: int = 0, lib_path: Optional[str] = None,
                 plan: Optional[FamilyPlan] = None, ordering: str = 'mindeg', full_output: bool = False,
                 build_options: Optional[Dict[str, float]] = None):
        """full_output: return the complete canonical solution (sol_x, sol_y) -- what the reference's
        `cpg_solve_and_gradient_info` hands to `cpg_gradient` (templates/cpg_solver.py.jinja2:122-173).
        build_options: OSQP settings without a setter in the generated shim (BUILD_OPTIONS); default = the
        OSQP >= 1.0 library defaults (BUILD_OPTION_DEFAULTS), BUILD_OPTIONS_FIXED_RHO for a solver that never
        adapts rho."""
        self.full_output = full_output
        self.build_options = dict(build_options or {})
        for k in self.build_options:
            if k not in BUILD_OPTIONS:
                raise AttributeError(f'Build option "{k}" not available.')
        self._settings_kwargs: Dict[str, float] = {}
        self._ref_key = None
        self._grad_loaded = False
        if desc.solver != 'OSQP':
            raise ValueError(f'BatchSolver handles OSQP families, not {desc.solver}')
        self.desc = desc
        self.plan = plan or build_family_plan(desc, ordering=ordering)
        if lib_path is None:
            from . import codegen
            if max(-(-desc.n_var // 64), -(-desc.m // 64)) > codegen.GENERIC_MAX_SLOTS:
                # the generic library carries slot classes up to 16 x 16 (n_var, m <= 1024): larger
                # families get the same table-driven kernels compiled for their own class
                tag = ''.join(ch if ch.isalnum() else '_' for ch in (desc.name or 'family'))
                out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'generated', f'{tag}_{desc.n_var}x{desc.m}')
                lib_path = codegen.build_streamed_family_library(self.plan, out, tag, verbose=True)
        self.lib = CpgLibrary(lib_path)
        self._keep: list = []
        self._update_key = None
        self._update_keep: list = []
        self.h = C.c_void_p()
        p, o = self.plan, self.plan.osqp
        keep = self._keep
        D = np.ascontiguousarray(o.scaling.D[p.ordx]); E = np.ascontiguousarray(o.scaling.E[p.ordz])
        ctype = np.ascontiguousarray(o.constr_type[p.ordz], dtype=np.int8)
        self._family_ctype = np.asarray(o.constr_type, dtype=np.int8)      # canonical order (BatchSolver.row_classes)
        fpos = np.ascontiguousarray(p.kkt.final_pos, dtype=np.uint16)
        prim_idx = np.ascontiguousarray(p.posx if full_output else p.prim_idx, dtype=np.int32)
        dual_idx = np.ascontiguousarray(p.posz if full_output else p.dual_idx, dtype=np.int32)
        ord_ = np.ascontiguousarray(np.concatenate([p.ordx, p.ordz]), dtype=np.int32)
        keep += [D, E, ctype, fpos, prim_idx, dual_idx, ord_]
        rg = p.kkt_ragged
        rg_ctab = np.ascontiguousarray(rg.ctab, dtype=np.int32)
        rg_desc = np.ascontiguousarray(rg.desc, dtype=np.uint32)
        rg_vals = np.ascontiguousarray(rg.vals, dtype=np.float64)
        rg_cols = np.ascontiguousarray(rg.cols, dtype=np.uint16)
        keep += [rg_ctab, rg_desc, rg_vals, rg_cols]
        ragged = _Ragged(rg.n_chunks, rg.nnz, rg_ctab.ctypes.data_as(_ip),
                         rg_desc.ctypes.data_as(C.POINTER(C.c_uint32)), _d(rg_vals),
                         rg_cols.ctypes.data_as(_u16p))
        fam = _Family(
            n=desc.n_var, m=desc.m, n_eq=desc.n_eq, is_maximization=int(desc.is_maximization),
            sigma=o.settings['sigma'], alpha=o.settings['alpha'], rho=o.settings['rho'],
            D=_d(D), E=_d(E), c=o.scaling.c, ctype=ctype.ctypes.data_as(_i8p),
            n_slots=p.kkt.n_slots, fpos=fpos.ctypes.data_as(_u16p),
            n_vary_x=p.n_vary_x, n_vary_z=p.n_vary_z,
            kkt=_program_struct(p.kkt, keep), A_rows=_program_struct(p.A_rows, keep),
            P_rows=_program_struct(p.P_rows, keep), At_rows=_program_struct(p.At_rows, keep),
            kkt_ragged=ragged,
            n_prim=len(prim_idx), prim_idx=prim_idx.ctypes.data_as(_ip),
            n_dual=len(dual_idx), dual_idx=dual_idx.ctypes.data_as(_ip), ord=ord_.ctypes.data_as(_ip))
        self.lib.check(self.lib.L.cpg_hip_create_osqp(C.byref(fam), device, C.byref(self.h)),
                       'cpg_hip_create_osqp')
        self._apply_build_options(self.h)
        self.device = device
        self.h_shared = self.h
        self.h_ref = C.c_void_p()          # canonical-order handle of the refactorisation path (structural pattern)
        self._rplan = None
        self._rplan_res = None             # resident_plan.ResidentPlan when the library carries this family's resident executor
        self.h_rs = C.c_void_p()           # ... and of shared-matrix mode (numerically non-zero pattern, plan.osqp_shared)
        self._rplan_s = None
        self._rs_key = None
        self.h_rg = C.c_void_p()           # ... and of the adjoint when no varying parameter enters P / A: stored pattern for the
        self._rplan_g = None               # canonicalisation and the gradients, pruned pattern for the factor (see _ensure_refactor_handle)
        self._rg_key = None
        self._grad_loaded_on = set()
        self._hybrid = False
        self.np_var = 0
        self._var_cols = np.zeros(0, dtype=np.int64)

    def close(self):
        for name in ('h_shared', 'h_ref', 'h_rs', 'h_rg'):
            hh = getattr(self, name, None)
            if hh is not None and hh.value:
                self.lib.L.cpg_hip_destroy(hh)
                setattr(self, name, C.c_void_p())
        self.h = C.c_void_p()

    def _ensure_refactor_handle(self, shared_mats: bool = False, mode: Optional[str] = None):
        """further handles in canonical ordering (no device permutation) carrying the structural tables of a
        per-instance factor path.  mode 'struct' (h_ref): on the stored pattern of P and A (parameters entering the
        matrices; the adjoint with matrix parameters); 'shared' (h_rs): on the numerically non-zero pattern of the
        code-generation-time workspace (shared-matrix mode); 'grad' (h_rg): the adjoint when no varying parameter
        enters P / A -- canonicalisation, refinement and the gradients d(P), d(A) on the STORED pattern (an entry
        that is zero still has a gradient), the factor of the masked KKT matrix and its substitution program on the
        pruned one (the zeros contribute nothing to it): nnz(L) 6 314 -> 1 110 on MPC 12/4/10."""
        mode = mode or ('shared' if shared_mats else 'struct')
        shared_mats = mode == 'shared'
        attr = {'struct': 'h_ref', 'shared': 'h_rs', 'grad': 'h_rg'}[mode]
        if getattr(self, attr).value:
            return
        from . import refactor_plan as _rp
        import dataclasses
        desc = self.desc
        if shared_mats:
            o = self.plan.osqp_shared or self.plan.osqp
            Ps, As = o.pruned(desc.P, desc.A)
            # The plan codegen.instance_header generated the library's instance executor from (planned for register-
            # resident coefficients: more, narrower steps) -- when this library has one for this family: its
            # cpg_instance_<name>.h sits next to it.  Any other library streams the program, and gets the streaming plan.
            rplan = None
            fps = _instance_fingerprints(self.lib.path)
            if fps:
                cand = _rp.shared_mode_plan(Ps, As, o)
                if cand.sol.fingerprint() in fps:
                    rplan = cand
            if rplan is None:
                rplan = _rp.build_refactor_plan(Ps, As, o)
            self._rplan_s = rplan
        elif mode == 'grad':
            o = self.plan.osqp
            if self._rplan is None:
                self._rplan = _rp.build_refactor_plan(desc.P, desc.A, o)
            os_ = self.plan.osqp_shared or self.plan.osqp
            Pk, Ak = os_.pruned(desc.P, desc.A)
            rpp = _rp.build_refactor_plan(Pk, Ak, os_)
            kidx = np.array(rpp.ksrc_idx, dtype=np.int64)
            if os_.keepP is not None:          # entry numbers of the pruned matrices -> entry numbers of the stored ones
                isP, isA = rpp.ksrc_kind == _rp.K_P, rpp.ksrc_kind == _rp.K_A
                kidx[isP] = os_.keepP[kidx[isP]]
                kidx[isA] = os_.keepA[kidx[isA]]
            rplan = dataclasses.replace(self._rplan, nnzL=rpp.nnzL, Lp=rpp.Lp, Li=rpp.Li, Lcol=rpp.Lcol, perm=rpp.perm,
                                        ksrc_kind=rpp.ksrc_kind, ksrc_idx=kidx.astype(np.int32), fac=rpp.fac, fac_a=rpp.fac_a,
                                        fac_b=rpp.fac_b, fac_k=rpp.fac_k, sol=rpp.sol, sol_kind=rpp.sol_kind, sol_idx=rpp.sol_idx,
                                        stats=rpp.stats)
            self._rplan_g = rplan
        else:
            o = self.plan.osqp
            rplan = self._rplan
            if rplan is not None and not getattr(self, '_res_probed', False):
                rplan = None               # (a gradient built the plain plan first: the resident / team candidate is still to be probed)
            self._res_probed = True
            if rplan is None:
                # the plan codegen.resident_header generated this library's resident executor from (merged levels,
                # register-resident coefficients) -- when the library has one for this family; its `base` is the plan
                # every other library streams
                self._rplan_res = None
                fps_t = _instance_fingerprints(self.lib.path, 'cpg_team', 'GENT')
                if fps_t:
                    # ... or the team executor (csrc/cpg_osqp_team.h): the same plan with its programs planned for W wavefronts
                    # per instance -- W is what the header next to the library says
                    from . import codegen as _cg
                    mine = (desc.n_var, desc.m, int(desc.A.nnz))
                    for Wt, Gt, groups, dims in sorted(_team_widths(self.lib.path), key=lambda t_: (t_[0], t_[1] or 0)):
                        if any(dv is not None and dv != mv for dv, mv in zip(dims[:3], mine)):
                            continue           # (a header of another family in the same directory: not worth a plan build)
                        try:
                            cand = _cg.build_team_plan(desc, o, Wt, Gt, groups=list(groups) if groups else None)
                        except ValueError:
                            continue           # (level groups that do not tile this family's factor)
                        if cand.sol.fingerprint() in fps_t:
                            self._rplan_res = cand
                            rplan = cand.base
                            break
                    if rplan is None:
                        import warnings
                        warnings.warn(f'{self.lib.path} carries a team executor (cpg_team_*.h) but none of the headers next to it matches '
                                      f'this family\'s plan: per-instance solves will run the streaming kernel', RuntimeWarning)
                if rplan is None and _instance_fingerprints(self.lib.path, 'cpg_resident', 'GENR'):
                    from . import resident_plan as _rs
                    cand = _rs.build_resident_plan(desc.P, desc.A, o)
                    if cand.sol.fingerprint() in _instance_fingerprints(self.lib.path, 'cpg_resident', 'GENR'):
                        self._rplan_res = cand
                        rplan = cand.base
                if rplan is None:
                    rplan = _rp.build_refactor_plan(desc.P, desc.A, o)
            self._rplan = rplan
        keep = self._keep
        n, m = desc.n_var, desc.m
        ones_n, ones_m = np.ones(n), np.ones(m)
        ctype = np.ascontiguousarray(o.constr_type, dtype=np.int8)
        fpos = np.ascontiguousarray(rplan.sol.final_pos, dtype=np.uint16)
        prim_idx = np.ascontiguousarray(np.concatenate([v.indices for v in desc.variables]), dtype=np.int32) \
            if desc.variables else np.zeros(0, dtype=np.int32)
        dual_idx = np.ascontiguousarray(np.concatenate([d.indices for d in desc.duals]), dtype=np.int32) \
            if desc.duals else np.zeros(0, dtype=np.int32)
        if self.full_output:
            prim_idx, dual_idx = np.arange(n, dtype=np.int32), np.arange(m, dtype=np.int32)
        keep += [ones_n, ones_m, ctype, fpos, prim_idx, dual_idx]
        empty = _Program(0, 0, None, None, None, None)
        rows_A = rows_P = rows_At = empty
        if shared_mats:
            # products of the termination test through natural-layout row programs of the workspace's matrices
            # (canonical order): coalesced and shared by every wavefront, against per-lane row walks
            from . import codegen as _cg
            rows_A, rows_P, rows_At = (_program_struct(pr, keep) for pr in _cg.shared_row_programs(Ps, As))
        fam = _Family(
            n=n, m=m, n_eq=desc.n_eq, is_maximization=int(desc.is_maximization),
            sigma=o.settings['sigma'], alpha=o.settings['alpha'], rho=o.settings['rho'],
            D=_d(ones_n), E=_d(ones_m), c=1.0, ctype=ctype.ctypes.data_as(_i8p),
            n_slots=rplan.sol.n_slots, fpos=fpos.ctypes.data_as(_u16p), n_vary_x=n, n_vary_z=m,
            kkt=empty, A_rows=rows_A, P_rows=rows_P, At_rows=rows_At,
            kkt_ragged=_Ragged(0, 0, None, None, None, None),
            n_prim=len(prim_idx), prim_idx=prim_idx.ctypes.data_as(_ip),
            n_dual=len(dual_idx), dual_idx=dual_idx.ctypes.data_as(_ip), ord=None)
        hh = C.c_void_p()
        self.lib.check(self.lib.L.cpg_hip_create_osqp(C.byref(fam), self.device, C.byref(hh)),
                       'cpg_hip_create_osqp (per-instance factor handle)')
        setattr(self, attr, hh)
        self._apply_build_options(hh)
        if getattr(self, '_launch', None):
            self.lib.check(self.lib.L.cpg_hip_set_launch(hh, *self._launch), 'set_launch')
        if getattr(self, '_placement', None) is not None:
            self.lib.check(self.lib.L.cpg_hip_set_program_placement(hh, self._placement), 'set_program_placement')

    def _apply_build_options(self, hh) -> None:
        for k, v in self.build_options.items():
            self.lib.check(self.lib.L.cpg_hip_set_build_option(hh, k.encode(), float(v)), 'cpg_hip_set_build_option')

    @property
    def adaptive_rho(self) -> bool:
        bo = {**BUILD_OPTION_DEFAULTS, **self.build_options}
        return bool(bo['adaptive_rho']) and int(bo['adaptive_rho_interval']) > 0

    def _set_refactor(self, cols: np.ndarray, th_fixed: np.ndarray, q_setup: Optional[np.ndarray] = None,
                      shared_mats: bool = False, mode: Optional[str] = None):
        """tables of the per-instance factor path for this column subset; solve and gradient share them,
        one signature (cols, fixed part of theta, q of the workspace, mode) decides whether they are current.
        shared_mats: no varying parameter enters P or A -- the workspace's equilibrated matrices serve every
        instance (pre-scaled q / u maps, no re-equilibration in the kernel): the instances handed over by the
        shared-factor kernel after a rho change, and rows that changed class."""
        mode = mode or ('shared' if shared_mats else 'struct')
        shared_mats = mode == 'shared'
        self._ensure_refactor_handle(mode=mode)
        desc = self.desc
        rp = {'struct': self._rplan, 'shared': self._rplan_s, 'grad': self._rplan_g}[mode]
        o = (self.plan.osqp_shared or self.plan.osqp) if shared_mats else self.plan.osqp
        hh = {'struct': self.h_ref, 'shared': self.h_rs, 'grad': self.h_rg}[mode]
        if q_setup is None:
            q_setup = desc.default_canon()['q']
        q_setup = np.ascontiguousarray(q_setup, dtype=np.float64)
        key = (np.asarray(cols).tobytes(), np.asarray(th_fixed).tobytes(), q_setup.tobytes())
        if key == {'struct': self._ref_key, 'shared': self._rs_key, 'grad': self._rg_key}[mode]:
            return
        keep: list = []

        def split(pid, clip=False, scale=None):
            Cm = sp.csr_matrix(desc.maps[pid])
            base = np.asarray(Cm @ th_fixed).ravel()
            if clip:
                base = np.clip(base, -CPG_INF, CPG_INF)
            Mv = sp.csr_matrix(Cm[:, cols]) if len(cols) else sp.csr_matrix((Cm.shape[0], 0))
            if scale is not None:
                base, Mv = scale * base, sp.csr_matrix(sp.diags(scale) @ Mv)
            base = np.ascontiguousarray(base)
            keep.append(base)
            return base, _csr_struct(Mv, keep)

        if shared_mats:
            # the matrices are the workspace's: equilibrated values on the pattern of this plan, no maps
            Psm, Asm = o.pruned(desc.P, desc.A)
            Ps = np.ascontiguousarray(Psm.data, dtype=np.float64); As = np.ascontiguousarray(Asm.data, dtype=np.float64)
            Pb, Ab = np.zeros(max(1, rp.nnzP)), np.zeros(max(1, rp.nnzA))
            MP = _csr_struct(sp.csr_matrix((rp.nnzP, len(cols))), keep); MA = _csr_struct(sp.csr_matrix((rp.nnzA, len(cols))), keep)
            keep += [Pb, Ab]
        else:
            Pb, MP = split('P'); Ab, MA = split('A')
            Ps = np.ascontiguousarray(o.Px, dtype=np.float64); As = np.ascontiguousarray(o.Ax, dtype=np.float64)
        qb, Mq = split('q', scale=(o.scaling.c * o.scaling.D) if shared_mats else None)
        ub, Mu = split('u', clip=True, scale=o.scaling.E if shared_mats else None)
        Dv = np.ascontiguousarray(o.scaling.D, dtype=np.float64); Ev = np.ascontiguousarray(o.scaling.E, dtype=np.float64)
        keep += [Ps, As, Dv, Ev]
        keep.append(q_setup)
        Cd = sp.csr_matrix(desc.maps['d'])
        d_base = float((Cd @ th_fixed)[0]) if desc.nonzero_d else 0.0
        Md = _csr_struct(sp.csr_matrix(Cd[:, cols]) if (len(cols) and desc.nonzero_d)
                         else sp.csr_matrix((1, len(cols))), keep)

        def i32(a):
            a = np.ascontiguousarray(a, dtype=np.int32); keep.append(a); return a.ctypes.data_as(_ip)

        def u32(a):
            a = np.ascontiguousarray(a, dtype=np.uint32); keep.append(a); return a.ctypes.data_as(C.POINTER(C.c_uint32))

        def u16(a):
            a = np.ascontiguousarray(a, dtype=np.uint16); keep.append(a); return a.ctypes.data_as(_u16p)
        rf = _Refactor(
            nnzP=rp.nnzP, nnzA=rp.nnzA, nnzL=rp.nnzL, scaling_iters=int(o.settings['scaling']),
            Ap=i32(rp.Ap), Ai=i32(rp.Ai), Arp=i32(rp.Arp), Aent=i32(rp.Aent), Acol=i32(rp.Acol),
            Pp=i32(rp.Pp), Pi=i32(rp.Pi), Prp=i32(rp.Prp), Pent=i32(rp.Pent), Pcol=i32(rp.Pcol),
            Lcol=i32(rp.Lcol), ksrc_kind=i32(rp.ksrc_kind), ksrc_idx=i32(rp.ksrc_idx),
            fac_chunks=rp.fac.n_chunks, fac_triples=len(rp.fac_a), fac_ctab=i32(rp.fac.ctab),
            fac_task=u32(rp.fac.task), fac_len=u32(rp.fac.tlen), fac_a=u32(rp.fac_a), fac_b=u32(rp.fac_b),
            fac_k=u32(rp.fac_k), sol_chunks=rp.sol.n_chunks, sol_nnz=rp.sol.nnz, sol_slots=rp.sol.n_slots,
            sol_ctab=i32(rp.sol.ctab), sol_desc=u32(rp.sol.desc), sol_cols=u16(rp.sol.cols),
            sol_kind=i32(rp.sol_kind), sol_idx=i32(rp.sol_idx), sol_fpos=u16(rp.sol.final_pos),
            np_var=len(cols), P_base=_d(Pb), A_base=_d(Ab), q_base=_d(qb), u_base=_d(ub), d_base=d_base,
            map_P=MP, map_A=MA, map_q=Mq, map_u=Mu, map_d=Md, q_setup=_d(q_setup),
            shared_mats=int(bool(shared_mats)), Ps=_d(Ps), As=_d(As), D=_d(Dv), E=_d(Ev), c=float(o.scaling.c))
        res = getattr(self, '_rplan_res', None) if mode == 'struct' else None
        if res is not None:
            def rows(prog, ent):
                return _RowsProgram(n_chunks=prog.n_chunks, nnz=prog.nnz, ctab=i32(prog.ctab), desc=u32(prog.desc), cols=u16(prog.cols), ent=i32(ent))
            rs = _Resident(
                nnzX=res.nnzX, fac_chunks=int(res.f_ctab.shape[0]), fac_triples=len(res.f_a), f_ctab=i32(res.f_ctab),
                f_task=u32(res.f_task), f_len=u32(res.f_len), f_a=u32(res.f_a), f_b=u32(res.f_b), f_k=u32(res.f_k),
                sol_chunks=res.sol.n_chunks, sol_nnz=res.sol.nnz, sol_slots=res.sol.n_slots, sol_ctab=i32(res.sol.ctab),
                sol_desc=u32(res.sol.desc), sol_cols=u16(res.sol.cols), sol_kind=i32(res.sol_kind), sol_idx=i32(res.sol_idx),
                sol_lcol=i32(res.sol_lcol), rows_A=rows(res.rows_A, res.rows_A_ent), rows_P=rows(res.rows_P, res.rows_P_ent),
                rows_At=rows(res.rows_At, res.rows_At_ent), out_ax=res.out_ax, out_px=res.out_px, out_aty=res.out_aty)
            self.lib.check(self.lib.L.cpg_hip_set_resident(hh, C.byref(rf), C.byref(rs)), 'cpg_hip_set_resident')
        else:
            self.lib.check(self.lib.L.cpg_hip_set_refactor(hh, C.byref(rf)), 'cpg_hip_set_refactor')
        if shared_mats:
            self._rs_keep, self._rs_key = keep, key
        elif mode == 'grad':
            self._rg_keep, self._rg_key = keep, key
        else:
            self._refactor_keep, self._ref_key = keep, key

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- settings --------------------------------------------------------------------------------
    def apply_settings(self, **kwargs) -> None:
        """Reference semantics: reset to defaults, then apply the keyword arguments
        (`cvxpygen/templates/cpg_solver.py.jinja2:55-60`)."""
        self._settings_kwargs = dict(kwargs)
        self._apply_settings_to(self.h)

    def _apply_settings_to(self, hh) -> None:
        L = self.lib.L
        self.lib.check(L.cpg_hip_set_default_settings(hh), 'set_default_settings')
        for k, v in self._settings_kwargs.items():
            name = SETTING_ALIASES.get(k, k)
            if L.cpg_hip_set_setting(hh, name.encode(), float(v)) != 0:
                raise AttributeError(f'Solver setting "{k}" not available.')

    def set_launch(self, waves_per_block=0, inst_per_wave=0, blocks_per_cu=0):
        """launch geometry of both handles (shared-factor and refactorisation path)"""
        self._launch = (waves_per_block, inst_per_wave, blocks_per_cu)
        for hh in (self.h_shared, self.h_ref, self.h_rs, self.h_rg):
            if hh is not None and hh.value:
                self.lib.check(self.lib.L.cpg_hip_set_launch(hh, *self._launch), 'set_launch')

    def set_program_placement(self, in_lds: int = -1):
        """-1 automatic, 0 stream the solve program from L2/HBM, 1 keep it resident in LDS, 2 (per-instance factor
        handles) the streaming executor with its entry words in LDS instead of a generated executor"""
        self._placement = in_lds
        for hh in (self.h_shared, self.h_ref, self.h_rs):
            if hh is not None and hh.value:
                self.lib.check(self.lib.L.cpg_hip_set_program_placement(hh, in_lds), 'set_program_placement')

    # ---- which parameters vary ----------------------------------------------------------------------
    def set_updated(self, updated_params: Optional[Sequence[str]] = None, theta_base: Optional[np.ndarray] = None,
                    q_setup: Optional[np.ndarray] = None, path: str = 'auto') -> None:
        """Which user parameters vary across the batch.  Every other parameter is folded into the base
        vectors at `theta_base` (default: the code-generation-time values theta0 -- what a fresh process of
        the reference holds; a sequential caller passes the values its workspace holds now).
        path: 'auto' -- shared factor unless a varying parameter enters P / A; with rho adaptation on (the
        default) the shared-factor kernel then hands instances whose rho changes to the per-instance factor
        kernel in shared-matrix mode ("hybrid": two kernels per solve, cpg_hip_set_handover);
        'shared' / 'refactor' force one kernel (a caller that forces 'shared' guarantees that P and A at these
        values are the family's; 'refactor' = every instance equilibrates and factors from iteration 0); q_setup: the unscaled q the workspace held when its matrices were last
        updated (the cost scaling of OSQP's re-equilibration sees that one, HISTORY.md 4.3)."""
        desc, p, o = self.desc, self.plan, self.plan.osqp
        if updated_params is None:
            updated_params = desc.param_names
        names = []
        for nm in updated_params:
            desc.param(nm)                       # AttributeError for unknown names, as the reference
            if nm not in names:
                names.append(nm)
        names = [q.name for q in desc.params if q.name in names]     # theta order
        dep = desc.user_p_name_to_canon_outdated()
        touched = set()
        for nm in names:
            touched.update(dep[nm])
        refactor = path == 'refactor' or (path == 'auto' and bool(touched & {'P', 'A'}))
        hybrid = (not refactor) and self.adaptive_rho
        base = desc.theta0 if theta_base is None else np.asarray(theta_base, dtype=np.float64)
        key = (tuple(names), refactor, hybrid, None if theta_base is None else base.tobytes(),
               None if q_setup is None else np.asarray(q_setup, dtype=np.float64).tobytes())
        if key == self._update_key:
            return
        cols = np.concatenate([np.arange(desc.param(nm).col, desc.param(nm).col + desc.param(nm).size)
                               for nm in names]).astype(np.int64) if names else np.zeros(0, np.int64)
        NP = desc.NP
        fixed = np.ones(NP + 1, dtype=bool)
        fixed[cols] = False
        th_fixed = np.where(fixed, base, 0.0)
        self._th_fixed = th_fixed
        if refactor:
            # per-instance equilibration + factorisation path
            self._set_refactor(cols, th_fixed, q_setup)
            self.h = self.h_ref
            self._update_key, self._update_keep = key, []
            self._var_cols, self.np_var = cols, len(cols)
            self._updated_names = names
            self._q_setup = q_setup
            return
        self.h = self.h_shared
        keep: list = []

        def split(pid, scale, order, clip):
            Cm = sp.csr_matrix(desc.maps[pid])
            base_v = np.asarray(Cm @ th_fixed).ravel()
            if clip:
                base_v = np.clip(base_v, -CPG_INF, CPG_INF)
            Mv = sp.csr_matrix(Cm[:, cols]) if len(cols) else sp.csr_matrix((Cm.shape[0], 0))
            Mv = sp.diags(scale) @ Mv
            return np.ascontiguousarray((scale * base_v)[order]), sp.csr_matrix(Mv)[order]

        qb, Mq = split('q', o.scaling.c * o.scaling.D, p.ordx, False)
        ub, Mu = split('u', o.scaling.E, p.ordz, True)
        Cd = sp.csr_matrix(desc.maps['d'])
        d_base = float((Cd @ th_fixed)[0]) if desc.nonzero_d else 0.0
        Md = sp.csr_matrix(Cd[:, cols]) if (len(cols) and desc.nonzero_d) else sp.csr_matrix((1, len(cols)))
        keep += [qb, ub]
        upd = _Update(np_var=len(cols), q_base=_d(qb), u_base=_d(ub), d_base=d_base,
                      map_q=_csr_struct(Mq, keep), map_u=_csr_struct(Mu, keep),
                      map_d=_csr_struct(Md, keep))
        self.lib.check(self.lib.L.cpg_hip_set_update(self.h, C.byref(upd)), 'cpg_hip_set_update')
        if hybrid:
            # rho adaptation: instances whose rho changes continue on their own factor of the SAME matrices
            self._set_refactor(cols, th_fixed, q_setup, shared_mats=True)
            self.lib.check(self.lib.L.cpg_hip_set_handover(self.h_shared, self.h_rs), 'cpg_hip_set_handover')
        else:
            self.lib.check(self.lib.L.cpg_hip_set_handover(self.h_shared, None), 'cpg_hip_set_handover')
        self._hybrid = hybrid
        self._update_key, self._update_keep = key, keep
        self._var_cols, self.np_var = cols, len(cols)
        self._updated_names = names
        self._q_setup = q_setup

    # ---- adjoint ---------------------------------------------------------------------------------------
    def _set_gradient(self, hh=None):
        desc = self.desc
        hh = hh or self.h_ref
        NP = desc.NP
        blocks, kinds = [], []
        sizes = {'q': desc.n_var, 'l': desc.n_eq, 'u': desc.m, 'P': desc.P.nnz, 'A': desc.A.nnz}
        for kind, pid in enumerate(('q', 'l', 'u', 'P', 'A')):
            Cm = sp.csr_matrix(desc.maps[pid])[:, :NP]
            if not desc.changes.get(pid, False):
                Cm = sp.csr_matrix((sizes[pid], NP))
            blocks.append(Cm)
            kinds.append(np.full(Cm.shape[0], kind, dtype=np.int32))
        S = sp.vstack(blocks).tocsc()
        S.sort_indices()
        kind_of_row = np.concatenate(kinds)
        off = np.concatenate([[0], np.cumsum([b.shape[0] for b in blocks])])
        rows = S.indices
        tkind = kind_of_row[rows]
        tidx = (rows - off[tkind]).astype(np.int32)
        keep = []

        def i32(a):
            a = np.ascontiguousarray(a, dtype=np.int32); keep.append(a); return a.ctypes.data_as(_ip)
        tcoef = np.ascontiguousarray(S.data, dtype=np.float64); keep.append(tcoef)
        Pcol = np.repeat(np.arange(desc.n_var), np.diff(desc.P.indptr))
        Acol = np.repeat(np.arange(desc.n_var), np.diff(desc.A.indptr))
        g = _Gradient(NP=NP, Pcolidx=i32(Pcol), Acolidx=i32(Acol), tptr=i32(S.indptr), tkind=i32(tkind),
                      tidx=i32(tidx), tcoef=_d(tcoef))
        self.lib.check(self.lib.L.cpg_hip_set_gradient(hh, C.byref(g)), 'cpg_hip_set_gradient')
        self._gradient_keep = getattr(self, '_gradient_keep', []) + [keep]

    def gradient(self, params: Dict[str, np.ndarray], sol_x: np.ndarray, sol_y: np.ndarray,
                 dvars: Dict[str, np.ndarray], updated_params: Optional[Sequence[str]] = None,
                 theta_base: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
        """Batched `cpg_gradient` (templates/cpg_solver.py.jinja2:135-173): given the canonical solution of
        the forward solve and the gradient of a scalar loss w.r.t. the user variables, returns the
        gradient w.r.t. every user parameter (F-order reshaped like the reference's `param.gradient`)."""
        desc = self.desc
        if updated_params is None:
            updated_params = [q.name for q in desc.params if q.name in params]
        names = [q.name for q in desc.params if q.name in updated_params]
        cols = np.concatenate([np.arange(desc.param(nm).col, desc.param(nm).col + desc.param(nm).size)
                               for nm in names]).astype(np.int64) if names else np.zeros(0, np.int64)
        fixed = np.ones(desc.NP + 1, dtype=bool)
        fixed[cols] = False
        base = desc.theta0 if theta_base is None else np.asarray(theta_base, dtype=np.float64)
        dep = desc.user_p_name_to_canon_outdated()
        touched = set().union(*[dep[nm] for nm in names]) if names else set()
        # matrices at their code-generation-time values in every instance: the masked KKT matrix is factored on the
        # numerically non-zero pattern (handle h_rg); otherwise on the stored one (h_ref, shared with the solve path)
        pruned = theta_base is None and not (touched & {'P', 'A'}) and os.environ.get('CPG_PRUNE', '1') != '0'
        if pruned:
            self._set_refactor(cols, np.where(fixed, base, 0.0), None, mode='grad')
            hg = self.h_rg
        else:
            before = self._ref_key
            self._set_refactor(cols, np.where(fixed, base, 0.0), None)     # no-op when these tables are loaded
            if self._ref_key != before:
                self._update_key = None          # a following solve must re-select its tables
            hg = self.h_ref
        if hg.value not in self._grad_loaded_on:
            self._set_gradient(hg)
            self._grad_loaded_on.add(hg.value)
        self.h_grad = hg
        tv = self.theta_var(params, names=names)      # (the solve-side selection -- _updated_names, _var_cols -- stays as it is)
        B = sol_x.shape[0]
        dx = np.zeros((B, desc.n_var))
        for v in desc.variables:
            if v.name in dvars:
                g = np.asarray(dvars[v.name], dtype=np.float64).reshape((B,) + tuple(v.shape))
                dx[:, v.indices] = g.transpose((0,) + tuple(range(len(v.shape), 0, -1))).reshape(B, -1) \
                    if len(v.shape) > 1 else g.reshape(B, -1)
        dth = np.empty((B, desc.NP))
        sx = np.ascontiguousarray(sol_x, dtype=np.float64); sy = np.ascontiguousarray(sol_y, dtype=np.float64)
        tv = np.ascontiguousarray(tv, dtype=np.float64)
        self.lib.check(self.lib.L.cpg_hip_gradient_batch(hg, B, _d(tv), _d(sx), _d(sy), _d(dx), _d(dth)),
                       'cpg_hip_gradient_batch')
        out = {'_flat': dth}
        for q in desc.params:
            blk = dth[:, q.col:q.col + q.size]
            if q.kind == 'dense' and len(q.shape) > 1:
                blk = blk.reshape((B,) + tuple(q.shape)[::-1]).transpose((0,) + tuple(range(len(q.shape), 0, -1)))
            elif q.kind == 'scalar':
                blk = blk[:, 0]
            out[q.name] = blk
        return out

    def theta_var(self, params: Dict[str, np.ndarray], B: Optional[int] = None,
                  names: Optional[Sequence[str]] = None) -> np.ndarray:
        """[B, np_var] C-contiguous array of the updated parameters, each flattened as the reference's
        `get_param_value` does (F-order / diagonal / stored non-zeros), instance-major.
        names: parameter set in theta order (default: the one selected by set_updated)."""
        if names is None:
            names = self._updated_names
        blocks = []
        for nm in names:
            if nm not in params:
                raise KeyError(f'value for updated parameter {nm} missing')
            up = self.desc.param(nm)
            v = np.asarray(params[nm], dtype=np.float64)
            Bn = v.shape[0]
            if B is None:
                B = Bn
            if Bn != B:
                raise ValueError('inconsistent batch sizes')
            if v.ndim == 2 and v.shape[1] == up.size:
                # already flattened the way the reference stores it: F-ORDER for dense matrices
                # (templates/cpg_solver.py.jinja2:26-34), the diagonal, or the stored non-zeros
                blocks.append(v.reshape(B, up.size))
            elif up.kind == 'diag' and v.ndim == 3:
                blocks.append(np.diagonal(v, axis1=1, axis2=2).reshape(B, up.size))
            elif up.kind == 'sparse' and v.ndim == 3:
                r, c = up.sparsity
                blocks.append(v[:, np.asarray(r), np.asarray(c)].reshape(B, up.size))
            elif up.kind == 'scalar':
                blocks.append(v.reshape(B, 1))
            else:
                blocks.append(v.reshape((B,) + tuple(up.shape)).transpose(
                    (0,) + tuple(range(len(up.shape), 0, -1))).reshape(B, up.size))
        if not blocks:
            return np.zeros((B or 0, 0))
        return np.ascontiguousarray(np.concatenate(blocks, axis=1))

    # ---- solve ---------------------------------------------------------------------------------------
    def solve(self, params: Optional[Dict[str, np.ndarray]] = None,
              updated_params: Optional[Sequence[str]] = None, B: Optional[int] = None,
              theta_var: Optional[np.ndarray] = None, state_in: Optional[np.ndarray] = None,
              return_state: bool = False, ctype_in: Optional[np.ndarray] = None, **kwargs) -> BatchResult:
        """state_in / return_state: the workspace a sequential caller carries from solve to solve
        ([B, n + 2 m + 1]: scaled iterates x | z | y in canonical order, then rho; include/cpg_hip.h).
        ctype_in [B, m] (int8): the row classes that workspace held (`BatchResult.ctype` of its previous solve; default:
        the family's, i.e. a workspace that never saw a class change).  OSQP's update_rho_vec rebuilds rho_vec from
        settings->rho -- the family's rho, restored by every cpg_solve -- when a bound moved a row to another class
        (oracle/osqp_oracle.c set_rho_vec): instances whose classes differ from ctype_in therefore start from that rho,
        not from the adapted one in state_in."""
        if not (updated_params is None and theta_var is not None and self._update_key is not None):
            self.set_updated(updated_params)      # None = every parameter, as in the reference
        self.apply_settings(**kwargs)
        if theta_var is None:
            theta_var = self.theta_var(params or {}, B)
        theta_var = np.ascontiguousarray(theta_var, dtype=np.float64)
        Bn = theta_var.shape[0] if theta_var.ndim == 2 and self.np_var else (B or theta_var.shape[0])
        if self.np_var and theta_var.shape != (Bn, self.np_var):
            raise ValueError(f'theta_var must have shape (B, {self.np_var})')
        ctype = None
        if (state_in is not None or return_state) and self.desc.solver == 'OSQP':
            ctype = self.row_classes(theta_var, Bn)
            if state_in is not None:
                prev = self._family_ctype[None, :] if ctype_in is None else np.asarray(ctype_in).reshape(Bn, -1)
                moved = (ctype != prev).any(axis=1)
                if moved.any():
                    state_in = np.array(state_in, dtype=np.float64)
                    state_in[moved, -1] = min(max(self.plan.osqp.settings['rho'], 1e-6), 1e6)
        t0 = time.time()
        out = self._solve_on(self.h, theta_var, Bn, state_in, return_state)
        t1 = time.time()
        ms = C.c_float(0)
        self.lib.L.cpg_hip_last_kernel_ms(self.h, C.byref(ms))
        self._resolve_class_changes(theta_var, out, state_in)
        res = self._result(*out[:7], t1 - t0, ms.value)
        res.state = out[7]
        res.ctype = ctype
        return res

    def row_classes(self, theta_var: np.ndarray, Bn: int) -> np.ndarray:
        """[B, m] int8 classes of the constraint rows at these parameter values, as OSQP's update_rho_vec sees them
        (-1 free, 1 equality, 0 inequality) with the scaling E of the family.  (After a matrix update the workspace's
        E is the instance's own; a bound of +-1e30 and l == u classify the same under any admissible E in
        [1e-4, 1e4], only a row with 0 < E (u - l) < 1e-4 could differ.)"""
        d, o = self.desc, self.plan.osqp
        # the maps in CSR, their fixed part and their columns over the varying parameters: built once per set_updated
        key = (self._var_cols.tobytes(), self._th_fixed.tobytes())
        cache = getattr(self, '_rc_cache', None)
        if cache is None or cache[0] != key:
            Mu, Ml = sp.csr_matrix(d.maps['u']), sp.csr_matrix(d.maps['l'])
            tf = np.array(self._th_fixed[:d.NP + 1], dtype=np.float64)
            if self.np_var:
                tf[self._var_cols] = 0.0
            ub, lb = np.asarray(Mu @ tf).ravel(), (np.asarray(Ml @ tf).ravel() if d.n_eq else np.zeros(0))
            Muv = sp.csr_matrix(Mu[:, self._var_cols]) if self.np_var else None
            Mlv = sp.csr_matrix(Ml[:, self._var_cols]) if (self.np_var and d.n_eq) else None
            cache = (key, ub, lb, Muv, Mlv)
            self._rc_cache = cache
        _, ub, lb, Muv, Mlv = cache
        if (Muv is None or Muv.nnz == 0) and (Mlv is None or Mlv.nnz == 0):
            # neither bound depends on a varying parameter: no row can have changed class
            u1 = np.clip(ub, -CPG_INF, CPG_INF)[None, :]
            l1 = np.full_like(u1, -CPG_INF)
            if d.n_eq:
                l1[:, :d.n_eq] = np.clip(lb, -CPG_INF, CPG_INF)
            E = np.asarray(o.scaling.E, dtype=np.float64)
            ls, us = l1 * E, u1 * E
            free = (ls < -CPG_INF * 1e-4) & (us > CPG_INF * 1e-4)
            eq = ~free & (us - ls < 1e-4)
            return np.broadcast_to(np.where(free, -1, np.where(eq, 1, 0)).astype(np.int8), (Bn, d.m)).copy()
        E = np.asarray(o.scaling.E, dtype=np.float64)
        u = ub[None, :] + (np.asarray((Muv @ theta_var.T).T) if Muv is not None and Muv.nnz else 0.0)
        u = np.clip(np.broadcast_to(u, (Bn, d.m)), -CPG_INF, CPG_INF)
        l = np.full_like(u, -CPG_INF)
        if d.n_eq:
            lv = lb[None, :] + (np.asarray((Mlv @ theta_var.T).T) if Mlv is not None and Mlv.nnz else 0.0)
            l[:, :d.n_eq] = np.clip(np.broadcast_to(lv, (Bn, d.n_eq)), -CPG_INF, CPG_INF)
        ls, us = l * E, u * E
        free = (ls < -CPG_INF * 1e-4) & (us > CPG_INF * 1e-4)
        eq = ~free & (us - ls < 1e-4)
        return np.where(free, -1, np.where(eq, 1, 0)).astype(np.int8)

    def _solve_on(self, hh, theta_var, Bn, state_in, return_state):
        n_prim, n_dual = self.n_out_prim, self.n_out_dual
        prim = np.empty((Bn, n_prim)); dual = np.empty((Bn, n_dual))
        obj = np.empty(Bn); pri = np.empty(Bn); dua = np.empty(Bn)
        it = np.empty(Bn, dtype=np.int32); st = np.empty(Bn, dtype=np.int32)
        slen = self.desc.n_var + 2 * self.desc.m + 1
        if state_in is not None:
            state_in = np.ascontiguousarray(state_in, dtype=np.float64)
            if state_in.shape != (Bn, slen):
                raise ValueError(f'state_in must have shape (B, {slen})')
        state_out = np.empty((Bn, slen)) if return_state else None
        self.lib.check(self.lib.L.cpg_hip_solve_batch_state(
            hh, Bn, _d(theta_var), None if state_in is None else _d(state_in),
            None if state_out is None else _d(state_out), _d(prim), _d(dual), _d(obj),
            it.ctypes.data_as(_ip), st.ctypes.data_as(_ip), _d(pri), _d(dua)), 'cpg_hip_solve_batch_state')
        return [prim, dual, obj, it, st, pri, dua, state_out]

    def _resolve_class_changes(self, theta_var, out, state_in=None) -> None:
        """Instances whose parameters moved a constraint row to another class (a bound that became
        +-infinite or finite again) cannot use the family's shared factor; the reference refactors
        inside osqp_update_data_vec (update_rho_vec).  They come back from the shared-factor kernel
        flagged and are solved here through the per-instance factor path, whose kernel classifies the
        rows of every instance itself."""
        st = out[4]
        bad = np.nonzero(st == STATUS_NEEDS_REFACTOR)[0]
        if not len(bad) or self.h is not self.h_shared:
            return
        # the reference keeps its scaling when a bound moves a row to another class (osqp_update_data_vec ->
        # update_rho_vec refactors K, nothing else): the workspace's matrices, a factor per instance
        self._set_refactor(self._var_cols, self._th_fixed, self._q_setup, shared_mats=True)
        self._apply_settings_to(self.h_rs)
        # (state_in: `solve` has already put the family's rho where the classes differ from the ones the workspace held)
        sub = self._solve_on(self.h_rs, np.ascontiguousarray(theta_var[bad]), len(bad),
                             None if state_in is None else state_in[bad], out[7] is not None)
        for k in range(8):
            if out[k] is not None:
                out[k][bad] = sub[k]

    @property
    def n_out_prim(self) -> int:
        return self.desc.n_var if self.full_output else len(self.plan.prim_idx)

    @property
    def n_out_dual(self) -> int:
        return self.desc.m if self.full_output else len(self.plan.dual_idx)

    def _result(self, prim, dual, obj, it, st, pri, dua, dt, ms) -> BatchResult:
        d = self.desc
        Bn = prim.shape[0]
        sol_x = sol_y = None
        if self.full_output:
            sol_x, sol_y = prim, dual
            prim = np.concatenate([sol_x[:, v.indices] for v in d.variables], axis=1) if d.variables else np.zeros((Bn, 0))
            dual = np.concatenate([sol_y[:, u.indices] for u in d.duals], axis=1) if d.duals else np.zeros((Bn, 0))
        pd, dd, k = {}, {}, 0
        for v in d.variables:
            sz = v.indices.size
            pd[v.name] = prim[:, k:k + sz].reshape((Bn,) + tuple(v.shape)[::-1]).transpose(
                (0,) + tuple(range(len(v.shape), 0, -1))) if len(v.shape) > 1 else prim[:, k:k + sz]
            k += sz
        k = 0
        for u in d.duals:
            sz = u.indices.size
            if len(u.shape) > 1:
                dd[u.name] = dual[:, k:k + sz].reshape((Bn,) + tuple(u.shape)[::-1]).transpose(
                    (0,) + tuple(range(len(u.shape), 0, -1)))
            elif len(u.shape) == 1:
                dd[u.name] = dual[:, k:k + sz]
            else:
                dd[u.name] = dual[:, k]
            k += sz
        # +-1e30 -> +-inf as the reference shim does (templates/cpg_solver.py.jinja2:98-101)
        obj = np.where(obj >= 1e30, np.inf, np.where(obj <= -1e30, -np.inf, obj))      # (NaN stays NaN)
        res = BatchResult(prim=pd, dual=dd, obj_val=obj, iter=it, status=st, pri_res=pri,
                          dua_res=dua, solve_time=dt, kernel_ms=ms, prim_flat=prim, dual_flat=dual)
        res.sol_x, res.sol_y = sol_x, sol_y
        return res


class DeviceBatch:
    """Device-resident buffers for `BatchSolver.solve_device`: theta_var in, results out, all in
    HBM (cpg_hip_malloc), so that a timed solve contains no PCIe traffic.  Bound to the parameter
    set that was selected (`set_updated`) when it was created."""

    def __init__(self, solver: BatchSolver, B: int):
        self.s, self.B = solver, int(B)
        self.n_prim, self.n_dual = solver.n_out_prim, solver.n_out_dual
        self.np_var, self._key = solver.np_var, solver._update_key
        self._theta_host = None
        self._ptrs = {}
        sizes = dict(theta=B * max(solver.np_var, 1) * 8, prim=B * self.n_prim * 8,
                     dual=B * self.n_dual * 8, obj=B * 8, pri=B * 8, dua=B * 8, iter=B * 4, status=B * 4)
        for k, nbytes in sizes.items():
            p = C.c_void_p()
            solver.lib.check(solver.lib.L.cpg_hip_malloc(solver.h, nbytes, C.byref(p)), 'cpg_hip_malloc')
            self._ptrs[k] = p

    def check_current(self) -> None:
        if self.s._update_key != self._key or self.s.np_var != self.np_var:
            raise ValueError('the solver\'s set of updated parameters changed after this DeviceBatch was created')

    def upload(self, theta_var: np.ndarray) -> None:
        self.check_current()
        tv = np.ascontiguousarray(theta_var, dtype=np.float64)
        if tv.shape != (self.B, self.np_var) and not (self.np_var == 0 and tv.size == 0):
            raise ValueError(f'theta_var must have shape ({self.B}, {self.np_var}), got {tv.shape}')
        self._theta_host = tv
        if tv.size:
            self.s.lib.check(self.s.lib.L.cpg_hip_memcpy_h2d(self.s.h, self._ptrs['theta'],
                                                             tv.ctypes.data_as(C.c_void_p), tv.nbytes), 'h2d')

    def download(self) -> BatchResult:
        s, B = self.s, self.B
        out = dict(prim=np.empty((B, self.n_prim)), dual=np.empty((B, self.n_dual)), obj=np.empty(B),
                   pri=np.empty(B), dua=np.empty(B), iter=np.empty(B, dtype=np.int32),
                   status=np.empty(B, dtype=np.int32))
        for k, a in out.items():
            if a.nbytes:
                s.lib.check(s.lib.L.cpg_hip_memcpy_d2h(s.h, a.ctypes.data_as(C.c_void_p), self._ptrs[k],
                                                       a.nbytes), 'd2h')
        lst = [out['prim'], out['dual'], out['obj'], out['iter'], out['status'], out['pri'], out['dua'], None]
        if (out['status'] == STATUS_NEEDS_REFACTOR).any():
            if self._theta_host is None:
                raise RuntimeError('row-class changes need the host copy of theta_var (DeviceBatch.upload)')
            s._resolve_class_changes(self._theta_host, lst)
        return s._result(*lst[:7], 0.0, s.last_kernel_ms())

    def free(self) -> None:
        for p in self._ptrs.values():
            self.s.lib.L.cpg_hip_free(self.s.h, p)
        self._ptrs = {}


def _solve_device(self, dev: DeviceBatch) -> None:
    """Asynchronous launch on the solver's stream; pair with synchronize()."""
    dev.check_current()
    P = dev._ptrs
    self.lib.check(self.lib.L.cpg_hip_solve_batch_device(
        self.h, dev.B, P['theta'], P['prim'], P['dual'], P['obj'], P['iter'], P['status'], P['pri'],
        P['dua']), 'cpg_hip_solve_batch_device')


def _synchronize(self) -> None:
    self.lib.check(self.lib.L.cpg_hip_synchronize(self.h), 'cpg_hip_synchronize')


def _last_kernel_ms(self) -> float:
    ms = C.c_float(0)
    self.lib.check(self.lib.L.cpg_hip_last_kernel_ms(self.h, C.byref(ms)), 'cpg_hip_last_kernel_ms')
    return float(ms.value)


def _last_phase_ms(self):
    """(ms of the shared-factor kernel, ms of the per-instance factor kernel behind it, instances handed over)
    of the most recent solve on the current handle"""
    a, b, n = C.c_float(0), C.c_float(0), C.c_int64(0)
    self.lib.check(self.lib.L.cpg_hip_last_phase_ms(self.h, C.byref(a), C.byref(b), C.byref(n)), 'cpg_hip_last_phase_ms')
    return float(a.value), float(b.value), int(n.value)


BatchSolver.last_phase_ms = _last_phase_ms
BatchSolver.solve_device = _solve_device
BatchSolver.synchronize = _synchronize
BatchSolver.last_kernel_ms = _last_kernel_ms


class PinnedStream:
    """`n_batches` consecutive batches of B instances in page-locked HOST memory, solved through
    cpg_hip_solve_batches_pipelined: H2D of batch i + 1, the solve of batch i and D2H of batch i - 1
    overlap, so that in steady state the PCIe transfers hide behind the kernel -- the whole-path rate of
    SURVEY.md 8(d) (theta in host memory -> results in host memory)."""

    def __init__(self, solver: BatchSolver, B: int, n_batches: int):
        self.s, self.B, self.n = solver, int(B), int(n_batches)
        self._key = solver._update_key
        tot = self.B * self.n
        self._raw = []

        def pinned(shape, dtype):
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            p = C.c_void_p()
            solver.lib.check(solver.lib.L.cpg_hip_host_malloc(solver.h, max(nbytes, 8), C.byref(p)), 'cpg_hip_host_malloc')
            self._raw.append(p)
            buf = (C.c_char * max(nbytes, 8)).from_address(p.value)
            return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self.theta = pinned((tot, max(solver.np_var, 1)), np.float64)[:, :solver.np_var]
        self.prim = pinned((tot, solver.n_out_prim), np.float64)
        self.dual = pinned((tot, solver.n_out_dual), np.float64)
        self.obj, self.pri, self.dua = pinned((tot,), np.float64), pinned((tot,), np.float64), pinned((tot,), np.float64)
        self.iter, self.status = pinned((tot,), np.int32), pinned((tot,), np.int32)

    def run(self) -> None:
        s = self.s
        if s._update_key != self._key:
            raise ValueError('the solver\'s set of updated parameters changed after this PinnedStream was created')
        th = self.theta if self.theta.flags['C_CONTIGUOUS'] else np.ascontiguousarray(self.theta)
        s.lib.check(s.lib.L.cpg_hip_solve_batches_pipelined(
            s.h, self.B, self.n, _d(th), _d(self.prim), _d(self.dual), _d(self.obj), self.iter.ctypes.data_as(_ip),
            self.status.ctypes.data_as(_ip), _d(self.pri), _d(self.dua)), 'cpg_hip_solve_batches_pipelined')

    def result(self) -> BatchResult:
        lst = [np.array(self.prim), np.array(self.dual), np.array(self.obj), np.array(self.iter), np.array(self.status),
               np.array(self.pri), np.array(self.dua), None]
        if (lst[4] == STATUS_NEEDS_REFACTOR).any():
            self.s._resolve_class_changes(np.array(self.theta), lst)
        return self.s._result(*lst[:7], 0.0, self.s.last_kernel_ms())

    def free(self) -> None:
        for p in self._raw:
            self.s.lib.L.cpg_hip_host_free(self.s.h, p)
        self._raw = []
