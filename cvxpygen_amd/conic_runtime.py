"""
Host side of the conic interior-point path: the batched equivalent of
the reference's generated Clarabel solver (`cvxpygen/solvers/clarabel.py:172-204`; shim
`cvxpygen/templates/cpg_solver.py.jinja2:36-118`).  One `ConicBatchSolver` per problem family and
GPU; `solve(params)` = `cpg_solve(prob, updated_params, **kwargs)` for B instances at once.

The reference builds a new Clarabel solver in every solve, so there is no code-generation-time
factor to share: the kernel canonicalises, equilibrates and factors per instance; this module only
uploads the family's fixed patterns / schedules (conic_plan.py) and the canonicalisation maps.
No CPU fallback: without the HIP library construction fails.
"""

from __future__ import annotations

import ctypes as C
import time
from typing import Dict, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from . import conic_plan as _cp
from .descriptor import FamilyDescriptor
from .runtime import BatchSolver, CpgLibrary, _Csr, _csr_struct, _d, _dp, _ip, _u16p

_u32p = C.POINTER(C.c_uint32)

CLARABEL_STATUS = {0: 'Unsolved', 1: 'Solved', 2: 'PrimalInfeasible', 3: 'DualInfeasible',
                   4: 'AlmostSolved', 5: 'AlmostPrimalInfeasible', 6: 'AlmostDualInfeasible',
                   7: 'MaxIterations', 8: 'MaxTime', 9: 'NumericalError', 10: 'InsufficientProgress'}


class _ConicFamily(C.Structure):     # include/cpg_hip.h: cpg_conic_family_t
    _fields_ = [('n', C.c_int32), ('m', C.c_int32), ('is_maximization', C.c_int32),
                ('n_zero', C.c_int32), ('n_nonneg', C.c_int32), ('n_soc', C.c_int32), ('soc_dims', _ip),
                ('nnzP', C.c_int32), ('nnzA', C.c_int32), ('nnzL', C.c_int32),
                ('Ap', _ip), ('Ai', _ip), ('Arp', _ip), ('Aent', _ip), ('Acol', _ip),
                ('Pp', _ip), ('Pi', _ip), ('Prp', _ip), ('Pent', _ip), ('Pcol', _ip),
                ('Lcol', _ip), ('ksrc_kind', _ip), ('ksrc_idx', _ip),
                ('fac_chunks', C.c_int32), ('fac_triples', C.c_int32), ('fac_ctab', _ip),
                ('fac_task', _u32p), ('fac_len', _u32p), ('fac_a', _u32p), ('fac_b', _u32p), ('fac_k', _u32p),
                ('sol_chunks', C.c_int32), ('sol_nnz', C.c_int32), ('sol_slots', C.c_int32),
                ('sol_ctab', _ip), ('sol_desc', _u32p), ('sol_cols', _u16p),
                ('sol_kind', _ip), ('sol_idx', _ip), ('sol_fpos', _u16p),
                ('np_var', C.c_int32), ('P_base', _dp), ('A_base', _dp), ('q_base', _dp), ('b_base', _dp),
                ('d_base', C.c_double),
                ('map_P', _Csr), ('map_A', _Csr), ('map_q', _Csr), ('map_b', _Csr), ('map_d', _Csr),
                ('n_prim', C.c_int32), ('prim_idx', _ip), ('n_dual', C.c_int32), ('dual_idx', _ip),
                ('n_exp', C.c_int32), ('n_pow', C.c_int32), ('pow_alpha', _dp), ('n_psd', C.c_int32), ('psd_dims', _ip)]


class _PlanView:
    """the attributes of runtime.FamilyPlan the shared result / bench code looks at"""

    def __init__(self, cp: _cp.ConicPlan, prim_idx, dual_idx):
        self.conic = cp
        self.prim_idx, self.dual_idx = prim_idx, dual_idx
        self.stats = dict(cp.stats)


class ConicBatchSolver(BatchSolver):
    SETTING_ALIASES = {'max_iters': 'max_iter'}        # name_cvxpy, clarabel.py:65

    def __init__(self, desc: FamilyDescriptor, device: int = 0, lib_path: Optional[str] = None,
                 plan: Optional[_cp.ConicPlan] = None, ordering: str = 'auto', full_output: bool = False):
        if desc.solver != 'CLARABEL' or not desc.cones:
            raise ValueError(f'ConicBatchSolver handles conic (CLARABEL) families, not {desc.solver}')
        self.full_output = full_output
        self.desc = desc
        self.lib = CpgLibrary(lib_path)
        if not hasattr(self.lib.L, 'cpg_hip_create_clarabel'):
            raise RuntimeError(f'{self.lib.path} has no conic interior-point kernel (family-specialised '
                               'OSQP build?)')
        self.lib.L.cpg_hip_create_clarabel.argtypes = [C.POINTER(_ConicFamily), C.c_int, C.POINTER(C.c_void_p)]
        t0 = time.time()
        cp = plan or _cp.build_conic_plan(desc, ordering=ordering)
        prim_idx = np.concatenate([v.indices for v in desc.variables]).astype(np.int32) \
            if desc.variables else np.zeros(0, dtype=np.int32)
        dual_idx = np.concatenate([d.indices for d in desc.duals]).astype(np.int32) \
            if desc.duals else np.zeros(0, dtype=np.int32)
        self.plan = _PlanView(cp, prim_idx, dual_idx)
        self.plan.stats['compile_s'] = time.time() - t0
        self.device = device
        self.h = C.c_void_p()
        self.h_shared = C.c_void_p()
        self.h_ref = C.c_void_p()
        self._update_key = None
        self._keep: list = []
        self.np_var = 0
        self._var_cols = np.zeros(0, dtype=np.int64)
        self._updated_names = []

    def close(self):
        if getattr(self, 'h', None) is not None and self.h.value:
            self.lib.L.cpg_hip_destroy(self.h)
        self.h = C.c_void_p()
        self._update_key = None

    def apply_settings(self, **kwargs) -> None:
        L = self.lib.L
        self.lib.check(L.cpg_hip_set_default_settings(self.h), 'set_default_settings')
        for k, v in kwargs.items():
            name = self.SETTING_ALIASES.get(k, k)
            rc = L.cpg_hip_set_setting(self.h, name.encode(), float(v))
            if rc == -4:           # CPG_E_UNSUPPORTED: a setting of the reference's interface this backend accepts at its default only
                raise NotImplementedError(L.cpg_hip_last_error().decode())
            if rc != 0:
                raise AttributeError(f'Solver setting "{k}" not available.')

    def set_program_placement(self, in_lds: int = -1):
        """-1 / 1: block-shared LDS copy of the family's index tables when it fits; 0: tables stay in L2"""
        self._placement = in_lds
        if self.h.value:
            self.lib.check(self.lib.L.cpg_hip_set_program_placement(self.h, in_lds), 'set_program_placement')

    def gradient(self, *a, **k):
        raise NotImplementedError('a conic family is differentiated through its OSQP form '
                                  '(cvxpygen/generator.py:76-80): cvxpygen_amd.two_stage.TwoStageBatchSolver')

    def set_updated(self, updated_params: Optional[Sequence[str]] = None) -> None:
        """(Re)creates the device handle for this set of per-instance parameters: every other
        parameter is folded into the base vectors at its code-generation-time value."""
        desc, cp = self.desc, self.plan.conic
        if updated_params is None:
            updated_params = desc.param_names
        names = []
        for nm in updated_params:
            desc.param(nm)
            if nm not in names:
                names.append(nm)
        names = [q.name for q in desc.params if q.name in names]
        key = tuple(names)
        if key == self._update_key and self.h.value:
            return
        self.close()
        cols = np.concatenate([np.arange(desc.param(nm).col, desc.param(nm).col + desc.param(nm).size)
                               for nm in names]).astype(np.int64) if names else np.zeros(0, np.int64)
        fixed = np.ones(desc.NP + 1, dtype=bool)
        fixed[cols] = False
        th_fixed = np.where(fixed, desc.theta0, 0.0)
        keep: list = []

        def split(pid):
            Cm = sp.csr_matrix(desc.maps[pid])
            base = np.ascontiguousarray(np.asarray(Cm @ th_fixed).ravel(), dtype=np.float64)
            Mv = sp.csr_matrix(Cm[:, cols]) if len(cols) else sp.csr_matrix((Cm.shape[0], 0))
            keep.append(base)
            return base, _csr_struct(Mv, keep)

        Pb, MP = split('P'); Ab, MA = split('A'); qb, Mq = split('q'); bb, Mb = split('b')
        Cd = sp.csr_matrix(desc.maps['d'])
        d_base = float((Cd @ th_fixed)[0]) if desc.nonzero_d else 0.0
        Md = _csr_struct(sp.csr_matrix(Cd[:, cols]) if (len(cols) and desc.nonzero_d)
                         else sp.csr_matrix((1, len(cols))), keep)

        def i32(a):
            a = np.ascontiguousarray(a, dtype=np.int32); keep.append(a); return a.ctypes.data_as(_ip)

        def u32(a):
            a = np.ascontiguousarray(a, dtype=np.uint32); keep.append(a); return a.ctypes.data_as(_u32p)

        def u16(a):
            a = np.ascontiguousarray(a, dtype=np.uint16); keep.append(a); return a.ctypes.data_as(_u16p)

        def f64(a):
            a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a); return a.ctypes.data_as(_dp)
        prim_idx = np.arange(desc.n_var, dtype=np.int32) if self.full_output else self.plan.prim_idx
        dual_idx = np.arange(desc.m, dtype=np.int32) if self.full_output else self.plan.dual_idx
        fam = _ConicFamily(
            n=desc.n_var, m=desc.m, is_maximization=int(desc.is_maximization),
            n_zero=cp.n_zero, n_nonneg=cp.n_nonneg, n_soc=len(cp.soc_dims), soc_dims=i32(cp.soc_dims),
            nnzP=cp.nnzP, nnzA=cp.nnzA, nnzL=cp.nnzL,
            Ap=i32(cp.Ap), Ai=i32(cp.Ai), Arp=i32(cp.Arp), Aent=i32(cp.Aent), Acol=i32(cp.Acol),
            Pp=i32(cp.Pp), Pi=i32(cp.Pi), Prp=i32(cp.Prp), Pent=i32(cp.Pent), Pcol=i32(cp.Pcol),
            Lcol=i32(cp.Lcol), ksrc_kind=i32(cp.ksrc_kind), ksrc_idx=i32(cp.ksrc_idx),
            fac_chunks=cp.fac.n_chunks, fac_triples=len(cp.fac_a), fac_ctab=i32(cp.fac.ctab),
            fac_task=u32(cp.fac.task), fac_len=u32(cp.fac.tlen), fac_a=u32(cp.fac_a), fac_b=u32(cp.fac_b),
            fac_k=u32(cp.fac_k), sol_chunks=cp.sol.n_chunks, sol_nnz=cp.sol.nnz, sol_slots=cp.sol.n_slots,
            sol_ctab=i32(cp.sol.ctab), sol_desc=u32(cp.sol.desc), sol_cols=u16(cp.sol.cols),
            sol_kind=i32(cp.sol_kind), sol_idx=i32(cp.sol_idx), sol_fpos=u16(cp.sol.final_pos),
            np_var=len(cols), P_base=_d(Pb), A_base=_d(Ab), q_base=_d(qb), b_base=_d(bb), d_base=d_base,
            map_P=MP, map_A=MA, map_q=Mq, map_b=Mb, map_d=Md,
            n_prim=len(prim_idx), prim_idx=i32(prim_idx), n_dual=len(dual_idx), dual_idx=i32(dual_idx),
            n_exp=cp.n_exp, n_pow=len(cp.pow_alpha), pow_alpha=f64(cp.pow_alpha), n_psd=len(cp.psd_dims), psd_dims=i32(cp.psd_dims))
        self.lib.check(self.lib.L.cpg_hip_create_clarabel(C.byref(fam), self.device, C.byref(self.h)),
                       'cpg_hip_create_clarabel')
        self._update_key, self._keep = key, keep
        self._var_cols, self.np_var = cols, len(cols)
        self._updated_names = names
        if getattr(self, '_placement', None) is not None:
            self.lib.check(self.lib.L.cpg_hip_set_program_placement(self.h, self._placement), 'set_program_placement')
        launch = getattr(self, '_launch', None)
        if launch:
            self.lib.check(self.lib.L.cpg_hip_set_launch(self.h, *launch), 'set_launch')

    def set_launch(self, waves_per_block=0, inst_per_wave=0, blocks_per_cu=0):
        self._launch = (waves_per_block, inst_per_wave, blocks_per_cu)
        if self.h.value:
            self.lib.check(self.lib.L.cpg_hip_set_launch(self.h, *self._launch), 'set_launch')

    def status_str(self, status) -> list:
        return [CLARABEL_STATUS.get(int(s), 'unknown') for s in status]
