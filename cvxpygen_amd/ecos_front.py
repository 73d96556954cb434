"""
ECOS-form families (`cvxpygen/solvers/ecos.py`) on the conic interior-point kernel.

The reference's ECOS interface canonicalises to

        minimize c'x + d     s.t.   A x = b,   G x + s = h,   s in  R+^l  x  SOC(q_1) x ... x SOC(q_k)

(`ecos.py:20-22`: canonical parameters c, d, A, b, G, h; duals split into y for A x = b and z for the cone rows,
`ecos.py:75-77`; `ECOS_setup(n, m, p, l, ncones, q, e, G, A, c, h, b)`, `ecos.py:86-97`) and links the ECOS C solver.
That form is the conic kernel's own form with P = 0, rows [A; G], right-hand side [b; h] and the cones
zero(p) x nonneg(l) x soc(q): this module stacks it, translates setting names and exit flags, and splits the duals.

PARITY UNPINNED, and weaker than for the other solvers: the kernel runs the Clarabel-style interior point method
(cvxpygen_amd/csrc/cpg_clarabel_kernel.h), not ECOS's -- neither ECOS nor cvxpy is in the build image.  What an
ECOS-generated solver returns is matched at the level of the optimisation problem (solution, objective, duals,
infeasibility verdicts to the tolerances below), not iteration by iteration: `iter` counts this kernel's iterations.
"""

from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from .descriptor import FamilyDescriptor, UserDual
from .runtime import BatchResult

# `ecos.py:61-69`: name -> (kernel setting, default)
ECOS_SETTINGS = {
    'feastol': ('tol_feas', 1e-8), 'abstol': ('tol_gap_abs', 1e-8), 'reltol': ('tol_gap_rel', 1e-8),
    'feastol_inacc': ('reduced_tol_feas', 1e-4), 'abstol_inacc': ('reduced_tol_gap_abs', 5e-5),
    'reltol_inacc': ('reduced_tol_gap_rel', 5e-5), 'maxit': ('max_iter', 100),
}
ECOS_SETTING_ALIASES = {'max_iters': 'maxit'}                   # name_cvxpy, ecos.py:68
# ecos.h exit codes: ECOS_OPTIMAL 0, ECOS_PINF 1, ECOS_DINF 2, + ECOS_INACC_OFFSET 10, ECOS_MAXIT -1, ECOS_NUMERICS -2
ECOS_FROM_KERNEL_STATUS = {0: -2, 1: 0, 2: 1, 3: 2, 4: 10, 5: 11, 6: 12, 7: -1, 8: -1, 9: -2, 10: -2}
ECOS_DOCU = 'https://github.com/embotech/ecos/wiki/Usage-from-C'


def _blocks(desc: FamilyDescriptor):
    """per entry of the stacked pattern [A; G] (CSC order): True where it belongs to the A block"""
    Ast = sp.csc_matrix(desc.A)
    return Ast, Ast.indices < desc.n_eq


def ecos_from_conic(desc: FamilyDescriptor) -> FamilyDescriptor:
    """The ECOS form of a conic family without quadratic objective (what cvxpy hands ECOS for the same problem):
    canonical parameters c, d, A, b, G, h; the stacked pattern [A; G] is kept in `.A`."""
    if desc.solver != 'CLARABEL' or not desc.cones:
        raise ValueError('expects a conic (CLARABEL-form) family')
    if sp.csc_matrix(desc.P).nnz:
        raise ValueError('ECOS has no quadratic objective: the family must be canonicalised with its squares as cones')
    Ast, in_A = _blocks(desc)
    p = desc.cones['zero']
    mA = sp.csr_matrix(desc.maps['A'])
    mb = sp.csr_matrix(desc.maps['b'])
    maps = {'c': sp.csr_matrix(desc.maps['q']), 'd': sp.csr_matrix(desc.maps['d']),
            'A': mA[np.nonzero(in_A)[0]], 'b': mb[:p], 'G': mA[np.nonzero(~in_A)[0]], 'h': mb[p:]}
    changes = {pid: bool(Cm.tocsc()[:, :desc.NP].nnz > 0) for pid, Cm in maps.items()}
    duals = []
    for u in desc.duals:                                         # dual_var_split: y (rows of A) / z (rows of G)
        idx = np.asarray(u.indices)
        if (idx < p).all():
            duals.append(UserDual(u.name, idx.astype(np.int32), u.shape, 'y'))
        elif (idx >= p).all():
            duals.append(UserDual(u.name, (idx - p).astype(np.int32), u.shape, 'z'))
        else:
            raise ValueError(f'dual {u.name} mixes equality and cone rows')
    return FamilyDescriptor(name=desc.name, n_var=desc.n_var, n_eq=p, n_ineq=desc.m - p, P=sp.csc_matrix((desc.n_var, desc.n_var)),
                            A=Ast, maps=maps, changes=changes, theta0=desc.theta0.copy(), params=list(desc.params),
                            variables=list(desc.variables), duals=duals, is_maximization=desc.is_maximization,
                            nonzero_d=desc.nonzero_d, solver='ECOS', cones=dict(desc.cones))


def conic_from_ecos(desc: FamilyDescriptor) -> FamilyDescriptor:
    """stack an ECOS-form family into the kernel's form: rows [A; G], right-hand side [b; h], P = 0"""
    if desc.solver != 'ECOS':
        raise ValueError('expects an ECOS-form family')
    Ast, in_A = _blocks(desc)
    nnz = Ast.nnz
    pos = np.empty(nnz, dtype=np.int64)
    pos[np.nonzero(in_A)[0]] = np.arange(int(in_A.sum()))
    pos[np.nonzero(~in_A)[0]] = int(in_A.sum()) + np.arange(int((~in_A).sum()))
    both = sp.vstack([sp.csr_matrix(desc.maps['A']), sp.csr_matrix(desc.maps['G'])]).tocsr()
    maps = {'P': sp.csr_matrix((0, desc.theta0.shape[0])), 'q': sp.csr_matrix(desc.maps['c']),
            'd': sp.csr_matrix(desc.maps['d']), 'A': both[pos],
            'b': sp.vstack([sp.csr_matrix(desc.maps['b']), sp.csr_matrix(desc.maps['h'])]).tocsr()}
    changes = {'P': False, 'q': bool(desc.changes.get('c', False)), 'd': bool(desc.changes.get('d', False)),
               'A': bool(desc.changes.get('A', False) or desc.changes.get('G', False)),
               'b': bool(desc.changes.get('b', False) or desc.changes.get('h', False))}
    m = desc.n_eq + desc.n_ineq
    duals = [UserDual('yz', np.arange(m, dtype=np.int32), (m,), 'z')]
    # (ClarabelInterface's bookkeeping counts every row as an "equality": solvers/clarabel.py)
    return FamilyDescriptor(name=desc.name + '_stacked', n_var=desc.n_var, n_eq=m, n_ineq=0,
                            P=sp.csc_matrix((desc.n_var, desc.n_var)), A=Ast, maps=maps, changes=changes,
                            theta0=desc.theta0.copy(), params=list(desc.params), variables=list(desc.variables),
                            duals=duals, is_maximization=desc.is_maximization, nonzero_d=desc.nonzero_d,
                            solver='CLARABEL', cones=dict(desc.cones))


class EcosBatchSolver:
    """`cpg_solve` of an ECOS-generated solver for B instances: ECOS's setting names, exit flags and y / z split
    around the conic interior-point kernel."""

    def __init__(self, desc: FamilyDescriptor, device: int = 0, lib_path: Optional[str] = None):
        from .conic_runtime import ConicBatchSolver
        self.desc = desc
        self.conic_desc = conic_from_ecos(desc)
        self.conic = ConicBatchSolver(self.conic_desc, device=device, lib_path=lib_path, full_output=True)
        self.adaptive_rho = False

    def close(self):
        self.conic.close()

    def gradient(self, *a, **k):
        raise NotImplementedError('a conic family is differentiated through its OSQP form '
                                  '(cvxpygen/generator.py:76-80): cvxpygen_amd.two_stage')

    def kernel_settings(self, kwargs: Dict[str, float]) -> Dict[str, float]:
        out = {knl: dflt for knl, dflt in ECOS_SETTINGS.values()}      # ECOS defaults, not Clarabel's
        for k, v in kwargs.items():
            k = ECOS_SETTING_ALIASES.get(k, k)
            if k not in ECOS_SETTINGS:
                raise AttributeError(f'Solver setting "{k}" not available.')
            out[ECOS_SETTINGS[k][0]] = float(v)
        return out

    def solve(self, params: Optional[Dict[str, np.ndarray]] = None, updated_params: Optional[Sequence[str]] = None,
              B: Optional[int] = None, theta_var: Optional[np.ndarray] = None, **kwargs) -> BatchResult:
        d = self.desc
        r = self.conic.solve(params, updated_params=updated_params, B=B, theta_var=theta_var,
                             **self.kernel_settings(kwargs))
        x, yz = np.asarray(r.sol_x), np.asarray(r.sol_y)
        p = d.n_eq
        vec = {'y': yz[:, :p], 'z': yz[:, p:]}
        Bn = x.shape[0]
        prim = {v.name: x[:, v.indices].reshape((Bn,) + tuple(v.shape), order='F') if v.shape else x[:, v.indices[0]]
                for v in d.variables}
        dual = {u.name: vec[u.vec][:, u.indices].reshape((Bn,) + tuple(u.shape)) if u.shape else vec[u.vec][:, u.indices[0]]
                for u in d.duals}
        flag = np.array([ECOS_FROM_KERNEL_STATUS.get(int(s), -2) for s in r.status], dtype=np.int32)
        out = BatchResult(prim=prim, dual=dual, obj_val=r.obj_val, iter=r.iter, status=flag, pri_res=r.pri_res,
                          dua_res=r.dua_res, kernel_ms=r.kernel_ms)
        out.sol_x, out.sol_y = x, yz
        out.prim_flat = np.concatenate([x[:, v.indices] for v in d.variables], axis=1) if d.variables else x[:, :0]
        return out
