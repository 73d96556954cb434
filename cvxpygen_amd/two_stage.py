"""
Two-stage differentiation: a QP family SOLVED by the conic interior-point kernel and DIFFERENTIATED through its
OSQP form.

The reference, for `generate_code(problem, solver=<conic solver>, gradient=True)` (`cvxpygen/generator.py:76-80`):
  1. canonicalises the problem for OSQP (`canonicalizer.py:60`): theta -> (P, q, d, A, l, u);
  2. builds an OSQP-shaped cvxpy problem whose PARAMETERS are those canonical parameters
     (`_get_osqp_problem`, `canonicalizer.py:334-361`): P must be constant and diagonal,
         minimize 0.5 * sum_squares(multiply(x, sqrt(diag P))) + q'x   s.t.   l <= A[:n_eq] x,   A x <= u;
  3. canonicalises THAT for the conic solver and composes the two affine maps (`_merge`, `canonicalizer.py:363-406`);
  4. solves with the conic solver and differentiates with the OSQP-form adjoint (`cpg_osqp_grad_*`).
Step 3 runs cvxpy, which the build image does not have.  For the only shape step 2 can produce the conic form is
written down here directly (Clarabel takes the quadratic objective as it is; every constraint is an inequality):

    minimize 0.5 x'Px + q'x      s.t.   [-A_eq; A] x + s = [-l; u],   s >= 0         (n_eq + m nonnegative rows)

with the multipliers z = [z_low; z_up] >= 0.  Stationarity P x + q - A_eq' z_low + A' z_up = 0 is OSQP's
P x + q + A'y = 0 with  y = z_up - [z_low; 0]:  that y and x go into the OSQP-form adjoint kernel unchanged.
The ORDER of the rows cvxpy would emit for step 3 is not pinned (no cvxpy): it only permutes z, never x or y.
"""

from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from .descriptor import FamilyDescriptor, UserDual
from .runtime import BatchResult, BatchSolver


def qp_to_conic(desc: FamilyDescriptor) -> FamilyDescriptor:
    """The Clarabel-form family of the same QP (module docstring): same variables and parameters, rows
    [-A_eq; A], right-hand side [-l; u], all in the nonnegative cone."""
    if desc.solver != 'OSQP':
        raise ValueError('two-stage differentiation starts from the OSQP form of the problem')
    if desc.changes.get('P', False):
        # `canonicalizer.py:338-342`
        raise ValueError('Problem does not follow extended DPP rules for differentiation with general solvers '
                         '(other than OSQP). Quadratics cannot be multiplied with parameters.')
    P = sp.coo_matrix(desc.P)
    if np.any(P.row != P.col):
        raise ValueError('two-stage differentiation needs a diagonal P (canonicalizer.py:344-345)')
    n, m, n_eq = desc.n_var, desc.m, desc.n_eq
    A = sp.csc_matrix(desc.A)
    NP1 = desc.theta0.shape[0]
    # entries of the conic A in CSC order: per column the negated equality-row entries, then all entries shifted
    rows_c, src, sign = [], [], []
    indptr = [0]
    for j in range(n):
        a, e = A.indptr[j], A.indptr[j + 1]
        r = A.indices[a:e]
        k = np.arange(a, e)
        eq = r < n_eq
        rows_c.extend(r[eq].tolist()); src.extend(k[eq].tolist()); sign.extend([-1.0] * int(eq.sum()))
        rows_c.extend((r + n_eq).tolist()); src.extend(k.tolist()); sign.extend([1.0] * len(k))
        indptr.append(len(rows_c))
    src = np.asarray(src, dtype=np.int64); sign = np.asarray(sign)
    A_c = sp.csc_matrix((sign * A.data[src], np.asarray(rows_c, dtype=np.int32), np.asarray(indptr, dtype=np.int32)),
                        shape=(n_eq + m, n))
    pick = sp.csr_matrix((sign, (np.arange(len(src)), src)), shape=(len(src), A.nnz))
    map_A = sp.csr_matrix(pick @ sp.csr_matrix(desc.maps['A']))
    map_b = sp.vstack([-sp.csr_matrix(desc.maps['l']), sp.csr_matrix(desc.maps['u'])]).tocsr()
    maps = {'P': sp.csr_matrix(desc.maps['P']), 'q': sp.csr_matrix(desc.maps['q']), 'd': sp.csr_matrix(desc.maps['d']),
            'A': map_A, 'b': map_b}
    changes = {'P': False, 'q': bool(desc.changes.get('q', False)), 'd': bool(desc.changes.get('d', False)),
               'A': bool(desc.changes.get('A', False)),
               'b': bool(desc.changes.get('l', False) or desc.changes.get('u', False))}
    assert map_b.shape == (n_eq + m, NP1)
    duals = [UserDual(name='z', indices=np.arange(n_eq + m, dtype=np.int32), shape=(n_eq + m,), vec='z')]
    # ClarabelInterface counts every row as "equality" in its bookkeeping (solvers/clarabel.py; n_ineq = 0)
    return FamilyDescriptor(name=desc.name + '_conic', n_var=n, n_eq=n_eq + m, n_ineq=0, P=sp.csc_matrix(desc.P), A=A_c,
                            maps=maps, changes=changes, theta0=desc.theta0.copy(), params=list(desc.params),
                            variables=list(desc.variables), duals=duals, is_maximization=desc.is_maximization,
                            nonzero_d=desc.nonzero_d, solver='CLARABEL',
                            cones={'zero': 0, 'nonneg': n_eq + m, 'soc': []})


def qp_duals_from_conic(z: np.ndarray, n_eq: int) -> np.ndarray:
    """y of the OSQP form from the conic multipliers z = [z_low (n_eq); z_up (m)]"""
    y = np.array(z[..., n_eq:], dtype=np.float64)
    y[..., :n_eq] -= z[..., :n_eq]
    return y


class TwoStageBatchSolver:
    """solve(): conic interior-point kernel on the Clarabel form; gradient(): OSQP-form adjoint kernel at that
    solution.  Results are reported in the terms of the OSQP form (the user's variables and constraints)."""

    def __init__(self, desc: FamilyDescriptor, device: int = 0, lib_path: Optional[str] = None):
        from .conic_runtime import ConicBatchSolver
        self.desc = desc
        self.conic_desc = qp_to_conic(desc)
        self.conic = ConicBatchSolver(self.conic_desc, device=device, lib_path=lib_path, full_output=True)
        self.qp = BatchSolver(desc, device=device, lib_path=lib_path, full_output=True)
        self.adaptive_rho = False

    def close(self):
        self.conic.close(); self.qp.close()

    def status_str(self, status):
        return self.conic.status_str(status)

    def solve(self, params: Optional[Dict[str, np.ndarray]] = None, updated_params: Optional[Sequence[str]] = None,
              B: Optional[int] = None, theta_var: Optional[np.ndarray] = None, **kwargs) -> BatchResult:
        d = self.desc
        r = self.conic.solve(params, updated_params=updated_params, B=B, theta_var=theta_var, **kwargs)
        x = np.asarray(r.sol_x)
        y = qp_duals_from_conic(np.asarray(r.sol_y), d.n_eq)
        prim = {v.name: x[:, v.indices].reshape((x.shape[0],) + tuple(v.shape), order='F') if v.shape else x[:, v.indices[0]]
                for v in d.variables}
        dual = {u.name: y[:, u.indices].reshape((y.shape[0],) + tuple(u.shape)) if u.shape else y[:, u.indices[0]]
                for u in d.duals}
        out = BatchResult(prim=prim, dual=dual, obj_val=r.obj_val, iter=r.iter, status=r.status, pri_res=r.pri_res,
                          dua_res=r.dua_res, kernel_ms=r.kernel_ms)
        out.sol_x, out.sol_y = x, y
        out.prim_flat = np.concatenate([x[:, v.indices] for v in d.variables], axis=1) if d.variables else x[:, :0]
        out.state = None
        return out

    def gradient(self, params, sol_x, sol_y, dvars, updated_params=None, **kwargs):
        return self.qp.gradient(params, sol_x, sol_y, dvars, updated_params=updated_params, **kwargs)
