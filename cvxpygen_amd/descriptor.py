"""
Problem-family descriptor: everything the generated solver needs to know about ONE
parametrised problem family, with no cvxpy objects inside.

It is the cvxpy-free equivalent of what the reference keeps in `Canon`
(`cvxpygen/mappings.py:129-138`): `ParameterCanon` (canonical-parameter defaults, the sparse
affine maps `p_id_to_mapping`, `p_id_to_changes`, `is_maximization`, `nonzero_d`;
`cvxpygen/mappings.py:33-48`), `ParameterInfo` (user parameter name -> column/size/shape;
`cvxpygen/mappings.py:50-68`), `PrimalVariableInfo` / `DualVariableInfo`
(`cvxpygen/mappings.py:70-98`) plus the dimensions `n_var, n_eq, n_ineq` a
`SolverInterface` carries (`cvxpygen/solvers/_interface.py:87-100`).

Conventions kept from the reference:
  * theta = all user parameters concatenated, each flattened in F-order (sparse parameters:
    stored non-zeros only; diag parameters: the diagonal), with a trailing constant 1
    (`cvxpygen/canonicalizer.py:226-271`, `cvxpygen/templates/cpg_solver.py.jinja2:26-34`).
  * every canonical parameter p in {P, q, d, A, l, u} is  p = C_p @ theta  with C_p a CSR
    matrix (`cvxpygen/canonicalizer.py:283-332`, `cvxpygen/utils.py:279-294`).
  * OSQP canonical form: minimise 1/2 x'Px + q'x + d  s.t.  l <= Ax <= u, rows ordered
    equalities first (l = u) then inequalities with l = -inf
    (`cvxpygen/solvers/_interface.py:39-79`); P upper-triangular CSC, A CSC.
  * +-inf is stored as +-1e30 (`cvxpygen/utils.py:213-228`).
  * conic families (solver 'CLARABEL'; `cvxpygen/solvers/clarabel.py:19-46, 133-155`): canonical
    parameters {P, q, d, A, b} for  minimise 1/2 x'Px + q'x + d  s.t.  Ax + s = b, s in K, with the
    rows ordered zero cone, nonnegative cone, second-order cones, PSD cones, exponential cones, 3-d power cones
    (cvxpy's stacking for this solver); `cones` holds {'zero': int, 'nonneg': int, 'soc': [dims]} and, when present,
    'psd': [matrix orders], 'exp': count, 'pow': [exponents]; n_eq = zero-cone rows, n_ineq = all other rows;
    the dual vector is called 'z' (`cvxpygen/solvers/clarabel.py:33-35`).
"""

from __future__ import annotations

import io
import json
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import scipy.sparse as sp

CPG_INF = 1e30

CANON_IDS_QP = ('P', 'q', 'd', 'A', 'l', 'u')
CANON_IDS_CONIC = ('P', 'q', 'd', 'A', 'b')
CANON_IDS_ECOS = ('c', 'd', 'A', 'b', 'G', 'h')          # cvxpygen/solvers/ecos.py:20
CANON_IDS_ALL = ('P', 'q', 'd', 'A', 'l', 'u', 'b', 'c', 'G', 'h')


@dataclass
class UserParam:
    """One user-defined parameter (a slice of theta)."""
    name: str
    col: int                 # first column in theta
    size: int                # number of stored entries (nnz for sparse, n for diag)
    shape: Tuple[int, ...]   # user-facing shape
    kind: str = 'dense'      # 'dense' | 'diag' | 'sparse' | 'scalar'
    sparsity: Tuple[Tuple[int, ...], Tuple[int, ...]] = None  # (rows, cols) for kind == 'sparse'


@dataclass
class UserVar:
    """One user-defined primal variable: indices into the canonical x."""
    name: str
    indices: np.ndarray
    shape: Tuple[int, ...]
    sym: bool = False


@dataclass
class UserDual:
    """One user-facing dual variable (one per constraint): indices into the canonical y."""
    name: str
    indices: np.ndarray
    shape: Tuple[int, ...]
    vec: str = 'y'


@dataclass
class FamilyDescriptor:
    name: str
    n_var: int
    n_eq: int
    n_ineq: int
    # patterns (scipy CSC with *default* values = values at theta0)
    P: sp.csc_matrix                     # upper triangular, n_var x n_var
    A: sp.csc_matrix                     # (n_eq + n_ineq) x n_var
    # canonical-parameter maps, CSR, shape (size_p, NP + 1)
    maps: Dict[str, sp.csr_matrix]
    changes: Dict[str, bool]             # p_id_to_changes
    theta0: np.ndarray                   # NP + 1, last entry 1
    params: List[UserParam] = field(default_factory=list)
    variables: List[UserVar] = field(default_factory=list)
    duals: List[UserDual] = field(default_factory=list)
    is_maximization: bool = False
    nonzero_d: bool = True
    solver: str = 'OSQP'
    cones: Dict[str, object] = None      # conic families only

    # ---- derived --------------------------------------------------------------------------
    @property
    def m(self) -> int:
        return self.n_eq + self.n_ineq

    @property
    def NP(self) -> int:
        return int(self.theta0.shape[0] - 1)

    @property
    def n_prim_user(self) -> int:
        return int(sum(v.indices.size for v in self.variables))

    @property
    def n_dual_user(self) -> int:
        return int(sum(d.indices.size for d in self.duals))

    def param(self, name: str) -> UserParam:
        for p in self.params:
            if p.name == name:
                return p
        raise AttributeError(f"{name} is not a parameter.")

    @property
    def param_names(self) -> List[str]:
        return [p.name for p in self.params]

    def user_p_name_to_canon_outdated(self) -> Dict[str, List[str]]:
        """adjacency user parameter -> canonical parameters that depend on it
        (`cvxpygen/canonicalizer.py:117-120, 439-446`)."""
        out = {}
        for p in self.params:
            deps = []
            for pid in CANON_IDS_ALL:
                if pid not in self.maps:
                    continue
                Cm = self.maps[pid].tocsc()
                if Cm[:, p.col:p.col + p.size].nnz > 0:
                    deps.append(pid)
            out[p.name] = deps
        return out

    def canon_at(self, theta: np.ndarray) -> Dict[str, np.ndarray]:
        """Canonical parameter values at theta (host, numpy): the batched-free restatement of
        `cpg_canonicalize_<p>` (`cvxpygen/utils.py:279-294`) used by setup code and tests."""
        theta = np.asarray(theta, dtype=np.float64)
        out = {}
        for pid, Cm in self.maps.items():
            out[pid] = np.asarray(Cm @ theta).ravel()
        return out

    def default_canon(self) -> Dict[str, np.ndarray]:
        return self.canon_at(self.theta0)

    def flatten_param(self, name: str, value) -> np.ndarray:
        """User value -> stored entries, as `get_param_value` does
        (`cvxpygen/templates/cpg_solver.py.jinja2:26-34`)."""
        p = self.param(name)
        v = np.asarray(value, dtype=np.float64)
        if p.kind == 'scalar' or p.size == 1 and v.size == 1:
            return v.reshape(1)
        if p.kind == 'diag':
            if v.ndim == 2:
                return np.diag(v).copy()
            return v.reshape(p.size)
        if p.kind == 'sparse':
            if v.ndim == 1 and v.size == p.size:
                return v.copy()
            rows, cols = p.sparsity
            return v[np.asarray(rows), np.asarray(cols)].astype(np.float64)
        return v.reshape(p.shape).flatten(order='F')

    def theta_from_values(self, values: Dict[str, np.ndarray]) -> np.ndarray:
        th = self.theta0.copy()
        for name, val in values.items():
            p = self.param(name)
            th[p.col:p.col + p.size] = self.flatten_param(name, val)
        return th

    # ---- (de)serialisation ---------------------------------------------------------------
    def save(self, path: str) -> None:
        meta = {
            'name': self.name, 'n_var': self.n_var, 'n_eq': self.n_eq, 'n_ineq': self.n_ineq,
            'is_maximization': self.is_maximization, 'nonzero_d': self.nonzero_d,
            'solver': self.solver, 'changes': self.changes, 'cones': self.cones,
            'params': [dict(name=p.name, col=p.col, size=p.size, shape=list(p.shape), kind=p.kind,
                            sparsity=[list(map(int, s)) for s in p.sparsity] if p.sparsity else None)
                       for p in self.params],
            'variables': [dict(name=v.name, shape=list(v.shape), sym=v.sym) for v in self.variables],
            'duals': [dict(name=d.name, shape=list(d.shape), vec=d.vec) for d in self.duals],
            'map_ids': list(self.maps.keys()),
        }
        arrays = {'theta0': self.theta0}
        for tag, M in (('P', self.P), ('A', self.A)):
            M = M.tocsc()
            arrays[f'{tag}_indptr'] = M.indptr.astype(np.int32)
            arrays[f'{tag}_indices'] = M.indices.astype(np.int32)
            arrays[f'{tag}_data'] = M.data.astype(np.float64)
        for pid, Cm in self.maps.items():
            Cm = Cm.tocsr()
            arrays[f'map_{pid}_indptr'] = Cm.indptr.astype(np.int32)
            arrays[f'map_{pid}_indices'] = Cm.indices.astype(np.int32)
            arrays[f'map_{pid}_data'] = Cm.data.astype(np.float64)
            arrays[f'map_{pid}_shape'] = np.asarray(Cm.shape, dtype=np.int64)
        for v in self.variables:
            arrays[f'var_{v.name}'] = np.asarray(v.indices, dtype=np.int32)
        for d in self.duals:
            arrays[f'dual_{d.name}'] = np.asarray(d.indices, dtype=np.int32)
        arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez(path, **arrays)

    @staticmethod
    def load(path: str) -> 'FamilyDescriptor':
        z = np.load(path, allow_pickle=False)
        meta = json.loads(bytes(z['meta']).decode())
        n_var, m = meta['n_var'], meta['n_eq'] + meta['n_ineq']
        P = sp.csc_matrix((z['P_data'], z['P_indices'], z['P_indptr']), shape=(n_var, n_var))
        A = sp.csc_matrix((z['A_data'], z['A_indices'], z['A_indptr']), shape=(m, n_var))
        maps = {}
        for pid in meta['map_ids']:
            maps[pid] = sp.csr_matrix((z[f'map_{pid}_data'], z[f'map_{pid}_indices'],
                                       z[f'map_{pid}_indptr']), shape=tuple(z[f'map_{pid}_shape']))
        params = [UserParam(p['name'], p['col'], p['size'], tuple(p['shape']), p['kind'],
                            tuple(tuple(s) for s in p['sparsity']) if p['sparsity'] else None)
                  for p in meta['params']]
        variables = [UserVar(v['name'], z[f"var_{v['name']}"], tuple(v['shape']), v['sym'])
                     for v in meta['variables']]
        duals = [UserDual(d['name'], z[f"dual_{d['name']}"], tuple(d['shape']), d['vec'])
                 for d in meta['duals']]
        return FamilyDescriptor(
            name=meta['name'], n_var=n_var, n_eq=meta['n_eq'], n_ineq=meta['n_ineq'], P=P, A=A,
            maps=maps, changes=meta['changes'], theta0=z['theta0'], params=params,
            variables=variables, duals=duals, is_maximization=meta['is_maximization'],
            nonzero_d=meta['nonzero_d'], solver=meta['solver'], cones=meta.get('cones'))
