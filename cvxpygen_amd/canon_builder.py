"""
cvxpy-free front door: a tiny builder that produces the OSQP-form canonicalisation of a
parametrised QP the way cvxpy's QP path + `cvxpygen/canonicalizer.py:283-332` would, i.e. as
sparse *affine maps* from the user-parameter vector theta to the canonical parameters
(P, q, d, A, l, u), with equalities first and inequalities (l = -1e30) after
(`cvxpygen/solvers/_interface.py:39-79`).

cvxpy is not installed in the build container nor on the GPU box, so benchmark / test families
are written against this builder.  When cvxpy is
importable, `cvxpygen_amd.canonicalizer` fills the same `FamilyDescriptor` from
`problem.get_problem_data(...)` instead.

A coefficient is "affine in theta":  {theta_index: weight, ..., CONST: weight}.
"""

from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple, Union

import numpy as np
import scipy.sparse as sp

from .descriptor import CPG_INF, FamilyDescriptor, UserDual, UserParam, UserVar

CONST = -1
Coef = Union[float, int, Dict[int, float]]


def _as_coef(c: Coef) -> Dict[int, float]:
    if isinstance(c, dict):
        return c
    return {CONST: float(c)}


def cmul(c: Coef, s: float) -> Dict[int, float]:
    return {k: v * s for k, v in _as_coef(c).items()}


def cadd(a: Coef, b: Coef) -> Dict[int, float]:
    out = dict(_as_coef(a))
    for k, v in _as_coef(b).items():
        out[k] = out.get(k, 0.0) + v
    return out


class ParamRef:
    """Handle to a user parameter; `ref[i, j]` / `ref[i]` -> coefficient {theta_idx: 1.0}."""

    def __init__(self, up: UserParam):
        self.up = up
        if up.kind == 'sparse':
            rows, cols = up.sparsity
            self._pos = {(int(r), int(c)): k for k, (r, c) in enumerate(zip(rows, cols))}

    def idx(self, *ij) -> int:
        up = self.up
        if up.kind == 'scalar':
            return up.col
        if len(up.shape) == 1:
            return up.col + int(ij[0])
        i, j = int(ij[0]), int(ij[1])
        if up.kind == 'diag':
            return up.col + i if i == j else None
        if up.kind == 'sparse':
            k = self._pos.get((i, j))
            return None if k is None else up.col + k
        return up.col + i + j * up.shape[0]          # F-order

    def __getitem__(self, ij) -> Dict[int, float]:
        if not isinstance(ij, tuple):
            ij = (ij,)
        k = self.idx(*ij)
        return {} if k is None else {k: 1.0}

    def structurally_nonzero(self, i, j) -> bool:
        return self.idx(i, j) is not None


class CanonBuilder:
    def __init__(self, name: str):
        self.name = name
        self.params: List[UserParam] = []
        self.param_values: Dict[str, np.ndarray] = {}
        self.NP = 0
        self.n_var = 0
        self.variables: List[UserVar] = []
        self._eq: List[Tuple[List[Tuple[int, Dict[int, float]]], Dict[int, float]]] = []
        self._ineq: List[Tuple[List[Tuple[int, Dict[int, float]]], Dict[int, float]]] = []
        self._P: Dict[Tuple[int, int], Dict[int, float]] = {}
        self._q: Dict[int, Dict[int, float]] = {}
        self._d: Dict[int, float] = {}
        self._duals: List[Tuple[str, List[Tuple[str, int]], Tuple[int, ...]]] = []
        self._soc: List[List[Tuple[List[Tuple[int, Dict[int, float]]], Dict[int, float]]]] = []
        self._exp: List[List[Tuple[List[Tuple[int, Dict[int, float]]], Dict[int, float]]]] = []
        self._pow: List[Tuple[float, List[Tuple[List[Tuple[int, Dict[int, float]]], Dict[int, float]]]]] = []
        self._psd: List[Tuple[int, List[Tuple[List[Tuple[int, Dict[int, float]]], Dict[int, float]]]]] = []
        self.is_maximization = False

    # ---- declarations -------------------------------------------------------------------------
    def param(self, name, shape=(), kind='dense', sparsity=None) -> ParamRef:
        shape = tuple(shape) if not isinstance(shape, int) else (shape,)
        if shape == ():
            kind, size = 'scalar', 1
        elif kind == 'diag':
            size = shape[0]
        elif kind == 'sparse':
            size = len(sparsity[0])
        else:
            size = int(np.prod(shape))
        up = UserParam(name, self.NP, size, shape, kind,
                       tuple(tuple(int(v) for v in s) for s in sparsity) if sparsity else None)
        self.params.append(up)
        self.NP += size
        return ParamRef(up)

    def var(self, name, shape) -> np.ndarray:
        """User variable; returns canonical-x indices shaped `shape` (F-order numbering)."""
        shape = tuple(shape) if not isinstance(shape, int) else (shape,)
        size = int(np.prod(shape))
        idx = np.arange(self.n_var, self.n_var + size)
        self.n_var += size
        self.variables.append(UserVar(name, idx.copy(), shape))
        return idx.reshape(shape, order='F')

    def aux(self, n: int) -> np.ndarray:
        idx = np.arange(self.n_var, self.n_var + n)
        self.n_var += n
        return idx

    # ---- constraints / objective --------------------------------------------------------------
    def eq(self, entries: Iterable[Tuple[int, Coef]], rhs: Coef = 0.0) -> Tuple[str, int]:
        self._eq.append(([(int(c), _as_coef(v)) for c, v in entries], _as_coef(rhs)))
        return ('eq', len(self._eq) - 1)

    def ineq(self, entries: Iterable[Tuple[int, Coef]], rhs: Coef = 0.0) -> Tuple[str, int]:
        """sum_j entries_j * x_j <= rhs"""
        self._ineq.append(([(int(c), _as_coef(v)) for c, v in entries], _as_coef(rhs)))
        return ('ineq', len(self._ineq) - 1)

    def soc(self, rows: Sequence[Tuple[Iterable[Tuple[int, Coef]], Coef]]) -> Tuple[str, int]:
        """second-order cone over len(rows) slack entries: s_k = rhs_k - sum_j entries_kj * x_j and
        s_0 >= ||s_1:||  (conic families only; `cvxpygen/solvers/clarabel.py:133-155`)"""
        self._soc.append([([(int(c), _as_coef(v)) for c, v in entries], _as_coef(rhs)) for entries, rhs in rows])
        return ('soc', len(self._soc) - 1)

    def exp_cone(self, rows: Sequence[Tuple[Iterable[Tuple[int, Coef]], Coef]]) -> Tuple[str, int]:
        """exponential cone over three slack entries s_k = rhs_k - sum_j entries_kj * x_j:  s_1 > 0, s_1 exp(s_0 / s_1) <= s_2
        (`cvxpygen/solvers/clarabel.py:136, 142`: ClarabelExponentialConeT)"""
        if len(rows) != 3:
            raise ValueError('an exponential cone has three rows')
        self._exp.append([([(int(c), _as_coef(v)) for c, v in entries], _as_coef(rhs)) for entries, rhs in rows])
        return ('exp', len(self._exp) - 1)

    def pow_cone(self, alpha: float, rows: Sequence[Tuple[Iterable[Tuple[int, Coef]], Coef]]) -> Tuple[str, int]:
        """three-dimensional power cone  s_0^alpha s_1^(1 - alpha) >= |s_2|, s_0, s_1 >= 0  (`cvxpygen/solvers/clarabel.py:139, 147`:
        ClarabelPowerConeT(alpha)); alpha is structure, not a parameter"""
        if len(rows) != 3 or not 0.0 < float(alpha) < 1.0:
            raise ValueError('a power cone has three rows and an exponent in (0, 1)')
        self._pow.append((float(alpha), [([(int(c), _as_coef(v)) for c, v in entries], _as_coef(rhs)) for entries, rhs in rows]))
        return ('pow', len(self._pow) - 1)

    def psd_cone(self, p: int, rows: Sequence[Tuple[Iterable[Tuple[int, Coef]], Coef]]) -> Tuple[str, int]:
        """PSD cone of order p over p (p + 1) / 2 slack entries: the upper triangle of the matrix column by column, the OFF-diagonal
        rows carrying the entry times sqrt 2 (`cvxpygen/solvers/clarabel.py:138, 146`: ClarabelPSDTriangleConeT(p); cvxpy applies the
        scaling when it stacks the rows for this solver)"""
        if len(rows) != p * (p + 1) // 2:
            raise ValueError('a PSD cone of order p has p (p + 1) / 2 rows')
        self._psd.append((int(p), [([(int(c), _as_coef(v)) for c, v in entries], _as_coef(rhs)) for entries, rhs in rows]))
        return ('psd', len(self._psd) - 1)

    def quad(self, i: int, j: int, coef: Coef) -> None:
        """objective += 1/2 * coef * x_i x_j * (2 if i != j else 1), i.e. P[i, j] += coef (upper)."""
        i, j = (int(i), int(j)) if i <= j else (int(j), int(i))
        self._P[(i, j)] = cadd(self._P.get((i, j), {}), coef)

    def sum_squares(self, cols: Sequence[int]) -> None:
        """objective += sum_i x_i^2  (P_ii = 2, as cvxpy's QP canonicalisation yields)."""
        for c in cols:
            self.quad(c, c, 2.0)

    def lin(self, col: int, coef: Coef) -> None:
        self._q[int(col)] = cadd(self._q.get(int(col), {}), coef)

    def const(self, coef: Coef) -> None:
        self._d = cadd(self._d, coef)

    def dual(self, name: str, rows: Sequence[Tuple[str, int]], shape) -> None:
        shape = tuple(shape) if not isinstance(shape, int) else (shape,)
        self._duals.append((name, list(rows), shape))

    # ---- finalise --------------------------------------------------------------------------------
    def _theta_idx(self, k: int) -> int:
        return self.NP if k == CONST else k

    def _map_from_rows(self, rows: List[Dict[int, float]]) -> sp.csr_matrix:
        r, c, v = [], [], []
        for i, coef in enumerate(rows):
            for k, w in sorted((self._theta_idx(k), w) for k, w in coef.items()):
                if w != 0.0:
                    r.append(i); c.append(k); v.append(w)
        return sp.csr_matrix((v, (r, c)), shape=(len(rows), self.NP + 1))

    def build(self, values: Dict[str, np.ndarray], solver: str = 'OSQP') -> FamilyDescriptor:
        conic = solver != 'OSQP'
        if (self._soc or self._exp or self._pow or self._psd) and not conic:
            raise ValueError('second-order, PSD, exponential and power cones need a conic solver')
        # (rows behind the nonnegative cone in cvxpy's order for Clarabel: soc | psd | exp | p3d)
        soc_rows = [row for cone in self._soc for row in cone] + [row for _, cone in self._psd for row in cone] + \
            [row for cone in self._exp for row in cone] + \
            [row for _, cone in self._pow for row in cone]
        n, n_eq, n_ineq = self.n_var, len(self._eq), len(self._ineq) + len(soc_rows)
        m = n_eq + n_ineq

        # theta0
        desc_tmp = FamilyDescriptor(self.name, n, n_eq, n_ineq, None, None, {}, {},
                                    np.zeros(self.NP + 1), params=self.params)
        theta0 = np.zeros(self.NP + 1)
        theta0[-1] = 1.0
        for up in self.params:
            if up.name not in values:
                raise ValueError(f'no default value for parameter {up.name}')
            theta0[up.col:up.col + up.size] = desc_tmp.flatten_param(up.name, values[up.name])

        # A: gather entries, CSC order (column-major, rows ascending)
        ent: Dict[Tuple[int, int], Dict[int, float]] = {}
        for r, (entries, _) in enumerate(self._eq + self._ineq + soc_rows):
            for c, coef in entries:
                if coef:
                    ent[(r, c)] = cadd(ent.get((r, c), {}), coef)
        keys = sorted(ent.keys(), key=lambda rc: (rc[1], rc[0]))
        A_rows = np.array([k[0] for k in keys], dtype=np.int64)
        A_cols = np.array([k[1] for k in keys], dtype=np.int64)
        map_A = self._map_from_rows([ent[k] for k in keys])
        A_vals = np.asarray(map_A @ theta0).ravel()
        A = sp.csc_matrix((A_vals, (A_rows, A_cols)), shape=(m, n))
        A.sort_indices()
        assert A.nnz == len(keys), 'duplicate / dropped entries in A'
        # explicit zeros must stay in the pattern: rebuild with indptr by hand
        indptr = np.zeros(n + 1, dtype=np.int32)
        np.add.at(indptr, A_cols + 1, 1)
        indptr = np.cumsum(indptr).astype(np.int32)
        A = sp.csc_matrix((A_vals, A_rows.astype(np.int32), indptr), shape=(m, n))

        # P upper triangular
        pkeys = sorted(self._P.keys(), key=lambda rc: (rc[1], rc[0]))
        map_P = self._map_from_rows([self._P[k] for k in pkeys])
        P_vals = np.asarray(map_P @ theta0).ravel()
        pindptr = np.zeros(n + 1, dtype=np.int32)
        if pkeys:
            np.add.at(pindptr, np.array([k[1] for k in pkeys]) + 1, 1)
        pindptr = np.cumsum(pindptr).astype(np.int32)
        P = sp.csc_matrix((P_vals, np.array([k[0] for k in pkeys], dtype=np.int32), pindptr),
                          shape=(n, n))

        # q, d
        map_q = self._map_from_rows([self._q.get(i, {}) for i in range(n)])
        map_d = self._map_from_rows([self._d])

        # l (n_eq rows, padded later with -inf), u (m rows)
        map_l = self._map_from_rows([rhs for _, rhs in self._eq])
        map_u = self._map_from_rows([rhs for _, rhs in self._eq + self._ineq + soc_rows])

        maps = {'P': map_P, 'q': map_q, 'd': map_d, 'A': map_A, 'l': map_l, 'u': map_u}
        cones = None
        if conic:       # Ax + s = b, s in K: the same rows, one right-hand side b (clarabel.py:19-46)
            maps = {'P': map_P, 'q': map_q, 'd': map_d, 'A': map_A, 'b': map_u}
            cones = {'zero': n_eq, 'nonneg': len(self._ineq), 'soc': [len(c) for c in self._soc]}
            if self._psd:
                cones['psd'] = [q for q, _ in self._psd]
            if self._exp:
                cones['exp'] = len(self._exp)
            if self._pow:
                cones['pow'] = [a for a, _ in self._pow]
        # p_id_to_changes: depends on any non-constant theta column
        # (`cvxpygen/canonicalizer.py:324`)
        changes = {}
        for pid, Cm in maps.items():
            changes[pid] = bool(Cm.tocsc()[:, :self.NP].nnz > 0)

        duals = []
        for name, rows, shape in self._duals:
            idx = np.array([r if kind == 'eq' else n_eq + r for kind, r in rows], dtype=np.int32)
            duals.append(UserDual(name, idx, shape, 'z' if conic else 'y'))

        nonzero_d = bool(map_d.nnz > 0)
        return FamilyDescriptor(
            name=self.name, n_var=n, n_eq=n_eq, n_ineq=n_ineq, P=P, A=A, maps=maps,
            changes=changes, theta0=theta0, params=self.params, variables=self.variables,
            duals=duals, is_maximization=self.is_maximization, nonzero_d=nonzero_d, solver=solver,
            cones=cones)


def canon_lu(desc: FamilyDescriptor, canon: Dict[str, np.ndarray]):
    """Full-length l, u as handed to OSQP: l padded with -1e30 for the inequality rows
    (`cvxpygen/solvers/_interface.py:76-79`, `cvxpygen/utils.py:213-228`)."""
    l = np.concatenate([canon['l'], -CPG_INF * np.ones(desc.n_ineq)])
    u = canon['u'].copy()
    return np.clip(l, -CPG_INF, CPG_INF), np.clip(u, -CPG_INF, CPG_INF)
