"""
Code-generation-time setup of the embedded-OSQP workspace for one problem family (host, numpy).

This is the work the reference delegates to `osqp.OSQP().setup(P, q, A, l, u)` +
`.codegen(..., parameters='matrices')` (`cvxpygen/solvers/osqp.py:126-131`): Ruiz equilibration
of the data at the code-generation-time parameter values, the per-row step-size vector, the
quasi-definite KKT matrix, a fill-reducing permutation and its LDL' factorisation.  The third-party
OSQP / QDLDL sources are not part of the reference checkout; the algorithm is
restated from the OSQP paper (Stellato et al., Math. Prog. Comp. 2020, sections 3, 5) and the public
description of QDLDL.

Everything here is product code (it builds the plan the HIP kernels consume); the independent
scalar-C restatement used as test oracle lives in `oracle/`.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import scipy.sparse as sp

from . import ordering as _ord

OSQP_INFTY = 1e30
MIN_SCALING = 1e-4
MAX_SCALING = 1e4
RHO_MIN = 1e-6
RHO_MAX = 1e6
RHO_TOL = 1e-4
RHO_EQ_OVER_RHO_INEQ = 1e3

# OSQP library defaults that the reference never overrides (setup() is called with no settings,
# `cvxpygen/solvers/osqp.py:126-129`) + the defaults it re-applies on every solve
# (`cvxpygen/solvers/osqp.py:102-115`, `cvxpygen/templates/cpg_solver.py.jinja2:55`).
DEFAULT_SETTINGS = dict(
    rho=0.1, sigma=1e-6, alpha=1.6, scaling=10,
    max_iter=4000, eps_abs=1e-3, eps_rel=1e-3, eps_prim_inf=1e-4, eps_dual_inf=1e-4,
    scaled_termination=0, check_termination=25, warm_starting=1,
    adaptive_rho=0, adaptive_rho_interval=50, adaptive_rho_tolerance=5.0, check_dualgap=0,
)


def _limit_scaling(v: np.ndarray) -> np.ndarray:
    v = np.where(v < MIN_SCALING, 1.0, v)
    return np.where(v > MAX_SCALING, MAX_SCALING, v)


@dataclass
class Scaling:
    D: np.ndarray
    E: np.ndarray
    c: float

    @property
    def Dinv(self):
        return 1.0 / self.D

    @property
    def Einv(self):
        return 1.0 / self.E

    @property
    def cinv(self):
        return 1.0 / self.c


def ruiz_scale(P: sp.csc_matrix, q: np.ndarray, A: sp.csc_matrix, iters: int = 10
               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, Scaling]:
    """Ruiz equilibration of [[P, A'], [A, 0]] with cost scaling (OSQP paper, Algorithm 2).
    P upper-triangular CSC.  Returns scaled (P.data, q, A.data) and the scaling."""
    n, m = P.shape[0], A.shape[0]
    Pc = sp.coo_matrix(P)
    Ac = sp.coo_matrix(A)
    # COO of a CSC keeps the data order of the CSC
    Pr, Pcn, Px = Pc.row, Pc.col, P.data.astype(np.float64).copy()
    Ar, Acn, Ax = Ac.row, Ac.col, A.data.astype(np.float64).copy()
    q = q.astype(np.float64).copy()
    D = np.ones(n)
    E = np.ones(m)
    c = 1.0
    for _ in range(iters):
        # column inf-norms of the KKT matrix
        Dt = np.zeros(n)
        if Px.size:
            ap = np.abs(Px)
            np.maximum.at(Dt, Pcn, ap)
            np.maximum.at(Dt, Pr, ap)
        Et = np.zeros(m)
        if Ax.size:
            aa = np.abs(Ax)
            np.maximum.at(Dt, Acn, aa)
            np.maximum.at(Et, Ar, aa)
        Dt = 1.0 / np.sqrt(_limit_scaling(Dt))
        Et = 1.0 / np.sqrt(_limit_scaling(Et))
        Px = Px * Dt[Pr] * Dt[Pcn]
        Ax = Ax * Et[Ar] * Dt[Acn]
        q = q * Dt
        D *= Dt
        E *= Et
        # cost scaling
        Pn = np.zeros(n)
        if Px.size:
            ap = np.abs(Px)
            np.maximum.at(Pn, Pcn, ap)
            np.maximum.at(Pn, Pr, ap)
        c_tmp = Pn.mean() if n else 0.0
        qn = float(_limit_scaling(np.array([np.max(np.abs(q)) if n else 0.0]))[0])
        c_tmp = float(_limit_scaling(np.array([max(c_tmp, qn)]))[0])
        c_tmp = 1.0 / c_tmp
        Px = Px * c_tmp
        q = q * c_tmp
        c *= c_tmp
    return Px, q, Ax, Scaling(D, E, c)


def compute_rho_vec(l: np.ndarray, u: np.ndarray, rho: float) -> Tuple[np.ndarray, np.ndarray]:
    """Per-row step sizes from the (scaled) bounds: unconstrained rows -> RHO_MIN, equality rows
    -> 1e3 * rho, inequality rows -> rho.  Returns (rho_vec, constr_type)."""
    rho = min(max(rho, RHO_MIN), RHO_MAX)
    ctype = np.zeros(l.shape[0], dtype=np.int32)
    uncon = (l < -OSQP_INFTY * MIN_SCALING) & (u > OSQP_INFTY * MIN_SCALING)
    eq = (~uncon) & ((u - l) < RHO_TOL)
    ctype[uncon] = -1
    ctype[eq] = 1
    rv = np.full(l.shape[0], rho)
    rv[uncon] = RHO_MIN
    rv[eq] = RHO_EQ_OVER_RHO_INEQ * rho
    return rv, ctype


def kkt_upper(P: sp.csc_matrix, A: sp.csc_matrix, sigma: float, rho_vec: np.ndarray
              ) -> Tuple[sp.csc_matrix, Dict[str, np.ndarray]]:
    """K = [[P + sigma I, A'], [A, -diag(1/rho_vec)]], upper triangle, CSC.  Also returns index
    maps from the entries of P / A / the two diagonals into K.data so that the numeric values
    can be refreshed without touching the pattern."""
    n, m = P.shape[0], A.shape[0]
    N = n + m
    Pc, Ac = sp.coo_matrix(P), sp.coo_matrix(A)
    rows = np.concatenate([Pc.row, np.arange(n), Ac.col, n + np.arange(m)])
    cols = np.concatenate([Pc.col, np.arange(n), n + Ac.row, n + np.arange(m)])
    tag = np.concatenate([np.zeros(P.nnz), np.ones(n), 2 * np.ones(A.nnz), 3 * np.ones(m)]).astype(int)
    src = np.concatenate([np.arange(P.nnz), np.arange(n), np.arange(A.nnz), np.arange(m)])
    # merge duplicates (P diagonal + sigma): build unique (col, row) keys
    key = cols.astype(np.int64) * N + rows
    uniq, inv = np.unique(key, return_inverse=True)
    ucols, urows = uniq // N, uniq % N
    indptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(indptr, ucols + 1, 1)
    indptr = np.cumsum(indptr)
    vals = np.zeros(len(uniq))
    np.add.at(vals, inv[tag == 0], P.data)
    np.add.at(vals, inv[tag == 1], sigma)
    np.add.at(vals, inv[tag == 2], A.data)
    np.add.at(vals, inv[tag == 3], -1.0 / rho_vec)
    K = sp.csc_matrix((vals, urows.astype(np.int32), indptr.astype(np.int32)), shape=(N, N))
    idx = {'P': inv[tag == 0], 'sigma': inv[tag == 1], 'A': inv[tag == 2], 'rho': inv[tag == 3]}
    return K, idx


def permute_upper(K: sp.csc_matrix, perm: np.ndarray) -> Tuple[sp.csc_matrix, np.ndarray]:
    """Symmetric permutation of an upper-triangular matrix: Kp = (P K P')_upper with
    Kp[i, j] = K[perm[i], perm[j]].  Returns Kp and the map Kp.data[k] = K.data[src[k]]."""
    N = K.shape[0]
    pinv = np.empty(N, dtype=np.int64)
    pinv[perm] = np.arange(N)
    Kc = sp.coo_matrix(K)
    r, c = pinv[Kc.row], pinv[Kc.col]
    rr, cc = np.minimum(r, c), np.maximum(r, c)
    order = np.lexsort((rr, cc))
    indptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(indptr, cc + 1, 1)
    indptr = np.cumsum(indptr)
    Kp = sp.csc_matrix((K.data[order], rr[order].astype(np.int32), indptr.astype(np.int32)),
                       shape=(N, N))
    return Kp, order


def numeric_ldl(Kp: sp.csc_matrix, Lp: np.ndarray, Li: np.ndarray
                ) -> Tuple[np.ndarray, np.ndarray]:
    """Numeric LDL' (no pivoting) of the permuted upper-triangular Kp on the fixed pattern
    (Lp, Li).  Row-by-row sparse up-looking factorisation (the scheme QDLDL_factor uses), here
    on a dense work row; returns (Lx, D)."""
    N = Kp.shape[0]
    Lx = np.zeros(len(Li))
    D = np.zeros(N)
    # CSR view of L (row k -> (col j, position in Lx)) built from the CSC pattern
    rows = Li
    cols = np.repeat(np.arange(N), np.diff(Lp))
    order = np.lexsort((cols, rows))
    rptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(rptr, rows + 1, 1)
    rptr = np.cumsum(rptr)
    rcol = cols[order]
    rpos = order
    Kpp, Kpi, Kpx = Kp.indptr, Kp.indices, Kp.data
    y = np.zeros(N)
    for k in range(N):
        # scatter column k of Kp (upper) into y
        s, e = Kpp[k], Kpp[k + 1]
        ii = Kpi[s:e]
        y[ii] = Kpx[s:e]
        dk = y[k]
        y[k] = 0.0
        # solve for row k of L: process columns in increasing order
        for t in range(rptr[k], rptr[k + 1]):
            j = rcol[t]
            yj = y[j]
            y[j] = 0.0
            # y[rows of column j below j and < k] -= L[:, j] * yj
            ps, pe = Lp[j], Lp[j + 1]
            seg = Li[ps:pe]
            cnt = np.searchsorted(seg, k)
            if cnt:
                y[seg[:cnt]] -= Lx[ps:ps + cnt] * yj
            lkj = yj / D[j]
            dk -= yj * lkj
            Lx[rpos[t]] = lkj
        D[k] = dk
    return Lx, D


@dataclass
class OsqpPlan:
    """Everything the batched solver needs that is fixed at code-generation time."""
    n: int
    m: int
    settings: Dict[str, float]
    scaling: Scaling
    # scaled data at theta0
    Px: np.ndarray
    Ax: np.ndarray
    q: np.ndarray
    l: np.ndarray
    u: np.ndarray
    rho_vec: np.ndarray
    constr_type: np.ndarray
    # KKT + factor
    K: sp.csc_matrix
    K_idx: Dict[str, np.ndarray]
    perm: np.ndarray                    # new -> old
    Kp_src: np.ndarray                  # Kp.data = K.data[Kp_src]
    Kp: sp.csc_matrix
    Lp: np.ndarray
    Li: np.ndarray
    Lx: np.ndarray
    D: np.ndarray
    etree: np.ndarray
    # setup(prune=True): positions (in P.data / A.data order) of the entries the KKT pattern was built from --
    # the numerically non-zero ones; None: every stored entry
    keepP: Optional[np.ndarray] = None
    keepA: Optional[np.ndarray] = None

    @property
    def N(self):
        return self.n + self.m

    def pruned(self, P: sp.csc_matrix, A: sp.csc_matrix):
        """(P, A) SCALED, on the pattern the KKT matrix of this plan was built from"""
        Ps = sp.csc_matrix((self.Px, P.indices, P.indptr), shape=P.shape)
        As = sp.csc_matrix((self.Ax, A.indices, A.indptr), shape=A.shape)
        if self.keepP is not None:
            Ps, As = _keep_entries(Ps, self.keepP), _keep_entries(As, self.keepA)
        return Ps, As


def _keep_entries(M: sp.csc_matrix, keep: np.ndarray) -> sp.csc_matrix:
    """M restricted to the stored entries `keep` (positions in M.data), same column order"""
    cols = np.repeat(np.arange(M.shape[1]), np.diff(M.indptr))[keep]
    indptr = np.zeros(M.shape[1] + 1, dtype=np.int64)
    np.add.at(indptr, cols + 1, 1)
    return sp.csc_matrix((M.data[keep], M.indices[keep], np.cumsum(indptr).astype(np.int32)), shape=M.shape)


def choose_ordering(K: sp.csc_matrix, method: str = 'auto') -> np.ndarray:
    if method == 'mindeg':
        return _ord.min_degree(K)
    if method == 'nd':
        return _ord.nested_dissection(K)
    if method == 'natural':
        return np.arange(K.shape[0])
    # auto: keep the shallower elimination tree unless it costs > 1.6x the fill
    cands = {}
    for name, fn in (('mindeg', _ord.min_degree), ('nd', _ord.nested_dissection)):
        p = fn(K)
        Kp, _ = permute_upper(K, p)
        et, lnz = _ord.etree_and_counts(Kp)
        cands[name] = (p, int(lnz.sum()), _ord.etree_height(et))
    pm, fm, hm = cands['mindeg']
    pn, fn_, hn = cands['nd']
    if hn < hm and fn_ <= 1.6 * fm:
        return pn
    return pm


def setup(P: sp.csc_matrix, q: np.ndarray, A: sp.csc_matrix, l: np.ndarray, u: np.ndarray,
          settings: Optional[Dict[str, float]] = None, ordering: str = 'auto', prune: bool = False) -> OsqpPlan:
    """prune: build the KKT pattern, the ordering and the factor from the numerically NON-ZERO entries of P and A
    only.  A parametrised matrix carries every position a parameter can reach (MPC 12/4/10: 4 168 stored entries of
    A, 896 of them non-zero at the code-generation-time values); as long as no varying parameter enters P or A the
    others are exact zeros in every instance, and a factorisation that skips them computes the same numbers from
    a fraction of the fill (nnz(L) 6 314 -> 1 110, elimination-tree height 242 -> 35 on that family).  Px / Ax
    keep the full stored order; keepP / keepA name the entries that were used."""
    stg = dict(DEFAULT_SETTINGS)
    if settings:
        stg.update(settings)
    n, m = P.shape[0], A.shape[0]
    l = np.maximum(np.asarray(l, dtype=np.float64), -OSQP_INFTY)
    u = np.minimum(np.asarray(u, dtype=np.float64), OSQP_INFTY)
    if stg['scaling']:
        Px, qs, Ax, sc = ruiz_scale(P, q, A, int(stg['scaling']))
    else:
        Px, qs, Ax, sc = P.data.copy(), q.copy(), A.data.copy(), Scaling(np.ones(n), np.ones(m), 1.0)
    ls, us = sc.E * l, sc.E * u
    rho_vec, ctype = compute_rho_vec(ls, us, stg['rho'])
    Ps = sp.csc_matrix((Px, P.indices, P.indptr), shape=P.shape)
    As = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
    keepP = keepA = None
    if prune:
        keepP, keepA = np.nonzero(Px != 0.0)[0], np.nonzero(Ax != 0.0)[0]
        Ps, As = _keep_entries(Ps, keepP), _keep_entries(As, keepA)
    K, K_idx = kkt_upper(Ps, As, stg['sigma'], rho_vec)
    perm = choose_ordering(K, ordering)
    Kp, Kp_src = permute_upper(K, perm)
    Lp, Li, etree = _ord.symbolic_ldl(Kp)
    Lx, D = numeric_ldl(Kp, Lp, Li)
    return OsqpPlan(n=n, m=m, settings=stg, scaling=sc, Px=Px, Ax=Ax, q=qs, l=ls, u=us,
                    rho_vec=rho_vec, constr_type=ctype, K=K, K_idx=K_idx, perm=perm,
                    Kp_src=Kp_src, Kp=Kp, Lp=Lp, Li=Li, Lx=Lx, D=D, etree=etree, keepP=keepP, keepA=keepA)


def ldl_solve(plan: OsqpPlan, b: np.ndarray) -> np.ndarray:
    """x = K^{-1} b through the permuted factor (reference host implementation for tests)."""
    N = plan.N
    x = b[plan.perm].astype(np.float64).copy()
    Lp, Li, Lx = plan.Lp, plan.Li, plan.Lx
    for j in range(N):
        s, e = Lp[j], Lp[j + 1]
        if e > s:
            x[Li[s:e]] -= Lx[s:e] * x[j]
    x /= plan.D
    for j in range(N - 1, -1, -1):
        s, e = Lp[j], Lp[j + 1]
        if e > s:
            x[j] -= np.dot(Lx[s:e], x[Li[s:e]])
    out = np.empty(N)
    out[plan.perm] = x
    return out
