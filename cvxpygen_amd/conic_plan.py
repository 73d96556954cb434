"""
Shared (per-family) tables for the conic interior-point path (the
reference builds a new Clarabel solver per solve: `cvxpygen/solvers/clarabel.py:172-204`).

Per instance the kernel (csrc/cpg_clarabel_kernel.h) canonicalises P, A, q, b, equilibrates, and
in every interior-point iteration factors

    K = [[P + eps I, A'], [A, -(W'W) - eps I]]

whose pattern is fixed for the family: P (upper), A, the diagonal of the (2,2) block and one dense
block per second-order cone, per PSD cone (d x d over its svec rows, entries computed from the cone's
NT point Q) and per exponential / power cone (3 x 3: the scaling block H_s of a nonsymmetric cone
stands where W'W stands for a symmetric one).  Everything structural is computed once here: fill-reducing
permutation, symbolic LDL', where every KKT entry comes from, the level-scheduled dot-product
schedule of the numeric factorisation and the ragged substitution program with value sources
(shared machinery: refactor_plan.build_schedules).
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import scipy.sparse as sp

from . import ordering as _ord
from . import osqp_setup as _setup
from . import solve_program as _sp
from .refactor_plan import RaggedTable, build_schedules

# KKT value sources (device: cpg_clarabel_kernel.h)
K_NONE, K_P, K_A, K_DIAGX, K_HDIAG, K_HSOC, K_HNS, K_HPSD = 0, 1, 2, 3, 5, 6, 7, 8
PSD_MAX = 8             # largest PSD cone order of the kernel (csrc/cpg_clarabel_psd.h)
# what the planner of the substitution program charges for a reduction stage, relative to the defaults
# tuned on the large OSQP programs: small KKT systems (ADP: 36 rows) want wider rows and fewer steps
# (29 -> 18 steps per solve, +5 % instances/s)
CONIC_STAGE_SCALE = float(os.environ.get('CPG_CONIC_STAGE_SCALE', 0.5))


@dataclass
class ConicPlan:
    n: int
    m: int
    nnzP: int
    nnzA: int
    nnzL: int
    n_zero: int
    n_nonneg: int
    soc_dims: np.ndarray
    n_exp: int
    pow_alpha: np.ndarray            # exponents of the 3-d power cones (rows: zero | nonneg | soc | psd | exp | pow)
    psd_dims: np.ndarray             # matrix orders of the PSD cones
    Ap: np.ndarray; Ai: np.ndarray
    Arp: np.ndarray; Aent: np.ndarray; Acol: np.ndarray
    Pp: np.ndarray; Pi: np.ndarray
    Prp: np.ndarray; Pent: np.ndarray; Pcol: np.ndarray
    perm: np.ndarray
    Lp: np.ndarray; Li: np.ndarray; Lcol: np.ndarray
    ksrc_kind: np.ndarray; ksrc_idx: np.ndarray
    fac: RaggedTable
    fac_a: np.ndarray; fac_b: np.ndarray; fac_k: np.ndarray
    sol: _sp.RaggedProgram
    sol_kind: np.ndarray; sol_idx: np.ndarray
    stats: Dict[str, float]


def matrix_views(P: sp.csc_matrix, A: sp.csc_matrix):
    """CSR view of A (entry numbers of the CSC storage) and full symmetric row view of upper P"""
    n, m = P.shape[0], A.shape[0]
    nnzP, nnzA = P.nnz, A.nnz
    Ac = sp.coo_matrix((np.arange(nnzA) + 1, (A.indices, np.repeat(np.arange(n), np.diff(A.indptr)))),
                       shape=(m, n)).tocsr()
    Ac.sort_indices()
    Arp, Aent, Acol = Ac.indptr.astype(np.int32), (Ac.data - 1).astype(np.int32), Ac.indices.astype(np.int32)
    pr = P.indices
    pc = np.repeat(np.arange(n), np.diff(P.indptr))
    rows = np.concatenate([pr, pc[pr != pc]])
    cols = np.concatenate([pc, pr[pr != pc]])
    ent = np.concatenate([np.arange(nnzP), np.arange(nnzP)[pr != pc]])
    o = np.lexsort((cols, rows))
    Prp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(Prp, rows + 1, 1)
    Prp = np.cumsum(Prp).astype(np.int32)
    return Arp, Aent, Acol, Prp, ent[o].astype(np.int32), cols[o].astype(np.int32)


def build_conic_plan(desc, ordering: str = 'auto') -> ConicPlan:
    if not desc.cones:
        raise ValueError('not a conic family')
    for key, val in desc.cones.items():
        if key not in ('zero', 'nonneg', 'soc', 'exp', 'pow', 'psd') and np.size(val) and np.any(val):
            raise NotImplementedError(f'cone type "{key}" is not supported by the interior-point kernel')
    P, A = sp.csc_matrix(desc.P), sp.csc_matrix(desc.A)
    n, m = desc.n_var, desc.m
    N = n + m
    nz, nn = int(desc.cones['zero']), int(desc.cones['nonneg'])
    soc = np.asarray(desc.cones.get('soc', []), dtype=np.int32)
    n_exp = int(desc.cones.get('exp', 0) or 0)
    pow_alpha = np.asarray(desc.cones.get('pow', []), dtype=np.float64).ravel()
    if len(pow_alpha) and not ((pow_alpha > 0.0) & (pow_alpha < 1.0)).all():
        raise ValueError('power cone exponents must lie in (0, 1)')
    psd = np.asarray(desc.cones.get('psd', []), dtype=np.int32).ravel()
    if len(psd) and (psd.min() < 1 or psd.max() > PSD_MAX):
        raise NotImplementedError(f'PSD cones of order 1 .. {PSD_MAX} only (the kernel factors them one cone per lane)')
    if nz + nn + int(soc.sum()) + int((psd * (psd + 1) // 2).sum()) + 3 * (n_exp + len(pow_alpha)) != m:
        raise ValueError('cone dimensions do not add up to the number of rows')
    if N >= 0xFFFF:
        raise ValueError('family too large for 16-bit slot indices')
    Arp, Aent, Acol, Prp, Pent, Pcol = matrix_views(P, A)

    # ---- KKT pattern (upper triangle, natural order) with value sources
    src_nat: Dict[tuple, tuple] = {}
    pr, pc = P.indices, np.repeat(np.arange(n), np.diff(P.indptr))
    for k in range(P.nnz):
        src_nat[(int(pr[k]), int(pc[k]))] = (K_P, k)
    for j in range(n):
        src_nat.setdefault((j, j), (K_DIAGX, j))
    ar, ac = A.indices, np.repeat(np.arange(n), np.diff(A.indptr))
    for k in range(A.nnz):
        src_nat[(int(ac[k]), n + int(ar[k]))] = (K_A, k)
    for i in range(m):
        src_nat[(n + i, n + i)] = (K_HDIAG, i)
    o = nz + nn
    for d in soc:
        for a in range(d):
            for b in range(a + 1, d):
                src_nat[(n + o + a, n + o + b)] = (K_HSOC, (o + a) | ((o + b) << 16))
        o += int(d)
    qoff = 0
    for pp in (int(v) for v in psd):                 # PSD cones: dense block of W'W = Q (x)s Q over the svec rows; Q sits at `qoff` of the slice's PSD store
        ij = [(i, j) for j in range(pp) for i in range(j + 1)]
        for a, (i, j) in enumerate(ij):
            for b in range(a + 1, len(ij)):
                k, l = ij[b]
                src_nat[(n + o + a, n + o + b)] = (K_HPSD, qoff | (pp << 12) | (i << 16) | (j << 19) | (k << 22) | (l << 25))
        o += len(ij)
        qoff += 11 * pp * pp + 3 * pp          # (Q | R | R^-1 | lambda | workspace: csrc/cpg_clarabel_psd.h CPG_PSD_WORK)
    if qoff > 0xFFF:
        raise NotImplementedError('PSD cones too large for the 12-bit offsets of their KKT sources')
    for _ in range(n_exp + len(pow_alpha)):          # 3 x 3 scaling blocks: off-diagonals (0,1), (0,2), (1,2) at wv[o + 0 .. 2]
        src_nat[(n + o, n + o + 1)] = (K_HNS, o)
        src_nat[(n + o, n + o + 2)] = (K_HNS, o + 1)
        src_nat[(n + o + 1, n + o + 2)] = (K_HNS, o + 2)
        o += 3
    keys = list(src_nat.keys())
    K = sp.csc_matrix((np.ones(len(keys)), ([k[0] for k in keys], [k[1] for k in keys])), shape=(N, N))
    perm = _setup.choose_ordering(K, ordering)
    Kp, _ = _setup.permute_upper(K, perm)
    Lp, Li, etree = _ord.symbolic_ldl(Kp)
    pinv = np.empty(N, dtype=np.int64)
    pinv[perm] = np.arange(N)
    src = {}
    for (r, c), v in src_nat.items():
        a, b = int(pinv[r]), int(pinv[c])
        src[(min(a, b), max(a, b))] = v
    (Lcol, ksrc_kind, ksrc_idx, fac, fac_a, fac_b, fac_k, sol, sol_kind, sol_idx, stats) = \
        build_schedules(N, perm, Lp, Li, src, stage_scale=CONIC_STAGE_SCALE)
    stats = dict(stats)
    stats['etree_height'] = int(_ord.etree_height(etree))
    return ConicPlan(n=n, m=m, nnzP=P.nnz, nnzA=A.nnz, nnzL=len(Li), n_zero=nz, n_nonneg=nn, soc_dims=soc, n_exp=n_exp, pow_alpha=pow_alpha, psd_dims=psd,
                     Ap=A.indptr.astype(np.int32), Ai=A.indices.astype(np.int32), Arp=Arp, Aent=Aent,
                     Acol=Acol, Pp=P.indptr.astype(np.int32), Pi=P.indices.astype(np.int32), Prp=Prp,
                     Pent=Pent, Pcol=Pcol, perm=perm.astype(np.int32), Lp=np.asarray(Lp, dtype=np.int32),
                     Li=np.asarray(Li, dtype=np.int32), Lcol=Lcol.astype(np.int32), ksrc_kind=ksrc_kind,
                     ksrc_idx=ksrc_idx, fac=fac, fac_a=fac_a, fac_b=fac_b, fac_k=fac_k, sol=sol,
                     sol_kind=sol_kind, sol_idx=sol_idx, stats=stats)
