"""
Shared (per-family) tables for the per-instance refactorisation path: what the reference does in
`osqp_update_data_mat` when a parameter enters P or A (`cvxpygen/solvers/osqp.py:20-33`; third-party
OSQP: unscale -> overwrite values -> Ruiz-equilibrate again from scratch -> numeric LDL' on the
fixed symbolic pattern), done here for every instance of a batch on the GPU.

Everything structural is computed once on the host:
  * row / column views of A and P for the equilibration sweeps,
  * where every entry of the permuted KKT matrix comes from (P, A, sigma, -1/rho),
  * a level-scheduled "dot-product" schedule for the numeric LDL': for every entry (i, j) of L and
    every pivot, the list of products L_ik d_k L_jk it needs,
  * the (unmerged) level-scheduled substitution program with, for every coefficient, its source
    in the per-instance factor (-L_ij, 1/d_i or a constant).
The numpy functions below replay exactly what the kernel does and are what the tests check the
tables with.
"""

from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import scipy.sparse as sp

from . import ordering as _ord
from . import osqp_setup as _setup
from . import solve_program as _sp

LANES = 64
SRC_ZERO, SRC_ONE, SRC_NEG_L, SRC_DINV = 0, 1, 2, 3         # value sources of the solve program
K_NONE, K_P, K_A, K_SIGMA, K_RHO = 0, 1, 2, 3, 4            # value sources of KKT entries


@dataclass
class RaggedTable:
    """ragged chunks of 64 lane-tasks: ctab [n_chunks, 4] = (max len, flag, first entry, 0);
    task id and entry count per lane; lanes ordered by non-increasing entry count"""
    ctab: np.ndarray
    task: np.ndarray        # uint32 [n_chunks, 64]  (0xFFFFFFFF: none)
    tlen: np.ndarray        # uint32 [n_chunks, 64]

    @property
    def n_chunks(self):
        return int(self.ctab.shape[0])


NO_TASK = 0xFFFFFFFF


def _pack_tasks(levels: List[List[int]], lens: np.ndarray, split: bool = True, team: int = 1):
    """levels: lists of task ids (one dot product of lens[t] terms each).  Returns (RaggedTable, entry
    order: list of (task, term) or None for a padding entry).

    A chunk holds up to 64 lanes.  When a level has few tasks their dot products are SPLIT over
    g = 2^lg adjacent lanes (lane j of a task takes terms j, j + g, ...; the chunk's lg is ctab[:, 3],
    the first lane of the group owns the task and receives the group sum): the schedule's length is
    set by the longest dot product divided by g instead of the longest dot product.  Entry addressing
    is `base + lane` over the lanes whose ADDRESSING length (low 16 bits of tlen; the same for all
    lanes of a task, so that lengths stay non-increasing across the chunk) exceeds the step; the high
    16 bits of tlen are the lane's REAL number of terms, the rest are padding entries.
    team: wavefronts that share a level (see below)."""
    ctab, task, tlen, order = [], [], [], []
    first = 0
    for li, tasks in enumerate(levels):
        tasks = sorted(tasks, key=lambda t: -int(lens[t]))
        s0 = 0
        while s0 < len(tasks) or (s0 == 0 and not tasks):
            rest = len(tasks) - s0
            g = 1
            if split and 0 < rest <= (LANES * max(1, team)) // 2:
                # (team > 1: the chunks of a level run side by side on the wavefronts of a team -- csrc/cpg_osqp_team.h --, so a level's
                # dot products are split until they fill 64 * team lanes: fewer steps per chunk, the level's latency, for more chunks)
                longest = int(lens[tasks[s0]])
                while 2 * g * rest <= LANES * max(1, team) and 2 * g <= max(1, longest) and 2 * g <= LANES:
                    g *= 2
            sel = tasks[s0:s0 + LANES // g]
            T = np.full(LANES, NO_TASK, dtype=np.uint32)
            AL = np.zeros(LANES, dtype=np.int64)
            RL = np.zeros(LANES, dtype=np.int64)
            owner = [None] * LANES
            for i, t in enumerate(sel):
                n_t = int(lens[t])
                T[i * g] = t
                for j in range(g):
                    AL[i * g + j] = -(-n_t // g)
                    RL[i * g + j] = max(0, -(-(n_t - j) // g))
                    owner[i * g + j] = (t, j)
            assert np.all(np.diff(AL) <= 0) and AL.max(initial=0) < 0x10000
            L = int(AL.max()) if len(sel) else 0
            n_ent = 0
            for s in range(L):
                for ln in range(LANES):
                    if AL[ln] > s:
                        t, j = owner[ln]
                        term = s * g + j
                        order.append((t, term) if term < int(lens[t]) else None)
                        n_ent += 1
            s0 += len(sel)
            last = 1 if s0 >= len(tasks) else 0
            ctab.append([L, last, first, int(np.log2(g))])
            task.append(T)
            tlen.append((AL | (RL << 16)).astype(np.uint32))
            first += n_ent
            if not tasks:
                break
    return (RaggedTable(np.asarray(ctab, dtype=np.int32).reshape(-1, 4),
                        np.asarray(task, dtype=np.uint32).reshape(-1, LANES),
                        np.asarray(tlen, dtype=np.uint32).reshape(-1, LANES)), order)


@dataclass
class RefactorPlan:
    n: int
    m: int
    nnzP: int
    nnzA: int
    nnzL: int
    # views for the equilibration sweeps
    Ap: np.ndarray; Ai: np.ndarray                      # CSC
    Arp: np.ndarray; Aent: np.ndarray; Acol: np.ndarray  # CSR view: entry index into CSC order
    Prp: np.ndarray; Pent: np.ndarray; Pcol: np.ndarray  # full symmetric row view of upper-tri P
    Pp: np.ndarray; Pi: np.ndarray
    # factor pattern + KKT sources for every destination (L entries, then the N pivots)
    Lp: np.ndarray; Li: np.ndarray; Lcol: np.ndarray
    perm: np.ndarray
    ksrc_kind: np.ndarray; ksrc_idx: np.ndarray
    # numeric LDL' schedule
    fac: RaggedTable
    fac_a: np.ndarray; fac_b: np.ndarray; fac_k: np.ndarray
    # substitution program (ragged, unmerged) with value sources
    sol: _sp.RaggedProgram
    sol_kind: np.ndarray; sol_idx: np.ndarray
    stats: Dict[str, float]


# forward_in_place='auto': the forward sweep drops its unit-diagonal entries (8 streamed bytes per row
# and iteration) at the price of a read-modify-write at the end of every forward chunk; that pays when
# a level holds many rows.  Measured: +5 % on the portfolio family (1 442 rows in 24 levels), -8 % on
# MPC 12/4/10 (504 rows in 242 levels).
IN_PLACE_MIN_ROWS_PER_LEVEL = 8
# A step of the streaming executor is a memory request (~16 cycles of the CU's address unit plus
# latency), a reduction stage a handful of VALU instructions: plan with cheaper stages than the
# LDS-resident executor's defaults (MPC 12/4/10: 1 159 -> 671 steps per iteration).
STREAM_STAGE_SCALE = float(os.environ.get('CPG_STREAM_STAGE_SCALE', 0.5))


def build_schedules(N: int, perm: np.ndarray, Lp: np.ndarray, Li: np.ndarray, src: Dict[tuple, tuple],
                    forward_in_place: bool = False, stage_scale: float = 1.0):
    """Everything that only depends on the pattern of the permuted factor: for a symmetric
    quasi-definite matrix whose permuted upper-triangle entries (r <= c) have the value sources
    `src[(r, c)] = (kind, idx)`, returns (Lcol, ksrc_kind, ksrc_idx, fac, fac_a, fac_b, fac_k, sol,
    sol_kind, sol_idx, stats): the KKT source of every destination (L entries, then the N pivots),
    the level-scheduled dot-product schedule of the numeric LDL' and the ragged substitution
    program with value sources.  forward_in_place: the rows of the forward sweep ACCUMULATE into their
    own slot (w[r] += -sum L_rk w[k]) instead of carrying a unit-diagonal entry, and rows without
    off-diagonal entries disappear -- one entry per row less to stream where the coefficients live
    in HBM (executors that understand the chunk flag: execute_ragged, run_program_stream)."""
    Lp, Li = np.asarray(Lp, dtype=np.int64), np.asarray(Li, dtype=np.int64)
    nnzL = len(Li)
    Lcol = np.repeat(np.arange(N), np.diff(Lp)).astype(np.int64)
    # diagonal P entries carry "+ sigma": flag by kind K_P with idx and diag -> handled in kernel:
    ksrc_kind = np.zeros(nnzL + N, dtype=np.int32)
    ksrc_idx = np.zeros(nnzL + N, dtype=np.int32)
    for d in range(nnzL):
        kind, idx = src.get((Lcol[d], Li[d]), (K_NONE, 0))
        ksrc_kind[d], ksrc_idx[d] = kind, idx
    for j in range(N):
        kind, idx = src[(j, j)]
        ksrc_kind[nnzL + j], ksrc_idx[nnzL + j] = kind, idx

    # ---- dot-product schedule: dest (i, j) needs sum_k L_ik d_k L_jk over k in rowpat(i) & rowpat(j)
    Lcsr = sp.csr_matrix((np.arange(nnzL) + 1, (Li, Lcol)), shape=(N, N))
    Lcsr.sort_indices()
    rp_ptr, rp_col, rp_pos = Lcsr.indptr, Lcsr.indices, Lcsr.data - 1
    lev = np.zeros(N, dtype=np.int64)
    for j in range(N):
        s, e = Lp[j], Lp[j + 1]
        if e > s:
            np.maximum.at(lev, Li[s:e], lev[j] + 1)
    nlev = int(lev.max()) + 1
    trip_a: List[np.ndarray] = []
    trip_b: List[np.ndarray] = []
    trip_k: List[np.ndarray] = []
    lens = np.zeros(nnzL + N, dtype=np.int64)
    for d in range(nnzL + N):
        if d < nnzL:
            i, j = Li[d], Lcol[d]
        else:
            i = j = d - nnzL
        ki, pi = rp_col[rp_ptr[i]:rp_ptr[i + 1]], rp_pos[rp_ptr[i]:rp_ptr[i + 1]]
        kj, pj = rp_col[rp_ptr[j]:rp_ptr[j + 1]], rp_pos[rp_ptr[j]:rp_ptr[j + 1]]
        common, ia, ib = np.intersect1d(ki, kj, assume_unique=True, return_indices=True)
        trip_a.append(pi[ia]); trip_b.append(pj[ib]); trip_k.append(common)
        lens[d] = len(common)
    # levels: pivots and entries of column j are computed in the level of column j; the division by
    # the pivot happens in a second sweep of the same level, so pivots come first
    lv_piv = [[] for _ in range(nlev)]
    lv_ent = [[] for _ in range(nlev)]
    for j in range(N):
        lv_piv[lev[j]].append(nnzL + j)
    for d in range(nnzL):
        lv_ent[lev[Lcol[d]]].append(d)
    levels = []
    for a in range(nlev):
        levels.append(lv_piv[a] + lv_ent[a])
    fac, order = _pack_tasks(levels, lens)
    fac_a = np.array([trip_a[o[0]][o[1]] if o else 0 for o in order], dtype=np.uint32)
    fac_b = np.array([trip_b[o[0]][o[1]] if o else 0 for o in order], dtype=np.uint32)
    fac_k = np.array([trip_k[o[0]][o[1]] if o else 0 for o in order], dtype=np.uint32)

    # ---- substitution program with value sources (codes instead of values)
    def code(kind, idx):
        return float(kind * (1 << 32) + idx)
    rows_f, cols_f, vals_f = [], [], []
    for r in range(N):
        c = rp_col[rp_ptr[r]:rp_ptr[r + 1]]
        ps = rp_pos[rp_ptr[r]:rp_ptr[r + 1]]
        cols_f.append(np.concatenate([[r], c]).astype(np.int64))
        vals_f.append(np.array([code(SRC_ONE, 0)] + [code(SRC_NEG_L, p) for p in ps]))
    blev = np.zeros(N, dtype=np.int64)
    for j in range(N - 1, -1, -1):
        s, e = Lp[j], Lp[j + 1]
        if e > s:
            blev[j] = blev[Li[s:e]].max() + 1
    phases = []
    if forward_in_place == 'auto':
        forward_in_place = N >= IN_PLACE_MIN_ROWS_PER_LEVEL * nlev
    for a in range(nlev):
        rr = np.nonzero(lev == a)[0]
        if forward_in_place:
            rr = np.array([r for r in rr if len(cols_f[r]) > 1], dtype=np.int64)
            if len(rr):
                phases.append(_sp.Phase(perm[rr], [perm[cols_f[r][1:]] for r in rr], [vals_f[r][1:] for r in rr],
                                        False, f'F{a}', accumulate=True))
            continue
        phases.append(_sp.Phase(perm[rr], [perm[cols_f[r]] for r in rr], [vals_f[r] for r in rr], False, f'F{a}'))
    for a in range(int(blev.max()) + 1):
        rr = np.nonzero(blev == a)[0]
        cs, vs = [], []
        for r in rr:
            s, e = Lp[r], Lp[r + 1]
            cs.append(perm[np.concatenate([[r], Li[s:e]]).astype(np.int64)])
            vs.append(np.array([code(SRC_DINV, r)] + [code(SRC_NEG_L, p) for p in range(s, e)]))
        phases.append(_sp.Phase(perm[rr], cs, vs, False, f'B{a}'))
    sol = _sp.pack_ragged(phases, N, balanced='auto', stage_scale=stage_scale)
    codes = sol.vals.astype(np.int64)
    sol_kind = (codes >> 32).astype(np.int32)
    sol_idx = (codes & 0xFFFFFFFF).astype(np.int32)
    # entries of the streaming executor's layout (cpg_hip_set_refactor pairs consecutive steps; a lane
    # active in one step of a pair gets a zero entry in the other): what one iteration streams
    cnt = []
    for c in range(sol.n_chunks):
        ln = (sol.desc[c] >> 16) & (0xFFF if sol.ctab[c, 3] & 1 else 0xFFFF)
        cnt += [int((ln > s).sum()) for s in range(int(sol.ctab[c, 0]))]
    cnt = np.asarray(cnt + [0] * (len(cnt) % 2), dtype=np.int64)
    stream_entries = int(2 * np.maximum(cnt[0::2], cnt[1::2]).sum()) if len(cnt) else 0
    stats = dict(nnzL=nnzL, fac_chunks=fac.n_chunks, sol_stream_entries=stream_entries, fac_triples=len(fac_a), fac_steps=int(fac.ctab[:, 0].sum()),
                 sol_chunks=sol.n_chunks, sol_steps=int(sol.ctab[:, 0].sum()), sol_nnz=sol.nnz, levels=nlev)
    return (Lcol, ksrc_kind, ksrc_idx, fac, fac_a, fac_b, fac_k, sol, sol_kind, sol_idx, stats)


# what the planner charges for a reduction stage in the per-instance program of SHARED-MATRIX mode when it runs on the
# generated instance executor (register-resident coefficients): that executor is a chain of latencies, a reduction stage
# costs it ~3 dependent DPP steps and a further multiply-add step almost nothing, so rows get one lane each where they
# are short (MPC 12/4/10: 165 steps / 56 stages instead of 86 / 135; 13.35 instead of 14.03 ms per 96 487 instances)
INSTANCE_STAGE_SCALE = float(os.environ.get('CPG_INSTANCE_STAGE_SCALE', 0.7))


def shared_mode_plan(Ps: sp.csc_matrix, As: sp.csc_matrix, osqp: _setup.OsqpPlan) -> 'RefactorPlan':
    """The refactorisation plan of shared-matrix mode (rho adaptation hand-over, rows that changed class).  Planned for
    the generated instance executor (INSTANCE_STAGE_SCALE) when the resulting program fits it -- coefficient registers,
    LDS tables: codegen.instance_program_fits --, else as every streaming plan.  Used by BOTH the code generator
    (codegen.instance_header) and the runtime (BatchSolver._ensure_refactor_handle): the library checks the program's
    fingerprint."""
    from . import codegen as _cg
    if INSTANCE_STAGE_SCALE != STREAM_STAGE_SCALE:
        cand = build_refactor_plan(Ps, As, osqp, stage_scale=INSTANCE_STAGE_SCALE)
        if _cg.instance_program_fits(cand):
            return cand
    return build_refactor_plan(Ps, As, osqp)


def build_refactor_plan(P: sp.csc_matrix, A: sp.csc_matrix, osqp: _setup.OsqpPlan, stage_scale: Optional[float] = None) -> RefactorPlan:
    n, m = P.shape[0], A.shape[0]
    N = n + m
    P, A = sp.csc_matrix(P), sp.csc_matrix(A)
    nnzP, nnzA = P.nnz, A.nnz
    # ---- A row view
    Ac = sp.coo_matrix((np.arange(nnzA) + 1, (A.indices, np.repeat(np.arange(n), np.diff(A.indptr)))),
                       shape=(m, n)).tocsr()
    Ac.sort_indices()
    Arp, Aent, Acol = Ac.indptr.astype(np.int32), (Ac.data - 1).astype(np.int32), Ac.indices.astype(np.int32)
    # ---- P symmetric row view (entry k of the upper triangle appears in row i and, if i != j, row j)
    pr = P.indices
    pc = np.repeat(np.arange(n), np.diff(P.indptr))
    rows = np.concatenate([pr, pc[pr != pc]])
    cols = np.concatenate([pc, pr[pr != pc]])
    ent = np.concatenate([np.arange(nnzP), np.arange(nnzP)[pr != pc]])
    o = np.lexsort((cols, rows))
    Prp = np.zeros(n + 1, dtype=np.int32)
    np.add.at(Prp, rows + 1, 1)
    Prp = np.cumsum(Prp).astype(np.int32)
    Pent, Pcol = ent[o].astype(np.int32), cols[o].astype(np.int32)

    # ---- factor pattern (same permutation / symbolic analysis as the shared-factor plan)
    perm, Lp, Li = osqp.perm, osqp.Lp.astype(np.int64), osqp.Li.astype(np.int64)
    nnzL = len(Li)
    Lcol = np.repeat(np.arange(N), np.diff(Lp)).astype(np.int64)
    pinv = np.empty(N, dtype=np.int64); pinv[perm] = np.arange(N)
    # KKT sources keyed by permuted (row <= col)
    src: Dict[tuple, tuple] = {}

    def put(r, c, kind, idx):
        r, c = (pinv[r], pinv[c])
        key = (min(r, c), max(r, c))
        if key in src:                       # P diagonal entry + sigma
            k0, i0 = src[key]
            assert {k0, kind} == {K_P, K_SIGMA}
            src[key] = (K_P, i0 if k0 == K_P else idx)
        else:
            src[key] = (kind, idx)
    for k in range(nnzP):
        put(pr[k], pc[k], K_P, k)
    for j in range(n):
        put(j, j, K_SIGMA, j)
    Ar = A.indices
    Acn = np.repeat(np.arange(n), np.diff(A.indptr))
    for k in range(nnzA):
        put(n + Ar[k], Acn[k], K_A, k)
    for i in range(m):
        put(n + i, n + i, K_RHO, i)
    (Lcol, ksrc_kind, ksrc_idx, fac, fac_a, fac_b, fac_k, sol, sol_kind, sol_idx, stats) = \
        build_schedules(N, perm, Lp, Li, src, forward_in_place='auto',
                        stage_scale=STREAM_STAGE_SCALE if stage_scale is None else stage_scale)
    return RefactorPlan(n=n, m=m, nnzP=nnzP, nnzA=nnzA, nnzL=nnzL, Ap=A.indptr.astype(np.int32),
                        Ai=A.indices.astype(np.int32), Arp=Arp, Aent=Aent, Acol=Acol, Prp=Prp, Pent=Pent,
                        Pcol=Pcol, Pp=P.indptr.astype(np.int32), Pi=P.indices.astype(np.int32),
                        Lp=Lp.astype(np.int32), Li=Li.astype(np.int32), Lcol=Lcol.astype(np.int32),
                        perm=perm.astype(np.int32), ksrc_kind=ksrc_kind, ksrc_idx=ksrc_idx, fac=fac,
                        fac_a=fac_a, fac_b=fac_b, fac_k=fac_k, sol=sol, sol_kind=sol_kind, sol_idx=sol_idx,
                        stats=stats)


# ------------------------------------------------------------------------------------------------
# numpy replay of the kernel's algorithms (tests)

def replay_equilibrate(rp: RefactorPlan, Px, q, Ax, iters=10):
    """cumulative-scaling form of Ruiz equilibration: D, E, c such that the scaled data are
    c D P D, E A D, c D q (what the kernel computes; equals the in-place sweeps up to rounding)"""
    n, m = rp.n, rp.m
    lim = lambda v: np.where(v < 1e-4, 1.0, np.where(v > 1e4, 1e4, v))
    D, E, c = np.ones(n), np.ones(m), 1.0
    pr = rp.Pi; pc = np.repeat(np.arange(n), np.diff(rp.Pp))
    ar = rp.Ai; ac = np.repeat(np.arange(n), np.diff(rp.Ap))
    for _ in range(iters):
        ps = np.abs(c * D[pr] * Px * D[pc]); as_ = np.abs(E[ar] * Ax * D[ac])
        dn = np.zeros(n); en = np.zeros(m)
        np.maximum.at(dn, pc, ps); np.maximum.at(dn, pr, ps); np.maximum.at(dn, ac, as_)
        np.maximum.at(en, ar, as_)
        D = D / np.sqrt(lim(dn)); E = E / np.sqrt(lim(en))
        ps = np.abs(c * D[pr] * Px * D[pc])
        pn = np.zeros(n); np.maximum.at(pn, pc, ps); np.maximum.at(pn, pr, ps)
        qn = lim(np.array([np.abs(c * D * q).max() if n else 0.0]))[0]
        c = c / lim(np.array([max(pn.mean(), qn)]))[0]
    return D, E, c


def replay_factor(rp: RefactorPlan, Ps, As, sigma, rho_inv):
    """numeric LDL' through the dot-product schedule; returns (Lx, Dg)"""
    nnzL, N = rp.nnzL, rp.n + rp.m

    def kval(d):
        kind, idx = rp.ksrc_kind[d], rp.ksrc_idx[d]
        if kind == K_P:
            v = Ps[idx]
            if d >= nnzL:
                v = v + sigma
            return v
        if kind == K_A:
            return As[idx]
        if kind == K_SIGMA:
            return sigma
        if kind == K_RHO:
            return -rho_inv[idx]
        return 0.0
    Lx, Dg = np.zeros(nnzL), np.zeros(N)
    pending = []
    for c in range(rp.fac.n_chunks):
        L, last, first, nd = rp.fac.ctab[c]
        T, Ln = rp.fac.task[c], rp.fac.tlen[c]
        acc = np.zeros(LANES)
        base = first
        for s in range(L):
            act = Ln > s
            e = base + np.arange(LANES)[act]
            acc[act] += Lx[rp.fac_a[e]] * Dg[rp.fac_k[e]] * Lx[rp.fac_b[e]]
            base += int(act.sum())
        for t in range(nd):
            d = int(T[t])
            v = kval(d) - acc[t]
            if d >= nnzL:
                Dg[d - nnzL] = v
            else:
                Lx[d] = v                     # raw; divided by the pivot once the level is complete
                pending.append(d)
        if last:
            for d in pending:
                Lx[d] /= Dg[rp.Lcol[d]]
            pending = []
    return Lx, Dg


def replay_solve_vals(rp: RefactorPlan, Lx, Dg):
    v = np.zeros(rp.sol.nnz)
    k, i = rp.sol_kind, rp.sol_idx
    v[k == SRC_ONE] = 1.0
    v[k == SRC_NEG_L] = -Lx[i[k == SRC_NEG_L]]
    v[k == SRC_DINV] = 1.0 / Dg[i[k == SRC_DINV]]
    return v
