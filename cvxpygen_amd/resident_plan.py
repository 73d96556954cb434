"""
Plan of the RESIDENT per-instance factor kernel (csrc/cpg_osqp_resident.h): families whose parameters enter P or A
(`osqp_update_data_mat` per instance, cvxpygen/solvers/osqp.py:20-33) solved with the instance's factor kept ON
THE CU -- numeric LDL' in the wavefront's LDS slice, substitution coefficients in registers -- instead of
streaming 70-150 KB of coefficients per instance and ADMM iteration from HBM (refactor_plan.py).

What this module adds to a RefactorPlan:

* **merged levels**: plain level scheduling leaves one dependent phase per level of the elimination tree and sweep
  (portfolio family: 24 levels of which 19 hold ONE row -- the dense trailing block of the factor).  Consecutive
  levels are merged into groups G whose unit-triangular diagonal block is inverted NUMERICALLY, per instance and
  factorisation, on the device (X_G = L_GG^-1; same pattern algebra as solve_program._compile_lower, which does
  it on the host for the family's shared factor).  A merged group costs two phases per sweep,
      forward   t_G = b_G - L_GE y_E ,   y_G = X_G t_G          (E: earlier groups)
      backward  t_G = D_G^-1 y_G - L_LG' x_L ,   x_G = X_G' t_G  (L: later groups)
  instead of one per level: portfolio 47 -> 13 dependent phases per KKT solve.
* the **inverse schedule**: dot products X_ij = -(l_ij + sum_k l_ik X_kj) in the format of the numeric LDL'
  (refactor_plan._pack_tasks), appended to the factorisation schedule; all indices are ABSOLUTE positions in one
  array  fac = [ M (nnzL) | 1/d (N) | X (nnzX) | 1.0 | 0.0 ]  (M_ij = l_ij d_j, the undivided entries of numeric_ldl_m).
* **row programs** of the termination test's three products (A x, P x, A' y) in the ragged layout, with the
  entry of the instance's scaled matrix behind every coefficient: the kernel keeps per-instance copies of A and P in
  program order and streams them coalesced (cpg_osqp_kernel.h run_program_stream).
* **coalesced canonicalisation maps** (ELL): theta is staged in LDS once, entry k of P / A / q / u is
  base[k] + sum_j coef[j][k] theta[idx[j][k]].
* the **entry table** of the equilibration sweeps (row | column of every stored entry of A and P).

The numpy functions at the end replay what the kernel computes; tests/test_resident.py checks them against dense
linear algebra and against refactor_plan's replay of the unmerged path.
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import scipy.sparse as sp

from . import refactor_plan as _rp
from . import solve_program as _sp

LANES = 64
SRC_ZERO, SRC_ONE, SRC_NEG_L, SRC_DINV, SRC_X = 0, 1, 2, 3, 4
MAX_GROUP_ROWS = int(os.environ.get('CPG_RESIDENT_MAX_GROUP_ROWS', 64))
# what the planner charges for a reduction stage relative to a multiply-add step (refactor_plan.INSTANCE_STAGE_SCALE:
# the register-resident executor is a chain of latencies)
RESIDENT_STAGE_SCALE = float(os.environ.get('CPG_RESIDENT_STAGE_SCALE', 0.7))
PIVOT_FLAG = 0x80000000
NO_TASK = 0xFFFFFFFF


def _levels(N, Lp, Li):
    lev = np.zeros(N, dtype=np.int64)
    for j in range(N):
        s, e = Lp[j], Lp[j + 1]
        if e > s:
            np.maximum.at(lev, Li[s:e], lev[j] + 1)
    return lev


def _closure(rows: np.ndarray, rowpat: List[np.ndarray]) -> np.ndarray:
    """Boolean pattern of (I - L_GG)^-1 for the rows G (ascending elimination order): R[a, b] = row a reaches b."""
    ng = len(rows)
    pos = {int(r): a for a, r in enumerate(rows)}
    R = np.eye(ng, dtype=bool)
    for a, r in enumerate(rows):
        for k in rowpat[int(r)]:
            b = pos.get(int(k))
            if b is not None:
                R[a] |= R[b]
    return R


def plan_groups(N: int, Lp: np.ndarray, Li: np.ndarray, max_rows: int = None) -> List[Tuple[int, int]]:
    """Level ranges (a, b) of the groups, by dynamic programming on the packer's cost model: a single level costs
    one phase per sweep, a merged group two (off-group products, then the inverse of its diagonal block)."""
    max_rows = MAX_GROUP_ROWS if max_rows is None else max_rows
    Lcol = np.repeat(np.arange(N), np.diff(Lp)).astype(np.int64)
    Lcsr = sp.csr_matrix((np.ones(len(Li)), (Li, Lcol)), shape=(N, N))
    Lcsc = sp.csc_matrix(Lcsr)
    rowpat = [Lcsr.indices[Lcsr.indptr[i]:Lcsr.indptr[i + 1]] for i in range(N)]
    colpat = [Lcsc.indices[Lcsc.indptr[j]:Lcsc.indptr[j + 1]] for j in range(N)]
    lev = _levels(N, Lp, Li)
    nlev = int(lev.max()) + 1 if N else 0
    rows_of = [np.nonzero(lev == a)[0] for a in range(nlev)]

    def cost_single(a):
        rr = rows_of[a]
        f = np.array([len(rowpat[r]) for r in rr], dtype=np.int64)
        b = np.array([len(colpat[r]) + 1 for r in rr], dtype=np.int64)
        c = _sp._approx_cost(b, False)
        if f.sum():
            c += _sp._approx_cost(f[f > 0], False)
        return c

    def cost_group(a, b):
        G = np.concatenate([rows_of[v] for v in range(a, b + 1)])
        G.sort()
        inG = np.zeros(N, dtype=bool); inG[G] = True
        R = _closure(G, rowpat)
        xin = R.sum(axis=1) - 1                       # X entries per row (forward) ...
        xout = R.sum(axis=0) - 1                      # ... and per column (backward)
        fo = np.array([int((~inG[rowpat[r]]).sum()) for r in G], dtype=np.int64)
        bo = np.array([int((~inG[colpat[r]]).sum()) + 1 for r in G], dtype=np.int64)
        c = _sp._approx_cost(bo, False)
        for v in (fo, xin, xout):
            if v.sum():
                c += _sp._approx_cost(v[v > 0], False)
        return c
    best = np.full(nlev + 1, np.inf); best[0] = 0.0
    choice = np.zeros(nlev + 1, dtype=np.int64)
    for a in range(nlev):
        nrows = 0
        for b in range(a, nlev):
            nrows += len(rows_of[b])
            if b > a and nrows > max_rows:
                break
            c = cost_single(a) if b == a else cost_group(a, b)
            if best[a] + c < best[b + 1]:
                best[b + 1] = best[a] + c
                choice[b + 1] = a
    out = []
    b = nlev
    while b > 0:
        a = int(choice[b])
        out.append((a, b - 1))
        b = a
    return out[::-1]


@dataclass
class ResidentPlan:
    base: _rp.RefactorPlan
    groups: List[Tuple[int, int]]
    nnzX: int
    x_row: np.ndarray; x_col: np.ndarray          # elimination indices of the X entries
    # combined factorisation + inverse schedule (absolute indices into fac = [M | 1/d | X | 1.0])
    f_ctab: np.ndarray; f_task: np.ndarray; f_len: np.ndarray
    f_a: np.ndarray; f_b: np.ndarray; f_k: np.ndarray
    k_kind: np.ndarray; k_idx: np.ndarray         # KKT source of fac[0 .. nnzL + N)
    # merged substitution program (value sources instead of values)
    sol: _sp.RaggedProgram
    sol_kind: np.ndarray; sol_idx: np.ndarray
    sol_lcol: np.ndarray                          # column of the L entry behind a kind-2 coefficient
    # products of the termination test: ragged programs on the work vector [x | y | A x | P x | A' y]
    rows_A: _sp.RaggedProgram; rows_P: _sp.RaggedProgram; rows_At: _sp.RaggedProgram
    rows_A_ent: np.ndarray; rows_P_ent: np.ndarray; rows_At_ent: np.ndarray
    out_ax: int; out_px: int; out_aty: int; w_slots: int
    stats: Dict[str, float] = field(default_factory=dict)
    team: int = 1                                 # wavefronts per instance the programs were planned for (1: the resident kernel)

    @property
    def fac_len(self) -> int:                      # [M | 1/d | X | 1.0 | 0.0]
        return self.base.nnzL + self.base.n + self.base.m + self.nnzX + 2

    @property
    def one_pos(self) -> int:
        return self.fac_len - 2


def build_resident_plan(P: sp.csc_matrix, A: sp.csc_matrix, osqp, stage_scale: Optional[float] = None,
                        groups: Optional[List[Tuple[int, int]]] = None, team: int = 1, inplace_x: Optional[bool] = None) -> ResidentPlan:
    """team = W > 1: the substitution program and the row programs planned for a team of W wavefronts per instance
    (csrc/cpg_osqp_team.h): everything else -- groups, schedules, sources -- is the plan of the resident kernel.
    inplace_x (default: team <= 1): the products with the inverses of merged diagonal blocks as in-place (`deferred`) phases;
    False: to slots of their own (what the team kernel's executor runs, also with a team of one)."""
    return _build(P, A, osqp, stage_scale, groups, int(team), (int(team) <= 1) if inplace_x is None else bool(inplace_x))


def _build(P, A, osqp, stage_scale, groups, team=1, inplace_x=True) -> ResidentPlan:
    # (experiments, CPG_TEAM_GROUP_SECTIONS=1: the block inverses group by group with the marks that let wavefronts 1 .. W - 1 FOLLOW
    # the LDL' chain level by level -- measured slower than inverting all groups side by side behind the chain, 215 against 204 us per
    # factorisation: the chain pays a signal per level and the followers do not keep its pace; profiles/r5_s8_*)
    group_sections = (not inplace_x) and os.environ.get('CPG_TEAM_GROUP_SECTIONS', '0') == '1'
    base = _rp.build_refactor_plan(P, A, osqp)
    n, m, nnzL = base.n, base.m, base.nnzL
    N = n + m
    Lp, Li, Lcol = base.Lp.astype(np.int64), base.Li.astype(np.int64), base.Lcol.astype(np.int64)
    perm = base.perm.astype(np.int64)
    Lcsr = sp.csr_matrix((np.arange(nnzL) + 1, (Li, Lcol)), shape=(N, N)); Lcsr.sort_indices()
    rp_ptr, rp_col, rp_pos = Lcsr.indptr, Lcsr.indices, Lcsr.data - 1
    rowpat = [rp_col[rp_ptr[i]:rp_ptr[i + 1]] for i in range(N)]
    rowpos = [rp_pos[rp_ptr[i]:rp_ptr[i + 1]] for i in range(N)]
    lev = _levels(N, Lp, Li)
    nlev = int(lev.max()) + 1 if N else 0
    groups = plan_groups(N, Lp, Li) if groups is None else list(groups)
    # (groups handed in -- a library's header records its own -- must tile the levels 0 .. nlev - 1 of THIS family's factor:
    # a ValueError, not an assert, so that a caller can tell a foreign header from a bug, python -O or not)
    if not groups or groups[0][0] != 0 or groups[-1][1] != nlev - 1 or any(a > b for a, b in groups) or \
            any(groups[k + 1][0] != groups[k][1] + 1 for k in range(len(groups) - 1)):
        raise ValueError(f'level groups do not tile the {nlev} levels of this family\'s factor')
    grp_of = np.zeros(N, dtype=np.int64)
    g_rows = []
    for gi, (a, b) in enumerate(groups):
        rr = np.nonzero((lev >= a) & (lev <= b))[0]
        grp_of[rr] = gi
        g_rows.append(rr)

    # ---- X = L_GG^-1 of the multi-level groups: pattern, numbering, dot-product terms
    xid: Dict[Tuple[int, int], int] = {}
    x_row, x_col = [], []
    depth = np.zeros(N, dtype=np.int64)
    for gi, (a, b) in enumerate(groups):
        if a == b:
            continue
        G = g_rows[gi]
        R = _closure(G, rowpat)
        pos = {int(r): t for t, r in enumerate(G)}
        for t, r in enumerate(G):
            dd = [depth[k] + 1 for k in rowpat[int(r)] if int(k) in pos]
            depth[r] = max(dd) if dd else 0
            for s in np.nonzero(R[t, :t])[0]:
                xid[(int(r), int(G[s]))] = len(x_row)
                x_row.append(int(r)); x_col.append(int(G[s]))
    nnzX = len(x_row)
    X0 = nnzL + N                                      # position of X[0] in fac
    ONE = X0 + nnzX
    ta, tb, tk, lens = [], [], [], np.zeros(nnzX, dtype=np.int64)
    for x, (i, j) in enumerate(zip(x_row, x_col)):
        a_, b_, k_ = [], [], []
        for k, p in zip(rowpat[i], rowpos[i]):
            k = int(k)
            if grp_of[k] != grp_of[i] or k < j:
                continue
            if k == j:
                a_.append(int(p)); k_.append(nnzL + k); b_.append(ONE)
            elif (k, j) in xid:
                a_.append(int(p)); k_.append(nnzL + k); b_.append(X0 + xid[(k, j)])
        ta.append(a_); tb.append(b_); tk.append(k_); lens[x] = len(a_)
        assert lens[x] >= 1
    nxl = int(depth.max()) + 1 if nnzX else 0
    # ---- combined schedule: the LDL' chunks of the base plan, then the inverse
    fb = base.fac
    f_ctab = [fb.ctab.copy()]
    f_task = [np.where(fb.task == NO_TASK, NO_TASK,
                       np.where(fb.task >= nnzL, fb.task | PIVOT_FLAG, fb.task)).astype(np.uint32)]
    f_len = [fb.tlen.copy()]
    f_a = [base.fac_a.astype(np.int64)]; f_b = [base.fac_b.astype(np.int64)]; f_k = [base.fac_k.astype(np.int64) + nnzL]

    def append_inverse(xlv, n_before, need_levels=False):
        # (splitting a level's dot products over the lanes of the whole team -- _pack_tasks(team=W) -- was tried: more, shorter chunks,
        # but a batch of the table walk costs the same for 2 steps as for 8: 182 instead of 156 batches on the busiest wavefront)
        ft, order = _rp._pack_tasks(xlv, lens)
        ct = ft.ctab.copy(); ct[:, 2] += n_before
        if need_levels:
            # (team kernel) a chunk of the inverse can run as soon as the LDL' chain has passed the LAST row it reads the factor of:
            # number of LDL' levels that must be complete, in the bits above the marks of column 1
            for c_ in range(ct.shape[0]):
                tk_ = ft.task[c_][ft.task[c_] != NO_TASK]
                need = int(max(lev[x_row[int(t_)]] for t_ in tk_)) + 1 if len(tk_) else 0
                ct[c_, 1] |= need << 8
        f_ctab.append(ct)
        f_task.append(np.where(ft.task == NO_TASK, NO_TASK, ft.task + X0).astype(np.uint32))
        f_len.append(ft.tlen)
        f_a.append(np.array([ta[o[0]][o[1]] if o else 0 for o in order], dtype=np.int64))
        f_b.append(np.array([tb[o[0]][o[1]] if o else 0 for o in order], dtype=np.int64))
        f_k.append(np.array([tk[o[0]][o[1]] if o else nnzL for o in order], dtype=np.int64))
        return len(order)
    if nnzX and not group_sections:
        # every group's inverse level by level side by side: the fewest levels for one wavefront
        xlevels = [[] for _ in range(nxl)]
        for x, i in enumerate(x_row):
            xlevels[depth[i]].append(x)
        append_inverse([l for l in xlevels if l], len(base.fac_a))
    elif nnzX:
        # (team kernel) one SECTION per merged group, in group order: a group's inverse needs the factor's columns of that group
        # only -- row by row it can FOLLOW the LDL' chain through the group, on another wavefront (cpg_osqp_team.h).  Marks in
        # column 1 of the chunk table (bit 0 stays "level complete"): 4 on the LDL' chunk that completes a merged group, 2 on
        # the last chunk of a group's section, and from bit 8 up the number of LDL' levels an inverse chunk waits for.
        lvl_end = np.nonzero(f_ctab[0][:, 1] != 0)[0]              # chunk that ends LDL' level k
        n_before = len(base.fac_a)
        for gi, (a, b) in enumerate(groups):
            if a == b:
                continue
            xs = [x for x, i in enumerate(x_row) if grp_of[i] == gi]
            if not xs:
                continue
            f_ctab[0][lvl_end[b], 1] |= 4
            xl = [[] for _ in range(nxl)]
            for x in xs:
                xl[depth[x_row[x]]].append(x)
            n_before += append_inverse([l for l in xl if l], n_before, need_levels=True)
            f_ctab[-1][-1, 1] |= 2
    f_ctab = np.concatenate(f_ctab).astype(np.int32)
    f_task = np.concatenate(f_task).astype(np.uint32); f_len = np.concatenate(f_len).astype(np.uint32)
    f_a = np.concatenate(f_a).astype(np.uint32); f_b = np.concatenate(f_b).astype(np.uint32); f_k = np.concatenate(f_k).astype(np.uint32)

    # ---- merged substitution program
    def code(kind, idx):
        return float(kind * (1 << 32) + idx)
    phases = []
    for gi, (a, b) in enumerate(groups):              # forward
        G = g_rows[gi]
        rr, cs, vs = [], [], []
        for r in G:
            sel = grp_of[rowpat[r]] != gi
            if sel.any():
                rr.append(r); cs.append(perm[rowpat[r][sel]]); vs.append(np.array([code(SRC_NEG_L, p) for p in rowpos[r][sel]]))
        if rr:
            phases.append(_sp.Phase(perm[np.array(rr)], cs, vs, False, f'F{a}-{b}', accumulate=True))
        if a != b:
            rr, cs, vs = [], [], []
            byrow: Dict[int, List[Tuple[int, int]]] = {}
            for x, (i, j) in enumerate(zip(x_row, x_col)):
                if grp_of[i] == gi:
                    byrow.setdefault(i, []).append((j, x))
            if not inplace_x:
                # (a team of wavefronts: the in-place form below would need a barrier between the last gather and the first
                # store of the phase; y_G goes to slots of its own instead -- assign_slots' `intra` phases -- with the unit
                # diagonal of X as an explicit coefficient: one barrier per phase, solution not in place (sol.final_pos))
                for i in G:
                    ent = byrow.get(int(i), [])
                    rr.append(int(i)); cs.append(perm[np.array([int(i)] + [j for j, _ in ent])])
                    vs.append(np.array([code(SRC_ONE, 0)] + [code(SRC_X, x) for _, x in ent]))
                phases.append(_sp.Phase(perm[np.array(rr)], cs, vs, True, f'FX{a}-{b}'))
                continue
            for i in sorted(byrow):
                rr.append(i); cs.append(perm[np.array([j for j, _ in byrow[i]])]); vs.append(np.array([code(SRC_X, x) for _, x in byrow[i]]))
            if rr:
                phases.append(_sp.Phase(perm[np.array(rr)], cs, vs, False, f'FX{a}-{b}', accumulate=True, deferred=True))
    Lcsc_ptr = Lp
    for gi in range(len(groups) - 1, -1, -1):         # backward
        a, b = groups[gi]
        G = g_rows[gi]
        cs, vs = [], []
        for r in G:
            s, e = Lcsc_ptr[r], Lcsc_ptr[r + 1]
            ks = Li[s:e]; ps = np.arange(s, e)
            sel = grp_of[ks] != gi
            cs.append(perm[np.concatenate([[r], ks[sel]]).astype(np.int64)])
            vs.append(np.array([code(SRC_DINV, r)] + [code(SRC_NEG_L, p) for p in ps[sel]]))
        phases.append(_sp.Phase(perm[G], cs, vs, False, f'B{a}-{b}'))
        if a != b:
            bycol: Dict[int, List[Tuple[int, int]]] = {}
            for x, (i, j) in enumerate(zip(x_row, x_col)):
                if grp_of[i] == gi:
                    bycol.setdefault(j, []).append((i, x))
            rr, cs, vs = [], [], []
            if not inplace_x:
                for j in G:
                    ent = bycol.get(int(j), [])
                    rr.append(int(j)); cs.append(perm[np.array([int(j)] + [i for i, _ in ent])])
                    vs.append(np.array([code(SRC_ONE, 0)] + [code(SRC_X, x) for _, x in ent]))
                phases.append(_sp.Phase(perm[np.array(rr)], cs, vs, True, f'BX{a}-{b}'))
                continue
            for j in sorted(bycol):
                rr.append(j); cs.append(perm[np.array([i for i, _ in bycol[j]])]); vs.append(np.array([code(SRC_X, x) for _, x in bycol[j]]))
            if rr:
                phases.append(_sp.Phase(perm[np.array(rr)], cs, vs, False, f'BX{a}-{b}', accumulate=True, deferred=True))
    sol = _sp._pack_ragged(phases, N, balanced='auto', stage_scale=RESIDENT_STAGE_SCALE if stage_scale is None else stage_scale,
                           team=team)
    codes = sol.vals.astype(np.int64)
    sol_kind = (codes >> 32).astype(np.int32); sol_idx = (codes & 0xFFFFFFFF).astype(np.int32)
    sol_lcol = np.where(sol_kind == SRC_NEG_L, Lcol[np.where(sol_kind == SRC_NEG_L, sol_idx, 0)], 0).astype(np.int32)

    # ---- products of the termination test on w = [x (n) | y (m) | .. | A x (m)  or  P x (n) | A' y (n)]
    Acsr = sp.csr_matrix((np.arange(base.nnzA) + 1.0, (base.Ai, np.repeat(np.arange(n), np.diff(base.Ap)))), shape=(m, n))
    Acsr.sort_indices()
    # (the results sit behind the executor's work vector and the instance's q / u in the wavefront's LDS slice; A x is
    # consumed before P x and A' y are formed -- update_info's primal residual, then its dual residual -- and shares their slots)
    out_ax = sol.n_slots + _sp.GEN_EXTRA_SLOTS + N
    out_px, out_aty = out_ax, out_ax + n
    w_slots = out_ax + max(m, 2 * n)

    def product(M: sp.csr_matrix, col_off: int, out_off: int, name: str):
        M = sp.csr_matrix(M); M.sort_indices()
        rows = np.arange(M.shape[0]) + out_off
        cs = [M.indices[M.indptr[r]:M.indptr[r + 1]].astype(np.int64) + col_off for r in range(M.shape[0])]
        vs = [M.data[M.indptr[r]:M.indptr[r + 1]].astype(np.float64) for r in range(M.shape[0])]
        keep = [r for r in range(M.shape[0]) if len(cs[r])]
        ph = _sp.Phase(rows[keep], [cs[r] for r in keep], [vs[r] for r in keep], False, name)
        prog = _sp._pack_ragged([ph], w_slots, balanced='auto', stage_scale=_rp.STREAM_STAGE_SCALE if team <= 1 else RESIDENT_STAGE_SCALE,
                                team=team)
        return prog, (prog.vals.astype(np.int64) - 1).astype(np.int32)      # entry behind every coefficient (-1: padding)
    rows_A, rows_A_ent = product(Acsr, 0, out_ax, 'A')
    pr = base.Pi.astype(np.int64); pc = np.repeat(np.arange(n), np.diff(base.Pp)).astype(np.int64)
    off = pr != pc
    Pfull = sp.csr_matrix((np.concatenate([np.arange(base.nnzP), np.arange(base.nnzP)[off]]) + 1.0,
                           (np.concatenate([pr, pc[off]]), np.concatenate([pc, pr[off]]))), shape=(n, n))
    rows_P, rows_P_ent = product(Pfull, 0, out_px, 'P')
    rows_At, rows_At_ent = product(sp.csr_matrix(Acsr.T), n, out_aty, 'At')

    stats = dict(base.stats)
    stats.update(groups=len(groups), merged=sum(1 for a, b in groups if a != b), nnzX=nnzX, phases=sol.n_phases,
                 sol_steps=int(sol.ctab[:, 0].sum()), sol_chunks=sol.n_chunks, inv_levels=int((np.concatenate(f_ctab)[len(fb.ctab):, 1] != 0).sum()) if isinstance(f_ctab, list) else int((f_ctab[len(fb.ctab):, 1] != 0).sum()),
                 fac_chunks=int(f_ctab.shape[0]), fac_steps=int(f_ctab[:, 0].sum()), fac_len=nnzL + N + nnzX + 2,
                 rows_steps=[int(p.ctab[:, 0].sum()) for p in (rows_A, rows_P, rows_At)])
    return ResidentPlan(base=base, groups=groups, nnzX=nnzX, x_row=np.asarray(x_row, dtype=np.int32), x_col=np.asarray(x_col, dtype=np.int32),
                        f_ctab=f_ctab, f_task=f_task, f_len=f_len, f_a=f_a, f_b=f_b, f_k=f_k,
                        k_kind=base.ksrc_kind.copy(), k_idx=base.ksrc_idx.copy(),
                        sol=sol, sol_kind=sol_kind, sol_idx=sol_idx, sol_lcol=sol_lcol,
                        rows_A=rows_A, rows_P=rows_P, rows_At=rows_At, rows_A_ent=rows_A_ent, rows_P_ent=rows_P_ent,
                        rows_At_ent=rows_At_ent, out_ax=out_ax, out_px=out_px, out_aty=out_aty, w_slots=w_slots, stats=stats,
                        team=team)


# ------------------------------------------------------------------------------------------------
# coalesced (ELL) form of a canonicalisation map: out[k] = base[k] + sum_j coef[j, k] * theta[idx[j, k]]
def ell_map(M: sp.csr_matrix):
    M = sp.csr_matrix(M); M.sort_indices()
    rows = M.shape[0]
    cnt = np.diff(M.indptr)
    J = int(cnt.max()) if rows and M.nnz else 0
    idx = np.zeros((max(J, 1), max(rows, 1)), dtype=np.int32)
    coef = np.zeros((max(J, 1), max(rows, 1)), dtype=np.float64)
    for r in range(rows):
        s, e = M.indptr[r], M.indptr[r + 1]
        idx[:e - s, r] = M.indices[s:e]
        coef[:e - s, r] = M.data[s:e]
    return J, idx, coef


# ------------------------------------------------------------------------------------------------
# numpy replay of the kernel's algorithms (tests)
def replay_factor(pl: ResidentPlan, Ps, As, sigma, rho_inv) -> np.ndarray:
    """fac = [M | 1/d | X | 1] through the combined schedule (the M-form of numeric_ldl_m: undivided column
    entries, reciprocal pivots), K values preloaded into the destinations"""
    b = pl.base
    nnzL, N = b.nnzL, b.n + b.m
    fac = np.zeros(pl.fac_len)
    for d in range(nnzL + N):
        kind, idx = pl.k_kind[d], pl.k_idx[d]
        v = 0.0
        if kind == _rp.K_P:
            v = Ps[idx] + (sigma if d >= nnzL else 0.0)
        elif kind == _rp.K_A:
            v = As[idx]
        elif kind == _rp.K_SIGMA:
            v = sigma
        elif kind == _rp.K_RHO:
            v = -rho_inv[idx]
        fac[d] = v
    fac[pl.one_pos] = 1.0
    lane = np.arange(LANES)
    for c in range(pl.f_ctab.shape[0]):
        L, last, first, lg = (int(v) for v in pl.f_ctab[c])
        T, lw = pl.f_task[c], pl.f_len[c]
        al, rl = (lw & 0xFFFF).astype(np.int64), (lw >> 16).astype(np.int64)
        acc = np.zeros(LANES)
        base_e = first
        for s in range(L):
            act = al > s
            e = base_e + np.cumsum(act) - 1
            real = act & (rl > s)
            acc[real] += fac[pl.f_a[e[real]]] * fac[pl.f_k[e[real]]] * fac[pl.f_b[e[real]]]
            base_e += int(act.sum())
        g = 1 << lg
        red = acc.reshape(LANES // g, g).sum(axis=1)
        for t in range(0, LANES, g):
            if T[t] == NO_TASK:
                continue
            dest = int(T[t] & 0x7FFFFFFF)
            v = fac[dest] - red[t // g]
            fac[dest] = 1.0 / v if (T[t] & PIVOT_FLAG) else v
    return fac


def replay_solve_vals(pl: ResidentPlan, fac: np.ndarray) -> np.ndarray:
    nnzL, N = pl.base.nnzL, pl.base.n + pl.base.m
    k, i = pl.sol_kind, pl.sol_idx
    v = np.zeros(pl.sol.nnz)
    v[k == SRC_ONE] = 1.0
    s = k == SRC_NEG_L
    v[s] = -(fac[i[s]] * fac[nnzL + pl.sol_lcol[s]])
    s = k == SRC_DINV
    v[s] = fac[nnzL + i[s]]
    s = k == SRC_X
    v[s] = fac[nnzL + N + i[s]]
    return v


def replay_product(prog: _sp.RaggedProgram, ent: np.ndarray, vals: np.ndarray, w: np.ndarray) -> np.ndarray:
    """one product of the termination test through its row program, coefficients = the instance's matrix entries"""
    import dataclasses
    pv = np.where(ent >= 0, np.asarray(vals)[np.maximum(ent, 0)], 0.0)
    return _sp.execute_ragged(dataclasses.replace(prog, vals=pv), w)
