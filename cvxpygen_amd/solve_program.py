"""
Compiles the fixed LDL' factor of the family KKT matrix into a *solve program* for the GPU:
a short sequence of fully parallel sparse phases

        w[r]  <-  sum_k coef[k] * w[col[k]]        for all rows r of the phase at once,

executed by one wavefront per problem instance on a work vector w held in LDS.

Why not plain substitution: QDLDL-style forward / backward substitution (the reference's
`QDLDL_solve`, called inside `osqp_solve`, `cvxpygen/solvers/osqp.py:62`) is a chain of
N dependent column steps; level scheduling still leaves one step per level of the elimination tree
(242 levels for the MPC family: a Riccati-like sweep with one row per level).  A wavefront pays a
full LDS write->read round trip per level, so the phase count, not the flop count, bounds the
solve.  Here consecutive levels are merged into groups G and the unit-triangular diagonal block of
each group is inverted on the host (partitioned-inverse representation of L^-1):

    forward  :  y_G = T_G b_G - (T_G L_GE) y_E ,          T_G = L_GG^-1,  E = earlier groups
    backward :  x_G = T_G' D_G^-1 y_G - (T_G' L_LG') x_L ,                L = later groups

which is exact in exact arithmetic, costs a bounded amount of extra fill, and cuts the number of
dependent phases by an order of magnitude.  Group boundaries are chosen by dynamic programming on a
cost model (phase latency vs. multiply-add steps).  Products are formed in extended precision.

The same phase format also carries the sparse products of the termination check
(A x, P x, A' y) so that one executor serves the whole ADMM iteration.

Phase storage ("chunks"): every phase is cut into chunks of 64 lane-tasks.  In a chunk, a row is
split over g = 2^k consecutive lanes (g constant inside the chunk); lane t performs `len`
multiply-adds  acc += val[s][t] * w[col[s][t]]  and the g partial sums are reduced by a butterfly;
the first lane of the group stores to w[row].  val/col are stored step-major, lane-minor so that a
wavefront reads 512 contiguous bytes of values per step.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

import os as _os
import threading as _threading

# The planner's stage costs: pack_ragged(stage_scale=...) changes them for the duration of a call, and
# __graft_entry__.build() / generate_code plan several families from worker threads.  They are therefore PER THREAD
# (`_costs`): a scaled call of one thread is invisible to the plans the others are building, and no plan builder has to
# serialise on a lock (round 3 held one around every whole plan, annealing included).
class _Costs(_threading.local):
    def __init__(self):
        self.stage = STAGE_COST
        self.group = GROUP_STAGE_COST
        self.team = 1           # wavefronts that share a phase (pack_ragged(team=W)): chunk costs add up per wavefront only


LANES = 64
# cost model of the packer, in units of one multiply-add step of a wavefront (environment overrides are
# tuning knobs for experiments; a family library and the runtime must be built with the same values)
PHASE_COST = float(_os.environ.get('CPG_PHASE_COST', 14.0))      # fixed cost of a phase
CHUNK_COST = float(_os.environ.get('CPG_CHUNK_COST', 3.0))       # fixed cost of one more chunk inside a phase
GROUP_STAGE_COST = float(_os.environ.get('CPG_GROUP_STAGE_COST', 1.5))   # one stage of a power-of-two group reduction
MAX_GROUP_ROWS = int(_os.environ.get('CPG_MAX_GROUP_ROWS', 128))     # rows of a merged phase (its inverse fills in)


@dataclass
class Phase:
    """out rows (indices into w) and their sparse rows (lists of (col, coef))."""
    rows: np.ndarray
    cols: List[np.ndarray]
    vals: List[np.ndarray]
    intra: bool = False      # some row reads another row of the same phase -> deferred stores
    name: str = ''
    accumulate: bool = False  # w[row] += sum instead of w[row] = sum (ragged packing only; in-place rows)
    # rows read other rows' slots of the SAME phase (accumulate form of a merged group's diagonal-block inverse,
    # resident_plan.py): every gather of the phase must come before its first store.  The generated executors issue a
    # phase's gathers first anyway; chunk-by-chunk executors (run_program_stream) refuse such a program (chunk kind bit 2)
    deferred: bool = False

    @property
    def nnz(self) -> int:
        return int(sum(len(c) for c in self.cols))


def _levels(N: int, Lp: np.ndarray, Li: np.ndarray) -> np.ndarray:
    lev = np.zeros(N, dtype=np.int64)
    for j in range(N):
        s, e = Lp[j], Lp[j + 1]
        if e > s:
            rows = Li[s:e]
            np.maximum.at(lev, rows, lev[j] + 1)
    return lev


# ------------------------------------------------------------------------------------------------
def _team_cost(costs: Sequence[float]) -> float:
    """What the chunks of one phase cost: their sum on one wavefront; with a TEAM of W wavefronts per instance
    (pack_ragged(team=W), csrc/cpg_osqp_team.h) the chunks of a phase run side by side, longest first to the least
    loaded wavefront, and the phase takes as long as the busiest one."""
    W = int(_costs.team)
    if W <= 1 or len(costs) <= 1:
        return float(sum(costs))
    loads = [0.0] * W
    for c in sorted(costs, reverse=True):
        k = loads.index(min(loads))
        loads[k] += c
    # (a little for the total too: of two plans with the same busiest wavefront the one with fewer steps)
    return max(loads) + 1e-3 * float(sum(costs))


def _chunk_plan(lens: Sequence[int]) -> Tuple[float, int, List[Tuple[int, int, List[int]]]]:
    """Best split of rows with `lens` multiply-adds into chunks.  Returns (cost, n_chunks,
    [(g, len, [row positions]) ...])."""
    lens = np.asarray(lens, dtype=np.int64)
    R = len(lens)
    if R == 0:
        return 0.0, 0, []
    best = None
    total = int(lens.sum())
    cand = sorted(set([1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128,
                       max(1, -(-total // LANES)), max(1, -(-total // (LANES * max(1, int(_costs.team))))), int(lens.max())]))
    for T in cand:
        g = np.ones(R, dtype=np.int64)
        need = -(-lens // T)
        g = np.where(need <= 1, 1, 2 ** np.ceil(np.log2(np.maximum(need, 1))).astype(np.int64))
        g = np.minimum(g, LANES)
        chunks = []
        costs = []
        for gv in np.unique(g):
            idx = np.nonzero(g == gv)[0]
            idx = idx[np.argsort(-lens[idx], kind='stable')]
            per = LANES // gv
            for s in range(0, len(idx), per):
                sel = idx[s:s + per]
                ln = int(-(-lens[sel].max() // gv)) if len(sel) else 0
                ln = max(ln, 1)
                chunks.append((int(gv), ln, [int(v) for v in sel]))
                costs.append(ln + CHUNK_COST + _costs.group * np.log2(gv))
        cost = _team_cost(costs)
        if best is None or cost < best[0]:
            best = (cost, len(chunks), chunks)
    return best


def _phase_cost(lens: Sequence[int], intra: bool) -> float:
    cost, nch, _ = _chunk_plan(lens)
    return PHASE_COST + cost


# ------------------------------------------------------------------------------------------------
def _approx_cost(lens: np.ndarray, intra: bool) -> float:
    """Cheap stand-in for `_phase_cost` used inside the dynamic programme."""
    total = float(lens.sum())
    steps = max(np.ceil(total / LANES), np.ceil(lens.max() / LANES), 1.0)
    lanes_needed = total / steps
    nch = max(1.0, np.ceil(lanes_needed / LANES))
    split = max(1.0, lens.max() / steps)
    return PHASE_COST + 1.2 * total / LANES + CHUNK_COST * nch + 1.5 * np.log2(split) + 1.0


class _GroupBuilder:
    """Grows a group of consecutive levels a..b one level at a time, maintaining
    T = M_GG^-1 and W = -T M_GE (dense over all columns) in extended precision.  Rows are kept in
    (level, index) order, itself a topological order, so a new level only appends rows."""

    def __init__(self, N, M: sp.csr_matrix, sc: np.ndarray, dtype=np.longdouble):
        self.N, self.M, self.sc, self.dt = N, M, sc, dtype
        self.G = np.zeros(0, dtype=np.int64)
        self.pos = np.full(N, -1, dtype=np.int64)
        self.T = np.zeros((0, 0), dtype=dtype)
        self.W = np.zeros((0, N), dtype=dtype)

    def extend(self, rows: np.ndarray) -> None:
        M, N = self.M, self.N
        ng, nr = len(self.G), len(rows)
        Mi = np.zeros((nr, ng), dtype=self.dt)
        Me = np.zeros((nr, N), dtype=self.dt)
        for k, r in enumerate(rows):
            c = M.indices[M.indptr[r]:M.indptr[r + 1]]
            v = M.data[M.indptr[r]:M.indptr[r + 1]]
            p = self.pos[c]
            inn = p >= 0
            Mi[k, p[inn]] = v[inn]
            Me[k, c[~inn]] = v[~inn]
        Tn = np.zeros((ng + nr, ng + nr), dtype=self.dt)
        Tn[:ng, :ng] = self.T
        if ng:
            Tn[ng:, :ng] = -(Mi @ self.T)
        Tn[ng:, ng:] = np.eye(nr, dtype=self.dt)
        Wn = -Me
        if ng:
            Wn = Wn - Mi @ self.W
        self.T = Tn
        self.W = np.vstack([self.W, Wn])
        self.pos[rows] = ng + np.arange(nr)
        self.G = np.concatenate([self.G, rows])

    def row_lens(self) -> np.ndarray:
        return (np.count_nonzero(self.T, axis=1) + np.count_nonzero(self.W, axis=1)).astype(np.int64)

    def phase(self, name: str) -> Phase:
        G, ng = self.G, len(self.G)
        Ts = self.T * self.sc[G][None, :]
        cols, vals = [], []
        for k in range(ng):
            ti = np.nonzero(Ts[k, :k + 1])[0]
            ti = np.concatenate([[k], ti[ti != k]]).astype(np.int64)
            wi = np.nonzero(self.W[k])[0]
            cols.append(np.concatenate([G[ti], wi]).astype(np.int64))
            vals.append(np.concatenate([Ts[k, ti], self.W[k, wi]]).astype(np.float64))
        intra = bool(np.count_nonzero(self.T) > ng)
        return Phase(G.copy(), cols, vals, intra=intra, name=name)


def _compile_lower(N: int, M: sp.csr_matrix, scale: Optional[np.ndarray], merge: bool, tag: str
                   ) -> List[Phase]:
    """Phases computing  w <- M^-1 (scale * w)  for a unit lower-triangular M (CSR, strict part
    stored, unit diagonal implied)."""
    M = sp.csr_matrix(M)
    M.sort_indices()
    Mc = sp.csc_matrix(M)
    lev = _levels(N, Mc.indptr, Mc.indices)
    nlev = int(lev.max()) + 1 if N else 0
    order = np.lexsort((np.arange(N), lev))
    lev_ptr = np.zeros(nlev + 1, dtype=np.int64)
    np.add.at(lev_ptr, lev + 1, 1)
    lev_ptr = np.cumsum(lev_ptr)
    rows_of = [np.sort(order[lev_ptr[a]:lev_ptr[a + 1]]) for a in range(nlev)]
    sc = np.ones(N, dtype=np.longdouble) if scale is None else scale.astype(np.longdouble)

    def build(a: int, b: int) -> Phase:
        gb = _GroupBuilder(N, M, sc)
        for lv in range(a, b + 1):
            gb.extend(rows_of[lv])
        return gb.phase(f'{tag}{a}' if a == b else f'{tag}{a}-{b}')

    if not merge:
        return [build(a, a) for a in range(nlev)]

    # dynamic programme over level boundaries (cheap cost model, incremental group growth)
    best = np.full(nlev + 1, np.inf)
    best[0] = 0.0
    choice = np.zeros(nlev + 1, dtype=np.int64)
    for a in range(nlev):
        gb = _GroupBuilder(N, M, sc, dtype=np.float64)   # cost model only needs the pattern
        nrows = 0
        for b in range(a, nlev):
            nrows += len(rows_of[b])
            if b > a and nrows > MAX_GROUP_ROWS:
                break
            gb.extend(rows_of[b])
            cst = _approx_cost(gb.row_lens(), b > a)
            if best[a] + cst < best[b + 1]:
                best[b + 1] = best[a] + cst
                choice[b + 1] = a
            if b > a and cst > 4.0 * PHASE_COST * (b - a + 1):
                break
    bounds = []
    b = nlev
    while b > 0:
        a = int(choice[b])
        bounds.append((a, b - 1))
        b = a
    return [build(a, b) for a, b in bounds[::-1]]


def compile_ldl(N: int, Lp: np.ndarray, Li: np.ndarray, Lx: np.ndarray, D: np.ndarray,
                perm: np.ndarray, merge: bool = True, devpos: Optional[np.ndarray] = None
                ) -> List[Phase]:
    """Phases for  w <- K^-1 w  with K = P' L D L' P.  Row / column indices are LOGICAL entries of
    w: natural KKT index i, or devpos[i] when a device ordering is given (both the fill-reducing
    permutation and the device ordering are folded into the indices; nothing is permuted at run
    time)."""
    L = sp.csc_matrix((Lx, Li, Lp), shape=(N, N))
    fwd = _compile_lower(N, sp.csr_matrix(L), None, merge, 'F')
    # backward: L' x = D^-1 y.  Reverse the index order to obtain a lower-triangular system.
    J = np.arange(N)[::-1]
    Lt_rev = sp.csr_matrix(L.T)[J][:, J]
    bwd = _compile_lower(N, sp.csr_matrix(Lt_rev), (1.0 / D)[J], merge, 'B')
    mp = perm if devpos is None else devpos[perm]
    out = []
    for ph in fwd:
        out.append(Phase(mp[ph.rows], [mp[c] for c in ph.cols], ph.vals, ph.intra, ph.name))
    for ph in bwd:
        out.append(Phase(mp[J[ph.rows]], [mp[J[c]] for c in ph.cols], ph.vals, ph.intra, ph.name))
    return out


def spmv_phase(M: sp.spmatrix, col_offset: int, name: str) -> Phase:
    """Phase computing (M v) with v = w[col_offset : col_offset + M.shape[1]]; row r of the result
    is delivered to lane-task r (natural layout, one row per lane, see `pack(natural=True)`)."""
    M = sp.csr_matrix(M)
    rows = np.arange(M.shape[0], dtype=np.int64)
    cols = [M.indices[M.indptr[r]:M.indptr[r + 1]].astype(np.int64) + col_offset for r in rows]
    vals = [M.data[M.indptr[r]:M.indptr[r + 1]].astype(np.float64) for r in rows]
    return Phase(rows, cols, vals, intra=False, name=name)


# ------------------------------------------------------------------------------------------------
@dataclass
class PackedProgram:
    """Flat arrays consumed by the HIP executor.

    hdr  int32 [n_chunks, 4]: (len, log2 g, flags, first step)
    rows uint16 [n_chunks, 64]: LDS slot the lane stores its result to (0xFFFF: none)
    vals float64 [n_steps, 64];  cols uint16 [n_steps, 64]: LDS slot of the operand
    n_slots: size of the LDS work vector (>= N; merged phases write to fresh slots)
    final_pos int32 [N]: slot holding logical entry i after the whole program ran
    """
    hdr: np.ndarray
    rows: np.ndarray
    vals: np.ndarray
    cols: np.ndarray
    n_phases: int
    nnz: int
    n_slots: int = 0
    final_pos: Optional[np.ndarray] = None

    @property
    def n_chunks(self) -> int:
        return int(self.hdr.shape[0])

    @property
    def steps(self) -> int:
        return int(self.vals.shape[0])


FLAG_LAST = 2
NO_ROW = 0xFFFF


def assign_slots(phases: List[Phase], N: int, slot_perm: Optional[np.ndarray] = None):
    """LDS slot allocation.  Entry i starts in slot i (slot slot_perm[i] when a bank-aware numbering is
    given, cvxpygen_amd/slot_layout.py: every slot number below is then mapped through it).  A phase whose rows read each other
    (`intra`) writes its results to free slots and releases the slots its rows occupied, so every
    chunk can store immediately; other phases update in place.  Returns (per-phase output slots,
    per-phase column slots, n_slots, final_pos)."""
    cur = np.arange(N, dtype=np.int64)
    free: List[int] = []
    n_slots = N
    outs, ins = [], []
    for ph in phases:
        ins.append([cur[c] for c in ph.cols])
        if not ph.intra:
            outs.append(cur[ph.rows].copy())
            continue
        new = np.zeros(len(ph.rows), dtype=np.int64)
        for k in range(len(ph.rows)):
            if free:
                new[k] = free.pop()
            else:
                new[k] = n_slots
                n_slots += 1
        old = cur[ph.rows].copy()
        cur[ph.rows] = new
        free.extend(int(v) for v in old[::-1])
        outs.append(new)
    if slot_perm is not None:
        sp_ = np.asarray(slot_perm, dtype=np.int64)
        assert len(sp_) == n_slots and np.array_equal(np.sort(sp_), np.arange(n_slots))
        outs = [sp_[o] for o in outs]
        ins = [[sp_[c] for c in ph_in] for ph_in in ins]
        return outs, ins, n_slots, sp_[cur]
    return outs, ins, n_slots, cur.copy()


def pack(phases: List[Phase], natural: bool = False, N: Optional[int] = None,
         slot_perm: Optional[np.ndarray] = None) -> PackedProgram:
    return _pack(phases, natural, N, slot_perm)


def _pack(phases: List[Phase], natural: bool = False, N: Optional[int] = None,
          slot_perm: Optional[np.ndarray] = None) -> PackedProgram:
    """natural=True: chunk c holds rows 64c .. 64c+63, one row per lane (results are consumed in
    registers by the lane that owns the element; nothing is stored to w)."""
    hdr, rows_out, vals_out, cols_out = [], [], [], []
    off = 0
    nnz = 0
    if natural:
        outs = [ph.rows for ph in phases]
        ins = [ph.cols for ph in phases]
        n_slots, final_pos = 0, None
    else:
        if N is None:
            N = int(max(int(ph.rows.max()) for ph in phases)) + 1 if phases else 0
        outs, ins, n_slots, final_pos = assign_slots(phases, N, slot_perm)
    for ph, out_slots, col_slots in zip(phases, outs, ins):
        lens = [len(c) for c in ph.cols]
        nnz += sum(lens)
        if natural:
            plan = []
            for s in range(0, len(ph.rows), LANES):
                sel = list(range(s, min(s + LANES, len(ph.rows))))
                plan.append((1, max(1, max(lens[k] for k in sel)), sel))
        else:
            _, _, plan = _chunk_plan(lens)
        for ci, (g, ln, sel) in enumerate(plan):
            V = np.zeros((ln, LANES))
            Cc = np.zeros((ln, LANES), dtype=np.uint16)
            R = np.full(LANES, NO_ROW, dtype=np.uint16)
            for k, rp in enumerate(sel):
                base = k * g
                R[base] = out_slots[rp] if not natural else ph.rows[rp] % LANES
                c, v = col_slots[rp], ph.vals[rp]
                seg = -(-len(c) // g) if len(c) else 0
                for t in range(g):
                    cs, vs = c[t * seg:(t + 1) * seg], v[t * seg:(t + 1) * seg]
                    V[:len(vs), base + t] = vs
                    Cc[:len(cs), base + t] = cs
            hdr.append([ln, int(np.log2(g)), 0, off])
            rows_out.append(R)
            vals_out.append(V)
            cols_out.append(Cc)
            off += ln
    if hdr:
        hdr[-1][2] |= FLAG_LAST
    return PackedProgram(
        hdr=np.asarray(hdr, dtype=np.int32).reshape(-1, 4),
        rows=np.asarray(rows_out, dtype=np.uint16).reshape(-1, LANES),
        vals=np.concatenate(vals_out, axis=0) if vals_out else np.zeros((0, LANES)),
        cols=np.concatenate(cols_out, axis=0) if cols_out else np.zeros((0, LANES), dtype=np.uint16),
        n_phases=len(phases), nnz=nnz, n_slots=n_slots, final_pos=final_pos)


def execute_packed(prog: PackedProgram, w: np.ndarray, natural: bool = False):
    """Host emulation of the HIP executor (tests).  `w` must have prog.n_slots entries; chunk by
    chunk: multiply-add steps, group reduction, immediate store -- exactly as `run_program` in
    csrc/cpg_osqp_kernel.h.  With natural=True returns the per-chunk lane results instead."""
    nat = []
    for c in range(prog.n_chunks):
        ln, lg, _, off = prog.hdr[c]
        g = 1 << lg
        acc = np.zeros(LANES)
        for s in range(ln):
            acc += prog.vals[off + s] * w[prog.cols[off + s]]
        if natural:
            nat.append(acc)
            continue
        red = acc.reshape(LANES // g, g).sum(axis=1)
        R = prog.rows[c][::g]
        ok = R != NO_ROW
        w[R[ok]] = red[ok]
    return nat if natural else w


# ------------------------------------------------------------------------------------------------
@dataclass
class RaggedProgram:
    """Compact (padding-free) form of a solve program, small enough to stay resident in LDS.

    ctab int32 [n_chunks, 4]: (max len, log2 g, first entry, 0)  -- uniform chunk: every row is split
                              over g = 2^k lanes, butterfly reduction;
                              (max len, stages S, first entry, 1) -- balanced chunk: a row occupies a
                              variable number (<= 16) of ADJACENT lanes inside one 16-lane DPP row;
                              segmented reduction in S masked shift-add stages (lane t adds lane
                              t + 2^j iff bit j of its mask is set), the row's first lane stores
    desc uint32 [n_chunks, 64]: output slot (low 16 bits, 0xFFFF none) | number of entries << 16
                              (12 bits) | stage mask << 28 (balanced chunks)
    vals float64 [nnz + 1]; cols uint16 [nnz + 1]: entries step-major.  Inside a chunk the lanes are
    ordered by non-increasing number of entries, so the lanes that still have an entry at step s
    are a prefix [0, cnt_s) and lane t finds its entry at  first + sum_{s' < s} cnt_s' + t.
    `cols` holds BYTE offsets into the work vector (slot * 8).  The last entry (index nnz) is a
    zero coefficient on slot 0: lanes without an entry read it instead of being masked off.
    """
    ctab: np.ndarray
    desc: np.ndarray
    vals: np.ndarray
    cols: np.ndarray
    n_phases: int
    n_slots: int
    final_pos: np.ndarray
    chunk_phase: Optional[np.ndarray] = None    # phase index of every chunk
    # [n_chunks, 64]: the row (numbered inside its chunk) whose partial sum a lane accumulates, -1 for dummy lanes.  The entries
    # of a row may sit in ANY of its (lane, step) cells -- slot_layout.optimise_entries permutes them to avoid LDS bank conflicts
    lane_row: Optional[np.ndarray] = None

    @property
    def n_chunks(self) -> int:
        return int(self.ctab.shape[0])

    @property
    def nnz(self) -> int:
        return int(self.vals.shape[0])          # including the trailing dummy entry

    def fingerprint(self) -> int:
        """32-bit FNV-1a over the structural tables; ties a generated library to its family"""
        h = 0x811C9DC5
        for arr in (self.ctab.astype(np.int32), self.desc.astype(np.uint32), self.cols.astype(np.uint16)):
            for b in arr.tobytes():
                h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
        return h

    def lds_bytes(self) -> int:
        """bytes the program occupies in LDS (values, indices, descriptors, chunk table)"""
        return 8 * self.nnz + 8 * (-(-self.nnz // 4)) + 4 * 64 * self.n_chunks + 16 * self.n_chunks


DPP_ROW = 16            # cross-lane shifts of the segmented reduction stay inside 16-lane DPP rows
SEG_KMAX = 8            # lanes per row in a balanced chunk (3 mask bits next to a 13-bit slot)
AUTO_BALANCED_MARGIN = float(_os.environ.get('CPG_AUTO_BALANCED_MARGIN', 0.9))
STAGE_COST = float(_os.environ.get('CPG_STAGE_COST', 1.0))       # one stage of the segmented reduction (default of _costs.stage)
_costs = _Costs()


def _balanced_layout(lens: np.ndarray, seg_len: int):
    """Rows split into k = ceil(len / seg_len) <= 16 adjacent lanes of s = ceil(len / k) entries each
    (short tails padded with zero coefficients), placed in order of non-increasing s so that lane
    lengths are monotone inside a chunk; a row never straddles a 16-lane DPP row (dummy lanes of the
    current length fill the gap).  Returns [chunk]: chunk = [(row position | -1, lane of the row, k, s)]."""
    rows = []
    for r, ln in enumerate(lens):
        ln = max(int(ln), 1)
        k = min(SEG_KMAX, -(-ln // seg_len))
        rows.append((-(-ln // k), k, r))
    rows.sort(key=lambda t: (-t[0], -t[1], t[2]))
    chunks, cur = [], []
    pending = list(rows)
    while pending:
        s_cur = pending[0][0]
        off = len(cur) % DPP_ROW
        room_row = DPP_ROW - off
        room_chunk = LANES - len(cur)
        # among the rows with the current (largest remaining) segment length take the widest that fits
        pick = None
        for i, (sr, k, r) in enumerate(pending):
            if sr != s_cur:
                break
            if k <= room_row and k <= room_chunk:
                pick = i
                break
        if pick is None:
            if room_chunk < DPP_ROW and all(k > room_chunk for (sr, k, r) in pending if sr == s_cur):
                chunks.append(cur)      # nothing of this length fits the rest of the chunk
                cur = []
                continue
            cur.append((-1, 0, 1, s_cur))                              # dummy lane: s_cur zero entries
            if len(cur) == LANES:
                chunks.append(cur)
                cur = []
            continue
        sr, k, r = pending.pop(pick)
        for j in range(k):
            cur.append((r, j, k, sr))
        if len(cur) == LANES:
            chunks.append(cur)
            cur = []
    if cur:
        chunks.append(cur)
    return chunks


def _balanced_plan(lens: Sequence[int]):
    """Segment length that minimises  sum_chunks (steps + CHUNK_COST + STAGE_COST * stages)."""
    lens = np.asarray(lens, dtype=np.int64)
    if len(lens) == 0:
        return 0.0, []
    best = None
    mx = int(max(lens.max(), 1))
    for seg_len in sorted(set(list(range(1, min(mx, 24) + 1)) + [mx, -(-mx // 2), -(-mx // 3), -(-mx // 4)])):
        chunks = _balanced_layout(lens, seg_len)
        costs = []
        for ch in chunks:
            kmax = max(k for (_, _, k, _) in ch)
            costs.append(ch[0][3] + CHUNK_COST + _costs.stage * int(np.ceil(np.log2(kmax))) if kmax > 1 else ch[0][3] + CHUNK_COST)
        cost = _team_cost(costs)
        if best is None or cost < best[0]:
            best = (cost, chunks)
    return best




def pack_ragged(phases: List[Phase], N: int, balanced=False, stage_scale: float = 1.0,
                slot_perm: Optional[np.ndarray] = None, team: int = 1) -> RaggedProgram:
    return _pack_ragged(phases, N, balanced, stage_scale, slot_perm, team)


def _pack_ragged(phases: List[Phase], N: int, balanced=False, stage_scale: float = 1.0,
                 slot_perm: Optional[np.ndarray] = None, team: int = 1) -> RaggedProgram:
    """balanced=False: rows of a chunk are split over a uniform power-of-two number of lanes;
    balanced=True: variable number of adjacent lanes per row + segmented reduction (fewer steps when
    row lengths are uneven); balanced='auto': per phase whichever of the two the cost model prefers.
    stage_scale scales what the planner charges for a reduction stage relative to a step: the
    executors whose steps are memory requests (run_program_stream) want fewer, wider steps than the
    LDS-resident one the defaults were tuned on.
    team = W > 1: the program is planned for a team of W wavefronts per instance (csrc/cpg_osqp_team.h) -- the chunks of
    a phase are spread over the wavefronts (team_assignment), so the planner prefers plans whose BUSIEST wavefront
    is done first: more, shorter chunks per phase."""
    if int(team) != int(_costs.team):
        saved_team = _costs.team
        _costs.team = int(team)
        try:
            return _pack_ragged(phases, N, balanced, stage_scale, slot_perm, team)
        finally:
            _costs.team = saved_team
    if stage_scale != 1.0:
        saved = (_costs.stage, _costs.group)
        _costs.stage, _costs.group = saved[0] * stage_scale, saved[1] * stage_scale
        try:
            return _pack_ragged(phases, N, balanced, slot_perm=slot_perm, team=team)
        finally:
            _costs.stage, _costs.group = saved
    outs, ins, n_slots, final_pos = assign_slots(phases, N, slot_perm)
    if n_slots * 8 > 0xFFFF:
        raise NotImplementedError('work vector too large for 16-bit byte offsets')
    ctab, desc, vals, cols, chunk_phase, lane_row = [], [], [], [], [], []
    first = 0
    for pi, (ph, out_slots, col_slots) in enumerate(zip(phases, outs, ins)):
        lens = [len(c) for c in ph.cols]
        use_balanced = bool(balanced) and max(lens, default=0) < 4096 and n_slots < 0x1FFF
        if use_balanced and balanced == 'auto':
            # per phase whichever layout the cost model prefers (few long rows: power-of-two groups
            # across the wave; many short rows: variable segments inside 16-lane rows)
            # (the segmented reduction is dearer than its stage count suggests: demand a clear win)
            use_balanced = _balanced_plan(lens)[0] < AUTO_BALANCED_MARGIN * _chunk_plan(lens)[0]
        if use_balanced:
            _, chunks = _balanced_plan(lens)
            for ch in chunks:
                chunk_phase.append(pi)
                lane_c = [np.zeros(0, dtype=np.int64)] * LANES
                lane_v = [np.zeros(0)] * LANES
                D = np.full(LANES, NO_ROW, dtype=np.uint32)
                mask = np.zeros(LANES, dtype=np.uint32)
                kmax = 1
                lr = np.full(LANES, -1, dtype=np.int32)
                for t, (rp, j, k, sr) in enumerate(ch):
                    if rp < 0:
                        lane_c[t], lane_v[t] = np.zeros(sr, dtype=np.int64), np.zeros(sr)
                        continue
                    lr[t] = rp
                    c, v = np.asarray(col_slots[rp], dtype=np.int64), np.asarray(ph.vals[rp], dtype=np.float64)
                    cs, vs = c[j * sr:(j + 1) * sr], v[j * sr:(j + 1) * sr]
                    pad = sr - len(cs)
                    lane_c[t] = np.concatenate([cs, np.zeros(pad, dtype=np.int64)])
                    lane_v[t] = np.concatenate([vs, np.zeros(pad)])
                    if j == 0:
                        D[t] = out_slots[rp]
                        assert t % DPP_ROW + k <= DPP_ROW
                    kmax = max(kmax, k)
                    for st in range(3):
                        if j + (1 << st) < k:
                            mask[t] |= 1 << st
                S = int(np.ceil(np.log2(kmax))) if kmax > 1 else 0
                ll = np.array([len(x) for x in lane_c], dtype=np.int64)
                assert np.all(np.diff(ll) <= 0), 'lane lengths must be non-increasing'
                L = int(ll.max())
                D = D | (ll.astype(np.uint32) << 16) | (mask << 28)
                n_ent = 0
                for s in range(L):
                    cnt = int((ll > s).sum())
                    vals.append(np.array([lane_v[t][s] for t in range(cnt)]))
                    cols.append(np.array([8 * lane_c[t][s] for t in range(cnt)], dtype=np.uint16))
                    n_ent += cnt
                ctab.append([L, S, first, 1 | (2 if ph.accumulate else 0) | (4 if ph.deferred else 0)])
                desc.append(D)
                lane_row.append(lr)
                first += n_ent
            continue
        _, _, plan = _chunk_plan(lens)
        for g, ln, sel in plan:
            chunk_phase.append(pi)
            # rows of a chunk in order of non-increasing segment length; every lane of a row gets
            # the same number of entries (short segments are padded with zero coefficients)
            segs = [(-(-len(col_slots[rp]) // g) if len(col_slots[rp]) else 0) for rp in sel]
            order = np.argsort(-np.asarray(segs), kind='stable')
            lane_c = [np.zeros(0, dtype=np.int64)] * LANES
            lane_v = [np.zeros(0)] * LANES
            D = np.full(LANES, NO_ROW, dtype=np.uint32)
            lr = np.full(LANES, -1, dtype=np.int32)
            for k, oi in enumerate(order):
                rp, seg = sel[oi], segs[oi]
                base = k * g
                D[base] = out_slots[rp]
                lr[base:base + g] = rp
                c, v = np.asarray(col_slots[rp], dtype=np.int64), np.asarray(ph.vals[rp], dtype=np.float64)
                for t in range(g):
                    cs, vs = c[t * seg:(t + 1) * seg], v[t * seg:(t + 1) * seg]
                    pad = seg - len(cs)
                    lane_c[base + t] = np.concatenate([cs, np.zeros(pad, dtype=np.int64)])
                    lane_v[base + t] = np.concatenate([vs, np.zeros(pad)])
            ll = np.array([len(x) for x in lane_c], dtype=np.int64)
            assert np.all(np.diff(ll) <= 0), 'lane lengths must be non-increasing'
            L = int(ll.max()) if len(ll) else 0
            D = D | (ll.astype(np.uint32) << 16)
            n_ent = 0
            for s in range(L):
                cnt = int((ll > s).sum())
                vals.append(np.array([lane_v[t][s] for t in range(cnt)]))
                cols.append(np.array([8 * lane_c[t][s] for t in range(cnt)], dtype=np.uint16))
                n_ent += cnt
            ctab.append([L, int(np.log2(g)), first, (2 if ph.accumulate else 0) | (4 if ph.deferred else 0)])
            desc.append(D)
            lane_row.append(lr)
            first += n_ent
    vals.append(np.zeros(1))
    cols.append(np.zeros(1, dtype=np.uint16))
    return RaggedProgram(
        ctab=np.asarray(ctab, dtype=np.int32).reshape(-1, 4),
        desc=np.asarray(desc, dtype=np.uint32).reshape(-1, LANES),
        vals=np.concatenate(vals), cols=np.concatenate(cols).astype(np.uint16),
        n_phases=len(phases), n_slots=n_slots, final_pos=final_pos,
        chunk_phase=np.asarray(chunk_phase, dtype=np.int32),
        lane_row=np.asarray(lane_row, dtype=np.int32).reshape(-1, LANES))


def execute_ragged(prog: RaggedProgram, w: np.ndarray) -> np.ndarray:
    """Host emulation of `run_program_lds` (tests).  Chunks of a `deferred` phase (kind bit 2) store when their
    phase is complete: all of its gathers see the values from before the phase."""
    lane = np.arange(LANES)
    dummy = prog.nnz - 1
    pending = []                        # (slots, values, accumulate) of the open deferred phase

    def flush():
        for R_, v_, acc_ in pending:
            w[R_] = (w[R_] + v_) if acc_ else v_
        pending.clear()
    for c in range(prog.n_chunks):
        L, lg, first, kind = (int(v) for v in prog.ctab[c])
        if pending and (not (kind & 4) or prog.chunk_phase is None or prog.chunk_phase[c] != prog.chunk_phase[c - 1]):
            flush()
        d = prog.desc[c]
        row, ln = d & 0xFFFF, d >> 16
        accumulate = bool(kind & 2)
        if kind & 1:
            ln = ln & 0xFFF            # balanced chunk: segmented shift-add reduction
            acc = np.zeros(LANES)
            base = first
            for s in range(L):
                act = ln > s
                e = np.where(act, base + lane, dummy)
                acc += prog.vals[e] * w[prog.cols[e] // 8]
                base += int(act.sum())
            for st in range(lg):
                sh = np.zeros(LANES)
                src = lane + (1 << st)
                okl = (src < LANES) & (src // DPP_ROW == lane // DPP_ROW)
                sh[okl] = acc[src[okl]]
                acc = acc + np.where(((d >> 28) >> st) & 1, sh, 0.0)
            ok = row != NO_ROW
            if kind & 4:
                pending.append((row[ok].copy(), acc[ok].copy(), accumulate))
            else:
                w[row[ok]] = (w[row[ok]] + acc[ok]) if accumulate else acc[ok]
            continue
        g = 1 << lg
        acc = np.zeros(LANES)
        base = first
        for s in range(L):
            act = ln > s
            e = np.where(act, base + lane, dummy)
            acc += prog.vals[e] * w[prog.cols[e] // 8]
            base += int(act.sum())
        red = acc.reshape(LANES // g, g).sum(axis=1)
        R = row[::g]
        ok = R != NO_ROW
        if kind & 4:
            pending.append((R[ok].copy(), red[ok].copy(), accumulate))
        else:
            w[R[ok]] = (w[R[ok]] + red[ok]) if accumulate else red[ok]
    flush()
    return w


GEN_DUMMY_SLOTS = 16        # generated executor: store targets of lanes without a row (one per lane of a 16-lane store group)
GEN_EXTRA_SLOTS = GEN_DUMMY_SLOTS + 1     # ... and one slot that always holds 0.0 (operand of idle lanes)


def execution_steps(prog: RaggedProgram):
    """The multiply-add steps of the GENERATED executor in the order it runs them (cvxpygen_amd/codegen.py):
    phase by phase, round-robin over the chunks of a phase.  [(phase index, chunk, first entry, active lanes)]"""
    lens_all = np.where((prog.ctab[:, 3:4] & 1) == 1, (prog.desc >> 16) & 0xFFF, prog.desc >> 16)
    phases = {}
    for c in range(prog.n_chunks):
        phases.setdefault(int(prog.chunk_phase[c]), []).append(c)
    steps = []
    for pi_, p in enumerate(sorted(phases)):
        per_chunk = []
        for c in phases[p]:
            L, first = int(prog.ctab[c, 0]), int(prog.ctab[c, 2])
            e, lst = first, []
            for s_ in range(L):
                cnt = int((lens_all[c] > s_).sum())
                lst.append((pi_, c, e, cnt))
                e += cnt
            per_chunk.append(lst)
        for s_ in range(max(len(l) for l in per_chunk)):
            for lst in per_chunk:
                if s_ < len(lst):
                    steps.append(lst[s_])
    return steps


def team_assignment(prog: RaggedProgram, W: int) -> np.ndarray:
    """Wavefront of every chunk when a team of W wavefronts executes the program (csrc/cpg_osqp_team.h, codegen.
    emit_team_program): inside a phase the chunk with the most steps goes to the least loaded wavefront (the lowest
    numbered one on ties)."""
    wave = np.zeros(prog.n_chunks, dtype=np.int32)
    phases = {}
    for c in range(prog.n_chunks):
        phases.setdefault(int(prog.chunk_phase[c]), []).append(c)
    for p in sorted(phases):
        loads = [0.0] * W
        for c in sorted(phases[p], key=lambda c_: (-int(prog.ctab[c_, 0]), c_)):
            k = min(range(W), key=lambda k_: (loads[k_], k_))
            loads[k] += int(prog.ctab[c, 0]) + CHUNK_COST
            wave[c] = k
    return wave


def padded_offsets_fit(prog: RaggedProgram, N: int, waves: int = 8, lds_bytes: int = 160 * 1024) -> bool:
    """Should the generated executor store the operand offsets for all 64 lanes of every step (idle lanes point
    at the zero slot: no masking of partial steps)?  Yes when that costs no resident wavefront (of at most
    `waves`) against the ragged layout.  Same arithmetic as cpg_hip.cpp / cpg_osqp_kernel.h."""
    n_steps = int(prog.ctab[:, 0].sum())
    nnzp = prog.nnz + 64
    per_wave = 8 * (prog.n_slots + GEN_EXTRA_SLOTS)

    def fit(n_off):
        fixed = 8 * N + 8 * (nnzp + (n_off + 3) // 4 + 16 * ((prog.n_chunks + 3) & ~3))
        return max(0, min(waves, (lds_bytes - fixed) // per_wave))
    return fit(64 * ((n_steps + 3) & ~3)) >= max(1, fit(nnzp))


def gathered_slots(prog: RaggedProgram, idle_zero: bool = False) -> np.ndarray:
    """[n_steps, 64]: the slot every lane of every multiply-add step of the GENERATED executor gathers
    (cvxpygen_amd/codegen.py).  Lanes past the active prefix of a partial step read either the zero slot
    behind the work vector (idle_zero: offsets stored for all 64 lanes) or, in the ragged layout, the entries
    that follow in the flat program (their lanes then sit out the multiply-add), past its end the zero padding
    (offset 0)."""
    flat = np.concatenate([prog.cols.astype(np.int64) // 8, np.zeros(2 * LANES, dtype=np.int64)])
    rows = []
    for c in range(prog.n_chunks):
        L, _, first, kind = (int(v) for v in prog.ctab[c])
        d = prog.desc[c]
        ln = ((d >> 16) & 0xFFF) if (kind & 1) else (d >> 16)
        e = first
        for s_ in range(L):
            cnt = int((ln > s_).sum())
            row = flat[e:e + LANES].copy()
            if idle_zero:
                row[cnt:] = prog.n_slots + GEN_DUMMY_SLOTS
            rows.append(row)
            e += cnt
    return np.asarray(rows, dtype=np.int64).reshape(-1, LANES)
