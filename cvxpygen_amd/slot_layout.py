"""
Bank-aware numbering of the LDS work-vector slots of a solve program.

Every multiply-add step of the solve program gathers one 8-byte operand per lane from the work vector
(`ds_read_b64` with per-lane addresses).  On CDNA4 the LDS services such a read in two groups of 32 lanes;
inside a group, lanes that address DIFFERENT 8-byte slots on the same pair of banks (slot mod 32) are
serialised: a group costs as many LDS cycles as the most loaded bank pair has distinct slots
(MI355X_MICROARCH.md, LDS section; calibrated with scripts/micro/lds_conflicts.hip: the measured cycles of a
gather are exactly that count).  The reduce-stores at the phase ends are scattered too: a `ds_write_b64` is served
in four groups of 16 lanes over 32 four-byte banks, so 8-byte slots of one group collide modulo 16.
With the natural numbering the gathers of the MPC 12/4/10 program
spend 32 % of all LDS cycles on such conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) while the LDS
pipe is 87 % busy -- the binding resource of the headline kernel.

Which slot a value lives in is free: entries of the KKT right-hand side may be placed in any order inside
their region (x entries / rows of each class, parameter-dependent ones first: the device ordering of
cvxpygen_amd/runtime.py is a permutation the host picks), and the spill slots of merged phases are
arbitrary.  This module picks the numbering by simulated annealing on the exact cost of the generated
executor's gathers: sum over (step, 32-lane group) of (max distinct slots on one bank pair - 1).

Code-generation time only (numpy); deterministic for a given seed.
"""

from __future__ import annotations

from typing import List, Tuple

import numpy as np

BANK_PAIRS = 32          # 64 banks x 4 B; an 8-byte slot covers one aligned pair
GROUP = 32               # lanes serviced together by a ds_read_b64


def gather_groups(step_slots: np.ndarray, lane_group=None) -> List[np.ndarray]:
    """step_slots [n_steps, 64]: slot gathered by every lane of every step (what the executor really
    reads, idle lanes included).  Returns the distinct slots of every (step, lane group): the two 32-lane halves of a
    ds_read_b64, or the groups lane_group[lane] names (B128_LANE_GROUP: the four 16-lane groups of a ds_read_b128)."""
    out = []
    for row in step_slots:
        if lane_group is None:
            for g in range(0, row.shape[0], GROUP):
                out.append(np.unique(row[g:g + GROUP]))
        else:
            lg = np.asarray(lane_group)
            for g in range(int(lg.max()) + 1):
                out.append(np.unique(row[lg == g]))
    return out


def conflict_cycles(groups: List[np.ndarray], pi: np.ndarray, mods=None) -> int:
    """extra LDS cycles of all gathers (and stores, modulus 16) under the numbering pi (slot -> position)"""
    tot = 0
    for gi, sl in enumerate(groups):
        m = BANK_PAIRS if mods is None else int(mods[gi])
        tot += int(np.bincount(pi[sl] % m, minlength=m).max()) - 1
    return tot


STORE_GROUP = 16         # lanes served together by a ds_write_b64 (four groups per wavefront) ...
STORE_BANK_PAIRS = 16    # ... over 32 four-byte banks: 8-byte slots collide modulo 16


def store_groups(out_slots: np.ndarray, no_row: int, group: int = 16) -> List[np.ndarray]:
    """out_slots [n_chunks, 64]: slot every lane of a chunk's reduce-store writes (no_row: none, the lane
    writes a dummy slot).  Returns the distinct real slots of every (chunk, 16-lane group) with more than one."""
    out = []
    for row in out_slots:
        for g in range(0, row.shape[0], group):
            sl = np.unique(row[g:g + group])
            sl = sl[sl != no_row]
            if len(sl) > 1:
                out.append(sl.astype(np.int64))
    return out


_SOURCE_VERSION = None


def _source_version() -> bytes:
    """digest of this module's source: part of the disk cache key of `optimise`, so that a change of the annealer
    (temperatures, move rule, best-seen logic) cannot silently reuse layouts -- and with them family-library
    fingerprints -- of the version before"""
    global _SOURCE_VERSION
    if _SOURCE_VERSION is None:
        import hashlib
        try:
            with open(__file__, 'rb') as f:
                _SOURCE_VERSION = hashlib.sha256(f.read()).digest()
        except OSError:
            _SOURCE_VERSION = b'unknown'
    return _SOURCE_VERSION


def optimise(step_slots: np.ndarray, region: np.ndarray, sweeps: int = 60, seed: int = 0,
             stores: List[np.ndarray] = ()) -> Tuple[np.ndarray, int, int]:
    """Returns (pi, cost before, cost after).  pi[slot] = new position; pi permutes the slots of every
    region among themselves (region[slot] = region id; regions are contiguous ranges of positions).
    stores: groups of slots written by one 16-lane group of a reduce-store (`store_groups`); they are charged
    with the store's banking (8-byte slots collide modulo 16), measured like the gathers' with
    scripts/micro/lds_conflicts.hip: a ds_write_b64 costs the sum over its four lane groups of the largest number
    of distinct slots on one bank pair."""
    n = region.shape[0]
    # the result is a pure function of the arguments: cached on disk (next to the generated family libraries), so
    # that constructing a solver for a family that was laid out before does not anneal again
    import hashlib
    import os
    hsh = hashlib.sha256()
    hsh.update(_source_version())               # the annealer itself: a cached layout of another version of this file is not reused
    for a_ in (np.ascontiguousarray(step_slots, dtype=np.int64), np.ascontiguousarray(region, dtype=np.int64),
               np.asarray([sweeps, seed, BANK_PAIRS, STORE_BANK_PAIRS], dtype=np.int64),
               *[np.ascontiguousarray(g_, dtype=np.int64) for g_ in stores]):
        hsh.update(a_.tobytes()); hsh.update(b'|')
    cdir = os.environ.get('CPG_LAYOUT_CACHE', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'generated', '.layout_cache'))
    cfile = os.path.join(cdir, hsh.hexdigest()[:32] + '.npy')
    if os.path.exists(cfile):
        try:
            rec = np.load(cfile)
            if rec.shape == (n + 2,):
                return rec[2:].astype(np.int64), int(rec[0]), int(rec[1])
        except (OSError, ValueError):
            pass
    rng = np.random.default_rng(seed)
    groups = gather_groups(step_slots)
    mods = [BANK_PAIRS] * len(groups) + [STORE_BANK_PAIRS] * len(stores)
    groups = groups + [np.asarray(g_, dtype=np.int64) for g_ in stores]
    mods = np.asarray(mods, dtype=np.int64)
    ng = len(groups)
    pi = np.arange(n, dtype=np.int64)
    # slot -> groups that contain it
    occ: List[List[int]] = [[] for _ in range(n)]
    for gi, sl in enumerate(groups):
        for s in sl:
            occ[int(s)].append(gi)
    occ_arr = [np.asarray(o, dtype=np.int64) for o in occ]
    cnt = np.zeros((ng, BANK_PAIRS), dtype=np.int32)
    for gi, sl in enumerate(groups):
        np.add.at(cnt[gi], pi[sl] % mods[gi], 1)
    gmax = cnt.max(axis=1)
    cost0 = int(gmax.sum() - ng)
    members = [np.nonzero(region == r)[0] for r in np.unique(region)]
    members = [m for m in members if len(m) > 1]
    used = np.array([len(o) > 0 for o in occ])
    cost = cost0
    best_cost, best_pi = cost0, pi.copy()        # the annealer wanders: what it returns is the best numbering it SAW
    n_moves = sweeps * int(used.sum())
    T0, T1 = 1.0, 0.05
    for it in range(n_moves):
        T = T0 * (T1 / T0) ** (it / max(1, n_moves - 1))
        m = members[int(rng.integers(len(members)))]
        a = int(m[int(rng.integers(len(m)))])
        b = int(m[int(rng.integers(len(m)))])
        if pi[a] % BANK_PAIRS == pi[b] % BANK_PAIRS or not (used[a] or used[b]):
            continue
        ga, gb = occ_arr[a], occ_arr[b]
        # groups that contain both are unaffected
        only_a = np.setdiff1d(ga, gb, assume_unique=True)
        only_b = np.setdiff1d(gb, ga, assume_unique=True)
        aff = np.concatenate([only_a, only_b])
        if not len(aff):
            pi[a], pi[b] = pi[b], pi[a]
            continue
        sub = cnt[aff].copy()
        na = len(only_a)
        ra, rb = pi[a] % mods[aff], pi[b] % mods[aff]      # residues under every affected group's banking
        ia, ib = np.arange(na), np.arange(na, len(aff))
        np.subtract.at(sub, (ia, ra[:na]), 1); np.add.at(sub, (ia, rb[:na]), 1)
        np.subtract.at(sub, (ib, rb[na:]), 1); np.add.at(sub, (ib, ra[na:]), 1)
        new_max = sub.max(axis=1)
        delta = int(new_max.sum() - gmax[aff].sum())
        if delta <= 0 or rng.random() < np.exp(-delta / T):
            cnt[aff] = sub
            gmax[aff] = new_max
            pi[a], pi[b] = pi[b], pi[a]
            cost += delta
            if cost < best_cost:
                best_cost, best_pi = cost, pi.copy()
    assert cost == conflict_cycles(groups, pi, mods)
    # never worse than the natural numbering (identity when nothing better was seen)
    assert best_cost == conflict_cycles(groups, best_pi, mods) and best_cost <= cost0
    try:
        os.makedirs(cdir, exist_ok=True)
        tmp = cfile + f'.{os.getpid()}.tmp.npy'
        np.save(tmp, np.concatenate([[cost0, best_cost], best_pi]).astype(np.int64))
        os.replace(tmp, cfile)
    except OSError:
        pass
    return best_pi, cost0, best_cost


# ------------------------------------------------------------------------------------------------
# Joint optimisation of the slot numbering AND of the order of a row's entries (round 6).
#
# The numbering alone leaves ~0.85 extra LDS cycles per 32-lane gather group on MPC 12/4/10 (252 modelled cycles per
# solve against 296 conflict-free gather cycles; SQ_LDS_BANK_CONFLICT = 26 % of SQ_LDS_IDX_ACTIVE with the LDS array
# 85 % busy, profiles/r5_final_pmc_config2.txt).  A second freedom was unused: WHICH of its (lane, step) cells an
# entry of a row occupies does not matter to the row's sum, so the entries of a row may be permuted among the row's
# cells.  With every lane free to choose the step at which it reads a given operand, a (chunk, 32-lane half) is a
# bipartite edge-colouring problem (lanes x bank pairs, colours = steps): conflict-free whenever no bank pair holds
# more distinct operands of the half than the chunk has steps -- which the numbering can arrange.  Both freedoms are
# annealed together on the exact cost (same model as `optimise`: calibrated with scripts/micro/lds_conflicts.hip).
def entry_cells(prog, idle_zero: bool, lane_group=None):
    """The (lane, step) cells of a RaggedProgram's flat entry array: per entry e its gather group (2 * step + half, steps
    numbered chunk by chunk as in `gathered_slots`), its row (unique per (chunk, row); pad cells of dummy lanes get a row of
    their own), its slot (cols // 8) and whether it is a pad (zero coefficient: any address will do).  Also the
    (group, slot) pairs of the idle lanes of partial steps when their offsets point at the zero slot (idle_zero)."""
    n_ent = prog.nnz - 1
    lane_group = np.arange(64) // GROUP if lane_group is None else np.asarray(lane_group, dtype=np.int64)
    NLG = int(lane_group.max()) + 1               # lane groups a gather instruction is served in
    grp = np.zeros(n_ent, dtype=np.int64)
    row = np.zeros(n_ent, dtype=np.int64)
    slot = (prog.cols[:n_ent].astype(np.int64)) // 8
    pad = np.asarray(prog.vals[:n_ent]) == 0.0
    fixed = []
    step0 = 0
    next_row = 0
    for c in range(prog.n_chunks):
        L, _, first, kind = (int(v) for v in prog.ctab[c])
        d = prog.desc[c]
        ln = (((d >> 16) & 0xFFF) if (kind & 1) else (d >> 16)).astype(np.int64)
        lr = prog.lane_row[c].astype(np.int64)
        ids = {}
        lane_rowid = np.zeros(64, dtype=np.int64)
        for t in range(64):
            key = int(lr[t]) if lr[t] >= 0 else -(t + 1)          # dummy lanes: a row each
            if key not in ids:
                ids[key] = next_row
                next_row += 1
            lane_rowid[t] = ids[key]
        e = first
        for s_ in range(L):
            cnt = int((ln > s_).sum())
            lanes = np.arange(cnt)
            grp[e:e + cnt] = NLG * (step0 + s_) + lane_group[lanes]
            row[e:e + cnt] = lane_rowid[:cnt]
            if idle_zero and cnt < 64:
                zs = prog.n_slots + 16          # GEN_DUMMY_SLOTS: the slot that always holds 0.0
                for h in sorted({int(v) for v in lane_group[cnt:]}):
                    fixed.append((NLG * (step0 + s_) + h, zs))
            e += cnt
        step0 += L
    return grp, row, slot, pad, fixed, NLG * step0


def fresh_entries(prog) -> np.ndarray:
    """per entry of the flat array: does it read a slot the PREVIOUS phase stores?  (The generated LDS executor issues the gathers
    of a step before the previous phase has stored when no lane of the step does: codegen.emit_program_header `early`.)"""
    n_ent = prog.nnz - 1
    slot = prog.cols[:n_ent].astype(np.int64) // 8
    fresh = np.zeros(n_ent, dtype=bool)
    plist = sorted({int(p_) for p_ in prog.chunk_phase})
    outs = {}
    for c in range(prog.n_chunks):
        o_ = (prog.desc[c] & 0xFFFF).astype(np.int64)
        outs.setdefault(int(prog.chunk_phase[c]), set()).update(int(x) for x in o_ if int(x) != 0xFFFF)
    for c in range(prog.n_chunks):
        k_ = plist.index(int(prog.chunk_phase[c]))
        if k_ == 0:
            continue
        prev = np.fromiter(outs[plist[k_ - 1]], dtype=np.int64)
        first = int(prog.ctab[c, 2])
        last = int(prog.ctab[c + 1, 2]) if c + 1 < prog.n_chunks else n_ent
        fresh[first:last] = np.isin(slot[first:last], prev)
    return fresh & (np.asarray(prog.vals[:n_ent]) != 0.0)


# lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS section): four groups of 16 lanes, one LDS cycle each; 16-byte slots
# collide modulo 16
B128_LANE_GROUP = np.zeros(64, dtype=np.int64)
for _g, _ranges in enumerate((((0, 4), (12, 16), (20, 28)), ((4, 12), (16, 20), (28, 32)), ((32, 36), (44, 48), (52, 60)), ((36, 44), (48, 52), (60, 64)))):
    for _a, _b in _ranges:
        B128_LANE_GROUP[_a:_b] = _g


def optimise_entries(prog, region: np.ndarray, idle_zero: bool, sweeps: int = 60, seed: int = 0,
                     stores: List[np.ndarray] = (), keep_fresh: bool = True, lane_group=None, mod: int = BANK_PAIRS,
                     store_mod: int = STORE_BANK_PAIRS, pi0=None, slot_moves: bool = True
                     ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int, int]:
    """Returns (pi, eperm, pad_slot, cost before, cost after) for a RaggedProgram packed with the NATURAL numbering:
    pi[slot] = new position (as `optimise`); eperm: the entry that moves to flat position e is the old entry eperm[e]
    (entries only move between cells of their own row); pad_slot[e] >= 0: the (zero-coefficient) entry at position e
    reads this OLD slot -- an address another lane of its gather group reads anyway (a broadcast costs nothing).
    Cost: extra LDS cycles of all gathers and reduce-stores per run of the program."""
    import hashlib
    import os
    n = region.shape[0]
    grp, row, slot0, pad, fixed, n_groups = entry_cells(prog, idle_zero, lane_group)
    n_ent = len(grp)
    # keep_fresh: an entry that reads what the previous phase stores only trades places with another one that does -- the steps
    # whose gathers the LDS executor may issue early stay what they are
    cls_ = fresh_entries(prog).astype(np.int64) if keep_fresh else np.zeros(n_ent, dtype=np.int64)
    row = row * 2 + cls_
    hsh = hashlib.sha256()
    hsh.update(_source_version())
    for a_ in (grp, row, slot0, pad.astype(np.int64), np.asarray(fixed, dtype=np.int64).reshape(-1),
               np.ascontiguousarray(region, dtype=np.int64), np.asarray([sweeps, seed, mod, store_mod, 7, int(keep_fresh), int(slot_moves)], dtype=np.int64), np.arange(n) if pi0 is None else np.asarray(pi0, dtype=np.int64), np.asarray([1e6 * float(os.environ.get(k_, v_)) for k_, v_ in (('CPG_ANNEAL_T0', 0.4), ('CPG_ANNEAL_T1', 0.05), ('CPG_ANNEAL_PENTRY', 0.6), ('CPG_ANNEAL_TENTRY', 0.3))], dtype=np.int64),
               *[np.ascontiguousarray(g_, dtype=np.int64) for g_ in stores]):
        hsh.update(np.ascontiguousarray(a_, dtype=np.int64).tobytes()); hsh.update(b'|')
    cdir = os.environ.get('CPG_LAYOUT_CACHE', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'generated', '.layout_cache'))
    cfile = os.path.join(cdir, 'e' + hsh.hexdigest()[:31] + '.npz')
    if os.path.exists(cfile):
        try:
            rec = np.load(cfile)
            if rec['pi'].shape == (n,) and rec['eperm'].shape == (n_ent,):
                return rec['pi'], rec['eperm'], rec['pad_slot'], int(rec['cost'][0]), int(rec['cost'][1])
        except (OSError, ValueError, KeyError):
            pass
    import random
    rnd = random.Random(seed)
    NG = n_groups
    MODG = int(mod)
    SMOD = int(store_mod)
    # ---- state (plain Python containers: the moves touch a handful of small integers each)
    pos = list(range(n)) if pi0 is None else [int(v) for v in pi0]        # slot -> position
    cell_slot = [(-1 if pad[e] else int(slot0[e])) for e in range(n_ent)]      # what the cell holds now (-1: a pad)
    cell_src = list(range(n_ent))                          # ... and which old entry that is
    cell_grp = [int(g_) for g_ in grp]
    mult = [dict() for _ in range(NG)]                     # gather group -> {slot: multiplicity}
    fixed_in = [set() for _ in range(NG)]
    for g_, s_ in fixed:
        fixed_in[g_].add(int(s_))
    cnt = [[0] * MODG for _ in range(NG)]                  # distinct slots per bank pair
    occ = [dict() for _ in range(n)]                       # slot -> {gather group: multiplicity}

    def add(g_, x):
        m_ = mult[g_]
        k_ = m_.get(x, 0)
        m_[x] = k_ + 1
        if k_ == 0:
            cnt[g_][pos[x] % MODG] += 1
            occ[x][g_] = 1

    def rem(g_, x):
        m_ = mult[g_]
        k_ = m_[x] - 1
        if k_ == 0:
            del m_[x]
            cnt[g_][pos[x] % MODG] -= 1
            del occ[x][g_]
        else:
            m_[x] = k_
    for g_ in range(NG):
        for x in fixed_in[g_]:
            add(g_, x); add(g_, x)                         # (multiplicity 2: a fixed member never leaves)
    for e in range(n_ent):
        if cell_slot[e] >= 0:
            add(cell_grp[e], cell_slot[e])
    gmax = [max(c_) if any(c_) else 1 for c_ in cnt]
    # store groups: static membership, modulus 16
    st_groups = [[int(v) for v in g_] for g_ in stores]
    st_of = [[] for _ in range(n)]
    for k_, g_ in enumerate(st_groups):
        for x in g_:
            st_of[x].append(k_)
    st_cnt = []
    for g_ in st_groups:
        c_ = [0] * SMOD
        for x in g_:
            c_[pos[x] % SMOD] += 1
        st_cnt.append(c_)
    st_max = [max(c_) for c_ in st_cnt]
    cost = sum(gmax) - NG + sum(st_max) - len(st_groups)
    cost0 = cost
    # rows with more than one cell in different groups can move entries; slots of a region can swap
    by_row = {}
    for e in range(n_ent):
        by_row.setdefault(int(row[e]), []).append(e)
    rows_mv = [v for v in by_row.values() if len({cell_grp[e] for e in v}) > 1]
    row_of_cell = {}
    for v in rows_mv:
        for e in v:
            row_of_cell[e] = v
    cells_of_grp = [[] for _ in range(NG)]
    for e in range(n_ent):
        if e in row_of_cell:
            cells_of_grp[cell_grp[e]].append(e)
    members = [np.nonzero(region == r)[0] for r in np.unique(region)]
    members = [[int(v) for v in m_] for m_ in members if len(m_) > 1]
    used = [bool(occ[x]) or bool(st_of[x]) for x in range(n)]
    n_used = sum(used)
    best = (cost, list(pos), list(cell_src), list(cell_slot))
    n_moves = sweeps * (n_used + n_ent // 2)
    T0, T1 = float(os.environ.get('CPG_ANNEAL_T0', 0.4)), float(os.environ.get('CPG_ANNEAL_T1', 0.05))
    P_ENTRY = float(os.environ.get('CPG_ANNEAL_PENTRY', 0.6))
    T_ENTRY = float(os.environ.get('CPG_ANNEAL_TENTRY', 0.3))
    import math
    hot = set(g_ for g_ in range(NG) if gmax[g_] > 1)
    hot_list, hot_at = [], 0
    for it in range(n_moves):
        T = T0 * (T1 / T0) ** (it / max(1, n_moves - 1))
        if rows_mv and (not slot_moves or (T < T_ENTRY and rnd.random() < P_ENTRY)):
            # ---- entry move: two cells of one row exchange what they hold
            if hot and rnd.random() < 0.7:
                if it >= hot_at:
                    hot_list = sorted(hot)
                    hot_at = it + 64
                g1 = rnd.choice(hot_list) if hot_list else 0
                if gmax[g1] <= 1:
                    continue
                cl_ = cells_of_grp[g1]
                if not cl_:
                    hot.discard(g1)
                    continue
                # an entry of the most loaded bank pair of a conflicted group
                bm = cnt[g1].index(gmax[g1])
                cand = [e for e in cl_ if cell_slot[e] >= 0 and pos[cell_slot[e]] % MODG == bm]
                e1 = rnd.choice(cand) if cand else rnd.choice(cl_)
            else:
                e1 = rnd.choice(rnd.choice(rows_mv))
            e2 = rnd.choice(row_of_cell[e1])
            g1, g2 = cell_grp[e1], cell_grp[e2]
            x1, x2 = cell_slot[e1], cell_slot[e2]
            if g1 == g2 or x1 == x2:
                continue
            if x1 >= 0:
                rem(g1, x1)
            if x2 >= 0:
                rem(g2, x2)
            if x1 >= 0:
                add(g2, x1)
            if x2 >= 0:
                add(g1, x2)
            m1, m2 = max(cnt[g1]), max(cnt[g2])
            m1, m2 = max(m1, 1), max(m2, 1)
            delta = m1 + m2 - gmax[g1] - gmax[g2]
            if delta <= 0 or rnd.random() < math.exp(-delta / T):
                gmax[g1], gmax[g2] = m1, m2
                cell_slot[e1], cell_slot[e2] = x2, x1
                cell_src[e1], cell_src[e2] = cell_src[e2], cell_src[e1]
                cost += delta
                for g_, m_ in ((g1, m1), (g2, m2)):
                    if m_ > 1:
                        hot.add(g_)
                    else:
                        hot.discard(g_)
            else:
                if x1 >= 0:
                    rem(g2, x1)
                if x2 >= 0:
                    rem(g1, x2)
                if x1 >= 0:
                    add(g1, x1)
                if x2 >= 0:
                    add(g2, x2)
        else:
            # ---- slot move: two slots of one region exchange their positions
            if not slot_moves or not members:
                continue
            m_ = rnd.choice(members)
            a, b = rnd.choice(m_), rnd.choice(m_)
            pa, pb = pos[a], pos[b]
            if (pa % MODG == pb % MODG and pa % SMOD == pb % SMOD) or not (used[a] or used[b]):
                continue
            oa, ob = occ[a], occ[b]
            aff = [g_ for g_ in oa if g_ not in ob]
            affb = [g_ for g_ in ob if g_ not in oa]
            ca, cb = pa % MODG, pb % MODG
            delta = 0
            newmax = []
            for g_ in aff:
                c_ = cnt[g_]
                c_[ca] -= 1; c_[cb] += 1
                mm = max(c_)
                newmax.append(mm)
                delta += mm - gmax[g_]
            for g_ in affb:
                c_ = cnt[g_]
                c_[cb] -= 1; c_[ca] += 1
                mm = max(c_)
                newmax.append(mm)
                delta += mm - gmax[g_]
            sa, sb = pa % SMOD, pb % SMOD
            saff, sbff, snew = [], [], []
            if sa != sb:
                sta, stb = st_of[a], st_of[b]
                saff = [k_ for k_ in sta if k_ not in stb]
                sbff = [k_ for k_ in stb if k_ not in sta]
                for k_ in saff:
                    c_ = st_cnt[k_]
                    c_[sa] -= 1; c_[sb] += 1
                    mm = max(c_)
                    snew.append(mm)
                    delta += mm - st_max[k_]
                for k_ in sbff:
                    c_ = st_cnt[k_]
                    c_[sb] -= 1; c_[sa] += 1
                    mm = max(c_)
                    snew.append(mm)
                    delta += mm - st_max[k_]
            if delta <= 0 or rnd.random() < math.exp(-delta / T):
                pos[a], pos[b] = pb, pa
                for g_, mm in zip(aff + affb, newmax):
                    gmax[g_] = mm
                    if mm > 1:
                        hot.add(g_)
                    else:
                        hot.discard(g_)
                for k_, mm in zip(saff + sbff, snew):
                    st_max[k_] = mm
                cost += delta
            else:
                for g_ in aff:
                    c_ = cnt[g_]
                    c_[ca] += 1; c_[cb] -= 1
                for g_ in affb:
                    c_ = cnt[g_]
                    c_[cb] += 1; c_[ca] -= 1
                for k_ in saff:
                    c_ = st_cnt[k_]
                    c_[sa] += 1; c_[sb] -= 1
                for k_ in sbff:
                    c_ = st_cnt[k_]
                    c_[sb] += 1; c_[sa] -= 1
        if cost < best[0]:
            best = (cost, list(pos), list(cell_src), list(cell_slot))
    cost1, pos_b, src_b, slot_b = best
    pi = np.asarray(pos_b, dtype=np.int64)
    eperm = np.asarray(src_b, dtype=np.int64)
    # pads: an address some other lane of the gather group reads anyway (the group's first real entry; a group of pads only:
    # its first fixed member, else slot 0)
    pad_slot = np.full(n_ent, -1, dtype=np.int64)
    first_real = {}
    for e in range(n_ent):
        if slot_b[e] >= 0 and cell_grp[e] not in first_real:
            first_real[cell_grp[e]] = slot_b[e]
    for e in range(n_ent):
        if slot_b[e] < 0:
            g_ = cell_grp[e]
            pad_slot[e] = first_real.get(g_, min(fixed_in[g_]) if fixed_in[g_] else 0)
    assert sorted(eperm.tolist()) == list(range(n_ent)) and sorted(pi.tolist()) == list(range(n))
    try:
        os.makedirs(cdir, exist_ok=True)
        tmp = cfile + f'.{os.getpid()}.tmp.npz'
        np.savez(tmp, pi=pi, eperm=eperm, pad_slot=pad_slot, cost=np.asarray([cost0, cost1], dtype=np.int64))
        os.replace(tmp, cfile)
    except OSError:
        pass
    return pi, eperm, pad_slot, int(cost0), int(cost1)


def apply_entries(prog, eperm: np.ndarray, pad_slot: np.ndarray, pi=None):
    """The RaggedProgram with its entries permuted (`optimise_entries`): vals / cols of flat position e come from old entry
    eperm[e]; pads read pad_slot[e] (an OLD slot number: mapped through pi when the program was packed with it)."""
    import dataclasses
    n_ent = prog.nnz - 1
    vals = prog.vals.copy()
    cols = prog.cols.copy()
    vals[:n_ent] = prog.vals[eperm]
    cols[:n_ent] = prog.cols[eperm]
    isp = pad_slot >= 0
    assert (vals[:n_ent][isp] == 0.0).all() and (vals[:n_ent][~isp] != 0.0).all()
    ps = pad_slot[isp]
    if pi is not None:
        pi = np.asarray(pi, dtype=np.int64)
        ps = np.where(ps < len(pi), pi[np.minimum(ps, len(pi) - 1)], ps)
    c2 = cols[:n_ent].astype(np.int64)
    c2[isp] = 8 * ps
    cols[:n_ent] = c2.astype(cols.dtype)
    return dataclasses.replace(prog, vals=vals, cols=cols)
