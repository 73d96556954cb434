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


def gather_groups(step_slots: np.ndarray) -> List[np.ndarray]:
    """step_slots [n_steps, 64]: slot gathered by every lane of every step (what the executor really
    reads, idle lanes included).  Returns the distinct slots of every (step, 32-lane group)."""
    out = []
    for row in step_slots:
        for g in range(0, row.shape[0], GROUP):
            out.append(np.unique(row[g:g + GROUP]))
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


def store_groups(out_slots: np.ndarray, no_row: int) -> List[np.ndarray]:
    """out_slots [n_chunks, 64]: slot every lane of a chunk's reduce-store writes (no_row: none, the lane
    writes a dummy slot).  Returns the distinct real slots of every (chunk, 16-lane group) with more than one."""
    out = []
    for row in out_slots:
        for g in range(0, row.shape[0], STORE_GROUP):
            sl = np.unique(row[g:g + STORE_GROUP])
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
