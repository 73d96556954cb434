"""Team per-instance factor kernel (csrc/cpg_osqp_team.h): the path of the resident kernel -- `osqp_update_data_mat` +
`osqp_solve` per instance (cvxpygen/solvers/osqp.py:20-62) -- with one WORKGROUP of W wavefronts per instance, for families
whose merged substitution program does not fit one wavefront's registers (MPC 12/4/10 with every parameter per instance).

CPU tier: the team plan's algebra (programs planned for W wavefronts, products with the block inverses out of place), the
wavefront assignment, and the kernel SOURCES on the lock-step emulator against the C oracle.  GPU tier: the all-parameters
MPC 12/4/10 family through the C-ABI against the oracle."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

from cvxpygen_amd import codegen, families, resident_plan as rs, solve_program as spm
from cvxpygen_amd.runtime import BatchSolver, build_family_plan

from test_resident import _dense_kkt, _portfolio_values, _random_values


def _team_in_use(bs) -> bool:
    v = C.c_double(-1)
    bs.lib.check(bs.lib.L.cpg_hip_get_setting(bs.h_ref, b'team_executor', C.byref(v)), 'cpg_hip_get_setting')
    return v.value > 0.0            # (the width W of the team)


def test_factor_batch_size_follows_the_schedule(monkeypatch):
    """steps per batch of the factorisation table walk (codegen.team_fac_batch -> CPG_GENT_FAC_BATCH): a chunk of L steps costs
    ceil(L / S) batches of S step slots each, padding included -- chains of 6- and 12-step levels want S = 6 (what the all-parameters
    MPC 12/4/10 family has: 143 -> 123 us of LDL' chain on MI355X, profiles/r6_t3_*), long chunks S = 8, 4-step levels S = 4;
    an even number always (two steps share three table words); the environment overrides"""
    import types
    monkeypatch.delenv('CPG_TEAM_FAC_BATCH', raising=False)
    mk = lambda lens: types.SimpleNamespace(f_ctab=np.array([[L, 1, 0, 0] for L in lens], dtype=np.int32))
    assert codegen.team_fac_batch(mk([6] * 72 + [12] * 74 + [5] * 36 + [7] * 31 + [4] * 13)) == 6
    assert codegen.team_fac_batch(mk([64, 48, 40, 33] * 20)) == 8
    assert codegen.team_fac_batch(mk([4, 3, 4, 2] * 50)) == 4
    assert codegen.team_fac_batch(mk([0, 0, 6])) in (4, 6, 8)          # (empty chunks still cost a batch)
    monkeypatch.setenv('CPG_TEAM_FAC_BATCH', '8')
    assert codegen.team_fac_batch(mk([6] * 10)) == 8


@pytest.mark.parametrize('fam', ['portfolio', 'mpc6'])
@pytest.mark.parametrize('W', [1, 2, 4])
def test_team_plan_solves_the_kkt_system(fam, W):
    """the programs planned for a team: no deferred (in-place) phases -- the products with the inverses of merged diagonal
    blocks write to slots of their own and the solution still comes out in place --, every chunk belongs to one wavefront,
    the busiest wavefront has fewer steps than the single-wavefront plan, and the program == a dense solve"""
    d = families.portfolio(30, 4) if fam == 'portfolio' else families.mpc(6, 3, 10)
    plan = build_family_plan(d, bank_layout=False)
    pl1 = rs.build_resident_plan(d.P, d.A, plan.osqp)
    pl = rs.build_resident_plan(d.P, d.A, plan.osqp, team=W, inplace_x=False)
    b = pl.base
    N = b.n + b.m
    assert pl.team == W and pl.nnzX == pl1.nnzX and pl.groups == pl1.groups
    assert not (pl.sol.ctab[:, 3] & 4).any() and (pl1.sol.ctab[:, 3] & 4).any()
    assert np.array_equal(pl.sol.final_pos, np.arange(N)) and pl.sol.n_slots > N
    steps, sw, sl, cw, cl = codegen.team_steps(pl.sol, W)
    assert len(steps) == int(pl.sol.ctab[:, 0].sum())
    for v in range(W):
        mine = np.nonzero(sw == v)[0]
        assert np.array_equal(np.sort(sl[mine]), np.arange(len(mine)))          # a wavefront's steps are numbered 0 ..
        assert all(cw[steps[t][1]] == v for t in mine)
    per_phase = {}
    for t, (p, c, e, cnt) in enumerate(steps):
        per_phase.setdefault(p, np.zeros(W, dtype=int))[sw[t]] += 1
    if W > 1:
        assert sum(int(v.max()) for v in per_phase.values()) < int(pl1.sol.ctab[:, 0].sum())
    rng, Ps, As, rho_inv = _random_values(b, d.n_eq)
    fac = rs.replay_factor(pl, Ps, As, 1e-6, rho_inv)
    _, _, K = _dense_kkt(b, Ps, As, 1e-6, rho_inv)
    rhs = rng.standard_normal(N)
    w = np.zeros(pl.sol.n_slots); w[:N] = rhs
    w = spm.execute_ragged(dataclasses.replace(pl.sol, vals=rs.replay_solve_vals(pl, fac)), w)
    xr = np.linalg.solve(K, rhs)
    assert np.abs(w[:N] - xr).max() <= 1e-9 * np.abs(xr).max()


@pytest.mark.parametrize('W,gen_fac', [(1, True), (2, True), (4, True), (4, False), (2, False)])
def test_team_kernel_on_the_emulator_vs_oracle(oracle_lib, tmp_path, monkeypatch, W, gen_fac):
    """the kernel sources of the team path (set-up on 64 W threads, factorisation on wavefront 0, per-wavefront coefficient /
    offset / slot registers, one barrier per phase, team reductions of the termination test) in a family library of a small
    portfolio family: the oracle's iterates, iteration counts and statuses in the default mode (rho adapted at 50, 100 ...),
    at a cut-off between two tests (approximate second test) and with tight tolerances; the streaming kernel of the same
    library gives the same counts"""
    from sim import build_sim
    from test_sim_kernel import _assert_parity, _oracle_flat
    monkeypatch.setattr(codegen, 'TEAM_WAVES', W)
    # (gen_fac False: the factorisation as the team's batched table walk -- LDL' part on wavefront 0, the levels of the block
    # inverses spread over the team -- what families with long schedules get; True: straight-line code on wavefront 0)
    monkeypatch.setattr(codegen, 'TEAM_FAC_GENERATED_MAX', 600 if gen_fac else 0)
    n, m, B = 20, 3, 3
    d = families.portfolio(n, m)
    plan = build_family_plan(d)
    _, defs = codegen.family_library_defs(plan, str(tmp_path), 'pf20')
    assert any('CPG_GENT_HEADER' in x for x in defs) and not any('CPG_GENR_HEADER' in x for x in defs)
    assert ('CPG_GENT_FAC_GENERATED' in open(str(tmp_path / 'cpg_team_pf20.h')).read()) == gen_fac
    lib = build_sim.build_family(plan, str(tmp_path), 'pf20')
    vals, th, upd = _portfolio_values(d, B, n, m)
    bs = BatchSolver(d, lib_path=lib, plan=plan)
    for stg in ({}, dict(max_iter=60), dict(eps_abs=1e-7, eps_rel=1e-7)):
        r = bs.solve(vals, updated_params=upd, **stg)
        assert _team_in_use(bs)
        o, prim, dual = _oracle_flat(oracle_lib, d, th, upd, **stg)
        _assert_parity(r, o, prim, dual, tol=1e-8)
    bs.set_program_placement(0)                       # the streaming kernel (unmerged program from HBM)
    r2 = bs.solve(vals, updated_params=upd, eps_abs=1e-7, eps_rel=1e-7)
    assert not _team_in_use(bs)
    assert r2.iter.tolist() == r.iter.tolist() and np.abs(r2.prim_flat - r.prim_flat).max() < 1e-8
    bs.close()


def test_team_kernel_all_parameters_mpc_on_the_emulator(oracle_lib, tmp_path, monkeypatch):
    """MPC 6/3/10 with EVERY parameter per instance (the shape of the family the kernel was built for: a chain of levels,
    several merged groups) on the emulator, a team of four wavefronts: iteration counts / statuses = oracle, 1e-8"""
    from sim import build_sim
    from test_sim_kernel import _assert_parity, _oracle_flat
    monkeypatch.setattr(codegen, 'TEAM_WAVES', 4)
    monkeypatch.setattr(codegen, 'TEAM_FAC_GENERATED_MAX', 0)          # (the batched factorisation: several merged groups to invert)
    d = families.mpc(6, 3, 10)
    plan = build_family_plan(d)
    _, defs = codegen.family_library_defs(plan, str(tmp_path), 'mpc6t')
    assert any('CPG_GENT_HEADER' in x for x in defs)
    lib = build_sim.build_family(plan, str(tmp_path), 'mpc6t')
    B = 2
    rng = np.random.default_rng(17)
    th = np.tile(d.theta0, (B, 1))
    th[:, :d.NP] *= 1 + 0.1 * rng.standard_normal((B, d.NP))
    p = d.param('x_init')
    th[:, p.col:p.col + p.size] = -2 + 4 * rng.random((B, p.size))
    vals = {q.name: th[:, q.col:q.col + q.size] for q in d.params}
    bs = BatchSolver(d, lib_path=lib, plan=plan)
    r = bs.solve(vals)
    assert _team_in_use(bs)
    o, prim, dual = _oracle_flat(oracle_lib, d, th, None)
    _assert_parity(r, o, prim, dual, tol=1e-8)
    bs.close()


@pytest.mark.gpu
def test_all_parameters_mpc12_runs_the_team_kernel_vs_oracle(oracle_lib):
    """MPC 12/4/10 with EVERY parameter per instance (sub-mode ii) in its family library on the GPU: the
    team kernel is what runs (the family's merged program does not fit the resident kernel's single wavefront), iteration
    counts / statuses = oracle, prim / dual within 1e-6 -- default mode, a cut-off between two tests, and against the streaming
    kernel of the same library"""
    import os
    from test_sim_kernel import _assert_parity, _oracle_flat
    d = families.mpc(12, 4, 10)
    plan = build_family_plan(d)
    out = os.path.join(os.path.dirname(os.path.abspath(codegen.__file__)), 'generated', 'mpc12')
    lib = codegen.build_family_library(plan, out, 'mpc12')
    B = 96
    rng = np.random.default_rng(23)
    th = np.tile(d.theta0, (B, 1))
    th[:, :d.NP] *= 1 + 0.05 * rng.standard_normal((B, d.NP))
    p = d.param('x_init')
    th[:, p.col:p.col + p.size] = -2 + 4 * rng.random((B, p.size))
    vals = {q.name: th[:, q.col:q.col + q.size] for q in d.params}
    bs = BatchSolver(d, lib_path=lib, plan=plan)
    for stg in ({}, dict(max_iter=60)):
        r = bs.solve(vals, **stg)
        assert _team_in_use(bs)
        o, prim, dual = _oracle_flat(oracle_lib, d, th, None, **stg)
        _assert_parity(r, o, prim, dual, tol=1e-6)
    bs.set_program_placement(0)
    r2 = bs.solve(vals, max_iter=60)
    assert not _team_in_use(bs)
    assert r2.iter.tolist() == r.iter.tolist() and np.abs(r2.prim_flat - r.prim_flat).max() < 1e-8
    bs.close()


@pytest.mark.gpu
def test_all_parameters_full_batch_is_its_distinct_instances_repeated(oracle_lib):
    """the all-parameters batch at the size the bench line is quoted on (20 000): 50 distinct instances repeated -- every copy
    comes back bit-identical to the first (teams share nothing but read-only tables and pull instances in an order that depends on
    the run) and the 50 equal the oracle's (counts exact, 1e-6)"""
    import os
    from test_sim_kernel import _assert_parity, _oracle_flat
    d = families.mpc(12, 4, 10)
    plan = build_family_plan(d)
    out = os.path.join(os.path.dirname(os.path.abspath(codegen.__file__)), 'generated', 'mpc12')
    lib = codegen.build_family_library(plan, out, 'mpc12')
    K, B = 50, 20_000
    rng = np.random.default_rng(29)
    th = np.tile(d.theta0, (K, 1))
    th[:, :d.NP] *= 1 + 0.05 * rng.standard_normal((K, d.NP))
    p = d.param('x_init')
    th[:, p.col:p.col + p.size] = -2 + 4 * rng.random((K, p.size))
    bs = BatchSolver(d, lib_path=lib, plan=plan)
    bs.set_updated(None)
    tv = th[:, bs._var_cols]
    tv_full = np.ascontiguousarray(np.tile(tv, (B // K, 1)))
    r = bs.solve(theta_var=tv_full, B=B)
    assert _team_in_use(bs)
    o, prim, dual = _oracle_flat(oracle_lib, d, th, None)
    first = type('R', (), dict(iter=r.iter[:K], status=r.status[:K], prim_flat=r.prim_flat[:K], dual_flat=r.dual_flat[:K],
                               obj_val=r.obj_val[:K], pri_res=r.pri_res[:K], dua_res=r.dua_res[:K]))()
    _assert_parity(first, o, prim, dual, tol=1e-6)
    idx = np.arange(B) % K
    assert np.array_equal(r.iter, r.iter[:K][idx]) and np.array_equal(r.status, r.status[:K][idx])
    assert np.array_equal(r.prim_flat, r.prim_flat[:K][idx]) and np.array_equal(r.dual_flat, r.dual_flat[:K][idx])
    bs.close()
