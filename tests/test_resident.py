"""Resident per-instance factor kernel (csrc/cpg_osqp_resident.h, cvxpygen_amd/resident_plan.py): the path of families whose
parameters enter P or A -- `osqp_update_data_mat` + `osqp_solve` per instance (cvxpygen/solvers/osqp.py:20-62).

CPU tier: the plan's algebra against dense linear algebra (merged levels, numeric block inverses through the combined
schedule, row programs of the termination test's products), and the kernel SOURCES on the lock-step emulator against the
C oracle.  GPU tier: BASELINE config 3's family through the C-ABI against the oracle."""
import ctypes as C
import dataclasses

import numpy as np
import pytest
import scipy.sparse as sp

from cvxpygen_amd import codegen, families, resident_plan as rs, solve_program as spm
from cvxpygen_amd.runtime import BatchSolver, build_family_plan


def _dense_kkt(b, Ps, As, sigma, rho_inv):
    n, m = b.n, b.m
    pr = b.Pi; pc = np.repeat(np.arange(n), np.diff(b.Pp))
    Pm = sp.csc_matrix((Ps, (pr, pc)), shape=(n, n)).toarray(); Pm = Pm + np.triu(Pm, 1).T
    Am = sp.csc_matrix((As, b.Ai, b.Ap), shape=(m, n)).toarray()
    return Pm, Am, np.block([[Pm + sigma * np.eye(n), Am.T], [Am, -np.diag(rho_inv)]])


def _random_values(b, n_eq, seed=0):
    rng = np.random.default_rng(seed)
    pr = b.Pi; pc = np.repeat(np.arange(b.n), np.diff(b.Pp))
    Ps = rng.standard_normal(b.nnzP) * 0.1
    Ps = np.where(pr == pc, np.abs(Ps) + 0.5, Ps * 0.01)
    As = rng.standard_normal(b.nnzA)
    rho_inv = 1.0 / np.where(np.arange(b.m) < n_eq, 100.0, 0.1)
    return rng, Ps, As, rho_inv


@pytest.mark.parametrize('fam', ['portfolio', 'mpc6'])
def test_merged_plan_solves_the_kkt_system(fam):
    """merged level groups + numeric inverses of their diagonal blocks (combined schedule) == dense solve; the three
    row programs == the dense products.  portfolio: a dense trailing block (one merged group); MPC: a chain (several)."""
    d = families.portfolio(30, 4) if fam == 'portfolio' else families.mpc(6, 3, 10)
    plan = build_family_plan(d, bank_layout=False)
    pl = rs.build_resident_plan(d.P, d.A, plan.osqp)
    b = pl.base
    N = b.n + b.m
    assert any(a != c for a, c in pl.groups) and pl.nnzX > 0
    assert pl.sol.n_phases < b.sol.n_phases                       # fewer dependent phases than plain level scheduling
    assert np.array_equal(pl.sol.final_pos, np.arange(N))
    rng, Ps, As, rho_inv = _random_values(b, d.n_eq)
    sigma = 1e-6
    fac = rs.replay_factor(pl, Ps, As, sigma, rho_inv)
    Pm, Am, K = _dense_kkt(b, Ps, As, sigma, rho_inv)
    rhs = rng.standard_normal(N)
    w = np.zeros(pl.sol.n_slots); w[:N] = rhs
    w = spm.execute_ragged(dataclasses.replace(pl.sol, vals=rs.replay_solve_vals(pl, fac)), w)
    xr = np.linalg.solve(K, rhs)
    assert np.abs(w[pl.sol.final_pos] - xr).max() <= 1e-9 * np.abs(xr).max()
    # the inverses themselves: X = L_GG^-1 with L rebuilt from the M-form factor
    nnzL = b.nnzL
    L = sp.csc_matrix((fac[:nnzL] * np.repeat(fac[nnzL:nnzL + N], np.diff(b.Lp)), b.Li, b.Lp), shape=(N, N)).toarray() + np.eye(N)
    lev = rs._levels(N, b.Lp.astype(np.int64), b.Li.astype(np.int64))
    for a, c in pl.groups:
        if a == c:
            continue
        G = np.nonzero((lev >= a) & (lev <= c))[0]
        Xd = np.linalg.inv(L[np.ix_(G, G)])
        pos = {int(r): t for t, r in enumerate(G)}
        sel = [k for k in range(pl.nnzX) if int(pl.x_row[k]) in pos]
        got = fac[nnzL + N + np.array(sel)]
        want = np.array([Xd[pos[int(pl.x_row[k])], pos[int(pl.x_col[k])]] for k in sel])
        assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    xx, yy = rng.standard_normal(b.n), rng.standard_normal(b.m)
    ww = np.zeros(pl.w_slots); ww[:b.n] = xx; ww[b.n:N] = yy
    ww = rs.replay_product(pl.rows_A, pl.rows_A_ent, As, ww)
    assert np.abs(ww[pl.out_ax:pl.out_ax + b.m] - Am @ xx).max() < 1e-12
    ww[pl.out_ax:pl.out_ax + max(b.m, 2 * b.n)] = 0.0            # (A x shares the slots of P x | A' y)
    ww = rs.replay_product(pl.rows_P, pl.rows_P_ent, Ps, ww)
    ww = rs.replay_product(pl.rows_At, pl.rows_At_ent, As, ww)
    assert np.abs(ww[pl.out_px:pl.out_px + b.n] - Pm @ xx).max() < 1e-12
    assert np.abs(ww[pl.out_aty:pl.out_aty + b.n] - Am.T @ yy).max() < 1e-12


def test_deferred_phase_reads_the_values_from_before_the_phase():
    """a phase whose rows read each other's slots (in-place product with a merged group's inverse): execute_ragged holds
    its stores back until the phase is complete, chunk by chunk"""
    rows = np.arange(70)                               # two chunks
    cols = [np.array([(r + 1) % 70]) for r in rows]
    vals = [np.array([2.0]) for _ in rows]
    ph = spm.Phase(rows, cols, vals, False, 'X', accumulate=True, deferred=True)
    prog = spm.pack_ragged([ph], 70, balanced='auto')
    assert (prog.ctab[:, 3] & 4).all()
    w0 = np.arange(70, dtype=float)
    w = spm.execute_ragged(prog, w0.copy())
    assert np.array_equal(w, w0 + 2.0 * np.roll(w0, -1))


def _portfolio_values(d, B, n, m, seed=5):
    rng = np.random.default_rng(seed)
    sig = np.zeros((B, m, m)); sig[:, np.arange(m), np.arange(m)] = rng.random((B, m))
    vals = {'a': rng.standard_normal((B, n)), 'F': np.round(rng.standard_normal((B, n, m))), 'Sig_f_sqrt': sig,
            'd_sqrt': rng.random((B, n)), 'w_prev': np.zeros((B, n))}
    th = np.stack([d.theta_from_values({k: v[i] for k, v in vals.items()}) for i in range(B)])
    return vals, th, ['a', 'F', 'Sig_f_sqrt', 'd_sqrt', 'w_prev']


def _resident_in_use(bs) -> bool:
    v = C.c_double(-1)
    bs.lib.check(bs.lib.L.cpg_hip_get_setting(bs.h_ref, b'resident_executor', C.byref(v)), 'cpg_hip_get_setting')
    return v.value == 1.0


@pytest.mark.parametrize('four_waves', [False, True])
def test_resident_kernel_on_the_emulator_vs_oracle(oracle_lib, tmp_path, monkeypatch, four_waves):
    """the kernel sources of the resident path (set-up in LDS, combined factorisation + inverse stream, merged program
    through the generated executor, termination test through the streamed row programs) in a family library of a small
    portfolio family: the oracle's iterates, iteration counts and statuses in the default mode (rho adapted at 50, 100 ...),
    at a cut-off between two tests (approximate second test) and with tight tolerances; the streaming kernel of the same
    library gives the same counts"""
    from sim import build_sim
    from test_sim_kernel import _assert_parity, _oracle_flat
    # (four_waves: the layout with the executor's tables in global memory and four slices per CU -- correct, slower on the GPU,
    # off by default: codegen.resident_four_waves)
    monkeypatch.setenv('CPG_RES_FOUR_WAVES', '1' if four_waves else '0')
    n, m, B = 20, 3, 3
    d = families.portfolio(n, m)
    plan = build_family_plan(d)
    _, defs = codegen.family_library_defs(plan, str(tmp_path), 'pf20')
    assert any('CPG_GENR_HEADER' in x for x in defs)
    assert ('CPG_GENR_TABLES_GLOBAL' in open(str(tmp_path / 'cpg_resident_pf20.h')).read()) == four_waves
    lib = build_sim.build_family(plan, str(tmp_path), 'pf20')
    vals, th, upd = _portfolio_values(d, B, n, m)
    bs = BatchSolver(d, lib_path=lib, plan=plan)
    for stg in ({}, dict(max_iter=60), dict(eps_abs=1e-7, eps_rel=1e-7)):
        r = bs.solve(vals, updated_params=upd, **stg)
        assert _resident_in_use(bs)
        o, prim, dual = _oracle_flat(oracle_lib, d, th, upd, **stg)
        _assert_parity(r, o, prim, dual, tol=1e-8)
    bs.set_program_placement(0)                       # the streaming kernel (unmerged program from HBM)
    r2 = bs.solve(vals, updated_params=upd, eps_abs=1e-7, eps_rel=1e-7)
    assert not _resident_in_use(bs)
    assert r2.iter.tolist() == r.iter.tolist() and np.abs(r2.prim_flat - r.prim_flat).max() < 1e-8
    bs.close()


def test_generic_library_keeps_the_streaming_kernel(sim_lib, oracle_lib):
    """a library without this family's resident executor (the generic table-driven one): cpg_hip_set_resident is never
    offered the tables, the streaming kernel serves the handle"""
    from test_sim_kernel import _assert_parity, _oracle_flat
    d = families.portfolio(12, 2)
    vals, th, upd = _portfolio_values(d, 2, 12, 2, seed=7)
    bs = BatchSolver(d, lib_path=sim_lib)
    r = bs.solve(vals, updated_params=upd, max_iter=50)
    assert not _resident_in_use(bs)
    o, prim, dual = _oracle_flat(oracle_lib, d, th, upd, max_iter=50)
    _assert_parity(r, o, prim, dual, tol=1e-8)
    bs.close()


@pytest.mark.gpu
def test_config3_family_runs_the_resident_kernel_vs_oracle(oracle_lib):
    """BASELINE config 3's family (portfolio n=100 m=10) in its family library on the GPU: the resident kernel is what runs,
    iteration counts / statuses = oracle, prim / dual within 1e-6 -- default mode, a cut-off, and against the streaming
    kernel of the same library"""
    import os
    from test_sim_kernel import _assert_parity, _oracle_flat
    d = families.portfolio(100, 10)
    plan = build_family_plan(d)
    out = os.path.join(os.path.dirname(os.path.abspath(codegen.__file__)), 'generated', 'portfolio')
    lib = codegen.build_family_library(plan, out, 'portfolio')
    B = 48
    vals, th, upd = _portfolio_values(d, B, 100, 10)
    bs = BatchSolver(d, lib_path=lib, plan=plan)
    for stg in ({}, dict(max_iter=60)):
        r = bs.solve(vals, updated_params=upd, **stg)
        assert _resident_in_use(bs)
        o, prim, dual = _oracle_flat(oracle_lib, d, th, upd, **stg)
        _assert_parity(r, o, prim, dual, tol=1e-6)
    bs.set_program_placement(0)
    r2 = bs.solve(vals, updated_params=upd, max_iter=60)
    assert not _resident_in_use(bs)
    assert r2.iter.tolist() == r.iter.tolist() and np.abs(r2.prim_flat - r.prim_flat).max() < 1e-8
    bs.close()


@pytest.mark.gpu
def test_config3_full_shard_is_its_64_distinct_instances_repeated(oracle_lib):
    """BASELINE config 3 at its full per-GPU shard (125 000 instances): a size-independent property -- the shard is 64 distinct
    instances repeated, every copy must come back bit-identical to the first (instances share nothing but read-only tables and
    are handed to wavefronts in an order that depends on the run), and the 64 equal the oracle's (counts exact, 1e-6)"""
    import os
    from test_sim_kernel import _assert_parity, _oracle_flat
    d = families.portfolio(100, 10)
    plan = build_family_plan(d)
    out = os.path.join(os.path.dirname(os.path.abspath(codegen.__file__)), 'generated', 'portfolio')
    lib = codegen.build_family_library(plan, out, 'portfolio')
    K, B = 64, 125_000
    vals, th, upd = _portfolio_values(d, K, 100, 10, seed=11)
    bs = BatchSolver(d, lib_path=lib, plan=plan)
    bs.set_updated(upd)
    tv = bs.theta_var(vals)                                   # [K, np_var]
    reps = -(-B // K)
    tv_full = np.ascontiguousarray(np.tile(tv, (reps, 1))[:B])
    r = bs.solve(theta_var=tv_full, B=B)
    assert _resident_in_use(bs)
    o, prim, dual = _oracle_flat(oracle_lib, d, th, upd)
    first = type('R', (), dict(iter=r.iter[:K], status=r.status[:K], prim_flat=r.prim_flat[:K], dual_flat=r.dual_flat[:K],
                               obj_val=r.obj_val[:K], pri_res=r.pri_res[:K], dua_res=r.dua_res[:K]))()
    _assert_parity(first, o, prim, dual, tol=1e-6)
    idx = np.arange(B) % K
    assert np.array_equal(r.iter, r.iter[:K][idx]) and np.array_equal(r.status, r.status[:K][idx])
    assert np.array_equal(r.prim_flat, r.prim_flat[:K][idx]) and np.array_equal(r.dual_flat, r.dual_flat[:K][idx])
    assert np.array_equal(r.obj_val, r.obj_val[:K][idx])
    bs.close()
