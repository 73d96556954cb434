"""QP adjoint (gradient=True), SURVEY.md section 8 rows G1-G6.  The reference tests it against
cvxpylayers / finite differences on nonneg least squares (tests/test_diff.py:14-69, 147-164); here:
restated adjoint (numpy + C) vs finite differences of the forward oracle, emulator build of the HIP
adjoint kernel vs the oracle, and the reference-facing cpg_gradient / forward / backward surface."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import scipy.sparse as sp

from cvxpygen_amd import cpg, families
from cvxpygen_amd.lite import LiteProblem
from cvxpygen_amd.runtime import BatchSolver
from oracle.osqp_grad_numpy import dtheta_from_canonical, qp_adjoint


def _tight_solve(oracle_lib, d, th):
    o = oracle_lib.cpg_solve_batch(d, th[None, :], None, eps_abs=1e-11, eps_rel=1e-11, max_iter=300000)
    assert o['status'][0] == 1
    return o['sol_x'][0], o['sol_y'][0]


@pytest.mark.parametrize('make', [lambda: families.nonneg_ls(10, 5, sparsity=None, seed=0),
                                  lambda: families.nonneg_ls(),
                                  lambda: families.mpc(4, 2, 3)])
def test_adjoint_oracles_match_finite_differences(oracle_lib, make):
    d = make()
    rng = np.random.default_rng(1)
    th = np.append(d.theta0[:-1] * (1 + 0.05 * rng.standard_normal(d.NP)), 1.0)
    x, y = _tight_solve(oracle_lib, d, th)
    wts = np.zeros(d.n_var)
    for v in d.variables:
        wts[v.indices] = 0.1                          # loss = 0.1 * sum(variables), tests/test_diff.py:38
    c = d.canon_at(th)
    Pd = sp.csc_matrix((c['P'], d.P.indices, d.P.indptr), shape=d.P.shape).toarray()
    Ad = sp.csc_matrix((c['A'], d.A.indices, d.A.indptr), shape=d.A.shape).toarray()
    dth = dtheta_from_canonical(d, qp_adjoint(Pd, Ad, x, y, wts))
    gc = oracle_lib.qp_adjoint(d, c, x, y, wts)
    assert np.abs(gc['dtheta'] - dth).max() <= 1e-12 * max(1.0, np.abs(dth).max())
    for k in rng.choice(d.NP, size=min(6, d.NP), replace=False):
        h = 1e-6 * max(1.0, abs(th[k]))
        tp, tm = th.copy(), th.copy()
        tp[k] += h; tm[k] -= h
        fd = (wts @ _tight_solve(oracle_lib, d, tp)[0] - wts @ _tight_solve(oracle_lib, d, tm)[0]) / (2 * h)
        assert abs(fd - dth[k]) <= 1e-6 * max(1.0, abs(fd))


def test_adjoint_kernel_vs_oracle_emulator(sim_lib, oracle_lib):
    d = families.nonneg_ls(10, 5, sparsity=None, seed=0)
    B = 2
    rng = np.random.default_rng(1)
    th = np.tile(d.theta0, (B, 1))
    th[:, :d.NP] *= 1 + 0.05 * rng.standard_normal((B, d.NP))
    bs = BatchSolver(d, lib_path=sim_lib, full_output=True)
    vals = {p.name: th[:, p.col:p.col + p.size] for p in d.params}
    r = bs.solve(vals, eps_abs=1e-4, eps_rel=1e-4, max_iter=100)     # the adjoint is checked at whatever (x, y) comes out
    assert r.sol_x.shape == (B, d.n_var) and r.prim['x'].shape == (B, 5)
    g = bs.gradient(vals, r.sol_x, r.sol_y, {'x': 0.1 * np.ones((B, 5))})
    assert g['A'].shape == (B, 10, 5) and g['b'].shape == (B, 10)
    for k in range(B):
        wts = np.zeros(d.n_var); wts[d.variables[0].indices] = 0.1
        go = oracle_lib.qp_adjoint(d, d.canon_at(th[k]), r.sol_x[k], r.sol_y[k], wts)
        assert np.abs(g['_flat'][k] - go['dtheta']).max() <= 1e-10 * np.abs(go['dtheta']).max()
        pA = d.param('A')
        assert np.allclose(g['A'][k].flatten(order='F'), go['dtheta'][pA.col:pA.col + pA.size], rtol=1e-9, atol=1e-14)
    bs.close()


def test_reference_gradient_surface(sim_lib, oracle_lib, tmp_path):
    """generate_code(gradient=True) -> cpg_solve_and_gradient_info / cpg_gradient / forward / backward"""
    d = families.nonneg_ls()                                      # examples/main.py family
    prob = LiteProblem.from_descriptor(d)
    cpg.generate_code(prob, code_dir=str(tmp_path / 'grad_code'), solver='OSQP', gradient=True, wrapper=False)   # (no hipcc step: the emulator library is injected)
    mod = cpg.load_generated(str(tmp_path / 'grad_code'), prob)
    mod._SOLVER.lib_path = sim_lib
    val, gp, gd = mod.cpg_solve_and_gradient_info(prob, eps_abs=1e-9, eps_rel=1e-9)
    assert len(gp) == d.n_var and len(gd) == d.m and prob.status == 'solved'
    prob.var_dict['x'].gradient = np.array([0.1, 0.1])
    mod.cpg_gradient(prob, gp, gd)
    wts = np.zeros(d.n_var); wts[d.variables[0].indices] = 0.1
    go = oracle_lib.qp_adjoint(d, d.default_canon(), np.array(gp), np.array(gd), wts)
    pA, pb = d.param('A'), d.param('b')
    assert np.allclose(np.ravel(prob.param_dict['b'].gradient), go['dtheta'][pb.col:pb.col + pb.size], rtol=1e-8, atol=1e-12)
    assert np.allclose(np.ravel(prob.param_dict['A'].gradient), go['dtheta'][pA.col:pA.col + pA.size], rtol=1e-8, atol=1e-12)
    assert np.abs(go['dtheta']).max() > 1e-3
    # cvxpylayers custom_method protocol
    ctx = SimpleNamespace(solver_args={'problem': prob}, param_ids=[id(p) for p in prob.parameters()],
                          variables=prob.variables(), info=None)
    for p in prob.parameters():
        p.id = id(p)
    sol, info = mod.forward([p.value for p in prob.parameters()], ctx)
    ctx.info = info
    grads, _ = mod.backward([np.array([0.1, 0.1])], ctx)
    assert len(grads) == 2 and all(g is not None for g in grads)


def test_batched_forward_backward_adapter(sim_lib, oracle_lib, tmp_path):
    """cvxpylayers protocol with a leading batch axis on the parameters (templates/cpg_solver.py.jinja2:176-212;
    the reference's users loop over the batch, examples/paper_grad/ADP.py:80-88): ONE batched solve + ONE
    batched adjoint, equal to the single-instance forward / backward instance by instance"""
    d = families.nonneg_ls()
    prob = LiteProblem.from_descriptor(d)
    cpg.generate_code(prob, code_dir=str(tmp_path / 'fb_code'), solver='OSQP', gradient=True, wrapper=False)   # (no hipcc step: the emulator library is injected)
    mod = cpg.load_generated(str(tmp_path / 'fb_code'), prob)
    mod._SOLVER.lib_path = sim_lib
    B = 3
    rng = np.random.default_rng(12)
    Ab, bb = rng.standard_normal((B, 3)), rng.standard_normal((B, 3))
    ctx = SimpleNamespace(solver_args={'problem': prob, 'eps_abs': 1e-9, 'eps_rel': 1e-9},
                          param_ids=[p.id for p in prob.parameters()], variables=prob.variables(), info=None)
    order = [p.name() for p in prob.parameters()]
    batch_vals = [dict(A=Ab, b=bb)[nm] for nm in order]
    sol, info = mod.forward(batch_vals, ctx)
    assert sol[0].shape == (B, 2) and info['batched'] and (info['status'] == 1).all()
    ctx.info = info
    up = 0.1 * np.ones((B, 2))
    grads, _ = mod.backward([up], ctx)
    assert [g.shape for g in grads] == [(B, 3) if nm == 'A' else (B, 3) for nm in order]
    wts = np.zeros(d.n_var); wts[d.variables[0].indices] = 0.1
    for k in range(B):
        th = d.theta0.copy()
        th[d.param('A').col:d.param('A').col + 3] = Ab[k]
        th[d.param('b').col:d.param('b').col + 3] = bb[k]
        o = oracle_lib.cpg_solve_batch(d, th[None, :], ['A', 'b'], eps_abs=1e-9, eps_rel=1e-9)
        assert np.abs(sol[0][k] - o['sol_x'][0, d.variables[0].indices]).max() <= 1e-9
        go = oracle_lib.qp_adjoint(d, d.canon_at(th), info['gradient_primal'][k], info['gradient_dual'][k], wts)
        for nm, g in zip(order, grads):
            p = d.param(nm)
            assert np.abs(np.ravel(g[k]) - go['dtheta'][p.col:p.col + p.size]).max() <= 1e-8 * max(1.0, np.abs(go['dtheta']).max())
    # a parameter shared by the whole batch may come without the batch axis
    sol2, _ = mod.forward([dict(A=Ab[0], b=bb)[nm] for nm in order], ctx)
    assert np.abs(sol2[0][0] - sol[0][0]).max() <= 1e-12


def test_adjoint_on_the_pruned_factor_pattern(sim_lib, oracle_lib):
    """config 5 shape: only x_init varies, so the masked KKT matrix of every instance has the workspace's P and A --
    its factor is built on their numerically non-zero pattern (handle h_rg) while d(P), d(A) still cover every
    STORED entry (an entry that is zero has a gradient).  Must equal the oracle's adjoint and the stored-pattern path."""
    d = families.mpc(4, 2, 3)
    B = 3
    rng = np.random.default_rng(6)
    x0 = -2 + 4 * rng.random((B, 4))
    bs = BatchSolver(d, lib_path=sim_lib, full_output=True)
    r = bs.solve({'x_init': x0}, updated_params=['x_init'], eps_abs=1e-6, eps_rel=1e-6)
    dv = {v.name: 0.1 * np.ones((B,) + tuple(v.shape)) for v in d.variables}
    g = bs.gradient({'x_init': x0}, r.sol_x, r.sol_y, dv, updated_params=['x_init'])
    assert bs.h_grad is bs.h_rg and bs._rplan_g.nnzL < bs._rplan.nnzL          # the pruned factor was used
    full = {q.name: np.tile(d.theta0[q.col:q.col + q.size], (B, 1)) for q in d.params}
    full['x_init'] = x0
    g2 = bs.gradient(full, r.sol_x, r.sol_y, dv, updated_params=d.param_names)     # matrix parameters listed: stored pattern
    assert bs.h_grad is bs.h_ref
    assert np.abs(g['_flat'] - g2['_flat']).max() <= 1e-9 * max(1.0, np.abs(g2['_flat']).max())
    wts = np.zeros(d.n_var)
    for v in d.variables:
        wts[v.indices] = 0.1
    p = d.param('x_init')
    for k in range(B):
        th = d.theta0.copy(); th[p.col:p.col + p.size] = x0[k]
        go = oracle_lib.qp_adjoint(d, d.canon_at(th), r.sol_x[k], r.sol_y[k], wts)
        assert np.abs(g['_flat'][k] - go['dtheta']).max() <= 1e-8 * max(1.0, np.abs(go['dtheta']).max())
        assert np.abs(go['dtheta']).max() > 1e-4
    # a solve after the adjoint still uses its own tables (the advisor's round-2 finding)
    r2 = bs.solve({'x_init': x0}, updated_params=['x_init'], eps_abs=1e-6, eps_rel=1e-6)
    assert r2.iter.tolist() == r.iter.tolist() and np.array_equal(r2.prim_flat, r.prim_flat)
    bs.close()


def test_alternating_solve_subset_and_gradient_all(sim_lib, oracle_lib):
    """round-2 advisor finding: gradient() over all parameters must not disturb a solve that lists a subset --
    three rounds of solve(['b']) / gradient(['A', 'b']) give what a fresh solver gives every time"""
    d = families.nonneg_ls()
    rng = np.random.default_rng(30)
    B = 4
    bs = BatchSolver(d, lib_path=sim_lib, full_output=True)
    for rnd in range(3):
        bv = rng.standard_normal((B, 3))
        r = bs.solve({'b': bv}, updated_params=['b'])
        th = np.tile(d.theta0, (B, 1)); pb = d.param('b'); th[:, pb.col:pb.col + 3] = bv
        o = oracle_lib.cpg_solve_batch(d, th, ['b'])
        assert r.iter.tolist() == o['iter'].tolist()
        assert np.abs(r.sol_x - o['sol_x']).max() <= 1e-9 * max(1.0, np.abs(o['sol_x']).max()), rnd
        vals = {'A': np.tile(d.theta0[d.param('A').col:d.param('A').col + 3], (B, 1)), 'b': bv}
        g = bs.gradient(vals, r.sol_x, r.sol_y, {'x': 0.1 * np.ones((B, 2))}, updated_params=['A', 'b'])
        wts = np.zeros(d.n_var); wts[d.variables[0].indices] = 0.1
        go = oracle_lib.qp_adjoint(d, d.canon_at(th[0]), r.sol_x[0], r.sol_y[0], wts)
        assert np.abs(g['_flat'][0] - go['dtheta']).max() <= 1e-9 * max(1.0, np.abs(go['dtheta']).max())
    bs.close()
