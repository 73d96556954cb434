"""GPU tier (-m gpu) for the surface users touch (templates/cpg_solver.py.jinja2:40-212) and for the
workspace semantics around the hot path, on the real HIP library:
  * cpg.generate_code(problem, solver='OSQP' | 'CLARABEL', gradient=...) -> prob.solve(method='CPG')
    -> values against the oracle; cpg_gradient / forward / backward;
  * successive solves keep the reference's static-workspace state (parameter values, scaling of the
    last osqp_update_data_mat, warm start);
  * the OSQP build options (rho adaptation, duality-gap test) against the oracle in the same mode on
    BASELINE configs 2 and 3; row-class changes are solved, not flagged;
  * BASELINE config 5 at its full batch size: properties + a 256-instance oracle sample."""
from types import SimpleNamespace

import numpy as np
import pytest

from cvxpygen_amd import cpg, families
from cvxpygen_amd.lite import LiteProblem
from cvxpygen_amd.runtime import BatchSolver

pytestmark = pytest.mark.gpu
REL_TOL = 1e-6


def _theta(desc, values):
    B = next(iter(values.values())).shape[0]
    th = np.tile(desc.theta0, (B, 1))
    for name, v in values.items():
        p = desc.param(name)
        for k in range(B):
            th[k, p.col:p.col + p.size] = desc.flatten_param(name, v[k])
    return th


def _check(r, o, desc, tol=REL_TOL):
    prim = np.concatenate([o['sol_x'][:, v.indices] for v in desc.variables], axis=1)
    dual = np.concatenate([o['sol_y'][:, d.indices] for d in desc.duals], axis=1)
    assert (r.iter == o['iter']).all(), f'{int((r.iter != o["iter"]).sum())} iteration-count mismatches'
    assert (r.status == o['status']).all()
    ok = np.isin(o['status'], (1, 2, 7))
    assert np.abs(r.prim_flat[ok] - prim[ok]).max() <= tol * np.abs(prim[ok]).max()
    assert np.abs(r.dual_flat[ok] - dual[ok]).max() <= tol * np.abs(dual[ok]).max()
    assert np.abs(r.obj_val[ok] - o['obj_val'][ok]).max() <= tol * np.abs(o['obj_val'][ok]).max()


def test_generate_code_and_solve_method_cpg_osqp(oracle_lib, tmp_path):
    """the reference's own workflow on the example of examples/main.py, with gradient=True"""
    d = families.nonneg_ls()
    prob = LiteProblem.from_descriptor(d)
    mod = cpg.generate_code(prob, code_dir=str(tmp_path / 'nnls_code'), solver='OSQP', gradient=True, wrapper=True)
    ses = oracle_lib.CpgSession(d)
    rng = np.random.default_rng(2)
    A1, b1, b2 = rng.standard_normal(3), rng.standard_normal(3), rng.standard_normal(3)
    prob.param_dict['A'].value = A1
    prob.param_dict['b'].value = b1
    val = prob.solve(method='CPG', eps_abs=1e-7, eps_rel=1e-7)
    o = ses.solve({'A': A1, 'b': b1}, eps_abs=1e-7, eps_rel=1e-7)
    xi = d.variables[0].indices
    assert prob.status == 'solved' and prob._solution.attr['num_iters'] == o['iter']
    assert abs(val - o['obj_val']) <= REL_TOL * abs(o['obj_val'])
    assert np.abs(prob.var_dict['x'].value - o['x'][xi]).max() <= REL_TOL * np.abs(o['x']).max()
    assert np.abs(prob.constraints[0].dual_value - o['y'][d.duals[0].indices]).max() <= REL_TOL * max(1.0, np.abs(o['y']).max())
    assert prob._solver_stats.solver_name == 'OSQP' and prob._solution.opt_val == val
    # second call: only b listed -- A of call 1 stays, warm start from call 1 (static workspace of the reference)
    prob.param_dict['b'].value = b2
    val = prob.solve(method='CPG', updated_params=['b'], eps_abs=1e-7, eps_rel=1e-7)
    o = ses.solve({'b': b2}, eps_abs=1e-7, eps_rel=1e-7)
    assert prob._solution.attr['num_iters'] == o['iter']
    assert np.abs(prob.var_dict['x'].value - o['x'][xi]).max() <= REL_TOL * max(1.0, np.abs(o['x']).max())
    # unknown names raise like the reference (templates/cpg_solver.py.jinja2:48-60)
    with pytest.raises(AttributeError):
        prob.solve(method='CPG', updated_params=['nope'])
    with pytest.raises(AttributeError):
        prob.solve(method='CPG', not_a_setting=1)
    # gradient surface: cpg_solve_and_gradient_info + cpg_gradient, then the cvxpylayers protocol
    val, gp, gd = mod.cpg_solve_and_gradient_info(prob, eps_abs=1e-9, eps_rel=1e-9, warm_start=False)
    prob.var_dict['x'].gradient = np.array([0.1, 0.1])
    mod.cpg_gradient(prob, gp, gd)
    wts = np.zeros(d.n_var); wts[xi] = 0.1
    th = d.theta0.copy()
    th[d.param('A').col:d.param('A').col + 3] = A1
    th[d.param('b').col:d.param('b').col + 3] = b2
    go = oracle_lib.qp_adjoint(d, d.canon_at(th), np.array(gp), np.array(gd), wts)
    for nm in ('A', 'b'):
        p = d.param(nm)
        ref = go['dtheta'][p.col:p.col + p.size]
        assert np.abs(np.ravel(prob.param_dict[nm].gradient) - ref).max() <= REL_TOL * np.abs(go['dtheta']).max() + 1e-12
    for p in prob.parameters():
        p.id = id(p)
    ctx = SimpleNamespace(solver_args={'problem': prob}, param_ids=[p.id for p in prob.parameters()],
                          variables=prob.variables(), info=None)
    sol, info = mod.forward([p.value for p in prob.parameters()], ctx)
    ctx.info = info
    grads, _ = mod.backward([np.array([0.1, 0.1])], ctx)
    assert len(sol) == 1 and len(grads) == 2
    # forward = one more cpg_solve on the static workspace (warm start from the previous call); backward = the
    # adjoint at THAT solution, checked against the oracle's adjoint like cpg_gradient above
    go2 = oracle_lib.qp_adjoint(d, d.canon_at(th), np.array(info['gradient_primal']), np.array(info['gradient_dual']), wts)
    for p_, g in zip(prob.parameters(), grads):
        q = d.param(p_.name())
        ref = go2['dtheta'][q.col:q.col + q.size]
        assert np.abs(np.ravel(g) - ref).max() <= REL_TOL * np.abs(go2['dtheta']).max() + 1e-12
    assert np.abs(np.asarray(sol[0]) - np.asarray(info['gradient_primal'])[xi]).max() == 0.0


@pytest.mark.parametrize('fam,B', [('nonneg_LS', 300), ('mpc6', 256)])
def test_batched_forward_backward_on_gpu(oracle_lib, tmp_path, fam, B):
    """row (f)3: cvxpylayers' custom_method protocol (templates/cpg_solver.py.jinja2:176-212) with a leading batch
    axis on the parameters -- ONE batched solve and ONE batched adjoint on the GPU against the oracle, instance
    by instance: forward = oracle.cpg_solve_batch, backward = oracle.qp_adjoint at that solution"""
    d = families.nonneg_ls() if fam == 'nonneg_LS' else families.mpc(6, 3, 10)
    prob = LiteProblem.from_descriptor(d)
    mod = cpg.generate_code(prob, code_dir=str(tmp_path / f'fb_{fam}'), solver='OSQP', gradient=True, wrapper=True)
    rng = np.random.default_rng(55)
    if fam == 'nonneg_LS':
        vals = {'A': rng.standard_normal((B, 3)), 'b': rng.standard_normal((B, 3))}
        stg = dict(eps_abs=1e-9, eps_rel=1e-9)
    else:
        vals = {'x_init': -2 + 4 * rng.random((B, 6))}
        stg = dict(eps_abs=1e-7, eps_rel=1e-7)
    for p_ in prob.parameters():
        p_.id = id(p_)
    plist = [p_ for p_ in prob.parameters() if p_.name() in vals]
    ctx = SimpleNamespace(solver_args={'problem': prob, **stg}, param_ids=[p_.id for p_ in plist],
                          variables=prob.variables(), info=None)
    sol, info = mod.forward([vals[p_.name()] for p_ in plist], ctx)
    assert info['batched'] and (info['status'] == 1).all()
    th = np.tile(d.theta0, (B, 1))
    for nm, v in vals.items():
        q = d.param(nm)
        th[:, q.col:q.col + q.size] = v
    o = oracle_lib.cpg_solve_batch(d, th, list(vals), **stg)
    assert info['iter'].tolist() == o['iter'].tolist()
    for var, sv in zip(ctx.variables, sol):
        v = next(x for x in d.variables if x.name == var.name())
        ref = o['sol_x'][:, v.indices]
        got = np.asarray(sv).reshape(B, -1) if len(v.shape) <= 1 else np.asarray(sv).transpose(0, 2, 1).reshape(B, -1)
        assert np.abs(got - ref).max() <= REL_TOL * np.abs(ref).max()
    ctx.info = info
    ups = [0.1 * np.ones((B,) + tuple(var.shape)) for var in ctx.variables]       # 0.1 * sol.sum(), tests/test_diff.py:38
    grads, _ = mod.backward(ups, ctx)
    wts = np.zeros(d.n_var)
    for v in d.variables:
        wts[v.indices] = 0.1
    for k in range(0, B, 16):
        go = oracle_lib.qp_adjoint(d, d.canon_at(th[k]), info['gradient_primal'][k], info['gradient_dual'][k], wts)
        for p_, g in zip(plist, grads):
            q = d.param(p_.name())
            ref = go['dtheta'][q.col:q.col + q.size]
            assert np.abs(np.ravel(g[k]) - ref).max() <= REL_TOL * np.abs(go['dtheta']).max() + 1e-12, (fam, k, p_.name())


def test_generate_code_and_solve_method_cpg_clarabel(tmp_path):
    from oracle import clarabel_numpy as cl
    d = families.adp()
    prob = LiteProblem.from_descriptor(d)
    cpg.generate_code(prob, code_dir=str(tmp_path / 'adp_code'), solver='CLARABEL', wrapper=True)
    vals = families.adp_values(-2 + 4 * np.random.RandomState(5).rand(6))
    for k, v in vals.items():
        prob.param_dict[k].value = v
    val = prob.solve(method='CPG')
    o = cl.cpg_solve_batch(d, d.theta_from_values(vals)[None, :])
    assert prob.status.startswith('1 ') and prob._solution.attr['num_iters'] == int(o['iter'][0])
    assert abs(val - o['obj_val'][0]) <= REL_TOL * max(1.0, abs(o['obj_val'][0]))
    u = np.concatenate([np.ravel(prob.var_dict[v.name].value, order='F') for v in d.variables])
    uo = np.concatenate([o['sol_x'][0, v.indices] for v in d.variables])
    assert np.abs(u - uo).max() <= REL_TOL * np.abs(uo).max()


@pytest.mark.parametrize('fam,B', [('mpc6', 256), ('mpc12', 256)])
@pytest.mark.parametrize('opts', [dict(adaptive_rho=1, adaptive_rho_interval=50, check_dualgap=1),
                                  dict(adaptive_rho=1, adaptive_rho_interval=25), dict(check_dualgap=1)])
def test_build_options_vs_oracle_config2(oracle_lib, fam, B, opts):
    """BASELINE config 2 in the other rho / termination modes of the OSQP build"""
    d = families.mpc(6, 3, 10) if fam == 'mpc6' else families.mpc(12, 4, 10)
    x0 = -2 + 4 * np.random.default_rng(17).random((B, d.param('x_init').size))
    bs = BatchSolver(d, build_options=opts)
    r = bs.solve({'x_init': x0}, updated_params=['x_init'])
    o = oracle_lib.cpg_solve_batch(d, _theta(d, {'x_init': x0}), ['x_init'], **opts)
    _check(r, o, d)
    assert (r.status == 1).all()
    bs.close()


def test_adaptive_rho_vs_oracle_config3(oracle_lib):
    """BASELINE config 3 (portfolio: matrix parameters) with rho adaptation and the duality-gap test"""
    d = families.portfolio(100, 10)
    B = 48
    rng = np.random.default_rng(31)
    sig = np.zeros((B, 10, 10)); sig[:, np.arange(10), np.arange(10)] = rng.random((B, 10))
    vals = {'a': rng.standard_normal((B, 100)), 'F': np.round(rng.standard_normal((B, 100, 10))),
            'Sig_f_sqrt': sig, 'd_sqrt': rng.random((B, 100))}
    opts = dict(adaptive_rho=1, adaptive_rho_interval=50, check_dualgap=1)
    bs = BatchSolver(d, build_options=opts)
    r = bs.solve(vals, updated_params=list(vals))
    o = oracle_lib.cpg_solve_batch(d, _theta(d, vals), list(vals), **opts)
    _check(r, o, d)
    bs.close()


def test_row_class_changes_are_solved_on_gpu(oracle_lib):
    d = families.toy_box()
    B = 300
    rng = np.random.default_rng(9)
    th = np.tile(d.theta0, (B, 1))
    th[:, d.param('a').col] = 3 * rng.standard_normal(B)
    free = rng.random(B) < 0.25
    th[free, d.param('ub').col] = 1e30                       # inequality row -> free row
    infeas = (~free) & (rng.random(B) < 0.2)
    th[infeas, d.param('lb').col] = 2.0
    th[infeas, d.param('ub').col] = 1.0
    bs = BatchSolver(d)
    vals = {p.name: th[:, p.col:p.col + p.size] for p in d.params}
    r = bs.solve(vals)
    o = oracle_lib.cpg_solve_batch(d, th, None)
    assert (r.status != -2).all() and free.any() and infeas.any()
    _check(r, o, d, tol=1e-6)
    bs.close()


def test_config5_full_batch_forward_and_adjoint(oracle_lib):
    """BASELINE config 5: MPC 12/4/10, gradient=True, 100 000 instances -- forward + batched adjoint"""
    d = families.mpc(12, 4, 10)
    B = 100000
    rng = np.random.default_rng(41)
    x0 = -2 + 4 * rng.random((B, 12))
    x0[1] = x0[0]
    bs = BatchSolver(d, full_output=True)
    r = bs.solve({'x_init': x0}, updated_params=['x_init'])
    assert (r.status == 1).all()
    dv = {v.name: 0.1 * np.ones((B,) + tuple(v.shape)) for v in d.variables}    # 0.1 * sol.sum(), tests/test_diff.py:38
    g = bs.gradient({'x_init': x0}, r.sol_x, r.sol_y, dv, updated_params=['x_init'])
    dth = g['_flat']
    assert dth.shape == (B, d.NP) and np.isfinite(dth).all()
    assert np.array_equal(dth[0], dth[1])                                         # duplicates: identical bits
    # linearity of the adjoint in the upstream gradient (size-independent property)
    dv2 = {k: 2.0 * v for k, v in dv.items()}
    g2 = bs.gradient({'x_init': x0[:2048]}, r.sol_x[:2048], r.sol_y[:2048], {k: v[:2048] for k, v in dv2.items()},
                     updated_params=['x_init'])
    assert np.abs(g2['_flat'] - 2.0 * dth[:2048]).max() <= 1e-9 * max(1.0, np.abs(dth[:2048]).max())
    # oracle sample
    wts = np.zeros(d.n_var)
    for v in d.variables:
        wts[v.indices] = 0.1
    th = _theta(d, {'x_init': x0[:256]})
    o = oracle_lib.cpg_solve_batch(d, th, ['x_init'])
    assert (r.iter[:256] == o['iter']).all()
    for k in range(0, 256, 8):
        go = oracle_lib.qp_adjoint(d, d.canon_at(th[k]), r.sol_x[k], r.sol_y[k], wts)
        assert np.abs(dth[k] - go['dtheta']).max() <= 1e-6 * np.abs(go['dtheta']).max() + 1e-10
    bs.close()


@pytest.mark.parametrize('opts', [{}, dict(adaptive_rho=1, adaptive_rho_interval=50, check_dualgap=1)])
def test_portfolio_family_library_vs_oracle(oracle_lib, opts):
    """BASELINE config 3 on what generate_code builds for it: the family library (per-instance factor kernel
    compiled for the family's exact slot class), in both rho modes"""
    import os
    from cvxpygen_amd import codegen
    from cvxpygen_amd.runtime import build_family_plan
    d = families.portfolio(100, 10)
    plan = build_family_plan(d)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = codegen.build_family_library(plan, os.path.join(root, 'cvxpygen_amd', 'generated', 'portfolio'), 'portfolio')
    B = 64
    rng = np.random.default_rng(77)
    sig = np.zeros((B, 10, 10)); sig[:, np.arange(10), np.arange(10)] = rng.random((B, 10))
    vals = {'a': rng.standard_normal((B, 100)), 'F': np.round(rng.standard_normal((B, 100, 10))),
            'Sig_f_sqrt': sig, 'd_sqrt': rng.random((B, 100))}
    bs = BatchSolver(d, lib_path=lib, plan=plan, build_options=opts)
    r = bs.solve(vals, updated_params=list(vals))
    o = oracle_lib.cpg_solve_batch(d, _theta(d, vals), list(vals), **opts)
    _check(r, o, d)
    bs.close()


@pytest.mark.parametrize('lib', ['family', 'generic'])
def test_hybrid_execution_edge_cases(oracle_lib, lib):
    """the two-kernel execution of the default mode (shared factor until an instance's rho changes, per-instance
    factor behind it) where its bookkeeping is most exposed: cut-offs at / between the adaptation points, an
    adaptation interval that is not a multiple of check_termination, two instances per wavefront, and a batch whose
    instances arrive with different workspace rhos (hand-over at iteration 0) -- all against the oracle."""
    import os
    d = families.mpc(6, 3, 10)
    B = 200
    x0 = -2 + 4 * np.random.default_rng(23).random((B, 6))
    th = _theta(d, {'x_init': x0})
    gen = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cvxpygen_amd', 'generated', 'mpc6', 'libcpg_mpc6.so')
    lib_path = gen if (lib == 'family' and os.path.exists(gen)) else None
    for opts, stg, G in (({}, dict(max_iter=50), 1), ({}, dict(max_iter=63), 1), ({}, dict(max_iter=30), 1),
                         (dict(adaptive_rho_interval=35), {}, 1), ({}, dict(eps_abs=1e-6, eps_rel=1e-6), 1), ({}, {}, 2)):
        bs = BatchSolver(d, lib_path=lib_path, build_options=opts)
        bs.set_launch(0, G, 0)
        r = bs.solve({'x_init': x0}, updated_params=['x_init'], **stg)
        o = oracle_lib.cpg_solve_batch(d, th, ['x_init'], **opts, **stg)
        _check(r, o, d)
        assert bs._hybrid and (r.status != -3).all() and (r.status != -2).all()
        bs.close()
    # workspaces with their own rho (a sequential caller's state): per instance, against CpgSession
    bs = BatchSolver(d, lib_path=lib_path)
    r1 = bs.solve({'x_init': x0[:8]}, updated_params=['x_init'], return_state=True)
    x1 = -2 + 4 * np.random.default_rng(24).random((8, 6))
    r2 = bs.solve({'x_init': x1}, updated_params=['x_init'], state_in=r1.state, return_state=True)
    assert len(set(np.round(r1.state[:, -1], 12))) > 1                  # the batch really carries different rhos
    for k in range(8):
        ses = oracle_lib.CpgSession(d)
        ses.solve({'x_init': x0[k]})
        o = ses.solve({'x_init': x1[k]})
        assert r2.iter[k] == o['iter'] and r2.status[k] == o['status'], k
        got = np.concatenate([np.ravel(r2.prim[v.name][k], order='F') for v in d.variables])
        ref = np.concatenate([o['x'][v.indices] for v in d.variables])
        assert np.abs(got - ref).max() <= REL_TOL * max(1.0, np.abs(ref).max()), k
        assert abs(r2.state[k, -1] - o['rho']) <= 1e-9 * o['rho']
    bs.close()


@pytest.mark.gpu
def test_family_libraries_run_their_generated_instance_executor():
    """the per-instance phase of the default mode must run on the generated instance executor of the family library
    (register-resident coefficients), not fall back silently to the streaming one -- e.g. because its LDS tables no
    longer fit: the handle says which (`generated_instance_executor`), for the headline family and the notebook one"""
    import ctypes as C
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, d in (('mpc12', families.mpc(12, 4, 10)), ('mpc6', families.mpc(6, 3, 10))):
        gen = os.path.join(root, 'cvxpygen_amd', 'generated', name, f'libcpg_{name}.so')
        assert os.path.exists(gen), gen                       # __graft_entry__.build() made it
        bs = BatchSolver(d, lib_path=gen)
        x0 = -2 + 4 * np.random.default_rng(3).random((64, d.param('x_init').size))
        r = bs.solve({'x_init': x0}, updated_params=['x_init'])
        assert bs._hybrid and (r.status == 1).all()
        v = C.c_double(-1)
        bs.lib.check(bs.lib.L.cpg_hip_get_setting(bs.h_rs, b'generated_instance_executor', C.byref(v)), 'get_setting')
        assert v.value == 1.0, name
        ms = bs.last_phase_ms()
        assert ms[1] > 0.0 and ms[2] > 0                      # the second kernel ran (instances changed rho)
        bs.close()
