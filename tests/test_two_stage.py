"""Two-stage differentiation (cvxpygen/generator.py:76-80, canonicalizer.py:54-65, 334-406): an
OSQP-form family solved by the conic interior-point kernel and differentiated through its OSQP form."""
from types import SimpleNamespace

import numpy as np
import pytest
import scipy.sparse as sp

from cvxpygen_amd import cpg, families
from cvxpygen_amd.lite import LiteProblem
from cvxpygen_amd.runtime import BatchSolver
from cvxpygen_amd.two_stage import TwoStageBatchSolver, qp_duals_from_conic, qp_to_conic


def test_conic_form_of_the_osqp_problem_is_the_same_problem():
    """host logic: rows [-A_eq; A], rhs [-l; u] as affine maps of theta; stationarity carries over with
    y = z_up - [z_low; 0]"""
    for d in (families.nonneg_ls(), families.toy_box(), families.mpc(2, 1, 3)):
        c = qp_to_conic(d)
        rng = np.random.default_rng(0)
        th = d.theta0.copy(); th[:d.NP] += 0.1 * rng.standard_normal(d.NP)
        q, cc = d.canon_at(th), c.canon_at(th)
        A = sp.csc_matrix((q['A'], d.A.indices, d.A.indptr), shape=d.A.shape).toarray()
        Ac = sp.csc_matrix((cc['A'], c.A.indices, c.A.indptr), shape=c.A.shape).toarray()
        assert c.cones == {'zero': 0, 'nonneg': d.n_eq + d.m, 'soc': []} and c.m == d.n_eq + d.m
        assert np.array_equal(Ac, np.vstack([-A[:d.n_eq], A]))
        assert np.array_equal(cc['b'], np.concatenate([-q['l'], q['u']]))
        assert np.array_equal(cc['P'], q['P']) and np.array_equal(cc['q'], q['q'])
        z = rng.random(c.m)
        assert np.allclose(Ac.T @ z, A.T @ qp_duals_from_conic(z, d.n_eq))
    bad = families.portfolio(6, 2)                      # P depends on parameters: the reference refuses (canonicalizer.py:338-342)
    if bad.changes.get('P', False):
        with pytest.raises(ValueError, match='extended DPP'):
            qp_to_conic(bad)


def _check_two_stage(d, vals, names, lib, oracle_lib, dv):
    B = next(iter(vals.values())).shape[0]
    ts = TwoStageBatchSolver(d, lib_path=lib)
    r = ts.solve(vals, updated_params=names)
    assert (r.status == 1).all() and r.iter.max() <= 30           # interior point: a handful of iterations
    qs = BatchSolver(d, lib_path=lib, full_output=True)
    q = qs.solve(vals, updated_params=names, eps_abs=1e-10, eps_rel=1e-10, max_iter=50000)
    assert (q.status == 1).all()
    # same problem, two algorithms: solutions agree to solver accuracy (interior point at Clarabel's default
    # tolerances against ADMM at 1e-10; the objective is second-order flat at the solution)
    assert np.abs(r.sol_x - q.sol_x).max() <= 1e-4 * max(1.0, np.abs(q.sol_x).max())
    assert np.abs(r.sol_y - q.sol_y).max() <= 1e-4 * max(1.0, np.abs(q.sol_y).max())
    assert np.abs(r.obj_val - q.obj_val).max() <= 1e-6 * max(1.0, np.abs(q.obj_val).max())
    for v in d.variables:
        assert np.array_equal(r.prim[v.name].reshape(B, -1), r.sol_x[:, v.indices])
    # the OSQP-form adjoint at the conic solution = the oracle's adjoint at the same (x, y)
    g = ts.gradient(vals, r.sol_x, r.sol_y, dv, updated_params=names)
    th = np.tile(d.theta0, (B, 1))
    for nm in names:
        p = d.param(nm)
        th[:, p.col:p.col + p.size] = np.stack([d.flatten_param(nm, vals[nm][k]) for k in range(B)])
    wts = np.zeros(d.n_var)
    for v in d.variables:
        wts[v.indices] = 0.1
    for k in range(min(B, 4)):
        go = oracle_lib.qp_adjoint(d, d.canon_at(th[k]), r.sol_x[k], r.sol_y[k], wts)
        cols = np.concatenate([np.arange(d.param(nm).col, d.param(nm).col + d.param(nm).size) for nm in names])
        assert np.abs(g['_flat'][k] - go['dtheta'][cols]).max() <= 1e-8 * max(1e-6, np.abs(go['dtheta']).max())
    ts.close(); qs.close()


def test_two_stage_solver_in_emulator(sim_lib, oracle_lib):
    d = families.nonneg_ls()
    rng = np.random.default_rng(0)
    vals = {'A': rng.standard_normal((3, 3)), 'b': rng.standard_normal((3, 3))}
    _check_two_stage(d, vals, ['A', 'b'], sim_lib, oracle_lib, {'x': 0.1 * np.ones((3, 2))})


def test_generate_code_two_stage_surface(sim_lib, oracle_lib, tmp_path):
    """generate_code(solver='CLARABEL', gradient=True) on an OSQP-form family: cpg_solve through the conic kernel,
    cpg_gradient / forward / backward through the OSQP-form adjoint"""
    d = families.nonneg_ls()
    prob = LiteProblem.from_descriptor(d)
    cpg.generate_code(prob, code_dir=str(tmp_path / 'ts_code'), solver='CLARABEL', gradient=True, wrapper=False)
    mod = cpg.load_generated(str(tmp_path / 'ts_code'), prob)
    assert mod._SOLVER.two_stage
    mod._SOLVER.lib_path = sim_lib
    val, gp, gd = mod.cpg_solve_and_gradient_info(prob)
    assert len(gp) == d.n_var and len(gd) == d.m and prob.status.startswith('1 ')      # Clarabel status code, as the reference prints it
    ref = cpg.generate_code(LiteProblem.from_descriptor(d), code_dir=str(tmp_path / 'qp_code'), solver='OSQP', gradient=True, wrapper=False)
    p2 = LiteProblem.from_descriptor(d)
    m2 = cpg.load_generated(str(tmp_path / 'qp_code'), p2)
    m2._SOLVER.lib_path = sim_lib
    v2, gp2, gd2 = m2.cpg_solve_and_gradient_info(p2, eps_abs=1e-10, eps_rel=1e-10, max_iter=50000)
    assert abs(val - v2) <= 1e-6 * max(1.0, abs(v2)) and np.abs(np.array(gp) - np.array(gp2)).max() <= 1e-5
    assert np.allclose(prob.var_dict['x'].value, p2.var_dict['x'].value, atol=1e-5)
    prob.var_dict['x'].gradient = np.array([0.1, 0.1])
    mod.cpg_gradient(prob, gp, gd)
    wts = np.zeros(d.n_var); wts[d.variables[0].indices] = 0.1
    go = oracle_lib.qp_adjoint(d, d.default_canon(), np.array(gp), np.array(gd), wts)
    pb = d.param('b')
    assert np.allclose(np.ravel(prob.param_dict['b'].gradient), go['dtheta'][pb.col:pb.col + pb.size], rtol=1e-8, atol=1e-12)
    with pytest.raises(NotImplementedError, match='OSQP form'):
        cpg.generate_code(families.adp(), code_dir=str(tmp_path / 'x'), solver='CLARABEL', gradient=True, wrapper=False)


@pytest.mark.gpu
def test_two_stage_on_the_gpu(oracle_lib):
    """conic kernel forward, OSQP-form adjoint kernel backward, 4 096 instances of the least-squares family with
    every parameter per instance (the interior-point kernel keeps the whole state of an instance in LDS: small
    families only, like the reference's conic examples)"""
    B = 4096
    rng = np.random.default_rng(9)
    d = families.nonneg_ls(10, 5, sparsity=None, seed=0)
    th = np.tile(d.theta0, (B, 1)); th[:, :d.NP] *= 1 + 0.05 * rng.standard_normal((B, d.NP))
    vals = {p.name: th[:, p.col:p.col + p.size].reshape((B,) + tuple(p.shape), order='F') for p in d.params}
    dv = {v.name: 0.1 * np.ones((B,) + tuple(v.shape)) for v in d.variables}
    _check_two_stage(d, vals, d.param_names, None, oracle_lib, dv)
