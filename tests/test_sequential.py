"""CPU tier (lock-step emulator of the kernel sources) for the workspace semantics around the hot path:
  * the OSQP build options the reference inherits from `osqp.OSQP().setup()` -- rho adaptation and the
    duality-gap test -- in the per-instance factor kernel, against the oracle in the same mode;
  * a row whose bound moves it to another class is solved (the reference refactors), never returned
    as an internal status;
  * the B = 1 drop-in keeps the reference's static-workspace state between calls: parameter values of
    earlier updates, the scaling of the last osqp_update_data_mat, the iterates (warm_starting = 1)."""
import numpy as np
import pytest

from cvxpygen_amd import cpg, families
from cvxpygen_amd.lite import LiteProblem
from cvxpygen_amd.runtime import BatchSolver


def _theta(desc, values):
    B = next(iter(values.values())).shape[0]
    th = np.tile(desc.theta0, (B, 1))
    for name, v in values.items():
        p = desc.param(name)
        for k in range(B):
            th[k, p.col:p.col + p.size] = desc.flatten_param(name, v[k])
    return th


def _flat(o, desc):
    prim = np.concatenate([o['sol_x'][:, v.indices] for v in desc.variables], axis=1)
    dual = np.concatenate([o['sol_y'][:, d.indices] for d in desc.duals], axis=1)
    return prim, dual


def _parity(r, o, desc, tol=1e-9):
    prim, dual = _flat(o, desc)
    assert r.iter.tolist() == o['iter'].tolist()
    assert r.status.tolist() == o['status'].tolist()
    ok = np.isin(o['status'], (1, 2, 7))
    assert np.abs(r.prim_flat[ok] - prim[ok]).max() <= tol * max(1.0, np.abs(prim[ok]).max())
    assert np.abs(r.dual_flat[ok] - dual[ok]).max() <= tol * max(1.0, np.abs(dual[ok]).max())
    assert np.abs(r.obj_val[ok] - o['obj_val'][ok]).max() <= tol * max(1.0, np.abs(o['obj_val'][ok]).max())


@pytest.mark.parametrize('opts', [dict(adaptive_rho=1, adaptive_rho_interval=10, check_dualgap=0),
                                  dict(adaptive_rho=1, adaptive_rho_interval=25, check_dualgap=1),
                                  dict(adaptive_rho=1, adaptive_rho_interval=35),        # not a multiple of check_termination
                                  dict(adaptive_rho=0, check_dualgap=1), dict(adaptive_rho=0, check_dualgap=0), {}])
def test_build_options_vs_oracle_vector_parameters(sim_lib, oracle_lib, opts):
    """only q / l / u vary: the shared-factor kernel serves every instance until its rho changes, the per-instance
    factor kernel behind it the rest (hybrid execution); without rho adaptation one kernel"""
    d = families.nonneg_ls()
    rng = np.random.default_rng(3)
    vals = {'b': 3.0 * rng.standard_normal((5, 3))}
    bs = BatchSolver(d, lib_path=sim_lib, build_options=opts)
    for stg in ({}, dict(eps_abs=1e-8, eps_rel=1e-8), dict(max_iter=30), dict(max_iter=20, check_termination=0)):
        r = bs.solve(vals, updated_params=['b'], **stg)
        o = oracle_lib.cpg_solve_batch(d, _theta(d, vals), ['b'], **opts, **stg)
        _parity(r, o, d)
    assert bs.h is bs.h_shared and bs._hybrid == bool(opts.get('adaptive_rho', 1))     # default: OSQP >= 1.0, rho adapted
    bs.close()


def test_adaptive_rho_vs_oracle_mpc_and_matrix_parameters(sim_lib, oracle_lib):
    opts = dict(adaptive_rho=1, adaptive_rho_interval=50, check_dualgap=1)      # an OSQP >= 1.0 build
    d = families.mpc(6, 3, 10)
    x0 = -2 + 4 * np.random.default_rng(5).random((2, 6))
    bs = BatchSolver(d, lib_path=sim_lib, build_options=opts)
    bs.set_launch(waves_per_block=2)
    r = bs.solve({'x_init': x0}, updated_params=['x_init'])
    o = oracle_lib.cpg_solve_batch(d, _theta(d, {'x_init': x0}), ['x_init'], **opts)
    _parity(r, o, d)
    o_fixed = oracle_lib.cpg_solve_batch(d, _theta(d, {'x_init': x0}), ['x_init'], adaptive_rho=0, check_dualgap=0)
    assert o['iter'].tolist() != o_fixed['iter'].tolist()      # the mode matters on this family
    bs.close()
    d2 = families.nonneg_ls()                                    # A is a parameter: osqp_update_data_mat path
    rng = np.random.default_rng(8)
    vals = {'A': np.stack([rng.standard_normal(3) for _ in range(3)]), 'b': rng.standard_normal((3, 3))}
    bs2 = BatchSolver(d2, lib_path=sim_lib, build_options=dict(adaptive_rho=1, adaptive_rho_interval=10))
    r2 = bs2.solve(vals, updated_params=['A', 'b'], eps_abs=1e-7, eps_rel=1e-7)
    o2 = oracle_lib.cpg_solve_batch(d2, _theta(d2, vals), ['A', 'b'], adaptive_rho=1, adaptive_rho_interval=10,
                                    eps_abs=1e-7, eps_rel=1e-7)
    _parity(r2, o2, d2)
    bs2.close()


def test_row_class_change_is_solved_not_flagged(sim_lib, oracle_lib):
    """an upper bound of 1e30 turns an inequality row into a free row (and back): the reference's
    osqp_update_data_vec refactors; the batch gets the same result, no internal status"""
    d = families.toy_box()
    B = 5
    th = np.tile(d.theta0, (B, 1))
    ub = d.param('ub')
    th[1, ub.col] = 1e30                       # free row
    th[3, ub.col] = np.inf                     # same after replace_inf (utils.py:213-228)
    th[4, d.param('a').col] = 4.0
    th = np.clip(th, -1e30, 1e30)
    bs = BatchSolver(d, lib_path=sim_lib)
    upd = ['a', 'lb', 'ub']
    tv = np.concatenate([th[:, d.param(nm).col:d.param(nm).col + d.param(nm).size] for nm in
                         [q.name for q in d.params if q.name in upd]], axis=1)
    r = bs.solve(updated_params=upd, theta_var=tv)
    o = oracle_lib.cpg_solve_batch(d, th, upd)
    assert (r.status != -2).all()
    _parity(r, o, d)
    bs.close()


def test_drop_in_keeps_the_reference_workspace_between_calls(sim_lib, oracle_lib, tmp_path):
    d = families.nonneg_ls()                                       # parameters A (sparse) and b
    prob = LiteProblem.from_descriptor(d)
    cpg.generate_code(prob, code_dir=str(tmp_path / 'seq_code'), solver='OSQP', wrapper=False)   # (no hipcc step: the emulator library is injected)
    mod = cpg.load_generated(str(tmp_path / 'seq_code'), prob)
    mod._SOLVER.lib_path = sim_lib
    ses = oracle_lib.CpgSession(d)
    rng = np.random.default_rng(21)
    A1, b1 = rng.standard_normal(3), rng.standard_normal(3)
    b2, b3 = rng.standard_normal(3), 2.0 * rng.standard_normal(3)
    A2 = rng.standard_normal(3)

    def step(updates, updated_params, **kw):
        for name, v in updates.items():
            prob.param_dict[name].value = v
        val = prob.solve(method='CPG', updated_params=updated_params, **kw)
        o = ses.solve({k: updates[k] for k in (updated_params or updates)}, warm=bool(kw.get('warm_start', 1)),
                      **{k: v for k, v in kw.items() if k != 'warm_start'})
        return val, o

    # 1. everything updated
    val, o = step({'A': A1, 'b': b1}, ['A', 'b'], eps_abs=1e-6, eps_rel=1e-6)
    x1 = prob.var_dict['x'].value.copy()
    assert prob._solution.attr['num_iters'] == o['iter'] and abs(val - o['obj_val']) <= 1e-9 * max(1, abs(val))
    # 2. only b listed: A keeps the value of call 1 (not the code-generation one), the scaling of call 1's
    #    osqp_update_data_mat stays, and the iterates of call 1 warm-start the solve
    val, o = step({'b': b2}, ['b'], eps_abs=1e-6, eps_rel=1e-6)
    assert prob._solution.attr['num_iters'] == o['iter']
    assert np.abs(prob.var_dict['x'].value - o['x'][d.variables[0].indices]).max() <= 1e-9
    # 3. A changed in `prob` but NOT listed: the reference does not read it
    prob.param_dict['A'].value = A2
    val, o = step({'b': b3}, ['b'], eps_abs=1e-6, eps_rel=1e-6)
    assert prob._solution.attr['num_iters'] == o['iter']
    assert np.abs(prob.var_dict['x'].value - o['x'][d.variables[0].indices]).max() <= 1e-9
    # 4. now A is listed; warm_start=False cold-starts (cvxpy alias, solvers/osqp.py:110)
    val, o = step({'A': A2}, ['A'], eps_abs=1e-6, eps_rel=1e-6, warm_start=False)
    assert prob._solution.attr['num_iters'] == o['iter'] and abs(val - o['obj_val']) <= 1e-9 * max(1, abs(val))
    assert np.abs(prob.var_dict['x'].value - o['x'][d.variables[0].indices]).max() <= 1e-9
    assert np.abs(prob.var_dict['x'].value - x1).max() > 1e-6


def test_enabled_optional_settings_are_fields_without_effect(sim_lib, tmp_path):
    """`enable_settings=['polishing', ...]` (solvers/osqp.py:111-114): the names become accepted keywords; in the
    embedded OSQP build the reference emits they change a settings field and not the solve -- same here; a name that
    was not enabled is refused like any unknown setting"""
    d = families.toy_box()
    prob = LiteProblem.from_descriptor(d)
    cpg.generate_code(prob, code_dir=str(tmp_path / 'opt_code'), solver='OSQP', wrapper=False,
                      enable_settings=['polishing', 'polish_refine_iter', 'delta'])
    mod = cpg.load_generated(str(tmp_path / 'opt_code'), prob)
    mod._SOLVER.lib_path = sim_lib
    v0 = prob.solve(method='CPG', warm_start=False)
    it0, x0 = prob._solution.attr['num_iters'], prob.var_dict['x'].value.copy()
    v1 = prob.solve(method='CPG', warm_start=False, polishing=1, polish_refine_iter=3, delta=1e-7)
    assert v1 == v0 and prob._solution.attr['num_iters'] == it0 and np.array_equal(prob.var_dict['x'].value, x0)
    with pytest.raises(AttributeError, match='not available'):
        prob.solve(method='CPG', verbose=1)                        # not among the enabled names


def test_drop_in_sequence_on_a_vector_only_family(sim_lib, oracle_lib, tmp_path):
    """only vector parameters (q, l, u): shared factor in every call, warm start from the previous solution"""
    d = families.toy_box()
    prob = LiteProblem.from_descriptor(d)
    cpg.generate_code(prob, code_dir=str(tmp_path / 'seq_box'), solver='OSQP', wrapper=False)   # (no hipcc step: the emulator library is injected)
    mod = cpg.load_generated(str(tmp_path / 'seq_box'), prob)
    mod._SOLVER.lib_path = sim_lib
    ses = oracle_lib.CpgSession(d)
    rng = np.random.default_rng(4)
    iters = []
    a = np.asarray(prob.param_dict['a'].value, dtype=float).copy()
    for k in range(3):
        a = a + (rng.standard_normal(a.shape) if k < 2 else 1e-3)
        prob.param_dict['a'].value = a
        val = prob.solve(method='CPG', updated_params=['a'], eps_abs=1e-7, eps_rel=1e-7)
        o = ses.solve({'a': a}, eps_abs=1e-7, eps_rel=1e-7)
        assert prob._solution.attr['num_iters'] == o['iter'] and prob.status == 'solved'
        assert abs(val - o['obj_val']) <= 1e-9 * max(1.0, abs(val))
        iters.append(o['iter'])
    bs = mod._SOLVER.batch_solver
    assert bs.h.value == bs.h_shared.value
    assert iters[2] <= iters[0]                                       # a nearby problem does not converge slower warm


def test_row_class_change_after_an_adapted_rho_restarts_from_the_family_rho(sim_lib, oracle_lib, tmp_path):
    """call 1 adapts the workspace's rho (0.1 -> 1.66); call 2 frees the upper bound (ub = 1e30: the row changes class):
    OSQP's update_rho_vec rebuilds rho_vec from settings->rho -- 0.1 again after the per-call reset of the settings --
    and not from the adapted value the workspace carries; call 3 makes the bound finite again (oracle/osqp_oracle.c
    set_rho_vec; cvxpygen/solvers/osqp.py:100-101)"""
    d = families.toy_box()
    prob = LiteProblem.from_descriptor(d)
    cpg.generate_code(prob, code_dir=str(tmp_path / 'seq_cls'), solver='OSQP', wrapper=False)
    mod = cpg.load_generated(str(tmp_path / 'seq_cls'), prob)
    mod._SOLVER.lib_path = sim_lib
    ses = oracle_lib.CpgSession(d)
    seq = [({'a': 1000.0, 'lb': -1.0, 'ub': 1.0}, None), ({'ub': 1e30}, ['ub']), ({'ub': 2.0}, ['ub'])]
    rhos = []
    for k, (upd, names) in enumerate(seq):
        for nm, v in upd.items():
            prob.param_dict[nm].value = np.array(v)
        val = prob.solve(method='CPG', updated_params=names, eps_abs=1e-9, eps_rel=1e-9)
        o = ses.solve({nm: np.array(v) for nm, v in upd.items()}, eps_abs=1e-9, eps_rel=1e-9)
        assert prob._solution.attr['num_iters'] == o['iter'] and prob.status == 'solved', k
        assert abs(val - o['obj_val']) <= 1e-9 * max(1.0, abs(o['obj_val'])), k
        ws = mod._SOLVER._workspace()['state']
        assert abs(ws[0, -1] - o['rho']) <= 1e-9 * o['rho'], k
        rhos.append(o['rho'])
    assert rhos[0] > 1.0 and rhos[1] < 0.1                             # adapted up, then restarted at 0.1 and adapted down


def _three_call_sequence_with_a_rho_change(lib_path, oracle_lib, tmp_path, wrapper):
    """the reference's static workspace under OSQP >= 1.0 (solvers/osqp.py:100-101): every call resets the SETTINGS
    (rho back to 0.1), the workspace keeps the rho and factor of its last adapt_rho.  Call 1 adapts rho (MPC: the
    estimate leaves [rho / 5, 5 rho] at iteration 50), call 2 warm-starts on the adapted factor and compares its
    estimates with 0.1 again, call 3 cold-starts the iterates but not the workspace's rho."""
    d = families.mpc(6, 3, 10)
    prob = LiteProblem.from_descriptor(d)
    mod = cpg.generate_code(prob, code_dir=str(tmp_path / 'seq_rho'), solver='OSQP', wrapper=wrapper)
    if not wrapper:
        mod = cpg.load_generated(str(tmp_path / 'seq_rho'), prob)
        mod._SOLVER.lib_path = lib_path
    ses = oracle_lib.CpgSession(d)
    rng = np.random.default_rng(3)
    rhos = []
    for k, kw in enumerate(({}, {}, {'warm_start': False})):
        x0 = -2 + 4 * rng.random(6)
        prob.param_dict['x_init'].value = x0
        val = prob.solve(method='CPG', updated_params=['x_init'], **kw)
        o = ses.solve({'x_init': x0}, warm=bool(kw.get('warm_start', 1)))
        assert prob._solution.attr['num_iters'] == o['iter'] and prob.status == 'solved', k
        assert abs(val - o['obj_val']) <= 1e-6 * max(1.0, abs(o['obj_val'])), k
        for v in d.variables:
            got = np.ravel(prob.var_dict[v.name].value, order='F')
            assert np.abs(got - o['x'][v.indices]).max() <= 1e-6 * max(1.0, np.abs(o['x']).max()), (k, v.name)
        ws = mod._SOLVER._workspace()['state']
        assert abs(ws[0, -1] - o['rho']) <= 1e-9 * o['rho'], k              # the workspace's rho travels with the state
        rhos.append(o['rho'])
    assert rhos[0] != 0.1                                                    # call 1 did adapt


def test_three_call_sequence_with_a_rho_change(sim_lib, oracle_lib, tmp_path):
    _three_call_sequence_with_a_rho_change(sim_lib, oracle_lib, tmp_path, wrapper=False)


@pytest.mark.gpu
def test_three_call_sequence_with_a_rho_change_on_gpu(oracle_lib, tmp_path):
    _three_call_sequence_with_a_rho_change(None, oracle_lib, tmp_path, wrapper=True)
