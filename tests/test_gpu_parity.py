"""GPU tier (-m gpu): the HIP path through the C-ABI against the oracle on seeded inputs, against
the committed known answers, and -- at BASELINE.json's full batch size -- through size-independent
properties (batch-order invariance, duplicates, termination criteria)."""
import json
import os

import numpy as np
import pytest

from cvxpygen_amd import families
from cvxpygen_amd.runtime import BatchSolver, DeviceBatch

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'known_answers.json')))
REL_TOL = 1e-6          # north_star: 1e-6 relative on primal / dual vectors, identical iteration counts


def _theta(desc, name, values):
    th = np.tile(desc.theta0, (values.shape[0], 1))
    p = desc.param(name)
    th[:, p.col:p.col + p.size] = values
    return th


def _check(r, o, desc, tol=REL_TOL):
    prim = np.concatenate([o['sol_x'][:, v.indices] for v in desc.variables], axis=1)
    dual = np.concatenate([o['sol_y'][:, d.indices] for d in desc.duals], axis=1)
    assert (r.iter == o['iter']).all(), f'{int((r.iter != o["iter"]).sum())} iteration-count mismatches'
    assert (r.status == o['status']).all()
    assert np.abs(r.prim_flat - prim).max() <= tol * np.abs(prim).max()
    assert np.abs(r.dual_flat - dual).max() <= tol * np.abs(dual).max()
    assert np.abs(r.obj_val - o['obj_val']).max() <= tol * np.abs(o['obj_val']).max()


@pytest.mark.parametrize('G', [1, 2])
def test_nonneg_ls_vs_oracle(oracle_lib, G):
    d = families.nonneg_ls()
    rng = np.random.default_rng(0)
    b = rng.standard_normal((1001, 3))                 # ragged: not a multiple of anything
    bs = BatchSolver(d)
    bs.set_launch(0, G, 0)
    for stg in ({}, dict(eps_abs=1e-8, eps_rel=1e-8), dict(max_iter=60)):
        r = bs.solve({'b': b}, updated_params=['b'], **stg)
        _check(r, oracle_lib.cpg_solve_batch(d, _theta(d, 'b', b), ['b'], **stg), d)
    bs.close()


def test_known_answers_on_gpu():
    d = families.nonneg_ls()
    g = GOLD['nonneg_LS']
    bs = BatchSolver(d)
    r = bs.solve({'b': np.array([g['b']])}, updated_params=['b'], eps_abs=1e-10, eps_rel=1e-10)
    assert r.status[0] == 1
    assert np.allclose(r.prim['x'][0], g['x'], atol=1e-8)
    assert np.allclose(r.dual['d0'][0], g['dual_x_ge_0'], atol=1e-7)
    assert abs(r.obj_val[0] - g['obj']) < 1e-7
    bs.close()
    d = families.mpc(6, 3, 10)
    g = GOLD['MPC_6_3_10']
    bs = BatchSolver(d)
    r = bs.solve({'x_init': np.array([g['x_init']])}, updated_params=['x_init'], eps_abs=1e-9,
                 eps_rel=1e-9, max_iter=20000)
    assert r.status[0] == 1 and abs(r.obj_val[0] - g['obj']) / g['obj'] < 1e-7
    assert np.allclose(r.prim['U'][0], np.array(g['U']), atol=1e-6)
    bs.close()


@pytest.mark.parametrize('n,m,B,G', [(6, 3, 512, 1), (6, 3, 257, 2), (12, 4, 384, 1), (12, 4, 255, 2)])
def test_mpc_vs_oracle(oracle_lib, n, m, B, G):
    d = families.mpc(n, m, 10)
    rng = np.random.default_rng(11)
    x0 = -2 + 4 * rng.random((B, n))
    bs = BatchSolver(d)
    bs.set_launch(0, G, 0)
    r = bs.solve({'x_init': x0}, updated_params=['x_init'])
    _check(r, oracle_lib.cpg_solve_batch(d, _theta(d, 'x_init', x0), ['x_init']), d)
    # tight tolerances: the iterates agree long after the 1e-3 stopping point
    r = bs.solve({'x_init': x0[:64]}, updated_params=['x_init'], eps_abs=1e-7, eps_rel=1e-7)
    _check(r, oracle_lib.cpg_solve_batch(d, _theta(d, 'x_init', x0[:64]), ['x_init'],
                                         eps_abs=1e-7, eps_rel=1e-7), d)
    bs.close()


def test_infeasible_and_empty_batches(oracle_lib):
    d = families.toy_box()
    B = 130
    th = np.tile(d.theta0, (B, 1))
    rng = np.random.default_rng(3)
    th[:, d.param('a').col] = 3 * rng.standard_normal(B)
    bad = rng.random(B) < 0.3
    th[bad, d.param('lb').col] = 2.0
    th[bad, d.param('ub').col] = 1.0
    bs = BatchSolver(d)
    vals = {p.name: th[:, p.col:p.col + p.size] for p in d.params}
    r = bs.solve(vals)
    o = oracle_lib.cpg_solve_batch(d, th, None)
    assert (r.status == o['status']).all() and (r.iter == o['iter']).all()
    assert (r.status[bad] == 3).all() and np.isnan(r.prim_flat[bad]).all() and (r.obj_val[bad] == np.inf).all()
    assert np.allclose(r.prim_flat[~bad], o['sol_x'][~bad][:, d.variables[0].indices], atol=1e-9)
    r0 = bs.solve({p.name: np.zeros((0, p.size)) for p in d.params}, B=0)       # empty batch
    assert r0.iter.shape == (0,)
    bs.close()


def test_full_size_properties(oracle_lib):
    """100 000 instances of the benchmark family: properties that need no oracle, and a random sample of 512 instances drawn
    from inside the big batch against the oracle (counts exact, 1e-6)."""
    d = families.mpc(12, 4, 10)
    B = 100000
    rng = np.random.default_rng(5)
    x0 = -2 + 4 * rng.random((B, 12))
    x0[1] = x0[0]                                              # duplicates
    bs = BatchSolver(d)
    r = bs.solve({'x_init': x0}, updated_params=['x_init'])
    assert (r.status == 1).all()
    assert (r.iter % 25 == 0).all() and r.iter.max() <= 4000
    # duplicates give bit-identical results
    assert np.array_equal(r.prim_flat[0], r.prim_flat[1]) and r.iter[0] == r.iter[1]
    # batch-order invariance: reversing the batch reverses the results bit for bit
    rr = bs.solve({'x_init': x0[::-1].copy()}, updated_params=['x_init'])
    assert np.array_equal(rr.prim_flat[::-1], r.prim_flat) and np.array_equal(rr.iter[::-1], r.iter)
    assert np.array_equal(rr.dual_flat[::-1], r.dual_flat)
    # a random sample of the big batch against the oracle: what ran at 100 000 is what the oracle computes, not merely self-consistent
    pick = np.sort(np.random.default_rng(6).choice(B, 512, replace=False))
    sub = type('R', (), dict(iter=r.iter[pick], status=r.status[pick], prim_flat=r.prim_flat[pick], dual_flat=r.dual_flat[pick],
                             obj_val=r.obj_val[pick], pri_res=r.pri_res[pick], dua_res=r.dua_res[pick]))()
    _check(sub, oracle_lib.cpg_solve_batch(d, _theta(d, 'x_init', x0[pick]), ['x_init']), d)
    # the constraint X[:,0] == x_init holds to the ADMM tolerance; |U| <= 1 likewise
    assert np.abs(r.prim['X'][:, :, 0] - x0).max() < 5e-2
    assert np.abs(r.prim['U']).max() < 1 + 5e-2
    # device-resident entry point gives the same bits as the host-pointer one
    dev = DeviceBatch(bs, 4096)
    dev.upload(x0[:4096])
    bs.solve_device(dev); bs.synchronize()
    rd = dev.download()
    assert np.array_equal(rd.prim_flat, r.prim_flat[:4096]) and np.array_equal(rd.iter, r.iter[:4096])
    dev.free(); bs.close()


def test_generated_family_library_vs_oracle(oracle_lib, tmp_path):
    """what generate_code() compiles: executor specialised for the family (cvxpygen_amd/codegen.py)"""
    from cvxpygen_amd import codegen
    from cvxpygen_amd.runtime import build_family_plan
    d = families.mpc(12, 4, 10)
    plan = build_family_plan(d)
    pre = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cvxpygen_amd',
                       'generated', 'mpc12', 'libcpg_mpc12.so')
    lib = pre if os.path.exists(pre) else codegen.build_family_library(plan, str(tmp_path), 'mpc12')
    rng = np.random.default_rng(21)
    x0 = -2 + 4 * rng.random((777, 12))
    for G in (1, 2):
        bs = BatchSolver(d, lib_path=lib, plan=plan)
        bs.set_launch(0, G, 0)
        bs.set_program_placement(-1)       # "automatic" (what bench.py passes): the library's own choice
        r = bs.solve({'x_init': x0}, updated_params=['x_init'])
        _check(r, oracle_lib.cpg_solve_batch(d, _theta(d, 'x_init', x0), ['x_init']), d)
        bs.close()
    # full batch, several launches in a row (stale LDS contents between launches): the specialised
    # executor must agree with the table-driven kernels instance by instance
    x0 = -2 + 4 * np.random.default_rng(22).random((100_000, 12))
    ref = BatchSolver(d, plan=plan)
    r0 = ref.solve({'x_init': x0}, updated_params=['x_init'])
    ref.close()
    bs = BatchSolver(d, lib_path=lib, plan=plan)
    for _ in range(3):
        r = bs.solve({'x_init': x0}, updated_params=['x_init'])
        assert (r.status == 1).all()
        assert (r.iter == r0.iter).all()
        assert np.abs(r.prim_flat - r0.prim_flat).max() <= 1e-9 * np.abs(r0.prim_flat).max()
    # ... and a random 512 of the 100 000 against the oracle (the comparison above is GPU against GPU)
    pick = np.sort(np.random.default_rng(23).choice(len(x0), 512, replace=False))
    sub = type('R', (), dict(iter=r.iter[pick], status=r.status[pick], prim_flat=r.prim_flat[pick], dual_flat=r.dual_flat[pick],
                             obj_val=r.obj_val[pick]))()
    _check(sub, oracle_lib.cpg_solve_batch(d, _theta(d, 'x_init', x0[pick]), ['x_init']), d)
    bs.close()


def test_squad_executor_on_gpu(oracle_lib, tmp_path):
    """csrc/cpg_osqp_squad.h (round 6; placement 3 of a family library): the family's solve program in the registers of a squad
    of four wavefronts that solves four instances at a time -- against the oracle on a ragged batch (the last squad is not
    full), in both forks of the default, and at the full batch size instance by instance against the LDS-resident executor of
    the same library"""
    import ctypes as C
    from cvxpygen_amd import codegen
    from cvxpygen_amd.runtime import build_family_plan, BUILD_OPTIONS_FIXED_RHO
    d = families.mpc(12, 4, 10)
    plan = build_family_plan(d)
    assert plan.kkt_squad is not None
    pre = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cvxpygen_amd',
                       'generated', 'mpc12', 'libcpg_mpc12.so')
    lib = pre if os.path.exists(pre) else codegen.build_family_library(plan, str(tmp_path), 'mpc12')
    x0 = -2 + 4 * np.random.default_rng(31).random((1001, 12))
    for bo, mode in (({}, {}), (dict(BUILD_OPTIONS_FIXED_RHO), dict(adaptive_rho=0, check_dualgap=0))):
        bs = BatchSolver(d, lib_path=lib, plan=plan, build_options=bo)
        bs.set_program_placement(3)
        r = bs.solve({'x_init': x0}, updated_params=['x_init'])
        v = C.c_double(0)
        bs.lib.L.cpg_hip_get_setting(bs.h_shared, b'squad_executor', C.byref(v))
        assert v.value == 1.0
        _check(r, oracle_lib.cpg_solve_batch(d, _theta(d, 'x_init', x0), ['x_init'], **mode), d)
        bs.close()
    x0 = -2 + 4 * np.random.default_rng(32).random((100_000, 12))
    res = []
    for placement in (1, 3):
        bs = BatchSolver(d, lib_path=lib, plan=plan)
        bs.set_program_placement(placement)
        res.append(bs.solve({'x_init': x0}, updated_params=['x_init']))
        bs.close()
    assert (res[1].status == 1).all() and (res[0].iter == res[1].iter).all()
    assert np.abs(res[0].prim_flat - res[1].prim_flat).max() <= 1e-9 * np.abs(res[0].prim_flat).max()


@pytest.mark.parametrize('name', ['mpc8', 'nnls40'])
def test_generated_executor_other_shapes(oracle_lib, tmp_path, name):
    """two more families through the generated executor (libraries built by __graft_entry__.build): step and chunk
    counts that are not multiples of four (grouped offsets / output-slot table padding), other phase structures"""
    from cvxpygen_amd import codegen
    from cvxpygen_amd.runtime import build_family_plan
    rng = np.random.default_rng(31)
    if name == 'mpc8':
        d = families.mpc(8, 3, 7)
        pname, vals = 'x_init', -2 + 4 * rng.random((300, 8))
    else:
        d = families.nonneg_ls(40, 20, sparsity=None, seed=1)
        pname, vals = 'b', d.theta0[d.param('b').col:d.param('b').col + 40] * (1 + 0.3 * rng.standard_normal((300, 40)))
    plan = build_family_plan(d)
    pre = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cvxpygen_amd', 'generated', name,
                       f'libcpg_{name}.so')
    lib = pre if os.path.exists(pre) else codegen.build_family_library(plan, str(tmp_path), name)
    o = oracle_lib.cpg_solve_batch(d, _theta(d, pname, vals), [pname])
    for G in (1, 2):
        bs = BatchSolver(d, lib_path=lib, plan=plan)
        bs.set_launch(0, G, 0)
        r = bs.solve({pname: vals}, updated_params=[pname])
        _check(r, o, d)
        bs.close()


@pytest.mark.parametrize('make,B', [(lambda: families.nonneg_ls(10, 5, sparsity=None, seed=0), 200),
                                    (lambda: families.mpc(6, 3, 10, sparse_params=True, terminal_index=9, const=1.0), 96),
                                    (lambda: families.mpc(6, 3, 10), 64),
                                    (lambda: families.mpc(12, 4, 10), 48)])
def test_refactor_path_vs_oracle(oracle_lib, make, B):
    """every parameter varies per instance (matrices included): reference osqp_update_data_mat path"""
    d = make()
    rng = np.random.default_rng(17)
    th = np.tile(d.theta0, (B, 1))
    th[:, :d.NP] *= 1 + 0.1 * rng.standard_normal((B, d.NP))
    if 'x_init' in d.param_names:
        p = d.param('x_init')
        th[:, p.col:p.col + p.size] = -2 + 4 * rng.random((B, p.size))
    bs = BatchSolver(d)
    vals = {p.name: th[:, p.col:p.col + p.size] for p in d.params}
    r = bs.solve(vals)
    _check(r, oracle_lib.cpg_solve_batch(d, th, None), d)
    bs.close()


def test_large_family_beyond_the_generic_slot_classes(oracle_lib):
    """MPC 12/4 with horizon 30: n_var = 1104, m = 1224 (KKT dimension 2328) -- more than the 16 x 16
    slot classes (1024) of the generic library, and a solve program (294 KB) that cannot be LDS
    resident: BatchSolver compiles the table-driven kernels for the family's own class
    (codegen.build_streamed_family_library); shared-factor and per-instance-factor paths vs oracle"""
    d = families.mpc(12, 4, 30)
    rng = np.random.default_rng(5)
    p = d.param('x_init')
    bs = BatchSolver(d)
    assert bs.lib.path.endswith('_streamed.so')
    B = 12
    th = np.tile(d.theta0, (B, 1)); th[:, p.col:p.col + p.size] = -2 + 4 * rng.random((B, p.size))
    r = bs.solve({'x_init': th[:, p.col:p.col + p.size]}, updated_params=['x_init'])
    _check(r, oracle_lib.cpg_solve_batch(d, th, ['x_init']), d)
    B = 4
    th = np.tile(d.theta0, (B, 1)); th[:, :d.NP] *= 1 + 0.05 * rng.standard_normal((B, d.NP))
    th[:, p.col:p.col + p.size] = -2 + 4 * rng.random((B, p.size))
    r = bs.solve({q.name: th[:, q.col:q.col + q.size] for q in d.params})
    _check(r, oracle_lib.cpg_solve_batch(d, th, None), d)
    bs.close()


@pytest.mark.parametrize('make,B,upd', [
    (lambda: families.nonneg_ls(10, 5, sparsity=None, seed=0), 300, None),        # tests/test_diff.py family
    (lambda: families.mpc(6, 3, 10), 24, ['x_init']),                             # BASELINE config 5 shape
    (lambda: families.mpc(12, 4, 10), 8, ['x_init'])])
def test_adjoint_vs_oracle(oracle_lib, make, B, upd):
    """gradient=True: forward solve with canonical output, then the batched adjoint kernel; compared
    with the C restatement of cpg_osqp_gradient on the SAME forward solution"""
    d = make()
    rng = np.random.default_rng(23)
    th = np.tile(d.theta0, (B, 1))
    if upd is None:
        th[:, :d.NP] *= 1 + 0.05 * rng.standard_normal((B, d.NP))
        names = d.param_names
    else:
        p = d.param('x_init')
        th[:, p.col:p.col + p.size] = -2 + 4 * rng.random((B, p.size))
        names = upd
    vals = {nm: th[:, d.param(nm).col:d.param(nm).col + d.param(nm).size] for nm in names}
    bs = BatchSolver(d, full_output=True)
    r = bs.solve(vals, updated_params=names, eps_abs=1e-6, eps_rel=1e-6)
    assert (r.status == 1).all()
    dv = {v.name: 0.1 * np.ones((B,) + tuple(v.shape)) for v in d.variables}      # loss = 0.1 * sum(vars)
    g = bs.gradient(vals, r.sol_x, r.sol_y, dv, updated_params=names)
    wts = np.zeros(d.n_var)
    for v in d.variables:
        wts[v.indices] = 0.1
    for k in range(B):
        go = oracle_lib.qp_adjoint(d, d.canon_at(th[k]), r.sol_x[k], r.sol_y[k], wts)
        # instances whose gradient vanishes identically (solution at a vertex) need an absolute floor
        assert np.abs(g['_flat'][k] - go['dtheta']).max() <= 1e-6 * np.abs(go['dtheta']).max() + 1e-10
    bs.close()


def test_portfolio_config3_vs_oracle(oracle_lib):
    """BASELINE config 3 (examples/portfolio.ipynb, n=100, m=10): a, F, Sig_f_sqrt, d_sqrt, w_prev per
    instance -> matrix parameters -> per-instance refactorisation path; maximisation problem"""
    d = families.portfolio(100, 10)
    B = 48
    rng = np.random.default_rng(31)
    sig = np.zeros((B, 10, 10)); sig[:, np.arange(10), np.arange(10)] = rng.random((B, 10))
    pv = {'a': rng.standard_normal((B, 100)), 'F': np.round(rng.standard_normal((B, 100, 10))),
          'Sig_f_sqrt': sig, 'd_sqrt': rng.random((B, 100)), 'w_prev': np.zeros((B, 100))}
    th = np.tile(d.theta0, (B, 1))
    for k in range(B):
        th[k] = d.theta_from_values({nm: v[k] for nm, v in pv.items()})
    bs = BatchSolver(d)
    r = bs.solve(pv, updated_params=list(pv.keys()))
    o = oracle_lib.cpg_solve_batch(d, th, list(pv.keys()))
    _check(r, o, d)
    assert r.prim['w'].shape == (B, 100) and r.dual['d3'].shape == (B, 100)
    assert np.abs(r.prim['w'].sum(axis=1) - 1).max() < 1e-2          # 1'w == 1
    bs.close()


def test_actuator_parameter_in_P_vs_oracle(oracle_lib):
    """tests/test_E2E_QP.py:14-41, 104-112: the family whose P depends on a parameter (lamb_sm)"""
    d = families.actuator()
    B = 257
    rng = np.random.default_rng(5)
    th = np.tile(d.theta0, (B, 1))
    th[:, d.param('lamb_sm').col] = rng.random(B)
    th[:, d.param('w').col:d.param('w').col + 3] += rng.standard_normal((B, 3))
    th[:, d.param('kappa').col] = 0.1 + 0.2 * rng.random(B)
    bs = BatchSolver(d)
    r = bs.solve({p.name: th[:, p.col:p.col + p.size] for p in d.params})
    _check(r, oracle_lib.cpg_solve_batch(d, th, None), d)
    # reference data of seed 0: u is clipped at u_max = 1, objective 21 + lamb_sm + 0.1
    r0 = bs.solve({p.name: d.theta0[None, p.col:p.col + p.size] for p in d.params})
    lam = d.theta0[d.param('lamb_sm').col]
    assert abs(r0.prim["u"][0, 0] - 1.0) < 2e-2 and abs(r0.obj_val[0] - (21.1 + lam)) < 0.1
    bs.close()


def test_portfolio_full_shard_properties():
    """config 3 at shard scale (20 000 instances here; the kernel's memory does not grow with B):
    every instance solved, budget and leverage constraints hold to the solver tolerance,
    batch-order invariance and duplicates"""
    d = families.portfolio(100, 10)
    B = 20_000
    rng = np.random.default_rng(41)
    sig = np.zeros((B, 10, 10)); sig[:, np.arange(10), np.arange(10)] = rng.random((B, 10))
    pv = {'a': rng.standard_normal((B, 100)), 'F': np.round(rng.standard_normal((B, 100, 10))),
          'Sig_f_sqrt': sig, 'd_sqrt': rng.random((B, 100)), 'w_prev': np.zeros((B, 100))}
    pv['a'][1] = pv['a'][0]; pv['F'][1] = pv['F'][0]; pv['Sig_f_sqrt'][1] = pv['Sig_f_sqrt'][0]; pv['d_sqrt'][1] = pv['d_sqrt'][0]
    bs = BatchSolver(d)
    r = bs.solve(pv, updated_params=list(pv.keys()))
    assert (r.status == 1).all()
    w = r.prim['w']
    # ||w||_1 <= L is 100 epigraph rows |w_i| <= t_i plus sum(t) <= L, each met to ~eps * ||Ax||: the sum may exceed L by ~0.2
    assert np.abs(w.sum(axis=1) - 1).max() < 2e-2 and (np.abs(w).sum(axis=1) <= 1.6 + 0.3).all()
    assert r.iter[0] == r.iter[1] and np.array_equal(w[0], w[1])
    perm = rng.permutation(B)[:4000]
    r2 = bs.solve({k: v[perm] for k, v in pv.items()}, updated_params=list(pv.keys()))
    assert (r2.iter == r.iter[perm]).all() and np.array_equal(r2.prim_flat, r.prim_flat[perm])
    bs.close()
