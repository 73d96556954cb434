"""Row C1, exponential and three-dimensional power cones of the reference's Clarabel path
(`cvxpygen/solvers/clarabel.py:133-155, 308-323`: ClarabelExponentialConeT, ClarabelPowerConeT).  The reference's own tests hold no
problem with these cones, and Clarabel itself is absent (PARITY UNPINNED, oracle/clarabel_numpy.py): what pins the arithmetic here
is mathematics that does not depend on the restatement --

  * the barrier calculus: gradient / Hessian / third derivative of the dual barriers against finite differences, the primal
    gradient through the conjugacy identity grad f*(-grad f(s)) = -s, the secant equations of the primal-dual scaling block;
  * closed-form optima: softmax (entropy maximisation), Cobb-Douglas demand (geometric mean under a budget), the proximal
    point of sum exp through the Wright omega function;

then the kernel (lock-step emulator on the CPU tier, HIP on the GPU) against the oracle: the first iterations agree to rounding,
the final answers to the accuracy of the method.  The primal-dual scaling of a nonsymmetric cone is ill-conditioned next to the
central path (a 1e-15 perturbation of the data moves the iterates of the ORACLE by 1e-5 at termination and its iteration count
by one or two: test_oracle_sensitivity): final iterates are compared at 1e-4, objective values at 1e-7, iteration counts +-3.
"""
import os

import numpy as np
import pytest
from scipy.special import logsumexp, wrightomega

from cvxpygen_amd import families
from cvxpygen_amd.conic_plan import build_conic_plan
from cvxpygen_amd.conic_runtime import ConicBatchSolver
from oracle import clarabel_numpy as cl

np.seterr(all='ignore')


def _theta(desc, pv):
    B = next(iter(pv.values())).shape[0]
    return np.stack([desc.theta_from_values({k: v[i] for k, v in pv.items()}) for i in range(B)])


def _cases(B, seed):
    rs = np.random.RandomState(seed)
    return [(families.softmax_entropy(4), {'c': rs.randn(B, 4)}),
            (families.cobb_douglas(0.3), {'p': 0.5 + rs.rand(B, 2), 'budget': 1.0 + rs.rand(B)}),
            (families.exp_prox(3), {'a': rs.randn(B, 3), 'ub': 5.0 * np.ones((B, 3))}),
            (families.exp_prox(3, radius=1.0, name='exp_prox_active'), {'a': 2.0 * rs.randn(B, 3), 'ub': 0.3 * np.ones((B, 3))})]


def _assert_lockstep(r, o, tol):
    assert r.iter.tolist() == o['iter'].tolist() and r.status.tolist() == o['status'].tolist()
    assert np.abs(r.sol_x - o['sol_x']).max() <= tol * max(1.0, np.abs(o['sol_x']).max())
    assert np.abs(r.sol_y - o['sol_z']).max() <= tol * max(1.0, np.abs(o['sol_z']).max())


def _assert_final(r, o, xtol=1e-4, otol=1e-7, iters=3):
    assert r.status.tolist() == o['status'].tolist()
    assert np.abs(r.iter.astype(int) - o['iter'].astype(int)).max() <= iters
    ok = o['status'] == 1
    assert np.abs(r.sol_x[ok] - o['sol_x'][ok]).max() <= xtol * max(1.0, np.abs(o['sol_x'][ok]).max())
    assert np.abs(r.sol_y[ok] - o['sol_z'][ok]).max() <= xtol * max(1.0, np.abs(o['sol_z'][ok]).max())
    assert np.abs(r.obj_val[ok] - o['obj_val'][ok]).max() <= otol * max(1.0, np.abs(o['obj_val'][ok]).max())
    assert np.isnan(r.obj_val[~ok]).all()


# ------------------------------------------------------------------------------------ barrier calculus (independent of any solver)
def _rand_point(rs, alpha, dual):
    inside = cl.ns_dual_feasible if dual else cl.ns_primal_feasible
    while True:
        v = 2.0 * rs.randn(3)
        if dual and alpha is None:
            v[0], v[2] = -abs(v[0]), abs(v[2])
        if inside(v, alpha):
            return v


@pytest.mark.parametrize('alpha', [None, 0.3, 0.5, 0.85])
def test_barrier_calculus(alpha):
    rs = np.random.RandomState(3)
    h = 1e-6
    E = np.eye(3)
    for _ in range(25):
        z = _rand_point(rs, alpha, True)
        g, H = cl.ns_dual_grad_hess(z, alpha)
        gn = np.array([(cl.ns_barrier_dual(z + h * e, alpha) - cl.ns_barrier_dual(z - h * e, alpha)) / (2 * h) for e in E])
        Hn = np.array([(cl.ns_dual_grad_hess(z + h * e, alpha)[0] - cl.ns_dual_grad_hess(z - h * e, alpha)[0]) / (2 * h) for e in E])
        assert np.abs(gn - g).max() <= 1e-5 * max(1.0, np.abs(g).max())
        assert np.abs(Hn - H).max() <= 1e-5 * max(1.0, np.abs(H).max())
        assert np.linalg.eigvalsh(H).min() > 0.0
        assert abs(float(g @ z) + 3.0) <= 1e-10                  # logarithmic homogeneity, degree 3
        # third-order correction: 1/2 d/dt [hess f*(z + t dz)] u at t = 0, u = hess^-1 ds
        ds, dz = rs.randn(3), rs.randn(3)
        eta = cl.ns_higher_correction(z, alpha, ds, dz)
        u = np.linalg.solve(H, ds)
        Tn = (cl.ns_dual_grad_hess(z + h * dz, alpha)[1] - cl.ns_dual_grad_hess(z - h * dz, alpha)[1]) / (2 * h) @ u
        assert np.abs(0.5 * Tn - eta).max() <= 1e-4 * max(1.0, np.abs(eta).max())
        # primal barrier: conjugate of the dual one
        s = _rand_point(rs, alpha, False)
        gp = cl.ns_gradient_primal(s, alpha)
        assert cl.ns_dual_feasible(-gp, alpha)
        assert np.abs(cl.ns_dual_grad_hess(-gp, alpha)[0] + s).max() <= 1e-9 * max(1.0, np.abs(s).max())
        assert abs(cl.ns_barrier_primal(s, alpha) - (-3.0 - cl.ns_barrier_dual(-gp, alpha))) <= 1e-9
        gpn = np.array([(cl.ns_barrier_primal(s + h * e, alpha) - cl.ns_barrier_primal(s - h * e, alpha)) / (2 * h) for e in E])
        assert np.abs(gpn - gp).max() <= 1e-4 * max(1.0, np.abs(gp).max())
        # primal-dual scaling block: H_s z = s, H_s (z + mu grad f(s)) = s + mu grad f*(z), positive definite
        Hs = cl.ns_primal_dual_Hs(s, z, alpha, g, H)
        mu = float(s @ z) / 3.0
        assert np.abs(Hs @ z - s).max() <= 1e-8 * max(1.0, np.abs(s).max())
        assert np.abs(Hs @ (z + mu * gp) - (s + mu * g)).max() <= 1e-7 * max(1.0, np.abs(s).max(), np.abs(mu * g).max())
        assert np.linalg.eigvalsh(Hs).min() > 0.0
    # the central points the iteration starts from: s = z = -grad f*(z)
    c = np.array(cl.EXP_CENTRAL) if alpha is None else np.array([np.sqrt(1.0 + alpha), np.sqrt(2.0 - alpha), 0.0])
    assert np.abs(cl.ns_dual_grad_hess(c, alpha)[0] + c).max() <= 1e-8
    assert np.abs(cl.ns_gradient_primal(c, alpha) + c).max() <= 1e-8


# ------------------------------------------------------------------------------------ oracle against closed forms
def test_oracle_softmax_closed_form():
    d = families.softmax_entropy(5)
    assert d.cones == {'zero': 1, 'nonneg': 0, 'soc': [], 'exp': 5} and d.m == 16
    rs = np.random.RandomState(0)
    c = 1.5 * rs.randn(6, 5)
    o = cl.cpg_solve_batch(d, _theta(d, {'c': c}))
    assert (o['status'] == cl.SOLVED).all() and o['iter'].max() <= 15
    assert np.abs(o['obj_val'] + logsumexp(-c, axis=1)).max() <= 1e-8
    x = np.exp(-c) / np.exp(-c).sum(axis=1, keepdims=True)
    assert np.abs(o['prim']['x'] - x).max() <= 1e-4
    assert np.abs(o['dual']['nu'][:, 0] - (logsumexp(-c, axis=1) - 1.0)).max() <= 2e-4      # c_i + log x_i + 1 + nu = 0 (z = -nu)


def test_oracle_cobb_douglas_closed_form():
    rs = np.random.RandomState(1)
    for alpha in (0.3, 0.5, 0.9):
        d = families.cobb_douglas(alpha)
        assert d.cones['pow'] == [alpha]
        p, b = 0.5 + rs.rand(5, 2), 1.0 + rs.rand(5)
        o = cl.cpg_solve_batch(d, _theta(d, {'p': p, 'budget': b}))
        assert (o['status'] == cl.SOLVED).all()
        x, y = alpha * b / p[:, 0], (1.0 - alpha) * b / p[:, 1]
        assert np.abs(o['obj_val'] - x ** alpha * y ** (1.0 - alpha)).max() <= 1e-7
        assert np.abs(o['prim']['v'] - np.stack([x, y], axis=1)).max() <= 1e-3


def test_oracle_exp_prox_closed_form():
    d = families.exp_prox(3)
    assert d.cones == {'zero': 0, 'nonneg': 3, 'soc': [4], 'exp': 3}
    rs = np.random.RandomState(2)
    a = rs.randn(5, 3)
    o = cl.cpg_solve_batch(d, _theta(d, {'a': a, 'ub': 5.0 * np.ones((5, 3))}))
    assert (o['status'] == cl.SOLVED).all()
    xs = a - wrightomega(a).real                          # x + exp(x) = a
    assert np.abs(o['prim']['x'] - xs).max() <= 1e-4
    val = 2.0 * (np.exp(xs).sum(axis=1) + 0.5 * ((xs - a) ** 2).sum(axis=1)) - (a ** 2).sum(axis=1)
    assert np.abs(o['obj_val'] - val).max() <= 1e-7


def test_oracle_sensitivity():
    """why final iterates are compared at 1e-4: the ORACLE against itself on data perturbed in the last place"""
    d = families.softmax_entropy(4)
    c = np.random.RandomState(0).randn(4)
    a = cl.cpg_solve_batch(d, d.theta_from_values({'c': c})[None])
    b = cl.cpg_solve_batch(d, d.theta_from_values({'c': c * (1.0 + 1e-15)})[None])
    assert abs(int(a['iter'][0]) - int(b['iter'][0])) <= 3
    assert np.abs(a['sol_x'] - b['sol_x']).max() <= 1e-4
    assert abs(a['obj_val'][0] - b['obj_val'][0]) <= 1e-8
    a3 = cl.cpg_solve_batch(d, d.theta_from_values({'c': c})[None], max_iter=3)
    b3 = cl.cpg_solve_batch(d, d.theta_from_values({'c': c * (1.0 + 1e-15)})[None], max_iter=3)
    assert np.abs(a3['sol_x'] - b3['sol_x']).max() <= 1e-10


def test_plan_holds_the_scaling_blocks():
    d = families.exp_prox(3)
    cp = build_conic_plan(d)
    assert cp.n_exp == 3 and len(cp.pow_alpha) == 0 and list(cp.soc_dims) == [4]
    kinds = np.asarray(cp.ksrc_kind)
    assert (kinds == 7).sum() == 9 and (kinds == 6).sum() == 6           # three off-diagonals per exponential cone, C(4, 2) for the ball
    first = d.m - 9
    assert sorted(np.asarray(cp.ksrc_idx)[kinds == 7].tolist()) == list(range(first, d.m))
    d.cones['gen_pow'] = [3]                 # (a cone type the reference's array does not have)
    with pytest.raises(NotImplementedError, match='gen_pow'):
        build_conic_plan(d)


# ------------------------------------------------------------------------------------ emulator tier
def test_kernel_in_emulator_lockstep_and_final(sim_lib):
    for d, pv in _cases(3, 0):
        bs = ConicBatchSolver(d, lib_path=sim_lib, full_output=True)
        th = _theta(d, pv)
        for k in (1, 3):          # the first iterations, operation for operation
            r = bs.solve(pv, max_iter=k)
            o = cl.cpg_solve_batch(d, th, max_iter=k)
            assert (o['status'] == cl.MAX_ITERATIONS).all()
            _assert_lockstep(r, o, 1e-9)
        _assert_final(bs.solve(pv), cl.cpg_solve_batch(d, th))
        bs.close()


def test_dual_scaling_strategy_in_emulator(sim_lib):
    """min_switch_step_length above 1: the first small-step checkpoint moves every instance to the dual scaling H_s = mu H*(z)
    with the centrality test on the barrier sum -- the better-conditioned strategy, compared tightly to the end"""
    for d, pv in _cases(2, 1)[:3]:
        bs = ConicBatchSolver(d, lib_path=sim_lib, full_output=True)
        th = _theta(d, pv)
        r = bs.solve(pv, min_switch_step_length=1.1, max_iter=4)
        _assert_lockstep(r, cl.cpg_solve_batch(d, th, min_switch_step_length=1.1, max_iter=4), 1e-10)
        r = bs.solve(pv, min_switch_step_length=1.1)
        o = cl.cpg_solve_batch(d, th, min_switch_step_length=1.1)
        assert (o['status'] == cl.SOLVED).all()
        _assert_final(r, o, xtol=1e-6, iters=1)
        bs.close()


def test_infeasible_instances_in_emulator(sim_lib):
    d = families.cobb_douglas(0.3)
    pv = {'p': np.array([[1.0, 2.0], [1.0, 2.0]]), 'budget': np.array([-1.0, 2.0])}     # a negative budget: no x, y >= 0
    bs = ConicBatchSolver(d, lib_path=sim_lib, full_output=True)
    r = bs.solve(pv)
    o = cl.cpg_solve_batch(d, _theta(d, pv))
    assert o['status'].tolist() == [cl.PRIMAL_INFEASIBLE, cl.SOLVED]
    _assert_final(r, o)
    bs.close()
    d = families.exp_prox(3, radius=1.0)
    pv = {'a': np.zeros((2, 3)), 'ub': np.array([[-5.0, -5.0, -5.0], [1.0, 1.0, 1.0]])}       # x <= -5 outside the unit ball
    bs = ConicBatchSolver(d, lib_path=sim_lib, full_output=True)
    r = bs.solve(pv)
    o = cl.cpg_solve_batch(d, _theta(d, pv))
    assert o['status'].tolist() == [cl.PRIMAL_INFEASIBLE, cl.SOLVED]
    _assert_final(r, o)
    bs.close()


def _fact(bs, name):
    import ctypes as C
    v = C.c_double(-1)
    bs.lib.check(bs.lib.L.cpg_hip_get_setting(bs.h, name.encode(), C.byref(v)), 'get_setting')
    return v.value


def test_generated_family_library_in_emulator(tmp_path):
    """the family's own library (generated executor, factorisation with the scaling-block sources, row words, dimensions
    compiled in) gives the table-driven kernel's results bit for bit"""
    from tests.sim import build_sim
    d, pv = _cases(3, 4)[2]
    cp = build_conic_plan(d)
    lib = build_sim.build_conic_family(cp, str(tmp_path), 'exp_prox')
    fh = open(os.path.join(str(tmp_path), 'cpg_conic_exp_prox_factor.h')).read()
    assert 'CPG_CK_HNS' in fh
    bs = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
    r = bs.solve(pv)
    assert _fact(bs, 'generated_executor') == 1.0 and _fact(bs, 'specialised_kernel') == 1.0
    bg = ConicBatchSolver(d, lib_path=build_sim.build(), plan=cp, full_output=True)
    rg = bg.solve(pv)
    assert np.array_equal(r.sol_x, rg.sol_x) and np.array_equal(r.sol_y, rg.sol_y)
    assert r.iter.tolist() == rg.iter.tolist() and r.status.tolist() == rg.status.tolist()
    _assert_final(r, cl.cpg_solve_batch(d, _theta(d, pv)))
    bs.close(); bg.close()


def test_c_abi_refuses_bad_cone_counts(sim_lib):
    d = families.cobb_douglas(0.3)
    cp = build_conic_plan(d)
    cp.pow_alpha = np.array([1.5])
    bs = ConicBatchSolver(d, lib_path=sim_lib, plan=cp)
    with pytest.raises(RuntimeError, match='exponent'):
        bs.solve({'p': np.ones((1, 2)), 'budget': np.ones(1)})


# ------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_kernel_on_gpu_lockstep_and_final():
    for d, pv in _cases(64, 5):
        bs = ConicBatchSolver(d, full_output=True)
        th = _theta(d, pv)
        r = bs.solve(pv, max_iter=2)
        _assert_lockstep(r, cl.cpg_solve_batch(d, th, max_iter=2), 1e-8)
        _assert_final(bs.solve(pv), cl.cpg_solve_batch(d, th))
        # dual scaling strategy from the first checkpoint on.  Forced like this it is the less robust one (the oracle itself ends
        # 0.5 % of softmax instances "almost solved" / on insufficient progress) and its centrality backtracking is a discrete
        # decision: instances solved on both sides are compared, a borderline instance may end differently
        r = bs.solve(pv, min_switch_step_length=1.1)
        o = cl.cpg_solve_batch(d, th, min_switch_step_length=1.1)
        both = (r.status == 1) & (o['status'] == 1)
        assert both.sum() >= len(both) - 3 and (r.status != o['status']).sum() <= 3
        assert np.abs(r.iter[both].astype(int) - o['iter'][both].astype(int)).max() <= 8        # (the centrality backtracking decides at a threshold)
        assert np.abs(r.sol_x[both] - o['sol_x'][both]).max() <= 1e-5 * max(1.0, np.abs(o['sol_x'][both]).max())
        assert np.abs(r.obj_val[both] - o['obj_val'][both]).max() <= 1e-7 * max(1.0, np.abs(o['obj_val'][both]).max())
        bs.close()


@pytest.mark.gpu
def test_closed_forms_on_gpu_at_batch_size():
    """20 000 instances per family against the closed forms (no oracle in the loop).  A handful in 10^5 end "almost solved" (status 4:
    insufficient progress next to the optimum, accepted at the reduced tolerances) -- the oracle does the same on the same data
    (1 of these 20 000 softmax instances, objective error 1e-10)"""

    def solved(r):
        assert np.isin(r.status, (1, 4)).all() and (r.status == 1).mean() >= 0.999

    B = 20000
    rs = np.random.RandomState(6)
    d = families.softmax_entropy(4)
    c = 1.5 * rs.randn(B, 4)
    bs = ConicBatchSolver(d)
    r = bs.solve({'c': c})
    solved(r)
    assert r.iter.max() <= 25
    assert np.abs(r.obj_val + logsumexp(-c, axis=1)).max() <= 1e-7
    assert np.abs(r.prim['x'] - np.exp(-c) / np.exp(-c).sum(axis=1, keepdims=True)).max() <= 1e-3
    bs.close()
    d = families.cobb_douglas(0.3)
    p, b = 0.5 + rs.rand(B, 2), 1.0 + rs.rand(B)
    bs = ConicBatchSolver(d)
    r = bs.solve({'p': p, 'budget': b})
    solved(r)
    x, y = 0.3 * b / p[:, 0], 0.7 * b / p[:, 1]
    assert np.abs(r.obj_val - x ** 0.3 * y ** 0.7).max() <= 1e-6
    bs.close()
    d = families.exp_prox(3)
    a = rs.randn(B, 3)
    bs = ConicBatchSolver(d)
    r = bs.solve({'a': a, 'ub': 5.0 * np.ones((B, 3))})
    solved(r)
    xs = a - wrightomega(a).real
    assert np.abs(r.prim['x'] - xs).max() <= 1e-3
    val = 2.0 * (np.exp(xs).sum(axis=1) + 0.5 * ((xs - a) ** 2).sum(axis=1)) - (a ** 2).sum(axis=1)
    assert np.abs(r.obj_val - val).max() <= 1e-6
    bs.close()


@pytest.mark.gpu
def test_generated_family_library_on_gpu():
    from cvxpygen_amd import codegen
    d, pv = _cases(2000, 7)[2]
    cp = build_conic_plan(d)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = codegen.build_conic_library(cp, os.path.join(root, 'cvxpygen_amd', 'generated', 'exp_prox'), 'exp_prox')   # no-op when fresh
    bs = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
    r = bs.solve(pv)
    assert _fact(bs, 'generated_executor') == 1.0 and _fact(bs, 'specialised_kernel') == 1.0
    bg = ConicBatchSolver(d, plan=cp, full_output=True)
    rg = bg.solve(pv)
    assert np.array_equal(r.sol_x, rg.sol_x) and r.iter.tolist() == rg.iter.tolist() and r.status.tolist() == rg.status.tolist()
    xs = pv['a'] - wrightomega(pv['a']).real
    assert np.abs(r.sol_x[:, :3] - xs).max() <= 1e-3
    bs.close(); bg.close()


def test_generate_code_drop_in_with_exponential_cones(sim_lib, tmp_path):
    """generate_code(prob, solver='CLARABEL') -> prob.solve(method='CPG') on a family with exponential cones (the shape of
    tests/test_E2E_SOCP.py:119-121), values against the closed form"""
    from cvxpygen_amd import cpg
    from cvxpygen_amd.lite import LiteProblem
    d = families.softmax_entropy(4)
    prob = LiteProblem.from_descriptor(d)
    mod = cpg.generate_code(prob, code_dir=str(tmp_path / 'softmax_CLARABEL'), solver='CLARABEL', prefix='softmax')
    mod._SOLVER.lib_path = sim_lib
    c = np.array([0.2, -0.7, 1.1, 0.4])
    prob.param_dict['c'].value = c
    val = prob.solve(method='CPG')
    assert abs(val + logsumexp(-c)) <= 1e-8 and prob.status.startswith('1 ')
    assert np.abs(prob.var_dict['x'].value - np.exp(-c) / np.exp(-c).sum()).max() <= 1e-4
    assert prob.solver_stats.solver_name == 'CLARABEL' and prob.solver_stats.num_iters <= 15
