"""ECOS-form families (cvxpygen/solvers/ecos.py: c, d, A, b, G, h; y / z duals; ECOS setting names and exit flags)
on the conic interior-point kernel -- cvxpygen_amd/ecos_front.py.  Parity is at the level of the optimisation problem
(see the module header): the oracle is the conic restatement on the stacked form."""
import numpy as np
import pytest

from cvxpygen_amd import cpg, families
from cvxpygen_amd.ecos_front import (ECOS_SETTINGS, EcosBatchSolver, conic_from_ecos, ecos_from_conic)
from cvxpygen_amd.lite import LiteProblem


def _batch(B, seed=0):
    rng = np.random.default_rng(seed)
    states = -2 + 4 * rng.random((B, 6))
    return {k: np.stack([families.adp_values(s_)[k] for s_ in states]) for k in ('f', 'G')}


@pytest.mark.parametrize('tie', [True, False])
def test_ecos_form_round_trip(tie):
    """host logic: split into (A, b) / (G, h) and back; p = 0 (no equality block) included"""
    c = families.adp_norm(tie=tie)
    e = ecos_from_conic(c)
    p = 3 if tie else 0
    assert e.solver == 'ECOS' and set(e.maps) == {'c', 'd', 'A', 'b', 'G', 'h'} and e.n_eq == p and e.n_ineq == c.m - p
    assert e.cones == {'zero': p, 'nonneg': 2, 'soc': [7, 4, 4, 4]}
    assert [u.vec for u in e.duals] == (['z', 'y'] if tie else ['z'])
    th = c.theta0.copy(); th[:c.NP] += 0.1 * np.random.default_rng(1).standard_normal(c.NP)
    back = conic_from_ecos(e)
    for pid in ('A', 'b', 'q', 'd'):
        assert np.array_equal(back.canon_at(th)[pid], c.canon_at(th)[pid])
    ce = e.canon_at(th)
    A = c.A.copy(); A.data = c.canon_at(th)['A']
    import scipy.sparse as sp
    Ad = sp.csc_matrix(A).toarray()
    blkA = sp.csc_matrix((ce['A'], *_pattern(Ad[:p])), shape=(p, c.n_var)).toarray() if p else np.zeros((0, c.n_var))
    blkG = sp.csc_matrix((ce['G'], *_pattern(Ad[p:])), shape=(c.m - p, c.n_var)).toarray()
    assert np.array_equal(np.vstack([blkA, blkG]), Ad)
    assert np.array_equal(np.concatenate([ce['b'], ce['h']]), c.canon_at(th)['b'])
    with pytest.raises(ValueError, match='no quadratic objective'):
        ecos_from_conic(families.adp())                     # sum_squares objective: P != 0


def _pattern(M):
    """CSC (indices, indptr) of the structural pattern used above: explicit entries of the stacked pattern that
    fall into the block -- rebuilt from the dense block, which is fine here because no stored entry is zero"""
    import scipy.sparse as sp
    S = sp.csc_matrix(M)
    return S.indices, S.indptr


def test_ecos_solver_in_emulator(sim_lib, oracle_lib):
    from oracle import clarabel_numpy
    c = families.adp_norm()
    e = ecos_from_conic(c)
    vals = _batch(3)
    es = EcosBatchSolver(e, lib_path=sim_lib)
    r = es.solve(vals, updated_params=['f', 'G'])
    assert (r.status == 0).all() and r.iter.max() <= 30                         # ECOS_OPTIMAL
    assert r.dual['d0'].shape == (3, 2) and r.dual['d1'].shape == (3, 3) and r.prim['u'].shape == (3, 2, 3)
    # oracle: the conic restatement on the stacked form with ECOS's tolerances
    stg = {knl: dflt for knl, dflt in ECOS_SETTINGS.values()}
    th = np.tile(c.theta0, (3, 1))
    for nm in ('f', 'G'):
        q = c.param(nm)
        th[:, q.col:q.col + q.size] = np.stack([c.flatten_param(nm, vals[nm][k]) for k in range(3)])
    o = clarabel_numpy.cpg_solve_batch(c, th, **stg)
    assert r.iter.tolist() == o['iter'].tolist() and (o['status'] == 1).all()
    # (the epigraph variables tn_i of inactive norm bounds are not unique: those components agree to 1e-8 only)
    assert np.abs(r.sol_x - o['sol_x']).max() <= 1e-6 * max(1.0, np.abs(o['sol_x']).max())
    assert np.abs(r.sol_y - o['sol_z']).max() <= 1e-6 * max(1.0, np.abs(o['sol_z']).max())
    assert np.abs(r.obj_val - o['obj_val']).max() <= 1e-9 * max(1.0, np.abs(o['obj_val']).max())
    # ECOS's setting names; unknown names are refused like cpg_set_solver_<name>
    r2 = es.solve(vals, updated_params=['f', 'G'], maxit=2)
    assert (r2.status == -1).all() and (r2.iter == 2).all()                      # ECOS_MAXIT
    es.solve(vals, updated_params=['f', 'G'], max_iters=50, feastol=1e-9)
    with pytest.raises(AttributeError, match='not available'):
        es.solve(vals, updated_params=['f', 'G'], tol_feas=1e-9)
    es.close()


def test_generate_code_ecos_surface(sim_lib, tmp_path):
    e = ecos_from_conic(families.adp_norm())
    prob = LiteProblem.from_descriptor(e)
    cpg.generate_code(prob, code_dir=str(tmp_path / 'ecos_code'), solver='ECOS', wrapper=False)
    mod = cpg.load_generated(str(tmp_path / 'ecos_code'), prob)
    mod._SOLVER.lib_path = sim_lib
    val = prob.solve(method='CPG')
    assert prob.status.startswith('0 (for description visit https://github.com/embotech/ecos')
    assert np.isfinite(val) and prob.var_dict['u'].value.shape == (2, 3)
    assert prob.constraints[0].dual_value.shape == (2,) and prob.constraints[1].dual_value.shape == (3,)
    assert prob.solver_stats.solver_name == 'ECOS'


@pytest.mark.gpu
def test_ecos_solver_on_the_gpu(oracle_lib):
    from oracle import clarabel_numpy
    c = families.adp_norm()
    e = ecos_from_conic(c)
    B = 20000
    vals = _batch(B, seed=3)
    vals['f'][1], vals['G'][1] = vals['f'][0], vals['G'][0]
    es = EcosBatchSolver(e)
    r = es.solve(vals, updated_params=['f', 'G'])
    # ECOS_OPTIMAL; the optimum sits at the kink u = 0 of the norms, where a handful of 20 000 instances stall at
    # 1e-8 (the kernel's InsufficientProgress / NumericalError -> ECOS_NUMERICS, or ECOS_OPTIMAL + ECOS_INACC_OFFSET)
    assert np.isin(r.status, (0, 10, -2)).all() and (r.status == 0).mean() >= 0.999
    assert np.array_equal(r.sol_x[0], r.sol_x[1])                                # duplicates: identical bits
    stg = {knl: dflt for knl, dflt in ECOS_SETTINGS.values()}
    th = np.tile(c.theta0, (64, 1))
    for nm in ('f', 'G'):
        q = c.param(nm)
        th[:, q.col:q.col + q.size] = np.stack([c.flatten_param(nm, vals[nm][k]) for k in range(64)])
    o = clarabel_numpy.cpg_solve_batch(c, th[::4], **stg)
    assert r.iter[:64:4].tolist() == o['iter'].tolist()
    assert np.abs(r.sol_x[:64:4] - o['sol_x']).max() <= 1e-6 * max(1.0, np.abs(o['sol_x']).max())
    es.close()


@pytest.mark.gpu
def test_generate_code_ecos_builds_a_family_library_on_the_gpu(tmp_path):
    """generate_code(solver='ECOS') with its compile step: the library's generated executor belongs to the stacked conic
    form of THIS family (handle facts), and prob.solve(method='CPG') through it gives the generic library's result"""
    import ctypes as C
    e = ecos_from_conic(families.adp_norm())
    prob = LiteProblem.from_descriptor(e)
    mod = cpg.generate_code(prob, code_dir=str(tmp_path / 'ecos_lib'), solver='ECOS')
    val = prob.solve(method='CPG')
    bs = mod._SOLVER.batch_solver.conic
    assert 'ecos_lib' in bs.lib.path
    for fact in (b'generated_executor', b'specialised_kernel'):
        v = C.c_double(-1)
        bs.lib.check(bs.lib.L.cpg_hip_get_setting(bs.h, fact, C.byref(v)), 'get_setting')
        assert v.value == 1.0, fact
    es = EcosBatchSolver(e)                               # generic library, table-driven executor
    r = es.solve({}, updated_params=[], B=1)            # every parameter at its code-generation-time value, like `prob`
    assert abs(val - float(r.obj_val[0])) <= 1e-12 * max(1.0, abs(val))
    es.close()
