"""CPU tier for the kernel logic: the product's kernel sources compiled for the lock-step emulator
(tests/sim/build_sim.py) and driven through the real C-ABI + Python runtime, compared with the C
oracle.  Small batches only (64 host threads per emulated wavefront)."""
import os

import numpy as np
import pytest

from cvxpygen_amd import families
from cvxpygen_amd.runtime import BatchSolver


def _oracle_flat(oracle_lib, desc, theta, upd, **stg):
    o = oracle_lib.cpg_solve_batch(desc, theta, upd, **stg)
    prim = np.concatenate([o['sol_x'][:, v.indices] for v in desc.variables], axis=1)
    dual = np.concatenate([o['sol_y'][:, d.indices] for d in desc.duals], axis=1)
    return o, prim, dual


def _theta(desc, name, values):
    B = values.shape[0]
    th = np.tile(desc.theta0, (B, 1))
    p = desc.param(name)
    th[:, p.col:p.col + p.size] = values
    return th


def _assert_parity(r, o, prim, dual, tol=1e-9):
    assert r.iter.tolist() == o['iter'].tolist()
    assert r.status.tolist() == o['status'].tolist()
    ok = np.isin(o['status'], (1, 2, 7))
    if ok.any():
        assert np.abs(r.prim_flat[ok] - prim[ok]).max() <= tol * max(1.0, np.abs(prim[ok]).max())
        assert np.abs(r.dual_flat[ok] - dual[ok]).max() <= tol * max(1.0, np.abs(dual[ok]).max())
        assert np.abs(r.obj_val[ok] - o['obj_val'][ok]).max() <= tol * max(1.0, np.abs(o['obj_val'][ok]).max())
        assert np.allclose(r.pri_res[ok], o['pri_res'][ok], rtol=1e-6, atol=1e-9)
        assert np.allclose(r.dua_res[ok], o['dua_res'][ok], rtol=1e-6, atol=1e-9)
    if (~ok).any():
        assert np.isnan(r.prim_flat[~ok]).all()


@pytest.mark.parametrize('G', [1, 2])
def test_nonneg_ls_parity(sim_lib, oracle_lib, G):
    d = families.nonneg_ls()
    rng = np.random.default_rng(0)
    B = 5                                  # odd: exercises the padded lane group for G = 2
    bvals = rng.standard_normal((B, 3))
    bs = BatchSolver(d, lib_path=sim_lib)
    bs.set_launch(waves_per_block=2, inst_per_wave=G)
    for stg in ({}, dict(eps_abs=1e-7, eps_rel=1e-7), dict(max_iter=60), dict(check_termination=10)):
        r = bs.solve({'b': bvals}, updated_params=['b'], **stg)
        o, prim, dual = _oracle_flat(oracle_lib, d, _theta(d, 'b', bvals), ['b'], **stg)
        _assert_parity(r, o, prim, dual)
    assert r.prim['x'].shape == (B, 2) and r.dual['d0'].shape == (B, 2)
    bs.close()


def test_mpc_parity_and_shapes(sim_lib, oracle_lib):
    d = families.mpc(6, 3, 10)
    rng = np.random.default_rng(1)
    x0 = -2 + 4 * rng.random((2, 6))
    bs = BatchSolver(d, lib_path=sim_lib)
    bs.set_launch(waves_per_block=2, inst_per_wave=1)
    r = bs.solve({'x_init': x0}, updated_params=['x_init'])
    o, prim, dual = _oracle_flat(oracle_lib, d, _theta(d, 'x_init', x0), ['x_init'])
    _assert_parity(r, o, prim, dual)
    # user-facing arrays are F-order reshapes of the flat vectors (templates/cpg_solver.py.jinja2:78)
    assert r.prim['U'].shape == (2, 3, 10) and r.prim['X'].shape == (2, 6, 11)
    Xo = o['sol_x'][0, d.variables[1].indices].reshape(6, 11, order='F')
    assert np.allclose(r.prim['X'][0], Xo, atol=1e-9)
    assert np.allclose(r.prim['X'][:, :, 0], x0, atol=1e-3)       # X[:,0] == x_init up to ADMM accuracy
    bs.close()


def test_infeasible_instances_in_a_batch(sim_lib, oracle_lib):
    d = families.toy_box()
    B = 4
    th = np.tile(d.theta0, (B, 1))
    th[1, d.param('lb').col], th[1, d.param('ub').col] = 2.0, 1.0      # infeasible
    th[3, d.param('a').col] = 5.0                                      # active upper bound
    bs = BatchSolver(d, lib_path=sim_lib)
    bs.set_launch(waves_per_block=1, inst_per_wave=1)
    vals = {p.name: th[:, p.col:p.col + p.size] for p in d.params}
    r = bs.solve(vals)
    o, prim, dual = _oracle_flat(oracle_lib, d, th, None)
    _assert_parity(r, o, prim, dual)
    assert r.status[1] == 3 and r.obj_val[1] == np.inf
    assert abs(r.prim['x'][3, 0] - 1.0) < 1e-2
    d2 = families.toy_lp()
    th2 = np.tile(d2.theta0, (2, 1)); th2[1, 0] = -1.0
    bs2 = BatchSolver(d2, lib_path=sim_lib)
    bs2.set_launch(waves_per_block=1, inst_per_wave=1)
    r2 = bs2.solve({'c': th2[:, :1]})
    o2, prim2, dual2 = _oracle_flat(oracle_lib, d2, th2, None)
    _assert_parity(r2, o2, prim2, dual2)
    assert r2.status[1] == 5 and r2.obj_val[1] == -np.inf
    bs.close(); bs2.close()


def test_settings_and_errors(sim_lib):
    d = families.nonneg_ls()
    bs = BatchSolver(d, lib_path=sim_lib)
    with pytest.raises(AttributeError, match='not available'):
        bs.solve({'b': np.zeros((1, 3))}, updated_params=['b'], polish=True)       # disabled setting
    with pytest.raises(AttributeError, match='is not a parameter'):
        bs.solve({'b': np.zeros((1, 3))}, updated_params=['nope'])
    bs.solve({'b': np.ones((1, 3))}, updated_params=['b'], warm_start=False)     # cvxpy alias accepted
    bs.close()


@pytest.mark.parametrize('fam,G,gen', [('nnls', 1, {}), ('nnls', 2, {}), ('mpc6', 1, {}),
                                       ('mpc6', 1, dict(cross=0, pad_offsets=False, early=0)),
                                       ('nnls', 2, dict(cross=9, depth=3, group_offsets=2, early=9)), ('mpc6', 1, dict(group_offsets=1, batch=2))])
def test_generated_executor_parity(oracle_lib, tmp_path, fam, G, gen):
    """cvxpygen_amd.codegen: the family-specialised straight-line executor (emulator build of the
    generated source) gives the oracle's results; a library generated for one family refuses another."""
    from cvxpygen_amd.runtime import build_family_plan
    from sim import build_sim
    if fam == 'nnls':
        d, name, vals = families.nonneg_ls(), 'b', np.random.default_rng(0).standard_normal((5, 3))
    else:
        d, name, vals = families.mpc(6, 3, 10), 'x_init', -2 + 4 * np.random.default_rng(1).random((3, 6))
    plan = build_family_plan(d)
    lib = build_sim.build_family(plan, str(tmp_path), fam, **gen)   # gen: pipeline options of the
    bs = BatchSolver(d, lib_path=lib, plan=plan)                    # generated executor
    bs.set_launch(waves_per_block=2, inst_per_wave=G)
    r = bs.solve({name: vals}, updated_params=[name])
    o, prim, dual = _oracle_flat(oracle_lib, d, _theta(d, name, vals), [name])
    _assert_parity(r, o, prim, dual)
    bs.close()
    other = families.toy_box()
    with pytest.raises(RuntimeError, match='different problem family'):
        BatchSolver(other, lib_path=lib)


def test_program_placements_parity(sim_lib, oracle_lib):
    """table-driven kernels with the solve program streamed (run_program_stream: what large families
    and the generic library's automatic placement use) and LDS resident: both give the oracle's
    results (the streamed family library itself, codegen.build_streamed_family_library, is compiled
    by __graft_entry__.build and exercised on the GPU by test_large_family_beyond_the_generic_slot_classes)"""
    d = families.mpc(6, 3, 10)
    vals = -2 + 4 * np.random.default_rng(4).random((2, 6))
    bs = BatchSolver(d, lib_path=sim_lib)
    o, prim, dual = _oracle_flat(oracle_lib, d, _theta(d, 'x_init', vals), ['x_init'], max_iter=50)
    for placement in (0, 1):
        bs.set_program_placement(placement)
        bs.set_launch(waves_per_block=2, inst_per_wave=1)
        r = bs.solve({'x_init': vals}, updated_params=['x_init'], max_iter=50)
        _assert_parity(r, o, prim, dual)
    bs.close()


def test_refactor_path_all_parameters(sim_lib, oracle_lib):
    """parameters entering A: per-instance canonicalisation, Ruiz equilibration from scratch, numeric
    LDL' and ADMM with the instance's own factor (reference: osqp_update_data_mat)"""
    d = families.nonneg_ls()
    rng = np.random.default_rng(0)
    B = 3
    th = np.tile(d.theta0, (B, 1))
    th[:, :d.NP] *= 1 + 0.1 * rng.standard_normal((B, d.NP))
    bs = BatchSolver(d, lib_path=sim_lib)
    vals = {p.name: th[:, p.col:p.col + p.size] for p in d.params}
    r = bs.solve(vals)                                  # updated_params=None: every parameter
    o, prim, dual = _oracle_flat(oracle_lib, d, th, None)
    _assert_parity(r, o, prim, dual)
    # switching back to a vector-only update uses the shared factor again
    r2 = bs.solve({'b': th[:, 3:6]}, updated_params=['b'])
    o2, prim2, dual2 = _oracle_flat(oracle_lib, d, _theta(d, 'b', th[:, 3:6]), ['b'])
    _assert_parity(r2, o2, prim2, dual2)
    bs.close()


def test_refactor_path_matrix_and_vector_update_order(sim_lib, oracle_lib):
    """q and A both outdated: the reference calls osqp_update_data_mat first (re-equilibration, cost
    scaling sees the OLD q) and osqp_update_data_vec second (cvxpygen/solvers/osqp.py:20-59)"""
    d = families.toy_qa()
    rng = np.random.default_rng(3)
    B = 3
    th = np.tile(d.theta0, (B, 1))
    th[:, :d.NP] += 0.5 * rng.standard_normal((B, d.NP))
    th[:, d.param('c').col:d.param('c').col + 3] *= 40.0      # make ||q|| decide the cost scaling
    bs = BatchSolver(d, lib_path=sim_lib)
    r = bs.solve({p.name: th[:, p.col:p.col + p.size] for p in d.params}, max_iter=100)
    o, prim, dual = _oracle_flat(oracle_lib, d, th, None, max_iter=100)
    _assert_parity(r, o, prim, dual)
    bs.close()


def test_refactor_path_parameter_in_P(sim_lib, oracle_lib):
    """tests/test_E2E_QP.py 'actuator': lamb_sm enters P -> osqp_update_data_mat with new P values"""
    d = families.actuator()
    rng = np.random.default_rng(1)
    B = 3
    th = np.tile(d.theta0, (B, 1))
    th[:, d.param('lamb_sm').col] = rng.random(B)                     # np.random.rand() per seed in the reference test
    th[:, d.param('w').col:d.param('w').col + 3] += rng.standard_normal((B, 3))
    bs = BatchSolver(d, lib_path=sim_lib)
    # (three checks and one rho adaptation are enough here; the cut-off exercises the approximate second test too)
    r = bs.solve({p.name: th[:, p.col:p.col + p.size] for p in d.params}, max_iter=75)
    o, prim, dual = _oracle_flat(oracle_lib, d, th, None, max_iter=75)
    assert np.isin(o['status'], (2, 7)).all() and (o['status'] == 7).any()
    # (slow family, cut off far from its solution right after a rho change: rounding differences of the two
    # factorisations show at 1e-9; the contract is 1e-6)
    _assert_parity(r, o, prim, dual, tol=1e-8)
    assert r.prim['delta_u'].shape == (B, 1, 1) and r.dual['d2'].shape == (B, 1, 1)
    bs.close()



def test_generated_instance_executor_and_both_ldl_forms(oracle_lib, tmp_path):
    """shared-matrix mode of the per-instance factor kernel (rho adaptation hand-over) in a family library: the
    generated instance executor (register-resident coefficients, LDL' in LDS) and the streaming executor (LDL' in
    the wavefront's global buffer) both use the one-step-per-level factorisation and must give the oracle's
    results -- through several rho adaptations (tight tolerances) and a max_iter cut-off."""
    import ctypes as C
    from cvxpygen_amd.runtime import build_family_plan
    from sim import build_sim
    d = families.mpc(6, 3, 10)
    plan = build_family_plan(d)
    lib = build_sim.build_family(plan, str(tmp_path), 'mpc6')
    vals = -2 + 4 * np.random.default_rng(4).random((4, 6))
    for executor in ('generated', 'stream'):
        bs = BatchSolver(d, lib_path=lib, plan=plan)
        bs.set_launch(waves_per_block=2)
        bs.set_updated(['x_init'])
        assert bs._hybrid
        if executor == 'stream':
            bs.lib.check(bs.lib.L.cpg_hip_set_program_placement(bs.h_rs, 0), 'placement')
        v = C.c_double(-1)
        bs.lib.check(bs.lib.L.cpg_hip_get_setting(bs.h_rs, b'generated_instance_executor', C.byref(v)), 'get')
        assert v.value == (1.0 if executor == 'generated' else 0.0)
        for stg in ({}, dict(eps_abs=1e-7, eps_rel=1e-7), dict(max_iter=60)):
            r = bs.solve({'x_init': vals}, updated_params=['x_init'], **stg)
            o, prim, dual = _oracle_flat(oracle_lib, d, _theta(d, 'x_init', vals), ['x_init'], **stg)
            _assert_parity(r, o, prim, dual)
            if 'eps_abs' in stg:
                assert o['iter'].max() > 100          # more than one adaptation point was passed
        bs.close()


def test_entry_words_in_lds_for_the_portfolio_family(oracle_lib, tmp_path):
    """family library of BASELINE config 3's family (portfolio n=100 m=10: 10 + 13 slots): its per-instance-matrix kernel
    keeps the streaming executor's entry words in a block-shared LDS copy (CPG_REFACTOR_CR_LDS, eight wavefronts per
    workgroup); the same solve with the entry words read through L2 (placement 0) must give identical bits, and both
    the oracle's iterates after 30 iterations"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sim import build_sim
    from cvxpygen_amd import codegen
    from cvxpygen_amd.runtime import build_family_plan
    d = families.portfolio(100, 10)
    plan = build_family_plan(d)
    _, defs = codegen.family_library_defs(plan, str(tmp_path), 'portfolio')
    assert '-DCPG_REFACTOR_CR_LDS=1' in defs
    lib = build_sim.build_family(plan, str(tmp_path), 'portfolio')
    rng = np.random.default_rng(5)
    B, n, m = 2, 100, 10
    sig = np.zeros((B, m, m)); sig[:, np.arange(m), np.arange(m)] = rng.random((B, m))
    vals = {'a': rng.standard_normal((B, n)), 'F': np.round(rng.standard_normal((B, n, m))), 'Sig_f_sqrt': sig,
            'd_sqrt': rng.random((B, n)), 'w_prev': np.zeros((B, n))}
    th = np.stack([d.theta_from_values({k: v[i] for k, v in vals.items()}) for i in range(B)])
    upd = ['a', 'F', 'Sig_f_sqrt', 'd_sqrt', 'w_prev']
    stg = dict(max_iter=30)
    out = []
    for placement in (2, 0):                  # (-1 would select the resident kernel of this library: tests/test_resident.py)
        bs = BatchSolver(d, lib_path=lib, plan=plan)
        bs.set_program_placement(placement)
        out.append(bs.solve(vals, updated_params=upd, **stg))
        bs.close()
    o, prim, dual = _oracle_flat(oracle_lib, d, th, upd, **stg)
    _assert_parity(out[0], o, prim, dual, tol=1e-8)
    assert np.array_equal(out[0].prim_flat, out[1].prim_flat) and np.array_equal(out[0].dual_flat, out[1].dual_flat)


def test_shared_mode_plan_follows_the_library(sim_lib, tmp_path):
    """the refactorisation plan of shared-matrix mode is the one planned for the generated instance executor only when the
    library carries that executor for this family (its cpg_instance_<name>.h next to it); the generic library streams
    the program and must get the streaming plan (fewer, wider steps)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sim import build_sim
    from cvxpygen_amd import refactor_plan as _rp
    from cvxpygen_amd.runtime import build_family_plan
    d = families.mpc(6, 3, 10)
    plan = build_family_plan(d)
    o = plan.osqp_shared or plan.osqp
    Ps, As = o.pruned(d.P, d.A)
    inst, stream = _rp.shared_mode_plan(Ps, As, o).sol.fingerprint(), _rp.build_refactor_plan(Ps, As, o).sol.fingerprint()
    assert inst != stream
    x0 = -2 + 4 * np.random.default_rng(2).random((2, 6))
    for lib, want in ((sim_lib, stream), (build_sim.build_family(plan, str(tmp_path), 'mpc6'), inst)):
        bs = BatchSolver(d, lib_path=lib, plan=plan)
        bs.solve({'x_init': x0}, updated_params=['x_init'])
        assert bs._hybrid and bs._rplan_s.sol.fingerprint() == want
        bs.close()


def test_raw_entry_points_refuse_rho_adaptation_on_a_shared_factor_handle_without_a_linked_one(sim_lib, oracle_lib):
    """the C entry points driven WITHOUT the host layer's re-solve (device and pipelined paths): a shared-factor handle
    with rho adaptation on and no per-instance factor handle linked is refused (CPG_E_UNSUPPORTED, loud message) instead
    of returning CPG_OK with unsolved status -2 rows; opting in (build option flag_rho_changes) gives exactly the rows
    whose rho changed as -2; with adaptation off every row is solved (include/cpg_hip.h, cpg_hip_set_handover)"""
    import ctypes as C
    from cvxpygen_amd.runtime import DeviceBatch, PinnedStream
    d = families.mpc(6, 3, 10)
    B = 6
    x0 = -2 + 4 * np.random.default_rng(3).random((B, 6))
    bs = BatchSolver(d, lib_path=sim_lib)
    bs.set_updated(['x_init'])
    L = bs.lib.L
    dev = DeviceBatch(bs, B)
    dev.upload(x0)

    def raw_status():
        bs.solve_device(dev); bs.synchronize()
        st = np.empty(B, dtype=np.int32)
        bs.lib.check(L.cpg_hip_memcpy_d2h(bs.h, st.ctypes.data_as(C.c_void_p), dev._ptrs['status'], st.nbytes), 'd2h')
        it = np.empty(B, dtype=np.int32)
        bs.lib.check(L.cpg_hip_memcpy_d2h(bs.h, it.ctypes.data_as(C.c_void_p), dev._ptrs['iter'], it.nbytes), 'd2h')
        return st, it
    o, _, _ = _oracle_flat(oracle_lib, d, _theta(d, 'x_init', x0), ['x_init'])
    st, it = raw_status()                                   # linked (what the host layer sets up): two kernels, nothing flagged
    assert st.tolist() == o['status'].tolist() and it.tolist() == o['iter'].tolist() and (o['iter'] > 50).any()
    bs.lib.check(L.cpg_hip_set_handover(bs.h_shared, None), 'cpg_hip_set_handover')
    with pytest.raises(RuntimeError, match=r'\(-4\).*cpg_hip_set_handover'):
        bs.solve_device(dev)
    ps = PinnedStream(bs, B, 2)
    ps.theta[:] = np.tile(x0, (2, 1))
    with pytest.raises(RuntimeError, match='cpg_hip_set_handover'):
        ps.run()
    bs.lib.check(L.cpg_hip_set_build_option(bs.h_shared, b'flag_rho_changes', 1.0), 'cpg_hip_set_build_option')
    st2, _ = raw_status()
    assert (st2 == -2).any() and ((st2 == -2) <= (o['iter'] > 50)).all()      # only instances that reached the first adaptation
    keep = st2 != -2
    assert st2[keep].tolist() == o['status'][keep].tolist()
    bs.lib.check(L.cpg_hip_set_build_option(bs.h_shared, b'flag_rho_changes', 0.0), 'cpg_hip_set_build_option')
    bs.lib.check(L.cpg_hip_set_build_option(bs.h_shared, b'adaptive_rho', 0.0), 'cpg_hip_set_build_option')
    st3, _ = raw_status()
    assert (st3 == 1).all()
    ps.run()
    assert (np.array(ps.status) == 1).all()
    ps.free(); dev.free(); bs.close()


def test_generated_streaming_executor_for_per_instance_matrices(sim_lib, oracle_lib, tmp_path, monkeypatch):
    """a family whose parameters enter P / A and whose merged program does not fit the resident kernel (an MPC chain): with
    CPG_GENS=1 its family library runs the per-instance substitution program as generated straight-line code with
    coefficients and tables from global memory (cpg_stream_<name>.h, CPG_GENS_HEADER) -- every parameter varying, against
    the oracle, against the table-driven streaming executor of the generic library, and through a rho adaptation.  (Off by
    default: slower than the table-driven executor on the GPU, codegen.stream_header says why; the path stays correct.)"""
    monkeypatch.setenv('CPG_GENS', '1')
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sim import build_sim
    from cvxpygen_amd import codegen
    from cvxpygen_amd.runtime import build_family_plan
    d = families.mpc(8, 3, 7)
    plan = build_family_plan(d)
    _, defs = codegen.family_library_defs(plan, str(tmp_path), 'mpc8')
    if not any('CPG_GENS_HEADER' in x for x in defs):
        pytest.skip('this family fits the resident kernel')
    lib = build_sim.build_family(plan, str(tmp_path), 'mpc8')
    rng = np.random.default_rng(2)
    B = 3
    th = np.tile(d.theta0, (B, 1))
    th[:, :d.NP] *= 1 + 0.05 * rng.standard_normal((B, d.NP))
    p = d.param('x_init')
    th[:, p.col:p.col + p.size] = -2 + 4 * rng.random((B, p.size))
    B = 2; th = th[:2]
    vals = {q.name: th[:, q.col:q.col + q.size] for q in d.params}
    out = {}
    for tag, path in (('family', lib), ('generic', sim_lib)):
        bs = BatchSolver(d, lib_path=path, plan=plan if tag == 'family' else None)
        for stg in ({}, dict(eps_abs=1e-7, eps_rel=1e-7)):
            r = bs.solve(vals, **stg)
            o, prim, dual = _oracle_flat(oracle_lib, d, th, None, **stg)
            _assert_parity(r, o, prim, dual, tol=1e-8)
        out[tag] = r
        bs.close()
    assert out['family'].iter.tolist() == out['generic'].iter.tolist()
    assert np.abs(out['family'].prim_flat - out['generic'].prim_flat).max() < 1e-9


@pytest.mark.parametrize('fam', ['nnls', 'mpc6', 'toy_box'])
def test_squad_executor_parity(oracle_lib, tmp_path, fam):
    """csrc/cpg_osqp_squad.h + codegen.emit_squad_program (round 6): the family's solve program in the registers of a
    squad of four wavefronts that solves four instances at a time -- placement 3 of a family library (an experiment: on MI355X
    it lost against the LDS-resident program, HISTORY.md round 6; kept selectable and tested).
    Against the oracle AND against the LDS-program executor of the same library (placement 1): batch sizes that do not fill
    the last squad, one instance, events that are not aligned across a squad (check_termination 7 next to the adaptation
    interval 50), the fixed-rho fork, max_iter reached, an infeasible instance among feasible ones."""
    from cvxpygen_amd.runtime import build_family_plan, BUILD_OPTIONS_FIXED_RHO
    from sim import build_sim
    import ctypes as C
    rng = np.random.default_rng(3)
    if fam == 'nnls':
        d, name = families.nonneg_ls(), 'b'
        draw = lambda B: rng.standard_normal((B, 3))
    elif fam == 'mpc6':
        d, name = families.mpc(6, 3, 10), 'x_init'
        draw = lambda B: -2 + 4 * rng.random((B, 6))
    else:
        d, name = families.toy_box(), None
    plan = build_family_plan(d)
    assert plan.kkt_squad is not None and 'squad_conflict_cycles' in plan.stats
    lib = build_sim.build_family(plan, str(tmp_path), fam)
    assert os.path.exists(os.path.join(str(tmp_path), f'cpg_squad_{fam}.h'))

    def run(vals, placement, build_options=None, **stg):
        bs = BatchSolver(d, lib_path=lib, plan=plan, build_options=build_options or {})
        bs.set_program_placement(placement)
        r = bs.solve(vals, updated_params=list(vals.keys()), **stg)
        v = C.c_double(0)
        bs.lib.L.cpg_hip_get_setting(bs.h_shared, b'squad_executor', C.byref(v))
        assert v.value == (1.0 if placement == 3 else 0.0)
        bs.close()
        return r
    if fam == 'toy_box':
        B = 6
        th = np.tile(d.theta0, (B, 1))
        th[1, d.param('lb').col], th[1, d.param('ub').col] = 2.0, 1.0      # infeasible
        th[4, d.param('a').col] = 5.0                                      # active upper bound
        vals = {p.name: th[:, p.col:p.col + p.size] for p in d.params}
        r = run(vals, 3)
        o, prim, dual = _oracle_flat(oracle_lib, d, th, None)
        _assert_parity(r, o, prim, dual)
        assert r.status[1] == 3 and r.obj_val[1] == np.inf
        return
    cases = [(5, {}, None), (1, {}, None), (9, dict(check_termination=7), None), (4, {}, dict(BUILD_OPTIONS_FIXED_RHO)),
             (3, dict(max_iter=30), None), (2, dict(max_iter=0), None)]
    if fam == 'mpc6':
        cases = cases[:3]                   # (64 host fibers per emulated wavefront: keep the larger family short)
    for B, stg, bo in cases:
        v = draw(B)
        mode = dict(adaptive_rho=0, check_dualgap=0) if bo else {}
        o, prim, dual = _oracle_flat(oracle_lib, d, _theta(d, name, v), [name], **mode, **stg)
        r3 = run({name: v}, 3, bo, **stg)
        _assert_parity(r3, o, prim, dual)
        r1 = run({name: v}, 1, bo, **stg)
        assert r1.iter.tolist() == r3.iter.tolist() and r1.status.tolist() == r3.status.tolist()
        ok = np.isin(r3.status, (1, 2, 7))
        assert np.allclose(r1.prim_flat[ok], r3.prim_flat[ok], rtol=1e-9, atol=1e-11)
