"""Row C1: the conic interior-point path (reference: Clarabel,
cvxpygen/solvers/clarabel.py) -- oracle against independent known answers, the kernel logic in the
lock-step emulator against the oracle (CPU tier), and the HIP kernel on the GPU against the oracle
and through size-independent properties (-m gpu)."""
import json
import os

import numpy as np
import pytest

from cvxpygen_amd import families
from cvxpygen_amd.conic_plan import build_conic_plan
from cvxpygen_amd.conic_runtime import ConicBatchSolver
from oracle import clarabel_numpy as cl

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'known_answers.json')))
REL_TOL = 1e-6


def _adp_batch(B, seed):
    rs = np.random.RandomState(seed)
    vals = [families.adp_values(-2 + 4 * rs.rand(6)) for _ in range(B)]     # tests/test_E2E_SOCP.py:57
    return {k: np.stack([v[k] for v in vals]) for k in vals[0]}


def _theta(desc, pv):
    B = next(iter(pv.values())).shape[0]
    return np.stack([desc.theta_from_values({k: v[i] for k, v in pv.items()}) for i in range(B)])


def _assert_parity(r, o, tol=REL_TOL):
    assert r.iter.tolist() == o['iter'].tolist()
    assert r.status.tolist() == o['status'].tolist()
    ok = o['status'] == 1
    assert np.abs(r.sol_x[ok] - o['sol_x'][ok]).max() <= tol * max(1.0, np.abs(o['sol_x'][ok]).max())
    assert np.abs(r.sol_y[ok] - o['sol_z'][ok]).max() <= tol * max(1.0, np.abs(o['sol_z'][ok]).max())
    assert np.abs(r.obj_val[ok] - o['obj_val'][ok]).max() <= tol * max(1.0, np.abs(o['obj_val'][ok]).max())
    assert np.isnan(r.obj_val[~ok]).all()


# ------------------------------------------------------------------------------------ oracle / host
def test_oracle_matches_exact_adp_solutions():
    """the restatement against the exact trust-region solution of the reference's ADP inputs
    (tests/golden/make_golden.py); the reference's own bar is 10 % (tests/test_E2E_SOCP.py:96-109)"""
    d = families.adp()
    for seed, g in GOLD['ADP'].items():
        th = d.theta_from_values({'Rsqrt': np.diag(g['Rsqrt_diag']), 'f': np.array(g['f']), 'G': np.array(g['G'])})
        o = cl.cpg_solve_batch(d, th[None])
        assert o['status'][0] == cl.SOLVED and o['iter'][0] <= 12
        u = o['prim']['u'][0].reshape((2, 3), order='F')
        assert abs(o['obj_val'][0] - g['obj']) <= 1e-7 * g['obj']
        assert np.abs(u[0] - g['u0']).max() <= 1e-5
        assert np.abs(u[1]).max() <= 1e-6                      # only has to stay in its ball: centre
        assert abs(o['dual']['d0'][0][0] - g['dual_norm_u0']) <= 1e-6
        assert o['pri_res'][0] < 1e-8 and o['dua_res'][0] < 1e-8


def test_oracle_on_the_nonneg_ls_example_in_conic_form():
    """`examples/main.py` family handed to the conic path (zero + nonnegative cones, P = 2I on t): the
    interior-point oracle against the exact NNLS answer of tests/golden, and against the OSQP
    oracle's solution of the same instance -- the two restatements share no code"""
    g = GOLD['nonneg_LS']
    d = families.nonneg_ls(solver='CLARABEL')
    th = d.theta_from_values({'A': np.array(g['A_data']), 'b': np.array(g['b'])})
    o = cl.cpg_solve_batch(d, th[None])
    assert o['status'][0] == cl.SOLVED
    assert np.abs(o['prim']['x'][0] - g['x']).max() < 1e-7 and abs(o['obj_val'][0] - g['obj']) < 1e-7
    assert np.abs(o['dual']['d0'][0] - g['dual_x_ge_0']).max() < 1e-6
    from oracle import binding
    dq = families.nonneg_ls()
    oq = binding.cpg_solve_batch(dq, dq.theta_from_values({'A': np.array(g['A_data']), 'b': np.array(g['b'])})[None],
                                 None, eps_abs=1e-9, eps_rel=1e-9)
    assert np.abs(o['prim']['x'][0] - oq['prim']['x'][0]).max() < 1e-6


def test_adp_descriptor_matches_survey_dimensions():
    d = families.adp()
    assert (d.n_var, d.n_eq, d.n_ineq, d.NP) == (17, 9, 10, 27)
    assert d.cones == {'zero': 9, 'nonneg': 2, 'soc': [4, 4]}
    assert d.user_p_name_to_canon_outdated() == {'Rsqrt': ['A'], 'f': ['b'], 'G': ['A']}
    # theta0 = the values of np.random.seed(0) in tests/test_E2E_SOCP.py:38-62
    g = GOLD['ADP']['0']
    assert np.allclose(d.theta0[3:9], g['f']) and np.allclose(d.theta0[9:27], np.array(g['G']).flatten(order='F'))


def test_conic_plan_factor_replay():
    """the LDL' schedule + substitution program of the plan, replayed in numpy with random
    quasi-definite values, solve K x = b"""
    from cvxpygen_amd import conic_plan as cp_
    d = families.adp()
    cp = build_conic_plan(d)
    rng = np.random.default_rng(0)
    n, m, N = d.n_var, d.m, d.n_var + d.m
    Pv, Av = np.abs(d.P.data) + 0.1, rng.standard_normal(d.A.nnz)
    hd = rng.random(m) + 0.5
    hd[:cp.n_zero] = 0.0
    wv = rng.random(m)
    o = cp.n_zero + cp.n_nonneg
    for dm in cp.soc_dims:                              # a valid NT scaling: w0^2 - |w1|^2 = 1
        wv[o] = np.sqrt(1.0 + wv[o + 1:o + dm] @ wv[o + 1:o + dm])
        hd[o:o + dm] = 2.0 * wv[o:o + dm] ** 2 + 1.0
        hd[o] -= 2.0
        o += dm
    eps = 1e-2                                          # schedule test, not a conditioning test
    # dense K from the sources
    K = np.zeros((N, N))
    pc = np.repeat(np.arange(n), np.diff(d.P.indptr))
    K[d.P.indices, pc] = Pv
    K[np.arange(n), np.arange(n)] += eps
    ac = np.repeat(np.arange(n), np.diff(d.A.indptr))
    K[ac, n + d.A.indices] = Av
    K[n + np.arange(m), n + np.arange(m)] = -hd - eps
    o = cp.n_zero + cp.n_nonneg
    for dm in cp.soc_dims:
        for a in range(dm):
            for b in range(a + 1, dm):
                K[n + o + a, n + o + b] = -2.0 * wv[o + a] * wv[o + b]
        o += dm
    K = np.triu(K) + np.triu(K, 1).T
    # numeric factor through the schedule
    nnzL = cp.nnzL
    Lx, Dg = np.zeros(nnzL), np.zeros(N)

    def kval(t):
        kind, idx = cp.ksrc_kind[t], cp.ksrc_idx[t]
        piv = t >= nnzL
        if kind == cp_.K_P: return Pv[idx] + (eps if piv else 0.0)
        if kind == cp_.K_A: return Av[idx]
        if kind == cp_.K_DIAGX: return eps
        if kind == cp_.K_HDIAG: return -hd[idx] - eps
        if kind == cp_.K_HSOC: return -2.0 * wv[idx & 0xFFFF] * wv[idx >> 16]
        return 0.0
    level_start = 0
    for c in range(cp.fac.n_chunks):
        L, last, base, lg = cp.fac.ctab[c]
        acc = np.zeros(64)
        alen, rlen = cp.fac.tlen[c] & 0xFFFF, cp.fac.tlen[c] >> 16      # addressing length | real terms per lane
        for s in range(L):
            act = np.nonzero(alen > s)[0]
            e = base + np.arange(len(act))
            acc[act] += np.where(rlen[act] > s, Lx[cp.fac_a[e]] * Dg[cp.fac_k[e]] * Lx[cp.fac_b[e]], 0.0)
            base += len(act)
        acc = np.repeat(acc.reshape(-1, 1 << lg).sum(axis=1), 1 << lg)   # group sums (first lane owns the task)
        for ln in range(64):
            t = int(cp.fac.task[c, ln])
            if t == 0xFFFFFFFF:
                continue
            v = kval(t) - acc[ln]
            if t >= nnzL: Dg[t - nnzL] = v
            else: Lx[t] = v
        if last:
            for c2 in range(level_start, c + 1):
                for t in cp.fac.task[c2]:
                    if t < nnzL:
                        Lx[t] /= Dg[cp.Lcol[t]]
            level_start = c + 1
    Lm = np.eye(N)
    Lm[cp.Li, cp.Lcol] = Lx
    Kp = K[np.ix_(cp.perm, cp.perm)]
    assert np.abs(Lm @ np.diag(Dg) @ Lm.T - Kp).max() < 1e-9 * np.abs(Kp).max()
    assert (np.sign(Dg) == np.where(cp.perm < n, 1, -1)).all()          # quasi-definite pivots


# ------------------------------------------------------------------------------------ emulator tier
def test_adp_in_emulator_vs_oracle(sim_lib):
    d = families.adp()
    pv = _adp_batch(3, 1)
    bs = ConicBatchSolver(d, lib_path=sim_lib, full_output=True)
    r = bs.solve(pv)
    _assert_parity(r, cl.cpg_solve_batch(d, _theta(d, pv)), tol=1e-9)
    assert r.prim['u'].shape == (3, 2, 3) and r.dual['d0'].shape == (3, 2)
    # tighter / looser settings change the iteration count the same way on both sides
    r2 = bs.solve(pv, tol_gap_abs=1e-4, tol_gap_rel=1e-4, tol_feas=1e-4)
    o2 = cl.cpg_solve_batch(d, _theta(d, pv), tol_gap_abs=1e-4, tol_gap_rel=1e-4, tol_feas=1e-4)
    _assert_parity(r2, o2, tol=1e-9)
    assert (r2.iter < r.iter).any()
    r3 = bs.solve(pv, max_iters=2)                      # cvxpy name of max_iter (clarabel.py:65)
    assert (r3.status == 7).all() and (r3.iter == 2).all()
    with pytest.raises(AttributeError, match='not available'):
        bs.solve(pv, eps_abs=1e-3)                      # an OSQP setting
    bs.close()


def _fact(bs, name):
    import ctypes as C
    v = C.c_double(-1)
    bs.lib.check(bs.lib.L.cpg_hip_get_setting(bs.h, name.encode(), C.byref(v)), 'get_setting')
    return v.value


def test_generated_conic_executor_in_emulator(tmp_path):
    """a conic family library (codegen.conic_header: straight-line executor of the substitution program, entries read
    from the wave's LDS array) gives the table-driven executor's results bit for bit -- same products, same
    accumulation order --, and hands any OTHER family to the table-driven executor it also carries"""
    from tests.sim import build_sim
    d = families.adp()
    cp = build_conic_plan(d)
    lib = build_sim.build_conic_family(cp, str(tmp_path), 'adp')
    hdr = open(os.path.join(str(tmp_path), 'cpg_conic_adp.h')).read()
    assert 'run_program_conic' in hdr and f'#define CPG_GENC_FINGERPRINT {cp.sol.fingerprint()}u' in hdr
    pv = _adp_batch(4, 5)
    bs = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
    r = bs.solve(pv)
    assert _fact(bs, 'generated_executor') == 1.0 and _fact(bs, 'specialised_kernel') == 1.0
    o = cl.cpg_solve_batch(d, _theta(d, pv))
    _assert_parity(r, o, tol=1e-9)
    os.environ['CPG_CONIC_GENERATED'] = '0'              # same library, table-driven executor
    try:
        bt = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
        rt = bt.solve(pv)
    finally:
        del os.environ['CPG_CONIC_GENERATED']
    assert _fact(bt, 'generated_executor') == 0.0 and _fact(bt, 'specialised_kernel') == 0.0
    assert np.array_equal(r.sol_x, rt.sol_x) and np.array_equal(r.sol_y, rt.sol_y) and r.iter.tolist() == rt.iter.tolist()
    # the factorisation schedule as straight-line code (codegen.emit_conic_factor, round 6) against the table walk of the same
    # library: same arithmetic in the same order -- identical bits
    assert _fact(bs, 'generated_factorisation') == 1.0 and os.path.exists(os.path.join(str(tmp_path), 'cpg_conic_adp_factor.h'))
    os.environ['CPG_CONIC_FACTOR'] = '0'
    try:
        bf = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
        rf = bf.solve(pv)
    finally:
        del os.environ['CPG_CONIC_FACTOR']
    assert _fact(bf, 'generated_factorisation') == 0.0 and _fact(bf, 'generated_executor') == 1.0
    assert np.array_equal(r.sol_x, rf.sol_x) and np.array_equal(r.sol_y, rf.sol_y) and r.iter.tolist() == rf.iter.tolist()
    bs.close(); bt.close(); bf.close()
    d2 = families.toy_box(solver='CLARABEL')             # another family through the ADP library
    th = np.tile(d2.theta0, (2, 1)); th[1, :3] = [0.3, 1.0, -1.0]
    b2 = ConicBatchSolver(d2, lib_path=lib, full_output=True)
    _assert_parity(b2.solve({'a': th[:, 0], 'lb': th[:, 1], 'ub': th[:, 2]}), cl.cpg_solve_batch(d2, th), tol=1e-9)
    assert _fact(b2, 'generated_executor') == 0.0
    b2.close()


def test_row_words_decode_to_the_patterns():
    """codegen.conic_row_tables: the words of every lane, decoded, are that row of P (full symmetric view), that column of A,
    that row of A -- entries in the order of the kernel's table-driven loops -- for a family with 16-bit words; an operand or
    entry count beyond one wavefront pass yields no tables"""
    import re
    import scipy.sparse as sp
    from cvxpygen_amd import codegen
    for d in (families.adp(), families.toy_box(solver='CLARABEL')):
        cp = build_conic_plan(d)
        t = codegen.conic_row_tables(cp)
        S = [int(re.search(rf'#define CPG_GENC_ROWS_{k} (\d+)', t).group(1)) for k in ('SP', 'SAT', 'SA')]
        words = [int(x, 16) for x in re.findall(r'0x[0-9a-f]{4}\b', t.split('genc_row_words[')[1])]
        assert len(words) == 64 * sum(S) == int(re.search(r'#define CPG_GENC_ROWS_WORDS (\d+)', t).group(1))
        lists = {'P': (0, S[0], [list(zip(cp.Pent[cp.Prp[j]:cp.Prp[j + 1]], cp.Pcol[cp.Prp[j]:cp.Prp[j + 1]])) for j in range(cp.n)]),
                 'At': (S[0], S[1], [[(cp.nnzP + k, cp.Ai[k]) for k in range(cp.Ap[j], cp.Ap[j + 1])] for j in range(cp.n)]),
                 'A': (S[0] + S[1], S[2], [list(zip(cp.nnzP + cp.Aent[cp.Arp[i]:cp.Arp[i + 1]], cp.Acol[cp.Arp[i]:cp.Arp[i + 1]])) for i in range(cp.m)])}
        for first, steps, rows in lists.values():
            assert steps == max(len(r) for r in rows)
            for lane in range(64):
                got = [(w & 0xFF, (w >> 8) & 0x7F) for w in (words[(first + s_) * 64 + lane] for s_ in range(steps)) if w >> 15]
                want = [(int(a), int(b)) for a, b in rows[lane]] if lane < len(rows) else []
                assert got == want
    import types
    big = types.SimpleNamespace(n=65, m=3, nnzP=0, nnzA=0)
    assert codegen.conic_row_tables(big) == ''


def _dense_lp():
    """minimise c'x  s.t.  G x <= h (24 dense rows), -1 <= x <= 1, x in R^20: 520 entries of A -> 32-bit row words"""
    from cvxpygen_amd.canon_builder import CanonBuilder
    rng = np.random.default_rng(11)
    G = rng.standard_normal((24, 20))
    cb = CanonBuilder('dense_lp')
    c = cb.param('c', (20,))
    h = cb.param('h', (24,))
    x = cb.var('x', (20,))
    for j in range(20):
        cb.lin(x[j], {c.idx(j): 1.0})
    rows = [cb.ineq([(x[j], float(G[i, j])) for j in range(20)], h[i]) for i in range(24)]
    for j in range(20):
        cb.ineq([(x[j], 1.0)], 1.0)
        cb.ineq([(x[j], -1.0)], 1.0)
    cb.dual('d0', rows, (24,))
    return cb.build({'c': rng.standard_normal(20), 'h': 1.0 + rng.random(24)}, solver='CLARABEL')


def test_generated_row_words_32_bit_in_emulator(tmp_path):
    """a family with more than 255 matrix entries: 32-bit row words (codegen.conic_row_tables), table-driven bits, oracle parity"""
    from tests.sim import build_sim
    from cvxpygen_amd import codegen
    d = _dense_lp()
    cp = build_conic_plan(d)
    assert cp.nnzP + cp.nnzA > 255 and 'typedef unsigned genc_row_word;' in codegen.conic_row_tables(cp)
    lib = build_sim.build_conic_family(cp, str(tmp_path), 'dlp')
    th = np.tile(d.theta0, (3, 1))
    th[1:, :-1] += 0.1 * np.random.default_rng(5).standard_normal((2, th.shape[1] - 1))
    bs = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
    bs.set_updated(None)
    r = bs.solve(theta_var=th[:, :-1])
    assert _fact(bs, 'specialised_kernel') == 1.0
    os.environ['CPG_CONIC_GENERATED'] = '0'
    try:
        bt = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
        bt.set_updated(None)
        rt = bt.solve(theta_var=th[:, :-1])
    finally:
        del os.environ['CPG_CONIC_GENERATED']
    assert _fact(bt, 'specialised_kernel') == 0.0
    assert (r.status == 1).all() and r.iter.tolist() == rt.iter.tolist()
    assert np.array_equal(r.sol_x, rt.sol_x) and np.array_equal(r.sol_y, rt.sol_y)
    _assert_parity(r, cl.cpg_solve_batch(d, th), tol=1e-8)
    bs.close(); bt.close()


def test_generated_row_words_in_emulator(tmp_path):
    """the generated row words of a conic family library (codegen.conic_row_tables: rows of P, columns and rows of A as padded
    per-lane lists, used by every sparse product and by the equilibration of the specialised kernel) on a family whose
    second-order-cone rows ARE rescaled by the equilibration and whose P is empty: the table-driven kernel's bits"""
    from tests.sim import build_sim
    from cvxpygen_amd.ecos_front import ecos_from_conic, conic_from_ecos
    from cvxpygen_amd import codegen
    d = conic_from_ecos(ecos_from_conic(families.adp_norm()))
    d = d[0] if isinstance(d, tuple) else d
    cp = build_conic_plan(d)
    assert '#define CPG_GENC_ROWS 1' in codegen.conic_row_tables(cp) and '#define CPG_GENC_ROWS_SP 0' in codegen.conic_row_tables(cp)
    lib = build_sim.build_conic_family(cp, str(tmp_path), 'adpn')
    assert '#define CPG_GENC_ROWS_HASH' in open(os.path.join(str(tmp_path), 'cpg_conic_adpn.h')).read()
    th = np.tile(d.theta0, (3, 1))
    th[1:, :-1] *= 1.0 + 0.05 * np.random.default_rng(3).standard_normal((2, th.shape[1] - 1))
    bs = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
    bs.set_updated(None)
    r = bs.solve(theta_var=th[:, :-1])
    assert _fact(bs, 'specialised_kernel') == 1.0
    os.environ['CPG_CONIC_GENERATED'] = '0'              # same library: table-driven executor, CSR / CSC loops
    try:
        bt = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
        bt.set_updated(None)
        rt = bt.solve(theta_var=th[:, :-1])
    finally:
        del os.environ['CPG_CONIC_GENERATED']
    assert _fact(bt, 'specialised_kernel') == 0.0
    assert (r.status == 1).all() and r.iter.tolist() == rt.iter.tolist()
    assert np.array_equal(r.sol_x, rt.sol_x) and np.array_equal(r.sol_y, rt.sol_y)
    _assert_parity(r, cl.cpg_solve_batch(d, th), tol=1e-9)
    bs.close(); bt.close()


def test_infeasible_and_lp_instances_in_emulator(sim_lib):
    """status integers of the conic path (Clarabel numbering) and the P == 0 initialisation"""
    d = families.toy_box(solver='CLARABEL')
    th = np.tile(d.theta0, (3, 1))
    th[1, :3] = [0.3, 1.0, -1.0]                        # lb > ub
    th[2, :3] = [5.0, -1.0, 1.0]
    bs = ConicBatchSolver(d, lib_path=sim_lib, full_output=True)
    r = bs.solve({'a': th[:, 0], 'lb': th[:, 1], 'ub': th[:, 2]})
    o = cl.cpg_solve_batch(d, th)
    assert o['status'].tolist() == [1, 2, 1]
    _assert_parity(r, o, tol=1e-9)
    bs.close()
    d = families.toy_lp(solver='CLARABEL')
    th = np.tile(d.theta0, (3, 1)); th[1, 0] = -1.0; th[2, 0] = 0.7
    bs = ConicBatchSolver(d, lib_path=sim_lib, full_output=True)
    r = bs.solve({'c': th[:, 0]})
    o = cl.cpg_solve_batch(d, th)
    assert o['status'].tolist() == [1, 3, 1]            # c < 0: unbounded below
    _assert_parity(r, o, tol=1e-9)
    bs.close()


def _almost_cases():
    """settings under which solves end at the iteration limit or on insufficient progress, where the reduced tolerances
    (cvxpygen/solvers/clarabel.py:79-84) decide between the "almost" statuses 4 / 5 / 6 and the raw 7 / 10"""
    d = families.adp()
    pv = _adp_batch(8, 1)
    th = _theta(d, pv)
    cases = [(d, pv, th, dict(max_iters=3)),            # a mix of AlmostSolved (4) and MaxIterations (7)
             (d, pv, th, dict(max_iters=4)),            # all AlmostSolved
             (d, pv, th, dict(max_iters=5)),            # Solved where 5 iterations are enough, AlmostSolved elsewhere
             # feasibility tolerance below round-off: the residuals stop improving -> insufficient progress, back to the
             # previous iterate, which passes the reduced tolerances
             # (what trips the test is round-off noise, which depends on the elimination order: the oracle's dense LDL' in
             # natural order and the kernel's sparse one agree on the status and the point, not on the exact iteration)
             (d, pv, th, dict(tol_feas=1e-16, tol_gap_abs=1e-3, tol_gap_rel=1e-3, _noise=True))]
    d3 = families.toy_lp(solver='CLARABEL')
    th3 = np.tile(d3.theta0, (3, 1)); th3[1, 0] = -1.0; th3[2, 0] = 0.7
    cases.append((d3, {'c': th3[:, 0]}, th3, dict(max_iters=4)))          # AlmostDualInfeasible (6) for the unbounded instance
    return cases


def _check_almost(lib_for):
    seen = set()
    for d, pv, th, stg in _almost_cases():
        stg = dict(stg)
        noise = stg.pop('_noise', False)
        bs = ConicBatchSolver(d, lib_path=lib_for(d), full_output=True)
        r = bs.solve(pv, **stg)
        o = cl.cpg_solve_batch(d, th, **{('max_iter' if k == 'max_iters' else k): v for k, v in stg.items()})
        assert r.status.tolist() == o['status'].tolist(), (stg, r.status, o['status'])
        if noise:       # the stopping iteration is decided by residuals at round-off level: both sides stop early, on a point
            assert (r.iter < 20).all() and (o['iter'] < 20).all() and (r.status == 4).all()      # good to the reduced tolerances
        else:
            assert r.iter.tolist() == o['iter'].tolist()
        tol = 1e-4 if noise else 1e-8
        fin = np.isin(o['status'], (1, 4))
        assert np.abs(r.sol_x[fin] - o['sol_x'][fin]).max() <= tol * max(1.0, np.abs(o['sol_x'][fin]).max())
        assert np.abs(r.obj_val[fin] - o['obj_val'][fin]).max() <= tol * max(1.0, np.abs(o['obj_val'][fin]).max())
        assert np.isnan(r.obj_val[np.isin(o['status'], (2, 3, 5, 6))]).all()
        seen |= set(o['status'].tolist())
        bs.close()
    assert {1, 4, 6, 7} <= seen


def test_almost_statuses_in_emulator(sim_lib):
    _check_almost(lambda d: sim_lib)


@pytest.mark.gpu
def test_almost_statuses_on_gpu():
    _check_almost(lambda d: None)


def test_generate_code_drop_in_for_conic_family(sim_lib, tmp_path):
    """tests/test_E2E_SOCP.py:119-121 shape: generate_code(prob, solver='CLARABEL') -> prob.solve(method='CPG')"""
    from cvxpygen_amd import cpg
    from cvxpygen_amd.lite import LiteProblem
    d = families.adp()
    prob = LiteProblem.from_descriptor(d)
    mod = cpg.generate_code(prob, code_dir=str(tmp_path / 'test_ADP_CLARABEL'), solver='CLARABEL', prefix='ADP_CLARABEL')
    mod._SOLVER.lib_path = sim_lib
    g = GOLD['ADP']['0']                                 # theta0 = seed 0 of the reference test
    val = prob.solve(method='CPG')
    assert abs(val - g['obj']) <= 1e-7 * g['obj'] and prob.value == val
    assert prob.status.startswith('1 (for description visit')          # integer status, utils.py:1598-1601
    assert np.abs(prob.var_dict['u'].value[0] - g['u0']).max() <= 1e-5
    assert abs(prob.constraints[0].dual_value[0] - g['dual_norm_u0']) <= 1e-6
    assert prob.solver_stats.solver_name == 'CLARABEL' and prob.solver_stats.num_iters <= 12
    # new parameter values, as assign_data(prob, name, seed=1) of the reference test does
    g1 = GOLD['ADP']['1']
    prob.param_dict['f'].value = np.array(g1['f'])
    prob.param_dict['G'].value = np.array(g1['G'])
    val = prob.solve(method='CPG', updated_params=['f', 'G'])
    assert abs(val - g1['obj']) <= 1e-7 * g1['obj']
    with pytest.raises(NotImplementedError):
        cpg.generate_code(d, code_dir=str(tmp_path / 'x'), solver='CLARABEL', gradient=True)
    with pytest.raises(ValueError, match='canonicalised for'):
        cpg.generate_code(d, code_dir=str(tmp_path / 'y'), solver='OSQP')


def test_settings_without_a_counterpart_are_accepted_at_their_defaults_only(sim_lib):
    """`time_limit`, `direct_kkt_solver`, `presolve_enable` of the reference's Clarabel interface (cvxpygen/solvers/clarabel.py:63-119):
    their defaults pass, a value that would change what the reference's solver does is refused -- not silently ignored"""
    cs = ConicBatchSolver(families.adp(), lib_path=sim_lib)
    cs.set_updated(None)                       # (the handle is created with the first set of updated parameters)
    cs.apply_settings(time_limit=1e10, direct_kkt_solver=1, presolve_enable=1, verbose=0)
    for bad in (dict(time_limit=0.5), dict(direct_kkt_solver=0), dict(presolve_enable=0)):
        with pytest.raises(NotImplementedError, match='default only'):
            cs.apply_settings(**bad)
    with pytest.raises(AttributeError):
        cs.apply_settings(no_such_setting=1)
    cs.close()


def test_conic_solver_rejects_wrong_family(sim_lib):
    with pytest.raises(ValueError, match='conic'):
        ConicBatchSolver(families.nonneg_ls(), lib_path=sim_lib)


# ------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
def test_adp_on_gpu_vs_oracle():
    d = families.adp()
    B = 500
    pv = _adp_batch(B, 7)
    bs = ConicBatchSolver(d, full_output=True)
    r = bs.solve(pv)
    o = cl.cpg_solve_batch(d, _theta(d, pv))
    assert (o['status'] == 1).all()
    _assert_parity(r, o)
    # only f varies: A stays at its code-generation-time values
    r2 = bs.solve({'f': pv['f']}, updated_params=['f'])
    th = np.tile(d.theta0, (B, 1)); th[:, 3:9] = pv['f']
    _assert_parity(r2, cl.cpg_solve_batch(d, th))
    bs.close()


@pytest.mark.gpu
def test_generated_conic_executor_on_gpu():
    """the ADP family library (__graft_entry__.build: generated executor of the substitution program) against the oracle
    and, bit for bit, against the table-driven executor of the generic library"""
    from cvxpygen_amd import codegen
    d = families.adp()
    cp = build_conic_plan(d)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = codegen.build_conic_library(cp, os.path.join(root, 'cvxpygen_amd', 'generated', 'adp'), 'adp')   # no-op when fresh
    pv = _adp_batch(300, 11)
    bs = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
    r = bs.solve(pv)
    assert _fact(bs, 'generated_executor') == 1.0 and _fact(bs, 'specialised_kernel') == 1.0
    _assert_parity(r, cl.cpg_solve_batch(d, _theta(d, pv)))
    pv = _adp_batch(5000, 12)                           # a batch larger than the resident wavefronts, executor against executor
    r = bs.solve(pv)
    bg = ConicBatchSolver(d, full_output=True)
    rg = bg.solve(pv)
    assert _fact(bg, 'generated_executor') == 0.0
    assert np.array_equal(r.sol_x, rg.sol_x) and np.array_equal(r.sol_y, rg.sol_y)
    assert r.iter.tolist() == rg.iter.tolist() and r.status.tolist() == rg.status.tolist()
    bs.close(); bg.close()


@pytest.mark.gpu
def test_adp_known_answers_on_gpu():
    d = families.adp()
    bs = ConicBatchSolver(d)
    for seed, g in GOLD['ADP'].items():
        r = bs.solve({'Rsqrt': np.array([g['Rsqrt_diag']]), 'f': np.array([g['f']]), 'G': np.array([g['G']])})
        assert r.status[0] == 1
        assert abs(r.obj_val[0] - g['obj']) <= 1e-7 * g['obj']
        assert np.abs(r.prim['u'][0][0] - g['u0']).max() <= 1e-5
        assert abs(r.dual['d0'][0][0] - g['dual_norm_u0']) <= 1e-6
    bs.close()


@pytest.mark.gpu
def test_nonneg_ls_in_conic_form_on_gpu():
    """QP with zero + nonnegative cones through the interior-point kernel: known NNLS answer, oracle parity"""
    g = GOLD['nonneg_LS']
    d = families.nonneg_ls(solver='CLARABEL')
    rng = np.random.default_rng(9)
    B = 333
    A = np.tile(np.array(g['A_data']), (B, 1)); b = np.tile(np.array(g['b']), (B, 1))
    A[1:] += 0.3 * rng.standard_normal((B - 1, 3)); b[1:] += rng.standard_normal((B - 1, 3))
    bs = ConicBatchSolver(d, full_output=True)
    r = bs.solve({'A': A, 'b': b})
    th = np.stack([d.theta_from_values({'A': A[k], 'b': b[k]}) for k in range(B)])
    _assert_parity(r, cl.cpg_solve_batch(d, th))
    assert np.abs(r.prim['x'][0] - g['x']).max() < 1e-7 and abs(r.obj_val[0] - g['obj']) < 1e-7
    bs.close()


@pytest.mark.gpu
def test_conic_statuses_on_gpu():
    d = families.toy_box(solver='CLARABEL')
    B = 300
    rng = np.random.default_rng(3)
    a, lb, ub = rng.standard_normal(B) * 2, -1 + 0.5 * rng.standard_normal(B), 1 + 0.5 * rng.standard_normal(B)
    swap = rng.random(B) < 0.3
    lb[swap], ub[swap] = ub[swap] + 0.5, lb[swap] - 0.5       # infeasible boxes
    th = np.tile(d.theta0, (B, 1)); th[:, 0], th[:, 1], th[:, 2] = a, lb, ub
    bs = ConicBatchSolver(d, full_output=True)
    r = bs.solve({'a': a, 'lb': lb, 'ub': ub})
    o = cl.cpg_solve_batch(d, th)
    assert set(o['status'].tolist()) == {1, 2}
    _assert_parity(r, o)
    assert np.allclose(r.prim['x'][o['status'] == 1, 0], np.clip(a, lb, ub)[o['status'] == 1], atol=1e-6)
    bs.close()


@pytest.mark.gpu
def test_adp_full_batch_properties():
    """BASELINE config 4 size (B = 100 000): every instance solved; objective equals the user's
    objective evaluated at the returned u; ||u_i|| <= 0.1; batch-order invariance; duplicates"""
    d = families.adp()
    B = 100_000
    rs = np.random.RandomState(11)
    states = -2 + 4 * rs.rand(B, 6)
    f = np.empty((B, 6)); G = np.zeros((B, 6, 3))
    f[:, :3] = states[:, :3] + 0.1 * states[:, 3:]
    f[:, 3:] = states[:, 3:] * (1 - 0.1 * states[:, 3:])
    G[:, 3, 0], G[:, 4, 1], G[:, 5, 2] = 0.1 * states[:, 3], 0.1 * states[:, 4], 0.1 * states[:, 5]
    chk = families.adp_values(states[0])
    assert np.allclose(f[0], chk['f']) and np.allclose(G[0], chk['G'])
    pv = {'f': f, 'G': G}
    bs = ConicBatchSolver(d)
    r = bs.solve(pv, updated_params=['f', 'G'])
    assert (r.status == 1).all() and r.iter.max() <= 15
    u0 = r.prim['u'][:, 0, :]
    obj = ((f + np.einsum('bij,bj->bi', G, u0)) ** 2).sum(axis=1) + 0.1 * (u0 ** 2).sum(axis=1)
    assert np.abs(obj - r.obj_val).max() <= 1e-6 * np.abs(obj).max()
    assert (np.linalg.norm(r.prim['u'], axis=2) <= 0.1 + 1e-7).all()
    perm = np.random.default_rng(0).permutation(B)
    r2 = bs.solve({'f': f[perm], 'G': G[perm]}, updated_params=['f', 'G'])
    assert (r2.iter == r.iter[perm]).all() and np.array_equal(r2.prim_flat, r.prim_flat[perm])
    bs.close()


def test_c_restatement_of_the_conic_oracle_follows_the_numpy_one():
    """oracle/clarabel_oracle.c (the CPU baseline of the conic workload) against oracle/clarabel_numpy.py, which shares no
    code with it beyond the algorithm: identical iteration counts and statuses, solutions to round-off on the ADP family of
    BASELINE config 4 (quadratic objective), objective and duals on the norm form whose optimum sits at a kink of the
    cones (x there is determined to ~1e-5 only), an infeasible instance, and a settings override"""
    from oracle import binding as ob
    from oracle import clarabel_numpy as cl
    rs = np.random.RandomState(11)
    states = -2 + 4 * rs.rand(10, 6)
    for d, xtol in ((families.adp(), 1e-10), (families.adp_norm(tie=True), 1e-4), (families.adp_norm(tie=False), 1e-4)):
        th = np.stack([d.theta_from_values(families.adp_values(s_)) for s_ in states])
        o = cl.cpg_solve_batch(d, th)
        c = ob.clarabel_solve_batch(d, th, nthreads=2)
        assert c['iter'].tolist() == o['iter'].tolist() and c['status'].tolist() == o['status'].tolist()
        assert np.abs(c['obj_val'] - o['obj_val']).max() <= 1e-9 * max(1.0, np.abs(o['obj_val']).max())
        assert np.abs(c['sol_x'] - o['sol_x']).max() <= xtol * max(1.0, np.abs(o['sol_x']).max())
        assert np.abs(c['sol_z'] - o['sol_z']).max() <= 1e-7 * max(1.0, np.abs(o['sol_z']).max())
        assert np.allclose(c['pri_res'], o['pri_res'], rtol=1e-2, atol=1e-10) and np.allclose(c['dua_res'], o['dua_res'], rtol=1e-2, atol=1e-10)   # (round-off-level figures)
    d = families.adp()
    th = np.stack([d.theta_from_values(families.adp_values(s_)) for s_ in states[:3]])
    o = cl.cpg_solve_batch(d, th, max_iter=3)
    c = ob.clarabel_solve_batch(d, th, max_iter=3)
    assert c['iter'].tolist() == o['iter'].tolist() == [3, 3, 3] and c['status'].tolist() == o['status'].tolist()
    with pytest.raises(KeyError):
        ob.clarabel_solve_batch(d, th, no_such_setting=1)
