"""Host logic (CPU): family descriptors, code-generation-time setup, solve-program compiler."""
import os
import tempfile

import numpy as np
import pytest
import scipy.sparse as sp

from cvxpygen_amd import families, osqp_setup as S, solve_program as SP, ordering as O
from cvxpygen_amd.canon_builder import canon_lu
from cvxpygen_amd.descriptor import FamilyDescriptor
from cvxpygen_amd.runtime import build_family_plan
from cvxpygen_amd.sharding import shard_bounds
from oracle.osqp_numpy import ruiz


# canonical dimensions of SURVEY.md Appendix B
@pytest.mark.parametrize('make,dims', [
    (lambda: families.nonneg_ls(), dict(n_var=5, n_eq=3, n_ineq=2, NP=6, prim=2, dual=2)),
    (lambda: families.nonneg_ls(10, 5, sparsity=None, seed=0), dict(n_var=15, n_eq=10, n_ineq=5, NP=60, prim=5, dual=5)),
    (lambda: families.mpc(6, 3, 10), dict(n_var=222, n_eq=162, n_ineq=90, NP=141, prim=96, dual=96)),
    (lambda: families.mpc(12, 4, 10), dict(n_var=384, n_eq=304, n_ineq=120, NP=508, prim=172, dual=172)),
    (lambda: families.portfolio(100, 10), dict(n_var=620, n_eq=221, n_ineq=601, NP=1601, prim=210, dual=112)),
])
def test_family_dimensions(make, dims):
    d = make()
    assert (d.n_var, d.n_eq, d.n_ineq, d.NP, d.n_prim_user, d.n_dual_user) == \
        (dims['n_var'], dims['n_eq'], dims['n_ineq'], dims['NP'], dims['prim'], dims['dual'])
    c = d.default_canon()
    assert np.allclose(c['A'], d.A.data) and np.allclose(c['P'], d.P.data)
    # P never depends on parameters for these families
    assert d.changes['P'] is False


def test_descriptor_roundtrip_and_dependencies():
    d = families.mpc(6, 3, 10)
    dep = d.user_p_name_to_canon_outdated()
    assert dep['x_init'] == ['l', 'u'] and dep['A'] == ['A'] and dep['Qsqrt'] == ['A']
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'fam.npz')
        d.save(path)
        e = FamilyDescriptor.load(path)
    assert (e.A != d.A).nnz == 0 and np.array_equal(e.theta0, d.theta0)
    assert [p.name for p in e.params] == [p.name for p in d.params]
    assert all((e.maps[k] != d.maps[k]).nnz == 0 for k in d.maps)
    with pytest.raises(AttributeError):
        d.param('nope')


def test_param_flattening_matches_reference_conventions():
    d = families.mpc(6, 3, 10, sparse_params=True)
    M = np.arange(36.0).reshape(6, 6)
    assert np.array_equal(d.flatten_param('Qsqrt', M), np.diag(M))                 # diag=True
    r, c = d.param('A').sparsity
    assert np.array_equal(d.flatten_param('A', M), M[np.array(r), np.array(c)])    # sparsity attr
    d2 = families.mpc(6, 3, 10)
    assert np.array_equal(d2.flatten_param('A', M), M.flatten(order='F'))          # dense: F-order


def test_ruiz_scaling_matches_independent_restatement():
    d = families.mpc(6, 3, 10)
    c = d.default_canon()
    Px, q, Ax, sc = S.ruiz_scale(d.P, c['q'], d.A, 10)
    Pf = (d.P + sp.triu(d.P, 1).T).toarray()
    P2, q2, A2, D2, E2, c2 = ruiz(Pf, c['q'], d.A.toarray(), 10)
    assert np.allclose(sc.D, D2, rtol=1e-13) and np.allclose(sc.E, E2, rtol=1e-13) and abs(sc.c - c2) < 1e-13 * c2
    As = sp.csc_matrix((Ax, d.A.indices, d.A.indptr), shape=d.A.shape).toarray()
    assert np.allclose(As, A2, rtol=1e-12, atol=1e-15)
    # equilibrated: KKT column norms close to one another
    K = np.block([[P2, A2.T], [A2, np.zeros((d.m, d.m))]])
    norms = np.abs(K).max(axis=0)
    assert norms.max() / norms[norms > 0].min() < 50


@pytest.mark.parametrize('method', ['mindeg', 'nd'])
def test_factorisation_solves_kkt(method):
    d = families.mpc(6, 3, 10)
    c = d.default_canon(); l, u = canon_lu(d, c)
    plan = S.setup(d.P, c['q'], d.A, l, u, ordering=method)
    assert sorted(plan.perm.tolist()) == list(range(plan.N))
    Kf = (plan.K + sp.triu(plan.K, 1).T).toarray()
    rng = np.random.default_rng(0)
    b = rng.standard_normal(plan.N)
    x = S.ldl_solve(plan, b)
    assert np.abs(Kf @ x - b).max() < 1e-7
    assert (plan.D[:] != 0).all()
    # quasi-definite: n positive and m negative pivots
    assert (plan.D > 0).sum() == d.n_var and (plan.D < 0).sum() == d.m


@pytest.mark.parametrize('fam', ['nnls', 'mpc6'])
@pytest.mark.parametrize('merge', [False, True])
def test_solve_program_equals_substitution(fam, merge):
    d = families.nonneg_ls() if fam == 'nnls' else families.mpc(6, 3, 10)
    c = d.default_canon(); l, u = canon_lu(d, c)
    plan = S.setup(d.P, c['q'], d.A, l, u, ordering='mindeg')
    N = plan.N
    phases = SP.compile_ldl(N, plan.Lp, plan.Li, plan.Lx, plan.D, plan.perm, merge=merge)
    prog = SP.pack(phases, N=N)
    assert prog.n_slots >= N and prog.cols.max() < prog.n_slots
    rng = np.random.default_rng(1)
    for _ in range(3):
        b = rng.standard_normal(N)
        w = np.zeros(prog.n_slots); w[:N] = b
        SP.execute_packed(prog, w)
        ref = S.ldl_solve(plan, b)
        assert np.abs(w[prog.final_pos] - ref).max() <= 1e-11 * np.abs(ref).max()
    if merge and fam == 'mpc6':
        unmerged = SP.pack(SP.compile_ldl(N, plan.Lp, plan.Li, plan.Lx, plan.D, plan.perm, merge=False), N=N)
        assert prog.n_phases * 8 < unmerged.n_phases      # the point of the transformation


@pytest.mark.parametrize('fam', ['nnls', 'mpc6', 'portfolio'])
@pytest.mark.parametrize('layout', [False, True, 'auto'])
def test_ragged_layouts_equal_substitution(fam, layout):
    """the ragged packing of the LDS-resident / streamed executors in its three lane layouts
    (power-of-two groups, segmented rows, per-phase choice) replays to the plain LDL' solve"""
    d = {'nnls': families.nonneg_ls, 'mpc6': lambda: families.mpc(6, 3, 10),
         'portfolio': lambda: families.portfolio(20, 4)}[fam]()
    c = d.default_canon(); l, u = canon_lu(d, c)
    plan = S.setup(d.P, c['q'], d.A, l, u, ordering='mindeg')
    N = plan.N
    phases = SP.compile_ldl(N, plan.Lp, plan.Li, plan.Lx, plan.D, plan.perm, merge=False)
    prog = SP.pack_ragged(phases, N, balanced=layout)
    kinds = set(prog.ctab[:, 3].tolist())
    assert kinds <= {0, 1} and (layout is not False or kinds == {0})
    # the streaming executor addresses entries as base + lane: active lanes must form a prefix
    for ch in range(prog.n_chunks):
        ln = (prog.desc[ch] >> 16) & (0xFFF if prog.ctab[ch, 3] & 1 else 0xFFFF)
        assert np.all(np.diff(ln.astype(np.int64)) <= 0)
    rng = np.random.default_rng(3)
    for _ in range(2):
        b = rng.standard_normal(N)
        w = np.zeros(prog.n_slots); w[:N] = b
        SP.execute_ragged(prog, w)
        ref = S.ldl_solve(plan, b)
        # (another summation order than the column sweeps of ldl_solve; the portfolio KKT matrix
        # with rho = 0.1, sigma = 1e-6 is the ill-conditioned one)
        assert np.abs(w[prog.final_pos] - ref).max() <= 1e-9 * np.abs(ref).max()
    if layout == 'auto':
        steps = lambda q: int(q.ctab[:, 0].sum()) + 3 * q.n_chunks
        # segmented rows are only taken where they clearly win: never worse than the uniform layout
        assert steps(prog) <= 1.05 * steps(SP.pack_ragged(phases, N, balanced=False))


def test_family_plan_device_ordering():
    d = families.mpc(6, 3, 10)
    p = build_family_plan(d)
    # the 6 rows X[:,0] == x_init are the only parameter-dependent bounds: they come first
    assert p.n_vary_z == 6 and p.n_vary_x == 0
    init_rows = d.duals[2].indices
    assert sorted(p.ordz[:6].tolist()) == sorted(init_rows.tolist())
    assert sorted(p.ordx.tolist()) == list(range(d.n_var)) and sorted(p.ordz.tolist()) == list(range(d.m))
    # natural programs reproduce the scaled products in device order
    o = p.osqp
    As = sp.csc_matrix((o.Ax, d.A.indices, d.A.indptr), shape=d.A.shape)
    rng = np.random.default_rng(2)
    xv = rng.standard_normal(d.n_var); yv = rng.standard_normal(d.m)
    w = np.concatenate([xv[p.ordx], yv[p.ordz]])
    ax = np.concatenate(SP.execute_packed(p.A_rows, w, natural=True))[:d.m]
    aty = np.concatenate(SP.execute_packed(p.At_rows, w, natural=True))[:d.n_var]
    assert np.allclose(ax, (As @ xv)[p.ordz], atol=1e-12)
    assert np.allclose(aty, (As.T @ yv)[p.ordx], atol=1e-12)


def test_shard_bounds_cover_batch():
    for B in (0, 1, 7, 100000, 1000003):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(B, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == B
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('make', [lambda: families.nonneg_ls(), lambda: families.mpc(6, 3, 10),
                                  lambda: families.portfolio(8, 3), lambda: families.adp(),
                                  lambda: families.toy_lp(solver='CLARABEL')])
def test_canonicalizer_core_round_trip(make):
    """the cvxpy-free core of the cvxpy front door: [A | b] entry maps in cvxpy's sign convention ->
    per-canonical-parameter maps (cvxpygen/solvers/_interface.py:39-79, 132-173); driven with the
    arrays rebuilt from a hand-made descriptor (cvxpy itself is not installed)"""
    from cvxpygen_amd import canonicalizer as cz
    d = make()
    red_P, P_index, q_map, red_A, A_index = cz.reduced_from_descriptor(d)
    d2 = cz.descriptor_from_reduced(d.name, d.solver, d.n_var, d.n_eq, d.n_ineq, red_P, P_index, q_map, red_A,
                                    A_index, d.theta0, d.params, d.variables, d.duals, d.is_maximization, d.cones)
    assert (d2.A.indptr == d.A.indptr).all() and (d2.A.indices == d.A.indices).all()
    assert np.allclose(d2.A.data, d.A.data) and np.allclose(d2.P.toarray(), d.P.toarray())
    rng = np.random.default_rng(0)
    th = d.theta0.copy(); th[:-1] += rng.standard_normal(d.NP)
    c1, c2 = d.canon_at(th), d2.canon_at(th)
    assert set(c1) == set(c2)
    for k in c1:
        assert np.allclose(c1[k], c2[k]), k
    assert d2.changes == d.changes and d2.nonzero_d == d.nonzero_d


def test_bank_aware_slot_numbering():
    """cvxpygen_amd/slot_layout.py: the numbering is a permutation inside every region, its cost bookkeeping is
    exact (gathers: 32-lane groups, slots collide modulo 32; reduce-stores: 16-lane groups, modulo 16 -- the model
    calibrated with scripts/micro/lds_conflicts.hip), and the plan built with it executes to the same solution"""
    from cvxpygen_amd import slot_layout as sl
    rng = np.random.default_rng(0)
    n = 200
    region = np.repeat(np.arange(4), 50)
    steps = rng.integers(0, n, size=(30, 64))
    outs = np.where(rng.random((6, 64)) < 0.7, rng.integers(0, n, size=(6, 64)), 0xFFFF)
    stores = sl.store_groups(outs, 0xFFFF)
    assert all(len(g) > 1 and (g != 0xFFFF).all() for g in stores)
    pi, c0, c1 = sl.optimise(steps, region, sweeps=20, seed=1, stores=stores)
    assert sorted(pi.tolist()) == list(range(n)) and c1 < c0
    for r in range(4):
        assert sorted(pi[region == r].tolist()) == list(range(50 * r, 50 * r + 50))
    groups = sl.gather_groups(steps)
    mods = [sl.BANK_PAIRS] * len(groups) + [sl.STORE_BANK_PAIRS] * len(stores)
    assert sl.conflict_cycles(groups + stores, pi, mods) == c1
    assert sl.conflict_cycles(groups + stores, np.arange(n), mods) == c0
    # the two banking rules on hand-made patterns: stride-2 slots collide pairwise in a 32-lane gather group,
    # consecutive slots never; 16 consecutive slots of a store group never, stride 16 always
    lanes = np.arange(32)
    assert sl.conflict_cycles([2 * lanes], np.arange(64)) == 1 and sl.conflict_cycles([lanes], np.arange(64)) == 0
    assert sl.conflict_cycles([np.arange(16)], np.arange(256), [16]) == 0
    assert sl.conflict_cycles([16 * np.arange(16)], np.arange(256), [16]) == 15
    # a family plan with and without the numbering: same solve program up to the relabelling
    d = families.mpc(2, 1, 3)
    pa, pb = build_family_plan(d, bank_layout=True), build_family_plan(d, bank_layout=False)
    assert pa.stats['bank_conflict_cycles'] <= pa.stats['bank_conflict_cycles_natural']
    assert pa.kkt_ragged.nnz == pb.kkt_ragged.nnz and pa.kkt_ragged.n_slots == pb.kkt_ragged.n_slots


@pytest.mark.parametrize('opts', [{}, dict(batch=2, early=9, cross=9, depth=3), dict(pad_offsets=False, early=0, cross=0),
                                  dict(group_offsets=2), dict(group_offsets=1, depth=1)])
def test_generated_executor_schedule_invariants(opts):
    """cvxpygen_amd/codegen.emit_program_header, without compiling anything: every step's offset / coefficient load,
    gather and multiply-add appear exactly once and in that order; multiply-adds of a chunk keep their order; a
    gather that reads what the previous phase stores comes after that phase's reduce / store, every other gather
    may come before (`early`); every chunk is stored exactly once, after its last multiply-add"""
    import re
    from cvxpygen_amd import codegen
    d = families.mpc(6, 3, 10)
    plan = build_family_plan(d)
    rp = plan.kkt_ragged
    text = codegen.emit_program_header(rp, 'mpc6', plan, pad_offsets=opts.pop('pad_offsets', True), **opts)
    steps = SP.execution_steps(rp)
    T = len(steps)
    body = text[text.index('run_program_gen'):]
    ev = []                                    # (kind, id) in program order
    for ln in body.splitlines():
        m = re.match(r'\s*CPG_GEN_LOAD_CV\w*\((\d+),', ln)
        if m: ev.append(('C', int(m.group(1)))); continue
        m = re.match(r'\s*CPG_GEN_LOAD_W\((\d+)\)', ln)
        if m: ev.append(('W', int(m.group(1)))); continue
        m = re.match(r'\s*CPG_GEN_FMA_\w+\(a(\d+), (\d+)', ln)
        if m: ev.append(('F', int(m.group(2)), int(m.group(1)))); continue
        m = re.match(r'\s*CPG_GEN_(?:SEG)?REDUCE_STORE(?:_LIT)?\(a(\d+),', ln)
        if m: ev.append(('S', int(m.group(1))))
    pos = {}
    for k, e in enumerate(ev):
        assert (e[0], e[1]) not in pos, e
        pos[(e[0], e[1])] = k
    assert all(('C', t) in pos and ('W', t) in pos and ('F', t) in pos for t in range(T))
    assert all(pos[('C', t)] < pos[('W', t)] < pos[('F', t)] for t in range(T))
    assert all(('S', c) in pos for c in range(rp.n_chunks))
    fma_order = [e[1] for e in ev if e[0] == 'F']
    assert fma_order == sorted(fma_order)                       # multiply-adds in execution order
    assert [e[1] for e in ev if e[0] == 'W'] == list(range(T))  # gathers too
    chunk_of = {t: st[1] for t, st in enumerate(steps)}
    for e in ev:
        if e[0] == 'F':
            assert e[2] == chunk_of[e[1]] and pos[('F', e[1])] < pos[('S', e[2])]
    # dependences through the work vector
    outs = {}
    for c in range(rp.n_chunks):
        outs.setdefault(int(rp.chunk_phase[c]), set()).update(int(x) for x in (rp.desc[c] & 0xFFFF) if int(x) != 0xFFFF)
    phase_ids = sorted(outs)
    n_early = 0
    for t, (pi_, c, e0, cnt) in enumerate(steps):
        if pi_ == 0:
            continue
        prev_chunks = [c2 for c2 in range(rp.n_chunks) if phase_ids.index(int(rp.chunk_phase[c2])) == pi_ - 1]
        prev_end = max(pos[('S', c2)] for c2 in prev_chunks)
        reads_prev = any(int(x) // 8 in outs[phase_ids[pi_ - 1]] for x in rp.cols[e0:e0 + cnt])
        if reads_prev:
            assert pos[('W', t)] > prev_end
        elif pos[('W', t)] < prev_end:
            n_early += 1
        # never earlier than the phase before the previous one has stored
        if pi_ >= 2:
            pp = [c2 for c2 in range(rp.n_chunks) if phase_ids.index(int(rp.chunk_phase[c2])) == pi_ - 2]
            assert pos[('W', t)] > max(pos[('S', c2)] for c2 in pp)
    assert (n_early > 0) == bool(opts.get('early', 4))


@pytest.mark.parametrize('make', [lambda: families.mpc(12, 4, 10), lambda: families.mpc(6, 3, 10),
                                  lambda: families.mpc(8, 3, 7), lambda: families.nonneg_ls(40, 20, sparsity=None, seed=1)])
def test_coefficient_register_sharing_of_the_instance_executor(make):
    """codegen.pack_step_registers: narrow chunks of the per-instance substitution program are shifted to lane offsets so
    that several steps share one coefficient register of the generated instance executor.  Host-side invariants the
    kernel relies on: no two steps on one lane of a register; every step of a chunk at the chunk's shift; shifts keep
    reduction groups inside their DPP rows; the emitted header carries the same map the C side rebuilds its tables
    from; and the executor's arithmetic (simulated here lane by lane on a shifted layout) is the program's."""
    import re
    from cvxpygen_amd import codegen, refactor_plan as _rp
    from cvxpygen_amd.runtime import build_family_plan
    from cvxpygen_amd.solve_program import execution_steps
    d = make()
    plan = build_family_plan(d)
    o = plan.osqp_shared or plan.osqp
    Ps, As = o.pruned(d.P, d.A)
    rpl = _rp.shared_mode_plan(Ps, As, o)
    assert codegen.instance_program_fits(rpl)
    rp = rpl.sol
    steps = execution_steps(rp)
    reg, shift, n_regs = codegen.pack_step_registers(rp, steps)
    assert n_regs <= len(steps) and max(reg) == n_regs - 1
    taken = {}
    for t, (_, c, e, cnt) in enumerate(steps):
        lg, kind = int(rp.ctab[c, 1]), int(rp.ctab[c, 3])
        width = max(s_[3] for s_ in steps if s_[1] == c)
        gran = 64 if lg >= 6 else 32 if lg == 5 else 8 if (width <= 8 and lg <= 3) else 16      # DPP rows; half rows for tiny chunks
        assert shift[c] % gran == 0 and shift[c] + cnt <= 64
        assert shift[c] // 16 == (shift[c] + width - 1) // 16 or gran >= 16                     # a half-row chunk stays in its row
        for l in range(shift[c], shift[c] + cnt):
            assert (reg[t], l) not in taken, (t, taken.get((reg[t], l)))
            taken[(reg[t], l)] = t
    # rows (output lanes) of a shifted chunk stay inside the wavefront
    for c in range(rp.n_chunks):
        used = np.nonzero((rp.desc[c] & 0xFFFF) != 0xFFFF)[0]
        assert used.size == 0 or used.max() + shift[c] < 64
    if len(steps) > 64:                                   # the headline families do share registers
        assert n_regs < len(steps) and n_regs <= codegen.GENI_MAX_REGS
    # the header carries exactly this map
    hdr = codegen.emit_instance_program(rp, 'fam')
    assert f'#define CPG_GENI_NREGS {n_regs}' in hdr
    tup = re.search(r'#define CPG_GENI_STEPS \{(.*)\}', hdr).group(1)
    got = [tuple(int(v) for v in m) for m in re.findall(r'\{(\d+), (\d+), (\d+), (\d+)\}', tup)]
    assert got == [(e, cnt, reg[t], shift[c]) for t, (_, c, e, cnt) in enumerate(steps)]
    assert set(re.findall(r'cf\[(\d+)\]', hdr.split('namespace cpg {')[1])) == {str(r) for r in set(reg)}


def test_plans_built_from_worker_threads_equal_the_sequential_ones():
    """__graft_entry__.build() and generate_code plan families from worker threads while pack_ragged(stage_scale=...)
    changes the planner's stage costs for the duration of a call: the costs are per thread (solve_program._costs), so the
    plans of concurrent builders do not see each other's scaled values and nobody serialises on a lock.  With module-level
    costs a family's solve program came out planned with another call's costs -- a header whose fingerprint no later
    process reproduces ("generated for a different problem family")."""
    from concurrent.futures import ThreadPoolExecutor
    from cvxpygen_amd import refactor_plan as _rp, solve_program as _spm
    from cvxpygen_amd.conic_plan import build_conic_plan
    from cvxpygen_amd.runtime import build_family_plan
    makes = [lambda: families.mpc(6, 3, 10), lambda: families.mpc(8, 3, 7), lambda: families.nonneg_ls(40, 20, sparsity=None, seed=1)]

    def prints(make):
        d = make()
        plan = build_family_plan(d)
        o = plan.osqp_shared or plan.osqp
        Ps, As = o.pruned(d.P, d.A)
        return (plan.kkt_ragged.fingerprint(), _rp.shared_mode_plan(Ps, As, o).sol.fingerprint(),
                _rp.build_refactor_plan(d.P, d.A, plan.osqp).sol.fingerprint())
    seq = [prints(mk) for mk in makes]
    conic = build_conic_plan(families.adp()).sol.fingerprint()
    costs = (_spm._costs.stage, _spm._costs.group)
    assert costs == (_spm.STAGE_COST, _spm.GROUP_STAGE_COST)
    for _ in range(2):
        with ThreadPoolExecutor(max_workers=7) as ex:
            futs = [ex.submit(prints, mk) for mk in makes + makes]
            fc = ex.submit(lambda: build_conic_plan(families.adp()).sol.fingerprint())
            got = [f.result() for f in futs]
            assert fc.result() == conic
        assert got == seq + seq
        assert (_spm._costs.stage, _spm._costs.group) == costs            # nothing left scaled


def test_cvxpy_front_door_forwards_solver_opts_and_routes_ecos(tmp_path):
    """`generate_code(cvxpy.Problem, solver=..., solver_opts=...)`: solver_opts are cvxpy's canonicalisation options and
    reach `problem.get_problem_data` verbatim (cvxpygen/canonicalizer.py:86-94), the OSQP build options travel under
    their own keyword; solver='ECOS' goes through cvxpy's ECOS chain and comes out in the ECOS form c, d, A, b, G, h
    with y / z duals (solvers/ecos.py:20-22, 75-84); `use_quad_obj` is honoured (canonicalizer.py:418-426).  cvxpy is
    not in the image: the problem objects are the test double of tests/sim/fake_cvxpy.py, built from hand-canonicalised
    families, so this checks the front door's plumbing, not cvxpy's canonicalisation."""
    import json
    import warnings
    from sim import fake_cvxpy
    from cvxpygen_amd import cpg
    from cvxpygen_amd.descriptor import FamilyDescriptor
    from cvxpygen_amd.ecos_front import ecos_from_conic
    remove = fake_cvxpy.install()
    try:
        def same(a, b):
            assert (a.solver, a.n_var, a.n_eq, a.n_ineq, a.is_maximization) == (b.solver, b.n_var, b.n_eq, b.n_ineq, b.is_maximization)
            assert set(a.maps) == set(b.maps)
            for k in a.maps:
                assert abs(sp.csr_matrix(a.maps[k]) - sp.csr_matrix(b.maps[k])).max() == 0, k
            assert np.array_equal(a.A.indices, b.A.indices) and np.array_equal(a.A.indptr, b.A.indptr)
            assert [(v.name, v.indices.tolist()) for v in a.variables] == [(v.name, v.indices.tolist()) for v in b.variables]
            assert [(u.vec, np.ravel(u.indices).tolist()) for u in a.duals] == [(u.vec, np.ravel(u.indices).tolist()) for u in b.duals]
            assert [(q.name, q.col, q.size) for q in a.params] == [(q.name, q.col, q.size) for q in b.params]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)
            # OSQP: solver_opts forwarded, build options under their own keyword
            d = families.mpc(2, 1, 3)
            prob = fake_cvxpy.Problem(d)
            opts = {'use_quad_obj': True, 'some_cvxpy_option': 7}
            cpg.generate_code(prob, code_dir=str(tmp_path / 'q'), solver='OSQP', solver_opts=opts, wrapper=False,
                              osqp_build_options={'adaptive_rho': 0, 'check_dualgap': 0})
            assert prob.calls == [dict(solver='OSQP', gp=False, enforce_dpp=True, verbose=False, solver_opts=opts)]
            same(FamilyDescriptor.load(str(tmp_path / 'q' / 'descriptor.npz')), d)
            assert json.load(open(tmp_path / 'q' / 'osqp_build.json')) == {'adaptive_rho': 0.0, 'check_dualgap': 0.0}
            # solver_opts are NOT read as build options any more: a build option left there would silently select nothing -> refused
            with pytest.raises(ValueError, match='osqp_build_options'):
                cpg.generate_code(fake_cvxpy.Problem(d), code_dir=str(tmp_path / 'q2'), solver='OSQP', solver_opts={'adaptive_rho': 0}, wrapper=False)
            with pytest.raises(ValueError, match='unknown OSQP build option'):
                cpg.generate_code(fake_cvxpy.Problem(d), code_dir=str(tmp_path / 'q3'), solver='OSQP', wrapper=False,
                                  osqp_build_options={'use_quad_obj': False})
            # CLARABEL and ECOS from the same conic problem
            c = families.adp()
            pc = fake_cvxpy.Problem(c)
            cpg.generate_code(pc, code_dir=str(tmp_path / 'c'), solver='CLARABEL', wrapper=False)
            assert pc.calls[0]['solver'] == 'CLARABEL' and pc.calls[0]['solver_opts'] is None
            same(FamilyDescriptor.load(str(tmp_path / 'c' / 'descriptor.npz')), c)
            c = families.adp_norm(tie=True)                    # (no quadratic objective: ECOS has none)
            pe = fake_cvxpy.Problem(c)
            cpg.generate_code(pe, code_dir=str(tmp_path / 'e'), solver='ECOS', solver_opts={'x': 1}, wrapper=False)
            assert pe.calls == [dict(solver='ECOS', gp=False, enforce_dpp=True, verbose=False, solver_opts={'x': 1})]
            e = FamilyDescriptor.load(str(tmp_path / 'e' / 'descriptor.npz'))
            assert e.solver == 'ECOS' and set(e.maps) == {'c', 'd', 'A', 'b', 'G', 'h'} and {u.vec for u in e.duals} <= {'y', 'z'}
            same(e, ecos_from_conic(c))
    finally:
        remove()
    # the same routing from the cvxpy-free front door: a conic LiteProblem asked for solver='ECOS'
    from cvxpygen_amd.lite import LiteProblem
    c = families.adp_norm(tie=True)
    lp = LiteProblem.from_descriptor(c)
    cpg.generate_code(lp, code_dir=str(tmp_path / 'le'), solver='ECOS', wrapper=False)
    e = FamilyDescriptor.load(str(tmp_path / 'le' / 'descriptor.npz'))
    assert e.solver == 'ECOS' and e.n_eq == c.cones['zero'] and e.n_ineq == c.m - c.cones['zero']
    with pytest.raises(ValueError, match='canonicalised for OSQP'):
        cpg.generate_code(LiteProblem.from_descriptor(families.mpc(2, 1, 3)), code_dir=str(tmp_path / 'bad'), solver='ECOS', wrapper=False)


def test_library_staleness_is_decided_by_content_not_by_times(tmp_path):
    """codegen.compile_if_stale: a library is rebuilt when the command or the CONTENT of a dependency changed -- digests taken
    before the compiler starts -- and not otherwise: a header edited while a long compile runs leaves a library that is newer
    than the edit without containing it (times alone called it fresh), and a copy of the checkout to another box changes every
    time stamp (and possibly the path) without changing anything that matters"""
    import time
    from cvxpygen_amd import codegen
    src = tmp_path / 'a.c'
    hdr = tmp_path / 'a.h'
    out = str(tmp_path / 'liba.so')
    hdr.write_text('#define V 1\n')
    src.write_text('#include "a.h"\nint v(void) { return V; }\n')
    cmd = ['gcc', '-shared', '-fPIC', '-o', out, str(src)]

    def build2():
        before = os.stat(out).st_mtime_ns if os.path.exists(out) else None
        time.sleep(0.02)
        codegen.compile_if_stale(cmd, out, [str(src), str(hdr)])
        return before is None or os.stat(out).st_mtime_ns != before
    assert build2()                                   # first build
    time.sleep(0.05)
    assert not build2()                               # nothing changed
    os.utime(hdr, None); os.utime(src, None)          # newer time stamps, same content (a copied checkout)
    assert not build2()
    # an edit "during the compile": the library ends up NEWER than the header and must still count as stale
    hdr.write_text('#define V 2\n')
    os.utime(out, None)
    assert os.path.getmtime(out) >= os.path.getmtime(hdr)
    assert build2()
    assert not build2()
    # another command (flags) -> rebuild; an interrupted build leaves no stamp behind
    cmd.insert(1, '-O1')
    assert build2()
    bad = ['gcc', '-shared', '-fPIC', '-o', out, str(tmp_path / 'missing.c')]
    with pytest.raises(Exception):
        codegen.compile_if_stale(bad, out, [str(src), str(hdr)])
    assert not os.path.exists(out + '.flags')
    assert build2()
