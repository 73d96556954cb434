"""Row C1, PSD cones of the reference's Clarabel path (`cvxpygen/solvers/clarabel.py:138, 146, 320-323`:
ClarabelPSDTriangleConeT).  As for the exponential / power cones (tests/test_nonsym_cones.py) the reference holds no problem with
this cone and Clarabel itself is absent; pinned here by mathematics independent of the restatement -- the Nesterov-Todd identities
of the scaling, closed-form optima (smallest eigenvalue, projection onto the cone) -- then the kernel against the oracle.  A PSD cone
is a symmetric cone: the comparison is tight to the end (iteration counts equal, 1e-9), except where an exponential cone in the same
family brings its own conditioning (trace_sdp: 1e-6)."""
import os

import numpy as np
import pytest

from cvxpygen_amd import families
from cvxpygen_amd.conic_plan import build_conic_plan
from cvxpygen_amd.conic_runtime import ConicBatchSolver
from oracle import clarabel_numpy as cl

np.seterr(all='ignore')


def _theta(desc, pv):
    B = next(iter(pv.values())).shape[0]
    return np.stack([desc.theta_from_values({k: v[i] for k, v in pv.items()}) for i in range(B)])


def _sym(rs, B, p):
    G = rs.randn(B, p, p)
    return G + G.transpose(0, 2, 1)


def _clip(C):
    w, V = np.linalg.eigh(C)
    return (V * np.maximum(w, 0.0)) @ V.T


def _assert_parity(r, o, tol=1e-9):
    assert r.iter.tolist() == o['iter'].tolist() and r.status.tolist() == o['status'].tolist()
    assert np.abs(r.sol_x - o['sol_x']).max() <= tol * max(1.0, np.abs(o['sol_x']).max())
    assert np.abs(r.sol_y - o['sol_z']).max() <= tol * max(1.0, np.abs(o['sol_z']).max())
    assert np.abs(r.obj_val - o['obj_val']).max() <= tol * max(1.0, np.abs(o['obj_val']).max())


# ------------------------------------------------------------------------------------ the scaling (independent of any solver)
@pytest.mark.parametrize('p', [1, 2, 3, 5])
def test_nesterov_todd_identities(p):
    rs = np.random.RandomState(p)
    c = cl.Cones(0, 0, [], psd=[p])
    sc = cl._Scaling(c)
    d = p * (p + 1) // 2
    for _ in range(5):
        G = rs.randn(p, p); S = G @ G.T + 0.1 * np.eye(p)
        G = rs.randn(p, p); Z = G @ G.T + 0.1 * np.eye(p)
        s, z = cl.mat_to_svec(S), cl.mat_to_svec(Z)
        assert abs(float(s @ z) - np.trace(S @ Z)) <= 1e-12 * abs(float(s @ z))          # svec is an isometry
        assert sc.update(s, z)
        lam = cl.mat_to_svec(np.diag(sc.psd_lam[0]))
        assert np.abs(sc.mul_W(z) - lam).max() <= 1e-10 * np.abs(lam).max()                           # W z = lambda
        assert np.abs(sc.mul_W(s, inv=True, trans=True) - lam).max() <= 1e-10 * np.abs(lam).max()     # W^-T s = lambda
        assert np.abs(sc.mul_Hs(z) - s).max() <= 1e-10 * np.abs(s).max()                              # W'W z = s
        v = rs.randn(d)
        assert np.abs(sc.Hs() @ v - sc.mul_Hs(v)).max() <= 1e-10 * np.abs(v).max() * np.abs(sc.Hs()).max()     # the dense block IS Q (x)s Q
        assert np.abs(sc.mul_W(sc.mul_W(v), trans=True) - sc.mul_Hs(v)).max() <= 1e-10 * np.abs(sc.mul_Hs(v)).max()
        assert np.abs(sc.mul_W(sc.mul_W(v), inv=True) - v).max() <= 1e-9 * np.abs(v).max()
        assert np.abs(sc.mul_W(sc.mul_W(v, trans=True), inv=True, trans=True) - v).max() <= 1e-9 * np.abs(v).max()
        assert np.abs(sc.circ(sc.lam, sc.inv_circ_lam(v)) - v).max() <= 1e-12 * np.abs(v).max()        # lambda o (lambda \ v) = v
        assert np.linalg.eigvalsh(sc.Hs()).min() > 0.0


# ------------------------------------------------------------------------------------ oracle against closed forms
def test_oracle_closed_forms():
    rs = np.random.RandomState(0)
    for p in (2, 3, 4, 6):
        C = _sym(rs, 3, p)
        d = families.min_eig(p)
        assert d.cones['psd'] == [p] and d.m == p * (p + 1) // 2
        o = cl.cpg_solve_batch(d, _theta(d, {'C': C}))
        assert (o['status'] == cl.SOLVED).all() and o['iter'].max() <= 12
        assert np.abs(o['obj_val'] - np.linalg.eigvalsh(C).min(axis=1)).max() <= 1e-7 * max(1.0, np.abs(C).max())
        d = families.psd_projection(p)
        o = cl.cpg_solve_batch(d, _theta(d, {'C': C}))
        assert (o['status'] == cl.SOLVED).all()
        for b in range(3):
            X = _clip(C[b])
            assert np.abs(cl.svec_to_mat(o['prim']['x'][b], p) - X).max() <= 1e-6
            assert abs(o['obj_val'][b] - (((X - C[b]) ** 2).sum() - (C[b] ** 2).sum())) <= 1e-6
    for p in (2, 3):
        C = _sym(rs, 3, p)
        d = families.trace_sdp(p)
        assert d.cones == {'zero': 1, 'nonneg': 0, 'soc': [p * (p + 1) // 2 + 1], 'psd': [p], 'exp': 1}
        o = cl.cpg_solve_batch(d, _theta(d, {'C': C}))
        assert (o['status'] == cl.SOLVED).all()
        assert np.abs(o['obj_val'] - np.linalg.eigvalsh(C).min(axis=1)).max() <= 1e-7 * max(1.0, np.abs(C).max())


def test_plan_holds_the_blocks_and_its_limits():
    d = families.min_eig(3)
    cp = build_conic_plan(d)
    assert list(cp.psd_dims) == [3] and cp.n_exp == 0
    kinds = np.asarray(cp.ksrc_kind)
    assert (kinds == 8).sum() == 15                       # C(6, 2) off-diagonal entries of the 6 x 6 block
    idx = np.asarray(cp.ksrc_idx)[kinds == 8]
    assert ((idx >> 12) & 0xF == 3).all() and (idx & 0xFFF == 0).all()
    d = families.min_eig(9)
    with pytest.raises(NotImplementedError, match='order 1 .. 8'):
        build_conic_plan(d)


# ------------------------------------------------------------------------------------ emulator tier
def test_kernel_in_emulator_vs_oracle(sim_lib):
    rs = np.random.RandomState(1)
    for p, fams in ((2, (families.min_eig, families.psd_projection, families.trace_sdp)), (3, (families.min_eig, families.psd_projection)),
                    (4, (families.trace_sdp,)), (8, (families.min_eig,))):
        C = _sym(rs, 2, p)
        for fam in fams:
            d = fam(p)
            bs = ConicBatchSolver(d, lib_path=sim_lib, full_output=True)
            th = _theta(d, {'C': C})
            _assert_parity(bs.solve({'C': C}, max_iter=2), cl.cpg_solve_batch(d, th, max_iter=2))
            o = cl.cpg_solve_batch(d, th)
            assert (o['status'] == cl.SOLVED).all()
            _assert_parity(bs.solve({'C': C}), o, tol=1e-6 if fam is families.trace_sdp else 1e-9)
            bs.close()


def _fact(bs, name):
    import ctypes as C
    v = C.c_double(-1)
    bs.lib.check(bs.lib.L.cpg_hip_get_setting(bs.h, name.encode(), C.byref(v)), 'get_setting')
    return v.value


def test_generated_family_library_in_emulator(tmp_path):
    from tests.sim import build_sim
    d = families.psd_projection(3)
    cp = build_conic_plan(d)
    lib = build_sim.build_conic_family(cp, str(tmp_path), 'psd_projection')
    assert 'CPG_CK_HPSD' in open(os.path.join(str(tmp_path), 'cpg_conic_psd_projection_factor.h')).read()
    C = _sym(np.random.RandomState(2), 3, 3)
    bs = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
    r = bs.solve({'C': C})
    assert _fact(bs, 'generated_executor') == 1.0 and _fact(bs, 'specialised_kernel') == 1.0
    bg = ConicBatchSolver(d, lib_path=build_sim.build(), plan=cp, full_output=True)
    rg = bg.solve({'C': C})
    assert np.array_equal(r.sol_x, rg.sol_x) and np.array_equal(r.sol_y, rg.sol_y) and r.iter.tolist() == rg.iter.tolist()
    _assert_parity(r, cl.cpg_solve_batch(d, _theta(d, {'C': C})))
    bs.close(); bg.close()


# ------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_kernel_on_gpu_vs_oracle():
    rs = np.random.RandomState(3)
    for p, fam, tol in ((2, families.min_eig, 1e-8), (3, families.psd_projection, 1e-8), (4, families.min_eig, 1e-8), (3, families.trace_sdp, 1e-5)):
        C = _sym(rs, 32, p)
        d = fam(p)
        bs = ConicBatchSolver(d, full_output=True)
        th = _theta(d, {'C': C})
        _assert_parity(bs.solve({'C': C}, max_iter=2), cl.cpg_solve_batch(d, th, max_iter=2), tol=1e-8)
        r, o = bs.solve({'C': C}), cl.cpg_solve_batch(d, th)
        if fam is families.trace_sdp:       # (an exponential cone in the family: its conditioning, tests/test_nonsym_cones.py)
            assert r.status.tolist() == o['status'].tolist() and np.abs(r.iter.astype(int) - o['iter'].astype(int)).max() <= 3
            assert np.abs(r.obj_val - o['obj_val']).max() <= 1e-7
        else:
            _assert_parity(r, o, tol=tol)
        bs.close()


@pytest.mark.gpu
def test_closed_forms_on_gpu_at_batch_size():
    B = 20000
    rs = np.random.RandomState(4)
    for p in (3, 6):
        C = _sym(rs, B, p)
        bs = ConicBatchSolver(families.min_eig(p))
        r = bs.solve({'C': C})
        assert (r.status == 1).all() and r.iter.max() <= 15
        assert np.abs(r.obj_val - np.linalg.eigvalsh(C).min(axis=1)).max() <= 1e-6
        bs.close()
    C = _sym(rs, B, 3)
    bs = ConicBatchSolver(families.psd_projection(3))
    r = bs.solve({'C': C})
    assert (r.status == 1).all()
    w, V = np.linalg.eigh(C)
    X = np.einsum('bij,bj,bkj->bik', V, np.maximum(w, 0.0), V)
    val = ((X - C) ** 2).sum(axis=(1, 2)) - (C ** 2).sum(axis=(1, 2))
    assert np.abs(r.obj_val - val).max() <= 1e-5
    bs.close()


@pytest.mark.gpu
def test_generated_family_library_on_gpu():
    from cvxpygen_amd import codegen
    d = families.psd_projection(3)
    cp = build_conic_plan(d)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = codegen.build_conic_library(cp, os.path.join(root, 'cvxpygen_amd', 'generated', 'psd_projection'), 'psd_projection')   # no-op when fresh
    C = _sym(np.random.RandomState(5), 2000, 3)
    bs = ConicBatchSolver(d, lib_path=lib, plan=cp, full_output=True)
    r = bs.solve({'C': C})
    assert _fact(bs, 'generated_executor') == 1.0 and _fact(bs, 'specialised_kernel') == 1.0
    bg = ConicBatchSolver(d, plan=cp, full_output=True)
    rg = bg.solve({'C': C})
    assert np.array_equal(r.sol_x, rg.sol_x) and r.iter.tolist() == rg.iter.tolist() and r.status.tolist() == rg.status.tolist()
    for b in range(0, 2000, 97):
        assert np.abs(cl.svec_to_mat(r.sol_x[b], 3) - _clip(C[b])).max() <= 1e-3       # (a rank-deficient optimum: the iterate is accurate to the square root of the gap)
    bs.close(); bg.close()
