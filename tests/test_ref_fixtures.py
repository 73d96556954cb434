"""Canonicalisation conventions pinned against the REFERENCE's own code: the fixtures under
tests/golden/ref_*.{npz,json} were produced by tests/golden/make_ref_fixtures.py from
cvxpygen/solvers/_interface.py (QPCanonMixin / SolverInterface.get_affine_map), the cvxpy-free
helpers of cvxpygen/canonicalizer.py (:425-486), cvxpygen/utils.py (replace_inf, and C code written
by write_canonicalize / write_mat_def / write_vec_def, compiled and run).  Here the repository's
front-door core (`canonicalizer.descriptor_from_reduced`), its data model (`FamilyDescriptor.maps`,
`canon_at`, `user_p_name_to_canon_outdated`, `canon_builder.canon_lu`) and the hand-canonicalised
families are checked against them."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from cvxpygen_amd import families
from cvxpygen_amd.canon_builder import canon_lu
from cvxpygen_amd.canonicalizer import descriptor_from_reduced, reduced_from_descriptor

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FAMS = {'nonneg_ls': lambda: families.nonneg_ls(), 'mpc_6_3_10': lambda: families.mpc(6, 3, 10),
        'toy_box': lambda: families.toy_box(), 'actuator': lambda: families.actuator(),
        'portfolio_8_3': lambda: families.portfolio(8, 3), 'adp_conic': lambda: families.adp(),
        'toy_lp': lambda: families.toy_lp(), 'nonneg_ls_conic': lambda: families.nonneg_ls(solver='CLARABEL')}


@pytest.fixture(scope='module')
def fx():
    z = np.load(os.path.join(GOLD, 'ref_canon_fixtures.npz'))
    meta = json.loads(bytes(z['meta']).decode())
    return z, meta


def _csr(z, key):
    return sp.csr_matrix((z[key + '/map_data'], z[key + '/map_indices'], z[key + '/map_indptr']),
                         shape=tuple(z[key + '/map_shape']))


def _same_sparse(A, B):
    A, B = sp.csr_matrix(A), sp.csr_matrix(B)
    assert A.shape == B.shape
    D = (A - B)
    assert D.nnz == 0 or np.abs(D.data).max() == 0.0


@pytest.mark.parametrize('tag', list(FAMS))
def test_maps_changes_and_defaults_match_the_reference(fx, tag):
    z, meta = fx
    desc0 = FAMS[tag]()
    m = meta[tag]
    # the front-door core on the arrays cvxpy would hand over, and the hand-built family itself
    red_P, P_index, q_map, red_A, A_index = reduced_from_descriptor(desc0)
    desc1 = descriptor_from_reduced(desc0.name, desc0.solver, desc0.n_var, desc0.n_eq, desc0.n_ineq, red_P, P_index,
                                    q_map, red_A, A_index, desc0.theta0, desc0.params, desc0.variables, desc0.duals,
                                    desc0.is_maximization, desc0.cones)
    for desc in (desc1, desc0):
        for p_id in m['ids']:
            if p_id in m['none']:                               # LP: the reference has no P map (_interface.py:136-138)
                assert desc.P.nnz == 0 and desc.maps['P'].shape[0] == 0
                continue
            key = f'{tag}/{p_id}'
            _same_sparse(desc.maps[p_id], _csr(z, key))         # rows, signs, dense scatter of l / u / b
            assert bool(desc.changes[p_id]) == m['changes'][p_id]
            assert desc.maps[p_id].shape[0] == m['size'][p_id]
            if p_id in ('P', 'A'):                              # default values as CSC (canonicalizer.py:448-480)
                M = sp.csc_matrix(getattr(desc, p_id))
                M.sort_indices()
                assert np.array_equal(M.indptr, z[key + '/default_indptr'])
                assert np.array_equal(M.indices, z[key + '/default_indices'])
                assert np.array_equal(M.data, z[key + '/default_data'])
        assert bool(desc.nonzero_d) == bool(m['nonzero_d'])
        c0 = desc.default_canon()
        if desc.solver == 'OSQP':
            l, u = canon_lu(desc, c0)                           # -inf padding + replace_inf (_interface.py:76-79)
            assert np.array_equal(l, z[f'{tag}/l/default']) and np.array_equal(u, z[f'{tag}/u/default'])
        else:
            assert np.array_equal(c0['b'], z[f'{tag}/b/default'])
        assert np.array_equal(c0['q'], z[f'{tag}/q/default'])
        assert np.array_equal(np.atleast_1d(c0['d']), z[f'{tag}/d/default'])
        # adjacency user parameter -> outdated canonical parameters (canonicalizer.py:436-446)
        adj = z[f'{tag}/adjacency']
        dep = desc.user_p_name_to_canon_outdated()
        for j, p in enumerate(desc.params):
            ref = sorted(pid for i, pid in enumerate(m['ids']) if adj[i, j])
            assert sorted(dep[p.name]) == ref, (p.name, dep[p.name], ref)


@pytest.mark.parametrize('tag', list(FAMS))
def test_canonical_vectors_match_the_emitted_c(tag):
    """p = C_p [theta; 1] as computed by C code the reference's emitters wrote (write_canonicalize over
    write_mat_def / write_vec_def data with %.20f literals) against FamilyDescriptor.canon_at"""
    em = json.load(open(os.path.join(GOLD, 'ref_emitted_c.json')))[tag]
    desc = FAMS[tag]()
    theta = np.asarray(em['theta'])
    c = desc.canon_at(theta)
    for p_id in ('q', 'd', 'l', 'u', 'b'):
        if p_id in em:
            ref = np.asarray(em[p_id])
            got = np.atleast_1d(c[p_id])
            assert got.shape == ref.shape
            if ref.size == 0:                                   # e.g. no equality rows
                continue
            assert np.abs(got - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max()), p_id


def test_replace_inf_convention():
    r = json.load(open(os.path.join(GOLD, 'ref_replace_inf.json')))
    from cvxpygen_amd.descriptor import CPG_INF
    v = np.array([float(s) for s in r['in']])
    assert np.array_equal(np.clip(v, -CPG_INF, CPG_INF), np.asarray(r['out']))
