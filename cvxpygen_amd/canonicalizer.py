"""
cvxpy front door: `cvxpy.Problem` -> `FamilyDescriptor` (reference: `cvxpygen/canonicalizer.py:86-122,
283-332`, `cvxpygen/solvers/_interface.py:39-79, 132-173`, `cvxpygen/solvers/clarabel.py:127-140`).

UNTESTED AGAINST cvxpy: cvxpy is not installed in the build container nor on the GPU box (SURVEY.md
F3).  The part that touches cvxpy objects (`descriptor_from_cvxpy`) only unpacks what
`problem.get_problem_data(solver, enforce_dpp=True)` returns; everything else lives in
`descriptor_from_reduced`, a pure numpy / scipy function over that data, which the CPU tests drive
with the same arrays rebuilt from hand-made descriptors (tests/test_host.py).

cvxpy hands over a parametrised cone / QP program as sparse maps from the parameter vector
[theta; 1] to the *stored entries* of the problem matrices:
  reduced_P.reduced_mat   rows = stored entries of P (CSC, `problem_data_index` = indices, indptr, shape)
  q                       (n_var + 1) rows: the linear cost, last row the constant d
  reduced_A.reduced_mat   rows = stored entries of [A | b] (CSC with n_var + 1 columns; the last
                          column is the constant vector), in the convention  A x + b  in K
The reference turns these into per-canonical-parameter maps with solver-specific signs and row
selections; this module does the same for the two solver forms the HIP backend implements.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from .descriptor import FamilyDescriptor, UserDual, UserParam, UserVar


def _split_constraint_entries(indices: np.ndarray, indptr: np.ndarray, n_var: int):
    """stored entries of [A | b]: those of the first n_var columns form the matrix part, the last
    column the vector part (`canonicalizer.py:270-281`)"""
    n_data = len(indices)
    n_vec = int(indptr[-1] - indptr[-2])
    n_mat = n_data - n_vec
    assert len(indptr) == n_var + 2
    return n_mat, n_vec


def descriptor_from_reduced(name: str, solver: str, n_var: int, n_eq: int, n_ineq: int,
                            red_P: Optional[sp.spmatrix], P_index: Optional[Tuple[np.ndarray, np.ndarray, tuple]],
                            q_map: sp.spmatrix, red_A: sp.spmatrix, A_index: Tuple[np.ndarray, np.ndarray, tuple],
                            theta0: np.ndarray, params: List[UserParam], variables: List[UserVar],
                            duals: List[UserDual], is_maximization: bool,
                            cones: Optional[Dict[str, object]] = None) -> FamilyDescriptor:
    """Pure-numpy core.  theta0 includes the trailing 1.  For solver 'OSQP' the rows of [A | b] are
    ordered equalities first (n_eq), then inequalities (n_ineq); the canonical form is
    l <= A x <= u (`_interface.py:39-79`).  For 'CLARABEL' every row is a cone row of A x + s = b
    (`clarabel.py:127-140`, base `get_affine_map`, `_interface.py:132-173`)."""
    NP1 = len(theta0)
    m = n_eq + n_ineq
    q_map = sp.csr_matrix(q_map)
    map_q, map_d = q_map[:n_var], q_map[n_var:n_var + 1]
    red_A = sp.csr_matrix(red_A)
    a_idx, a_ptr, _ = A_index
    a_idx, a_ptr = np.asarray(a_idx), np.asarray(a_ptr)
    n_mat, n_vec = _split_constraint_entries(a_idx, a_ptr, n_var)

    # ---- matrix part: keep every structurally possible entry (rows of the map that are not all zero)
    A_rows_map = sp.csr_matrix(red_A[:n_mat])
    A_row = a_idx[:n_mat]
    A_col = np.repeat(np.arange(n_var), np.diff(a_ptr[:n_var + 1]))
    if solver == 'OSQP':
        sign = np.where(A_row >= n_eq, -1.0, 1.0)            # `_interface.py:61`
    else:
        sign = -np.ones(n_mat)                               # `_interface.py:170-171`
    A_rows_map = sp.diags(sign) @ A_rows_map
    keep = np.diff(sp.csr_matrix(A_rows_map).indptr) > 0     # `canonicalizer.py:449-451`
    A_rows_map, A_row, A_col = sp.csr_matrix(A_rows_map)[keep], A_row[keep], A_col[keep]
    A_vals = np.asarray(A_rows_map @ theta0).ravel()
    indptr = np.zeros(n_var + 1, dtype=np.int32)
    np.add.at(indptr, A_col + 1, 1)
    A = sp.csc_matrix((A_vals, A_row.astype(np.int32), np.cumsum(indptr).astype(np.int32)), shape=(m, n_var))

    # ---- vector part, scattered to dense rows (`canonicalizer.py:425-433`)
    vec_rows = a_idx[n_mat:]
    dense = sp.lil_matrix((m, NP1))
    vec_map = sp.csr_matrix(red_A[n_mat:])
    for k, r in enumerate(vec_rows):
        dense[r, :] = vec_map[k]
    dense = sp.csr_matrix(dense)
    maps: Dict[str, sp.csr_matrix] = {}
    if solver == 'OSQP':
        sgn_u = np.concatenate([-np.ones(n_eq), np.ones(n_ineq)])       # `_interface.py:70-73`
        maps['l'] = sp.csr_matrix(-dense[:n_eq])                        # `_interface.py:62-67`
        maps['u'] = sp.csr_matrix(sp.diags(sgn_u) @ dense)
    else:
        maps['b'] = dense

    # ---- objective
    if red_P is not None and P_index is not None:
        p_idx, p_ptr, _ = P_index
        P_map = sp.csr_matrix(red_P)
        keep = np.diff(P_map.indptr) > 0
        P_row = np.asarray(p_idx)[keep]
        P_col = np.repeat(np.arange(n_var), np.diff(np.asarray(p_ptr)))[keep]
        P_map = P_map[keep]
        pptr = np.zeros(n_var + 1, dtype=np.int32)
        np.add.at(pptr, P_col + 1, 1)
        P = sp.csc_matrix((np.asarray(P_map @ theta0).ravel(), P_row.astype(np.int32),
                           np.cumsum(pptr).astype(np.int32)), shape=(n_var, n_var))
        if (P_row > P_col).any():
            raise NotImplementedError('expected the upper triangle of P')
    else:                                                     # LP: `_interface.py:136-138`
        P_map = sp.csr_matrix((0, NP1))
        P = sp.csc_matrix((n_var, n_var))
    maps.update({'P': sp.csr_matrix(P_map), 'q': sp.csr_matrix(map_q), 'd': sp.csr_matrix(map_d),
                 'A': sp.csr_matrix(A_rows_map)})
    changes = {pid: bool(sp.csc_matrix(M)[:, :NP1 - 1].nnz > 0) for pid, M in maps.items()}
    return FamilyDescriptor(name=name, n_var=n_var, n_eq=n_eq, n_ineq=n_ineq, P=P, A=A, maps=maps,
                            changes=changes, theta0=np.asarray(theta0, dtype=np.float64), params=params,
                            variables=variables, duals=duals, is_maximization=is_maximization,
                            nonzero_d=bool(sp.csr_matrix(map_d).nnz > 0), solver=solver, cones=cones)


def descriptor_from_cvxpy(problem, solver: str = 'OSQP', solver_opts=None, name: str = 'problem') -> FamilyDescriptor:
    """`Canonicalizer._extract` (`canonicalizer.py:86-122`) for the OSQP, CLARABEL and ECOS forms.  `solver_opts` has the
    reference's meaning: cvxpy canonicalisation options, forwarded verbatim to `get_problem_data` (`canonicalizer.py:89-95`);
    its `use_quad_obj` entry also decides whether a conic solver keeps a quadratic objective (`canonicalizer.py:418-426`).
    ECOS (`solvers/ecos.py:20-22, 75-84`): the chain of cvxpy's ECOS interface hands over the same [A | b] data as any conic
    solver; the rows of the zero cone become A x = b, the others G x + s = h (`_interface.py:132-173`, `ecos_front.py`)."""
    import warnings
    import cvxpy as cp
    from cvxpy.reductions.solvers.conic_solvers.conic_solver import ConicSolver
    warnings.warn('cvxpygen_amd: the cvxpy front door has not been exercised against a cvxpy installation '
                  '(none in the build image); what it unpacks from get_problem_data is checked only through '
                  'hand-canonicalised descriptors and the reference\'s own post-processing (tests/test_ref_fixtures.py). '
                  'Compare the first solve with prob.solve(solver=...) before relying on it.', RuntimeWarning, stacklevel=2)
    from cvxpy.reductions import InverseData
    try:
        from cvxpy.reductions.solvers.solving_chain import SolverInverseData
    except ImportError:                                     # older cvxpy
        SolverInverseData = ()
    if solver not in ('OSQP', 'CLARABEL', 'ECOS'):
        raise ValueError(f'Unsupported solver: {solver}.')
    data, _, inverse_data = problem.get_problem_data(solver=solver, gp=False, enforce_dpp=True, verbose=False,
                                                     solver_opts=solver_opts)
    pp = data['param_prob']
    if not pp.parameters:
        raise ValueError('Solution does not depend on parameters. Aborting code generation.')
    conic = solver in ('CLARABEL', 'ECOS')
    cones = None
    if conic:
        cd = pp.cone_dims
        if solver == 'ECOS' and cd.exp > 0:                  # `solvers/ecos.py:121-125`
            raise ValueError('Code generation with ECOS and exponential cones is not supported yet.')
        p3d = [float(a) for a in getattr(cd, 'p3d', [])]
        if (cd.exp or p3d or len(cd.psd)) and solver == 'ECOS':
            raise NotImplementedError('exponential / power / PSD cones: CLARABEL only')
        n_var, n_eq, n_ineq = int(pp.x.size), int(cd.zero), int(data['A'].shape[0]) - int(cd.zero)
        # rows as cvxpy stacks them for this solver: zero | nonneg | soc | (psd) | exp | p3d
        cones = {'zero': int(cd.zero), 'nonneg': int(cd.nonneg), 'soc': [int(v) for v in cd.soc]}
        if len(cd.psd):
            cones['psd'] = [int(v) for v in cd.psd]
        if cd.exp:
            cones['exp'] = int(cd.exp)
        if p3d:
            cones['pow'] = p3d
    else:
        n_var, n_eq, n_ineq = int(data['n_var']), int(data['n_eq']), int(data['n_ineq'])

    # ---- user parameters: column layout of cvxpy's parameter vector (`canonicalizer.py:226-271`)
    NP = int(pp.total_param_size)
    theta0 = np.zeros(NP + 1)
    theta0[-1] = 1.0
    params: List[UserParam] = []
    for p in pp.parameters:
        if p.value is None:
            p.project_and_assign(np.random.randn(*p.shape))
        col, size = int(pp.param_id_to_col[p.id]), int(pp.param_id_to_size[p.id])
        if p.attributes.get('diag'):
            kind, sparsity = 'diag', None
            val = np.asarray(p.value.toarray() if hasattr(p.value, 'toarray') else p.value)
            flat = np.diag(val)
        elif getattr(p, '_has_dim_reducing_attr', False) and p.attributes.get('sparsity') is not None:
            kind = 'sparse'
            sparsity = tuple(tuple(int(v) for v in s) for s in p.attributes['sparsity'])
            flat = np.asarray(p.value_sparse.data)
        elif p.size == 1:
            kind, sparsity, flat = 'scalar', None, np.asarray(p.value, dtype=float).reshape(1)
        else:
            kind, sparsity, flat = 'dense', None, np.asarray(p.value, dtype=float).flatten(order='F')
        theta0[col:col + size] = flat
        params.append(UserParam(p.name(), col, size, tuple(p.shape), kind, sparsity))
    params.sort(key=lambda u: u.col)

    # ---- primal variables (`canonicalizer.py:124-158`)
    offsets = inverse_data[-2].var_offsets
    variables: List[UserVar] = []
    for v in problem.variables():
        off = int(offsets[v.id])
        sym = bool(v.attributes['symmetric'] or v.attributes['PSD'] or v.attributes['NSD'])
        if sym:
            from cvxpy.atoms.affine.upper_tri import upper_tri_to_full
            (_, colx) = upper_tri_to_full(v.shape[0]).nonzero()
            idx = off + np.asarray(colx)
        else:
            idx = np.arange(off, off + int(np.prod(v.shape)) if v.shape else off + 1)
        variables.append(UserVar(v.name(), idx.astype(np.int32), tuple(v.shape), sym))

    # ---- dual variables: one per user constraint (`canonicalizer.py:160-224`)
    id_maps = []
    for inv in inverse_data:
        if isinstance(inv, InverseData) and not (SolverInverseData and isinstance(inv, SolverInverseData)):
            id_maps.append(inv.cons_id_map)
        if isinstance(inv, tuple) and len(inv) == 3:
            id_maps.append(inv[2])
    dual_ids = []
    for did in id_maps[0].keys():
        for mp in id_maps[1:]:
            did = mp[did]
        dual_ids.append(did)
    if conic:
        con_canon = inverse_data[-1][ConicSolver.EQ_CONSTR] + inverse_data[-1][ConicSolver.NEQ_CONSTR]
    else:
        con_canon = inverse_data[-2].constraints
    offs = np.cumsum([0] + [c.size for c in con_canon[:-1]])
    off_of = {c.id: int(o) for c, o in zip(con_canon, offs)}
    con_of = {c.id: c for c in con_canon}
    duals: List[UserDual] = []
    for k, did in enumerate(dual_ids):
        c = con_of[did]
        shape = tuple(c.shape) if int(np.prod(c.shape)) == c.size else (c.size,)
        duals.append(UserDual(f'd{k}', (off_of[did] + np.arange(c.size)).astype(np.int32), shape, 'z' if conic else 'y'))

    P_index = pp.reduced_P.problem_data_index
    # `canonicalizer.py:418-426`: quadratic solvers always; conic ones when they support it (Clarabel does, ECOS does
    # not: `supports_quad_obj`), the user did not switch it off and the objective has a quadratic term
    use_quad_obj = solver_opts.get('use_quad_obj', True) if solver_opts else True
    quad = P_index is not None and (not conic or (bool(use_quad_obj) and solver == 'CLARABEL'
                                                  and problem.objective.expr.has_quadratic_term()))
    desc = descriptor_from_reduced(
        name, 'CLARABEL' if conic else solver, n_var, n_eq, n_ineq,
        pp.reduced_P.reduced_mat if quad else None, P_index if quad else None,
        pp.q, pp.reduced_A.reduced_mat, pp.reduced_A.problem_data_index, theta0, params, variables, duals,
        isinstance(problem.objective, cp.Maximize), cones)
    if solver == 'ECOS':
        from .ecos_front import ecos_from_conic
        desc = ecos_from_conic(desc)
    return desc


def reduced_from_descriptor(desc: FamilyDescriptor):
    """Inverse of `descriptor_from_reduced` (test helper): the arrays cvxpy would hand over for this
    family -- reduced_P, P index data, q map, reduced_A ([A | b] entries, cvxpy sign convention),
    A index data."""
    n, m, n_eq = desc.n_var, desc.m, desc.n_eq
    NP1 = desc.NP + 1
    conic = desc.solver != 'OSQP'
    A = sp.csc_matrix(desc.A)
    A_row = A.indices
    if conic:
        sign = -np.ones(A.nnz)
        vec = sp.csr_matrix(desc.maps['b'])
    else:
        sign = np.where(A_row >= n_eq, -1.0, 1.0)
        sgn_u = np.concatenate([-np.ones(n_eq), np.ones(desc.n_ineq)])
        vec = sp.csr_matrix(sp.diags(sgn_u) @ sp.csr_matrix(desc.maps['u']))
    mat_rows = sp.diags(sign) @ sp.csr_matrix(desc.maps['A'])
    red_A = sp.vstack([mat_rows, vec]).tocsr()
    a_idx = np.concatenate([A_row, np.arange(m)])
    a_ptr = np.concatenate([A.indptr, [A.nnz + m]])
    q_map = sp.vstack([sp.csr_matrix(desc.maps['q']), sp.csr_matrix(desc.maps['d'])]).tocsr()
    P = sp.csc_matrix(desc.P)
    if P.nnz:
        red_P, P_index = sp.csr_matrix(desc.maps['P']), (P.indices, P.indptr, P.shape)
    else:
        red_P, P_index = None, None
    return red_P, P_index, q_map, red_A, (a_idx, a_ptr, (m, n + 1))
