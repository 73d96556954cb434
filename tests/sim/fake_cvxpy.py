"""Test double for the cvxpy objects `cvxpygen_amd.canonicalizer.descriptor_from_cvxpy` reads (no cvxpy in the build
image): a `Problem` built from a FamilyDescriptor whose `get_problem_data` hands back what cvxpy's solving chain would
for that family -- `param_prob` (reduced_P / q / reduced_A with their CSC index data, parameter columns, cone
dimensions), the inverse data of the chain (variable offsets, canonical constraints, constraint-id maps) -- and RECORDS
the arguments it was called with.  It checks the plumbing of the front door (which solver name and `solver_opts` reach
`get_problem_data`, how the pieces are unpacked for OSQP / CLARABEL / ECOS), not cvxpy's canonicalisation itself.

`install()` puts the stand-in modules into sys.modules and returns a function that removes them again."""
import sys
import types

import numpy as np
import scipy.sparse as sp


class InverseData:
    def __init__(self, cons_id_map, var_offsets, constraints):
        self.cons_id_map, self.var_offsets, self.constraints = cons_id_map, var_offsets, constraints


class ConicSolver:
    EQ_CONSTR, NEQ_CONSTR = 'eq_constr', 'other_constr'


class _Objective:
    def __init__(self, quad):
        self.expr = types.SimpleNamespace(has_quadratic_term=lambda: quad)


class Minimize(_Objective):
    pass


class Maximize(_Objective):
    pass


class _Con:
    def __init__(self, cid, size, shape):
        self.id, self.size, self.shape = cid, size, shape


class _Param:
    def __init__(self, up, theta0, pid):
        self.id, self._name, self.shape, self.size = pid, up.name, tuple(up.shape), int(up.size)
        self.attributes = {'diag': up.kind == 'diag', 'sparsity': None}
        flat = theta0[up.col:up.col + up.size]
        if up.kind == 'diag':
            self.value = np.diag(flat)
            self.size = int(np.prod(up.shape))
        elif up.kind == 'scalar':
            self.value = float(flat[0])
        elif up.kind == 'sparse':
            self._has_dim_reducing_attr = True
            self.attributes['sparsity'] = up.sparsity
            self.value_sparse = types.SimpleNamespace(data=np.array(flat))
            self.value = None
            self.size = int(np.prod(up.shape))
        else:
            self.value = flat.reshape(up.shape, order='F')

    def name(self):
        return self._name


class _Var:
    def __init__(self, uv, vid):
        self.id, self._name, self.shape = vid, uv.name, tuple(uv.shape)
        self.attributes = {'symmetric': False, 'PSD': False, 'NSD': False}

    def name(self):
        return self._name


class _Reduced:
    def __init__(self, mat, index):
        self.reduced_mat, self.problem_data_index = mat, index


class Problem:
    """the cvxpy.Problem of a family given as FamilyDescriptor (OSQP or CLARABEL form)"""

    def __init__(self, desc):
        from cvxpygen_amd.canonicalizer import reduced_from_descriptor
        self._family = desc
        self.calls = []
        quad = sp.csc_matrix(desc.P).nnz > 0
        self.objective = (Maximize if desc.is_maximization else Minimize)(quad)
        self._vars = [_Var(v, 100 + k) for k, v in enumerate(desc.variables)]
        self._params = [_Param(p, desc.theta0, 200 + k) for k, p in enumerate(desc.params)]
        self._reduced = reduced_from_descriptor(desc)
        self.solve_methods = {}

    def variables(self):
        return self._vars

    def register_solve(self, name, fn):
        self.solve_methods[name] = fn

    def get_problem_data(self, solver, gp=False, enforce_dpp=False, verbose=False, solver_opts=None):
        d = self._family
        self.calls.append(dict(solver=solver, gp=gp, enforce_dpp=enforce_dpp, verbose=verbose, solver_opts=solver_opts))
        conic = solver in ('CLARABEL', 'ECOS')
        assert conic == (d.solver != 'OSQP'), 'the double holds ONE canonical form of the family'
        red_P, P_index, q_map, red_A, A_index = self._reduced
        assert not (solver == 'ECOS' and red_P is not None), "cvxpy's ECOS chain never hands over a quadratic objective"
        pp = types.SimpleNamespace(
            parameters=self._params, total_param_size=d.NP,
            param_id_to_col={p.id: up.col for p, up in zip(self._params, d.params)},
            param_id_to_size={p.id: up.size for p, up in zip(self._params, d.params)},
            reduced_P=_Reduced(red_P, P_index), q=q_map, reduced_A=_Reduced(red_A, A_index),
            x=types.SimpleNamespace(size=d.n_var))
        data = {'param_prob': pp}
        # canonical constraints in row order: one per user dual, fillers for the rows in between
        rows = sorted((int(u.indices[0]), int(np.size(u.indices)), k) for k, u in enumerate(d.duals))
        cons, at, cid_of = [], 0, {}
        for lo, sz, k in rows:
            u = d.duals[k]
            assert np.array_equal(np.ravel(u.indices), lo + np.arange(sz)), 'user duals must cover contiguous rows'
            if lo > at:
                cons.append(_Con(900 + len(cons), lo - at, (lo - at,)))
            cons.append(_Con(500 + k, sz, tuple(u.shape)))
            cid_of[k] = 500 + k
            at = lo + sz
        if at < d.m:
            cons.append(_Con(900 + len(cons), d.m - at, (d.m - at,)))
        user_map = {300 + k: cid_of[k] for k in range(len(d.duals))}          # user constraint id -> canonical id
        var_offsets = {v.id: int(uv.indices[0]) for v, uv in zip(self._vars, d.variables)}
        # two reductions, as in a real chain: the first map's KEYS are the user's constraint ids, the later maps carry
        # them to the canonical ones (cvxpygen/canonicalizer.py:168-172)
        inv0 = InverseData({k: k for k in user_map}, {}, [])
        inv = InverseData(user_map, var_offsets, cons)
        if conic:
            c = d.cones
            pp.cone_dims = types.SimpleNamespace(zero=c['zero'], nonneg=c['nonneg'], soc=list(c['soc']), exp=0, psd=[], p3d=[])
            n_zero, eq, neq, at = c['zero'], [], [], 0
            for cn in cons:
                (eq if at < n_zero else neq).append(cn)
                at += cn.size
            data['A'] = sp.csc_matrix((d.m, d.n_var))
            data['G'] = sp.csc_matrix((d.m - n_zero, d.n_var))
            return data, None, [inv0, inv, {ConicSolver.EQ_CONSTR: eq, ConicSolver.NEQ_CONSTR: neq}]
        data.update(n_var=d.n_var, n_eq=d.n_eq, n_ineq=d.n_ineq)
        return data, None, [inv0, inv, {}]


def install():
    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        mods[name] = m
        return m
    mod('cvxpy', Maximize=Maximize, Minimize=Minimize, Problem=Problem)
    mod('cvxpy.reductions', InverseData=InverseData)
    mod('cvxpy.reductions.solvers')
    mod('cvxpy.reductions.solvers.conic_solvers')
    mod('cvxpy.reductions.solvers.conic_solvers.conic_solver', ConicSolver=ConicSolver)
    mod('cvxpy.reductions.solvers.solving_chain')                         # (no SolverInverseData: the "older cvxpy" branch)
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)

    def remove():
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return remove
