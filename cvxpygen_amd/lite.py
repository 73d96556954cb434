"""
A minimal stand-in for the handful of `cvxpy.Problem` attributes that the reference's generated
shim touches (`cvxpygen/templates/cpg_solver.py.jinja2:40-117`): `param_dict[...]` with `.value`,
`.size`, `.attributes`, `var_dict[...]` with `.save_value`, `constraints[i].save_dual_value`,
`_clear_solution`, `_status`, `_value`, `_solution`, `_solver_stats`, `register_solve` and
`solve(method=...)`.

cvxpy is not installed in the build container nor on the GPU box; `LiteProblem.from_descriptor`
lets the B = 1 drop-in path (`prob.register_solve('CPG', cpg_solve); prob.solve(method='CPG')`) be
exercised and tested without it.  With cvxpy present, a real `cvxpy.Problem` goes through exactly
the same `cpg_solve`.
"""

from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, Dict, List

import numpy as np
import scipy.sparse as sp

from .descriptor import FamilyDescriptor


class LiteParameter:
    def __init__(self, up, value):
        self.name_ = up.name
        self.shape = tuple(up.shape)
        self.size = int(np.prod(up.shape)) if up.shape else 1
        self.attributes = {'diag': up.kind == 'diag', 'sparsity': up.sparsity}
        self._has_dim_reducing_attr = up.kind == 'sparse'
        self._up = up
        self.id = id(self)
        self.gradient = None
        self.value = value

    def name(self):
        return self.name_

    @property
    def value_sparse(self):
        r, c = self._up.sparsity
        v = np.asarray(self.value)
        data = v if v.ndim == 1 else v[np.asarray(r), np.asarray(c)]
        return sp.coo_array((data, (np.asarray(r), np.asarray(c))), shape=self.shape)


class LiteVariable:
    def __init__(self, uv):
        self.name_ = uv.name
        self.shape = tuple(uv.shape)
        self.id = id(self)
        self.value = None
        self.gradient = None

    def name(self):
        return self.name_

    def save_value(self, v):
        self.value = v


class LiteConstraint:
    def __init__(self, ud):
        self.id = id(self)
        self.shape = tuple(ud.shape)
        self.dual_value = None

    def save_dual_value(self, v):
        self.dual_value = v


class LiteProblem:
    def __init__(self, desc: FamilyDescriptor):
        self.desc = desc
        self.param_dict: Dict[str, LiteParameter] = {}
        for up in desc.params:
            flat = desc.theta0[up.col:up.col + up.size]
            if up.kind == 'scalar':
                val = float(flat[0])
            elif up.kind == 'diag':
                val = np.diag(flat)
            elif up.kind == 'sparse':
                val = flat.copy()
            else:
                val = flat.reshape(up.shape, order='F')
            self.param_dict[up.name] = LiteParameter(up, val)
        self.var_dict = {v.name: LiteVariable(v) for v in desc.variables}
        self.constraints: List[LiteConstraint] = [LiteConstraint(d) for d in desc.duals]
        self._solve_methods: Dict[str, Callable] = {}
        self._status = None
        self._value = None
        self._solution = None
        self._solver_stats = None

    @staticmethod
    def from_descriptor(desc: FamilyDescriptor) -> 'LiteProblem':
        return LiteProblem(desc)

    # -- the cvxpy surface used by the shim ----------------------------------------------------
    def parameters(self):
        return list(self.param_dict.values())

    def variables(self):
        return list(self.var_dict.values())

    def _clear_solution(self):
        for v in self.var_dict.values():
            v.value = None
        for c in self.constraints:
            c.dual_value = None
        self._status = self._value = self._solution = None

    @property
    def status(self):
        return self._status

    @property
    def value(self):
        return self._value

    @property
    def solution(self):
        return self._solution

    @property
    def solver_stats(self):
        return self._solver_stats

    def register_solve(self, name: str, func: Callable) -> None:
        self._solve_methods[name] = func

    def solve(self, method: str = None, **kwargs):
        if method is None or method not in self._solve_methods:
            raise ValueError('LiteProblem can only be solved through a registered method '
                             "(e.g. method='CPG')")
        return self._solve_methods[method](self, **kwargs)


def make_solution(status, value, primal_vars, dual_vars, attr):
    """cvxpy.reductions.Solution when cvxpy is importable, else an attribute bag with the same
    field names (`opt_val` is what tests/test_E2E_QP.py:223 reads)."""
    try:
        from cvxpy.reductions import Solution  # type: ignore
        return Solution(status, value, primal_vars, dual_vars, attr)
    except Exception:
        return SimpleNamespace(status=status, opt_val=value, primal_vars=primal_vars,
                               dual_vars=dual_vars, attr=attr)


def make_solver_stats(results_dict, solver_name):
    try:
        from cvxpy.problems.problem import SolverStats  # type: ignore
        return SolverStats.from_dict(results_dict, solver_name)
    except Exception:
        return SimpleNamespace(solver_name=solver_name, solve_time=results_dict.get('solve_time'),
                               num_iters=results_dict.get('num_iters'),
                               extra_stats=results_dict.get('solver_specific_stats'))
