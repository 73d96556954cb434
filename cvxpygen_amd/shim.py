"""
Runtime behind the generated `cpg_solver.py`: the reference's Python shim
(`cvxpygen/templates/cpg_solver.py.jinja2`) re-implemented on top of the batched HIP backend.

  cpg_solve(prob, updated_params=None, **kwargs) -> float     same name / signature / side effects
  cpg_solve_batch(params, updated_params=None, **kwargs)      batched sibling
"""

from __future__ import annotations

import json
import os
import time
from typing import Dict, Optional, Sequence

import numpy as np

from .descriptor import FamilyDescriptor
from .lite import make_solution, make_solver_stats
from .runtime import STATUS_STRINGS, BatchResult, BatchSolver


def squeeze_scalar(val):
    if isinstance(val, np.ndarray):
        val = val.squeeze()
        if val.shape == ():
            return val.item()
    return val


def get_param_value(param):
    """`get_param_value` of the reference shim (templates/cpg_solver.py.jinja2:26-34)."""
    if param.size == 1:
        return squeeze_scalar(np.asarray(param.value))
    elif param.attributes["diag"]:
        v = np.asarray(param.value.toarray() if hasattr(param.value, 'toarray') else param.value)
        return list(np.diag(v))
    elif getattr(param, '_has_dim_reducing_attr', False):
        return list(param.value_sparse.data)
    else:
        return list(np.asarray(param.value).flatten(order="F"))


class GeneratedSolver:
    """One generated solver (= one code_dir).  Device resources are created on first use."""

    def __init__(self, code_dir: str, device: int = 0, lib_path: Optional[str] = None,
                 gradient: Optional[bool] = None):
        self.code_dir = code_dir
        self.desc = FamilyDescriptor.load(os.path.join(code_dir, 'descriptor.npz'))
        if gradient is None:
            gradient = os.path.exists(os.path.join(code_dir, 'GRADIENT'))
        self.gradient = bool(gradient)
        # gradient_two_stage (cvxpygen/generator.py:76-80): conic solve, OSQP-form adjoint (cvxpygen_amd/two_stage.py)
        self.two_stage = os.path.exists(os.path.join(code_dir, 'TWO_STAGE'))
        self.device = device
        if lib_path is None:
            # the library generate_code compiled for this family, else the generic table-driven one
            tag = ''.join(ch if ch.isalnum() else '_' for ch in self.desc.name)
            cands = [os.path.join(code_dir, f'libcpg_{tag}.so'), os.path.join(code_dir, f'libcpg_{tag}_streamed.so')]
            lib_path = None if self.two_stage else next((c for c in cands if os.path.exists(c)), None)
        self.lib_path = lib_path
        self._bs: Optional[BatchSolver] = None
        self._ws = None
        bo = os.path.join(code_dir, 'osqp_build.json')
        self.build_options = json.load(open(bo)) if os.path.exists(bo) else {}
        es = os.path.join(code_dir, 'enabled_settings.json')
        self.enabled_settings = json.load(open(es)) if os.path.exists(es) else []

    @property
    def batch_solver(self) -> BatchSolver:
        if self._bs is None:
            if self.two_stage:
                from .two_stage import TwoStageBatchSolver
                self._bs = TwoStageBatchSolver(self.desc, device=self.device, lib_path=self.lib_path)
            elif self.desc.solver == 'ECOS':
                from .ecos_front import EcosBatchSolver
                self._bs = EcosBatchSolver(self.desc, device=self.device, lib_path=self.lib_path)
            elif self.desc.solver == 'CLARABEL':
                from .conic_runtime import ConicBatchSolver
                self._bs = ConicBatchSolver(self.desc, device=self.device, lib_path=self.lib_path)
            else:
                self._bs = BatchSolver(self.desc, device=self.device, lib_path=self.lib_path,
                                       full_output=self.gradient, build_options=self.build_options)
        return self._bs

    # ---- batched entry point --------------------------------------------------------------------
    def cpg_solve_batch(self, params: Dict[str, np.ndarray],
                        updated_params: Optional[Sequence[str]] = None, **kwargs) -> BatchResult:
        if updated_params is None:
            updated_params = [p for p in self.desc.param_names if p in params] or None
        return self.batch_solver.solve(params, updated_params=updated_params, **kwargs)

    # ---- the reference's single-instance entry point -------------------------------------------------
    def _workspace(self):
        """Host mirror of the state the reference keeps in C static globals between cpg_solve() calls
        (cvxpygen/utils.py:470-689): cpg_params_vec, the Canon_Outdated flags (all raised at start,
        utils.py:559-562), and -- inside the OSQP workspace -- the unscaled q, the scaling of the last
        osqp_update_data_mat, the iterates (warm_starting = 1, solvers/osqp.py:110) and rho."""
        if self._ws is None:
            d = self.desc
            ids = [pid for pid in d.maps if d.changes.get(pid, False)]
            self._ws = dict(theta=np.array(d.theta0, dtype=np.float64), outdated=set(ids),
                            q_ws=np.array(d.default_canon().get('q', np.zeros(0)), dtype=np.float64), q_setup=None,
                            mat_touched=False, state=None)
        return self._ws

    def _filter_settings(self, kwargs):
        """settings the reference only offers through `enable_settings` (`solvers/osqp.py:111-114`): accepted like its
        `cpg_set_solver_<name>` once enabled and, like there, without effect on the iterates.  The generated solver is
        EMBEDDED OSQP: `osqp_solve` polishes only outside embedded mode (polish.c allocates a reduced KKT system and
        is not part of the emitted sources), printing is compiled out, and `delta` / `polish_refine_iter` are read by
        the polish step alone -- so `polishing=1` changes a field of the settings struct and nothing else."""
        out = dict(kwargs)
        if self.desc.solver == 'OSQP' and not self.two_stage:
            for name in ('verbose', 'polishing', 'polish_refine_iter', 'delta'):
                if name in out:
                    if name not in self.enabled_settings:
                        raise AttributeError(f'Solver setting "{name}" not available.')
                    out.pop(name)
        return out

    def reset_workspace(self):
        """back to the code-generation-time workspace (a fresh process of the reference)"""
        self._ws = None

    def cpg_solve(self, prob, updated_params=None, **kwargs):
        desc = self.desc
        if updated_params is None:
            updated_params = list(desc.param_names)
        for p in updated_params:
            if p not in desc.param_names:
                raise AttributeError(f"{p} is not a parameter.")
        kwargs = self._filter_settings(kwargs)
        ws = self._workspace()
        dep = desc.user_p_name_to_canon_outdated()
        param_dict = prob.param_dict
        # cpg_update_<param>: only the listed parameters are read; every other one keeps the value of its
        # last update (templates/cpg_solver.py.jinja2:44-68, utils.py:904-935)
        for name in updated_params:
            up = desc.param(name)
            v = np.asarray(get_param_value(param_dict[name]), dtype=np.float64).reshape(-1)
            ws['theta'][up.col:up.col + up.size] = v
            ws['outdated'].update(pid for pid in dep[name] if desc.changes.get(pid, False))
        theta_var = np.ascontiguousarray(ws['theta'][:desc.NP][None, :])
        bs = self.batch_solver
        t0 = time.time()
        conic = desc.solver in ('CLARABEL', 'ECOS') or self.two_stage
        if conic:
            # new solver per solve in the reference (solvers/clarabel.py:201-204): nothing but theta carries over
            res = bs.solve(updated_params=None, theta_var=theta_var, B=1, **kwargs)
        else:
            if ws['outdated'] & {'P', 'A'}:
                # osqp_update_data_mat runs before osqp_update_data_vec (solvers/osqp.py:20-59): the
                # re-equilibration sees the q the workspace held so far
                ws['q_setup'] = ws['q_ws'].copy()
                ws['mat_touched'] = True
            # matrices never updated: the workspace's P, A are the family's -- shared factor, and with rho
            # adaptation (the default) the per-instance factor kernel behind it (hybrid execution)
            path = 'refactor' if ws['mat_touched'] else 'shared'
            bs.set_updated(None, q_setup=ws['q_setup'], path=path)
            # the workspace always goes in: warm_starting = 0 (osqp_cold_start) zeroes the iterates only, the rho
            # its last adapt_rho left -- and the factor that goes with it -- stay (the kernels read the setting)
            res = bs.solve(theta_var=theta_var, B=1, state_in=ws['state'], return_state=True, ctype_in=ws.get('ctype'), **kwargs)
            ws['state'], ws['ctype'] = res.state, res.ctype
            if 'q' in ws['outdated']:
                ws['q_ws'] = np.asarray(desc.canon_at(ws['theta'])['q'], dtype=np.float64)
        ws['outdated'] = set()
        t1 = time.time()

        prob._clear_solution()
        k = 0
        for v in desc.variables:                    # templates/cpg_solver.py.jinja2:76-80
            sz = int(v.indices.size)
            prob.var_dict[v.name].save_value(np.array(res.prim_flat[0, k:k + sz]).reshape(v.shape, order='F'))
            k += sz
        for i, d in enumerate(desc.duals):
            dv = res.dual[d.name][0]
            prob.constraints[i].save_dual_value(np.array(dv).reshape(d.shape) if d.shape else float(dv))
        if conic:
            # integer status, formatted as the reference does (cvxpygen/utils.py:1598-1601)
            docu = 'https://github.com/embotech/ecos/wiki/Usage-from-C' if desc.solver == 'ECOS' else 'https://oxfordcontrol.github.io/ClarabelDocs/'
            status = '%d (for description visit %s)' % (int(res.status[0]), docu)
        else:
            status = STATUS_STRINGS.get(int(res.status[0]), 'unknown')
        prob._status = status
        obj = float(res.obj_val[0])
        prob._value = obj                           # +-1e30 already mapped to +-inf by the runtime
        primal_vars = {var.id: var.value for var in prob.variables()}
        dual_vars = {c.id: c.dual_value for c in prob.constraints}
        solver_specific_stats = {'obj_val': obj, 'status': status, 'iter': int(res.iter[0]),
                                 'pri_res': float(res.pri_res[0]), 'dua_res': float(res.dua_res[0]),
                                 'time': res.kernel_ms * 1e-3}
        attr = {'solve_time': t1 - t0, 'solver_specific_stats': solver_specific_stats,
                'num_iters': int(res.iter[0])}
        prob._solution = make_solution(prob.status, prob.value, primal_vars, dual_vars, attr)
        prob._solver_stats = make_solver_stats({'solver_specific_stats': solver_specific_stats,
                                                'num_iters': int(res.iter[0]),
                                                'solve_time': t1 - t0}, 'CLARABEL' if self.two_stage else desc.solver)
        self._last = (res.sol_x[0].copy(), res.sol_y[0].copy()) if res.sol_x is not None else None
        return prob.value

    # ---- gradient=True surface (templates/cpg_solver.py.jinja2:122-212) ---------------------------------
    def cpg_solve_and_gradient_info(self, prob, updated_params=None, **kwargs):
        if not self.gradient:
            raise AttributeError('code was generated with gradient=False')
        val = self.cpg_solve(prob, updated_params, **kwargs)
        gp, gd = self._last
        return val, list(gp), list(gd)

    def cpg_gradient(self, prob, gradient_sol_primal=None, gradient_sol_dual=None):
        """reads `var.gradient` of every variable, writes `param.gradient` of every parameter"""
        if not self.gradient:
            raise AttributeError('code was generated with gradient=False')
        desc = self.desc
        if gradient_sol_primal is not None and gradient_sol_dual is not None:
            sx, sy = np.asarray(gradient_sol_primal, dtype=np.float64), np.asarray(gradient_sol_dual, dtype=np.float64)
        else:
            sx, sy = self._last
        dvars = {}
        for v in desc.variables:
            g = prob.var_dict[v.name].gradient
            dvars[v.name] = np.asarray(0.0 if g is None else g, dtype=np.float64).reshape((1,) + tuple(v.shape))
        # the canonical P / A the adjoint differentiates are those of the workspace, i.e. of the last
        # cpg_solve (cpg_params_vec), not whatever `prob` holds now (writer.py:233-266)
        th = self._workspace()['theta']
        vals = {q.name: th[q.col:q.col + q.size].reshape(1, -1) for q in desc.params}
        out = self.batch_solver.gradient(vals, sx[None, :], sy[None, :], dvars, updated_params=desc.param_names)
        for q in desc.params:
            g = out[q.name][0]
            prob.param_dict[q.name].gradient = float(g) if q.kind == 'scalar' else np.array(g)

    def forward(self, params, context):
        """cvxpylayers `custom_method` forward (templates/cpg_solver.py.jinja2:176-193).  When a parameter
        value carries a leading batch axis (cvxpylayers hands the whole batch over; the reference's users loop
        over it one instance at a time, examples/paper_grad/ADP.py:80-88) the batch goes to the GPU as ONE
        solve: every instance from the code-generation-time workspace, cold start."""
        info = {}
        kwargs = context.solver_args.copy()
        prob = kwargs.pop("problem")
        parameters = prob.parameters()
        plist = [next(p for p in parameters if p.id == pid) for pid in context.param_ids]
        if any(self._batch_size(p, val) is not None for p, val in zip(plist, params)):
            return self._forward_batch(plist, params, context, prob, kwargs)
        for p, val in zip(plist, params):
            p.value = val
        updated_params = kwargs.pop("updated_params", None)
        _, info["gradient_primal"], info["gradient_dual"] = self.cpg_solve_and_gradient_info(prob, updated_params, **kwargs)
        info["prob"] = prob
        vars_ = prob.variables()
        return [next(v for v in vars_ if v.id == variable.id).value for variable in context.variables], info

    def _batch_size(self, p, val):
        """leading batch axis of a parameter value: [B, *shape], or [B, size] for values that are already
        flattened the way the reference stores them (F-order / diagonal / stored non-zeros); None: one instance"""
        v = np.asarray(val)
        shape, size = tuple(p.shape), self.desc.param(p.name()).size
        if v.shape == shape or (v.ndim <= 1 and v.size == size):
            return None
        if v.ndim >= 1 and (v.shape[1:] == shape or (v.ndim == 2 and v.shape[1] == size)):
            return int(v.shape[0])
        raise ValueError(f'value of parameter {p.name()} has shape {v.shape}, expected {shape} or a leading batch axis')

    def _forward_batch(self, plist, params, context, prob, kwargs):
        if not self.gradient:
            raise AttributeError('code was generated with gradient=False')
        sizes = [self._batch_size(p, val) for p, val in zip(plist, params)]
        B = max(b for b in sizes if b is not None)
        vals = {}
        for p, val, b in zip(plist, params, sizes):
            v = np.asarray(val, dtype=np.float64)
            if b is None:                                        # shared by the whole batch
                v = np.broadcast_to(v, (B,) + v.shape)
            elif b != B:
                raise ValueError('inconsistent batch sizes')
            vals[p.name()] = np.ascontiguousarray(v)
        kwargs.pop("updated_params", None)
        kwargs = self._filter_settings(kwargs)
        kwargs.pop('warm_start', None); kwargs.pop('warm_starting', None)     # independent instances: cold start
        names = [q.name for q in self.desc.params if q.name in vals]
        res = self.batch_solver.solve(vals, updated_params=names, **kwargs)
        info = {"batched": True, "gradient_primal": res.sol_x, "gradient_dual": res.sol_y, "prob": prob,
                "batch_params": vals, "batch_names": names, "status": res.status, "iter": res.iter}
        return [res.prim[variable.name()] for variable in context.variables], info

    def backward(self, dvars, context):
        prob = context.info["prob"]
        if context.info.get("batched"):
            vals, names = context.info["batch_params"], context.info["batch_names"]
            dv = {variable.name(): np.asarray(g, dtype=np.float64) for variable, g in zip(context.variables, dvars)}
            out = self.batch_solver.gradient(vals, context.info["gradient_primal"], context.info["gradient_dual"], dv,
                                             updated_params=names)
            params = prob.parameters()
            return [out[next(p for p in params if p.id == pid).name()] for pid in context.param_ids], {}
        vars_ = prob.variables()
        for variable, dv in zip(context.variables, dvars):
            next(v for v in vars_ if v.id == variable.id).gradient = dv
        self.cpg_gradient(prob, context.info["gradient_primal"], context.info["gradient_dual"])
        params = prob.parameters()
        return [next(p for p in params if p.id == pid).gradient for pid in context.param_ids], {}

    # batched siblings
    def cpg_solve_and_gradient_info_batch(self, params, updated_params=None, **kwargs):
        res = self.cpg_solve_batch(params, updated_params, **kwargs)
        return res, res.sol_x, res.sol_y

    def cpg_gradient_batch(self, params, sol_x, sol_y, dvars, updated_params=None):
        return self.batch_solver.gradient(params, sol_x, sol_y, dvars, updated_params)
