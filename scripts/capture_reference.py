#!/usr/bin/env python
"""
Capture outputs of the REAL reference -- cvxpygen-generated solvers -- for the five BASELINE configurations.

Runs only where `import cvxpy, cvxpygen, osqp, clarabel` succeeds (pip install cvxpygen; Python >= 3.11; a C
compiler + cmake for the generated extensions).  It cannot run in the build container of this repository (no
network, no cvxpy) and it never travels to a GPU box as code: what it writes is DATA,

    tests/golden/reference_outputs.npz      parameter values in, what the generated solver returned out
    tests/golden/reference_workspace.json   the OSQP settings block of every generated workspace.c + package versions

and tests/test_reference_outputs.py compares the HIP path (and the CPU oracle) with it: user-level primal / dual
values and objective within 1e-6 relative, iteration counts and status exactly.  While the files are absent that
test is skipped and every solver-parity statement of this repository stays "versus the restatement"
 -- running this script once is what turns it into "versus the reference", and it settles
which OSQP fork (rho adaptation / duality-gap test) the generated code really runs: the settings block is recorded
verbatim.

Problem definitions are the reference's own (file:line cited at each); parameter values are the seeded draws
SURVEY.md section 8(d) and bench.py use, stored in the npz so that the consumer needs no generator.

    python scripts/capture_reference.py [--instances 64] [--out tests/golden] [--keep-code DIR]
"""
import argparse
import importlib
import json
import os
import re
import shutil
import sys
import tempfile

import numpy as np


def _problems(cp):
    """name -> (problem, solver, gradient, function(rng, k) -> dict of parameter values for instance k,
    dict of code-generation-time parameter values, updated_params)"""
    out = {}

    # ---- config 1: examples/main.py:16-25 ---------------------------------------------------------------
    import scipy.sparse as sp
    m, n = 3, 2
    x = cp.Variable(n, name='x')
    A = cp.Parameter((m, n), name='A', sparsity=((0, 0, 1), (0, 1, 1)))
    b = cp.Parameter(m, name='b')
    p1 = cp.Problem(cp.Minimize(cp.sum_squares(A @ x - b)), [x >= 0])
    rs = np.random.RandomState(1)
    A0, b0 = rs.randn(3), rs.randn(m)

    def set1(vals):
        A.value_sparse = sp.coo_array((np.asarray(vals['A']), A.sparse_idx), shape=(m, n))
        b.value = np.asarray(vals['b'])
    out['config1_nonneg_LS'] = dict(problem=p1, solver='OSQP', gradient=False, setter=set1, gen={'A': A0, 'b': b0},
                                    draw=lambda rng, k: ({'A': A0, 'b': b0} if k == 0 else
                                                         {'A': rng.standard_normal(3), 'b': rng.standard_normal(m)}),
                                    updated=['A', 'b'])

    # ---- config 2 (+5): examples/MPC.ipynb cell 1 / 3 at (n, m) = (6, 3) and (12, 4) ---------------------------
    def mpc(nx, nu, H=10):
        U = cp.Variable((nu, H), name='U')
        X = cp.Variable((nx, H + 1), name='X')
        Psqrt = cp.Parameter((nx, nx), name='Psqrt'); Qsqrt = cp.Parameter((nx, nx), name='Qsqrt')
        Rsqrt = cp.Parameter((nu, nu), name='Rsqrt'); Ap = cp.Parameter((nx, nx), name='A')
        Bp = cp.Parameter((nx, nu), name='B'); x_init = cp.Parameter(nx, name='x_init')
        obj = cp.Minimize(cp.sum_squares(Psqrt @ X[:, H]) + cp.sum_squares(Qsqrt @ X[:, :H]) + cp.sum_squares(Rsqrt @ U))
        cons = [X[:, 1:] == Ap @ X[:, :H] + Bp @ U, cp.abs(U) <= 1, X[:, 0] == x_init]
        prob = cp.Problem(obj, cons)
        h = nx // 2
        A_cont = np.zeros((nx, nx)); A_cont[:h, h:] = np.eye(h)          # double integrator, SURVEY.md 8(d)
        B_cont = np.zeros((nx, nu)); B_cont[h:h + nu, :] = np.eye(nu)
        gen = {'Psqrt': np.eye(nx), 'Qsqrt': np.eye(nx), 'Rsqrt': np.sqrt(0.1) * np.eye(nu),
               'A': np.eye(nx) + 0.1 * A_cont, 'B': 0.1 * B_cont,
               'x_init': np.array([2, 2, 2, -1, -1, 1.0]) if nx == 6 else np.resize(np.array([2, 2, 2, -1, -1, 1.0]), nx)}

        def setter(vals):
            for k_, v in vals.items():
                prob.param_dict[k_].value = np.asarray(v)
        return prob, gen, setter
    for tag, (nx, nu) in (('config2_mpc_6_3_10', (6, 3)), ('config2_mpc_12_4_10', (12, 4))):
        prob, gen, setter = mpc(nx, nu)
        out[tag] = dict(problem=prob, solver='OSQP', gradient=False, setter=setter, gen=gen,
                        draw=(lambda nx_: lambda rng, k: {'x_init': -2 + 4 * rng.random(nx_)})(nx), updated=['x_init'])
    prob, gen, setter = mpc(12, 4)
    out['config5_mpc_12_4_10_gradient'] = dict(problem=prob, solver='OSQP', gradient=True, setter=setter, gen=gen,
                                               draw=lambda rng, k: {'x_init': -2 + 4 * rng.random(12)}, updated=['x_init'])

    # ---- config 3: examples/portfolio.ipynb cells 1 / 3 / 7 -----------------------------------------------------------
    n, m = 100, 10
    w = cp.Variable(n, name='w'); delta_w = cp.Variable(n, name='delta_w'); f = cp.Variable(m, name='f')
    a = cp.Parameter(n, name='a'); F = cp.Parameter((n, m), name='F'); Sig = cp.Parameter((m, m), name='Sig_f_sqrt')
    d_sqrt = cp.Parameter(n, name='d_sqrt'); k_tc = cp.Parameter(n, nonneg=True, name='k_tc')
    k_sh = cp.Parameter(n, nonneg=True, name='k_sh'); w_prev = cp.Parameter(n, name='w_prev')
    Lp = cp.Parameter(nonneg=True, name='L')
    obj = cp.Maximize(a @ w - cp.sum_squares(Sig @ f) - cp.sum_squares(cp.multiply(d_sqrt, w))
                      - k_tc @ cp.abs(delta_w) + k_sh @ cp.minimum(0, w))
    cons = [f == F.T @ w, np.ones(n) @ w == 1, cp.norm(w, 1) <= Lp, delta_w == w - w_prev]
    p3 = cp.Problem(obj, cons)
    rs = np.random.RandomState(0)
    alpha = rs.randn(n)
    gen3 = {'a': alpha, 'F': rs.randn(n, m), 'Sig_f_sqrt': rs.rand(m, m), 'd_sqrt': rs.rand(n),
            'k_tc': 0.01 * np.ones(n), 'k_sh': 0.05 * np.ones(n)}
    wp = rs.rand(n)
    gen3['w_prev'] = wp / np.linalg.norm(wp); gen3['L'] = 1.6

    def set3(vals):
        for k_, v in vals.items():
            p3.param_dict[k_].value = np.asarray(v) if np.ndim(v) else float(v)

    def draw3(rng, k):
        return {'a': rng.standard_normal(n), 'F': np.round(rng.standard_normal((n, m))),
                'Sig_f_sqrt': np.diag(rng.random(m)), 'd_sqrt': rng.random(n), 'w_prev': np.zeros(n)}
    out['config3_portfolio'] = dict(problem=p3, solver='OSQP', gradient=False, setter=set3, gen=gen3, draw=draw3,
                                    updated=['a', 'F', 'Sig_f_sqrt', 'd_sqrt', 'w_prev'])

    # ---- config 4: tests/test_E2E_SOCP.py:15-35, data :38-62 ----------------------------------------------------------
    u = cp.Variable((2, 3), name='u')
    Rsq = cp.Parameter((3, 3), name='Rsqrt', diag=True); fp = cp.Parameter(6, name='f'); G = cp.Parameter((6, 3), name='G')
    p4 = cp.Problem(cp.Minimize(cp.sum_squares(fp + G @ u[0]) + cp.sum_squares(Rsq @ u[0])), [cp.norm(u, 2, axis=1) <= 0.1])

    def adp_vals(state):
        Ac = np.zeros((6, 6)); Ac[:3, 3:] = np.eye(3); Ac[3:, 3:] = -np.diag(state[3:])
        Bc = np.concatenate((np.zeros((3, 3)), np.diag(state[3:])), axis=0)
        return {'Rsqrt': np.sqrt(0.1) * np.eye(3), 'f': (np.eye(6) + 0.1 * Ac) @ state, 'G': 0.1 * Bc}

    def set4(vals):
        for k_, v in vals.items():
            p4.param_dict[k_].value = np.asarray(v)
    np.random.seed(0)
    out['config4_adp_socp'] = dict(problem=p4, solver='CLARABEL', gradient=False, setter=set4,
                                   gen=adp_vals(-2 * np.ones(6) + 4 * np.random.rand(6)),
                                   draw=lambda rng, k: {kk: vv for kk, vv in adp_vals(-2 + 4 * rng.random(6)).items() if kk != 'Rsqrt'},
                                   updated=['f', 'G'])
    return out


def _settings_block(code_dir):
    """the initialiser of the OSQPSettings struct in the generated workspace.c, verbatim, + parsed name -> value"""
    for root, _, files in os.walk(code_dir):
        for fn in files:
            if fn == 'workspace.c':
                txt = open(os.path.join(root, fn)).read()
                mt = re.search(r'OSQPSettings\s+\w+\s*=\s*\{(.*?)\};', txt, re.S)
                if mt:
                    body = mt.group(1)
                    vals = [v.strip() for v in re.sub(r'/\*.*?\*/', '', body, flags=re.S).split(',') if v.strip()]
                    return {'verbatim': body.strip(), 'values': vals}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--instances', type=int, default=64)
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
    ap.add_argument('--keep-code', default=None)
    args = ap.parse_args()
    import cvxpy as cp
    from cvxpygen import cpg
    versions = {'python': sys.version.split()[0], 'cvxpy': cp.__version__}
    for mod in ('cvxpygen', 'osqp', 'clarabel', 'numpy', 'scipy'):
        try:
            versions[mod] = getattr(importlib.import_module(mod), '__version__', 'unknown')
        except ImportError:
            versions[mod] = None
    work = args.keep_code or tempfile.mkdtemp(prefix='cpg_ref_')
    os.makedirs(work, exist_ok=True)
    sys.path.insert(0, work)
    arrays, meta = {}, {'versions': versions, 'configs': {}}
    for name, cfg in _problems(cp).items():
        prob = cfg['problem']
        cfg['setter'](cfg['gen'])
        code_dir = os.path.join(work, name)
        cpg.generate_code(prob, code_dir=code_dir, solver=cfg['solver'], gradient=cfg['gradient'], wrapper=True,
                          prefix=name if cfg['gradient'] else '')
        mod = importlib.import_module(f'{name}.cpg_solver')
        prob.register_solve('CPG', mod.cpg_solve)
        rng = np.random.default_rng(1000)                        # bench.py: default_rng(1000 + rank), rank 0
        B = args.instances if name != 'config3_portfolio' else min(args.instances, 16)
        rec = {k: [] for k in ('obj', 'iter', 'status')}
        pv, xv, dv, gv = {}, {}, {}, {}
        for k in range(B):
            # The extension keeps ONE static workspace: instance 0 sees the code-generation-time workspace (= an
            # instance of a batch); every later one sees what its predecessors left -- parameter values, and inside
            # OSQP the rho / factor of the last adapt_rho (cold start via warm_start=False resets the iterates
            # only).  The consumer therefore REPLAYS the same call sequence through the B = 1 drop-in
            # (tests/test_reference_outputs.py), which models exactly that state; instance 0 doubles as the
            # batch-semantics check.
            vals = cfg['draw'](rng, k)
            cfg['setter'](vals)
            val = prob.solve(method='CPG', updated_params=cfg['updated'], warm_start=False) if cfg['solver'] == 'OSQP' \
                else prob.solve(method='CPG', updated_params=cfg['updated'])
            for kk, v in vals.items():
                pv.setdefault(kk, []).append(np.asarray(v, dtype=float))
            for v in prob.variables():
                xv.setdefault(v.name(), []).append(np.asarray(v.value, dtype=float))
            for i, c in enumerate(prob.constraints):
                dv.setdefault(f'd{i}', []).append(np.asarray(c.dual_value, dtype=float))
            st = prob.solver_stats
            rec['obj'].append(float(val)); rec['iter'].append(int(st.num_iters)); rec['status'].append(str(prob.status))
            if cfg['gradient']:
                for v in prob.variables():
                    v.gradient = 0.1 * np.ones(v.shape)           # 0.1 * sol.sum(), tests/test_diff.py:38
                mod.cpg_gradient(prob)
                for p in prob.parameters():
                    gv.setdefault(p.name(), []).append(np.asarray(p.gradient, dtype=float))
        for kk, lst in pv.items():
            arrays[f'{name}/param/{kk}'] = np.stack(lst)
        for kk, lst in xv.items():
            arrays[f'{name}/prim/{kk}'] = np.stack(lst)
        for kk, lst in dv.items():
            arrays[f'{name}/dual/{kk}'] = np.stack(lst)
        for kk, lst in gv.items():
            arrays[f'{name}/grad/{kk}'] = np.stack(lst)
        arrays[f'{name}/obj'] = np.array(rec['obj']); arrays[f'{name}/iter'] = np.array(rec['iter'], dtype=np.int32)
        meta['configs'][name] = {'solver': cfg['solver'], 'gradient': cfg['gradient'], 'updated_params': cfg['updated'],
                                 'status': rec['status'], 'instances': B,
                                 'gen_params': {k: np.asarray(v, dtype=float).tolist() for k, v in cfg['gen'].items()},
                                 'osqp_settings_in_workspace_c': _settings_block(code_dir) if cfg['solver'] == 'OSQP' else None,
                                 'var_order': [v.name() for v in prob.variables()],
                                 'first_solve_is_fresh_workspace': True}
        print(f'{name}: {B} instances, iterations {sorted(set(rec["iter"]))}, statuses {sorted(set(rec["status"]))}', flush=True)
    os.makedirs(args.out, exist_ok=True)
    np.savez_compressed(os.path.join(args.out, 'reference_outputs.npz'), **arrays)
    json.dump(meta, open(os.path.join(args.out, 'reference_workspace.json'), 'w'), indent=1)
    if not args.keep_code:
        shutil.rmtree(work, ignore_errors=True)
    print('wrote', os.path.join(args.out, 'reference_outputs.npz'), 'and reference_workspace.json')


if __name__ == '__main__':
    main()
