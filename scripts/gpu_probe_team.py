"""Team kernel on the GPU: parity with the oracle on a few instances, stage timing from inside (debug_stage 20: each
instance's stage time stamps replace its primal results) and the kernel time of a batch.
    python scripts/gpu_probe_team.py mpc12|portfolio <library> [B] [B_probe]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from cvxpygen_amd import families
from cvxpygen_amd.runtime import BatchSolver, build_family_plan

fam, lib = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
Bp = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
check = os.environ.get('CPG_PROBE_CHECK', '1') != '0'


def params(d, n, seed):
    if fam == 'portfolio':
        pv = bench.portfolio_params(d, n, seed)
        return pv, list(pv.keys())
    rng = np.random.default_rng(seed)
    full = np.tile(d.theta0[:-1], (n, 1)) * (1 + 0.05 * rng.standard_normal((n, d.NP)))
    p = d.param('x_init')
    full[:, p.col:p.col + p.size] = bench.make_theta(d, n, seed=seed)
    return {q.name: full[:, q.col:q.col + q.size] for q in d.params}, None


d = families.portfolio(100, 10) if fam == 'portfolio' else families.mpc(12, 4, 10)
plan = build_family_plan(d)
bs = BatchSolver(d, lib_path=lib, plan=plan)
import ctypes as C
if check:
    from oracle import binding as ob
    pv, upd = params(d, 48, 1000)
    r = bs.solve(pv, updated_params=upd)
    v = C.c_double(-1); bs.lib.L.cpg_hip_get_setting(bs.h_ref, b'team_executor', C.byref(v))
    th = np.tile(d.theta0, (48, 1))
    for q in d.params:
        if q.name in pv:
            th[:, q.col:q.col + q.size] = np.asarray(pv[q.name]).reshape(48, -1) if fam != 'portfolio' else th[:, q.col:q.col + q.size]
    if fam == 'portfolio':
        th = np.stack([d.theta_from_values({k: v_[i] for k, v_ in pv.items()}) for i in range(48)])
    o = ob.cpg_solve_batch(d, th, upd)
    po = np.concatenate([o['sol_x'][:, v_.indices] for v_ in d.variables], axis=1)
    do = np.concatenate([o['sol_y'][:, v_.indices] for v_ in d.duals], axis=1)
    print('team executor in use:', v.value, '| iteration mismatches', int((r.iter != o['iter']).sum()), 'of 48, status mismatches', int((r.status != o['status']).sum()),
          '| prim relerr %.2e dual relerr %.2e' % (np.abs(po - r.prim_flat).max() / np.abs(po).max(), np.abs(do - r.dual_flat).max() / np.abs(do).max()),
          '| iters', sorted(set(r.iter.tolist()))[:6])
pv, upd = params(d, Bp, 1001)
stg_probe = {'max_iter': 50} if os.environ.get('CPG_PROBE_TIMING_ONLY', '0') == '1' else {}      # (builds that compute garbage on purpose)
r = bs.solve(pv, updated_params=upd, debug_stage=20, **stg_probe)
ts = r.prim_flat[:, :8] * 0.01            # microseconds since the instance started
names = ['setup', 'factor', 'store', 'iterate', 'check', 'next1', 'next2']
d_ = np.diff(ts, axis=1)
print('probe instances', Bp, 'kernel ms', round(r.kernel_ms, 2))
for k in range(7):
    print(f'  {names[k]:8s} mean {d_[:, k].mean():9.1f} us   median {np.median(d_[:, k]):9.1f}   p90 {np.percentile(d_[:, k], 90):9.1f}')
if os.environ.get('CPG_PROBE_FACTOR', '0') == '1':
    r = bs.solve(pv, updated_params=upd, debug_stage=23)
    ts = r.prim_flat[:, :8] * 0.01
    d_ = np.diff(ts, axis=1)
    for k, nm in enumerate(['setup', 'KKT values', "LDL' part (wavefront 0)", 'block inverses (team)', 'store', 'iterate', 'check']):
        print(f'  [factor probe] {nm:26s} mean {d_[:, k].mean():9.1f} us   median {np.median(d_[:, k]):9.1f}')
pv, upd = params(d, B, 1002)
for rep in range(0 if os.environ.get('CPG_PROBE_TIMING_ONLY', '0') == '1' else 3):
    r = bs.solve(pv, updated_params=upd)
    print(f'B {B}: kernel {r.kernel_ms:.2f} ms = {B / r.kernel_ms:.1f} k instances/s, mean iter {r.iter.mean():.1f}, solved {(r.status == 1).sum()}')
bs.close()
