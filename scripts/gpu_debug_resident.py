"""GPU debugging aid of the resident per-instance factor kernel: a few portfolio instances through the family library,
against the oracle, at increasing iteration limits (python scripts/gpu_debug_resident.py [B])."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cvxpygen_amd import families, codegen
from cvxpygen_amd.runtime import BatchSolver, build_family_plan
from oracle import binding as oracle_lib
oracle_lib.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = families.portfolio(100, 10)
plan = build_family_plan(d)
lib = os.path.join(ROOT, 'cvxpygen_amd', 'generated', 'portfolio', 'libcpg_portfolio.so')
rng = np.random.default_rng(5)
n, m = 100, 10
sig = np.zeros((B, m, m)); sig[:, np.arange(m), np.arange(m)] = rng.random((B, m))
vals = {'a': rng.standard_normal((B, n)), 'F': np.round(rng.standard_normal((B, n, m))), 'Sig_f_sqrt': sig,
        'd_sqrt': rng.random((B, n)), 'w_prev': np.zeros((B, n))}
th = np.stack([d.theta_from_values({k: v[i] for k, v in vals.items()}) for i in range(B)])
upd = ['a', 'F', 'Sig_f_sqrt', 'd_sqrt', 'w_prev']
bs = BatchSolver(d, lib_path=lib, plan=plan)
import json
STGS = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [dict(max_iter=1), dict(max_iter=2), dict(max_iter=3), dict(max_iter=25), dict(max_iter=60), dict()]
for stg in STGS:
    print('>>', stg, flush=True)
    for placement in ((-1,) if 'debug_stage' in stg else (-1, 0)):
        bs.set_program_placement(placement)
        r = bs.solve(vals, updated_params=upd, **stg)
        v = C.c_double(-1)
        bs.lib.check(bs.lib.L.cpg_hip_get_setting(bs.h_ref, b'resident_executor', C.byref(v)), 'get')
        o = oracle_lib.cpg_solve_batch(d, th, upd, **{k_: v_ for k_, v_ in stg.items() if k_ != 'debug_stage'})
        prim = np.concatenate([o['sol_x'][:, v_.indices] for v_ in d.variables], axis=1)
        ok = np.isin(o['status'], (1, 2, 7)) & np.isin(r.status, (1, 2, 7))
        err = np.abs(r.prim_flat[ok] - prim[ok]).max() / max(1, np.abs(prim[ok]).max()) if ok.any() else float('nan')
        print(stg, 'resident' if v.value else 'stream', 'iter', r.iter.tolist(), o['iter'].tolist(), 'status', r.status.tolist(), o['status'].tolist(),
              'prim err %.2e' % err, 'pri_res', np.array2string(r.pri_res, precision=3), np.array2string(o['pri_res'], precision=3),
              'dua_res', np.array2string(r.dua_res, precision=3), np.array2string(o['dua_res'], precision=3), flush=True)
