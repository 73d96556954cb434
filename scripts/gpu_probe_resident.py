"""Stage timing of the resident kernel from inside (debug_stage 20: each instance's stage time stamps replace its primal
results): python scripts/gpu_probe_resident.py [B] [json settings]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench
from cvxpygen_amd import families, codegen
from cvxpygen_amd.runtime import BatchSolver, build_family_plan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
stg = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
d = families.portfolio(100, 10)
plan = build_family_plan(d)
lib = os.path.join(ROOT, 'cvxpygen_amd', 'generated', 'portfolio', os.environ.get('CPG_PROBE_LIB', 'libcpg_portfolio.so'))
pv = bench.portfolio_params(d, B, 1000)
bs = BatchSolver(d, lib_path=lib, plan=plan)
r = bs.solve(pv, updated_params=list(pv.keys()), debug_stage=int(os.environ.get('CPG_PROBE_STAGE', '20')), **stg)
ts = r.prim_flat[:, :8] * 0.01            # microseconds since the instance started
names = ['start', 'setup', 'factor', 'store', 'iterate', 'check', 'next', 'next2']
d_ = np.diff(ts, axis=1)
print('instances', B, 'iter', r.iter[:4])
for k in range(7):
    print(f'  {names[k+1]:8s} mean {d_[:, k].mean():9.1f} us   median {np.median(d_[:, k]):9.1f}   p90 {np.percentile(d_[:, k], 90):9.1f}')
