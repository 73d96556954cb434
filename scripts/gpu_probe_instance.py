"""Stage timing of the generated instance kernel (config 2's per-instance phase) from inside: debug_stage 20 makes every
handed-over instance write the 100 MHz time stamps of its stages over its first primal results.
    python scripts/gpu_probe_instance.py [B] [json settings]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from cvxpygen_amd import families
from cvxpygen_amd.runtime import BatchSolver, build_family_plan

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
stg = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
d = families.mpc(12, 4, 10)
plan = build_family_plan(d)
lib = os.path.join(ROOT, 'cvxpygen_amd', 'generated', 'mpc12', os.environ.get('CPG_PROBE_LIB', 'libcpg_mpc12.so'))
bs = BatchSolver(d, lib_path=lib, plan=plan)
bs.set_updated(['x_init'])
th = bench.make_theta(d, B, seed=1000)
r0 = bs.solve(theta_var=th, **stg)
r = bs.solve(theta_var=th, debug_stage=20, **stg)
ho = r0.iter > 50
print('instances', B, 'handed over', int(ho.sum()), 'mean iter', r0.iter.mean(), 'phase ms', bs.last_phase_ms())
ts = r.prim_flat[ho][:, :8] * 0.01            # microseconds since the instance started in the instance kernel
names = ['start', 'classes', 'factor', 'state', 'iterate', 'check', 'iterate2/final', 'check2/final', '...']
one = ts[(r0.iter[ho] == 75)]
print('instances that finish at the test of iteration 75:', len(one))
d_ = np.diff(one[:, :7], axis=1)
for k in range(6):
    print(f'  {names[k + 1]:16s} mean {d_[:, k].mean():9.1f} us   median {np.median(d_[:, k]):9.1f}   p90 {np.percentile(d_[:, k], 90):9.1f}')
print('  total per instance (mean)', one[:, 6].mean() if one.shape[1] > 6 else None)
