import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from cvxpygen_amd import families
from cvxpygen_amd.runtime import BatchSolver
from oracle import binding
d = families.portfolio(100, 10)
B = 16
rng = np.random.default_rng(31)
sig = np.zeros((B, 10, 10)); sig[:, np.arange(10), np.arange(10)] = rng.random((B, 10))
pv_all = {'a': rng.standard_normal((B, 100)), 'F': np.round(rng.standard_normal((B, 100, 10))),
      'Sig_f_sqrt': sig, 'd_sqrt': rng.random((B, 100)), 'w_prev': np.zeros((B, 100))}
print(d.user_p_name_to_canon_outdated())
for names in (['a'], ['F'], ['Sig_f_sqrt'], ['d_sqrt'], ['w_prev'], ['F', 'a'], list(pv_all)):
    pv = {k: pv_all[k] for k in names}
    th = np.tile(d.theta0, (B, 1))
    for k in range(B):
        th[k] = d.theta_from_values({nm: v[k] for nm, v in pv.items()})
    for mi in (25, 4000):
        o = binding.cpg_solve_batch(d, th, names, max_iter=mi)
        bs = BatchSolver(d, full_output=True)
        r = bs.solve(pv, updated_params=names, max_iter=mi)
        print(names, 'max_iter', mi, 'iter mismatches', int((r.iter != o['iter']).sum()),
              'x err %.2e' % np.abs(r.sol_x - o['sol_x']).max(), 'y err %.2e' % np.abs(r.sol_y - o['sol_y']).max(),
              'pri %.2e' % np.abs(r.pri_res - o['pri_res']).max(), 'dua %.2e' % np.abs(r.dua_res - o['dua_res']).max())
        bs.close()
