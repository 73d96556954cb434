"""PCIe-inclusive rate of the host-pointer entry point for the headline workload."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from cvxpygen_amd.runtime import BatchSolver
desc, _ = bench.make_workload('mpc12')
gen = os.path.join('cvxpygen_amd', 'generated', 'mpc12', 'libcpg_mpc12.so')
s = BatchSolver(desc, lib_path=gen if os.path.exists(gen) else None)
B = 100000
x0 = bench.make_theta(desc, B, 5)
s.solve({'x_init': x0}, updated_params=['x_init'])
ts = []
for _ in range(3):
    t0 = time.perf_counter(); r = s.solve({'x_init': x0}, updated_params=['x_init']); ts.append(time.perf_counter() - t0)
print('host-pointer solve (H2D + kernel + D2H + host staging): %.1f ms -> %.0f instances/s; kernel %.1f ms' % (1e3 * min(ts), B / min(ts), r.kernel_ms))
