"""Where does the conic interior-point kernel spend its time?  Kernel time of 100 000 ADP instances with pieces switched off through
the SETTINGS (no rebuild): iterative refinement, equilibration, a single iteration.  Results change, of course: timing only.
    python scripts/gpu_probe_conic.py [B]
    python scripts/gpu_probe_conic.py twice      # libraries built with -DCPG_CONIC_TWICE=k under cvxpygen_amd/generated/variants/adp_twice_k:
                                                 # piece k of an iteration executed twice (results unchanged) -> the time it adds is its cost"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from cvxpygen_amd import families, codegen
from cvxpygen_amd.conic_plan import build_conic_plan
from cvxpygen_amd.conic_runtime import ConicBatchSolver

TWICE = len(sys.argv) > 1 and sys.argv[1] == 'twice'
B = int(sys.argv[1]) if len(sys.argv) > 1 and not TWICE else 100000
d = families.adp()
cplan = build_conic_plan(d)
lib = codegen.build_conic_library(cplan, os.path.join(ROOT, 'cvxpygen_amd', 'generated', 'adp'), 'adp')
cs = ConicBatchSolver(d, device=0, lib_path=lib, plan=cplan)
pv = bench.adp_params(B, 1000)
if TWICE:
    names = {0: 'as it is', 1: 'factorisation', 2: 'substitution sweeps (every KKT solve and refinement pass)', 4: 'refinement residuals',
             8: 'NT scaling', 16: 'step lengths', 32: 'combined-step offset', 64: 'step assembly'}
    base = None
    for k, nm in names.items():
        lp = os.path.join(ROOT, 'cvxpygen_amd', 'generated', 'variants', f'adp_twice_{k}', 'libcpg_adp.so')
        if not os.path.exists(lp):
            continue
        c2 = ConicBatchSolver(d, device=0, lib_path=lp, plan=cplan)
        c2.set_updated(list(pv.keys()))
        th = c2.theta_var(pv)
        for rep in range(3):
            r = c2.solve(theta_var=th)
        base = r.kernel_ms if k == 0 else base
        print(f'{nm:62s} twice: kernel {r.kernel_ms:7.3f} ms  (+{r.kernel_ms - base:6.3f} ms = {100 * (r.kernel_ms - base) / base:5.1f} % of the kernel)   mean iter {r.iter.mean():.2f} solved {(r.status == 1).sum()}')
        c2.close()
    sys.exit(0)
cs.set_updated(list(pv.keys()))
theta = cs.theta_var(pv)
for name, stg in (('defaults', {}), ('no iterative refinement', dict(iterative_refinement_enable=0)), ('no equilibration', dict(equilibrate_enable=0)),
                  ('max_iter 1', dict(max_iter=1)), ('max_iter 2', dict(max_iter=2)), ('max_iter 3', dict(max_iter=3)),
                  ('max_iter 1, no equilibration', dict(max_iter=1, equilibrate_enable=0)),
                  ('no refinement, no static regularisation', dict(iterative_refinement_enable=0, static_regularization_enable=0))):
    for rep in range(2):
        r = cs.solve(theta_var=theta, **stg)
    print(f'{name:42s} kernel {r.kernel_ms:7.3f} ms   mean iter {r.iter.mean():5.2f}   solved {(r.status == 1).sum()}')
print('-- resident wavefronts per CU (waves per workgroup x workgroups per CU): what a lone pair of waves per SIMD sustains')
for wpb, bpc in ((0, 0), (16, 1), (14, 1), (12, 1), (8, 1), (7, 2), (6, 2), (4, 1)):
    cs.set_launch(wpb, 0, bpc)
    for rep in range(2):
        r = cs.solve(theta_var=theta)
    print(f'waves per workgroup {wpb}, workgroups per CU {bpc}: kernel {r.kernel_ms:7.3f} ms')
cs.close()
