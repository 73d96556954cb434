"""GPU probe of the exponential / power cone families (round 6): kernel time per batch, iteration statistics, closed-form errors.
Usage: python scripts/gpu_probe_nonsym.py [B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scipy.special import logsumexp, wrightomega
from cvxpygen_amd import families, codegen
from cvxpygen_amd.conic_plan import build_conic_plan
from cvxpygen_amd.conic_runtime import ConicBatchSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rs = np.random.RandomState(0)


def run(d, pv, check, lib=None, plan=None, **kw):
    bs = ConicBatchSolver(d, lib_path=lib, plan=plan)
    bs.solve({k: v[:64] for k, v in pv.items()}, **kw)
    best = 1e30
    for _ in range(3):
        r = bs.solve(pv, **kw)
        best = min(best, r.kernel_ms)
    it = r.iter
    print(f'{d.name:16s} {"lib" if lib else "generic":8s} {kw} B={B} kernel {best:8.3f} ms  {B / best / 1e3:8.3f} M inst/s  '
          f'iter mean {it.mean():.2f} max {it.max()}  solved {(r.status == 1).mean():.4f}  err {check(r):.2e}', flush=True)
    bs.close()


c = 1.5 * rs.randn(B, 4)
d = families.softmax_entropy(4)
run(d, {'c': c}, lambda r: np.abs(r.obj_val + logsumexp(-c, axis=1)).max())
run(d, {'c': c}, lambda r: np.abs(r.obj_val + logsumexp(-c, axis=1)).max(), min_switch_step_length=1.1)
p, b = 0.5 + rs.rand(B, 2), 1.0 + rs.rand(B)
d = families.cobb_douglas(0.3)
run(d, {'p': p, 'budget': b}, lambda r: np.abs(r.obj_val - (0.3 * b / p[:, 0]) ** 0.3 * (0.7 * b / p[:, 1]) ** 0.7).max())
a = rs.randn(B, 3)
d = families.exp_prox(3)
xs = a - wrightomega(a).real
val = 2.0 * (np.exp(xs).sum(axis=1) + 0.5 * ((xs - a) ** 2).sum(axis=1)) - (a ** 2).sum(axis=1)
pv = {'a': a, 'ub': 5.0 * np.ones((B, 3))}
run(d, pv, lambda r: np.abs(r.obj_val - val).max())
cp = build_conic_plan(d)
lib = codegen.build_conic_library(cp, os.path.join(ROOT, 'cvxpygen_amd', 'generated', 'exp_prox'), 'exp_prox')
run(d, pv, lambda r: np.abs(r.obj_val - val).max(), lib=lib, plan=cp)
# PSD cones
for p_ in (3, 6):
    G = rs.randn(B, p_, p_); Cm = G + G.transpose(0, 2, 1)
    run(families.min_eig(p_), {'C': Cm}, lambda r: np.abs(r.obj_val - np.linalg.eigvalsh(Cm).min(axis=1)).max())
G = rs.randn(B, 3, 3); Cm = G + G.transpose(0, 2, 1)
d = families.psd_projection(3)
w_, V_ = np.linalg.eigh(Cm)
X_ = np.einsum('bij,bj,bkj->bik', V_, np.maximum(w_, 0.0), V_)
val3 = ((X_ - Cm) ** 2).sum(axis=(1, 2)) - (Cm ** 2).sum(axis=(1, 2))
run(d, {'C': Cm}, lambda r: np.abs(r.obj_val - val3).max())
cp = build_conic_plan(d)
lib = codegen.build_conic_library(cp, os.path.join(ROOT, 'cvxpygen_amd', 'generated', 'psd_projection'), 'psd_projection')
run(d, {'C': Cm}, lambda r: np.abs(r.obj_val - val3).max(), lib=lib, plan=cp)
run(families.trace_sdp(3), {'C': Cm}, lambda r: np.abs(r.obj_val - np.linalg.eigvalsh(Cm).min(axis=1)).max())
