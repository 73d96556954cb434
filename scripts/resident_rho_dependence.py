"""How many of the resident kernel's substitution coefficients change when rho changes (round-5 review, item 3: "on a rho change re-extract
only coefficients that depend on rho -- measure how many of the 8 549 actually change").  CPU only: the plan's own replay of the numeric
factorisation at two values of rho on config 3's family, coefficient vectors compared entry by entry."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvxpygen_amd import families, resident_plan as rs
from cvxpygen_amd.runtime import build_family_plan

d = families.portfolio(100, 10)
plan = build_family_plan(d)
pl = rs.build_resident_plan(d.P, d.A, plan.osqp)
b = pl.base
cn = d.canon_at(d.theta0)
Ps, As = np.asarray(cn['P'], float), np.asarray(cn['A'], float)
out = {}
for rho in (0.1, 0.137):
    rho_inv = 1.0 / np.where(np.arange(b.m) < d.n_eq, 1e3 * rho, rho)
    fac = rs.replay_factor(pl, Ps, As, 1e-6, rho_inv)
    out[rho] = (fac.copy(), rs.replay_solve_vals(pl, fac))
(f0, v0), (f1, v1) = out[0.1], out[0.137]
nnzL, N = b.nnzL, b.n + b.m
chg = lambda a, c: int((np.abs(a - c) > 1e-14 * np.maximum(1.0, np.abs(a))).sum())
print(f'config 3 family: N = {N}, nnz(L) = {nnzL}, block-inverse entries {pl.nnzX}, substitution coefficients {len(v0)}')
print(f'factor entries that change with rho:   L {chg(f0[:nnzL], f1[:nnzL])} of {nnzL}, 1/d {chg(f0[nnzL:nnzL + N], f1[nnzL:nnzL + N])} of {N}, '
      f'X {chg(f0[nnzL + N:nnzL + N + pl.nnzX], f1[nnzL + N:nnzL + N + pl.nnzX])} of {pl.nnzX}')
k = pl.sol_kind
for name, code in (('-l_ij', rs.SRC_NEG_L), ('1/d_i', rs.SRC_DINV), ('X_ij', rs.SRC_X), ('1', rs.SRC_ONE)):
    s = k == code
    print(f'substitution coefficients {name:6s}: {chg(v0[s], v1[s])} of {int(s.sum())} change')
print(f'all: {chg(v0, v1)} of {len(v0)} change ({100.0 * chg(v0, v1) / len(v0):.1f} %)')
