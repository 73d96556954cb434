"""
TEST INFRASTRUCTURE ONLY -- never imported by the product package (cvxpygen_amd/).

Dense numpy restatement of the embedded-OSQP solve that cvxpygen's generated `cpg_solve()` runs
(`cvxpygen/utils.py:1008-1052` -> `osqp_update_data_vec/_mat` -> `osqp_solve`,
`cvxpygen/solvers/osqp.py:20-62`).  The OSQP sources are a third-party dependency that is absent
from /root/reference (PyPI `osqp >= 1.0.0b3`, `pyproject.toml:26`; submodule dir empty), so the
algorithm is restated from the published description (Stellato et al., "OSQP: an operator
splitting solver for quadratic programs", Math. Prog. Comp. 12, 2020: Algorithm 1, sections 3.4, 4,
5.1, 5.2) and the OSQP 1.0 default settings; SURVEY.md Appendix A lists the restated steps.

PARITY UNPINNED: the reference holds no golden vectors for this path;
this restatement is pinned only by independent mathematics (exact NNLS / BVLS answers and KKT
residuals, tests/golden/) and by agreement with the separately written C oracle
(oracle/osqp_oracle.c).

This file deliberately shares no code with cvxpygen_amd/: scaling, KKT solve (dense
numpy.linalg.solve instead of a sparse LDL'), iteration and termination are all re-derived here.
"""

import numpy as np

INFTY = 1e30
MIN_SCALING, MAX_SCALING = 1e-4, 1e4
RHO_MIN, RHO_MAX, RHO_TOL, RHO_EQ = 1e-6, 1e6, 1e-4, 1e3
DIV_TOL = 1.0 / INFTY

STATUS = {1: 'solved', 2: 'solved inaccurate', 3: 'primal infeasible',
          4: 'primal infeasible inaccurate', 5: 'dual infeasible', 6: 'dual infeasible inaccurate',
          7: 'maximum iterations reached', 9: 'problem non convex', 11: 'unsolved'}

DEFAULTS = dict(rho=0.1, sigma=1e-6, alpha=1.6, scaling=10, max_iter=4000, eps_abs=1e-3,
                eps_rel=1e-3, eps_prim_inf=1e-4, eps_dual_inf=1e-4, scaled_termination=0,
                check_termination=25, adaptive_rho=1, adaptive_rho_interval=50,
                adaptive_rho_tolerance=5.0, check_dualgap=1)     # OSQP >= 1.0 library defaults


def _limit(v):
    v = np.where(v < MIN_SCALING, 1.0, v)
    return np.where(v > MAX_SCALING, MAX_SCALING, v)


def ruiz(Pfull, q, A, iters):
    n, m = Pfull.shape[0], A.shape[0]
    P, q, A = Pfull.copy(), q.copy(), A.copy()
    D, E, c = np.ones(n), np.ones(m), 1.0
    for _ in range(iters):
        dn = np.maximum(np.abs(P).max(axis=0) if n else 0, np.abs(A).max(axis=0) if m else 0)
        en = np.abs(A).max(axis=1) if m else np.zeros(0)
        dt, et = 1 / np.sqrt(_limit(dn)), 1 / np.sqrt(_limit(en))
        P = dt[:, None] * P * dt[None, :]
        A = et[:, None] * A * dt[None, :]
        q = dt * q
        D, E = D * dt, E * et
        cn = np.abs(P).max(axis=0).mean()
        qn = _limit(np.array([np.abs(q).max()]))[0]
        ct = 1.0 / _limit(np.array([max(cn, qn)]))[0]
        P, q, c = P * ct, q * ct, c * ct
    return P, q, A, D, E, c


class DenseOSQP:
    """setup at (P, q, A, l, u); then update_vec / update_mat / solve (cold start)."""

    def __init__(self, P_upper, q, A, l, u, **settings):
        self.s = dict(DEFAULTS)
        self.s.update(settings)
        Pu = np.asarray(P_upper, dtype=float)
        self.P0 = np.triu(Pu) + np.triu(Pu, 1).T
        self.q0 = np.asarray(q, dtype=float).copy()
        self.A0 = np.asarray(A, dtype=float).copy()
        self.l0 = np.maximum(np.asarray(l, dtype=float), -INFTY)
        self.u0 = np.minimum(np.asarray(u, dtype=float), INFTY)
        self.n, self.m = self.P0.shape[0], self.A0.shape[0]
        self._scale()
        self.rho = min(max(self.s['rho'], RHO_MIN), RHO_MAX)
        self._rho_vec()

    def _scale(self):
        if self.s['scaling']:
            self.P, self.q, self.A, self.D, self.E, self.c = ruiz(self.P0, self.q0, self.A0,
                                                                 int(self.s['scaling']))
        else:
            self.P, self.q, self.A = self.P0.copy(), self.q0.copy(), self.A0.copy()
            self.D, self.E, self.c = np.ones(self.n), np.ones(self.m), 1.0
        self.l, self.u = self.E * self.l0, self.E * self.u0

    def _rho_vec(self):
        l, u = self.l, self.u
        unc = (l < -INFTY * MIN_SCALING) & (u > INFTY * MIN_SCALING)
        eq = ~unc & (u - l < RHO_TOL)
        self.ctype = np.where(unc, -1, np.where(eq, 1, 0))
        self.rho_vec = np.where(unc, RHO_MIN, np.where(eq, RHO_EQ * self.rho, self.rho))
        self._K = None

    def _kkt(self):
        if self._K is None:
            n, m = self.n, self.m
            K = np.zeros((n + m, n + m))
            K[:n, :n] = self.P + self.s['sigma'] * np.eye(n)
            K[:n, n:] = self.A.T
            K[n:, :n] = self.A
            K[n:, n:] = -np.diag(1.0 / self.rho_vec)
            self._K = np.linalg.inv(K)
        return self._K

    def update_vec(self, q=None, l=None, u=None):
        if q is not None:
            self.q0 = np.asarray(q, dtype=float).copy()
            self.q = self.c * self.D * self.q0
        if l is not None:
            self.l0 = np.asarray(l, dtype=float).copy()
            self.l = self.E * self.l0
        if u is not None:
            self.u0 = np.asarray(u, dtype=float).copy()
            self.u = self.E * self.u0
        if l is not None or u is not None:
            old = self.ctype.copy()
            self._rho_vec_keep(old)

    def _rho_vec_keep(self, old):
        K = self._K
        self._rho_vec()
        if np.array_equal(old, self.ctype):
            self._K = K

    def update_mat(self, P_upper=None, A=None):
        """unscale -> replace values -> re-equilibrate from scratch -> refactor."""
        if P_upper is not None:
            Pu = np.asarray(P_upper, dtype=float)
            self.P0 = np.triu(Pu) + np.triu(Pu, 1).T
        if A is not None:
            self.A0 = np.asarray(A, dtype=float).copy()
        self._scale()
        self._rho_vec()

    # -----------------------------------------------------------------------------------------
    def solve(self, x0=None, y0=None):
        s, n, m = self.s, self.n, self.m
        P, q, A, l, u = self.P, self.q, self.A, self.l, self.u
        D, E, c = self.D, self.E, self.c
        sig, alpha = s['sigma'], s['alpha']
        x, z, y = np.zeros(n), np.zeros(m), np.zeros(m)
        if x0 is not None:
            x = np.asarray(x0) / D
            z = A @ x
            y = c * np.asarray(y0) / E
        unscaled = bool(s['scaling']) and not s['scaled_termination']
        status, it = 11, 0
        info = {}
        for it in range(1, int(s['max_iter']) + 1):
            Kinv = self._kkt()
            rv, ri = self.rho_vec, 1.0 / self.rho_vec
            xp, zp = x, z
            rhs = np.concatenate([sig * xp - q, zp - ri * y])
            sol = Kinv @ rhs
            xt = sol[:n]
            zt = rhs[n:] + ri * sol[n:]
            x = alpha * xt + (1 - alpha) * xp
            dx = x - xp
            zz = alpha * zt + (1 - alpha) * zp
            z = np.minimum(np.maximum(zz + ri * y, l), u)
            dy = rv * (zz - z)
            y = y + dy
            chk = s['check_termination'] and it % int(s['check_termination']) == 0
            have_info = False
            if chk:
                info = self._info(x, z, y, unscaled)
                have_info = True
                status = self._check(info, dx, dy, unscaled, 1.0)
                if status != 11:
                    break
            if s['adaptive_rho'] and s['adaptive_rho_interval'] and \
                    it % int(s['adaptive_rho_interval']) == 0:
                if not have_info:
                    info = self._info(x, z, y, unscaled)
                self._adapt_rho(info)
        else:
            info = self._info(x, z, y, unscaled)
            status = self._check(info, dx, dy, unscaled, 10.0)
            if status == 11:
                status = 7
            elif status in (1, 3, 5):
                status += 1
        out = dict(iter=it, status=status, status_str=STATUS[status],
                   prim_res=info['prim_res'], dual_res=info['dual_res'],
                   rho=self.rho)
        if status in (1, 2, 7):
            out['x'] = D * x
            out['y'] = E * y / c
            out['obj_val'] = info['obj']
        else:
            out['x'] = np.full(n, np.nan)
            out['y'] = np.full(m, np.nan)
            out['obj_val'] = INFTY if status in (3, 4) else (-INFTY if status in (5, 6) else np.nan)
        return out

    def _info(self, x, z, y, unscaled):
        P, q, A = self.P, self.q, self.A
        Ax, Px, Aty = A @ x, P @ x, A.T @ y
        rp, rd = Ax - z, q + Px + Aty
        ninf = lambda v: np.abs(v).max() if v.size else 0.0
        info = dict(Ax=Ax, Px=Px, Aty=Aty, z=z, x=x, y=y,
                    sc_prim=ninf(rp), sc_dual=ninf(rd))
        if unscaled:
            Ei, Di, ci = 1 / self.E, 1 / self.D, 1 / self.c
            info['prim_res'] = ninf(Ei * rp)
            info['dual_res'] = ci * ninf(Di * rd)
            info['prim_nrm'] = max(ninf(Ei * z), ninf(Ei * Ax))
            info['dual_nrm'] = ci * max(ninf(Di * q), ninf(Di * Aty), ninf(Di * Px))
        else:
            info['prim_res'], info['dual_res'] = info['sc_prim'], info['sc_dual']
            info['prim_nrm'] = max(ninf(z), ninf(Ax))
            info['dual_nrm'] = max(ninf(q), ninf(Aty), ninf(Px))
        quad, lin = x @ Px, q @ x
        info['obj'] = (0.5 * quad + lin) / self.c
        if self.s['check_dualgap']:
            fin_u = self.u < INFTY * MIN_SCALING
            fin_l = self.l > -INFTY * MIN_SCALING
            sup = np.sum(np.where(fin_u, self.u, 0.0) * np.maximum(y, 0)) + \
                np.sum(np.where(fin_l, self.l, 0.0) * np.minimum(y, 0))
            info['dual_obj'] = (-0.5 * quad - sup) / self.c
            info['gap'] = abs(quad + lin + sup) / self.c
        return info

    def _check(self, info, dx, dy, unscaled, mult):
        s = self.s
        ea, er = s['eps_abs'] * mult, s['eps_rel'] * mult
        epi, edi = s['eps_prim_inf'] * mult, s['eps_dual_inf'] * mult
        if info['prim_res'] > INFTY or info['dual_res'] > INFTY:
            return 9
        pc = self.m == 0 or info['prim_res'] < ea + er * info['prim_nrm']
        dc = info['dual_res'] < ea + er * info['dual_nrm']
        gc = True
        if s['check_dualgap']:
            gc = info['gap'] < ea + er * max(abs(info['obj']), abs(info['dual_obj']))
        if pc and dc and gc:
            return 1
        if not pc and self._prim_inf(dy, unscaled, epi):
            return 3
        if not dc and self._dual_inf(dx, unscaled, edi):
            return 5
        return 11

    def _prim_inf(self, dy, unscaled, eps):
        l, u = self.l, self.u
        dy = dy.copy()
        iu, il = u > INFTY * MIN_SCALING, l < -INFTY * MIN_SCALING
        dy[iu & il] = 0.0
        dy[iu & ~il] = np.minimum(dy[iu & ~il], 0.0)
        dy[il & ~iu] = np.maximum(dy[il & ~iu], 0.0)
        nrm = np.abs(self.E * dy).max() if unscaled else np.abs(dy).max()
        if nrm > DIV_TOL:
            lhs = u @ np.maximum(dy, 0) + l @ np.minimum(dy, 0)
            if lhs < eps * nrm:
                Atdy = self.A.T @ dy
                if unscaled:
                    Atdy = Atdy / self.D
                return np.abs(Atdy).max() < eps * nrm
        return False

    def _dual_inf(self, dx, unscaled, eps):
        if unscaled:
            nrm, cs = np.abs(self.D * dx).max(), self.c
        else:
            nrm, cs = np.abs(dx).max(), 1.0
        if nrm > DIV_TOL:
            if self.q @ dx < -cs * eps * nrm:
                Pdx = self.P @ dx
                if unscaled:
                    Pdx = Pdx / self.D
                if np.abs(Pdx).max() < cs * eps * nrm:
                    Adx = self.A @ dx
                    if unscaled:
                        Adx = Adx / self.E
                    bad = ((self.u < INFTY * MIN_SCALING) & (Adx > eps * nrm)) | \
                          ((self.l > -INFTY * MIN_SCALING) & (Adx < -eps * nrm))
                    return not bad.any()
        return False

    def _adapt_rho(self, info):
        ninf = lambda v: np.abs(v).max() if v.size else 0.0
        pr = info['sc_prim'] / (max(ninf(info['z']), ninf(info['Ax'])) + DIV_TOL)
        dr = info['sc_dual'] / (max(ninf(self.q), ninf(info['Aty']), ninf(info['Px'])) + DIV_TOL)
        new = self.rho * np.sqrt(pr / dr)
        new = min(max(new, RHO_MIN), RHO_MAX)
        tol = self.s['adaptive_rho_tolerance']
        if new > self.rho * tol or new < self.rho / tol:
            self.rho = new
            self.rho_vec = np.where(self.ctype == -1, RHO_MIN,
                                    np.where(self.ctype == 1, RHO_EQ * new, new))
            self._K = None
