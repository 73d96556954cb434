"""
TEST INFRASTRUCTURE ONLY -- dense numpy restatement of the interior-point method behind the
reference's Clarabel path.  Imported by tests/ and
__graft_entry__.smoke(); never by cvxpygen_amd/.

PARITY UNPINNED.  The arithmetic of this path lives in a third-party dependency that is absent
from /root/reference: submodule `cvxpygen/solvers/Clarabel.cpp` (github.com/oxfordcontrol/
Clarabel.cpp wrapping Clarabel.rs; PyPI `clarabel >= 0.6.0`, pyproject.toml:28; commit unpinned,
directory empty, no Rust toolchain here).  What the reference itself fixes, and what this file
follows literally:
  * the call sequence per solve: copy canonical parameters, build a NEW solver object
    (equilibration + KKT assembly + factorisation from scratch), solve, read the solution
    (`cvxpygen/solvers/clarabel.py:172-204`),
  * the problem form  minimise 1/2 x'Px + q'x  s.t.  Ax + s = b, s in K  with cones ordered zero,
    nonnegative, second-order (`cvxpygen/solvers/clarabel.py:133-155, 308-323`),
  * every setting default (`cvxpygen/solvers/clarabel.py:63-119`),
  * the returned fields x, z, obj_val, iterations, status (integer), r_prim, r_dual
    (`cvxpygen/solvers/clarabel.py:37-46`).
The algorithm is restated from its published description (Goulart & Chen, "Clarabel: an
interior-point solver for conic programs with quadratic objectives", 2024):
homogeneous embedding with (tau, kappa), Ruiz equilibration, Nesterov-Todd scaling, quasi-definite
KKT system with static / dynamic regularisation and iterative refinement, Mehrotra
predictor-corrector with sigma = (1 - alpha)^3.  Where the paper leaves details open (order of the
equilibration clamps, which norms carry the cost scaling) the choice made is written next to the
code.  The only numbers this file is pinned against are independent ones: tests/golden holds
scipy solutions of the reference's ADP inputs (tests/test_E2E_SOCP.py:15-64).

Independent of the product: dense matrices, natural KKT ordering [x; z], its own LDL'.
"""

import numpy as np

DEFAULTS = dict(    # cvxpygen/solvers/clarabel.py:63-119
    max_iter=200, max_step_fraction=0.99,
    tol_gap_abs=1e-8, tol_gap_rel=1e-8, tol_feas=1e-8, tol_infeas_abs=1e-8, tol_infeas_rel=1e-8,
    tol_ktratio=1e-6,
    reduced_tol_gap_abs=5e-5, reduced_tol_gap_rel=5e-5, reduced_tol_feas=1e-4,
    reduced_tol_infeas_abs=5e-5, reduced_tol_infeas_rel=5e-5, reduced_tol_ktratio=1e-4,
    equilibrate_enable=1, equilibrate_max_iter=10, equilibrate_min_scaling=1e-4,
    equilibrate_max_scaling=1e4,
    linesearch_backtrack_step=0.8, min_switch_step_length=1e-1, min_terminate_step_length=1e-4,
    static_regularization_enable=1, static_regularization_constant=1e-8,
    static_regularization_proportional=2.2e-16,     # the reference's value (clarabel.py:104), not the solver's own
    dynamic_regularization_enable=1, dynamic_regularization_eps=1e-13,
    dynamic_regularization_delta=2e-7,
    iterative_refinement_enable=1, iterative_refinement_reltol=1e-13,
    iterative_refinement_abstol=1e-12, iterative_refinement_max_iter=10,
    iterative_refinement_stop_ratio=5.0)

# status integers (Clarabel SolverStatus)
UNSOLVED, SOLVED, PRIMAL_INFEASIBLE, DUAL_INFEASIBLE, ALMOST_SOLVED, ALMOST_PRIMAL_INFEASIBLE, \
    ALMOST_DUAL_INFEASIBLE, MAX_ITERATIONS, MAX_TIME, NUMERICAL_ERROR, INSUFFICIENT_PROGRESS = range(11)


class Cones:
    """rows of s / z: [zero | nonneg | soc_1 | soc_2 | ...]"""

    def __init__(self, zero, nonneg, soc):
        self.zero, self.nonneg, self.soc = int(zero), int(nonneg), [int(d) for d in soc]
        self.m = self.zero + self.nonneg + sum(self.soc)
        self.soc_start = []
        o = self.zero + self.nonneg
        for d in self.soc:
            self.soc_start.append(o)
            o += d
        self.degree = self.nonneg + len(self.soc)
        self.nn = slice(self.zero, self.zero + self.nonneg)

    def socs(self):
        return [slice(a, a + d) for a, d in zip(self.soc_start, self.soc)]


def _soc_res(v):
    return v[0] * v[0] - float(v[1:] @ v[1:])


def _ldl(K, signs, stg):
    """dense LDL' in natural order with dynamic regularisation of wrong-signed / tiny pivots"""
    N = K.shape[0]
    L = np.eye(N)
    d = np.zeros(N)
    Kw = K.copy()
    for k in range(N):
        dk = Kw[k, k]
        if stg['dynamic_regularization_enable'] and dk * signs[k] < stg['dynamic_regularization_eps']:
            dk = stg['dynamic_regularization_delta'] * signs[k]
        d[k] = dk
        L[k + 1:, k] = Kw[k + 1:, k] / dk
        Kw[k + 1:, k + 1:] -= np.outer(L[k + 1:, k], L[k + 1:, k]) * dk
    return L, d


def _ldl_solve(L, d, b):
    N = len(b)
    y = b.copy()
    for k in range(N):
        y[k + 1:] -= L[k + 1:, k] * y[k]
    y /= d
    for k in range(N - 1, -1, -1):
        y[k] -= L[k + 1:, k] @ y[k + 1:]
    return y


class _Kkt:
    """K = [[P, A'], [A, -Hs]] with the regularised factor and refinement against the true K"""

    def __init__(self, P, A, cones, stg):
        self.P, self.A, self.cones, self.stg = P, A, cones, stg
        self.n, self.m = P.shape[0], A.shape[0]
        self.signs = np.concatenate([np.ones(self.n), -np.ones(self.m)])

    def update(self, Hs):
        n, m, stg = self.n, self.m, self.stg
        K = np.zeros((n + m, n + m))
        K[:n, :n] = self.P
        K[n:, :n] = self.A
        K[:n, n:] = self.A.T
        K[n:, n:] = -Hs
        self.K = K
        Kr = K.copy()
        if stg['static_regularization_enable']:
            eps = stg['static_regularization_constant'] + \
                stg['static_regularization_proportional'] * np.abs(np.diag(K)).max()
            Kr[np.diag_indices(n + m)] += eps * self.signs
        self.L, self.d = _ldl(Kr, self.signs, stg)

    def solve(self, bx, bz):
        stg = self.stg
        b = np.concatenate([bx, bz])
        x = _ldl_solve(self.L, self.d, b)
        if stg['iterative_refinement_enable']:
            normb = np.abs(b).max() if b.size else 0.0
            e = b - self.K @ x
            norme = np.abs(e).max()
            for _ in range(int(stg['iterative_refinement_max_iter'])):
                if norme <= stg['iterative_refinement_abstol'] + stg['iterative_refinement_reltol'] * normb:
                    break
                lastnorme = norme
                xn = x + _ldl_solve(self.L, self.d, e)
                en = b - self.K @ xn
                norme = np.abs(en).max()
                ratio = lastnorme / norme if norme > 0 else np.inf
                if ratio < stg['iterative_refinement_stop_ratio']:
                    if ratio > 1.0:
                        x, e = xn, en
                    else:
                        norme = lastnorme
                    break
                x, e = xn, en
        return x[:self.n], x[self.n:]


def equilibrate(P, q, A, b, cones, stg):
    """Ruiz equilibration of [[P, A'], [A, 0]] with cost scaling.  Choices: per-pass norms are clamped
    to [min, max] scaling (a zero norm counts as 1) before the inverse square root; the second-order
    cone rows are made uniform (mean of the cone) once, after the passes."""
    n, m = P.shape[0], A.shape[0]
    D, E, c = np.ones(n), np.ones(m), 1.0
    P, q, A, b = P.copy(), q.copy(), A.copy(), b.copy()
    if not stg['equilibrate_enable']:
        return P, q, A, b, D, E, c
    lo, hi = stg['equilibrate_min_scaling'], stg['equilibrate_max_scaling']

    def lim(v):
        v = np.where(v == 0.0, 1.0, v)
        return np.clip(v, lo, hi)
    for _ in range(int(stg['equilibrate_max_iter'])):
        dn = np.zeros(n)
        if n:
            dn = np.abs(P).max(axis=0)
            if m:
                dn = np.maximum(dn, np.abs(A).max(axis=0))
        en = np.abs(A).max(axis=1) if (m and n) else np.zeros(m)
        dw, ew = 1.0 / np.sqrt(lim(dn)), 1.0 / np.sqrt(lim(en))
        P = dw[:, None] * P * dw[None, :]
        A = ew[:, None] * A * dw[None, :]
        q = dw * q
        b = ew * b
        D *= dw
        E *= ew
        pn = np.abs(P).max(axis=0).mean() if n else 0.0
        qn = np.abs(q).max() if n else 0.0
        if pn != 0.0 and qn != 0.0:
            ct = float(np.clip(1.0 / max(pn, qn), lo, hi))
            P *= ct
            q *= ct
            c *= ct
    for sl in cones.socs():
        ew = E[sl].mean() / E[sl]
        A[sl, :] *= ew[:, None]
        b[sl] *= ew
        E[sl] *= ew
    return P, q, A, b, D, E, c


def _margins(cones, v):
    """(min margin, sum of positive margins) over the nonnegative and second-order cones"""
    alpha, beta = np.inf, 0.0
    if cones.nonneg:
        alpha = min(alpha, v[cones.nn].min())
        beta += np.maximum(v[cones.nn], 0.0).sum()
    for sl in cones.socs():
        a = v[sl][0] - np.sqrt(v[sl][1:] @ v[sl][1:])
        alpha = min(alpha, a)
        beta += max(0.0, a)
    return alpha, beta


def _unit_shift(cones, v, a, primal):
    if cones.nonneg:
        v[cones.nn] += a
    for sl in cones.socs():
        v[sl.start] += a
    if primal and cones.zero:
        v[:cones.zero] = 0.0


def _shift_to_cone(cones, v, primal):
    if cones.degree == 0:
        _unit_shift(cones, v, 0.0, primal)
        return
    mn, pos = _margins(cones, v)
    target = max(1.0, 0.1 * pos / cones.degree)
    if mn <= 0.0:
        _unit_shift(cones, v, -mn, primal)
        _unit_shift(cones, v, target, primal)
    elif mn < target:
        _unit_shift(cones, v, target - mn, primal)
    else:
        _unit_shift(cones, v, 0.0, primal)


class _Scaling:
    """Nesterov-Todd scaling: nonneg w = sqrt(s/z); SOC W = eta [[w0, w1'], [w1, I + w1 w1'/(1+w0)]]"""

    def __init__(self, cones):
        self.c = cones
        self.identity()

    def identity(self):
        c = self.c
        self.w = np.ones(c.m)
        self.lam = np.ones(c.m)
        self.eta = [1.0] * len(c.soc)
        self.sw = []
        for d in c.soc:
            v = np.zeros(d)
            v[0] = 1.0
            self.sw.append(v)

    def update(self, s, z):
        c = self.c
        ok = True
        if c.nonneg:
            self.w[c.nn] = np.sqrt(s[c.nn] / z[c.nn])
            self.lam[c.nn] = np.sqrt(s[c.nn] * z[c.nn])
        for k, sl in enumerate(c.socs()):
            sk, zk = s[sl], z[sl]
            rs, rz = _soc_res(sk), _soc_res(zk)
            if not (rs > 0.0 and rz > 0.0):
                ok = False
                continue
            ss, zs = np.sqrt(rs), np.sqrt(rz)
            gamma = np.sqrt(0.5 * (1.0 + float(sk @ zk) / (ss * zs)))
            w = sk / (2.0 * ss * gamma)
            w[0] += zk[0] / (2.0 * zs * gamma)
            w[1:] -= zk[1:] / (2.0 * zs * gamma)
            w[0] = np.sqrt(1.0 + float(w[1:] @ w[1:]))
            self.sw[k] = w
            self.eta[k] = np.sqrt(ss / zs)
            self.lam[sl] = self.mul_W_cone(k, zk)
        return ok

    def mul_W_cone(self, k, v, inv=False):
        w, eta = self.sw[k], self.eta[k]
        out = np.empty_like(v)
        if not inv:
            zeta = float(w[1:] @ v[1:])
            out[0] = w[0] * v[0] + zeta
            out[1:] = v[1:] + (v[0] + zeta / (1.0 + w[0])) * w[1:]
            return eta * out
        zeta = float(w[1:] @ v[1:])
        out[0] = w[0] * v[0] - zeta
        out[1:] = v[1:] + (-v[0] + zeta / (1.0 + w[0])) * w[1:]
        return out / eta

    def Hs(self):
        """dense W'W (zero rows for the zero cone)"""
        c = self.c
        H = np.zeros((c.m, c.m))
        if c.nonneg:
            i = np.arange(c.zero, c.zero + c.nonneg)
            H[i, i] = self.w[c.nn] ** 2
        for k, sl in enumerate(c.socs()):
            w, eta = self.sw[k], self.eta[k]
            J = -np.eye(len(w))
            J[0, 0] = 1.0
            H[sl, sl] = eta * eta * (2.0 * np.outer(w, w) - J)
        return H

    def mul_Hs(self, v):
        c = self.c
        out = np.zeros(c.m)
        if c.nonneg:
            out[c.nn] = self.w[c.nn] ** 2 * v[c.nn]
        for k, sl in enumerate(c.socs()):
            w, eta = self.sw[k], self.eta[k]
            vk = v[sl]
            t = 2.0 * float(w @ vk)
            o = t * w
            o[0] -= vk[0]
            o[1:] += vk[1:]
            out[sl] = eta * eta * o
        return out

    def mul_W(self, v, inv=False):
        c = self.c
        out = np.zeros(c.m)
        if c.nonneg:
            out[c.nn] = v[c.nn] / self.w[c.nn] if inv else v[c.nn] * self.w[c.nn]
        for k, sl in enumerate(c.socs()):
            out[sl] = self.mul_W_cone(k, v[sl], inv)
        return out

    def circ(self, a, b):
        c = self.c
        out = np.zeros(c.m)
        out[c.nn] = a[c.nn] * b[c.nn]
        for sl in c.socs():
            out[sl.start] = float(a[sl] @ b[sl])
            out[sl.start + 1:sl.stop] = a[sl.start] * b[sl.start + 1:sl.stop] + b[sl.start] * a[sl.start + 1:sl.stop]
        return out

    def inv_circ_lam(self, d):
        """lambda \\ d"""
        c = self.c
        out = np.zeros(c.m)
        out[c.nn] = d[c.nn] / self.lam[c.nn]
        for sl in c.socs():
            lam, dk = self.lam[sl], d[sl]
            p = _soc_res(lam)
            u0 = (lam[0] * dk[0] - float(lam[1:] @ dk[1:])) / p
            out[sl.start] = u0
            out[sl.start + 1:sl.stop] = (dk[1:] - u0 * lam[1:]) / lam[0]
        return out

    def ds_offset(self, ds):
        """W'(lambda \\ ds)"""
        return self.mul_W(self.inv_circ_lam(ds))            # W is symmetric


def _step_length(cones, v, dv, amax):
    a = amax
    if cones.nonneg:
        vv, dd = v[cones.nn], dv[cones.nn]
        neg = dd < 0.0
        if neg.any():
            a = min(a, float((-vv[neg] / dd[neg]).min()))
    for sl in cones.socs():
        x, y = v[sl], dv[sl]
        qa = _soc_res(y)
        qb = 2.0 * (x[0] * y[0] - float(x[1:] @ y[1:]))
        qc = max(0.0, _soc_res(x))
        disc = qb * qb - 4.0 * qa * qc
        if (qa > 0.0 and qb > 0.0) or disc < 0.0:
            r = np.inf
        elif qa == 0.0:
            r = np.inf
        else:
            t = (-qb - np.sqrt(disc)) if qb >= 0.0 else (-qb + np.sqrt(disc))
            r1 = (2.0 * qc) / t if t != 0.0 else np.inf
            r2 = t / (2.0 * qa)
            r1 = np.inf if r1 < 0.0 else r1
            r2 = np.inf if r2 < 0.0 else r2
            r = min(r1, r2)
        # the cone also needs x0 + a*y0 >= 0
        a = min(a, r)
    return a


def solve(P, q, A, b, cones, p_is_zero=None, **settings):
    """P: dense symmetric (n x n), A: dense (m x n), cones: Cones.  Returns dict with the fields of
    `cvxpygen/solvers/clarabel.py:37-46` (+ s)."""
    stg = dict(DEFAULTS)
    stg.update(settings)
    n, m = P.shape[0], A.shape[0]
    normq = np.abs(q).max() if n else 0.0
    normb = np.abs(b).max() if m else 0.0
    Ph, qh, Ah, bh, D, E, c = equilibrate(np.asarray(P, float), np.asarray(q, float), np.asarray(A, float),
                                          np.asarray(b, float), cones, stg)
    Dinv, Einv = 1.0 / D, 1.0 / E
    kkt = _Kkt(Ph, Ah, cones, stg)
    sc = _Scaling(cones)

    # ---- initial point (symmetric cones, P != 0 or == 0 handled alike via the two-solve form)
    Hs0 = np.zeros((m, m))
    idx = np.arange(cones.zero, m)
    Hs0[idx, idx] = 1.0
    kkt.update(Hs0)
    if p_is_zero is None:
        p_is_zero = not np.any(Ph != 0.0)
    if not p_is_zero:      # the structural test nnz(P) == 0 of the solver
        x, z = kkt.solve(-qh, bh)
        s = -z.copy()
    else:
        x, s = kkt.solve(np.zeros(n), bh)
        s = -s
        _, z = kkt.solve(-qh, np.zeros(m))
    _shift_to_cone(cones, s, True)
    _shift_to_cone(cones, z, False)
    tau, kap = 1.0, 1.0

    status, it = UNSOLVED, 0
    info = {}
    prev = dict(res_p=np.inf, res_d=np.inf, gap_abs=np.inf, gap_rel=np.inf, cost_p=np.inf, cost_d=np.inf)
    prev_iterate = (x, z, s, tau, kap)
    while True:
        # ---- residuals
        Px = Ph @ x
        rx_inf = -(Ah.T @ z)
        rz_inf = Ah @ x + s
        dot_qx, dot_bz, dot_sz, xPx = float(qh @ x), float(bh @ z), float(s @ z), float(x @ Px)
        rx = rx_inf - Px - qh * tau
        rz = rz_inf - bh * tau
        rtau = dot_qx + dot_bz + kap + xPx / tau
        mu = (dot_sz + tau * kap) / (cones.degree + 1)
        # ---- info (unscaled)
        tinv = 1.0 / tau
        cinv = 1.0 / c
        cost_p = (dot_qx * tinv + 0.5 * xPx * tinv * tinv) * cinv
        cost_d = (-dot_bz * tinv - 0.5 * xPx * tinv * tinv) * cinv
        ninf = lambda v: float(np.abs(v).max()) if v.size else 0.0
        normx, normz, norms = ninf(D * x), ninf(E * z) * cinv, ninf(Einv * s)
        res_pinf = ninf(Dinv * rx_inf) / max(1.0, normz)
        res_dinf = max(ninf(Dinv * Px) / max(1.0, normx), ninf(Einv * rz_inf) / max(1.0, normx + norms))
        normx *= tinv
        normz *= tinv
        norms *= tinv
        res_p = ninf(Einv * rz) * tinv / max(1.0, normb + normx + norms)
        res_d = ninf(Dinv * rx) * tinv * cinv / max(1.0, normq + normx + normz)
        gap_abs = abs(cost_p - cost_d)
        gap_rel = gap_abs / max(1.0, min(abs(cost_p), abs(cost_d)))
        ktratio = kap / tau
        info = dict(cost_p=cost_p, cost_d=cost_d, res_p=res_p, res_d=res_d, gap_abs=gap_abs,
                    gap_rel=gap_rel, ktratio=ktratio, res_pinf=res_pinf, res_dinf=res_dinf,
                    dot_bz=dot_bz * cinv, dot_qx=dot_qx * cinv)

        def converged(pre):
            # check_convergence: optimality at kappa/tau <= 1, infeasibility certificates once kappa/tau has passed
            # 1000 / tol_ktratio (the published solver hard-codes the factor 1000 next to the setting)
            g = lambda k: stg[pre + k]
            if ktratio <= 1.0 and ((info['gap_abs'] < g('tol_gap_abs')) or (info['gap_rel'] < g('tol_gap_rel'))) \
                    and info['res_p'] < g('tol_feas') and info['res_d'] < g('tol_feas'):
                return ALMOST_SOLVED if pre else SOLVED
            if ktratio > 1000.0 / g('tol_ktratio'):
                if info['dot_bz'] < -g('tol_infeas_abs') and res_pinf < -g('tol_infeas_rel') * info['dot_bz']:
                    return ALMOST_PRIMAL_INFEASIBLE if pre else PRIMAL_INFEASIBLE
                if info['dot_qx'] < -g('tol_infeas_abs') and res_dinf < -g('tol_infeas_rel') * info['dot_qx']:
                    return ALMOST_DUAL_INFEASIBLE if pre else DUAL_INFEASIBLE
            return UNSOLVED
        status = converged('')
        # poor progress (check_termination of the published solver): the residuals went up ...
        if status == UNSOLVED and it > 1 and (res_d > prev['res_d'] or res_p > prev['res_p']):
            # ... at high accuracy: kappa/tau at round-off level and the previous gap already inside the tolerance
            if ktratio < 100.0 * np.finfo(float).eps and \
                    (prev['gap_abs'] < stg['tol_gap_abs'] or prev['gap_rel'] < stg['tol_gap_rel']):
                status = INSUFFICIENT_PROGRESS
            # ... or by a factor 100, out of the feasibility tolerance
            if (res_d > stg['tol_feas'] and res_d > 100.0 * prev['res_d']) or \
                    (res_p > stg['tol_feas'] and res_p > 100.0 * prev['res_p']):
                status = INSUFFICIENT_PROGRESS
            if status == INSUFFICIENT_PROGRESS:
                # "insufficient progress often involves actual degradation of results": back to the previous iterate
                # and its cost / residual / gap figures (kappa/tau and the certificate quantities stay)
                x, z, s, tau, kap = prev_iterate
                for k_ in ('cost_p', 'cost_d', 'res_p', 'res_d', 'gap_abs', 'gap_rel'):
                    info[k_] = prev[k_]
        if status == UNSOLVED and it >= int(stg['max_iter']):
            status = MAX_ITERATIONS
        if status != UNSOLVED:
            break
        prev = dict(info)
        it += 1
        # ---- scaling, factor, constant part of the solution
        if not sc.update(s, z):
            status = NUMERICAL_ERROR
            break
        kkt.update(sc.Hs())
        x2, z2 = kkt.solve(-qh, bh)

        def kkt_solve(rhs_x, rhs_z, rhs_tau, rhs_kap, ds_const):
            x1, z1 = kkt.solve(rhs_x, ds_const - rhs_z)
            xi = x / tau
            num = rhs_tau - rhs_kap / tau + float(qh @ x1) + float(bh @ z1) + 2.0 * float(xi @ (Ph @ x1))
            xm = xi - x2
            den = kap / tau - float(qh @ x2) - float(bh @ z2) + float(xm @ (Ph @ xm)) - float(x2 @ (Ph @ x2))
            dtau = num / den
            dx = x1 + dtau * x2
            dz = z1 + dtau * z2
            ds = -(sc.mul_Hs(dz) + ds_const)
            dkap = -(rhs_kap + kap * dtau) / tau
            return dx, dz, ds, dtau, dkap

        def step_len(dz, ds, dtau, dkap, combined):
            a = 1.0
            if dtau < 0.0:
                a = min(a, -tau / dtau)
            if dkap < 0.0:
                a = min(a, -kap / dkap)
            a = min(_step_length(cones, z, dz, a), _step_length(cones, s, ds, a))
            return a * stg['max_step_fraction'] if combined else a
        # ---- affine step
        dx, dz, ds, dtau, dkap = kkt_solve(rx, rz, rtau, tau * kap, s)
        alpha = step_len(dz, ds, dtau, dkap, False)
        sigma = (1.0 - alpha) ** 3
        # ---- combined step
        shift = sc.circ(sc.mul_W(ds, inv=True), sc.mul_W(dz))
        e = np.zeros(m)
        e[cones.nn] = 1.0
        for sl in cones.socs():
            e[sl.start] = 1.0
        d_s = sc.circ(sc.lam, sc.lam) + shift - sigma * mu * e
        d_s[:cones.zero] = 0.0
        rk = -sigma * mu + dtau * dkap + tau * kap
        dx, dz, ds, dtau, dkap = kkt_solve((1.0 - sigma) * rx, (1.0 - sigma) * rz, (1.0 - sigma) * rtau, rk,
                                           sc.ds_offset(d_s))
        alpha = step_len(dz, ds, dtau, dkap, True)
        if alpha <= max(0.0, stg['min_terminate_step_length']):      # undersized step: stop where we are
            status = INSUFFICIENT_PROGRESS
            break
        prev_iterate = (x, z, s, tau, kap)
        x = x + alpha * dx
        s = s + alpha * ds
        z = z + alpha * dz
        tau += alpha * dtau
        kap += alpha * dkap

    if status in (NUMERICAL_ERROR, INSUFFICIENT_PROGRESS, MAX_ITERATIONS):
        # post_process of the published solver: after an error or the iteration limit the last figures may still pass
        # the reduced tolerances -> "almost" statuses (cvxpygen/solvers/clarabel.py:79-84 carries their settings)
        almost = converged('reduced_')
        if almost != UNSOLVED:
            status = almost
    if status in (PRIMAL_INFEASIBLE, ALMOST_PRIMAL_INFEASIBLE, DUAL_INFEASIBLE, ALMOST_DUAL_INFEASIBLE):
        scale = 1.0                  # certificates are returned unnormalised by tau
        obj = np.nan
    else:
        scale = 1.0 / tau
        obj = info['cost_p']
    return dict(x=D * x * scale, z=E * z * scale / c, s=Einv * s * scale, obj_val=obj, iterations=it,
                status=status, r_prim=info['res_p'], r_dual=info['res_d'], info=info)


def cpg_solve_batch(desc, theta, **settings):
    """Reference semantics for a batch of one conic family: per instance canonicalise, build a new
    solver, solve, retrieve (cvxpygen/solvers/clarabel.py:172-204; cvxpygen/utils.py:1032-1052).
    theta: (B, NP) or (B, NP + 1)."""
    theta = np.asarray(theta, dtype=np.float64)
    B = theta.shape[0]
    if theta.shape[1] == desc.NP:
        theta = np.concatenate([theta, np.ones((B, 1))], axis=1)
    cones = Cones(desc.cones['zero'], desc.cones['nonneg'], desc.cones['soc'])
    n, m = desc.n_var, desc.m
    Pp, Pi = desc.P.indptr, desc.P.indices
    Ap, Ai = desc.A.indptr, desc.A.indices
    Pc = np.repeat(np.arange(n), np.diff(Pp))
    Ac = np.repeat(np.arange(n), np.diff(Ap))
    out = dict(sol_x=np.zeros((B, n)), sol_z=np.zeros((B, m)), obj_val=np.zeros(B),
               iter=np.zeros(B, dtype=np.int32), status=np.zeros(B, dtype=np.int32), pri_res=np.zeros(B),
               dua_res=np.zeros(B))
    for k in range(B):
        cn = desc.canon_at(theta[k])
        Pd = np.zeros((n, n))
        Pd[Pi, Pc] = cn['P']
        Pd[Pc, Pi] = cn['P']
        Ad = np.zeros((m, n))
        Ad[Ai, Ac] = cn['A']
        r = solve(Pd, cn['q'], Ad, cn['b'], cones, p_is_zero=(desc.P.nnz == 0), **settings)    # maps hold the minimisation form
        d = float(cn['d'][0]) if desc.nonzero_d else 0.0
        obj = r['obj_val'] + d
        out['sol_x'][k], out['sol_z'][k] = r['x'], r['z']
        out['obj_val'][k] = -obj if desc.is_maximization else obj
        out['iter'][k], out['status'][k] = r['iterations'], r['status']
        out['pri_res'][k], out['dua_res'][k] = r['r_prim'], r['r_dual']
    out['prim'] = {v.name: out['sol_x'][:, v.indices] for v in desc.variables}
    out['dual'] = {d.name: out['sol_z'][:, d.indices] for d in desc.duals}
    return out
