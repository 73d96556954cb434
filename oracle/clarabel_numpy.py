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
  * the problem form  minimise 1/2 x'Px + q'x  s.t.  Ax + s = b, s in K  with K a product of zero,
    nonnegative, second-order, PSD (triangle form: upper triangle column by column, off-diagonals times sqrt 2),
    exponential and three-dimensional power cones -- every type of the reference's `cones` array
    (`cvxpygen/solvers/clarabel.py:133-155, 308-323`).  ROW ORDER: zero | nonneg | soc | psd | exp | p3d -- the order
    in which cvxpy stacks the rows of A and b for this solver.  The reference's `cones` array lists the
    exponential cones AHEAD of the second-order cones (clarabel.py:316-319); for a family that has both kinds
    its solver is told cones that do not match the rows it is given.  Followed here: the rows' meaning; with
    only one of the two kinds present -- every case in which the reference is right -- both orders coincide,
  * every setting default (`cvxpygen/solvers/clarabel.py:63-119`),
  * the returned fields x, z, obj_val, iterations, status (integer), r_prim, r_dual
    (`cvxpygen/solvers/clarabel.py:37-46`).
The algorithm is restated from its published description (Goulart & Chen, "Clarabel: an
interior-point solver for conic programs with quadratic objectives", 2024):
homogeneous embedding with (tau, kappa), Ruiz equilibration, Nesterov-Todd scaling, quasi-definite
KKT system with static / dynamic regularisation and iterative refinement, Mehrotra
predictor-corrector with sigma = (1 - alpha)^3; for the exponential and power cones the paper's section on
nonsymmetric cones: unit initialisation, the dual barrier's Hessian as scaling block (primal-dual form: a
rank-three update H_s = s s'/<s,z> + ds ds'/<ds,dz> + t a a', fall-back mu * H*(z)), third-order correction
eta = 1/2 D^3 f*(z)[H*^-1 ds_aff, dz_aff], backtracking step lengths and -- under the dual scaling strategy --
the centrality test on the sum of the barriers.  Where the paper leaves details open (order of the
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
    """rows of s / z: [zero | nonneg | soc_1 | soc_2 | ... | psd_1 | ... | exp_1 | ... | pow_1 | ...] (exponential and power
    cones: 3 rows each; `pow` lists the exponents alpha of x^alpha y^(1-alpha) >= |z|; `psd` lists matrix orders p, p (p + 1) / 2
    rows each: the upper triangle column by column with the off-diagonal entries times sqrt 2)"""

    def __init__(self, zero, nonneg, soc, exp=0, pow=(), psd=()):
        self.zero, self.nonneg, self.soc = int(zero), int(nonneg), [int(d) for d in soc]
        self.exp, self.pow = int(exp), [float(a) for a in pow]
        self.psd = [int(p_) for p_ in psd]
        self.soc_start = []
        o = self.zero + self.nonneg
        for d in self.soc:
            self.soc_start.append(o)
            o += d
        self.psd_start = []
        for p_ in self.psd:
            self.psd_start.append(o)
            o += p_ * (p_ + 1) // 2
        self.ns = []                                  # nonsymmetric cones: (first row, alpha or None for the exponential cone)
        for a in [None] * self.exp + self.pow:
            self.ns.append((o, a))
            o += 3
        self.m = o
        self.degree = self.nonneg + len(self.soc) + sum(self.psd) + 3 * len(self.ns)
        self.symmetric = not self.ns
        self.nn = slice(self.zero, self.zero + self.nonneg)

    def socs(self):
        return [slice(a, a + d) for a, d in zip(self.soc_start, self.soc)]

    def psds(self):
        """(slice of rows, matrix order) per PSD cone"""
        return [(slice(a, a + p_ * (p_ + 1) // 2), p_) for a, p_ in zip(self.psd_start, self.psd)]


SQRT2 = np.sqrt(2.0)


def svec_to_mat(v, p):
    M = np.zeros((p, p))
    k = 0
    for j in range(p):
        for i in range(j + 1):
            M[i, j] = M[j, i] = v[k] if i == j else v[k] / SQRT2
            k += 1
    return M


def mat_to_svec(M):
    p = M.shape[0]
    v = np.zeros(p * (p + 1) // 2)
    k = 0
    for j in range(p):
        for i in range(j + 1):
            v[k] = M[i, j] if i == j else M[i, j] * SQRT2
            k += 1
    return v


def svec_diag(p):
    """positions of the diagonal entries"""
    return np.array([j * (j + 1) // 2 + j for j in range(p)])


def _soc_res(v):
    return v[0] * v[0] - float(v[1:] @ v[1:])


# ------------------------------------------------------------------------------------------------ nonsymmetric cones
# K_exp = {(x, y, z): y > 0, y e^(x / y) <= z},  K_pow(a) = {(x, y, z): x^a y^(1 - a) >= |z|, x, y >= 0}.  Both dual barriers
# have the form  f*(z) = -log zeta(z) - sum_i c_i log |z_i|  (degree 3):
#   exp:  zeta = z1 log(-z1 / z3) - z1 + z2,                         c = (1, 0, 1)
#   pow:  zeta = (z1 / a)^(2a) (z2 / (1 - a))^(2 - 2a) - z3^2,       c = (1 - a, a, 0)
# so gradient, Hessian and the third directional derivative follow from zeta's own derivatives by the chain rule.
EXP_CENTRAL = (-1.051383945322714, 0.556409619469370, 1.258967884768947)


def _logsafe(v):
    return np.log(v) if v > 0.0 else -np.inf


def ns_zeta(z, alpha):
    """(zeta, grad zeta, hess zeta, c)"""
    z0, z1, z2 = float(z[0]), float(z[1]), float(z[2])
    if alpha is None:
        l = np.log(-z0 / z2)
        zeta = z0 * l - z0 + z1
        g = np.array([l, 1.0, -z0 / z2])
        H = np.array([[1.0 / z0, 0.0, -1.0 / z2], [0.0, 0.0, 0.0], [-1.0 / z2, 0.0, z0 / (z2 * z2)]])
        return zeta, g, H, np.array([1.0, 0.0, 1.0])
    a, b = 2.0 * alpha, 2.0 - 2.0 * alpha
    phi = np.exp(a * np.log(z0 / alpha) + b * np.log(z1 / (1.0 - alpha)))
    zeta = phi - z2 * z2
    g = np.array([a * phi / z0, b * phi / z1, -2.0 * z2])
    h01 = a * b * phi / (z0 * z1)
    H = np.array([[a * (a - 1.0) * phi / (z0 * z0), h01, 0.0], [h01, b * (b - 1.0) * phi / (z1 * z1), 0.0], [0.0, 0.0, -2.0]])
    return zeta, g, H, np.array([1.0 - alpha, alpha, 0.0])


def ns_zeta3(z, alpha, u, v):
    """D^3 zeta(z)[u, v]"""
    z0, z1, z2 = float(z[0]), float(z[1]), float(z[2])
    if alpha is None:
        return np.array([-u[0] * v[0] / (z0 * z0) + u[2] * v[2] / (z2 * z2), 0.0,
                         (u[0] * v[2] + u[2] * v[0]) / (z2 * z2) - 2.0 * z0 * u[2] * v[2] / (z2 * z2 * z2)])
    a, b = 2.0 * alpha, 2.0 - 2.0 * alpha
    phi = np.exp(a * np.log(z0 / alpha) + b * np.log(z1 / (1.0 - alpha)))
    p000 = a * (a - 1.0) * (a - 2.0) * phi / (z0 * z0 * z0)
    p001 = a * (a - 1.0) * b * phi / (z0 * z0 * z1)
    p011 = a * b * (b - 1.0) * phi / (z0 * z1 * z1)
    p111 = b * (b - 1.0) * (b - 2.0) * phi / (z1 * z1 * z1)
    x = u[0] * v[1] + u[1] * v[0]
    return np.array([p000 * u[0] * v[0] + p001 * x + p011 * u[1] * v[1],
                     p001 * u[0] * v[0] + p011 * x + p111 * u[1] * v[1], 0.0])


def ns_dual_feasible(z, alpha):
    if alpha is None:
        if z[2] > 0.0 and z[0] < 0.0:
            return z[1] - z[0] - z[0] * np.log(-z[2] / z[0]) > 0.0
        return False
    if z[0] > 0.0 and z[1] > 0.0:
        return np.exp(2.0 * alpha * np.log(z[0] / alpha) + (2.0 - 2.0 * alpha) * np.log(z[1] / (1.0 - alpha))) - z[2] * z[2] > 0.0
    return False


def ns_primal_feasible(s, alpha):
    if alpha is None:
        if s[2] > 0.0 and s[1] > 0.0:
            return s[1] * np.log(s[2] / s[1]) - s[0] > 0.0
        return False
    if s[0] > 0.0 and s[1] > 0.0:
        return np.exp(2.0 * alpha * np.log(s[0]) + (2.0 - 2.0 * alpha) * np.log(s[1])) - s[2] * s[2] > 0.0
    return False


def _inv_where(c, z):
    """1 / z_i where the barrier has a log |z_i| term (c_i != 0), 0 elsewhere"""
    return np.array([1.0 / zi if ci != 0.0 else 0.0 for ci, zi in zip(c, z)])


def ns_dual_grad_hess(z, alpha):
    """gradient and Hessian of f* at z"""
    zeta, g, H, c = ns_zeta(z, alpha)
    iz = _inv_where(c, z)
    grad = -g / zeta - c * iz
    hess = np.outer(g, g) / (zeta * zeta) - H / zeta + np.diag(c * iz * iz)
    return grad, hess


def ns_barrier_dual(z, alpha):
    if not ns_dual_feasible(z, alpha):
        return np.inf
    zeta, _, _, c = ns_zeta(z, alpha)
    return -_logsafe(zeta) - float(sum(ci * np.log(abs(zi)) for ci, zi in zip(c, z) if ci != 0.0))


def wright_omega(x):
    """omega + log(omega) = x for x >= 1 (Newton from x - log x: error < 1e-16 after a few steps)"""
    w = x - np.log(x) if x > 1.0 else 1.0
    for _ in range(8):
        w = w - (w + np.log(w) - x) * w / (w + 1.0)
    return w


def _pow_root(s, alpha):
    """p >= 0 with  log(p^2 + 2p) - log s3^2 = 2a log((1+a+a p)/(a s1)) + 2(1-a) log((2-a+(1-a) p)/((1-a) s2)):
    the third component of the primal gradient is p / s3.  The left side minus the right increases in p from -inf to
    log(s1^2a s2^(2-2a) / s3^2) > 0: bisection-safeguarded Newton."""
    a = alpha
    l0 = np.log(s[2] * s[2])

    def F(p):
        return (np.log(p * p + 2.0 * p) - l0 - 2.0 * a * np.log((1.0 + a + a * p) / (a * s[0]))
                - 2.0 * (1.0 - a) * np.log((2.0 - a + (1.0 - a) * p) / ((1.0 - a) * s[1])))

    def dF(p):
        return (2.0 * p + 2.0) / (p * p + 2.0 * p) - 2.0 * a * a / (1.0 + a + a * p) - 2.0 * (1.0 - a) ** 2 / (2.0 - a + (1.0 - a) * p)
    lo, hi = 0.0, 1.0
    for _ in range(200):
        if F(hi) > 0.0:
            break
        lo, hi = hi, 2.0 * hi
    p = 0.5 * (lo + hi)
    for _ in range(100):
        f = F(p)
        if f > 0.0:
            hi = p
        else:
            lo = p
        pn = p - f / dF(p)
        if not (lo < pn < hi):
            pn = 0.5 * (lo + hi)
        if abs(pn - p) <= 1e-15 * pn:
            p = pn
            break
        p = pn
    return p


def ns_gradient_primal(s, alpha):
    """gradient of the primal barrier f(s) = sup_z {-<s, z> - f*(z)}:  g = -z~ with grad f*(z~) = -s"""
    g = np.zeros(3)
    if alpha is None:
        w = wright_omega(1.0 - s[0] / s[1] - np.log(s[1] / s[2]))
        g[0] = 1.0 / ((w - 1.0) * s[1])
        g[1] = g[0] + g[0] * np.log(w * s[1] / s[2]) - 1.0 / s[1]
        g[2] = w / ((1.0 - w) * s[2])
        return g
    if abs(s[2]) > np.finfo(float).eps:
        p = _pow_root(s, alpha)
        g[2] = p / s[2]
    else:
        p = 0.0
    g[0] = -(1.0 + alpha + alpha * p) / s[0]
    g[1] = -(2.0 - alpha + (1.0 - alpha) * p) / s[1]
    return g


def ns_barrier_primal(s, alpha):
    """f(s) = <s, g(s)> - f*(-g(s)) = -3 - f*(-g(s))"""
    if not ns_primal_feasible(s, alpha):
        return np.inf
    if alpha is None:
        w = wright_omega(1.0 - s[0] / s[1] - np.log(s[1] / s[2]))
        return -_logsafe((w - 1.0) * (w - 1.0) / w) - 2.0 * np.log(s[1]) - np.log(s[2]) - 3.0
    return -3.0 - ns_barrier_dual(-ns_gradient_primal(s, alpha), alpha)


def _chol3_solve(H, b):
    """H u = b by an explicit 3 x 3 Cholesky; None when H is not positive definite"""
    l00 = H[0, 0]
    if not l00 > 0.0:
        return None
    l00 = np.sqrt(l00)
    l10, l20 = H[1, 0] / l00, H[2, 0] / l00
    t = H[1, 1] - l10 * l10
    if not t > 0.0:
        return None
    l11 = np.sqrt(t)
    l21 = (H[2, 1] - l20 * l10) / l11
    t = H[2, 2] - l20 * l20 - l21 * l21
    if not t > 0.0:
        return None
    l22 = np.sqrt(t)
    y0 = b[0] / l00
    y1 = (b[1] - l10 * y0) / l11
    y2 = (b[2] - l20 * y0 - l21 * y1) / l22
    u2 = y2 / l22
    u1 = (y1 - l21 * u2) / l11
    u0 = (y0 - l10 * u1 - l20 * u2) / l00
    return np.array([u0, u1, u2])


def ns_higher_correction(z, alpha, ds, dz):
    """eta = 1/2 D^3 f*(z)[u, v],  u = (hess f*(z))^-1 ds,  v = dz"""
    zeta, g, H, c = ns_zeta(z, alpha)
    iz = _inv_where(c, z)
    hess = np.outer(g, g) / (zeta * zeta) - H / zeta + np.diag(c * iz * iz)
    u = _chol3_solve(hess, ds)
    if u is None:
        return np.zeros(3)
    v = dz
    gu, gv = float(g @ u), float(g @ v)
    Hu, Hv = H @ u, H @ v
    T = (-ns_zeta3(z, alpha, u, v) / zeta + (Hu * gv + Hv * gu + g * float(u @ Hv)) / (zeta * zeta)
         - 2.0 * g * gu * gv / (zeta * zeta * zeta) - 2.0 * c * u * v * iz * iz * iz)
    return 0.5 * T


def ns_primal_dual_Hs(s, z, alpha, grad, hess):
    """the scaling block of a nonsymmetric cone under the primal-dual strategy (grad, hess: of f* at z); mu * hess where
    the rank-three form is not safe"""
    st = grad
    zt = ns_gradient_primal(s, alpha)
    dot_sz = float(s @ z)
    mu = dot_sz / 3.0
    mut = float(zt @ st) / 3.0
    ds_ = s + mu * st
    dz_ = z + mu * zt
    dot_dsz = float(ds_ @ dz_)
    de1 = mu * mut - 1.0
    de2 = float(zt @ (hess @ zt)) - 3.0 * mut * mut
    eps = np.finfo(float).eps
    if abs(de1) > np.sqrt(eps) and abs(de2) > eps and dot_sz > 0.0 and dot_dsz > 0.0:
        tmp = mut * st - hess @ zt
        M = hess - np.outer(st, st) / 3.0 - np.outer(tmp, tmp) / de2
        t = mu * np.sqrt(float((M * M).sum()))
        ax = np.array([z[1] * zt[2] - z[2] * zt[1], z[2] * zt[0] - z[0] * zt[2], z[0] * zt[1] - z[1] * zt[0]])
        ax = ax / np.sqrt(float(ax @ ax))
        return np.outer(s, s) / dot_sz + np.outer(ds_, ds_) / dot_dsz + t * np.outer(ax, ax)
    return mu * hess


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
    for sl in cones.socs() + [sl_ for sl_, _ in cones.psds()]:
        ew = E[sl].mean() / E[sl]
        A[sl, :] *= ew[:, None]
        b[sl] *= ew
        E[sl] *= ew
    for st, _ in cones.ns:                  # exponential / power cones admit no row scaling at all: back to 1
        sl = slice(st, st + 3)
        ew = 1.0 / E[sl]
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
    for sl, p in cones.psds():
        e = np.linalg.eigvalsh(svec_to_mat(v[sl], p))
        alpha = min(alpha, float(e.min()))
        beta += float(np.maximum(e, 0.0).sum())
    return alpha, beta


def _unit_shift(cones, v, a, primal):
    if cones.nonneg:
        v[cones.nn] += a
    for sl in cones.socs():
        v[sl.start] += a
    for sl, p in cones.psds():
        v[sl.start + svec_diag(p)] += a
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
    """Nesterov-Todd scaling: nonneg w = sqrt(s/z); SOC W = eta [[w0, w1'], [w1, I + w1 w1'/(1+w0)]]; PSD cone: with S = L1 L1',
    Z = L2 L2' and the SVD L2'L1 = U diag(lambda) V':  R = L1 V diag(lambda)^-1/2, R^-1 = diag(lambda)^-1/2 U'L2',
    W x = svec(R' mat(x) R), W'W x = svec(Q mat(x) Q) with Q = R R' (the cone's NT point), W z = W^-T s = svec(diag(lambda))"""

    def __init__(self, cones):
        self.c = cones
        self.identity()

    def identity(self):
        c = self.c
        self.ns_grad = [np.zeros(3) for _ in c.ns]       # gradient of the dual barrier at z
        self.ns_Hs = [np.eye(3) for _ in c.ns]           # scaling blocks of the nonsymmetric cones
        self.w = np.ones(c.m)
        self.lam = np.ones(c.m)
        self.eta = [1.0] * len(c.soc)
        self.sw = []
        for d in c.soc:
            v = np.zeros(d)
            v[0] = 1.0
            self.sw.append(v)
        self.psd_R = [np.eye(p_) for p_ in c.psd]
        self.psd_Rinv = [np.eye(p_) for p_ in c.psd]
        self.psd_lam = [np.ones(p_) for p_ in c.psd]

    def update(self, s, z, mu=None, dual_strategy=False):
        c = self.c
        ok = True
        for k, (st, alpha) in enumerate(c.ns):
            zk, sk = z[st:st + 3], s[st:st + 3]
            grad, hess = ns_dual_grad_hess(zk, alpha)
            self.ns_grad[k] = grad
            self.ns_Hs[k] = mu * hess if dual_strategy else ns_primal_dual_Hs(sk, zk, alpha, grad, hess)
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
        for k, (sl, p_) in enumerate(c.psds()):
            try:
                L1 = np.linalg.cholesky(svec_to_mat(s[sl], p_))
                L2 = np.linalg.cholesky(svec_to_mat(z[sl], p_))
            except np.linalg.LinAlgError:
                ok = False
                continue
            U, sig, Vt = np.linalg.svd(L2.T @ L1)
            isq = 1.0 / np.sqrt(sig)
            self.psd_R[k] = (L1 @ Vt.T) * isq[None, :]
            self.psd_Rinv[k] = isq[:, None] * (U.T @ L2.T)
            self.psd_lam[k] = sig
            self.lam[sl] = mat_to_svec(np.diag(sig))
        return ok

    def psd_Hs(self, k):
        """dense (Q x_s Q): entry (a, b) = c_a c_b / 2 (Q_ik Q_jl + Q_il Q_jk), a <-> (i, j), b <-> (k, l), c = sqrt 2 off the diagonal"""
        Q = self.psd_R[k] @ self.psd_R[k].T
        p_ = Q.shape[0]
        idx = [(i, j) for j in range(p_) for i in range(j + 1)]
        H = np.zeros((len(idx), len(idx)))
        for a, (i, j) in enumerate(idx):
            ca = 1.0 if i == j else SQRT2
            for b, (kk, l) in enumerate(idx):
                cb = 1.0 if kk == l else SQRT2
                H[a, b] = (ca * cb * 0.5) * (Q[i, kk] * Q[j, l] + Q[i, l] * Q[j, kk])
        return H

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
        for k, (st, _) in enumerate(c.ns):
            H[st:st + 3, st:st + 3] = self.ns_Hs[k]
        for k, (sl, _) in enumerate(c.psds()):
            H[sl, sl] = self.psd_Hs(k)
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
        for k, (st, _) in enumerate(c.ns):
            out[st:st + 3] = self.ns_Hs[k] @ v[st:st + 3]
        for k, (sl, p_) in enumerate(c.psds()):
            Q = self.psd_R[k] @ self.psd_R[k].T
            out[sl] = mat_to_svec(Q @ svec_to_mat(v[sl], p_) @ Q)
        return out

    def mul_W(self, v, inv=False, trans=False):
        """W v, W'v, W^-1 v, W^-T v (the nonnegative and second-order cones' W is symmetric)"""
        c = self.c
        out = np.zeros(c.m)
        if c.nonneg:
            out[c.nn] = v[c.nn] / self.w[c.nn] if inv else v[c.nn] * self.w[c.nn]
        for k, sl in enumerate(c.socs()):
            out[sl] = self.mul_W_cone(k, v[sl], inv)
        for k, (sl, p_) in enumerate(c.psds()):
            X = svec_to_mat(v[sl], p_)
            R, Ri = self.psd_R[k], self.psd_Rinv[k]
            if not inv:
                Y = R @ X @ R.T if trans else R.T @ X @ R
            else:
                Y = Ri @ X @ Ri.T if trans else Ri.T @ X @ Ri
            out[sl] = mat_to_svec(Y)
        return out

    def circ(self, a, b):
        c = self.c
        out = np.zeros(c.m)
        out[c.nn] = a[c.nn] * b[c.nn]
        for sl in c.socs():
            out[sl.start] = float(a[sl] @ b[sl])
            out[sl.start + 1:sl.stop] = a[sl.start] * b[sl.start + 1:sl.stop] + b[sl.start] * a[sl.start + 1:sl.stop]
        for sl, p_ in c.psds():
            X, Y = svec_to_mat(a[sl], p_), svec_to_mat(b[sl], p_)
            out[sl] = mat_to_svec(0.5 * (X @ Y + Y @ X))
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
        for k, (sl, p_) in enumerate(c.psds()):
            lm = self.psd_lam[k]
            out[sl] = mat_to_svec(2.0 * svec_to_mat(d[sl], p_) / (lm[:, None] + lm[None, :]))
        return out

    def ds_offset(self, ds):
        """W'(lambda \\ ds)"""
        return self.mul_W(self.inv_circ_lam(ds), trans=True)

    def psd_step_length(self, dz, ds, amax):
        """largest a <= amax with lambda + a W dz and lambda + a W^-T ds in the cone: the smallest eigenvalue of
        diag(lambda)^-1/2 mat(.) diag(lambda)^-1/2"""
        a = amax
        for v in (self.mul_W(dz), self.mul_W(ds, inv=True, trans=True)):
            for k, (sl, p_) in enumerate(self.c.psds()):
                isq = 1.0 / np.sqrt(self.psd_lam[k])
                g = float(np.linalg.eigvalsh(isq[:, None] * svec_to_mat(v[sl], p_) * isq[None, :]).min())
                if g < 0.0:
                    a = min(a, -1.0 / g)
        return a


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

    if cones.symmetric:
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
    else:
        # ---- a nonsymmetric cone anywhere: every cone starts at its central point s = z = -grad f(s), x = 0
        x, s = np.zeros(n), np.zeros(m)
        s[cones.nn] = 1.0
        for sl in cones.socs():
            s[sl.start] = 1.0
        for sl, p_ in cones.psds():
            s[sl.start + svec_diag(p_)] = 1.0
        for st, alpha in cones.ns:
            s[st:st + 3] = EXP_CENTRAL if alpha is None else (np.sqrt(1.0 + alpha), np.sqrt(2.0 - alpha), 0.0)
        z = s.copy()
    tau, kap = 1.0, 1.0
    dual_strategy = False         # scaling strategy of the nonsymmetric cones: primal-dual until a checkpoint switches to dual

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
            if status == INSUFFICIENT_PROGRESS and not cones.symmetric and not dual_strategy:
                # strategy checkpoint: the primal-dual scaling gets a second chance as the dual scaling, from this iterate
                status, dual_strategy = UNSOLVED, True
                prev = dict(info)
                continue
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
        if not sc.update(s, z, mu, dual_strategy):
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
            a = min(_step_length(cones, z, dz, a), _step_length(cones, s, ds, a))      # symmetric cones first
            a = sc.psd_step_length(dz, ds, a)
            if not cones.symmetric:
                # back off from a full step so that the logarithms below are not taken at the boundary, then backtrack
                a = min(a, stg['max_step_fraction'])
                for st, alpha in cones.ns:
                    for v, dv, inside in ((z, dz, ns_dual_feasible), (s, ds, ns_primal_feasible)):
                        ak = a
                        while not inside(v[st:st + 3] + ak * dv[st:st + 3], alpha):
                            ak *= stg['linesearch_backtrack_step']
                            if ak < stg['min_terminate_step_length']:
                                ak = 0.0
                                break
                        a = min(a, ak)
            return a * stg['max_step_fraction'] if combined else a

        def barrier(dz, ds, dtau, dkap, a):
            """the centrality function of the dual scaling strategy at the trial point"""
            ct, ck = tau + a * dtau, kap + a * dkap
            sn, zn = s + a * ds, z + a * dz
            mu_ = (float(sn @ zn) + ct * ck) / (cones.degree + 1)
            val = (cones.degree + 1) * _logsafe(mu_) - _logsafe(ct) - _logsafe(ck)
            for i in range(cones.zero, cones.zero + cones.nonneg):
                val -= _logsafe(sn[i] * zn[i])
            for sl in cones.socs():
                rs_, rz_ = _soc_res(sn[sl]), _soc_res(zn[sl])
                val += -0.5 * _logsafe(rs_ * rz_) if (rs_ > 0.0 and rz_ > 0.0) else np.inf
            for st, alpha in cones.ns:
                val += ns_barrier_dual(zn[st:st + 3], alpha) + ns_barrier_primal(sn[st:st + 3], alpha)
            for sl, p_ in cones.psds():
                try:
                    l1 = np.linalg.cholesky(svec_to_mat(sn[sl], p_))
                    l2 = np.linalg.cholesky(svec_to_mat(zn[sl], p_))
                    val -= 2.0 * float(np.log(np.diag(l1)).sum()) + 2.0 * float(np.log(np.diag(l2)).sum())
                except np.linalg.LinAlgError:
                    val = np.inf
            return val
        # ---- affine step
        dx, dz, ds, dtau, dkap = kkt_solve(rx, rz, rtau, tau * kap, s)
        finite = np.isfinite(dtau) and np.isfinite(dz).all() and np.isfinite(dx).all()
        if not finite and not cones.symmetric:
            # strategy checkpoint (numerical error): dual scaling if not tried yet
            if not dual_strategy:
                dual_strategy = True
                continue
            status = NUMERICAL_ERROR
            break
        alpha = step_len(dz, ds, dtau, dkap, False)
        sigma = (1.0 - alpha) ** 3
        # ---- combined step
        shift = sc.circ(sc.mul_W(ds, inv=True, trans=True), sc.mul_W(dz))
        e = np.zeros(m)
        e[cones.nn] = 1.0
        for sl in cones.socs():
            e[sl.start] = 1.0
        for sl, p_ in cones.psds():
            e[sl.start + svec_diag(p_)] = 1.0
        d_s = sc.circ(sc.lam, sc.lam) + shift - sigma * mu * e
        d_s[:cones.zero] = 0.0
        rk = -sigma * mu + dtau * dkap + tau * kap
        ds_const = sc.ds_offset(d_s)
        for k, (st, alpha_k) in enumerate(cones.ns):       # ds = s + sigma mu grad f*(z) - eta, handed on as it is
            eta = ns_higher_correction(z[st:st + 3], alpha_k, ds[st:st + 3], dz[st:st + 3])
            ds_const[st:st + 3] = s[st:st + 3] + sigma * mu * sc.ns_grad[k] - eta
        dx, dz, ds, dtau, dkap = kkt_solve((1.0 - sigma) * rx, (1.0 - sigma) * rz, (1.0 - sigma) * rtau, rk, ds_const)
        finite = np.isfinite(dtau) and np.isfinite(dz).all() and np.isfinite(dx).all()
        if not finite and not cones.symmetric:
            if not dual_strategy:
                dual_strategy = True
                continue
            status = NUMERICAL_ERROR
            break
        alpha = step_len(dz, ds, dtau, dkap, True)
        if not cones.symmetric and dual_strategy:
            for _ in range(50):                                  # centrality: back to where the barrier sum is below 1
                if barrier(dz, ds, dtau, dkap, alpha) < 1.0:
                    break
                alpha *= stg['linesearch_backtrack_step']
        if not cones.symmetric and not dual_strategy and alpha < stg['min_switch_step_length']:
            dual_strategy = True                                 # strategy checkpoint (small step): redo with the dual scaling
            continue
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
    cones = Cones(desc.cones['zero'], desc.cones['nonneg'], desc.cones['soc'], desc.cones.get('exp', 0), desc.cones.get('pow', ()), desc.cones.get('psd', ()))
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
