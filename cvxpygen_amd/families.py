"""
Hand-canonicalised problem families (cvxpy-free), following SURVEY.md Appendix B: what cvxpy's
QP canonicalisation yields for the reference's example / test problems.

  nonneg_ls   `examples/main.py:16-25`, `tests/test_diff.py:14-25`
  mpc         `examples/MPC.ipynb` cell 1/3, `tests/test_E2E_QP.py:44-73`
  portfolio   `examples/portfolio.ipynb` cell 1/3/7, `tests/test_E2E_QP.py:76-110`

cvxpy replaces every `sum_squares(affine(theta, x))` by a new variable t with `t == affine` and an
objective term t't (P = 2I on the t block, constant), `abs(v)` by t with v <= t, -v <= t, and
`minimum(0, w)` by t <= 0, t <= w.  The ordering of variables inside x and of rows inside the
equality / inequality blocks follows cvxpy's internal `var_offsets` and is not derivable without
cvxpy; the orders chosen here are fixed and documented per family.
"""

from __future__ import annotations

import numpy as np

from .canon_builder import CanonBuilder, cmul
from .descriptor import FamilyDescriptor


# ------------------------------------------------------------------------------------------------
def nonneg_ls(m: int = 3, n: int = 2, sparsity=((0, 0, 1), (0, 1, 1)), seed: int = 1,
              name: str = 'nonneg_LS', solver: str = 'OSQP') -> FamilyDescriptor:
    """minimise ||A x - b||^2  s.t. x >= 0   (`examples/main.py:16-25`).

    x = [x (n); t (m)];  eq: A x - t = b (m rows);  ineq: -x <= 0 (n rows).
    Default values as the example: np.random.seed(1); A.data = randn(nnz); b = randn(m).
    """
    cb = CanonBuilder(name)
    if sparsity is not None:
        A = cb.param('A', (m, n), kind='sparse', sparsity=sparsity)
    else:
        A = cb.param('A', (m, n))
    b = cb.param('b', (m,))
    x = cb.var('x', (n,))
    t = cb.aux(m)
    cb.sum_squares(t)
    for i in range(m):
        ent = [(x[j], A[i, j]) for j in range(n) if A.structurally_nonzero(i, j)]
        ent.append((t[i], -1.0))
        cb.eq(ent, b[i])
    rows = [cb.ineq([(x[j], -1.0)], 0.0) for j in range(n)]
    cb.dual('d0', rows, (n,))

    rng = np.random.RandomState(seed)
    if sparsity is not None:
        Aval = rng.randn(len(sparsity[0]))
    else:
        Aval = rng.randn(m, n)
    bval = rng.randn(m)
    return cb.build({'A': Aval, 'b': bval}, solver=solver)


# ------------------------------------------------------------------------------------------------
def mpc_dynamics(n: int, m: int, td: float = 0.1):
    """Discrete double integrator: n/2 positions + n/2 velocities, m force inputs acting on the
    first m velocity states (for n=6, m=3 this is `examples/MPC.ipynb` cell 3 exactly; for
    n=12, m=4 it is the builder-defined extension named in SURVEY.md section 8(d))."""
    h = n // 2
    A_cont = np.zeros((n, n))
    A_cont[:h, h:] = np.eye(h)
    B_cont = np.zeros((n, m))
    B_cont[h:h + m, :] = np.eye(m)
    return np.eye(n) + td * A_cont, td * B_cont


def mpc(n: int = 6, m: int = 3, H: int = 10, sparse_params: bool = False, terminal_index=None,
        const: float = 0.0, x_init=None, name: str = 'MPC') -> FamilyDescriptor:
    """
    minimise ||Psqrt X[:,T]||^2 + ||Qsqrt X[:,:H]||_F^2 + ||Rsqrt U||_F^2 (+ const)
    s.t.     X[:,1:] == A X[:,:H] + B U ;  |U| <= 1 ;  X[:,0] == x_init
    (`examples/MPC.ipynb` cell 1: T = H, dense parameters;
     `tests/test_E2E_QP.py:44-73`: T = H-1, const = 1, diag cost parameters, sparse A, B.)

    x = [U (m*H); X (n*(H+1)); tP (n); tQ (n*H); tR (m*H); tA (m*H)]
    eq   = [tP - Psqrt X_T = 0 (n); tQ_k - Qsqrt X_k = 0 (n*H); tR_k - Rsqrt U_k = 0 (m*H);
            X_{k+1} - A X_k - B U_k = 0 (n*H); X_0 = x_init (n)]
    ineq = [U - tA <= 0 (m*H); -U - tA <= 0 (m*H); tA <= 1 (m*H)]
    user duals: dynamics (n x H), |U| <= 1 (m x H), init (n).
    """
    T = H if terminal_index is None else terminal_index
    cb = CanonBuilder(name)
    if sparse_params:
        Psqrt = cb.param('Psqrt', (n, n), kind='diag')
        Qsqrt = cb.param('Qsqrt', (n, n), kind='diag')
        Rsqrt = cb.param('Rsqrt', (m, m), kind='diag')
        nzA = [(i, i) for i in range(n)] + [(i, n // 2 + i) for i in range(n // 2)]
        Ap = cb.param('A', (n, n), kind='sparse', sparsity=tuple(zip(*nzA)))
        nzB = [(n // 2 + i, i) for i in range(min(m, n // 2))]
        Bp = cb.param('B', (n, m), kind='sparse', sparsity=tuple(zip(*nzB)))
    else:
        Psqrt = cb.param('Psqrt', (n, n))
        Qsqrt = cb.param('Qsqrt', (n, n))
        Rsqrt = cb.param('Rsqrt', (m, m))
        Ap = cb.param('A', (n, n))
        Bp = cb.param('B', (n, m))
    xi = cb.param('x_init', (n,))

    U = cb.var('U', (m, H))
    X = cb.var('X', (n, H + 1))
    tP = cb.aux(n)
    tQ = cb.aux(n * H).reshape((n, H), order='F')
    tR = cb.aux(m * H).reshape((m, H), order='F')
    tA = cb.aux(m * H).reshape((m, H), order='F')

    cb.sum_squares(tP)
    cb.sum_squares(tQ.ravel(order='F'))
    cb.sum_squares(tR.ravel(order='F'))
    if const:
        cb.const(const)

    def mat_rows(t_idx, Mp, nrow, ncol, v_idx):
        for i in range(nrow):
            ent = [(t_idx[i], 1.0)]
            ent += [(v_idx[j], cmul(Mp[i, j], -1.0)) for j in range(ncol)
                    if Mp.structurally_nonzero(i, j)]
            cb.eq(ent, 0.0)

    mat_rows(tP, Psqrt, n, n, X[:, T])
    for k in range(H):
        mat_rows(tQ[:, k], Qsqrt, n, n, X[:, k])
    for k in range(H):
        mat_rows(tR[:, k], Rsqrt, m, m, U[:, k])
    dyn_rows = []
    for k in range(H):
        for i in range(n):
            ent = [(X[i, k + 1], 1.0)]
            ent += [(X[j, k], cmul(Ap[i, j], -1.0)) for j in range(n)
                    if Ap.structurally_nonzero(i, j)]
            ent += [(U[j, k], cmul(Bp[i, j], -1.0)) for j in range(m)
                    if Bp.structurally_nonzero(i, j)]
            dyn_rows.append(cb.eq(ent, 0.0))
    init_rows = [cb.eq([(X[i, 0], 1.0)], xi[i]) for i in range(n)]
    for k in range(H):
        for i in range(m):
            cb.ineq([(U[i, k], 1.0), (tA[i, k], -1.0)], 0.0)
    for k in range(H):
        for i in range(m):
            cb.ineq([(U[i, k], -1.0), (tA[i, k], -1.0)], 0.0)
    abs_rows = []
    for k in range(H):
        for i in range(m):
            abs_rows.append(cb.ineq([(tA[i, k], 1.0)], 1.0))
    cb.dual('d0', dyn_rows, (n, H))
    cb.dual('d1', abs_rows, (m, H))
    cb.dual('d2', init_rows, (n,))

    Ad, Bd = mpc_dynamics(n, m)
    if x_init is None:
        x_init = np.array([2, 2, 2, -1, -1, 1], dtype=float) if n == 6 else \
            np.concatenate([2 * np.ones(n // 2), -np.ones(n // 2)])
    vals = {'Psqrt': np.eye(n), 'Qsqrt': np.eye(n), 'Rsqrt': np.sqrt(0.1) * np.eye(m),
            'A': Ad, 'B': Bd, 'x_init': np.asarray(x_init, dtype=float)}
    return cb.build(vals)


# ------------------------------------------------------------------------------------------------
def portfolio(n: int = 100, m: int = 10, seed: int = 0, name: str = 'portfolio') -> FamilyDescriptor:
    """
    maximise a'w - ||Sig_f_sqrt f||^2 - ||d_sqrt * w||^2 - k_tc'|delta_w| + k_sh' min(0, w)
    s.t. f == F'w ; 1'w == 1 ; ||w||_1 <= L ; delta_w == w - w_prev
    (`examples/portfolio.ipynb` cell 1; defaults = cell 3 with np.random.seed(0)).

    Canonical (minimise the negative):
    x = [w (n); delta_w (n); f (m); t1 (m); t2 (n); ta (n); tm (n); tn (n)]
    eq   = [t1 - Sig_f_sqrt f = 0 (m); t2 - d_sqrt*w = 0 (n); f - F'w = 0 (m); 1'w = 1 (1);
            delta_w - w = -w_prev (n)]
    ineq = [dw - ta <= 0 (n); -dw - ta <= 0 (n); tm <= 0 (n); tm - w <= 0 (n);
            w - tn <= 0 (n); -w - tn <= 0 (n); 1'tn <= L (1)]
    """
    cb = CanonBuilder(name)
    cb.is_maximization = True
    a = cb.param('a', (n,))
    F = cb.param('F', (n, m))
    Sig = cb.param('Sig_f_sqrt', (m, m))
    dsq = cb.param('d_sqrt', (n,))
    ktc = cb.param('k_tc', (n,))
    ksh = cb.param('k_sh', (n,))
    wprev = cb.param('w_prev', (n,))
    Lp = cb.param('L', ())

    w = cb.var('w', (n,))
    dw = cb.var('delta_w', (n,))
    f = cb.var('f', (m,))
    t1 = cb.aux(m)
    t2 = cb.aux(n)
    ta = cb.aux(n)
    tm = cb.aux(n)
    tn = cb.aux(n)

    cb.sum_squares(t1)
    cb.sum_squares(t2)
    for i in range(n):
        cb.lin(w[i], cmul(a[i], -1.0))
        cb.lin(ta[i], ktc[i])
        cb.lin(tm[i], cmul(ksh[i], -1.0))

    for i in range(m):
        cb.eq([(t1[i], 1.0)] + [(f[j], cmul(Sig[i, j], -1.0)) for j in range(m)], 0.0)
    for i in range(n):
        cb.eq([(t2[i], 1.0), (w[i], cmul(dsq[i], -1.0))], 0.0)
    rows_f = [cb.eq([(f[j], 1.0)] + [(w[i], cmul(F[i, j], -1.0)) for i in range(n)], 0.0)
              for j in range(m)]
    rows_sum = [cb.eq([(w[i], 1.0) for i in range(n)], 1.0)]
    rows_dw = [cb.eq([(dw[i], 1.0), (w[i], -1.0)], cmul(wprev[i], -1.0)) for i in range(n)]

    for i in range(n):
        cb.ineq([(dw[i], 1.0), (ta[i], -1.0)], 0.0)
    for i in range(n):
        cb.ineq([(dw[i], -1.0), (ta[i], -1.0)], 0.0)
    for i in range(n):
        cb.ineq([(tm[i], 1.0)], 0.0)
    for i in range(n):
        cb.ineq([(tm[i], 1.0), (w[i], -1.0)], 0.0)
    for i in range(n):
        cb.ineq([(w[i], 1.0), (tn[i], -1.0)], 0.0)
    for i in range(n):
        cb.ineq([(w[i], -1.0), (tn[i], -1.0)], 0.0)
    rows_l1 = [cb.ineq([(tn[i], 1.0) for i in range(n)], Lp[()] if False else {Lp.up.col: 1.0})]

    cb.dual('d0', rows_f, (m,))
    cb.dual('d1', rows_sum, ())
    cb.dual('d2', rows_l1, ())
    cb.dual('d3', rows_dw, (n,))

    rng = np.random.RandomState(seed)
    alpha = rng.randn(n)
    vals = {'a': alpha, 'F': rng.randn(n, m), 'Sig_f_sqrt': rng.rand(m, m), 'd_sqrt': rng.rand(n),
            'k_tc': 0.01 * np.ones(n), 'k_sh': 0.05 * np.ones(n)}
    wp = rng.rand(n)
    vals['w_prev'] = wp / np.linalg.norm(wp)
    vals['L'] = 1.6
    return cb.build(vals)


def toy_box(name: str = 'toy_box', solver: str = 'OSQP') -> FamilyDescriptor:
    """minimise (x - a)^2 s.t. lb <= x <= ub (test family: primal infeasible when lb > ub).
    x = [x; t];  eq: x - t = a;  ineq: x <= ub, -x <= -lb."""
    cb = CanonBuilder(name)
    a = cb.param('a', ())
    lb = cb.param('lb', ())
    ub = cb.param('ub', ())
    x = cb.var('x', (1,))
    t = cb.aux(1)
    cb.sum_squares(t)
    cb.eq([(x[0], 1.0), (t[0], -1.0)], {a.up.col: 1.0})
    r1 = cb.ineq([(x[0], 1.0)], {ub.up.col: 1.0})
    r2 = cb.ineq([(x[0], -1.0)], {lb.up.col: -1.0})
    cb.dual('d0', [r1], (1,))
    cb.dual('d1', [r2], (1,))
    return cb.build({'a': 0.3, 'lb': -1.0, 'ub': 1.0}, solver=solver)


def toy_lp(name: str = 'toy_lp', solver: str = 'OSQP') -> FamilyDescriptor:
    """minimise c*x s.t. x >= 0 (test family: dual infeasible / unbounded when c < 0)."""
    cb = CanonBuilder(name)
    c = cb.param('c', ())
    x = cb.var('x', (1,))
    cb.lin(x[0], {c.up.col: 1.0})
    r = cb.ineq([(x[0], -1.0)], 0.0)
    cb.dual('d0', [r], (1,))
    return cb.build({'c': 1.0}, solver=solver)


def actuator(name: str = 'actuator') -> FamilyDescriptor:
    """`tests/test_E2E_QP.py:14-41` (degenerate sizes n = 1, m = 3; data of :104-112 with seed 0):
    minimise ||A u - w||^2 + lamb_sm ||delta_u||^2 + kappa |u|   s.t. u_min <= u <= u_max,
    delta_u == u - u_prev.  The only reference family whose P depends on a parameter (lamb_sm), so
    it exercises osqp_update_data_mat with new P values.
    x = [u; delta_u; t (3) = A u - w; s = |u|]; eq 4 (t, delta_u), ineq 4 (bounds, |u| epigraph)."""
    cb = CanonBuilder(name)
    A = cb.param('A', (3, 1))
    w = cb.param('w', (3,))
    lamb = cb.param('lamb_sm', ())
    kappa = cb.param('kappa', (1,))
    u_prev = cb.param('u_prev', (1,))
    u_min = cb.param('u_min', (1,))
    u_max = cb.param('u_max', (1,))
    u = cb.var('u', (1,))
    du = cb.var('delta_u', (1, 1))
    t = cb.aux(3)
    s_ = cb.aux(1)
    cb.sum_squares(t)
    cb.quad(du[0, 0], du[0, 0], {lamb.up.col: 2.0})
    cb.lin(s_[0], {kappa.idx(0): 1.0})
    for i in range(3):
        cb.eq([(t[i], 1.0), (u[0], cmul(A[i, 0], -1.0))], cmul(w[i], -1.0))
    r_du = cb.eq([(du[0, 0], 1.0), (u[0], -1.0)], cmul(u_prev[0], -1.0))
    r_min = cb.ineq([(u[0], -1.0)], cmul(u_min[0], -1.0))
    r_max = cb.ineq([(u[0], 1.0)], u_max[0])
    cb.ineq([(u[0], 1.0), (s_[0], -1.0)], 0.0)
    cb.ineq([(u[0], -1.0), (s_[0], -1.0)], 0.0)
    cb.dual('d0', [r_min], (1,))
    cb.dual('d1', [r_max], (1,))
    cb.dual('d2', [r_du], (1, 1))
    lam0 = float(np.random.RandomState(0).rand())              # np.random.seed(0); np.random.rand()
    return cb.build({'A': np.ones((3, 1)), 'w': np.array([2.0, 3.0, 5.0]), 'lamb_sm': lam0,
                     'kappa': 0.1 * np.ones(1), 'u_prev': np.zeros(1), 'u_min': -np.ones(1), 'u_max': np.ones(1)})


def adp_dynamics(state: np.ndarray):
    """discrete-time dynamics of `tests/test_E2E_SOCP.py:42-55` (td = 0.1, unit mass)"""
    A_cont = np.zeros((6, 6))
    A_cont[0, 3] = A_cont[1, 4] = A_cont[2, 5] = 1.0
    A_cont[3, 3], A_cont[4, 4], A_cont[5, 5] = -state[3], -state[4], -state[5]
    B_cont = np.vstack([np.zeros((3, 3)), np.diag(state[3:])])
    return np.eye(6) + 0.1 * A_cont, 0.1 * B_cont


def adp_values(state: np.ndarray) -> Dict[str, np.ndarray]:
    """parameter values of `tests/test_E2E_SOCP.py:57-62` for one state"""
    A, B = adp_dynamics(state)
    return {'Rsqrt': np.sqrt(0.1) * np.eye(3), 'f': A @ state, 'G': B}


def adp(name: str = 'ADP') -> FamilyDescriptor:
    """BASELINE config 4: the ADP problem of `tests/test_E2E_SOCP.py:15-35` (norm form) in conic form
    the way cvxpy hands it to a quadratic-objective conic solver: sum_squares(affine) -> t == affine
    with objective t't (P = 2I on t), norm(u_i) <= 0.1 -> (tn_i, u_i) in SOC(4), tn_i <= 0.1.
    x = [u (2x3, F-order); t1 (6); t2 (3); tn (2)], rows: zero 9, nonneg 2, SOC [4, 4] (SURVEY.md
    Appendix B); theta = [diag(Rsqrt) 3, f 6, G 18 (F-order)]."""
    cb = CanonBuilder(name)
    n, m = 6, 3
    Rsqrt = cb.param('Rsqrt', (m, m), kind='diag')
    f = cb.param('f', (n,))
    G = cb.param('G', (n, m))
    u = cb.var('u', (2, m))
    t1, t2, tn = cb.aux(n), cb.aux(m), cb.aux(2)
    cb.sum_squares(t1)
    cb.sum_squares(t2)
    for i in range(n):
        cb.eq([(t1[i], 1.0)] + [(u[0, j], cmul(G[i, j], -1.0)) for j in range(m)], f[i])
    for i in range(m):
        cb.eq([(t2[i], 1.0), (u[0, i], cmul(Rsqrt[i, i], -1.0))], 0.0)
    rows = [cb.ineq([(tn[i], 1.0)], 0.1) for i in range(2)]
    for i in range(2):
        cb.soc([([(tn[i], -1.0)], 0.0)] + [([(u[i, j], -1.0)], 0.0) for j in range(m)])
    cb.dual('d0', rows, (2,))
    state = -2.0 + 4.0 * np.random.RandomState(0).rand(6)          # np.random.seed(0) of the test
    return cb.build(adp_values(state), solver='CLARABEL')


def adp_norm(name: str = 'ADP_norm', tie: bool = True) -> FamilyDescriptor:
    """Test family for the ECOS form (cvxpygen/solvers/ecos.py; no quadratic objective): the ADP problem of
    `tests/test_E2E_SOCP.py:15-35` with plain norms,  minimise ||f + G u_0|| + ||Rsqrt u_0||  s.t. ||u_i|| <= 0.1,
    as cvxpy hands it to a solver without quadratic objective: epigraph variables s1, s2 with
    (s1, f + G u_0) in SOC(7), (s2, Rsqrt u_0) in SOC(4), (tn_i, u_i) in SOC(4), tn_i <= 0.1; `tie` adds the
    equality rows u_1 = u_0 / 2 (so that the A x = b block of the ECOS form is not empty).  Returned in the
    stacked conic form with P = 0; `cvxpygen_amd.ecos_front.ecos_from_conic` splits it."""
    cb = CanonBuilder(name)
    n, m = 6, 3
    Rsqrt = cb.param('Rsqrt', (m, m), kind='diag')
    f = cb.param('f', (n,))
    G = cb.param('G', (n, m))
    u = cb.var('u', (2, m))
    s12, tn = cb.aux(2), cb.aux(2)
    cb.lin(s12[0], 1.0)
    cb.lin(s12[1], 1.0)
    eqs = [cb.eq([(u[1, j], 1.0), (u[0, j], -0.5)], 0.0) for j in range(m)] if tie else []
    rows = [cb.ineq([(tn[i], 1.0)], 0.1) for i in range(2)]
    cb.soc([([(s12[0], -1.0)], 0.0)] + [([(u[0, j], cmul(G[i, j], -1.0)) for j in range(m)], f[i]) for i in range(n)])
    cb.soc([([(s12[1], -1.0)], 0.0)] + [([(u[0, i], cmul(Rsqrt[i, i], -1.0))], 0.0) for i in range(m)])
    for i in range(2):
        cb.soc([([(tn[i], -1.0)], 0.0)] + [([(u[i, j], -1.0)], 0.0) for j in range(m)])
    cb.dual('d0', rows, (2,))
    if tie:
        cb.dual('d1', eqs, (m,))
    state = -2.0 + 4.0 * np.random.RandomState(0).rand(6)
    return cb.build(adp_values(state), solver='CLARABEL')


def toy_qa(n: int = 3, name: str = 'toy_qa') -> FamilyDescriptor:
    """minimise ||G x - h||^2 + c'x  s.t. 0 <= x <= 1 (test family: q AND A parameter-dependent, so
    osqp_update_data_mat and osqp_update_data_vec both fire; cvxpygen/solvers/osqp.py:20-59)"""
    cb = CanonBuilder(name)
    G = cb.param('G', (n, n))
    h = cb.param('h', (n,))
    c = cb.param('c', (n,))
    x = cb.var('x', (n,))
    t = cb.aux(n)
    cb.sum_squares(t)
    for j in range(n):
        cb.lin(x[j], {c.idx(j): 1.0})
    for i in range(n):
        cb.eq([(x[j], {G.idx(i, j): 1.0}) for j in range(n)] + [(t[i], -1.0)], {h.idx(i): 1.0})
    r1 = [cb.ineq([(x[j], 1.0)], 1.0) for j in range(n)]
    r2 = [cb.ineq([(x[j], -1.0)], 0.0) for j in range(n)]
    cb.dual('d0', r1, (n,))
    cb.dual('d1', r2, (n,))
    rng = np.random.default_rng(5)
    return cb.build({'G': np.eye(n) + 0.3 * rng.standard_normal((n, n)), 'h': rng.standard_normal(n),
                     'c': 0.5 * rng.standard_normal(n)})


FAMILIES = {'nonneg_LS': nonneg_ls, 'MPC': mpc, 'portfolio': portfolio, 'toy_box': toy_box,
            'toy_lp': toy_lp, 'toy_qa': toy_qa, 'ADP': adp, 'ADP_norm': adp_norm, 'actuator': actuator}


# ------------------------------------------------------------------------------------------------ exponential / power cones
# Test families for the nonsymmetric cones of the conic path (ClarabelExponentialConeT / ClarabelPowerConeT,
# `cvxpygen/solvers/clarabel.py:133-155`).  The reference's own tests hold no problem with these cones; each family below has a
# closed-form answer (tests/test_nonsym_cones.py).
def softmax_entropy(n: int = 4, name: str = 'softmax') -> FamilyDescriptor:
    """minimise c'x - sum_i entr(x_i)  s.t.  sum x = 1   (cvxpy: cp.Minimize(c @ x - cp.sum(cp.entr(x))), [cp.sum(x) == 1]):
    x* = softmax(-c), value -log sum exp(-c).  Conic form: t_i <= -x_i log x_i  <=>  (t_i, x_i, 1) in K_exp."""
    cb = CanonBuilder(name)
    c = cb.param('c', (n,))
    x = cb.var('x', (n,))
    t = cb.aux(n)
    for i in range(n):
        cb.lin(x[i], c[i])
        cb.lin(t[i], -1.0)
    r = cb.eq([(x[i], 1.0) for i in range(n)], 1.0)
    for i in range(n):
        cb.exp_cone([([(t[i], -1.0)], 0.0), ([(x[i], -1.0)], 0.0), ([], 1.0)])
    cb.dual('nu', [r], (1,))
    return cb.build({'c': np.linspace(-1.0, 1.0, n)}, solver='CLARABEL')


def cobb_douglas(alpha: float = 0.3, name: str = 'cobb_douglas') -> FamilyDescriptor:
    """maximise x^alpha y^(1 - alpha)  s.t.  p1 x + p2 y <= budget   (cvxpy: cp.Maximize(cp.geo_mean(..., p=[alpha, 1 - alpha]))):
    x* = alpha budget / p1, y* = (1 - alpha) budget / p2.  The prices are a MATRIX parameter (a row of A)."""
    cb = CanonBuilder(name)
    p = cb.param('p', (2,))
    budget = cb.param('budget', ())
    v = cb.var('v', (2,))
    z = cb.var('z', (1,))
    cb.lin(z[0], -1.0)
    cb.is_maximization = True
    r = cb.ineq([(v[0], p[0]), (v[1], p[1])], {budget.up.col: 1.0})
    cb.pow_cone(alpha, [([(v[0], -1.0)], 0.0), ([(v[1], -1.0)], 0.0), ([(z[0], -1.0)], 0.0)])
    cb.dual('lam', [r], (1,))
    return cb.build({'p': np.array([1.0, 2.0]), 'budget': 3.0}, solver='CLARABEL')


def exp_prox(n: int = 3, radius: float = 10.0, name: str = 'exp_prox') -> FamilyDescriptor:
    """minimise sum_i exp(x_i) + 1/2 ||x - a||^2  s.t.  ||x|| <= radius, x <= ub:  a quadratic objective, exponential,
    second-order and nonnegative cones in one family.  With the ball and the bound inactive x_i = a_i - omega(a_i) (Wright omega)."""
    cb = CanonBuilder(name)
    a = cb.param('a', (n,))
    ub = cb.param('ub', (n,))
    x = cb.var('x', (n,))
    t = cb.aux(n)
    cb.sum_squares(x)                     # P = 2 I on x: 1/2 x'Px = ||x||^2 -> scale the rest by 2
    for i in range(n):
        cb.lin(x[i], cmul(a[i], -2.0))
        cb.lin(t[i], 2.0)
    rows = [cb.ineq([(x[i], 1.0)], ub[i]) for i in range(n)]
    cb.soc([([], radius)] + [([(x[i], -1.0)], 0.0) for i in range(n)])
    for i in range(n):
        cb.exp_cone([([(x[i], -1.0)], 0.0), ([], 1.0), ([(t[i], -1.0)], 0.0)])      # exp(x_i) <= t_i
    cb.dual('mu', rows, (n,))
    return cb.build({'a': np.linspace(-1.0, 2.0, n), 'ub': 5.0 * np.ones(n)}, solver='CLARABEL')


# ------------------------------------------------------------------------------------------------ PSD cones
_SQRT2 = float(np.sqrt(2.0))


def _svec_pairs(p: int):
    return [(i, j) for j in range(p) for i in range(j + 1)]


def min_eig(p: int = 3, name: str = 'min_eig') -> FamilyDescriptor:
    """maximise t  s.t.  C - t I >= 0 (PSD):  t* = lambda_min(C); C (symmetric, p x p) is the parameter
    (cvxpy: cp.Maximize(t), [C - t * np.eye(p) >> 0])"""
    cb = CanonBuilder(name)
    C = cb.param('C', (p, p))
    t = cb.var('t', (1,))
    cb.lin(t[0], -1.0)
    cb.is_maximization = True
    cb.psd_cone(p, [([(t[0], 1.0)], C[i, j]) if i == j else ([], cmul(C[i, j], _SQRT2)) for i, j in _svec_pairs(p)])
    A = np.arange(p * p, dtype=float).reshape(p, p) / p
    return cb.build({'C': A + A.T}, solver='CLARABEL')


def psd_projection(p: int = 3, name: str = 'psd_projection') -> FamilyDescriptor:
    """minimise ||X - C||_F^2  s.t.  X >= 0 (PSD), over x = svec(X):  X* = C with its negative eigenvalues set to zero; value
    ||X* - C||^2 - ||C||^2 in the canonical form (the constant ||C||^2 is not part of it)"""
    cb = CanonBuilder(name)
    C = cb.param('C', (p, p))
    d = p * (p + 1) // 2
    x = cb.var('x', (d,))
    cb.sum_squares(x)
    for k, (i, j) in enumerate(_svec_pairs(p)):
        cb.lin(x[k], cmul(C[i, j], -2.0 if i == j else -2.0 * _SQRT2))
    cb.psd_cone(p, [([(x[k], -1.0)], 0.0) for k in range(d)])
    A = np.arange(p * p, dtype=float).reshape(p, p) / p - 1.0
    return cb.build({'C': A + A.T}, solver='CLARABEL')


def trace_sdp(p: int = 3, name: str = 'trace_sdp') -> FamilyDescriptor:
    """minimise tr(C X)  s.t.  tr X = 1, X >= 0, exp(x_00) <= 5, ||x|| <= 10:  lambda_min(C) with a zero, a PSD, an exponential and a
    second-order cone in one family (the last two inactive)"""
    cb = CanonBuilder(name)
    C = cb.param('C', (p, p))
    pairs = _svec_pairs(p)
    d = len(pairs)
    x = cb.var('x', (d,))
    for k, (i, j) in enumerate(pairs):
        cb.lin(x[k], C[i, j] if i == j else cmul(C[i, j], _SQRT2))
    cb.eq([(x[k], 1.0) for k, (i, j) in enumerate(pairs) if i == j], 1.0)
    cb.soc([([], 10.0)] + [([(x[k], -1.0)], 0.0) for k in range(d)])
    cb.psd_cone(p, [([(x[k], -1.0)], 0.0) for k in range(d)])
    cb.exp_cone([([(x[0], -1.0)], 0.0), ([], 1.0), ([], 5.0)])
    A = np.arange(p * p, dtype=float).reshape(p, p) / p
    return cb.build({'C': A + A.T}, solver='CLARABEL')
