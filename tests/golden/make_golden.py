"""
Generates tests/golden/known_answers.json: independent known answers for the reference's own
example inputs, computed with scipy (no solver from this repository involved).

  nonneg_LS  `examples/main.py:16-25` with np.random.seed(1): exact NNLS (scipy.optimize.nnls)
  MPC        `examples/MPC.ipynb` cells 1, 3 (x_init = [2,2,2,-1,-1,1]): the condensed problem is a
             bounded least-squares problem in U, solved exactly with scipy.optimize.lsq_linear
             (method='bvls', tol=1e-14)

  ADP        `tests/test_E2E_SOCP.py:15-64` (np.random.seed(seed), seed 0..4): the problem is a
             trust-region subproblem in u[0] (u[1] only has to stay in its ball), solved exactly
             through the secular equation ||(H + lam I)^-1 g|| = 0.1 with scipy.optimize.brentq;
             the multiplier of ||u[0]|| <= 0.1 is 2 * lam * 0.1

These are the fixtures SURVEY.md Appendix C asks to commit; the reference itself holds no golden
vectors.  Run:  python tests/golden/make_golden.py
"""
import json
import os

import numpy as np
from scipy.optimize import brentq, lsq_linear, nnls


def nonneg_ls():
    np.random.seed(1)
    data = np.random.randn(3)
    b = np.random.randn(3)
    A = np.zeros((3, 2))
    A[(0, 0, 1), (0, 1, 1)] = data
    x, _ = nnls(A, b)
    r = A @ x - b
    return dict(A_data=data.tolist(), b=b.tolist(), x=x.tolist(), obj=float(r @ r),
                dual_x_ge_0=(2 * A.T @ r).tolist())


def mpc(n=6, m=3, H=10):
    h = n // 2
    A_cont = np.zeros((n, n)); A_cont[:h, h:] = np.eye(h)
    B_cont = np.zeros((n, m)); B_cont[h:h + m, :] = np.eye(m)
    td = 0.1
    A, B = np.eye(n) + td * A_cont, td * B_cont
    x0 = np.array([2, 2, 2, -1, -1, 1], dtype=float)
    Rs = np.sqrt(0.1)
    # X_k = A^k x0 + sum_j A^(k-1-j) B u_j ; cost = sum_{k=0..H} |X_k|^2 + sum |Rs u_k|^2
    rows, rhs = [], []
    for k in range(H + 1):
        M = np.zeros((n, m * H))
        for j in range(k):
            M[:, j * m:(j + 1) * m] = np.linalg.matrix_power(A, k - 1 - j) @ B
        rows.append(M); rhs.append(-np.linalg.matrix_power(A, k) @ x0)
    rows.append(Rs * np.eye(m * H)); rhs.append(np.zeros(m * H))
    M, r = np.vstack(rows), np.concatenate(rhs)
    res = lsq_linear(M, r, bounds=(-1, 1), method='bvls', tol=1e-14, max_iter=10000)
    u = res.x
    U = u.reshape(H, m).T
    X = np.zeros((n, H + 1)); X[:, 0] = x0
    for k in range(H):
        X[:, k + 1] = A @ X[:, k] + B @ U[:, k]
    obj = float((X ** 2).sum() + 0.1 * (U ** 2).sum())
    return dict(x_init=x0.tolist(), U=U.tolist(), X=X.tolist(), obj=obj,
                n_at_bound=int((np.abs(np.abs(u) - 1) < 1e-9).sum()))


def adp(seed):
    np.random.seed(seed)
    state = -2 * np.ones(6) + 4 * np.random.rand(6)
    A_cont = np.zeros((6, 6))
    A_cont[0, 3] = A_cont[1, 4] = A_cont[2, 5] = 1.0
    A_cont[3, 3], A_cont[4, 4], A_cont[5, 5] = -state[3], -state[4], -state[5]
    B_cont = np.vstack([np.zeros((3, 3)), np.diag(state[3:])])
    A, B = np.eye(6) + 0.1 * A_cont, 0.1 * B_cont
    Rsqrt, f, G = np.sqrt(0.1) * np.eye(3), A @ state, B
    H, g, r = G.T @ G + Rsqrt.T @ Rsqrt, G.T @ f, 0.1
    u = -np.linalg.solve(H, g)
    lam = 0.0
    if np.linalg.norm(u) > r:
        fn = lambda l: np.linalg.norm(np.linalg.solve(H + l * np.eye(3), g)) - r
        hi = 1.0
        while fn(hi) > 0:
            hi *= 2
        lam = brentq(fn, 0.0, hi, xtol=1e-15, rtol=4 * np.finfo(float).eps, maxiter=500)
        u = -np.linalg.solve(H + lam * np.eye(3), g)
    obj = float(np.sum((f + G @ u) ** 2) + np.sum((Rsqrt @ u) ** 2))
    return dict(state=state.tolist(), Rsqrt_diag=np.diag(Rsqrt).tolist(), f=f.tolist(), G=G.tolist(),
                u0=u.tolist(), obj=obj, dual_norm_u0=float(2 * lam * r))


if __name__ == '__main__':
    out = dict(nonneg_LS=nonneg_ls(), MPC_6_3_10=mpc(), ADP={str(s): adp(s) for s in range(5)})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'known_answers.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print(path, out['nonneg_LS']['x'], out['MPC_6_3_10']['obj'])
