"""
TEST INFRASTRUCTURE ONLY -- dense numpy restatement of cvxpygen's QP adjoint
(`cvxpygen/templates/cpg_osqp_grad_compute.c.jinja2:432-531` driven by `cpg_gradient()`,
`cvxpygen/writer.py:233-312`), for ONE instance, following the template line by line:

  active set        a_i = -1 / +1 / 0 from y_i < -1e-12 / > 1e-12 / else          (:436-454)
  K                 [[P + 1e-6 I, A'], [A, -1e-6 I]]; an inactive row/column is removed by
                    cpg_ldl_delete, which leaves the exact factor of K with that row/column
                    zeroed and -1 on its diagonal                               (writer.py:362-365,
                                                                                   template :182-222)
  r = K^-1 [dx; 0], three refinement sweeps against K_true = [[P, A'], [A, 0]] with inactive
                    rows / columns skipped                                      (:456-490)
  dq = -r_x; dl / du = r_lambda routed by the sign; dP_ij = -1/2 (r_i x_j + x_i r_j);
  dA_ij = -(r_lambda_i x_j + y_i r_xj), 0 for inactive rows                     (:492-529)
  dtheta = sum_p C_p' d(p)  for q, l, u, P, A                                    (writer.py:268-311)

The sequential reference keeps the active set between calls (a row that starts "active upper"
and then sees y < 0 is not re-labelled, template :437-441); a batch has no history, so a_i is
taken from the sign of y_i as the code intends.  For the reference's canonical form this makes no
difference: l depends on parameters only on equality rows, where the l- and u-maps coincide.
"""
import numpy as np


def qp_adjoint(P_upper, A, x, y, dx, refine=3, eps=1e-6):
    Pu = np.asarray(P_upper, dtype=float)
    P = np.triu(Pu) + np.triu(Pu, 1).T
    A = np.asarray(A, dtype=float)
    n, m = P.shape[0], A.shape[0]
    a = np.where(y < -1e-12, -1, np.where(y > 1e-12, 1, 0))
    act = a != 0
    Am = A * act[:, None]
    K = np.block([[P + eps * np.eye(n), Am.T], [Am, np.diag(np.where(act, -eps, -1.0))]])
    Kt = np.block([[P, Am.T], [Am, np.zeros((m, m))]])
    rhs = np.concatenate([dx, np.zeros(m)])
    r = np.linalg.solve(K, rhs)
    for _ in range(refine):
        delta = rhs - Kt @ r
        delta[n:][~act] = 0.0
        r = r + np.linalg.solve(K, delta)
    rx, rl = r[:n], r[n:]
    dq = -rx
    dl = np.where(a == -1, rl, 0.0)
    du = np.where(a == 1, rl, 0.0)
    dP = -0.5 * (np.outer(rx, x) + np.outer(x, rx))          # full matrix; callers pick the pattern
    dA = -(np.outer(rl, x) + np.outer(y, rx)) * act[:, None]
    return dict(a=a, r=r, dq=dq, dl=dl, du=du, dP=dP, dA=dA)


def dtheta_from_canonical(desc, g):
    """un-canonicalise: dp = sum_p C_p' d(p) over the canonical parameters that change"""
    import scipy.sparse as sp
    NP = desc.NP
    out = np.zeros(NP + 1)
    Pc, Ac = sp.coo_matrix(desc.P), sp.coo_matrix(desc.A)
    parts = {'q': g['dq'], 'l': g['dl'][:desc.n_eq], 'u': g['du'],
             'P': g['dP'][Pc.row, Pc.col], 'A': g['dA'][Ac.row, Ac.col]}
    for pid, vec in parts.items():
        if desc.changes.get(pid, False):
            out += sp.csr_matrix(desc.maps[pid]).T @ vec
    return out[:NP]
