// Positive semidefinite cones of the conic interior-point kernel (reference:
// ClarabelPSDTriangleConeT(p) of the `cones` array, cvxpygen/solvers/clarabel.py:138, 146, 320-323): rows = the upper triangle of a
// p x p matrix column by column, off-diagonal entries times sqrt 2 (svec).  Dense arithmetic of ONE cone on small matrices
// (p <= CPG_PSD_MAX) -- the kernel runs one cone per lane (cpg_clarabel_kernel.h).
//
// Nesterov-Todd scaling (Goulart & Chen 2024; Vandenberghe, "The CVXOPT linear and quadratic cone program solvers"):
//   S = L1 L1', Z = L2 L2', L2'L1 = U diag(lambda) V'   ->   R = L1 V diag(lambda)^-1/2,  R^-1 = diag(lambda)^-1/2 U'L2'
//   W x = svec(R' X R),  W'x = svec(R X R'),  W^-1 x = svec(R^-T X R^-1),  W^-T x = svec(R^-1 X R^-T),  W z = W^-T s = svec(diag(lambda))
//   W'W x = svec(Q X Q), Q = R R'; as a matrix over svec indices a <-> (i, j), b <-> (k, l):
//   (W'W)_ab = c_a c_b / 2 (Q_ik Q_jl + Q_il Q_jk), c = sqrt 2 off the diagonal, 1 on it.
// The singular value decomposition comes from a cyclic Jacobi eigenvalue iteration on (L2'L1)'(L2'L1): every singular value is
// sqrt(mu) on the central path, so squaring costs nothing that matters here.
#pragma once

#ifndef CPG_PSD_MAX
#define CPG_PSD_MAX 8           // largest matrix order (36 rows)
#endif
#define CPG_PSD_LD CPG_PSD_MAX

namespace cpg {
namespace psd {

#define CPG_PSD_SQRT2 1.4142135623730951
#define CPG_PSD_ISQRT2 0.7071067811865476

// matrices: row-major with leading dimension CPG_PSD_LD
CPG_DEV void svec_to_mat(const double *v, int p, double *M) {
    int k = 0;
    for (int j = 0; j < p; j++)
        for (int i = 0; i <= j; i++, k++) {
            const double x = i == j ? v[k] : v[k] * CPG_PSD_ISQRT2;
            M[i * CPG_PSD_LD + j] = x; M[j * CPG_PSD_LD + i] = x;
        }
}
CPG_DEV void mat_to_svec(const double *M, int p, double *v) {
    int k = 0;
    for (int j = 0; j < p; j++)
        for (int i = 0; i <= j; i++, k++) v[k] = i == j ? M[i * CPG_PSD_LD + j] : M[i * CPG_PSD_LD + j] * CPG_PSD_SQRT2;
}
// C = op(A) op(B), op = transpose where the flag says so
CPG_DEV void matmul(int p, const double *A, bool ta, const double *B, bool tb, double *C) {
    for (int i = 0; i < p; i++)
        for (int j = 0; j < p; j++) {
            double acc = 0.0;
            for (int k = 0; k < p; k++)
                acc += (ta ? A[k * CPG_PSD_LD + i] : A[i * CPG_PSD_LD + k]) * (tb ? B[j * CPG_PSD_LD + k] : B[k * CPG_PSD_LD + j]);
            C[i * CPG_PSD_LD + j] = acc;
        }
}
// Y = A X A' (ta false) or A' X A (ta true); T: work
CPG_DEV void congruence(int p, const double *A, bool ta, const double *X, double *T, double *Y) {
    matmul(p, A, ta, X, false, T);
    matmul(p, T, false, A, !ta, Y);
}
// lower Cholesky factor; false when A is not (numerically) positive definite
CPG_DEV bool cholesky(int p, const double *A, double *L) {
    for (int i = 0; i < p; i++)
        for (int j = 0; j < p; j++) L[i * CPG_PSD_LD + j] = 0.0;
    for (int j = 0; j < p; j++) {
        double d = A[j * CPG_PSD_LD + j];
        for (int k = 0; k < j; k++) d -= L[j * CPG_PSD_LD + k] * L[j * CPG_PSD_LD + k];
        if (!(d > 0.0)) return false;
        const double ljj = sqrt(d);
        L[j * CPG_PSD_LD + j] = ljj;
        for (int i = j + 1; i < p; i++) {
            double v = A[i * CPG_PSD_LD + j];
            for (int k = 0; k < j; k++) v -= L[i * CPG_PSD_LD + k] * L[j * CPG_PSD_LD + k];
            L[i * CPG_PSD_LD + j] = v / ljj;
        }
    }
    return true;
}
// cyclic Jacobi on the symmetric matrix A (destroyed): eigenvalues ev, eigenvectors the COLUMNS of V (V == nullptr: values only)
CPG_DEV void jacobi(int p, double *A, double *V, double *ev) {
    if (V)
        for (int i = 0; i < p; i++)
            for (int j = 0; j < p; j++) V[i * CPG_PSD_LD + j] = i == j ? 1.0 : 0.0;
#pragma nounroll
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < p; i++) {
            dg += A[i * CPG_PSD_LD + i] * A[i * CPG_PSD_LD + i];
            for (int j = i + 1; j < p; j++) off += A[i * CPG_PSD_LD + j] * A[i * CPG_PSD_LD + j];
        }
        if (!(off > 1e-32 * dg)) break;
        for (int a = 0; a < p - 1; a++)
            for (int b = a + 1; b < p; b++) {
                const double apq = A[a * CPG_PSD_LD + b];
                if (apq == 0.0) continue;
                const double th = (A[b * CPG_PSD_LD + b] - A[a * CPG_PSD_LD + a]) / (2.0 * apq);
                const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < p; k++) {          // columns a, b
                    const double ka = A[k * CPG_PSD_LD + a], kb = A[k * CPG_PSD_LD + b];
                    A[k * CPG_PSD_LD + a] = c * ka - s * kb; A[k * CPG_PSD_LD + b] = s * ka + c * kb;
                }
                for (int k = 0; k < p; k++) {          // rows a, b
                    const double ak = A[a * CPG_PSD_LD + k], bk = A[b * CPG_PSD_LD + k];
                    A[a * CPG_PSD_LD + k] = c * ak - s * bk; A[b * CPG_PSD_LD + k] = s * ak + c * bk;
                }
                if (V)
                    for (int k = 0; k < p; k++) {
                        const double ka = V[k * CPG_PSD_LD + a], kb = V[k * CPG_PSD_LD + b];
                        V[k * CPG_PSD_LD + a] = c * ka - s * kb; V[k * CPG_PSD_LD + b] = s * ka + c * kb;
                    }
            }
    }
    for (int i = 0; i < p; i++) ev[i] = A[i * CPG_PSD_LD + i];
}
CPG_DEV double eig_min(int p, double *A) {
    double ev[CPG_PSD_MAX];
    jacobi(p, A, nullptr, ev);
    double m = ev[0];
    for (int i = 1; i < p; i++) m = ev[i] < m ? ev[i] : m;
    return m;
}
// (W'W)_ab from Q (p x p, leading dimension p, in the wavefront's slice): a <-> (i, j), b <-> (k, l)
CPG_DEV double kkt_entry(const double *Q, int p, int i, int j, int k, int l) {
    const double ca = i == j ? 1.0 : CPG_PSD_SQRT2, cb = k == l ? 1.0 : CPG_PSD_SQRT2;
    return (ca * cb * 0.5) * (Q[i * p + k] * Q[j * p + l] + Q[i * p + l] * Q[j * p + k]);
}

}  // namespace psd
}  // namespace cpg
