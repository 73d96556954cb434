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
// Working matrices: compact p x p (leading dimension p) in the cone's workspace of the wavefront's LDS slice -- private arrays would live
// in scratch memory (dynamic indices), and a cone's arithmetic is one lane's chain of dependent loads: 35 ms per instance at order 6 with
// scratch, see DESIGN.md 4.4.
#define CPG_PSD_WORK(p) (8 * (p) * (p) + 2 * (p))       // doubles of workspace per cone: eight matrices, two vectors

namespace cpg {
namespace psd {

#define CPG_PSD_SQRT2 1.4142135623730951
#define CPG_PSD_ISQRT2 0.7071067811865476

// matrices: row-major, compact (leading dimension p)
CPG_DEV void svec_to_mat(const double *v, int p, double *M) {
    const int ld = p;
    int k = 0;
    for (int j = 0; j < p; j++)
        for (int i = 0; i <= j; i++, k++) {
            const double x = i == j ? v[k] : v[k] * CPG_PSD_ISQRT2;
            M[i * ld + j] = x; M[j * ld + i] = x;
        }
}
CPG_DEV void mat_to_svec(const double *M, int p, double *v) {
    const int ld = p;
    int k = 0;
    for (int j = 0; j < p; j++)
        for (int i = 0; i <= j; i++, k++) v[k] = i == j ? M[i * ld + j] : M[i * ld + j] * CPG_PSD_SQRT2;
}
// C = op(A) op(B), op = transpose where the flag says so
CPG_DEV void matmul(int p, const double *A, bool ta, const double *B, bool tb, double *C) {
    const int ld = p;
    for (int i = 0; i < p; i++)
        for (int j = 0; j < p; j++) {
            double acc = 0.0;
            for (int k = 0; k < p; k++)
                acc += (ta ? A[k * ld + i] : A[i * ld + k]) * (tb ? B[j * ld + k] : B[k * ld + j]);
            C[i * ld + j] = acc;
        }
}
// Y = A X A' (ta false) or A' X A (ta true); T: work
CPG_DEV void congruence(int p, const double *A, bool ta, const double *X, double *T, double *Y) {
    matmul(p, A, ta, X, false, T);
    matmul(p, T, false, A, !ta, Y);
}
// lower Cholesky factor; false when A is not (numerically) positive definite
CPG_DEV bool cholesky(int p, const double *A, double *L) {
    const int ld = p;
    for (int i = 0; i < p; i++)
        for (int j = 0; j < p; j++) L[i * ld + j] = 0.0;
    for (int j = 0; j < p; j++) {
        double d = A[j * ld + j];
        for (int k = 0; k < j; k++) d -= L[j * ld + k] * L[j * ld + k];
        if (!(d > 0.0)) return false;
        const double ljj = sqrt(d);
        L[j * ld + j] = ljj;
        for (int i = j + 1; i < p; i++) {
            double v = A[i * ld + j];
            for (int k = 0; k < j; k++) v -= L[i * ld + k] * L[j * ld + k];
            L[i * ld + j] = v / ljj;
        }
    }
    return true;
}
// cyclic Jacobi on the symmetric matrix A (destroyed): eigenvalues ev, eigenvectors the COLUMNS of V (V == nullptr: values only)
CPG_DEV void jacobi(int p, double *A, double *V, double *ev) {
    const int ld = p;
    if (V)
        for (int i = 0; i < p; i++)
            for (int j = 0; j < p; j++) V[i * ld + j] = i == j ? 1.0 : 0.0;
#pragma nounroll
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < p; i++) {
            dg += A[i * ld + i] * A[i * ld + i];
            for (int j = i + 1; j < p; j++) off += A[i * ld + j] * A[i * ld + j];
        }
        if (!(off > 1e-32 * dg)) break;
        for (int a = 0; a < p - 1; a++)
            for (int b = a + 1; b < p; b++) {
                const double apq = A[a * ld + b];
                if (apq == 0.0) continue;
                const double th = (A[b * ld + b] - A[a * ld + a]) / (2.0 * apq);
                const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < p; k++) {          // columns a, b
                    const double ka = A[k * ld + a], kb = A[k * ld + b];
                    A[k * ld + a] = c * ka - s * kb; A[k * ld + b] = s * ka + c * kb;
                }
                for (int k = 0; k < p; k++) {          // rows a, b
                    const double ak = A[a * ld + k], bk = A[b * ld + k];
                    A[a * ld + k] = c * ak - s * bk; A[b * ld + k] = s * ak + c * bk;
                }
                if (V)
                    for (int k = 0; k < p; k++) {
                        const double ka = V[k * ld + a], kb = V[k * ld + b];
                        V[k * ld + a] = c * ka - s * kb; V[k * ld + b] = s * ka + c * kb;
                    }
            }
    }
    for (int i = 0; i < p; i++) ev[i] = A[i * ld + i];
}
CPG_DEV double eig_min(int p, double *A, double *ev) {
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
