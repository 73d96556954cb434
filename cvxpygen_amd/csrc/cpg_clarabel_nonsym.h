// Exponential and three-dimensional power cones of the conic interior-point kernel (SURVEY.md section 8 row C1;
// reference: ClarabelExponentialConeT / ClarabelPowerConeT of the `cones` array, cvxpygen/solvers/clarabel.py:133-155,
// 308-323).  Scalar arithmetic of ONE cone -- the kernel runs one cone per lane (cpg_clarabel_kernel.h).
//
//   K_exp    = {(x, y, z): y > 0, y exp(x / y) <= z}
//   K_pow(a) = {(x, y, z): x^a y^(1 - a) >= |z|, x, y >= 0},  0 < a < 1        (alpha == 0 below: the exponential cone)
//
// Both dual barriers have the form  f*(z) = -log zeta(z) - sum_i c_i log |z_i|  (degree 3),
//   exp:  zeta = z1 log(-z1 / z3) - z1 + z2,                        c = (1, 0, 1)
//   pow:  zeta = (z1 / a)^(2a) (z2 / (1 - a))^(2 - 2a) - z3^2,      c = (1 - a, a, 0)
// so the gradient, the Hessian and the third directional derivative of f* follow from zeta's own derivatives by the chain
// rule (Zeta below).  Method (Goulart & Chen 2024, nonsymmetric cones; Dahl & Andersen's primal-dual scaling): scaling block
// H_s = s s'/<s,z> + ds ds'/<ds,dz> + t a a' (fall-back mu H*(z)), third-order correction eta = 1/2 D^3 f*(z)[H*^-1 ds, dz],
// backtracking on the cone tests, centrality through the primal and dual barrier values.
#pragma once

namespace cpg {
namespace ns {

#define CPG_NS_EPS 2.220446049250313e-16
#define CPG_NS_INF __builtin_inf()      // (CPG_INFTY is OSQP's 1e30: the barrier sums below need the IEEE one)

CPG_DEV double logsafe(double v) { return v > 0.0 ? log(v) : -CPG_NS_INF; }

struct Zeta {
    double zeta, g[3], h00, h01, h02, h11, h12, h22, c[3], iz[3];     // iz_i = 1 / z_i where c_i != 0, else 0
};

CPG_DEV void zeta(const double z[3], double alpha, Zeta &Z) {
    const double z0 = z[0], z1 = z[1], z2 = z[2];
    if (alpha == 0.0) {
        const double l = log(-z0 / z2);
        Z.zeta = z0 * l - z0 + z1;
        Z.g[0] = l; Z.g[1] = 1.0; Z.g[2] = -z0 / z2;
        Z.h00 = 1.0 / z0; Z.h01 = 0.0; Z.h02 = -1.0 / z2; Z.h11 = 0.0; Z.h12 = 0.0; Z.h22 = z0 / (z2 * z2);
        Z.c[0] = 1.0; Z.c[1] = 0.0; Z.c[2] = 1.0;
        Z.iz[0] = 1.0 / z0; Z.iz[1] = 0.0; Z.iz[2] = 1.0 / z2;
    } else {
        const double a = 2.0 * alpha, b = 2.0 - 2.0 * alpha;
        const double phi = exp(a * log(z0 / alpha) + b * log(z1 / (1.0 - alpha)));
        Z.zeta = phi - z2 * z2;
        Z.g[0] = a * phi / z0; Z.g[1] = b * phi / z1; Z.g[2] = -2.0 * z2;
        Z.h00 = a * (a - 1.0) * phi / (z0 * z0); Z.h01 = a * b * phi / (z0 * z1); Z.h02 = 0.0;
        Z.h11 = b * (b - 1.0) * phi / (z1 * z1); Z.h12 = 0.0; Z.h22 = -2.0;
        Z.c[0] = 1.0 - alpha; Z.c[1] = alpha; Z.c[2] = 0.0;
        Z.iz[0] = 1.0 / z0; Z.iz[1] = 1.0 / z1; Z.iz[2] = 0.0;
    }
}

// D^3 zeta(z)[u, v]
CPG_DEV void zeta3(const double z[3], double alpha, const double u[3], const double v[3], double out[3]) {
    const double z0 = z[0], z1 = z[1], z2 = z[2];
    if (alpha == 0.0) {
        out[0] = -u[0] * v[0] / (z0 * z0) + u[2] * v[2] / (z2 * z2);
        out[1] = 0.0;
        out[2] = (u[0] * v[2] + u[2] * v[0]) / (z2 * z2) - 2.0 * z0 * u[2] * v[2] / (z2 * z2 * z2);
    } else {
        const double a = 2.0 * alpha, b = 2.0 - 2.0 * alpha;
        const double phi = exp(a * log(z0 / alpha) + b * log(z1 / (1.0 - alpha)));
        const double p000 = a * (a - 1.0) * (a - 2.0) * phi / (z0 * z0 * z0);
        const double p001 = a * (a - 1.0) * b * phi / (z0 * z0 * z1);
        const double p011 = a * b * (b - 1.0) * phi / (z0 * z1 * z1);
        const double p111 = b * (b - 1.0) * (b - 2.0) * phi / (z1 * z1 * z1);
        const double x = u[0] * v[1] + u[1] * v[0];
        out[0] = p000 * u[0] * v[0] + p001 * x + p011 * u[1] * v[1];
        out[1] = p001 * u[0] * v[0] + p011 * x + p111 * u[1] * v[1];
        out[2] = 0.0;
    }
}

CPG_DEV bool dual_feasible(const double z[3], double alpha) {
    if (alpha == 0.0) {
        if (z[2] > 0.0 && z[0] < 0.0) return z[1] - z[0] - z[0] * log(-z[2] / z[0]) > 0.0;
        return false;
    }
    if (z[0] > 0.0 && z[1] > 0.0)
        return exp(2.0 * alpha * log(z[0] / alpha) + (2.0 - 2.0 * alpha) * log(z[1] / (1.0 - alpha))) - z[2] * z[2] > 0.0;
    return false;
}

CPG_DEV bool primal_feasible(const double s[3], double alpha) {
    if (alpha == 0.0) {
        if (s[2] > 0.0 && s[1] > 0.0) return s[1] * log(s[2] / s[1]) - s[0] > 0.0;
        return false;
    }
    if (s[0] > 0.0 && s[1] > 0.0) return exp(2.0 * alpha * log(s[0]) + (2.0 - 2.0 * alpha) * log(s[1])) - s[2] * s[2] > 0.0;
    return false;
}

// gradient and Hessian (packed 00 01 02 11 12 22) of f* from zeta's derivatives
CPG_DEV void dual_grad_hess(const Zeta &Z, double grad[3], double H[6]) {
    const double zt = Z.zeta, zz = Z.zeta * Z.zeta;
    for (int i = 0; i < 3; i++) grad[i] = -Z.g[i] / zt - Z.c[i] * Z.iz[i];
    H[0] = Z.g[0] * Z.g[0] / zz - Z.h00 / zt + Z.c[0] * Z.iz[0] * Z.iz[0];
    H[1] = Z.g[0] * Z.g[1] / zz - Z.h01 / zt;
    H[2] = Z.g[0] * Z.g[2] / zz - Z.h02 / zt;
    H[3] = Z.g[1] * Z.g[1] / zz - Z.h11 / zt + Z.c[1] * Z.iz[1] * Z.iz[1];
    H[4] = Z.g[1] * Z.g[2] / zz - Z.h12 / zt;
    H[5] = Z.g[2] * Z.g[2] / zz - Z.h22 / zt + Z.c[2] * Z.iz[2] * Z.iz[2];
}

CPG_DEV double barrier_dual(const double z[3], double alpha) {
    if (!dual_feasible(z, alpha)) return CPG_NS_INF;
    Zeta Z;
    zeta(z, alpha, Z);
    double acc = 0.0;
    for (int i = 0; i < 3; i++) if (Z.c[i] != 0.0) acc += Z.c[i] * log(fabs(z[i]));
    return -logsafe(Z.zeta) - acc;
}

// omega + log(omega) = x for x >= 1
CPG_DEV double wright_omega(double x) {
    double w = x > 1.0 ? x - log(x) : 1.0;
#pragma nounroll
    for (int it = 0; it < 8; it++) w = w - (w + log(w) - x) * w / (w + 1.0);
    return w;
}

// p >= 0 with  log(p^2 + 2p) - log s3^2 = 2a log((1 + a + a p) / (a s1)) + 2(1 - a) log((2 - a + (1 - a) p) / ((1 - a) s2)):
// the left side minus the right increases in p from -inf to log(s1^2a s2^(2 - 2a) / s3^2) > 0; bisection-safeguarded Newton
CPG_DEV double pow_root(const double s[3], double a) {
    const double l0 = log(s[2] * s[2]);
    auto F = [&](double p) {
        return log(p * p + 2.0 * p) - l0 - 2.0 * a * log((1.0 + a + a * p) / (a * s[0])) -
               2.0 * (1.0 - a) * log((2.0 - a + (1.0 - a) * p) / ((1.0 - a) * s[1]));
    };
    auto dF = [&](double p) {
        return (2.0 * p + 2.0) / (p * p + 2.0 * p) - 2.0 * a * a / (1.0 + a + a * p) -
               2.0 * ((1.0 - a) * (1.0 - a)) / (2.0 - a + (1.0 - a) * p);
    };
    double lo = 0.0, hi = 1.0;
#pragma nounroll
    for (int it = 0; it < 200; it++) {
        if (F(hi) > 0.0) break;
        lo = hi; hi = 2.0 * hi;
    }
    double p = 0.5 * (lo + hi);
#pragma nounroll
    for (int it = 0; it < 100; it++) {
        const double f = F(p);
        if (f > 0.0) hi = p; else lo = p;
        double pn = p - f / dF(p);
        if (!(lo < pn && pn < hi)) pn = 0.5 * (lo + hi);
        const bool done = fabs(pn - p) <= 1e-15 * pn;
        p = pn;
        if (done) break;
    }
    return p;
}

// gradient of the primal barrier f(s) = sup_z {-<s, z> - f*(z)}:  g = -z~ with grad f*(z~) = -s
CPG_DEV void gradient_primal(const double s[3], double alpha, double g[3]) {
    if (alpha == 0.0) {
        const double w = wright_omega(1.0 - s[0] / s[1] - log(s[1] / s[2]));
        g[0] = 1.0 / ((w - 1.0) * s[1]);
        g[1] = g[0] + g[0] * log(w * s[1] / s[2]) - 1.0 / s[1];
        g[2] = w / ((1.0 - w) * s[2]);
        return;
    }
    double p = 0.0;
    g[2] = 0.0;
    if (fabs(s[2]) > CPG_NS_EPS) { p = pow_root(s, alpha); g[2] = p / s[2]; }
    g[0] = -(1.0 + alpha + alpha * p) / s[0];
    g[1] = -(2.0 - alpha + (1.0 - alpha) * p) / s[1];
}

// f(s) = <s, g(s)> - f*(-g(s)) = -3 - f*(-g(s))
CPG_DEV double barrier_primal(const double s[3], double alpha) {
    if (!primal_feasible(s, alpha)) return CPG_NS_INF;
    if (alpha == 0.0) {
        const double w = wright_omega(1.0 - s[0] / s[1] - log(s[1] / s[2]));
        return -logsafe((w - 1.0) * (w - 1.0) / w) - 2.0 * log(s[1]) - log(s[2]) - 3.0;
    }
    double g[3];
    gradient_primal(s, alpha, g);
    const double mg[3] = {-g[0], -g[1], -g[2]};
    return -3.0 - barrier_dual(mg, alpha);
}

// H u = b by an explicit 3 x 3 Cholesky; false when H is not positive definite
CPG_DEV bool chol3_solve(const double H[6], const double b[3], double u[3]) {
    if (!(H[0] > 0.0)) return false;
    const double l00 = sqrt(H[0]);
    const double l10 = H[1] / l00, l20 = H[2] / l00;
    double t = H[3] - l10 * l10;
    if (!(t > 0.0)) return false;
    const double l11 = sqrt(t);
    const double l21 = (H[4] - l20 * l10) / l11;
    t = H[5] - l20 * l20 - l21 * l21;
    if (!(t > 0.0)) return false;
    const double l22 = sqrt(t);
    const double y0 = b[0] / l00;
    const double y1 = (b[1] - l10 * y0) / l11;
    const double y2 = (b[2] - l20 * y0 - l21 * y1) / l22;
    u[2] = y2 / l22;
    u[1] = (y1 - l21 * u[2]) / l11;
    u[0] = (y0 - l10 * u[1] - l20 * u[2]) / l00;
    return true;
}

// eta = 1/2 D^3 f*(z)[u, v],  u = (hess f*(z))^-1 ds,  v = dz
CPG_DEV void higher_correction(const double z[3], double alpha, const double ds[3], const double dz[3], double eta[3]) {
    Zeta Z;
    zeta(z, alpha, Z);
    double grad[3], H[6], u[3];
    dual_grad_hess(Z, grad, H);
    if (!chol3_solve(H, ds, u)) { eta[0] = eta[1] = eta[2] = 0.0; return; }
    const double *v = dz;
    const double gu = Z.g[0] * u[0] + Z.g[1] * u[1] + Z.g[2] * u[2];
    const double gv = Z.g[0] * v[0] + Z.g[1] * v[1] + Z.g[2] * v[2];
    const double Hu[3] = {Z.h00 * u[0] + Z.h01 * u[1] + Z.h02 * u[2], Z.h01 * u[0] + Z.h11 * u[1] + Z.h12 * u[2],
                          Z.h02 * u[0] + Z.h12 * u[1] + Z.h22 * u[2]};
    const double Hv[3] = {Z.h00 * v[0] + Z.h01 * v[1] + Z.h02 * v[2], Z.h01 * v[0] + Z.h11 * v[1] + Z.h12 * v[2],
                          Z.h02 * v[0] + Z.h12 * v[1] + Z.h22 * v[2]};
    const double uHv = u[0] * Hv[0] + u[1] * Hv[1] + u[2] * Hv[2];
    double z3[3];
    zeta3(z, alpha, u, v, z3);
    const double zt = Z.zeta, zz = zt * zt, zzz = zz * zt;
    for (int i = 0; i < 3; i++) {
        const double T = -z3[i] / zt + (Hu[i] * gv + Hv[i] * gu + Z.g[i] * uHv) / zz - 2.0 * Z.g[i] * gu * gv / zzz -
                         2.0 * Z.c[i] * u[i] * v[i] * Z.iz[i] * Z.iz[i] * Z.iz[i];
        eta[i] = 0.5 * T;
    }
}

// scaling block (packed) under the primal-dual strategy; grad, H: of f* at z
CPG_DEV void primal_dual_Hs(const double s[3], const double z[3], double alpha, const double grad[3], const double H[6], double Hs[6]) {
    const double *st = grad;
    double zt[3];
    gradient_primal(s, alpha, zt);
    const double dot_sz = s[0] * z[0] + s[1] * z[1] + s[2] * z[2];
    const double mu = dot_sz / 3.0;
    const double mut = (zt[0] * st[0] + zt[1] * st[1] + zt[2] * st[2]) / 3.0;
    double ds[3], dz[3];
    for (int i = 0; i < 3; i++) { ds[i] = s[i] + mu * st[i]; dz[i] = z[i] + mu * zt[i]; }
    const double dot_dsz = ds[0] * dz[0] + ds[1] * dz[1] + ds[2] * dz[2];
    const double Hz[3] = {H[0] * zt[0] + H[1] * zt[1] + H[2] * zt[2], H[1] * zt[0] + H[3] * zt[1] + H[4] * zt[2],
                          H[2] * zt[0] + H[4] * zt[1] + H[5] * zt[2]};
    const double de1 = mu * mut - 1.0;
    const double de2 = (zt[0] * Hz[0] + zt[1] * Hz[1] + zt[2] * Hz[2]) - 3.0 * mut * mut;
    if (fabs(de1) > sqrt(CPG_NS_EPS) && fabs(de2) > CPG_NS_EPS && dot_sz > 0.0 && dot_dsz > 0.0) {
        double tmp[3];
        for (int i = 0; i < 3; i++) tmp[i] = mut * st[i] - Hz[i];
        double M[6];
        const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
        double fro = 0.0;
        for (int k = 0; k < 6; k++) {
            M[k] = H[k] - st[ia[k]] * st[ib[k]] / 3.0 - tmp[ia[k]] * tmp[ib[k]] / de2;
            fro += (ia[k] == ib[k] ? 1.0 : 2.0) * (M[k] * M[k]);
        }
        const double t = mu * sqrt(fro);
        double ax[3] = {z[1] * zt[2] - z[2] * zt[1], z[2] * zt[0] - z[0] * zt[2], z[0] * zt[1] - z[1] * zt[0]};
        const double an = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        for (int i = 0; i < 3; i++) ax[i] /= an;
        for (int k = 0; k < 6; k++)
            Hs[k] = s[ia[k]] * s[ib[k]] / dot_sz + ds[ia[k]] * ds[ib[k]] / dot_dsz + t * (ax[ia[k]] * ax[ib[k]]);
    } else {
        for (int k = 0; k < 6; k++) Hs[k] = mu * H[k];
    }
}

}  // namespace ns
}  // namespace cpg
