/*
 * TEST INFRASTRUCTURE ONLY -- scalar-C restatement of the interior-point method behind the reference's Clarabel path
 * (SURVEY.md section 8 row C1), statement for statement the algorithm of oracle/clarabel_numpy.py (whose header names
 * the sources: cvxpygen/solvers/clarabel.py:37-46, 63-119, 133-155, 172-204, 308-323 for what the reference fixes;
 * Goulart & Chen 2024 for the method).  Linked into oracle/liboracle.so; called by tests/ (against the numpy
 * restatement) and by bench.py's cpu_baseline leg of the conic workload; never by cvxpygen_amd/.
 *
 * PARITY UNPINNED, exactly as clarabel_numpy.py: Clarabel itself (Rust) is absent from the reference checkout.
 * Dense matrices in natural KKT order [x; z] with their own LDL': independent of the product's sparse plan.
 * One new solver per instance (clarabel.py:201-204), OpenMP over the instances of a batch.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { C_MAX_ITER, C_MAX_STEP_FRACTION, C_TOL_GAP_ABS, C_TOL_GAP_REL, C_TOL_FEAS, C_TOL_INFEAS_ABS, C_TOL_INFEAS_REL,
       C_TOL_KTRATIO, C_RED_TOL_GAP_ABS, C_RED_TOL_GAP_REL, C_RED_TOL_FEAS, C_RED_TOL_INFEAS_ABS, C_RED_TOL_INFEAS_REL,
       C_RED_TOL_KTRATIO, C_EQ_ENABLE, C_EQ_MAX_ITER, C_EQ_MIN, C_EQ_MAX, C_LS_BACKTRACK, C_MIN_SWITCH, C_MIN_TERMINATE,
       C_SREG_ENABLE, C_SREG_CONST, C_SREG_PROP, C_DREG_ENABLE, C_DREG_EPS, C_DREG_DELTA, C_IR_ENABLE, C_IR_RELTOL,
       C_IR_ABSTOL, C_IR_MAX_ITER, C_IR_STOP_RATIO, C_COUNT };     /* order of clarabel_numpy.DEFAULTS */
enum { ST_UNSOLVED, ST_SOLVED, ST_PINF, ST_DINF, ST_ALMOST_SOLVED, ST_ALMOST_PINF, ST_ALMOST_DINF, ST_MAX_ITER, ST_MAX_TIME,
       ST_NUMERICAL, ST_INSUFFICIENT };

typedef struct { int zero, nonneg, nsoc; const int *soc; int *start; int m, degree; } Cones;
typedef struct {
    int n, m, N; const double *stg; Cones c;
    double *P, *A, *q, *b, *D, *E, cs;          /* equilibrated data (dense row-major), scalings */
    double *K, *L, *d, *signs;                  /* KKT matrix, its regularised factor */
    double *w, *lam, *eta, *sw;                 /* NT scaling: nonneg w, lambda, per-SOC eta, SOC w vectors (at their rows) */
    double *t1, *t2, *t3, *t4;                  /* N-vectors of the linear solves */
} Ws;

static double *dv(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }
static double ninf(const double *v, int n) { double r = 0; for (int i = 0; i < n; i++) { double a = fabs(v[i]); if (a > r) r = a; } return r; }
static double sninf(const double *s, const double *v, int n) { double r = 0; for (int i = 0; i < n; i++) { double a = fabs(s[i] * v[i]); if (a > r) r = a; } return r; }
static double dot(const double *a, const double *b, int n) { double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }
static double soc_res(const double *v, int d) { return v[0] * v[0] - dot(v + 1, v + 1, d - 1); }
static double dmaxd(double a, double b) { return a > b ? a : b; }
static double dmind(double a, double b) { return a < b ? a : b; }

/* ---- equilibration (clarabel_numpy.equilibrate) */
static double lim(double v, double lo, double hi) { if (v == 0.0) v = 1.0; return v < lo ? lo : (v > hi ? hi : v); }
static void equilibrate(Ws *w) {
    const int n = w->n, m = w->m; const double *stg = w->stg;
    for (int j = 0; j < n; j++) w->D[j] = 1.0;
    for (int i = 0; i < m; i++) w->E[i] = 1.0;
    w->cs = 1.0;
    if (stg[C_EQ_ENABLE] == 0.0) return;
    const double lo = stg[C_EQ_MIN], hi = stg[C_EQ_MAX];
    double *dw = dv(n), *ew = dv(m);
    for (int it = 0; it < (int)stg[C_EQ_MAX_ITER]; it++) {
        for (int j = 0; j < n; j++) {
            double r = 0;
            for (int i = 0; i < n; i++) r = dmaxd(r, fabs(w->P[i * n + j]));
            for (int i = 0; i < m; i++) r = dmaxd(r, fabs(w->A[i * n + j]));
            dw[j] = 1.0 / sqrt(lim(r, lo, hi));
        }
        for (int i = 0; i < m; i++) { double r = 0; if (n) for (int j = 0; j < n; j++) r = dmaxd(r, fabs(w->A[i * n + j])); ew[i] = 1.0 / sqrt(lim(r, lo, hi)); }
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) w->P[i * n + j] = dw[i] * w->P[i * n + j] * dw[j];
        for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) w->A[i * n + j] = ew[i] * w->A[i * n + j] * dw[j];
        for (int j = 0; j < n; j++) { w->q[j] *= dw[j]; w->D[j] *= dw[j]; }
        for (int i = 0; i < m; i++) { w->b[i] *= ew[i]; w->E[i] *= ew[i]; }
        double pn = 0, qn = n ? ninf(w->q, n) : 0.0;
        if (n) { for (int j = 0; j < n; j++) { double r = 0; for (int i = 0; i < n; i++) r = dmaxd(r, fabs(w->P[i * n + j])); pn += r; } pn /= n; }
        if (pn != 0.0 && qn != 0.0) {
            double ct = 1.0 / dmaxd(pn, qn); ct = ct < lo ? lo : (ct > hi ? hi : ct);
            for (int k = 0; k < n * n; k++) w->P[k] *= ct;
            for (int j = 0; j < n; j++) w->q[j] *= ct;
            w->cs *= ct;
        }
    }
    for (int k = 0; k < w->c.nsoc; k++) {          /* second-order cone rows: one scaling per cone (the mean) */
        const int a = w->c.start[k], d = w->c.soc[k]; double mean = 0;
        for (int i = 0; i < d; i++) mean += w->E[a + i];
        mean /= d;
        for (int i = 0; i < d; i++) { const double e = mean / w->E[a + i];
            for (int j = 0; j < n; j++) w->A[(a + i) * n + j] *= e;
            w->b[a + i] *= e; w->E[a + i] *= e; }
    }
    free(dw); free(ew);
}

/* ---- KKT system K = [[P, A'], [A, -Hs]]: assembly, regularised dense LDL', solve with refinement */
static void ldl_solve(const Ws *w, const double *b, double *y) {
    const int N = w->N; const double *L = w->L;
    memcpy(y, b, sizeof(double) * N);
    for (int k = 0; k < N; k++) { const double yk = y[k]; for (int i = k + 1; i < N; i++) y[i] -= L[i * N + k] * yk; }
    for (int k = 0; k < N; k++) y[k] /= w->d[k];
    for (int k = N - 1; k >= 0; k--) { double s = 0; for (int i = k + 1; i < N; i++) s += L[i * N + k] * y[i]; y[k] -= s; }
}
static void kkt_update(Ws *w, const double *Hs /* m x m */) {
    const int n = w->n, m = w->m, N = w->N; const double *stg = w->stg;
    double *K = w->K;
    memset(K, 0, sizeof(double) * N * N);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) K[i * N + j] = w->P[i * n + j];
    for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) { K[(n + i) * N + j] = w->A[i * n + j]; K[j * N + n + i] = w->A[i * n + j]; }
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) K[(n + i) * N + n + j] = -Hs[i * m + j];
    double *Kw = w->L;                            /* factor in place of a copy: Kw becomes L column by column */
    memcpy(Kw, K, sizeof(double) * N * N);
    if (stg[C_SREG_ENABLE] != 0.0) {
        double mx = 0; for (int k = 0; k < N; k++) mx = dmaxd(mx, fabs(K[k * N + k]));
        const double eps = stg[C_SREG_CONST] + stg[C_SREG_PROP] * mx;
        for (int k = 0; k < N; k++) Kw[k * N + k] += eps * w->signs[k];
    }
    for (int k = 0; k < N; k++) {
        double dk = Kw[k * N + k];
        if (stg[C_DREG_ENABLE] != 0.0 && dk * w->signs[k] < stg[C_DREG_EPS]) dk = stg[C_DREG_DELTA] * w->signs[k];
        w->d[k] = dk;
        for (int i = k + 1; i < N; i++) Kw[i * N + k] = Kw[i * N + k] / dk;
        for (int i = k + 1; i < N; i++) { const double li = Kw[i * N + k];
            for (int j = k + 1; j < N; j++) Kw[i * N + j] -= (li * Kw[j * N + k]) * dk; }
    }
}
static void kkt_solve(Ws *w, const double *bx, const double *bz, double *ox, double *oz) {
    const int n = w->n, m = w->m, N = w->N; const double *stg = w->stg;
    double *b = w->t1, *x = w->t2, *e = w->t3, *xn = w->t4;
    memcpy(b, bx, sizeof(double) * n); memcpy(b + n, bz, sizeof(double) * m);
    ldl_solve(w, b, x);
    if (stg[C_IR_ENABLE] != 0.0) {
        double *corr = dv(N), *en = dv(N);
        const double normb = N ? ninf(b, N) : 0.0;
#define RESID(xx, out) do { for (int i_ = 0; i_ < N; i_++) { double s_ = 0; for (int j_ = 0; j_ < N; j_++) s_ += w->K[i_ * N + j_] * (xx)[j_]; (out)[i_] = b[i_] - s_; } } while (0)
        RESID(x, e);
        double norme = ninf(e, N);
        for (int it = 0; it < (int)stg[C_IR_MAX_ITER]; it++) {
            if (norme <= stg[C_IR_ABSTOL] + stg[C_IR_RELTOL] * normb) break;
            const double lastnorme = norme;
            ldl_solve(w, e, corr);
            for (int i = 0; i < N; i++) xn[i] = x[i] + corr[i];
            RESID(xn, en);
            norme = ninf(en, N);
            const double ratio = norme > 0 ? lastnorme / norme : INFINITY;
            if (ratio < stg[C_IR_STOP_RATIO]) {
                if (ratio > 1.0) { memcpy(x, xn, sizeof(double) * N); memcpy(e, en, sizeof(double) * N); }
                else norme = lastnorme;
                break;
            }
            memcpy(x, xn, sizeof(double) * N); memcpy(e, en, sizeof(double) * N);
        }
#undef RESID
        free(corr); free(en);
    }
    memcpy(ox, x, sizeof(double) * n); memcpy(oz, x + n, sizeof(double) * m);
}

/* ---- cones: margins, shift into the cone, Nesterov-Todd scaling */
static void margins(const Cones *c, const double *v, double *alpha, double *beta) {
    double a = INFINITY, b = 0;
    for (int i = c->zero; i < c->zero + c->nonneg; i++) { a = dmind(a, v[i]); b += dmaxd(v[i], 0.0); }
    for (int k = 0; k < c->nsoc; k++) { const double *s = v + c->start[k]; const double g = s[0] - sqrt(dot(s + 1, s + 1, c->soc[k] - 1)); a = dmind(a, g); b += dmaxd(0.0, g); }
    *alpha = a; *beta = b;
}
static void unit_shift(const Cones *c, double *v, double a, int primal) {
    for (int i = c->zero; i < c->zero + c->nonneg; i++) v[i] += a;
    for (int k = 0; k < c->nsoc; k++) v[c->start[k]] += a;
    if (primal) for (int i = 0; i < c->zero; i++) v[i] = 0.0;
}
static void shift_to_cone(const Cones *c, double *v, int primal) {
    if (c->degree == 0) { unit_shift(c, v, 0.0, primal); return; }
    double mn, pos; margins(c, v, &mn, &pos);
    const double target = dmaxd(1.0, 0.1 * pos / c->degree);
    if (mn <= 0.0) { unit_shift(c, v, -mn, primal); unit_shift(c, v, target, primal); }
    else if (mn < target) unit_shift(c, v, target - mn, primal);
    else unit_shift(c, v, 0.0, primal);
}
static void mul_W_cone(const Ws *w, int k, const double *v, double *out, int inv) {
    const int a = w->c.start[k], d = w->c.soc[k]; const double *sw = w->sw + a; const double eta = w->eta[k];
    const double zeta = dot(sw + 1, v + 1, d - 1);
    if (!inv) {
        out[0] = sw[0] * v[0] + zeta;
        const double f = v[0] + zeta / (1.0 + sw[0]);
        for (int i = 1; i < d; i++) out[i] = v[i] + f * sw[i];
        for (int i = 0; i < d; i++) out[i] = eta * out[i];
    } else {
        out[0] = sw[0] * v[0] - zeta;
        const double f = -v[0] + zeta / (1.0 + sw[0]);
        for (int i = 1; i < d; i++) out[i] = v[i] + f * sw[i];
        for (int i = 0; i < d; i++) out[i] = out[i] / eta;
    }
}
static int scaling_update(Ws *w, const double *s, const double *z) {
    const Cones *c = &w->c; int ok = 1;
    for (int i = c->zero; i < c->zero + c->nonneg; i++) { w->w[i] = sqrt(s[i] / z[i]); w->lam[i] = sqrt(s[i] * z[i]); }
    for (int k = 0; k < c->nsoc; k++) {
        const int a = c->start[k], d = c->soc[k]; const double *sk = s + a, *zk = z + a;
        const double rs = soc_res(sk, d), rz = soc_res(zk, d);
        if (!(rs > 0.0 && rz > 0.0)) { ok = 0; continue; }
        const double ss = sqrt(rs), zs = sqrt(rz);
        const double gamma = sqrt(0.5 * (1.0 + dot(sk, zk, d) / (ss * zs)));
        double *sw = w->sw + a;
        for (int i = 0; i < d; i++) sw[i] = sk[i] / (2.0 * ss * gamma);
        sw[0] += zk[0] / (2.0 * zs * gamma);
        for (int i = 1; i < d; i++) sw[i] -= zk[i] / (2.0 * zs * gamma);
        sw[0] = sqrt(1.0 + dot(sw + 1, sw + 1, d - 1));
        w->eta[k] = sqrt(ss / zs);
        mul_W_cone(w, k, zk, w->lam + a, 0);
    }
    return ok;
}
static void build_Hs(const Ws *w, double *H) {
    const Cones *c = &w->c; const int m = w->m;
    memset(H, 0, sizeof(double) * m * m);
    for (int i = c->zero; i < c->zero + c->nonneg; i++) H[i * m + i] = w->w[i] * w->w[i];
    for (int k = 0; k < c->nsoc; k++) { const int a = c->start[k], d = c->soc[k]; const double *sw = w->sw + a; const double e2 = w->eta[k] * w->eta[k];
        for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) {
            const double J = i == j ? (i == 0 ? 1.0 : -1.0) : 0.0;
            H[(a + i) * m + a + j] = e2 * (2.0 * (sw[i] * sw[j]) - J); } }
}
static void mul_Hs(const Ws *w, const double *v, double *out) {
    const Cones *c = &w->c;
    memset(out, 0, sizeof(double) * w->m);
    for (int i = c->zero; i < c->zero + c->nonneg; i++) out[i] = w->w[i] * w->w[i] * v[i];
    for (int k = 0; k < c->nsoc; k++) { const int a = c->start[k], d = c->soc[k]; const double *sw = w->sw + a, *vk = v + a; const double e2 = w->eta[k] * w->eta[k];
        const double t = 2.0 * dot(sw, vk, d);
        for (int i = 0; i < d; i++) { double o = t * sw[i]; if (i == 0) o -= vk[0]; else o += vk[i]; out[a + i] = e2 * o; } }
}
static void mul_W(const Ws *w, const double *v, double *out, int inv) {
    const Cones *c = &w->c;
    memset(out, 0, sizeof(double) * w->m);
    for (int i = c->zero; i < c->zero + c->nonneg; i++) out[i] = inv ? v[i] / w->w[i] : v[i] * w->w[i];
    for (int k = 0; k < c->nsoc; k++) mul_W_cone(w, k, v + c->start[k], out + c->start[k], inv);
}
static void circ(const Cones *c, int m, const double *a, const double *b, double *out) {
    memset(out, 0, sizeof(double) * m);
    for (int i = c->zero; i < c->zero + c->nonneg; i++) out[i] = a[i] * b[i];
    for (int k = 0; k < c->nsoc; k++) { const int s = c->start[k], d = c->soc[k];
        out[s] = dot(a + s, b + s, d);
        for (int i = 1; i < d; i++) out[s + i] = a[s] * b[s + i] + b[s] * a[s + i]; }
}
static void inv_circ_lam(const Ws *w, const double *dvv, double *out) {
    const Cones *c = &w->c;
    memset(out, 0, sizeof(double) * w->m);
    for (int i = c->zero; i < c->zero + c->nonneg; i++) out[i] = dvv[i] / w->lam[i];
    for (int k = 0; k < c->nsoc; k++) { const int s = c->start[k], d = c->soc[k]; const double *lam = w->lam + s, *dk = dvv + s;
        const double p = soc_res(lam, d);
        const double u0 = (lam[0] * dk[0] - dot(lam + 1, dk + 1, d - 1)) / p;
        out[s] = u0;
        for (int i = 1; i < d; i++) out[s + i] = (dk[i] - u0 * lam[i]) / lam[0]; }
}
static double step_length(const Cones *c, const double *v, const double *dvv, double amax) {
    double a = amax;
    for (int i = c->zero; i < c->zero + c->nonneg; i++) if (dvv[i] < 0.0) a = dmind(a, -v[i] / dvv[i]);
    for (int k = 0; k < c->nsoc; k++) { const int s = c->start[k], d = c->soc[k]; const double *x = v + s, *y = dvv + s;
        const double qa = soc_res(y, d), qb = 2.0 * (x[0] * y[0] - dot(x + 1, y + 1, d - 1)), qc = dmaxd(0.0, soc_res(x, d));
        const double disc = qb * qb - 4.0 * qa * qc;
        double r;
        if ((qa > 0.0 && qb > 0.0) || disc < 0.0) r = INFINITY;
        else if (qa == 0.0) r = INFINITY;
        else {
            const double t = qb >= 0.0 ? (-qb - sqrt(disc)) : (-qb + sqrt(disc));
            double r1 = t != 0.0 ? (2.0 * qc) / t : INFINITY, r2 = t / (2.0 * qa);
            if (r1 < 0.0) r1 = INFINITY;
            if (r2 < 0.0) r2 = INFINITY;
            r = dmind(r1, r2);
        }
        a = dmind(a, r); }
    return a;
}

typedef struct { double cost_p, cost_d, res_p, res_d, gap_abs, gap_rel, ktratio, res_pinf, res_dinf, dot_bz, dot_qx; } Info;
static int converged(const double *stg, int reduced, const Info *f) {
    const int o = reduced ? (C_RED_TOL_GAP_ABS - C_TOL_GAP_ABS) : 0;
    if (f->ktratio <= 1.0 && (f->gap_abs < stg[C_TOL_GAP_ABS + o] || f->gap_rel < stg[C_TOL_GAP_REL + o]) &&
        f->res_p < stg[C_TOL_FEAS + o] && f->res_d < stg[C_TOL_FEAS + o]) return reduced ? ST_ALMOST_SOLVED : ST_SOLVED;
    if (f->ktratio > 1000.0 / stg[C_TOL_KTRATIO + o]) {
        if (f->dot_bz < -stg[C_TOL_INFEAS_ABS + o] && f->res_pinf < -stg[C_TOL_INFEAS_REL + o] * f->dot_bz) return reduced ? ST_ALMOST_PINF : ST_PINF;
        if (f->dot_qx < -stg[C_TOL_INFEAS_ABS + o] && f->res_dinf < -stg[C_TOL_INFEAS_REL + o] * f->dot_qx) return reduced ? ST_ALMOST_DINF : ST_DINF;
    }
    return ST_UNSOLVED;
}

/* ---- one solve (clarabel_numpy.solve).  P (n x n, symmetric), A (m x n) dense row-major, overwritten. */
static void solve_one(int n, int m, double *P, double *q, double *A, double *b, const Cones *cones, int p_is_zero,
                      const double *stg, double *out_x, double *out_z, double *res /* obj, iter, status, r_prim, r_dual */) {
    Ws W; Ws *w = &W; memset(w, 0, sizeof(W));
    const int N = n + m;
    w->n = n; w->m = m; w->N = N; w->stg = stg; w->c = *cones;
    w->P = P; w->A = A; w->q = q; w->b = b;
    w->D = dv(n); w->E = dv(m); w->K = dv((size_t)N * N); w->L = dv((size_t)N * N); w->d = dv(N); w->signs = dv(N);
    w->w = dv(m); w->lam = dv(m); w->eta = dv(cones->nsoc); w->sw = dv(m);
    w->t1 = dv(N); w->t2 = dv(N); w->t3 = dv(N); w->t4 = dv(N);
    for (int k = 0; k < N; k++) w->signs[k] = k < n ? 1.0 : -1.0;
    const double normq = n ? ninf(q, n) : 0.0, normb = m ? ninf(b, m) : 0.0;
    equilibrate(w);
    const double c = w->cs;
    double *Dinv = dv(n), *Einv = dv(m);
    for (int j = 0; j < n; j++) Dinv[j] = 1.0 / w->D[j];
    for (int i = 0; i < m; i++) Einv[i] = 1.0 / w->E[i];
    /* identity scaling */
    for (int i = 0; i < m; i++) { w->w[i] = 1.0; w->lam[i] = 1.0; w->sw[i] = 0.0; }
    for (int k = 0; k < cones->nsoc; k++) { w->eta[k] = 1.0; w->sw[cones->start[k]] = 1.0; }
    double *Hs = dv((size_t)m * m), *x = dv(n), *z = dv(m), *s = dv(m), *zero_n = dv(n), *zero_m = dv(m), *nq = dv(n);
    double *px = dv(n), *pz = dv(m), *ps = dv(m);         /* previous iterate */
    double *Px = dv(n), *rx_inf = dv(n), *rz_inf = dv(m), *rx = dv(n), *rz = dv(m), *x2 = dv(n), *z2 = dv(m);
    double *x1 = dv(n), *z1 = dv(m), *dx = dv(n), *dz = dv(m), *ds = dv(m), *tmpm = dv(m), *tmpm2 = dv(m), *tmpm3 = dv(m), *tmpn = dv(n), *d_s = dv(m), *rhsx = dv(n), *rhsz = dv(m);
    for (int j = 0; j < n; j++) nq[j] = -q[j];
    for (int i = cones->zero; i < m; i++) Hs[i * m + i] = 1.0;
    kkt_update(w, Hs);
    if (!p_is_zero) { kkt_solve(w, nq, b, x, z); for (int i = 0; i < m; i++) s[i] = -z[i]; }
    else { kkt_solve(w, zero_n, b, x, s); for (int i = 0; i < m; i++) s[i] = -s[i]; kkt_solve(w, nq, zero_m, tmpn, z); }
    shift_to_cone(cones, s, 1);
    shift_to_cone(cones, z, 0);
    double tau = 1.0, kap = 1.0, ptau = 1.0, pkap = 1.0;
    memcpy(px, x, sizeof(double) * n); memcpy(pz, z, sizeof(double) * m); memcpy(ps, s, sizeof(double) * m);
    int status = ST_UNSOLVED, it = 0;
    Info info, prev; memset(&info, 0, sizeof(info));
    prev.res_p = prev.res_d = prev.gap_abs = prev.gap_rel = prev.cost_p = prev.cost_d = INFINITY;

#define MATVEC_P(xx, out) do { for (int i_ = 0; i_ < n; i_++) { double s_ = 0; for (int j_ = 0; j_ < n; j_++) s_ += w->P[i_ * n + j_] * (xx)[j_]; (out)[i_] = s_; } } while (0)
    for (;;) {
        /* residuals */
        MATVEC_P(x, Px);
        for (int j = 0; j < n; j++) { double t = 0; for (int i = 0; i < m; i++) t += w->A[i * n + j] * z[i]; rx_inf[j] = -t; }
        for (int i = 0; i < m; i++) { double t = 0; for (int j = 0; j < n; j++) t += w->A[i * n + j] * x[j]; rz_inf[i] = t + s[i]; }
        const double dot_qx = dot(w->q, x, n), dot_bz = dot(w->b, z, m), dot_sz = dot(s, z, m), xPx = dot(x, Px, n);
        for (int j = 0; j < n; j++) rx[j] = rx_inf[j] - Px[j] - w->q[j] * tau;
        for (int i = 0; i < m; i++) rz[i] = rz_inf[i] - w->b[i] * tau;
        const double rtau = dot_qx + dot_bz + kap + xPx / tau;
        const double mu = (dot_sz + tau * kap) / (cones->degree + 1);
        const double tinv = 1.0 / tau, cinv = 1.0 / c;
        info.cost_p = (dot_qx * tinv + 0.5 * xPx * tinv * tinv) * cinv;
        info.cost_d = (-dot_bz * tinv - 0.5 * xPx * tinv * tinv) * cinv;
        double normx = sninf(w->D, x, n), normz = sninf(w->E, z, m) * cinv, norms = sninf(Einv, s, m);
        info.res_pinf = sninf(Dinv, rx_inf, n) / dmaxd(1.0, normz);
        info.res_dinf = dmaxd(sninf(Dinv, Px, n) / dmaxd(1.0, normx), sninf(Einv, rz_inf, m) / dmaxd(1.0, normx + norms));
        normx *= tinv; normz *= tinv; norms *= tinv;
        info.res_p = sninf(Einv, rz, m) * tinv / dmaxd(1.0, normb + normx + norms);
        info.res_d = sninf(Dinv, rx, n) * tinv * cinv / dmaxd(1.0, normq + normx + normz);
        info.gap_abs = fabs(info.cost_p - info.cost_d);
        info.gap_rel = info.gap_abs / dmaxd(1.0, dmind(fabs(info.cost_p), fabs(info.cost_d)));
        info.ktratio = kap / tau;
        info.dot_bz = dot_bz * cinv; info.dot_qx = dot_qx * cinv;
        const double res_p = info.res_p, res_d = info.res_d;
        status = converged(stg, 0, &info);
        if (status == ST_UNSOLVED && it > 1 && (res_d > prev.res_d || res_p > prev.res_p)) {
            if (info.ktratio < 100.0 * 2.220446049250313e-16 && (prev.gap_abs < stg[C_TOL_GAP_ABS] || prev.gap_rel < stg[C_TOL_GAP_REL])) status = ST_INSUFFICIENT;
            if ((res_d > stg[C_TOL_FEAS] && res_d > 100.0 * prev.res_d) || (res_p > stg[C_TOL_FEAS] && res_p > 100.0 * prev.res_p)) status = ST_INSUFFICIENT;
            if (status == ST_INSUFFICIENT) {
                memcpy(x, px, sizeof(double) * n); memcpy(z, pz, sizeof(double) * m); memcpy(s, ps, sizeof(double) * m); tau = ptau; kap = pkap;
                info.cost_p = prev.cost_p; info.cost_d = prev.cost_d; info.res_p = prev.res_p; info.res_d = prev.res_d;
                info.gap_abs = prev.gap_abs; info.gap_rel = prev.gap_rel;
            }
        }
        if (status == ST_UNSOLVED && it >= (int)stg[C_MAX_ITER]) status = ST_MAX_ITER;
        if (status != ST_UNSOLVED) break;
        prev = info;
        it++;
        if (!scaling_update(w, s, z)) { status = ST_NUMERICAL; break; }
        build_Hs(w, Hs);
        kkt_update(w, Hs);
        kkt_solve(w, nq, w->b, x2, z2);
        double dtau = 0, dkap = 0, alpha = 0;
        for (int pass = 0; pass < 2; pass++) {
            /* pass 0: affine step (rhs = residuals, ds_const = s); pass 1: combined step */
            double rhs_tau, rhs_kap; const double *ds_const;
            if (pass == 0) { memcpy(rhsx, rx, sizeof(double) * n); memcpy(rhsz, rz, sizeof(double) * m); rhs_tau = rtau; rhs_kap = tau * kap; ds_const = s; }
            else {
                const double sigma = pow(1.0 - alpha, 3.0);
                mul_W(w, ds, tmpm, 1); mul_W(w, dz, tmpm2, 0);
                circ(cones, m, tmpm, tmpm2, tmpm3);                        /* shift */
                circ(cones, m, w->lam, w->lam, tmpm);
                for (int i = 0; i < m; i++) d_s[i] = tmpm[i] + tmpm3[i];
                for (int i = cones->zero; i < cones->zero + cones->nonneg; i++) d_s[i] -= sigma * mu;
                for (int k = 0; k < cones->nsoc; k++) d_s[cones->start[k]] -= sigma * mu;
                for (int i = 0; i < cones->zero; i++) d_s[i] = 0.0;
                rhs_kap = -sigma * mu + dtau * dkap + tau * kap;
                for (int j = 0; j < n; j++) rhsx[j] = (1.0 - sigma) * rx[j];
                for (int i = 0; i < m; i++) rhsz[i] = (1.0 - sigma) * rz[i];
                rhs_tau = (1.0 - sigma) * rtau;
                inv_circ_lam(w, d_s, tmpm); mul_W(w, tmpm, tmpm2, 0);     /* ds_offset = W (lambda \ d_s) */
                memcpy(d_s, tmpm2, sizeof(double) * m);
                ds_const = d_s;
            }
            for (int i = 0; i < m; i++) tmpm[i] = ds_const[i] - rhsz[i];
            kkt_solve(w, rhsx, tmpm, x1, z1);
            /* dtau from the two solves */
            double num = rhs_tau - rhs_kap / tau + dot(w->q, x1, n) + dot(w->b, z1, m);
            { MATVEC_P(x1, tmpn); double t = 0; for (int j = 0; j < n; j++) t += (x[j] / tau) * tmpn[j]; num += 2.0 * t; }
            double den = kap / tau - dot(w->q, x2, n) - dot(w->b, z2, m);
            { double *xm = rhsx;                   /* (rhsx is free from here on in this pass) */
              for (int j = 0; j < n; j++) xm[j] = x[j] / tau - x2[j];
              MATVEC_P(xm, tmpn); den += dot(xm, tmpn, n);
              MATVEC_P(x2, tmpn); den -= dot(x2, tmpn, n); }
            dtau = num / den;
            for (int j = 0; j < n; j++) dx[j] = x1[j] + dtau * x2[j];
            for (int i = 0; i < m; i++) dz[i] = z1[i] + dtau * z2[i];
            mul_Hs(w, dz, tmpm);
            for (int i = 0; i < m; i++) ds[i] = -(tmpm[i] + ds_const[i]);
            dkap = -(rhs_kap + kap * dtau) / tau;
            double a = 1.0;
            if (dtau < 0.0) a = dmind(a, -tau / dtau);
            if (dkap < 0.0) a = dmind(a, -kap / dkap);
            a = dmind(step_length(cones, z, dz, a), step_length(cones, s, ds, a));
            alpha = pass ? a * stg[C_MAX_STEP_FRACTION] : a;
        }
        if (alpha <= dmaxd(0.0, stg[C_MIN_TERMINATE])) { status = ST_INSUFFICIENT; break; }
        memcpy(px, x, sizeof(double) * n); memcpy(pz, z, sizeof(double) * m); memcpy(ps, s, sizeof(double) * m); ptau = tau; pkap = kap;
        for (int j = 0; j < n; j++) x[j] = x[j] + alpha * dx[j];
        for (int i = 0; i < m; i++) { s[i] = s[i] + alpha * ds[i]; z[i] = z[i] + alpha * dz[i]; }
        tau += alpha * dtau; kap += alpha * dkap;
    }
#undef MATVEC_P
    if (status == ST_NUMERICAL || status == ST_INSUFFICIENT || status == ST_MAX_ITER) {
        const int almost = converged(stg, 1, &info);
        if (almost != ST_UNSOLVED) status = almost;
    }
    double scale, obj;
    if (status == ST_PINF || status == ST_ALMOST_PINF || status == ST_DINF || status == ST_ALMOST_DINF) { scale = 1.0; obj = NAN; }
    else { scale = 1.0 / tau; obj = info.cost_p; }
    for (int j = 0; j < n; j++) out_x[j] = w->D[j] * x[j] * scale;
    for (int i = 0; i < m; i++) out_z[i] = w->E[i] * z[i] * scale / c;
    res[0] = obj; res[1] = it; res[2] = status; res[3] = info.res_p; res[4] = info.res_d;
    double *fr[] = {w->D, w->E, w->K, w->L, w->d, w->signs, w->w, w->lam, w->eta, w->sw, w->t1, w->t2, w->t3, w->t4, Dinv, Einv, Hs, x, z, s,
                    zero_n, zero_m, nq, px, pz, ps, Px, rx_inf, rz_inf, rx, rz, x2, z2, x1, z1, dx, dz, ds, tmpm, tmpm2, tmpm3, tmpn, d_s, rhsx, rhsz};
    for (size_t k = 0; k < sizeof(fr) / sizeof(fr[0]); k++) free(fr[k]);
}

/* ---- cpg_solve() of a conic family for a batch: per instance canonicalise, NEW solver, solve, retrieve
 *      (cvxpygen/solvers/clarabel.py:172-204, cvxpygen/utils.py:1032-1052).
 *   maps: 5 CSR maps over [theta; 1] in the order P, q, d, A, b (P: upper-triangle entries in CSC order of (Pp, Pi))
 *   out: sol_x B x n, sol_z B x m, info B x 5 = (obj_val, iterations, status, r_prim, r_dual) */
int clarabel_oracle_solve_batch(int n, int m, int n_zero, int n_nonneg, int n_soc, const int *soc_dims,
                                const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                                const int *map_rows, const int *const *map_p, const int *const *map_i, const double *const *map_x,
                                int is_max, int nonzero_d, int NP1, long B, const double *theta, const double *settings,
                                double *sol_x, double *sol_z, double *info, int nthreads) {
    Cones c; c.zero = n_zero; c.nonneg = n_nonneg; c.nsoc = n_soc; c.soc = soc_dims;
    c.start = (int *)calloc(n_soc ? n_soc : 1, sizeof(int));
    int o = n_zero + n_nonneg;
    for (int k = 0; k < n_soc; k++) { c.start[k] = o; o += soc_dims[k]; }
    c.m = o; c.degree = n_nonneg + n_soc;
    if (o != m) { free(c.start); return 1; }
    const int nnzP = Pp[n], nnzA = Ap[n];
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        double *P = dv((size_t)n * n), *A = dv((size_t)m * n), *q = dv(n), *b = dv(m), *vals = dv((size_t)(nnzP > nnzA ? nnzP : nnzA));
#pragma omp for schedule(dynamic, 4)
        for (long k = 0; k < B; k++) {
            const double *th = theta + (size_t)k * NP1; double d = 0;
#define CANON(id, out) do { for (int r_ = 0; r_ < map_rows[id]; r_++) { double s_ = 0; for (int e_ = map_p[id][r_]; e_ < map_p[id][r_ + 1]; e_++) s_ += map_x[id][e_] * th[map_i[id][e_]]; (out)[r_] = s_; } } while (0)
            memset(P, 0, sizeof(double) * n * n); memset(A, 0, sizeof(double) * m * n);
            if (map_rows[0]) { CANON(0, vals); for (int j = 0; j < n; j++) for (int e = Pp[j]; e < Pp[j + 1]; e++) { P[Pi[e] * n + j] = vals[e]; P[j * n + Pi[e]] = vals[e]; } }
            CANON(1, q);
            if (nonzero_d && map_rows[2]) CANON(2, &d);
            CANON(3, vals);
            for (int j = 0; j < n; j++) for (int e = Ap[j]; e < Ap[j + 1]; e++) A[Ai[e] * n + j] = vals[e];
            CANON(4, b);
#undef CANON
            double *io = info + (size_t)k * 5;
            solve_one(n, m, P, q, A, b, &c, nnzP == 0, settings, sol_x + (size_t)k * n, sol_z + (size_t)k * m, io);
            double ov = io[0] + d; if (is_max) ov = -ov;
            io[0] = ov;
        }
        free(P); free(A); free(q); free(b); free(vals);
    }
    free(c.start);
    return 0;
}
