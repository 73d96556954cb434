/*
 * TEST INFRASTRUCTURE ONLY.  Nothing under cvxpygen_amd/ may include, link or call this file;
 * it is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Scalar-C, double-precision restatement of the solver that cvxpygen's generated cpg_solve()
 * runs for the OSQP backend:
 *     cpg_canonicalize_<p>   (cvxpygen/utils.py:279-294, 937-944)        -> oracle_canonicalize
 *     osqp_update_data_mat / osqp_update_data_vec (cvxpygen/solvers/osqp.py:20-61)
 *                                                                        -> oracle_update_mat/_vec
 *     osqp_solve             (cvxpygen/solvers/osqp.py:62)               -> oracle_solve
 *     cpg_retrieve_info      (cvxpygen/utils.py:977-985)                 -> oracle_cpg_solve_batch
 *     setup at code-generation time: osqp.OSQP().setup(P,q,A,l,u)
 *                            (cvxpygen/solvers/osqp.py:126-131)          -> oracle_setup
 *
 * The OSQP / QDLDL / AMD sources are a THIRD-PARTY dependency that is absent from
 * /root/reference (PyPI `osqp >= 1.0.0b3`, pyproject.toml:26; git submodule
 * cvxpygen/solvers/osqp-python is an empty directory; commit unpinned in .gitmodules).  The
 * algorithm is therefore restated from its published description: B. Stellato, G. Banjac,
 * P. Goulart, A. Bemporad, S. Boyd, "OSQP: an operator splitting solver for quadratic programs",
 * Math. Prog. Comp. 12 (2020) -- Algorithm 1 (ADMM), sec. 3.4 (termination), sec. 3.4/4
 * (infeasibility), sec. 5.1 (Ruiz equilibration, Algorithm 2), sec. 5.2 (rho selection / adaptive
 * rho) -- with the OSQP 1.0 default settings (rho 0.1, sigma 1e-6, alpha 1.6, scaling 10,
 * infinity 1e30) and QDLDL's documented scheme (elimination tree + up-looking LDL' on the upper
 * triangle, no pivoting).  The fill-reducing ordering is a plain minimum-degree ordering, not
 * SuiteSparse AMD: any permutation gives the same iterates up to rounding.
 *
 * PARITY UNPINNED: the reference's tests hold no golden vectors for this path (they compare with
 * a live cvxpy solve at 10 % tolerance, tests/test_E2E_QP.py:205-216).  This restatement is pinned
 * by independent mathematics only: exact NNLS / BVLS answers (tests/golden/), KKT-residual
 * properties, and agreement with the separately written dense numpy restatement
 * (oracle/osqp_numpy.py).
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_INFTY 1e30
#define ORC_MIN_SCALING 1e-4
#define ORC_MAX_SCALING 1e4
#define ORC_RHO_MIN 1e-6
#define ORC_RHO_MAX 1e6
#define ORC_RHO_TOL 1e-4
#define ORC_RHO_EQ_OVER_INEQ 1e3
#define ORC_DIV_TOL (1.0 / ORC_INFTY)

/* status codes (OSQP 1.0 numbering) */
enum { ST_SOLVED = 1, ST_SOLVED_INACC = 2, ST_PINF = 3, ST_PINF_INACC = 4, ST_DINF = 5,
       ST_DINF_INACC = 6, ST_MAX_ITER = 7, ST_NON_CVX = 9, ST_UNSOLVED = 11 };

/* settings vector layout (doubles), shared with tests/oracle_binding.py */
enum { S_RHO = 0, S_SIGMA, S_ALPHA, S_SCALING, S_MAX_ITER, S_EPS_ABS, S_EPS_REL, S_EPS_PINF,
       S_EPS_DINF, S_SCALED_TERM, S_CHECK_TERM, S_WARM, S_ADAPT_RHO, S_ADAPT_INT, S_ADAPT_TOL,
       S_CHECK_GAP, S_COUNT };

typedef struct {
    int n, m, N, nnzP, nnzA;
    double stg[S_COUNT];
    /* patterns (CSC; P upper triangular) */
    int *Pp, *Pi, *Ap, *Ai;
    /* unscaled and scaled data */
    double *P0, *A0, *q0, *l0, *u0;
    double *Px, *Ax, *q, *l, *u;
    double *D, *E, *Dinv, *Einv, c, cinv;
    double rho, *rho_vec, *rho_inv; int *ctype;
    /* KKT (upper CSC, permuted) with value-source maps */
    int nnzK; int *Kp, *Ki; double *Kx;
    int *P2K, *A2K, *sig2K, *rho2K;
    int *perm;                     /* permuted index k  <- original index perm[k] */
    /* factor */
    int *etree, *Lnz, *Lp, *Li, nnzL; double *Lx, *Dg, *Dginv;
    int *iw; unsigned char *bw; double *fw;
    /* iterates */
    double *x, *z, *y, *x_prev, *z_prev, *xz, *dx, *dy, *tAx, *tPx, *tAty, *bp, *tn, *tm;
    /* info */
    int iter, status; double obj_val, prim_res, dual_res, sc_prim_res, sc_dual_res;
    double dual_obj, gap;
    int n_refactor;
} OracleWS;

static double *dvec(int n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }
static int *ivec(int n) { return (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int)); }
static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }
static double norm_inf(const double *v, int n) {
    double r = 0; for (int i = 0; i < n; i++) r = dmax(r, fabs(v[i])); return r; }
static double snorm_inf(const double *s, const double *v, int n) {
    double r = 0; for (int i = 0; i < n; i++) r = dmax(r, fabs(s[i] * v[i])); return r; }
static double limit_scaling(double v) {
    v = v < ORC_MIN_SCALING ? 1.0 : v; return v > ORC_MAX_SCALING ? ORC_MAX_SCALING : v; }

/* ------------------------------------------------------------------ sparse helpers */
static void spmv_csc(int ncol, const int *p, const int *i, const double *x, const double *v,
                     double *out, int nrow) { /* out = M v */
    for (int r = 0; r < nrow; r++) out[r] = 0;
    for (int j = 0; j < ncol; j++) { double vj = v[j];
        for (int k = p[j]; k < p[j + 1]; k++) out[i[k]] += x[k] * vj; }
}
static void spmtv_csc(int ncol, const int *p, const int *i, const double *x, const double *v,
                      double *out) { /* out = M' v */
    for (int j = 0; j < ncol; j++) { double s = 0;
        for (int k = p[j]; k < p[j + 1]; k++) s += x[k] * v[i[k]]; out[j] = s; }
}
static void symv_triu(int n, const int *p, const int *i, const double *x, const double *v,
                      double *out) { /* out = P v, P symmetric stored as upper triangle */
    for (int r = 0; r < n; r++) out[r] = 0;
    for (int j = 0; j < n; j++)
        for (int k = p[j]; k < p[j + 1]; k++) { int r = i[k];
            out[r] += x[k] * v[j]; if (r != j) out[j] += x[k] * v[r]; }
}

/* ------------------------------------------------------------------ Ruiz equilibration */
/* OSQP paper Algorithm 2 on M = [[P, A'], [A, 0]], `iters` passes, then cost scaling. */
static void scale_data(OracleWS *w) {
    int n = w->n, m = w->m, iters = (int)w->stg[S_SCALING];
    memcpy(w->Px, w->P0, sizeof(double) * w->nnzP);
    memcpy(w->Ax, w->A0, sizeof(double) * w->nnzA);
    memcpy(w->q, w->q0, sizeof(double) * n);
    for (int i = 0; i < n; i++) w->D[i] = 1.0;
    for (int i = 0; i < m; i++) w->E[i] = 1.0;
    w->c = 1.0;
    double *Dt = w->tn, *Et = w->tm;
    for (int it = 0; it < iters; it++) {
        for (int j = 0; j < n; j++) Dt[j] = 0;
        for (int i = 0; i < m; i++) Et[i] = 0;
        for (int j = 0; j < n; j++)
            for (int k = w->Pp[j]; k < w->Pp[j + 1]; k++) { double a = fabs(w->Px[k]); int r = w->Pi[k];
                Dt[j] = dmax(Dt[j], a); if (r != j) Dt[r] = dmax(Dt[r], a); }
        for (int j = 0; j < n; j++)
            for (int k = w->Ap[j]; k < w->Ap[j + 1]; k++) { double a = fabs(w->Ax[k]);
                Dt[j] = dmax(Dt[j], a); Et[w->Ai[k]] = dmax(Et[w->Ai[k]], a); }
        for (int j = 0; j < n; j++) Dt[j] = 1.0 / sqrt(limit_scaling(Dt[j]));
        for (int i = 0; i < m; i++) Et[i] = 1.0 / sqrt(limit_scaling(Et[i]));
        for (int j = 0; j < n; j++)
            for (int k = w->Pp[j]; k < w->Pp[j + 1]; k++) w->Px[k] *= Dt[w->Pi[k]] * Dt[j];
        for (int j = 0; j < n; j++)
            for (int k = w->Ap[j]; k < w->Ap[j + 1]; k++) w->Ax[k] *= Et[w->Ai[k]] * Dt[j];
        for (int j = 0; j < n; j++) { w->q[j] *= Dt[j]; w->D[j] *= Dt[j]; }
        for (int i = 0; i < m; i++) w->E[i] *= Et[i];
        /* cost scaling: mean column inf-norm of P vs ||q||_inf */
        for (int j = 0; j < n; j++) Dt[j] = 0;
        for (int j = 0; j < n; j++)
            for (int k = w->Pp[j]; k < w->Pp[j + 1]; k++) { double a = fabs(w->Px[k]); int r = w->Pi[k];
                Dt[j] = dmax(Dt[j], a); if (r != j) Dt[r] = dmax(Dt[r], a); }
        double cm = 0; for (int j = 0; j < n; j++) cm += Dt[j]; if (n) cm /= n;
        double qn = limit_scaling(norm_inf(w->q, n));
        double ct = 1.0 / limit_scaling(dmax(cm, qn));
        for (int k = 0; k < w->nnzP; k++) w->Px[k] *= ct;
        for (int j = 0; j < n; j++) w->q[j] *= ct;
        w->c *= ct;
    }
    for (int j = 0; j < n; j++) w->Dinv[j] = 1.0 / w->D[j];
    for (int i = 0; i < m; i++) { w->Einv[i] = 1.0 / w->E[i];
        w->l[i] = w->E[i] * w->l0[i]; w->u[i] = w->E[i] * w->u0[i]; }
    w->cinv = 1.0 / w->c;
}

/* ------------------------------------------------------------------ rho vector */
/* OSQP's update_rho_vec: re-classify the rows; returns 1 when a constraint changed class (equality /
 * inequality / free).  Only then is rho_vec rebuilt -- from settings->rho (w->stg[S_RHO]), which every
 * cpg_solve of the reference has reset to the library default (osqp_set_default_settings writes the settings
 * struct only, cvxpygen/solvers/osqp.py:100-101), not from the rho the workspace's last adapt_rho left
 * (w->rho, what rho_vec and the factor carry until then). */
static int set_rho_vec(OracleWS *w) {
    int changed = 0;
    for (int i = 0; i < w->m; i++) {
        int t;
        if (w->l[i] < -ORC_INFTY * ORC_MIN_SCALING && w->u[i] > ORC_INFTY * ORC_MIN_SCALING) t = -1;
        else if (w->u[i] - w->l[i] < ORC_RHO_TOL) t = 1;
        else t = 0;
        if (t != w->ctype[i]) changed = 1;
        w->ctype[i] = t;
    }
    if (!changed) return 0;
    w->rho = w->stg[S_RHO] = dmin(dmax(w->stg[S_RHO], ORC_RHO_MIN), ORC_RHO_MAX);
    for (int i = 0; i < w->m; i++) {
        w->rho_vec[i] = w->ctype[i] == -1 ? ORC_RHO_MIN : (w->ctype[i] == 1 ? ORC_RHO_EQ_OVER_INEQ * w->rho : w->rho);
        w->rho_inv[i] = 1.0 / w->rho_vec[i]; }
    return 1;
}

/* ------------------------------------------------------------------ ordering */
/* Minimum degree on the explicit elimination graph held as bit rows; ties -> lowest index. */
static void min_degree_order(int N, const int *Kp, const int *Ki, int *perm) {
    int W = (N + 63) / 64;
    unsigned long long *adj = (unsigned long long *)calloc((size_t)N * W, sizeof(unsigned long long));
    int *deg = ivec(N); unsigned char *gone = (unsigned char *)calloc(N, 1);
    for (int j = 0; j < N; j++)
        for (int k = Kp[j]; k < Kp[j + 1]; k++) { int i = Ki[k]; if (i == j) continue;
            adj[(size_t)i * W + j / 64] |= 1ULL << (j % 64); adj[(size_t)j * W + i / 64] |= 1ULL << (i % 64); }
    for (int v = 0; v < N; v++) { int d = 0;
        for (int t = 0; t < W; t++) d += __builtin_popcountll(adj[(size_t)v * W + t]); deg[v] = d; }
    for (int step = 0; step < N; step++) {
        int best = -1;
        for (int v = 0; v < N; v++) if (!gone[v] && (best < 0 || deg[v] < deg[best])) best = v;
        perm[step] = best; gone[best] = 1;
        unsigned long long *nb = adj + (size_t)best * W;
        for (int t = 0; t < W; t++) { unsigned long long bits = nb[t];
            while (bits) { int b = __builtin_ctzll(bits); bits &= bits - 1; int u = t * 64 + b;
                unsigned long long *ru = adj + (size_t)u * W;
                for (int s = 0; s < W; s++) ru[s] |= nb[s];
                ru[u / 64] &= ~(1ULL << (u % 64)); ru[best / 64] &= ~(1ULL << (best % 64));
                int d = 0; for (int s = 0; s < W; s++) d += __builtin_popcountll(ru[s]); deg[u] = d; } }
    }
    free(adj); free(deg); free(gone);
}

/* ------------------------------------------------------------------ KKT assembly */
/* K = [[P + sigma I, A'], [A, -diag(1/rho)]] permuted symmetrically, upper triangle, CSC. */
typedef struct { int r, c, tag, src; } Trip;
static int trip_cmp(const void *a, const void *b) {
    const Trip *x = (const Trip *)a, *y = (const Trip *)b;
    if (x->c != y->c) return x->c - y->c; if (x->r != y->r) return x->r - y->r; return x->tag - y->tag; }

static void build_kkt(OracleWS *w) {
    int n = w->n, m = w->m, N = w->N;
    int cap = w->nnzP + n + w->nnzA + m, t = 0;
    Trip *T = (Trip *)malloc(sizeof(Trip) * (size_t)cap);
    /* natural-order pattern first, to compute the ordering */
    for (int j = 0; j < n; j++) for (int k = w->Pp[j]; k < w->Pp[j + 1]; k++) T[t++] = (Trip){w->Pi[k], j, 0, k};
    for (int j = 0; j < n; j++) T[t++] = (Trip){j, j, 1, j};
    for (int j = 0; j < n; j++) for (int k = w->Ap[j]; k < w->Ap[j + 1]; k++) T[t++] = (Trip){j, n + w->Ai[k], 2, k};
    for (int i = 0; i < m; i++) T[t++] = (Trip){n + i, n + i, 3, i};
    /* ordering on the natural pattern */
    { int *cp = ivec(N + 1), *ci = ivec(cap), *fill = ivec(N);
      for (int k = 0; k < t; k++) cp[T[k].c + 1]++;
      for (int j = 0; j < N; j++) cp[j + 1] += cp[j];
      for (int k = 0; k < t; k++) ci[cp[T[k].c] + fill[T[k].c]++] = T[k].r;
      min_degree_order(N, cp, ci, w->perm); free(cp); free(ci); free(fill); }
    int *pinv = ivec(N); for (int k = 0; k < N; k++) pinv[w->perm[k]] = k;
    for (int k = 0; k < t; k++) { int r = pinv[T[k].r], c = pinv[T[k].c];
        if (r > c) { int s = r; r = c; c = s; } T[k].r = r; T[k].c = c; }
    qsort(T, (size_t)t, sizeof(Trip), trip_cmp);
    w->Kp = ivec(N + 1); w->Ki = ivec(t); w->Kx = dvec(t);
    w->P2K = ivec(w->nnzP); w->A2K = ivec(w->nnzA); w->sig2K = ivec(n); w->rho2K = ivec(m);
    int nz = -1, pr = -1, pc = -1;
    for (int k = 0; k < t; k++) {
        if (T[k].r != pr || T[k].c != pc) { nz++; w->Ki[nz] = T[k].r; w->Kp[T[k].c + 1]++; pr = T[k].r; pc = T[k].c; }
        switch (T[k].tag) { case 0: w->P2K[T[k].src] = nz; break; case 1: w->sig2K[T[k].src] = nz; break;
                            case 2: w->A2K[T[k].src] = nz; break; default: w->rho2K[T[k].src] = nz; }
    }
    w->nnzK = nz + 1;
    for (int j = 0; j < N; j++) w->Kp[j + 1] += w->Kp[j];
    free(T); free(pinv);
}
static void fill_kkt(OracleWS *w) {
    for (int k = 0; k < w->nnzK; k++) w->Kx[k] = 0;
    for (int k = 0; k < w->nnzP; k++) w->Kx[w->P2K[k]] += w->Px[k];
    for (int j = 0; j < w->n; j++) w->Kx[w->sig2K[j]] += w->stg[S_SIGMA];
    for (int k = 0; k < w->nnzA; k++) w->Kx[w->A2K[k]] += w->Ax[k];
    for (int i = 0; i < w->m; i++) w->Kx[w->rho2K[i]] += -w->rho_inv[i];
}

/* ------------------------------------------------------------------ LDL' (QDLDL scheme) */
/* elimination tree and column counts of L from the upper triangle (Liu's algorithm) */
static int ldl_etree(int N, const int *Kp, const int *Ki, int *work, int *Lnz, int *etree) {
    for (int i = 0; i < N; i++) { work[i] = 0; Lnz[i] = 0; etree[i] = -1; if (Kp[i] == Kp[i + 1]) return -1; }
    for (int j = 0; j < N; j++) { work[j] = j;
        for (int p = Kp[j]; p < Kp[j + 1]; p++) { int i = Ki[p]; if (i > j) return -1;
            while (work[i] != j) { if (etree[i] == -1) etree[i] = j; Lnz[i]++; work[i] = j; i = etree[i]; } } }
    int s = 0; for (int i = 0; i < N; i++) s += Lnz[i]; return s;
}
/* up-looking numeric factorisation: row k of L is obtained by a sparse triangular solve with
 * the already computed leading block; the non-zero pattern of the row is the reach of column k
 * of K in the elimination tree.  Returns the number of positive pivots, or -1 on a zero pivot. */
static int ldl_factor(OracleWS *w) {
    int N = w->N; const int *Kp = w->Kp, *Ki = w->Ki; const double *Kx = w->Kx;
    int *Lp = w->Lp, *Li = w->Li; double *Lx = w->Lx, *D = w->Dg, *Dinv = w->Dginv;
    int *ybuf = w->iw, *ebuf = w->iw + N, *nextcol = w->iw + 2 * N; unsigned char *mark = w->bw; double *yv = w->fw;
    int pos = 0;
    Lp[0] = 0;
    for (int i = 0; i < N; i++) { Lp[i + 1] = Lp[i] + w->Lnz[i]; mark[i] = 0; yv[i] = 0; D[i] = 0; nextcol[i] = Lp[i]; }
    for (int k = 0; k < N; k++) {
        int ny = 0;
        for (int p = Kp[k]; p < Kp[k + 1]; p++) {
            int b = Ki[p];
            if (b == k) { D[k] = Kx[p]; continue; }
            yv[b] = Kx[p];
            if (!mark[b]) { /* climb the tree from b, push the path in topological order */
                mark[b] = 1; ebuf[0] = b; int ne = 1; int nx = w->etree[b];
                while (nx != -1 && nx < k) { if (mark[nx]) break; mark[nx] = 1; ebuf[ne++] = nx; nx = w->etree[nx]; }
                while (ne) ybuf[ny++] = ebuf[--ne];
            }
        }
        for (int t = ny - 1; t >= 0; t--) {
            int c = ybuf[t]; int top = nextcol[c]; double yc = yv[c];
            for (int j = Lp[c]; j < top; j++) yv[Li[j]] -= Lx[j] * yc;
            Li[top] = k; Lx[top] = yc * Dinv[c]; D[k] -= yc * Lx[top]; nextcol[c]++;
            yv[c] = 0; mark[c] = 0;
        }
        if (D[k] == 0.0) return -1;
        if (D[k] > 0) pos++;
        Dinv[k] = 1.0 / D[k];
    }
    return pos;
}
static void ldl_solve(const OracleWS *w, double *x) {
    int N = w->N; const int *Lp = w->Lp, *Li = w->Li; const double *Lx = w->Lx;
    for (int i = 0; i < N; i++) { double v = x[i]; for (int j = Lp[i]; j < Lp[i + 1]; j++) x[Li[j]] -= Lx[j] * v; }
    for (int i = 0; i < N; i++) x[i] *= w->Dginv[i];
    for (int i = N - 1; i >= 0; i--) { double v = x[i]; for (int j = Lp[i]; j < Lp[i + 1]; j++) v -= Lx[j] * x[Li[j]]; x[i] = v; }
}
static int refactor(OracleWS *w) { fill_kkt(w); w->n_refactor++; return ldl_factor(w) < 0 ? -1 : 0; }

/* ------------------------------------------------------------------ public: setup */
void oracle_default_settings(double *s) {
    s[S_RHO] = 0.1; s[S_SIGMA] = 1e-6; s[S_ALPHA] = 1.6; s[S_SCALING] = 10; s[S_MAX_ITER] = 4000;
    s[S_EPS_ABS] = 1e-3; s[S_EPS_REL] = 1e-3; s[S_EPS_PINF] = 1e-4; s[S_EPS_DINF] = 1e-4;
    /* OSQP >= 1.0 library defaults (what the reference's generated code is linked with, pyproject.toml:26):
     * rho adapted every 50 iterations when the estimate leaves [rho / 5, 5 rho], duality-gap test */
    s[S_SCALED_TERM] = 0; s[S_CHECK_TERM] = 25; s[S_WARM] = 1; s[S_ADAPT_RHO] = 1; s[S_ADAPT_INT] = 50;
    s[S_ADAPT_TOL] = 5.0; s[S_CHECK_GAP] = 1;
}

static void alloc_common(OracleWS *w) {
    int n = w->n, m = w->m, N = w->N;
    w->P0 = dvec(w->nnzP); w->A0 = dvec(w->nnzA); w->q0 = dvec(n); w->l0 = dvec(m); w->u0 = dvec(m);
    w->Px = dvec(w->nnzP); w->Ax = dvec(w->nnzA); w->q = dvec(n); w->l = dvec(m); w->u = dvec(m);
    w->D = dvec(n); w->E = dvec(m); w->Dinv = dvec(n); w->Einv = dvec(m);
    w->rho_vec = dvec(m); w->rho_inv = dvec(m); w->ctype = ivec(m);
    w->perm = ivec(N);
    w->x = dvec(n); w->z = dvec(m); w->y = dvec(m); w->x_prev = dvec(n); w->z_prev = dvec(m);
    w->xz = dvec(N); w->dx = dvec(n); w->dy = dvec(m); w->tAx = dvec(m); w->tPx = dvec(n); w->tAty = dvec(n);
    w->bp = dvec(N); w->tn = dvec(n > m ? n : m); w->tm = dvec(n > m ? n : m);
}

OracleWS *oracle_setup(int n, int m, const int *Pp, const int *Pi, const double *Px, const double *q,
                       const int *Ap, const int *Ai, const double *Ax, const double *l, const double *u,
                       const double *settings) {
    OracleWS *w = (OracleWS *)calloc(1, sizeof(OracleWS));
    w->n = n; w->m = m; w->N = n + m; w->nnzP = Pp[n]; w->nnzA = Ap[n];
    memcpy(w->stg, settings, sizeof(double) * S_COUNT);
    w->Pp = ivec(n + 1); w->Pi = ivec(w->nnzP); w->Ap = ivec(n + 1); w->Ai = ivec(w->nnzA);
    memcpy(w->Pp, Pp, sizeof(int) * (n + 1)); memcpy(w->Pi, Pi, sizeof(int) * w->nnzP);
    memcpy(w->Ap, Ap, sizeof(int) * (n + 1)); memcpy(w->Ai, Ai, sizeof(int) * w->nnzA);
    alloc_common(w);
    memcpy(w->P0, Px, sizeof(double) * w->nnzP); memcpy(w->A0, Ax, sizeof(double) * w->nnzA);
    memcpy(w->q0, q, sizeof(double) * n);
    for (int i = 0; i < m; i++) { w->l0[i] = dmax(l[i], -ORC_INFTY); w->u0[i] = dmin(u[i], ORC_INFTY); w->ctype[i] = -2; }
    if (w->stg[S_SCALING] > 0) scale_data(w);
    else { memcpy(w->Px, w->P0, sizeof(double) * w->nnzP); memcpy(w->Ax, w->A0, sizeof(double) * w->nnzA);
           memcpy(w->q, w->q0, sizeof(double) * n); memcpy(w->l, w->l0, sizeof(double) * m); memcpy(w->u, w->u0, sizeof(double) * m);
           for (int j = 0; j < n; j++) w->D[j] = w->Dinv[j] = 1; for (int i = 0; i < m; i++) w->E[i] = w->Einv[i] = 1; w->c = w->cinv = 1; }
    w->rho = w->stg[S_RHO];
    set_rho_vec(w);
    build_kkt(w);
    int N = w->N;
    w->etree = ivec(N); w->Lnz = ivec(N); w->Lp = ivec(N + 1); w->iw = ivec(3 * N);
    w->bw = (unsigned char *)calloc(N > 0 ? N : 1, 1); w->fw = dvec(N); w->Dg = dvec(N); w->Dginv = dvec(N);
    fill_kkt(w);
    w->nnzL = ldl_etree(N, w->Kp, w->Ki, w->iw, w->Lnz, w->etree);
    if (w->nnzL < 0) { fprintf(stderr, "oracle: etree failed\n"); return NULL; }
    w->Li = ivec(w->nnzL); w->Lx = dvec(w->nnzL);
    if (ldl_factor(w) < 0) { fprintf(stderr, "oracle: zero pivot\n"); return NULL; }
    w->status = ST_UNSOLVED;
    return w;
}

#define DUPI(f, cnt) do { c->f = ivec(cnt); memcpy(c->f, w->f, sizeof(int) * (size_t)(cnt)); } while (0)
#define DUPD(f, cnt) do { c->f = dvec(cnt); memcpy(c->f, w->f, sizeof(double) * (size_t)(cnt)); } while (0)
OracleWS *oracle_clone(const OracleWS *w) {
    OracleWS *c = (OracleWS *)malloc(sizeof(OracleWS)); memcpy(c, w, sizeof(OracleWS));
    int n = w->n, m = w->m, N = w->N, mx = n > m ? n : m;
    DUPI(Pp, n + 1); DUPI(Pi, w->nnzP); DUPI(Ap, n + 1); DUPI(Ai, w->nnzA);
    DUPD(P0, w->nnzP); DUPD(A0, w->nnzA); DUPD(q0, n); DUPD(l0, m); DUPD(u0, m);
    DUPD(Px, w->nnzP); DUPD(Ax, w->nnzA); DUPD(q, n); DUPD(l, m); DUPD(u, m);
    DUPD(D, n); DUPD(E, m); DUPD(Dinv, n); DUPD(Einv, m); DUPD(rho_vec, m); DUPD(rho_inv, m); DUPI(ctype, m);
    DUPI(Kp, N + 1); DUPI(Ki, w->nnzK); DUPD(Kx, w->nnzK);
    DUPI(P2K, w->nnzP); DUPI(A2K, w->nnzA); DUPI(sig2K, n); DUPI(rho2K, m); DUPI(perm, N);
    DUPI(etree, N); DUPI(Lnz, N); DUPI(Lp, N + 1); DUPI(Li, w->nnzL); DUPD(Lx, w->nnzL); DUPD(Dg, N); DUPD(Dginv, N);
    DUPI(iw, 3 * N); c->bw = (unsigned char *)calloc(N > 0 ? N : 1, 1); DUPD(fw, N);
    DUPD(x, n); DUPD(z, m); DUPD(y, m); DUPD(x_prev, n); DUPD(z_prev, m); DUPD(xz, N); DUPD(dx, n); DUPD(dy, m);
    DUPD(tAx, m); DUPD(tPx, n); DUPD(tAty, n); DUPD(bp, N); DUPD(tn, mx); DUPD(tm, mx);
    return c;
}
/* copy the numeric state (data, scaling, factor, settings) of `w` into the existing clone `c` */
static void restore_from(OracleWS *c, const OracleWS *w) {
    int n = w->n, m = w->m, N = w->N;
    memcpy(c->stg, w->stg, sizeof(w->stg));
#define CPD(f, cnt) memcpy(c->f, w->f, sizeof(double) * (size_t)(cnt))
    CPD(P0, w->nnzP); CPD(A0, w->nnzA); CPD(q0, n); CPD(l0, m); CPD(u0, m);
    CPD(Px, w->nnzP); CPD(Ax, w->nnzA); CPD(q, n); CPD(l, m); CPD(u, m);
    CPD(D, n); CPD(E, m); CPD(Dinv, n); CPD(Einv, m); CPD(rho_vec, m); CPD(rho_inv, m);
    memcpy(c->ctype, w->ctype, sizeof(int) * m);
    CPD(Lx, w->nnzL); memcpy(c->Li, w->Li, sizeof(int) * w->nnzL); CPD(Dg, N); CPD(Dginv, N);
    c->c = w->c; c->cinv = w->cinv; c->rho = w->rho;
}
void oracle_free(OracleWS *w) {
    if (!w) return;
    void *ptrs[] = {w->Pp, w->Pi, w->Ap, w->Ai, w->P0, w->A0, w->q0, w->l0, w->u0, w->Px, w->Ax, w->q, w->l, w->u,
        w->D, w->E, w->Dinv, w->Einv, w->rho_vec, w->rho_inv, w->ctype, w->Kp, w->Ki, w->Kx, w->P2K, w->A2K,
        w->sig2K, w->rho2K, w->perm, w->etree, w->Lnz, w->Lp, w->Li, w->Lx, w->Dg, w->Dginv, w->iw, w->bw, w->fw,
        w->x, w->z, w->y, w->x_prev, w->z_prev, w->xz, w->dx, w->dy, w->tAx, w->tPx, w->tAty, w->bp, w->tn, w->tm};
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(ptrs[i]);
    free(w);
}
void oracle_set_settings(OracleWS *w, const double *s) { memcpy(w->stg, s, sizeof(double) * S_COUNT); }
void oracle_dims(const OracleWS *w, int *out) { out[0] = w->n; out[1] = w->m; out[2] = w->nnzL; out[3] = w->nnzK; out[4] = w->n_refactor; }
void oracle_get_scaling(const OracleWS *w, double *D, double *E, double *c) {
    memcpy(D, w->D, sizeof(double) * w->n); memcpy(E, w->E, sizeof(double) * w->m); *c = w->c; }

/* ------------------------------------------------------------------ public: data updates */
/* osqp_update_data_vec: new vectors are scaled with the stored D, E, c; bounds re-classify the
 * rows, and the KKT matrix is refactored only when a row changed class. */
int oracle_update_vec(OracleWS *w, const double *q, const double *l, const double *u) {
    int n = w->n, m = w->m;
    if (q) for (int j = 0; j < n; j++) { w->q0[j] = q[j]; w->q[j] = w->c * w->D[j] * q[j]; }
    if (l) for (int i = 0; i < m; i++) { w->l0[i] = l[i]; w->l[i] = w->E[i] * l[i]; }
    if (u) for (int i = 0; i < m; i++) { w->u0[i] = u[i]; w->u[i] = w->E[i] * u[i]; }
    if (l || u) { for (int i = 0; i < m; i++) if (w->l0[i] > w->u0[i]) return 1;
        if (set_rho_vec(w)) return refactor(w); }
    return 0;
}
/* osqp_update_data_mat: unscale, overwrite the values, equilibrate again from scratch, refactor */
int oracle_update_mat(OracleWS *w, const double *Px, const double *Ax) {
    if (Px) memcpy(w->P0, Px, sizeof(double) * w->nnzP);
    if (Ax) memcpy(w->A0, Ax, sizeof(double) * w->nnzA);
    if (w->stg[S_SCALING] > 0) scale_data(w);
    else { memcpy(w->Px, w->P0, sizeof(double) * w->nnzP); memcpy(w->Ax, w->A0, sizeof(double) * w->nnzA); }
    return refactor(w);
}
/* osqp_update_rho: settings->rho and the workspace (rho_vec, factor) */
static int update_rho(OracleWS *w, double rho_new) {
    w->rho = w->stg[S_RHO] = dmin(dmax(rho_new, ORC_RHO_MIN), ORC_RHO_MAX);
    for (int i = 0; i < w->m; i++) {
        w->rho_vec[i] = w->ctype[i] == -1 ? ORC_RHO_MIN : (w->ctype[i] == 1 ? ORC_RHO_EQ_OVER_INEQ * w->rho : w->rho);
        w->rho_inv[i] = 1.0 / w->rho_vec[i]; }
    return refactor(w);
}

/* ------------------------------------------------------------------ residuals / termination */
static void update_info(OracleWS *w, int iter) {
    int n = w->n, m = w->m; int unsc = w->stg[S_SCALING] > 0 && !(int)w->stg[S_SCALED_TERM];
    w->iter = iter;
    spmv_csc(n, w->Ap, w->Ai, w->Ax, w->x, w->tAx, m);
    for (int i = 0; i < m; i++) w->tm[i] = w->tAx[i] - w->z[i];
    w->sc_prim_res = norm_inf(w->tm, m);
    w->prim_res = unsc ? snorm_inf(w->Einv, w->tm, m) : w->sc_prim_res;
    symv_triu(n, w->Pp, w->Pi, w->Px, w->x, w->tPx);
    spmtv_csc(n, w->Ap, w->Ai, w->Ax, w->y, w->tAty);
    for (int j = 0; j < n; j++) w->tn[j] = w->q[j] + w->tPx[j] + w->tAty[j];
    w->sc_dual_res = norm_inf(w->tn, n);
    w->dual_res = unsc ? w->cinv * snorm_inf(w->Dinv, w->tn, n) : w->sc_dual_res;
    double quad = 0, lin = 0;
    for (int j = 0; j < n; j++) { quad += w->x[j] * w->tPx[j]; lin += w->q[j] * w->x[j]; }
    w->obj_val = (0.5 * quad + lin) * (w->stg[S_SCALING] > 0 ? w->cinv : 1.0);
    if ((int)w->stg[S_CHECK_GAP]) { double sup = 0;
        for (int i = 0; i < m; i++) {
            if (w->u[i] < ORC_INFTY * ORC_MIN_SCALING && w->y[i] > 0) sup += w->u[i] * w->y[i];
            if (w->l[i] > -ORC_INFTY * ORC_MIN_SCALING && w->y[i] < 0) sup += w->l[i] * w->y[i]; }
        double cs = w->stg[S_SCALING] > 0 ? w->cinv : 1.0;
        w->dual_obj = (-0.5 * quad - sup) * cs; w->gap = fabs(quad + lin + sup) * cs; }
}
static int is_primal_infeasible(OracleWS *w, double eps) {
    int n = w->n, m = w->m; int unsc = w->stg[S_SCALING] > 0 && !(int)w->stg[S_SCALED_TERM];
    for (int i = 0; i < m; i++) { /* project delta_y on the polar of the recession cone of [l,u] */
        int iu = w->u[i] > ORC_INFTY * ORC_MIN_SCALING, il = w->l[i] < -ORC_INFTY * ORC_MIN_SCALING;
        if (iu && il) w->dy[i] = 0; else if (iu) w->dy[i] = dmin(w->dy[i], 0); else if (il) w->dy[i] = dmax(w->dy[i], 0); }
    double nrm = unsc ? snorm_inf(w->E, w->dy, m) : norm_inf(w->dy, m);
    if (nrm > ORC_DIV_TOL) { double lhs = 0;
        for (int i = 0; i < m; i++) lhs += w->u[i] * dmax(w->dy[i], 0) + w->l[i] * dmin(w->dy[i], 0);
        if (lhs < eps * nrm) { spmtv_csc(n, w->Ap, w->Ai, w->Ax, w->dy, w->tn);
            double r = unsc ? snorm_inf(w->Dinv, w->tn, n) : norm_inf(w->tn, n); return r < eps * nrm; } }
    return 0;
}
static int is_dual_infeasible(OracleWS *w, double eps) {
    int n = w->n, m = w->m; int unsc = w->stg[S_SCALING] > 0 && !(int)w->stg[S_SCALED_TERM];
    double nrm = unsc ? snorm_inf(w->D, w->dx, n) : norm_inf(w->dx, n), cs = unsc ? w->c : 1.0;
    if (nrm > ORC_DIV_TOL) { double qdx = 0; for (int j = 0; j < n; j++) qdx += w->q[j] * w->dx[j];
        if (qdx < -cs * eps * nrm) { symv_triu(n, w->Pp, w->Pi, w->Px, w->dx, w->tn);
            double r = unsc ? snorm_inf(w->Dinv, w->tn, n) : norm_inf(w->tn, n);
            if (r < cs * eps * nrm) { spmv_csc(n, w->Ap, w->Ai, w->Ax, w->dx, w->tm, m);
                for (int i = 0; i < m; i++) { double a = unsc ? w->Einv[i] * w->tm[i] : w->tm[i];
                    if ((w->u[i] < ORC_INFTY * ORC_MIN_SCALING && a > eps * nrm) ||
                        (w->l[i] > -ORC_INFTY * ORC_MIN_SCALING && a < -eps * nrm)) return 0; }
                return 1; } } }
    return 0;
}
static int check_termination(OracleWS *w, int approximate) {
    int n = w->n, m = w->m; int unsc = w->stg[S_SCALING] > 0 && !(int)w->stg[S_SCALED_TERM];
    double mult = approximate ? 10.0 : 1.0;
    double ea = w->stg[S_EPS_ABS] * mult, er = w->stg[S_EPS_REL] * mult;
    double epi = w->stg[S_EPS_PINF] * mult, edi = w->stg[S_EPS_DINF] * mult;
    int pc = 0, dc = 0, pic = 0, dic = 0, gc = 1;
    if (w->prim_res > ORC_INFTY || w->dual_res > ORC_INFTY) { w->status = ST_NON_CVX; w->obj_val = NAN; return 1; }
    if (m == 0) pc = 1; else {
        double nz = unsc ? snorm_inf(w->Einv, w->z, m) : norm_inf(w->z, m);
        double na = unsc ? snorm_inf(w->Einv, w->tAx, m) : norm_inf(w->tAx, m);
        if (w->prim_res < ea + er * dmax(nz, na)) pc = 1; else pic = is_primal_infeasible(w, epi); }
    { double nq = unsc ? snorm_inf(w->Dinv, w->q, n) : norm_inf(w->q, n);
      double na = unsc ? snorm_inf(w->Dinv, w->tAty, n) : norm_inf(w->tAty, n);
      double np = unsc ? snorm_inf(w->Dinv, w->tPx, n) : norm_inf(w->tPx, n);
      double mx = dmax(nq, dmax(na, np)) * (unsc ? w->cinv : 1.0);
      if (w->dual_res < ea + er * mx) dc = 1; else dic = is_dual_infeasible(w, edi); }
    if ((int)w->stg[S_CHECK_GAP]) gc = w->gap < ea + er * dmax(fabs(w->obj_val), fabs(w->dual_obj));
    if (pc && dc && gc) { w->status = approximate ? ST_SOLVED_INACC : ST_SOLVED; return 1; }
    if (pic) { w->status = approximate ? ST_PINF_INACC : ST_PINF; w->obj_val = ORC_INFTY; return 1; }
    if (dic) { w->status = approximate ? ST_DINF_INACC : ST_DINF; w->obj_val = -ORC_INFTY; return 1; }
    return 0;
}
static double rho_estimate(const OracleWS *w) {
    double pn = dmax(norm_inf(w->z, w->m), norm_inf(w->tAx, w->m));
    double dn = dmax(norm_inf(w->q, w->n), dmax(norm_inf(w->tAty, w->n), norm_inf(w->tPx, w->n)));
    double pr = w->sc_prim_res / (pn + ORC_DIV_TOL), dr = w->sc_dual_res / (dn + ORC_DIV_TOL);
    /* compute_rho_estimate scales settings->rho, not the rho of the workspace */
    return dmin(dmax(w->stg[S_RHO] * sqrt(pr / dr), ORC_RHO_MIN), ORC_RHO_MAX);
}

/* ------------------------------------------------------------------ public: solve */
void oracle_warm_start(OracleWS *w, const double *x, const double *y) {
    int n = w->n, m = w->m;
    for (int j = 0; j < n; j++) w->x[j] = x[j] * w->Dinv[j];
    for (int i = 0; i < m; i++) w->y[i] = y[i] * w->Einv[i] * w->c;
    spmv_csc(n, w->Ap, w->Ai, w->Ax, w->x, w->z, m);
}
/* OSQP paper Algorithm 1 with relaxation alpha, per-row rho, termination every
 * check_termination iterations.  Outputs unscaled sol_x[n], sol_y[m]. */
int oracle_solve(OracleWS *w, double *sol_x, double *sol_y) {
    int n = w->n, m = w->m, N = w->N;
    double sigma = w->stg[S_SIGMA], alpha = w->stg[S_ALPHA];
    int max_iter = (int)w->stg[S_MAX_ITER], chk = (int)w->stg[S_CHECK_TERM];
    int ad = (int)w->stg[S_ADAPT_RHO], adi = (int)w->stg[S_ADAPT_INT];
    if (!(int)w->stg[S_WARM]) { memset(w->x, 0, sizeof(double) * n); memset(w->z, 0, sizeof(double) * m); memset(w->y, 0, sizeof(double) * m); }
    w->status = ST_UNSOLVED;
    int iter, can_check = 0, done = 0;
    for (iter = 1; iter <= max_iter; iter++) {
        { double *t = w->x; w->x = w->x_prev; w->x_prev = t; t = w->z; w->z = w->z_prev; w->z_prev = t; }
        /* KKT right-hand side and solve */
        for (int j = 0; j < n; j++) w->xz[j] = sigma * w->x_prev[j] - w->q[j];
        for (int i = 0; i < m; i++) w->xz[n + i] = w->z_prev[i] - w->rho_inv[i] * w->y[i];
        for (int k = 0; k < N; k++) w->bp[k] = w->xz[w->perm[k]];
        ldl_solve(w, w->bp);
        for (int k = 0; k < N; k++) { int o = w->perm[k];
            if (o < n) w->xz[o] = w->bp[k]; else w->xz[o] += w->rho_inv[o - n] * w->bp[k]; }
        /* relaxed updates, projection on [l,u], dual ascent */
        for (int j = 0; j < n; j++) { w->x[j] = alpha * w->xz[j] + (1.0 - alpha) * w->x_prev[j]; w->dx[j] = w->x[j] - w->x_prev[j]; }
        for (int i = 0; i < m; i++) { double zr = alpha * w->xz[n + i] + (1.0 - alpha) * w->z_prev[i];
            double zi = zr + w->rho_inv[i] * w->y[i]; zi = dmin(dmax(zi, w->l[i]), w->u[i]);
            w->z[i] = zi; w->dy[i] = w->rho_vec[i] * (zr - zi); w->y[i] += w->dy[i]; }
        can_check = chk && (iter % chk == 0);
        if (can_check) { update_info(w, iter); if (check_termination(w, 0)) { done = 1; break; } }
        if (ad && adi && (iter % adi == 0)) {
            if (!can_check) update_info(w, iter);
            double rn = rho_estimate(w), tol = w->stg[S_ADAPT_TOL];
            if (rn > w->stg[S_RHO] * tol || rn < w->stg[S_RHO] / tol) if (update_rho(w, rn)) return -1;
        }
    }
    if (!done) { iter = max_iter; if (!can_check) update_info(w, iter);
        if (!check_termination(w, 0) && !check_termination(w, 1)) w->status = ST_MAX_ITER; }
    w->iter = iter;
    int has_sol = w->status == ST_SOLVED || w->status == ST_SOLVED_INACC || w->status == ST_MAX_ITER;
    for (int j = 0; j < n; j++) sol_x[j] = has_sol ? w->D[j] * w->x[j] : NAN;
    for (int i = 0; i < m; i++) sol_y[i] = has_sol ? w->cinv * w->E[i] * w->y[i] : NAN;
    if (!has_sol) { memset(w->x, 0, sizeof(double) * n); memset(w->z, 0, sizeof(double) * m); memset(w->y, 0, sizeof(double) * m); }
    return 0;
}
void oracle_info(const OracleWS *w, double *out) {
    out[0] = w->obj_val; out[1] = w->iter; out[2] = w->status; out[3] = w->prim_res; out[4] = w->dual_res; out[5] = w->rho; }
/* K^{-1} b through the factor, for tests of the linear algebra only */
void oracle_kkt_solve(OracleWS *w, const double *b, double *out) {
    for (int k = 0; k < w->N; k++) w->bp[k] = b[w->perm[k]];
    ldl_solve(w, w->bp);
    for (int k = 0; k < w->N; k++) out[w->perm[k]] = w->bp[k];
}

/* ------------------------------------------------------------------ cpg_solve() for a batch */
/* One canonical-parameter map (CSR over [theta; 1]), cvxpygen/utils.py:279-294 */
typedef struct { int rows; const int *p, *i; const double *x; } CsrMap;
static void canonicalize(const CsrMap *mp, const double *theta, double *out) {
    for (int r = 0; r < mp->rows; r++) { double s = 0;
        for (int k = mp->p[r]; k < mp->p[r + 1]; k++) s += mp->x[k] * theta[mp->i[k]]; out[r] = s; }
}
/*
 * Runs, independently for every instance b of the batch, what one process of the reference does
 * on its first cpg_solve() after code generation with the user parameters of instance b:
 * canonicalise the outdated canonical parameters, push them with osqp_update_data_mat / _vec, cold
 * start, osqp_solve, cpg_retrieve_info (obj_val + d, sign flipped for maximisation).
 *   maps: 6 CSR maps in the order P, q, d, A, l, u (rows may be 0);  outdated[6]: flags
 *   theta: B x (NP+1) row-major, trailing 1 included
 *   out:  sol_x B x n, sol_y B x m, info B x 5 = (obj_val, iter, status, pri_res, dua_res)
 */
int oracle_cpg_solve_batch(const OracleWS *tmpl, int n_eq,
                           const int *map_rows, const int *const *map_p, const int *const *map_i,
                           const double *const *map_x, const int *outdated, int is_max, int NP1,
                           long B, const double *theta, double *sol_x, double *sol_y, double *info,
                           int nthreads) {
    int n = tmpl->n, m = tmpl->m; int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        OracleWS *w = oracle_clone(tmpl);
        double *Pn = dvec(tmpl->nnzP), *An = dvec(tmpl->nnzA), *qn = dvec(n), *ln = dvec(m), *un = dvec(m);
#pragma omp for schedule(static)
        for (long b = 0; b < B; b++) {
            const double *th = theta + (size_t)b * NP1; double d = 0;
            restore_from(w, tmpl);
            CsrMap mp[6]; for (int k = 0; k < 6; k++) mp[k] = (CsrMap){map_rows[k], map_p[k], map_i[k], map_x[k]};
            if (outdated[0]) canonicalize(&mp[0], th, Pn);
            if (outdated[1]) canonicalize(&mp[1], th, qn);
            if (mp[2].rows) canonicalize(&mp[2], th, &d);
            if (outdated[3]) canonicalize(&mp[3], th, An);
            if (outdated[4]) { canonicalize(&mp[4], th, ln); for (int i = n_eq; i < m; i++) ln[i] = -ORC_INFTY; }
            if (outdated[5]) canonicalize(&mp[5], th, un);
            int rc = 0;
            if (outdated[0] || outdated[3]) rc |= oracle_update_mat(w, outdated[0] ? Pn : NULL, outdated[3] ? An : NULL);
            if (outdated[1] || outdated[4] || outdated[5])
                rc |= oracle_update_vec(w, outdated[1] ? qn : NULL, outdated[4] ? ln : NULL, outdated[5] ? un : NULL);
            w->stg[S_WARM] = 0;
            rc |= oracle_solve(w, sol_x + (size_t)b * n, sol_y + (size_t)b * m);
            if (rc) {
#pragma omp atomic write
                fail = 1;
            }
            double ov = w->obj_val + d; if (is_max) ov = -ov;
            double *io = info + (size_t)b * 5;
            io[0] = ov; io[1] = w->iter; io[2] = w->status; io[3] = w->prim_res; io[4] = w->dual_res;
        }
        free(Pn); free(An); free(qn); free(ln); free(un); oracle_free(w);
    }
    return fail;
}
int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ QP adjoint (gradient) */
/*
 * Restatement of cpg_osqp_gradient() (cvxpygen/templates/cpg_osqp_grad_compute.c.jinja2:432-531)
 * for one instance; see oracle/osqp_grad_numpy.py for the line-by-line map.  The masked,
 * eps-regularised KKT matrix is factored densely (LDL' without pivoting, natural order as in the
 * reference, template :326-347) -- the rank-one add / delete machinery of the reference
 * (template :157-324) produces exactly this factor.
 *   P upper CSC (Pp, Pi, Px), A CSC (Ap, Ai, Ax), x[n], y[m], dx[n]
 *   out: r[n+m], dq[n], dl[m], du[m], dP[nnzP], dA[nnzA]
 */
int oracle_qp_adjoint(int n, int m, const int *Pp, const int *Pi, const double *Px, const int *Ap,
                      const int *Ai, const double *Ax, const double *x, const double *y, const double *dx,
                      double *r, double *dq, double *dl, double *du, double *dP, double *dA) {
    int N = n + m;
    double *K = dvec(N * N), *L = dvec(N * N), *D = dvec(N), *rhs = dvec(N), *delta = dvec(N), *t = dvec(N);
    int *a = ivec(m);
    for (int i = 0; i < m; i++) a[i] = y[i] < -1e-12 ? -1 : (y[i] > 1e-12 ? 1 : 0);
    for (int j = 0; j < n; j++) for (int k = Pp[j]; k < Pp[j + 1]; k++) { int i = Pi[k];
        K[i * N + j] = Px[k]; K[j * N + i] = Px[k]; }
    for (int j = 0; j < n; j++) K[j * N + j] += 1e-6;
    for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) { int i = Ai[k];
        if (a[i]) { K[(n + i) * N + j] = Ax[k]; K[j * N + n + i] = Ax[k]; } }
    for (int i = 0; i < m; i++) K[(n + i) * N + n + i] = a[i] ? -1e-6 : -1.0;
    /* dense LDL', no pivoting */
    for (int j = 0; j < N; j++) { double d = K[j * N + j];
        for (int k = 0; k < j; k++) d -= L[j * N + k] * L[j * N + k] * D[k];
        if (d == 0.0) { free(K); free(L); free(D); free(rhs); free(delta); free(t); free(a); return -1; }
        D[j] = d; L[j * N + j] = 1.0;
        for (int i = j + 1; i < N; i++) { double v = K[i * N + j];
            for (int k = 0; k < j; k++) v -= L[i * N + k] * L[j * N + k] * D[k];
            L[i * N + j] = v / d; } }
#define ADJ_SOLVE(v)                                                                     \
    do { for (int i_ = 0; i_ < N; i_++) { double s_ = v[i_]; for (int k_ = 0; k_ < i_; k_++) s_ -= L[i_ * N + k_] * v[k_]; v[i_] = s_; } \
         for (int i_ = 0; i_ < N; i_++) v[i_] /= D[i_];                                  \
         for (int i_ = N - 1; i_ >= 0; i_--) { double s_ = v[i_]; for (int k_ = i_ + 1; k_ < N; k_++) s_ -= L[k_ * N + i_] * v[k_]; v[i_] = s_; } } while (0)
    for (int i = 0; i < n; i++) { r[i] = dx[i]; rhs[i] = dx[i]; }
    for (int i = n; i < N; i++) { r[i] = 0; rhs[i] = 0; }
    ADJ_SOLVE(r);
    for (int l = 0; l < 3; l++) {
        /* delta = rhs - K_true r with inactive rows / columns skipped */
        for (int i = 0; i < N; i++) delta[i] = rhs[i];
        symv_triu(n, Pp, Pi, Px, r, t);
        for (int i = 0; i < n; i++) delta[i] -= t[i];
        for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) { int i = Ai[k];
            if (a[i]) { delta[j] -= Ax[k] * r[n + i]; delta[n + i] -= Ax[k] * r[j]; } }
        for (int i = 0; i < m; i++) if (!a[i]) delta[n + i] = 0.0;
        ADJ_SOLVE(delta);
        for (int i = 0; i < N; i++) r[i] += delta[i];
    }
    for (int i = 0; i < n; i++) dq[i] = -r[i];
    for (int i = 0; i < m; i++) { dl[i] = a[i] == -1 ? r[n + i] : 0.0; du[i] = a[i] == 1 ? r[n + i] : 0.0; }
    for (int j = 0; j < n; j++) for (int k = Pp[j]; k < Pp[j + 1]; k++) { int i = Pi[k];
        dP[k] = -0.5 * (r[i] * x[j] + x[i] * r[j]); }
    for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) { int i = Ai[k];
        dA[k] = a[i] ? -(r[n + i] * x[j] + y[i] * r[j]) : 0.0; }
    free(K); free(L); free(D); free(rhs); free(delta); free(t); free(a);
    return 0;
}
