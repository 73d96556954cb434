// Conic interior-point kernel: what the reference does per solve on
// its Clarabel path -- cpg_copy_* of the canonical parameters, clarabel_DefaultSolver_new
// (equilibration, KKT assembly, factorisation), clarabel_DefaultSolver_solve, _solution
// (cvxpygen/solvers/clarabel.py:172-204) -- for every instance of a batch.
//
// Mapping: ONE wavefront per instance, instances handed out through a global atomic counter.
// The whole interior-point state of an instance (scaled P / A values, iterates, step, NT scaling,
// the numeric factor L, D and the coefficients of the substitution program) lives in the
// wavefront's slice of LDS; HBM is touched for theta on the way in, the solution on the way out
// and the read-only family tables (patterns, schedules), which stay in L2.  Vectors are strided
// over the 64 lanes; second-order cones are handled one cone per lane.
//
// Algorithm (restated from Goulart & Chen 2024, "Clarabel"; defaults cvxpygen/solvers/clarabel.py:
// 63-119): homogeneous embedding (tau, kappa); Ruiz equilibration; Nesterov-Todd scaling;
// K = [[P + eps I, A'], [A, -W'W - eps I]] factored as LDL' (fixed pattern, level-scheduled
// dot-product form) with dynamic pivot regularisation and iterative refinement against the
// unregularised K; Mehrotra predictor-corrector with sigma = (1 - alpha)^3.
//
// Cones: zero, nonnegative, second-order in every instantiation; PSD (cpg_clarabel_psd.h), exponential and power cones
// (cpg_clarabel_nonsym.h) in the extended one (template switch NS / NONSYM: the symmetric kernel carries none of their code) --
// every type of the reference's `cones` array (clarabel.py:308-323), rows zero | nonneg | soc | psd | exp | pow.
#pragma once
#include "cpg_osqp_kernel.h"
#ifdef CPG_GENC_HEADER
// generated straight-line executor of this family's substitution program (codegen.emit_instance_program, coef='lds'):
// cpg::run_program_conic(entries, offsets, slots, w, lane)
#include CPG_GENC_HEADER
#endif
#include "cpg_clarabel_nonsym.h"
#include "cpg_clarabel_psd.h"

// (timing experiments, scripts/gpu_probe_conic.py: bit k set = piece k of an iteration is executed TWICE -- every piece is
// idempotent, so results and control flow stay what they are and the time added is the piece's cost.  1 factorisation, 2 substitution
// sweeps, 4 refinement residual, 8 NT scaling, 16 step lengths, 32 combined-step offset, 64 step assembly)
#ifndef CPG_CONIC_TWICE
#define CPG_CONIC_TWICE 0
#endif
#define CPG_CONIC_REPEAT(bit) for (int rep_ = 0; rep_ < ((CPG_CONIC_TWICE & (bit)) ? 2 : 1); rep_++)

namespace cpg {

#define CPG_CK_P 1
#define CPG_CK_A 2
#define CPG_CK_DIAGX 3
#define CPG_CK_HDIAG 5
#define CPG_CK_HSOC 6
#define CPG_CK_HNS 7            // off-diagonal entry of an exponential / power cone's 3 x 3 scaling block: -wv[idx]
#define CPG_CK_HPSD 8           // off-diagonal entry of a PSD cone's block: idx = offset of Q in the slice's PSD store | p << 12 | i << 16 | j << 19 | k << 22 | l << 25

// Clarabel's SolverStatus numbering (the reference hands the integer through: clarabel.py:37-46)
#define CPG_CL_UNSOLVED 0
#define CPG_CL_SOLVED 1
#define CPG_CL_PRIMAL_INFEASIBLE 2
#define CPG_CL_DUAL_INFEASIBLE 3
#define CPG_CL_ALMOST_SOLVED 4
#define CPG_CL_ALMOST_PRIMAL_INFEASIBLE 5
#define CPG_CL_ALMOST_DUAL_INFEASIBLE 6
#define CPG_CL_MAX_ITERATIONS 7
#define CPG_CL_NUMERICAL_ERROR 9
#define CPG_CL_INSUFFICIENT_PROGRESS 10

struct DevConicSettings {
    int max_iter, equilibrate_enable, equilibrate_max_iter, static_reg_enable, dynamic_reg_enable, ir_enable,
        ir_max_iter;
    double max_step_fraction, tol_gap_abs, tol_gap_rel, tol_feas, tol_infeas_abs, tol_infeas_rel, eq_min, eq_max,
        static_const, static_prop, dyn_eps, dyn_delta, ir_reltol, ir_abstol, ir_stop_ratio, min_terminate_step;
    double ls_backtrack, min_switch_step;      // nonsymmetric cones: backtracking factor, step length below which the scaling strategy switches
    // kappa/tau threshold of the infeasibility certificates and the reduced tolerances behind the "almost" statuses
    // (cvxpygen/solvers/clarabel.py:76-84)
    double tol_ktratio, red_gap_abs, red_gap_rel, red_feas, red_infeas_abs, red_infeas_rel, red_ktratio;
};

struct DevConic {
    int n, m, nnzP, nnzA, nnzL, n_zero, n_nonneg, n_soc, is_max, p_is_zero;
    int n_ns;                                // exponential + power cones: the last 3 n_ns rows, three per cone
    const double *ns_alpha;                  // [n_ns] exponent of a power cone, 0 for an exponential cone
    // PSD cones (rows between the second-order and the exponential cones): first row, matrix order and offset of the cone's
    // [Q | R | R^-1 | lambda] in the slice's PSD store, per cone (global memory); psd_first: first PSD row; psd_degree: sum of orders
    int n_psd, psd_first, psd_doubles, psd_degree;
    const int *psd_start, *psd_dim, *psd_off;
    const int *soc_start, *soc_dim;          // [n_soc]
    const int *row_cone;                     // [m] first row of the row's second-order cone, -1 otherwise
    const int *Ap, *Ai, *Arp, *Aent, *Acol, *Pp, *Pi, *Prp, *Pent, *Pcol;
    const int *Lcol, *ksrc_kind, *ksrc_idx;
    const int *fac_ctab;
    const unsigned *fac_task, *fac_len, *fac_a, *fac_b, *fac_k;
    int fac_chunks;
    const int *sol_ctab;
    const unsigned *sol_desc;
    const unsigned short *sol_cols;
    const int *sol_kind, *sol_idx;
    const unsigned short *sol_fpos;
    int sol_chunks, sol_nnz, sol_slots;
    int np_var;
    const double *P_base, *A_base, *q_base, *b_base;
    double d_base;
    DevCsr map_P, map_A, map_q, map_b, map_d;
    int n_prim, n_dual;
    const int *prim_idx, *dual_idx;
    int lds_doubles;                         // per wavefront
    int fac_triples, n_pfull;                // table lengths needed to stage the tables in LDS
    int tab_doubles;                         // LDS doubles of the block-shared copy of all index tables
    // generated executor of the substitution program (family library, CPG_GENC_HEADER): operand offsets
    // [step / 4][lane][4] and output slots [chunk / 4][lane][4]; the entries get `sv_pad` trailing zeros and the work
    // vector `w_extra` slots (dummy store targets + the zero slot)
    int gc_ok, sv_pad, w_extra, gc_ncols, gc_nrows;
    int gf_ok;                               // the numeric LDL' schedule handed in is the one this library's generated factorisation was emitted from
    const unsigned short *gc_cols, *gc_rows;
};

struct ConicBuf {
    double *P, *A, *q, *b, *D, *E, *x, *z, *s, *dx, *dz, *ds, *x2, *z2, *rx, *rz, *tx, *tz, *lam, *wv, *hd, *et,
        *dsc, *rb, *sol, *er, *cand, *Lx, *Dg, *Dginv, *sv, *w, *px, *pz, *ps, *psd;
};
// per-wavefront LDS: nnzP + nnzA + 7n + 14m + 6(n+m) + nnzL + sol_nnz + sv_pad + sol_slots + w_extra + psd_doubles doubles (host: cpg_hip.cpp)
CPG_DEV ConicBuf conic_carve(double *p, const DevConic &C) {
    ConicBuf o;
    const int n = C.n, m = C.m, N = n + m;
    o.P = p; p += C.nnzP; o.A = p; p += C.nnzA;
    o.q = p; p += n; o.D = p; p += n; o.x = p; p += n; o.dx = p; p += n; o.x2 = p; p += n; o.rx = p; p += n; o.tx = p; p += n;
    o.b = p; p += m; o.E = p; p += m; o.z = p; p += m; o.s = p; p += m; o.dz = p; p += m; o.ds = p; p += m;
    o.z2 = p; p += m; o.rz = p; p += m; o.tz = p; p += m; o.lam = p; p += m; o.wv = p; p += m; o.hd = p; p += m;
    o.et = p; p += m; o.dsc = p; p += m;
    o.rb = p; p += N; o.sol = p; p += N; o.er = p; p += N; o.cand = p; p += N; o.Dg = p; p += N; o.Dginv = p; p += N;
    o.Lx = p; p += C.nnzL; o.sv = p; p += C.sol_nnz + C.sv_pad; o.w = p;
    o.psd = p + C.sol_slots + C.w_extra;         // (PSD cones: Q | R | R^-1 | lambda per cone, psd_doubles in all)
    // previous iterate (insufficient progress falls back to it): in the step's place -- a step is dead from the moment it
    // is applied until the next iteration's solves write a new one, which is after the progress test
    o.px = o.dx; o.pz = o.dz; o.ps = o.ds;
    return o;
}

#ifndef CPG_GENC_ROWS
typedef unsigned genc_row_word;      // (no row words in this library: ConicCtx::rows stays null)
#endif
}  // namespace cpg
#ifdef CPG_GENC_FACTOR_HEADER
// generated straight-line numeric LDL' of this family (codegen.emit_conic_factor): cpg::conic_factor_gen(B, S, eps, lane)
#include CPG_GENC_FACTOR_HEADER
#endif
namespace cpg {

// NS: the family has exponential / power cones (a compile-time switch: the symmetric kernel carries none of their code).
// Their scaling state sits in the per-row vectors of the slice: hd = diagonal of the cone's 3 x 3 block H_s,
// wv = (H_s01, H_s02, H_s12), et = gradient of the dual barrier at z.
template <bool NS>
struct ConicCtxT {
    const DevConic &C;
    const DevConicSettings &S;
    ConicBuf B;
    int lane;
    unsigned n, m, N;
    // The family's own library, specialised kernel: the rows of P, the columns and the rows of A as padded per-lane word lists
    // (codegen.conic_row_tables; block-shared LDS copy, [step][lane]) -- null: the table-driven loops over the CSR / CSC arrays.
    // lane = row (n, m <= 64), a row's entries in the order of its loop, so both forms give the same bits.
    const genc_row_word *rows;

#ifdef CPG_GENC_ROWS
    static constexpr int ROWS_P = 0, ROWS_AT = CPG_GENC_ROWS_SP * 64, ROWS_A = (CPG_GENC_ROWS_SP + CPG_GENC_ROWS_SAT) * 64;
    // sum over the lane's list of (entry of P | A) x (operand of v): all words, then all values and operands are requested
    // before the first multiply-add -- two LDS round trips for the row instead of two per entry
    template <int S_>
    CPG_DEV double row_dot(int first, unsigned row, const double *v) const {
        constexpr int SS = S_ > 0 ? S_ : 1;
        unsigned w[SS];
        double a[SS], x[SS];
#pragma unroll
        for (int s = 0; s < S_; s++) w[s] = rows[first + s * 64 + (int)row];
#pragma unroll
        for (int s = 0; s < S_; s++) { a[s] = B.P[CPG_GENC_ROW_ENT(w[s])]; x[s] = v[CPG_GENC_ROW_OP(w[s])]; }
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < S_; s++) { const double t = fma(a[s], x[s], acc); acc = CPG_GENC_ROW_VALID(w[s]) ? t : acc; }
        return acc;
    }
    // largest |entry| of the lane's list
    template <int S_>
    CPG_DEV double row_absmax(int first, unsigned row, double acc) const {
        constexpr int SS = S_ > 0 ? S_ : 1;
        unsigned w[SS];
#pragma unroll
        for (int s = 0; s < S_; s++) w[s] = rows[first + s * 64 + (int)row];
#pragma unroll
        for (int s = 0; s < S_; s++) { const double t = cpgw::dmax2(acc, fabs(B.P[CPG_GENC_ROW_ENT(w[s])])); acc = CPG_GENC_ROW_VALID(w[s]) ? t : acc; }
        return acc;
    }
#endif

    // ---- sparse products with the instance's scaled matrices ------------------------------------
    CPG_DEV double row_P(unsigned j, const double *v) const {
#ifdef CPG_GENC_ROWS
        if (rows) return row_dot<CPG_GENC_ROWS_SP>(ROWS_P, j, v);
#endif
        double acc = 0.0;
        const unsigned a = (unsigned)cpgw::gld(C.Prp, j), e = (unsigned)cpgw::gld(C.Prp, j + 1u);
        for (unsigned k = a; k < e; k++) acc = fma(B.P[(unsigned)cpgw::gld(C.Pent, k)], v[(unsigned)cpgw::gld(C.Pcol, k)], acc);
        return acc;
    }
    CPG_DEV double row_A(unsigned i, const double *v) const {
#ifdef CPG_GENC_ROWS
        if (rows) return row_dot<CPG_GENC_ROWS_SA>(ROWS_A, i, v);
#endif
        double acc = 0.0;
        const unsigned a = (unsigned)cpgw::gld(C.Arp, i), e = (unsigned)cpgw::gld(C.Arp, i + 1u);
        for (unsigned k = a; k < e; k++) acc = fma(B.A[(unsigned)cpgw::gld(C.Aent, k)], v[(unsigned)cpgw::gld(C.Acol, k)], acc);
        return acc;
    }
    CPG_DEV double col_At(unsigned j, const double *v) const {
#ifdef CPG_GENC_ROWS
        if (rows) return row_dot<CPG_GENC_ROWS_SAT>(ROWS_AT, j, v);
#endif
        double acc = 0.0;
        const unsigned a = (unsigned)cpgw::gld(C.Ap, j), e = (unsigned)cpgw::gld(C.Ap, j + 1u);
        for (unsigned k = a; k < e; k++) acc = fma(B.A[k], v[(unsigned)cpgw::gld(C.Ai, k)], acc);
        return acc;
    }
    // a PSD cone's part of the slice: the NT point Q = R R', R, R^-1, lambda, then its workspace (eight p x p matrices, two p-vectors)
    struct PsdRef {
        unsigned st; int p, d; double *Q, *R, *Ri, *lam, *W;
        CPG_DEV double *mat(int i) const { return W + i * p * p; }
        CPG_DEV double *vec(int i) const { return W + 8 * p * p + i * p; }
    };
    CPG_DEV PsdRef psd_ref(int k) const {
        PsdRef r;
        r.st = (unsigned)cpgw::gld(C.psd_start, (unsigned)k); r.p = cpgw::gld(C.psd_dim, (unsigned)k);
        r.d = r.p * (r.p + 1) / 2;
        r.Q = B.psd + cpgw::gld(C.psd_off, (unsigned)k); r.R = r.Q + r.p * r.p; r.Ri = r.R + r.p * r.p; r.lam = r.Ri + r.p * r.p; r.W = r.lam + r.p;
        return r;
    }
    // per-cone dot products  sum_r wv[r] v[r]  are written to tmp[first row of the cone]; PSD cones: ALL rows of W'W v = svec(Q V Q)
    CPG_DEV void soc_dots(const double *v, double *tmp) const {
        if (NS) {
            for (int k = lane; k < C.n_psd; k += 64) {
                const PsdRef r = psd_ref(k);
                double *X = r.mat(0), *T = r.mat(1), *Y = r.mat(2);
                psd::svec_to_mat(v + r.st, r.p, X);
                psd::congruence(r.p, r.Q, false, X, T, Y);
                psd::mat_to_svec(Y, r.p, tmp + r.st);
            }
        }
        for (int k = lane; k < C.n_soc; k += 64) {
            const unsigned st = (unsigned)cpgw::gld(C.soc_start, (unsigned)k), dm = (unsigned)cpgw::gld(C.soc_dim, (unsigned)k);
            double acc = 0.0;
#ifdef CPG_GENC_SOC_MAXDIM
            if (rows) {      // the family's largest cone as a compile-time bound: every load before the first sum (rows past the
                             // cone's end read the neighbouring vectors of the slice and are not used)
                double a[CPG_GENC_SOC_MAXDIM], b[CPG_GENC_SOC_MAXDIM];
#pragma unroll
                for (int r = 0; r < CPG_GENC_SOC_MAXDIM; r++) { a[r] = B.wv[st + (unsigned)r]; b[r] = v[st + (unsigned)r]; }
#pragma unroll
                for (int r = 0; r < CPG_GENC_SOC_MAXDIM; r++) { const double t = acc + a[r] * b[r]; acc = (unsigned)r < dm ? t : acc; }
            } else
#endif
            for (unsigned r = 0; r < dm; r++) acc += B.wv[st + r] * v[st + r];
            tmp[st] = acc;
        }
        cpgw::lds_order();
    }
    CPG_DEV unsigned ns_first() const { return m - 3u * (unsigned)C.n_ns; }
    // (W'W v)_i for row i; `dots` from soc_dots(v)
    CPG_DEV double hs_row(unsigned i, const double *v, const double *dots) const {
        if (i < (unsigned)C.n_zero) return 0.0;
        if (NS && i >= (unsigned)C.psd_first && i < ns_first()) return dots[i];
        if (NS && i >= ns_first()) {
            const unsigned r = (i - ns_first()) % 3u, st = i - r;
            const double v0 = v[st], v1 = v[st + 1u], v2 = v[st + 2u];
            if (r == 0u) return (B.hd[st] * v0 + B.wv[st] * v1) + B.wv[st + 1u] * v2;
            if (r == 1u) return (B.wv[st] * v0 + B.hd[st + 1u] * v1) + B.wv[st + 2u] * v2;
            return (B.wv[st + 1u] * v0 + B.wv[st + 2u] * v1) + B.hd[st + 2u] * v2;
        }
        const int st = cpgw::gld(C.row_cone, i);
        if (st < 0) return B.hd[i] * v[i];
        const double eta = B.et[i];
        const double t = 2.0 * dots[(unsigned)st];
        const double o = t * B.wv[i] + ((unsigned)st == i ? -v[i] : v[i]);
        return (eta * eta) * o;
    }
    // er = rb - K v (K without regularisation); returns ||er||_inf
    CPG_DEV double kkt_residual(const double *v) const {
        soc_dots(v + n, B.tz);
        double nrm = 0.0;
        for (unsigned j = (unsigned)lane; j < n; j += 64u) {
            const double r = B.rb[j] - (row_P(j, v) + col_At(j, v + n));
            B.er[j] = r;
            nrm = cpgw::dmax2(nrm, fabs(r));
        }
        for (unsigned i = (unsigned)lane; i < m; i += 64u) {
            const double r = B.rb[n + i] - (row_A(i, v) - hs_row(i, v + n, B.tz));
            B.er[n + i] = r;
            nrm = cpgw::dmax2(nrm, fabs(r));
        }
        cpgw::lds_order();
        return cpgw::wave_max_nonneg(nrm);
    }
    // out = (LDL')^{-1} in  (+ add, if given)
    CPG_DEV void ldl_apply(const LdsProg &SP, const double *in, double *out, const double *add) const {
        CPG_CONIC_REPEAT(2) ldl_apply_once(SP, in, out, add);
    }
    CPG_DEV void ldl_apply_once(const LdsProg &SP, const double *in, double *out, const double *add) const {
        for (unsigned i = (unsigned)lane; i < N; i += 64u) B.w[i] = in[i];
        cpgw::lds_order();
#ifdef CPG_GENC_HEADER
        if (C.gc_ok) run_program_conic(B.sv, C.gc_cols, C.gc_rows, B.w, lane);
        else
#endif
        run_program_lds<1>(SP, B.w, C.sol_slots, lane);
        for (unsigned i = (unsigned)lane; i < N; i += 64u) {
            const double r = B.w[(unsigned)cpgw::gld(C.sol_fpos, i)];
            out[i] = add ? add[i] + r : r;
        }
        cpgw::lds_order();
    }
    // sol = K^{-1} rb with iterative refinement
    CPG_DEV void kkt_solve(const LdsProg &SP) const {
        ldl_apply(SP, B.rb, B.sol, nullptr);
        if (!S.ir_enable) return;
        double nb = 0.0;
        for (unsigned i = (unsigned)lane; i < N; i += 64u) nb = cpgw::dmax2(nb, fabs(B.rb[i]));
        nb = cpgw::wave_max_nonneg(nb);
        double norme = 0.0;
        CPG_CONIC_REPEAT(4) norme = kkt_residual(B.sol);
#pragma nounroll
        for (int it = 0; it < S.ir_max_iter; it++) {
            if (norme <= S.ir_abstol + S.ir_reltol * nb) break;
            const double last = norme;
            ldl_apply(SP, B.er, B.cand, B.sol);
            double nn = 0.0;
            CPG_CONIC_REPEAT(4) nn = kkt_residual(B.cand);
            const double ratio = nn > 0.0 ? last / nn : CPG_INFTY;
            const bool stop = ratio < S.ir_stop_ratio;
            if (!stop || ratio > 1.0) {
                for (unsigned i = (unsigned)lane; i < N; i += 64u) B.sol[i] = B.cand[i];
                cpgw::lds_order();
            }
            if (stop) break;
            norme = nn;
        }
    }

    // ---- numeric factorisation -------------------------------------------------------------------
    CPG_DEV void factor() const {
        // static regularisation from the largest diagonal entry of K
        double eps = 0.0;
        if (S.static_reg_enable) {
            double md = 0.0;
#ifdef CPG_GENC_ROWS
            if (rows) {
                if ((unsigned)lane < n) {         // (n <= 64: row = lane)
#pragma unroll
                    for (int s = 0; s < CPG_GENC_ROWS_SP; s++) {
                        const unsigned w = rows[ROWS_P + s * 64 + lane];
                        if (CPG_GENC_ROW_VALID(w) && CPG_GENC_ROW_OP(w) == (unsigned)lane) md = cpgw::dmax2(md, fabs(B.P[CPG_GENC_ROW_ENT(w)]));
                    }
                }
            } else
#endif
            for (unsigned j = (unsigned)lane; j < n; j += 64u) {
                const unsigned a = (unsigned)cpgw::gld(C.Prp, j), e = (unsigned)cpgw::gld(C.Prp, j + 1u);
                for (unsigned k = a; k < e; k++)
                    if ((unsigned)cpgw::gld(C.Pcol, k) == j) md = cpgw::dmax2(md, fabs(B.P[(unsigned)cpgw::gld(C.Pent, k)]));
            }
            for (unsigned i = (unsigned)lane; i < m; i += 64u) md = cpgw::dmax2(md, fabs(B.hd[i]));
            md = cpgw::wave_max_nonneg(md);
            eps = S.static_const + S.static_prop * md;
        }
#ifdef CPG_GENC_FACTOR_HEADER
        if (C.gf_ok) conic_factor_gen(B, S, eps, lane);      // (this library's family: the schedule as straight-line code)
        else {
#else
        {
#endif
        int level_start = 0;
#pragma nounroll
        for (int c = 0; c < C.fac_chunks; c++) {
            const int L = cpgw::read_first_lane(cpgw::gld(C.fac_ctab, 4u * (unsigned)c));
            const int last = cpgw::read_first_lane(cpgw::gld(C.fac_ctab, 4u * (unsigned)c + 1u));
            unsigned base = (unsigned)cpgw::read_first_lane(cpgw::gld(C.fac_ctab, 4u * (unsigned)c + 2u));
            const int lg = cpgw::read_first_lane(cpgw::gld(C.fac_ctab, 4u * (unsigned)c + 3u));
            const unsigned task = cpgw::gld(C.fac_task, (unsigned)c * 64u + (unsigned)lane);
            const unsigned lw = cpgw::gld(C.fac_len, (unsigned)c * 64u + (unsigned)lane);
            const int len = (int)(lw & 0xFFFFu), rlen = (int)(lw >> 16);   // addressing length | real terms of this lane
            double acc = 0.0;
#pragma nounroll
            for (int s = 0; s < L; s++) {
                const bool act = s < len;
                if (s < rlen) {
                    const unsigned e = base + (unsigned)lane;
                    const double la = B.Lx[cpgw::gld(C.fac_a, e)];
                    const double lb = B.Lx[cpgw::gld(C.fac_b, e)];
                    const double dk = B.Dg[cpgw::gld(C.fac_k, e)];
                    acc = fma(la * dk, lb, acc);
                }
                base += cpgw::popc64(cpgw::ballot(act));
            }
            acc = cpgw::group_sum_first_dyn(acc, lg);           // dot products split over 2^lg lanes (refactor_plan._pack_tasks)
            if (task != 0xFFFFFFFFu) {
                const int kind = cpgw::gld(C.ksrc_kind, task);
                const unsigned idx = (unsigned)cpgw::gld(C.ksrc_idx, task);
                const bool piv = task >= (unsigned)C.nnzL;
                double kv = 0.0, sign = 1.0;
                if (kind == CPG_CK_P) kv = B.P[idx] + (piv ? eps : 0.0);
                else if (kind == CPG_CK_A) kv = B.A[idx];
                else if (kind == CPG_CK_DIAGX) kv = eps;
                else if (kind == CPG_CK_HDIAG) { kv = -B.hd[idx] - eps; sign = -1.0; }
                else if (kind == CPG_CK_HSOC) {
                    const unsigned i = idx & 0xFFFFu, j = idx >> 16;
                    kv = -((B.et[i] * B.et[i]) * (2.0 * (B.wv[i] * B.wv[j])));
                }
                else if (NS && kind == CPG_CK_HNS) kv = -B.wv[idx];
                else if (NS && kind == CPG_CK_HPSD)
                    kv = -psd::kkt_entry(B.psd + (idx & 0xFFFu), (int)((idx >> 12) & 0xFu), (int)((idx >> 16) & 7u), (int)((idx >> 19) & 7u),
                                         (int)((idx >> 22) & 7u), (int)((idx >> 25) & 7u));
                double v = kv - acc;
                if (piv) {
                    if (S.dynamic_reg_enable && v * sign < S.dyn_eps) v = S.dyn_delta * sign;
                    B.Dg[task - (unsigned)C.nnzL] = v;
                    B.Dginv[task - (unsigned)C.nnzL] = 1.0 / v;
                } else B.Lx[task] = v;
            }
            if (last) {   // level complete: divide the new columns by their pivots
                cpgw::lds_order();
#pragma nounroll
                for (int c2 = level_start; c2 <= c; c2++) {
                    const unsigned t2 = cpgw::gld(C.fac_task, (unsigned)c2 * 64u + (unsigned)lane);
                    if (t2 < (unsigned)C.nnzL) B.Lx[t2] = B.Lx[t2] * B.Dginv[(unsigned)cpgw::gld(C.Lcol, t2)];
                }
                cpgw::lds_order();
                level_start = c + 1;
            }
        }
        }
        for (unsigned e = (unsigned)lane; e < (unsigned)C.sol_nnz; e += 64u) {
            const int kind = cpgw::gld(C.sol_kind, e);
            const unsigned idx = (unsigned)cpgw::gld(C.sol_idx, e);
            double v = 0.0;
            if (kind == 1) v = 1.0;
            else if (kind == 2) v = -B.Lx[idx];
            else if (kind == 3) v = B.Dginv[idx];
            B.sv[e] = v;
        }
        cpgw::lds_order();
    }

    // ---- cone operations -------------------------------------------------------------------------
    CPG_DEV void identity_scaling() const {
        for (unsigned i = (unsigned)lane; i < m; i += 64u) {
            const int st = cpgw::gld(C.row_cone, i);
            B.et[i] = 1.0;
            B.wv[i] = (st < 0 || (unsigned)st == i) ? 1.0 : 0.0;
            B.lam[i] = 1.0;
            B.hd[i] = i < (unsigned)C.n_zero ? 0.0 : 1.0;
        }
        if (NS) {
            for (int k = lane; k < C.n_psd; k += 64) {
                const PsdRef r = psd_ref(k);
                for (int i = 0; i < r.p; i++) {
                    for (int j = 0; j < r.p; j++) { const double e = i == j ? 1.0 : 0.0; r.Q[i * r.p + j] = e; r.R[i * r.p + j] = e; r.Ri[i * r.p + j] = e; }
                    r.lam[i] = 1.0;
                }
            }
        }
        cpgw::lds_order();
    }
    // (min margin, sum of positive margins) of v over the nonnegative and second-order cones
    CPG_DEV void margins(const double *v, double &mn, double &pos) const {
        double a = CPG_INFTY, bsum = 0.0;
        for (unsigned i = (unsigned)C.n_zero + (unsigned)lane; i < (unsigned)(C.n_zero + C.n_nonneg); i += 64u) {
            a = cpgw::dmin2(a, v[i]);
            bsum += cpgw::dmax2(v[i], 0.0);
        }
        for (int k = lane; k < C.n_soc; k += 64) {
            const unsigned st = (unsigned)cpgw::gld(C.soc_start, (unsigned)k), dm = (unsigned)cpgw::gld(C.soc_dim, (unsigned)k);
            double ss = 0.0;
            for (unsigned r = 1; r < dm; r++) ss += v[st + r] * v[st + r];
            const double mg = v[st] - sqrt(ss);
            a = cpgw::dmin2(a, mg);
            bsum += cpgw::dmax2(0.0, mg);
        }
        if (NS) {
            for (int k = lane; k < C.n_psd; k += 64) {
                const PsdRef r = psd_ref(k);
                double *X = r.mat(0), *ev = r.vec(0);
                psd::svec_to_mat(v + r.st, r.p, X);
                psd::jacobi(r.p, X, nullptr, ev);
                for (int i = 0; i < r.p; i++) { a = cpgw::dmin2(a, ev[i]); bsum += cpgw::dmax2(0.0, ev[i]); }
            }
        }
        mn = cpgw::wave_min(a);
        pos = cpgw::wave_sum(bsum);
    }
    CPG_DEV void unit_shift(double *v, double a, bool primal) const {
        for (unsigned i = (unsigned)lane; i < m; i += 64u) {
            if (i < (unsigned)C.n_zero) { if (primal) v[i] = 0.0; continue; }
            const int st = cpgw::gld(C.row_cone, i);
            if (st < 0 || (unsigned)st == i) v[i] += a;
        }
        cpgw::lds_order();
    }
    CPG_DEV void shift_to_cone(double *v, bool primal) const {
        const int degree = C.n_nonneg + C.n_soc + (NS ? C.psd_degree : 0);
        if (degree == 0) { unit_shift(v, 0.0, primal); return; }
        double mn, pos;
        margins(v, mn, pos);
        const double target = cpgw::dmax2(1.0, 0.1 * pos / (double)degree);
        if (mn <= 0.0) { unit_shift(v, -mn, primal); unit_shift(v, target, primal); }
        else if (mn < target) unit_shift(v, target - mn, primal);
        else unit_shift(v, 0.0, primal);
    }
    // Nesterov-Todd scaling from (s, z); false if a second-order cone iterate left the cone.  Exponential / power cones:
    // gradient and Hessian of the dual barrier at z, block H_s by the strategy in force (mu: the iterate's duality measure)
    CPG_DEV bool update_scaling(double mu, bool dual_strategy) const {
        bool ok = true;
        if (NS) {
            for (int k = lane; k < C.n_ns; k += 64) {
                const unsigned st = ns_first() + 3u * (unsigned)k;
                const double alpha = cpgw::gld(C.ns_alpha, (unsigned)k);
                const double zk[3] = {B.z[st], B.z[st + 1u], B.z[st + 2u]}, sk[3] = {B.s[st], B.s[st + 1u], B.s[st + 2u]};
                ns::Zeta Z;
                ns::zeta(zk, alpha, Z);
                double grad[3], H[6], Hs[6];
                ns::dual_grad_hess(Z, grad, H);
                if (dual_strategy) { for (int t = 0; t < 6; t++) Hs[t] = mu * H[t]; }
                else ns::primal_dual_Hs(sk, zk, alpha, grad, H, Hs);
                B.et[st] = grad[0]; B.et[st + 1u] = grad[1]; B.et[st + 2u] = grad[2];
                B.hd[st] = Hs[0]; B.hd[st + 1u] = Hs[3]; B.hd[st + 2u] = Hs[5];
                B.wv[st] = Hs[1]; B.wv[st + 1u] = Hs[2]; B.wv[st + 2u] = Hs[4];
            }
            // PSD cones: S = L1 L1', Z = L2 L2', L2'L1 = U diag(lambda) V' (through the eigenvectors of its Gram matrix),
            // R = L1 V diag(lambda)^-1/2, R^-1 = diag(lambda)^-1/2 U'L2', Q = R R'
            for (int k = lane; k < C.n_psd; k += 64) {
                const PsdRef r = psd_ref(k);
                const int p = r.p;
                double *A = r.mat(0), *L1 = r.mat(1), *L2 = r.mat(2), *M = r.mat(3), *V = r.mat(4), *T = r.mat(5), *sig = r.vec(0), *isq = r.vec(1);
                psd::svec_to_mat(B.s + r.st, p, A);
                bool pd = psd::cholesky(p, A, L1);
                psd::svec_to_mat(B.z + r.st, p, A);
                pd = psd::cholesky(p, A, L2) && pd;
                if (!pd) { ok = false; continue; }
                psd::matmul(p, L2, true, L1, false, M);                 // M = L2'L1
                psd::matmul(p, M, true, M, false, A);                   // M'M = V diag(lambda^2) V'
                psd::jacobi(p, A, V, sig);
                for (int i = 0; i < p; i++) { sig[i] = sqrt(sig[i]); isq[i] = 1.0 / sqrt(sig[i]); }
                psd::matmul(p, L1, false, V, false, r.R);               // R = L1 V diag(isq)
                for (int i = 0; i < p; i++) for (int j = 0; j < p; j++) r.R[i * p + j] *= isq[j];
                psd::matmul(p, r.R, false, r.R, true, r.Q);             // Q = R R'
                psd::matmul(p, M, false, V, false, T);                  // U = M V diag(1 / lambda);  R^-1 = diag(isq) U'L2'
                for (int i = 0; i < p; i++) for (int j = 0; j < p; j++) T[i * p + j] /= sig[j];
                psd::matmul(p, T, true, L2, true, r.Ri);
                for (int i = 0; i < p; i++) for (int j = 0; j < p; j++) r.Ri[i * p + j] *= isq[i];
                int a = 0;
                for (int j = 0; j < p; j++)
                    for (int i = 0; i <= j; i++, a++) {
                        B.lam[r.st + (unsigned)a] = i == j ? sig[i] : 0.0;
                        B.hd[r.st + (unsigned)a] = psd::kkt_entry(r.Q, p, i, j, i, j);     // (Q just stored by this lane)
                        B.et[r.st + (unsigned)a] = 1.0;
                    }
                for (int i = 0; i < p; i++) r.lam[i] = sig[i];
            }
        }
        for (unsigned i = (unsigned)C.n_zero + (unsigned)lane; i < (unsigned)(C.n_zero + C.n_nonneg); i += 64u) {
            const double w = sqrt(B.s[i] / B.z[i]);
            B.wv[i] = w; B.et[i] = 1.0;
            B.lam[i] = sqrt(B.s[i] * B.z[i]);
            B.hd[i] = w * w;
        }
        for (int k = lane; k < C.n_soc; k += 64) {
            const unsigned st = (unsigned)cpgw::gld(C.soc_start, (unsigned)k), dm = (unsigned)cpgw::gld(C.soc_dim, (unsigned)k);
            const double s0 = B.s[st], z0 = B.z[st];
            double ss = 0.0, zz = 0.0, sz = s0 * z0;
            for (unsigned r = 1; r < dm; r++) {
                ss += B.s[st + r] * B.s[st + r]; zz += B.z[st + r] * B.z[st + r]; sz += B.s[st + r] * B.z[st + r];
            }
            const double rs = s0 * s0 - ss, rz = z0 * z0 - zz;
            if (!(rs > 0.0 && rz > 0.0)) { ok = false; continue; }
            const double sscale = sqrt(rs), zscale = sqrt(rz);
            const double gamma = sqrt(0.5 * (1.0 + sz / (sscale * zscale)));
            const double fs = 2.0 * sscale * gamma, fz = 2.0 * zscale * gamma;
            double w1sq = 0.0;
            for (unsigned r = 1; r < dm; r++) {
                const double w = B.s[st + r] / fs - B.z[st + r] / fz;
                B.wv[st + r] = w;
                w1sq += w * w;
            }
            const double w0 = sqrt(1.0 + w1sq);
            B.wv[st] = w0;
            const double eta = sqrt(sscale / zscale);
            double zeta = 0.0;
            for (unsigned r = 1; r < dm; r++) zeta += B.wv[st + r] * B.z[st + r];
            B.lam[st] = eta * (w0 * z0 + zeta);
            const double f = z0 + zeta / (1.0 + w0);
            for (unsigned r = 1; r < dm; r++) B.lam[st + r] = eta * (B.z[st + r] + f * B.wv[st + r]);
            const double e2 = eta * eta;
            for (unsigned r = 0; r < dm; r++) {
                B.et[st + r] = eta;
                const double w = B.wv[st + r];
                B.hd[st + r] = e2 * (2.0 * (w * w) - (r == 0 ? 1.0 : -1.0));
            }
        }
        cpgw::lds_order();
        return !cpgw::wave_any(!ok);
    }
    // largest a in [0, amax] with v + a dv in the cone
    CPG_DEV double step_length(const double *v, const double *dv, double amax) const {
        double a = amax;
        for (unsigned i = (unsigned)C.n_zero + (unsigned)lane; i < (unsigned)(C.n_zero + C.n_nonneg); i += 64u)
            if (dv[i] < 0.0) a = cpgw::dmin2(a, -v[i] / dv[i]);
        for (int k = lane; k < C.n_soc; k += 64) {
            const unsigned st = (unsigned)cpgw::gld(C.soc_start, (unsigned)k), dm = (unsigned)cpgw::gld(C.soc_dim, (unsigned)k);
            double yy = 0.0, xy = 0.0, xx = 0.0;
#ifdef CPG_GENC_SOC_MAXDIM
            if (rows) {
                double a[CPG_GENC_SOC_MAXDIM], b[CPG_GENC_SOC_MAXDIM];
#pragma unroll
                for (int r = 1; r < CPG_GENC_SOC_MAXDIM; r++) { a[r] = v[st + (unsigned)r]; b[r] = dv[st + (unsigned)r]; }
#pragma unroll
                for (int r = 1; r < CPG_GENC_SOC_MAXDIM; r++) {
                    const bool in = (unsigned)r < dm;
                    const double t0 = yy + b[r] * b[r], t1 = xy + a[r] * b[r], t2 = xx + a[r] * a[r];
                    yy = in ? t0 : yy; xy = in ? t1 : xy; xx = in ? t2 : xx;
                }
            } else
#endif
            for (unsigned r = 1; r < dm; r++) { yy += dv[st + r] * dv[st + r]; xy += v[st + r] * dv[st + r]; xx += v[st + r] * v[st + r]; }
            const double qa = dv[st] * dv[st] - yy;
            const double qb = 2.0 * (v[st] * dv[st] - xy);
            const double qc = cpgw::dmax2(0.0, v[st] * v[st] - xx);
            const double disc = qb * qb - 4.0 * qa * qc;
            double r = CPG_INFTY;
            if (!((qa > 0.0 && qb > 0.0) || disc < 0.0) && qa != 0.0) {
                const double t = qb >= 0.0 ? (-qb - sqrt(disc)) : (-qb + sqrt(disc));
                double r1 = t != 0.0 ? (2.0 * qc) / t : CPG_INFTY;
                double r2 = t / (2.0 * qa);
                if (r1 < 0.0) r1 = CPG_INFTY;
                if (r2 < 0.0) r2 = CPG_INFTY;
                r = cpgw::dmin2(r1, r2);
            }
            a = cpgw::dmin2(a, r);
        }
        return cpgw::wave_min(a);
    }
    // PSD cones: largest a <= a0 with lambda + a W dz and lambda + a W^-T ds in the cone -- the smallest eigenvalue of
    // diag(lambda)^-1/2 mat(.) diag(lambda)^-1/2
    CPG_DEV double psd_step_length(double a0) const {
        double a = a0;
        for (int k = lane; k < C.n_psd; k += 64) {
            const PsdRef r = psd_ref(k);
            const int p = r.p;
            double *X = r.mat(0), *T = r.mat(1), *Y = r.mat(2), *isq = r.vec(1);
            for (int i = 0; i < p; i++) isq[i] = 1.0 / sqrt(r.lam[i]);
#pragma nounroll
            for (int side = 0; side < 2; side++) {
                psd::svec_to_mat((side == 0 ? B.dz : B.ds) + r.st, p, X);
                psd::congruence(p, side == 0 ? r.R : r.Ri, side == 0, X, T, Y);           // R' dZ R  |  R^-1 dS R^-T
                for (int i = 0; i < p; i++) for (int j = 0; j < p; j++) Y[i * p + j] *= isq[i] * isq[j];
                const double g = psd::eig_min(p, Y, r.vec(0));
                if (g < 0.0) a = cpgw::dmin2(a, -1.0 / g);
            }
        }
        return cpgw::wave_min(a);
    }
    // exponential / power cones: backtracking from a on the cone tests of z + a dz and s + a ds
    CPG_DEV double ns_step_length(double a0) const {
        double a = a0;
        for (int k = lane; k < C.n_ns; k += 64) {
            const unsigned st = ns_first() + 3u * (unsigned)k;
            const double alpha = cpgw::gld(C.ns_alpha, (unsigned)k);
#pragma nounroll
            for (int side = 0; side < 2; side++) {
                const double *v = side == 0 ? B.z : B.s, *dv = side == 0 ? B.dz : B.ds;
                const double v0 = v[st], v1 = v[st + 1u], v2 = v[st + 2u], d0 = dv[st], d1 = dv[st + 1u], d2 = dv[st + 2u];
                double ak = a0;
#pragma nounroll
                for (;;) {
                    const double w[3] = {v0 + ak * d0, v1 + ak * d1, v2 + ak * d2};
                    if (side == 0 ? ns::dual_feasible(w, alpha) : ns::primal_feasible(w, alpha)) break;
                    ak *= S.ls_backtrack;
                    if (ak < S.min_terminate_step) { ak = 0.0; break; }
                }
                a = cpgw::dmin2(a, ak);
            }
        }
        return cpgw::wave_min(a);
    }
    // centrality function of the dual scaling strategy at the trial point (s + a ds, z + a dz, tau + a dtau, kappa + a dkappa)
    CPG_DEV double barrier(double a, double tau, double kap, double dtau, double dkap) const {
        const double ct = tau + a * dtau, ck = kap + a * dkap;
        double sz = 0.0, acc = 0.0;
        for (unsigned i = (unsigned)lane; i < m; i += 64u) {
            const double sn = B.s[i] + a * B.ds[i], zn = B.z[i] + a * B.dz[i];
            sz = fma(sn, zn, sz);
            if (i >= (unsigned)C.n_zero && i < (unsigned)(C.n_zero + C.n_nonneg)) acc -= ns::logsafe(sn * zn);
        }
        for (int k = lane; k < C.n_soc; k += 64) {
            const unsigned st = (unsigned)cpgw::gld(C.soc_start, (unsigned)k), dm = (unsigned)cpgw::gld(C.soc_dim, (unsigned)k);
            double s0 = B.s[st] + a * B.ds[st], z0 = B.z[st] + a * B.dz[st], ss = 0.0, zz = 0.0;
            for (unsigned r = 1; r < dm; r++) {
                const double sn = B.s[st + r] + a * B.ds[st + r], zn = B.z[st + r] + a * B.dz[st + r];
                ss += sn * sn; zz += zn * zn;
            }
            const double rs = s0 * s0 - ss, rz = z0 * z0 - zz;
            acc += (rs > 0.0 && rz > 0.0) ? -0.5 * ns::logsafe(rs * rz) : CPG_NS_INF;
        }
        for (int k = lane; k < C.n_ns; k += 64) {
            const unsigned st = ns_first() + 3u * (unsigned)k;
            const double alpha = cpgw::gld(C.ns_alpha, (unsigned)k);
            const double zn[3] = {B.z[st] + a * B.dz[st], B.z[st + 1u] + a * B.dz[st + 1u], B.z[st + 2u] + a * B.dz[st + 2u]};
            const double sn[3] = {B.s[st] + a * B.ds[st], B.s[st + 1u] + a * B.ds[st + 1u], B.s[st + 2u] + a * B.ds[st + 2u]};
            acc += ns::barrier_dual(zn, alpha) + ns::barrier_primal(sn, alpha);
        }
        for (int k = lane; k < C.n_psd; k += 64) {          // -log det of both trial matrices through their Cholesky factors
            const PsdRef r = psd_ref(k);
            double *v = r.mat(2), *X = r.mat(0), *L = r.mat(1);
#pragma nounroll
            for (int side = 0; side < 2; side++) {
                const double *x = side == 0 ? B.s : B.z, *dx = side == 0 ? B.ds : B.dz;
                for (int t = 0; t < r.d; t++) v[t] = x[r.st + (unsigned)t] + a * dx[r.st + (unsigned)t];
                psd::svec_to_mat(v, r.p, X);
                if (!psd::cholesky(r.p, X, L)) { acc = CPG_NS_INF; continue; }
                for (int i = 0; i < r.p; i++) acc -= 2.0 * log(L[i * r.p + i]);
            }
        }
        sz = cpgw::wave_sum(sz);
        const bool inf = cpgw::wave_any(!(acc < CPG_NS_INF));       // (+inf in any lane: the sum is +inf whatever the other lanes hold)
        acc = cpgw::wave_sum(acc);
        const int degree = C.n_nonneg + C.n_soc + 3 * C.n_ns + C.psd_degree;
        const double mu = (sz + ct * ck) / (double)(degree + 1);
        const double val = (double)(degree + 1) * ns::logsafe(mu) - ns::logsafe(ct) - ns::logsafe(ck) + acc;
        return inf ? CPG_NS_INF : val;
    }
    // dsc = W'(lambda \ (lambda o lambda + (W^-1 ds) o (W dz) - sigma mu e)), zero-cone rows 0;
    // exponential / power cones: dsc = s + sigma mu grad f*(z) - eta(ds, dz)
    CPG_DEV void combined_ds_offset(double sigmamu) const {
        if (NS) {
            for (int k = lane; k < C.n_ns; k += 64) {
                const unsigned st = ns_first() + 3u * (unsigned)k;
                const double alpha = cpgw::gld(C.ns_alpha, (unsigned)k);
                const double zk[3] = {B.z[st], B.z[st + 1u], B.z[st + 2u]};
                const double dsk[3] = {B.ds[st], B.ds[st + 1u], B.ds[st + 2u]}, dzk[3] = {B.dz[st], B.dz[st + 1u], B.dz[st + 2u]};
                double eta[3];
                ns::higher_correction(zk, alpha, dsk, dzk, eta);
                for (unsigned r = 0; r < 3u; r++) B.dsc[st + r] = B.s[st + r] + sigmamu * B.et[st + r] - eta[r];
            }
            // PSD cones: a = W^-T ds = R^-1 dS R^-T, b = W dz = R' dZ R, d = diag(lambda^2) + (a b + b a) / 2 - sigma mu I,
            // u_ij = 2 d_ij / (lambda_i + lambda_j), dsc = W'u = R U R'
            for (int k = lane; k < C.n_psd; k += 64) {
                const PsdRef r = psd_ref(k);
                const int p = r.p;
                double *X = r.mat(0), *T = r.mat(1), *Am = r.mat(2), *Bm = r.mat(3), *AB = r.mat(4), *BA = r.mat(5);
                psd::svec_to_mat(B.ds + r.st, p, X);
                psd::congruence(p, r.Ri, false, X, T, Am);
                psd::svec_to_mat(B.dz + r.st, p, X);
                psd::congruence(p, r.R, true, X, T, Bm);
                psd::matmul(p, Am, false, Bm, false, AB);
                psd::matmul(p, Bm, false, Am, false, BA);
                for (int i = 0; i < p; i++)
                    for (int j = 0; j < p; j++) {
                        const double li = r.lam[i], lj = r.lam[j];
                        const double d = (i == j ? li * li - sigmamu : 0.0) + 0.5 * (AB[i * p + j] + BA[i * p + j]);
                        X[i * p + j] = 2.0 * d / (li + lj);
                    }
                psd::congruence(p, r.R, false, X, T, Am);              // R U R'
                psd::mat_to_svec(Am, p, B.dsc + r.st);
            }
        }
        for (unsigned i = (unsigned)lane; i < (unsigned)(C.n_zero + C.n_nonneg); i += 64u) {
            if (i < (unsigned)C.n_zero) { B.dsc[i] = 0.0; continue; }
            const double w = B.wv[i], lm = B.lam[i];
            const double d = lm * lm + (B.ds[i] / w) * (B.dz[i] * w) - sigmamu;
            B.dsc[i] = (d / lm) * w;
        }
        for (int k = lane; k < C.n_soc; k += 64) {
            const unsigned st = (unsigned)cpgw::gld(C.soc_start, (unsigned)k), dm = (unsigned)cpgw::gld(C.soc_dim, (unsigned)k);
            const double eta = B.et[st], w0 = B.wv[st];
            double *a = B.tz + st, *bb = B.dsc + st;
            const double *lm = B.lam + st, *w = B.wv + st;
            // a = W^-1 ds, bb = W dz
            double zs = 0.0, zz = 0.0;
            for (unsigned r = 1; r < dm; r++) { zs += w[r] * B.ds[st + r]; zz += w[r] * B.dz[st + r]; }
            const double ds0 = B.ds[st], dz0 = B.dz[st];
            a[0] = (w0 * ds0 - zs) / eta;
            bb[0] = eta * (w0 * dz0 + zz);
            const double fa = -ds0 + zs / (1.0 + w0), fb = dz0 + zz / (1.0 + w0);
            for (unsigned r = 1; r < dm; r++) {
                a[r] = (B.ds[st + r] + fa * w[r]) / eta;
                bb[r] = eta * (B.dz[st + r] + fb * w[r]);
            }
            // d = lam o lam + a o bb - sigma mu e   (into a)
            double ll = 0.0, ab = 0.0;
            for (unsigned r = 0; r < dm; r++) { ll += lm[r] * lm[r]; ab += a[r] * bb[r]; }
            const double a0 = a[0], b0 = bb[0];
            for (unsigned r = 1; r < dm; r++) a[r] = (lm[0] * lm[r] + lm[0] * lm[r]) + (a0 * bb[r] + b0 * a[r]);
            a[0] = ll + ab - sigmamu;
            // u = lam \ d   (into a)
            double l1 = 0.0, ld = 0.0;
            for (unsigned r = 1; r < dm; r++) { l1 += lm[r] * lm[r]; ld += lm[r] * a[r]; }
            const double p = lm[0] * lm[0] - l1;
            const double u0 = (lm[0] * a[0] - ld) / p;
            for (unsigned r = 1; r < dm; r++) a[r] = (a[r] - u0 * lm[r]) / lm[0];
            a[0] = u0;
            // out = W u   (into dsc)
            double zu = 0.0;
            for (unsigned r = 1; r < dm; r++) zu += w[r] * a[r];
            bb[0] = eta * (w0 * u0 + zu);
            const double fu = u0 + zu / (1.0 + w0);
            for (unsigned r = 1; r < dm; r++) bb[r] = eta * (a[r] + fu * w[r]);
        }
        cpgw::lds_order();
    }
};
typedef ConicCtxT<false> ConicCtx;

// Solves rb = (rhs_x, dsc - rhs_z) and assembles the step (dx, dz, ds, dtau, dkappa) of the
// homogeneous embedding; x2 / z2 is the constant part K^{-1}(-q, b), `den` its denominator.
// (The solve sol = (x1, z1) = K^{-1} rb itself is issued by the caller: the iteration has ONE inlined copy of kkt_solve.)
template <bool NS>
CPG_DEV void conic_step(const ConicCtxT<NS> &cx, double rhs_tau, double rhs_kap, double tau, double kap,
                        double den, double &dtau, double &dkap) {
    const ConicBuf &B = cx.B;
    const unsigned n = cx.n, m = cx.m;
    const int lane = cx.lane;
    for (unsigned j = (unsigned)lane; j < n; j += 64u) B.tx[j] = cx.row_P(j, B.sol);
    cpgw::lds_order();
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (unsigned j = (unsigned)lane; j < n; j += 64u) { a1 = fma(B.q[j], B.sol[j], a1); a3 = fma(B.x[j] / tau, B.tx[j], a3); }
    for (unsigned i = (unsigned)lane; i < m; i += 64u) a2 = fma(B.b[i], B.sol[n + i], a2);
    a1 = cpgw::wave_sum(a1); a2 = cpgw::wave_sum(a2); a3 = cpgw::wave_sum(a3);
    const double num = rhs_tau - rhs_kap / tau + a1 + a2 + 2.0 * a3;
    dtau = num / den;
    for (unsigned j = (unsigned)lane; j < n; j += 64u) B.dx[j] = B.sol[j] + dtau * B.x2[j];
    for (unsigned i = (unsigned)lane; i < m; i += 64u) B.dz[i] = B.sol[n + i] + dtau * B.z2[i];
    cpgw::lds_order();
    cx.soc_dots(B.dz, B.tz);
    for (unsigned i = (unsigned)lane; i < m; i += 64u) B.ds[i] = -(cx.hs_row(i, B.dz, B.tz) + B.dsc[i]);
    cpgw::lds_order();
    dkap = -(rhs_kap + kap * dtau) / tau;
}

// Block-shared LDS copy of every index table of the family (patterns, factorisation schedule,
// substitution program): the interior-point loop chases these indices in every sparse product and
// every chunk; an L2 round trip per index bounded the first version of the kernel (ADP: 20.0 ms with
// the tables in L2, 15.8 ms with the LDS copy, 8 waves per workgroup, 100 000 instances).
template <typename T>
CPG_DEV const T *conic_stage(const T *src, unsigned count, double *&cur) {
    T *dst = (T *)cur;
    for (unsigned t = cpgw::thread_in_block(); t < count; t += cpgw::block_threads()) dst[t] = cpgw::gld(src, t);
    cur += ((size_t)count * sizeof(T) + 7) / 8;
    return dst;
}

// SPECIALISED (family library, handle whose family is the one the library was generated for): the family's dimensions
// are compile-time constants (CPG_GENC_SPECIALISE of the generated header), so every vector of the interior-point state
// sits at a constant offset from the wave's LDS base, every staged table at a constant offset from the block's, and the
// loops over n / m / nnz have known trip counts -- instead of ~80 wave-uniform pointers kept in (and spilled from) SGPRs.
template <bool TABLES_IN_LDS, bool SPECIALISED = false, bool NONSYM = false>
CPG_DEV void clarabel_body(const DevConic &C0, const DevConicSettings &S, const DevBatch &Bt, double *lds, int /*wave_global*/) {
    DevConic C = C0;
#ifdef CPG_GENC_HEADER
    if (SPECIALISED) { CPG_GENC_SPECIALISE(C) }
#endif
    const genc_row_word *rows = nullptr;
    if (TABLES_IN_LDS) {
        double *cur = lds;
        const unsigned n1 = (unsigned)C.n + 1u, m1 = (unsigned)C.m + 1u, NN = (unsigned)(C.n + C.m);
        C.soc_start = conic_stage(C0.soc_start, (unsigned)C.n_soc, cur);
        C.soc_dim = conic_stage(C0.soc_dim, (unsigned)C.n_soc, cur);
        C.row_cone = conic_stage(C0.row_cone, (unsigned)C.m, cur);
#ifdef CPG_GENC_ROWS
        // the library's own family: the generated row words stand for every walk over the two patterns (ConicCtx::rows)
        if (SPECIALISED) rows = conic_stage(genc_row_words, (unsigned)CPG_GENC_ROWS_WORDS, cur);
        else
#endif
        {
            C.Ap = conic_stage(C0.Ap, n1, cur); C.Ai = conic_stage(C0.Ai, (unsigned)C.nnzA, cur);
            C.Arp = conic_stage(C0.Arp, m1, cur); C.Aent = conic_stage(C0.Aent, (unsigned)C.nnzA, cur);
            C.Acol = conic_stage(C0.Acol, (unsigned)C.nnzA, cur);
            C.Pp = conic_stage(C0.Pp, n1, cur); C.Pi = conic_stage(C0.Pi, (unsigned)C.nnzP, cur);
            C.Prp = conic_stage(C0.Prp, n1, cur); C.Pent = conic_stage(C0.Pent, (unsigned)C.n_pfull, cur);
            C.Pcol = conic_stage(C0.Pcol, (unsigned)C.n_pfull, cur);
        }
        C.Lcol = conic_stage(C0.Lcol, (unsigned)C.nnzL, cur);
        C.ksrc_kind = conic_stage(C0.ksrc_kind, (unsigned)C.nnzL + NN, cur);
        C.ksrc_idx = conic_stage(C0.ksrc_idx, (unsigned)C.nnzL + NN, cur);
        C.fac_ctab = conic_stage(C0.fac_ctab, (unsigned)C.fac_chunks * 4u, cur);
        C.fac_task = conic_stage(C0.fac_task, (unsigned)C.fac_chunks * 64u, cur);
        C.fac_len = conic_stage(C0.fac_len, (unsigned)C.fac_chunks * 64u, cur);
        C.fac_a = conic_stage(C0.fac_a, (unsigned)C.fac_triples, cur);
        C.fac_b = conic_stage(C0.fac_b, (unsigned)C.fac_triples, cur);
        C.fac_k = conic_stage(C0.fac_k, (unsigned)C.fac_triples, cur);
#ifdef CPG_GENC_HEADER
        // a family library: the generated executor reads its own two tables instead of the program's three.  (For another
        // family the program's tables stay in global memory: staging one set or the other under a run-time branch would
        // turn every table pointer of both executors into a flat one.)
        C.gc_cols = conic_stage(C0.gc_cols, (unsigned)C.gc_ncols, cur);     // (no entries for another family)
        C.gc_rows = conic_stage(C0.gc_rows, (unsigned)C.gc_nrows, cur);
#else
        C.sol_ctab = conic_stage(C0.sol_ctab, (unsigned)C.sol_chunks * 4u, cur);
        C.sol_desc = conic_stage(C0.sol_desc, (unsigned)C.sol_chunks * 64u, cur);
        C.sol_cols = conic_stage(C0.sol_cols, (unsigned)C.sol_nnz, cur);
#endif
        C.sol_kind = conic_stage(C0.sol_kind, (unsigned)C.sol_nnz, cur);
        C.sol_idx = conic_stage(C0.sol_idx, (unsigned)C.sol_nnz, cur);
        C.sol_fpos = conic_stage(C0.sol_fpos, NN, cur);
        lds += C.tab_doubles;
        cpgw::block_sync();
    }
    const int lane = cpgw::lane_id();
    const unsigned n = (unsigned)C.n, m = (unsigned)C.m, N = n + m;
    const ConicBuf B = conic_carve(lds + (size_t)cpgw::wave_in_block() * (size_t)C.lds_doubles, C);
    const ConicCtxT<NONSYM> cx{C, S, B, lane, n, m, N, rows};
    LdsProg SP;
    SP.ctab = C.sol_ctab; SP.desc = C.sol_desc; SP.vals = B.sv; SP.cols = C.sol_cols;
    SP.n_chunks = C.sol_chunks; SP.dummy = (unsigned)C.sol_nnz - 1u; SP.rows16 = nullptr;
    const int degree = C.n_nonneg + C.n_soc + (NONSYM ? 3 * C.n_ns + C.psd_degree : 0);
    const bool nonsym = NONSYM && C.n_ns > 0;     // (the extended instantiation also serves families whose only extra cones are PSD: symmetric)
    // zero padding behind the entries, dummy slots and zero slot behind the work vector (generated executor)
    for (int t = lane; t < C.sv_pad; t += 64) B.sv[C.sol_nnz + t] = 0.0;
    for (int t = lane; t < C.w_extra; t += 64) B.w[C.sol_slots + t] = 0.0;
    cpgw::lds_order();

    for (;;) {
        unsigned ig = 0;
        if (lane == 0) ig = cpgw::atomic_next(Bt.counter);
        ig = (unsigned)cpgw::read_first_lane((int)ig);
        if ((long long)ig >= Bt.B) break;
        const long long bi = (long long)ig;
        const double *theta = Bt.theta + (size_t)bi * C.np_var;

        // ---- 1. canonicalise: cpg_canonicalize_* + cpg_copy_* (utils.py:279-294, 987-1006)
        for (unsigned k = (unsigned)lane; k < (unsigned)C.nnzA; k += 64u) B.A[k] = csr_row(C.map_A, k, theta, cpgw::gld(C.A_base, k));
        for (unsigned k = (unsigned)lane; k < (unsigned)C.nnzP; k += 64u) B.P[k] = csr_row(C.map_P, k, theta, cpgw::gld(C.P_base, k));
        double normq = 0.0, normb = 0.0;
        for (unsigned j = (unsigned)lane; j < n; j += 64u) {
            const double v = csr_row(C.map_q, j, theta, cpgw::gld(C.q_base, j));
            B.q[j] = v; B.D[j] = 1.0; normq = cpgw::dmax2(normq, fabs(v));
        }
        for (unsigned i = (unsigned)lane; i < m; i += 64u) {
            const double v = csr_row(C.map_b, i, theta, cpgw::gld(C.b_base, i));
            B.b[i] = v; B.E[i] = 1.0; normb = cpgw::dmax2(normb, fabs(v));
        }
        const double dconst = csr_row(C.map_d, 0, theta, C.d_base);
        normq = cpgw::wave_max_nonneg(normq); normb = cpgw::wave_max_nonneg(normb);
        cpgw::lds_order();

        // ---- 2. equilibration (in place; D, E cumulative, c the cost scaling)
        double cs = 1.0;
        if (S.equilibrate_enable) {
#pragma nounroll
            for (int it = 0; it < S.equilibrate_max_iter; it++) {
                for (unsigned j = (unsigned)lane; j < n; j += 64u) {
                    double acc = 0.0;
#ifdef CPG_GENC_ROWS
                    if (rows) acc = cx.template row_absmax<CPG_GENC_ROWS_SAT>(ConicCtx::ROWS_AT, j, cx.template row_absmax<CPG_GENC_ROWS_SP>(ConicCtx::ROWS_P, j, 0.0));
                    else
#endif
                    {
                        unsigned a = (unsigned)cpgw::gld(C.Prp, j), e = (unsigned)cpgw::gld(C.Prp, j + 1u);
                        for (unsigned k = a; k < e; k++) acc = cpgw::dmax2(acc, fabs(B.P[(unsigned)cpgw::gld(C.Pent, k)]));
                        a = (unsigned)cpgw::gld(C.Ap, j); e = (unsigned)cpgw::gld(C.Ap, j + 1u);
                        for (unsigned k = a; k < e; k++) acc = cpgw::dmax2(acc, fabs(B.A[k]));
                    }
                    acc = acc == 0.0 ? 1.0 : acc;
                    acc = acc < S.eq_min ? S.eq_min : (acc > S.eq_max ? S.eq_max : acc);
                    B.tx[j] = 1.0 / sqrt(acc);
                }
                for (unsigned i = (unsigned)lane; i < m; i += 64u) {
                    double acc = 0.0;
#ifdef CPG_GENC_ROWS
                    if (rows) acc = cx.template row_absmax<CPG_GENC_ROWS_SA>(ConicCtx::ROWS_A, i, 0.0);
                    else
#endif
                    {
                        const unsigned a = (unsigned)cpgw::gld(C.Arp, i), e = (unsigned)cpgw::gld(C.Arp, i + 1u);
                        for (unsigned k = a; k < e; k++) acc = cpgw::dmax2(acc, fabs(B.A[(unsigned)cpgw::gld(C.Aent, k)]));
                    }
                    acc = acc == 0.0 ? 1.0 : acc;
                    acc = acc < S.eq_min ? S.eq_min : (acc > S.eq_max ? S.eq_max : acc);
                    B.tz[i] = 1.0 / sqrt(acc);
                }
                cpgw::lds_order();
                for (unsigned j = (unsigned)lane; j < n; j += 64u) {
                    const double dj = B.tx[j];
#ifdef CPG_GENC_ROWS
                    if (rows) {
                        // an entry of the upper triangle of P is scaled by the lane of its ROW (operand = its column >= the row;
                        // the mirrored occurrence in the column's list is skipped), an entry of A by the lane of its column
#pragma unroll
                        for (int s_ = 0; s_ < CPG_GENC_ROWS_SP; s_++) {
                            const unsigned w = rows[ConicCtx::ROWS_P + s_ * 64 + (int)j];
                            const unsigned e = CPG_GENC_ROW_ENT(w), o = CPG_GENC_ROW_OP(w);
                            if (CPG_GENC_ROW_VALID(w) && o >= j) B.P[e] = (dj * B.P[e]) * B.tx[o];
                        }
#pragma unroll
                        for (int s_ = 0; s_ < CPG_GENC_ROWS_SAT; s_++) {
                            const unsigned w = rows[ConicCtx::ROWS_AT + s_ * 64 + (int)j];
                            const unsigned e = CPG_GENC_ROW_ENT(w), o = CPG_GENC_ROW_OP(w);
                            if (CPG_GENC_ROW_VALID(w)) B.P[e] = (B.tz[o] * B.P[e]) * dj;
                        }
                    } else
#endif
                    {
                        unsigned a = (unsigned)cpgw::gld(C.Pp, j), e = (unsigned)cpgw::gld(C.Pp, j + 1u);
                        for (unsigned k = a; k < e; k++) B.P[k] = (B.tx[(unsigned)cpgw::gld(C.Pi, k)] * B.P[k]) * dj;
                        a = (unsigned)cpgw::gld(C.Ap, j); e = (unsigned)cpgw::gld(C.Ap, j + 1u);
                        for (unsigned k = a; k < e; k++) B.A[k] = (B.tz[(unsigned)cpgw::gld(C.Ai, k)] * B.A[k]) * dj;
                    }
                    B.q[j] = dj * B.q[j];
                    B.D[j] *= dj;
                }
                for (unsigned i = (unsigned)lane; i < m; i += 64u) { B.b[i] = B.tz[i] * B.b[i]; B.E[i] *= B.tz[i]; }
                cpgw::lds_order();
                double psum = 0.0, qn = 0.0;
                for (unsigned j = (unsigned)lane; j < n; j += 64u) {
                    double acc = 0.0;
#ifdef CPG_GENC_ROWS
                    if (rows) acc = cx.template row_absmax<CPG_GENC_ROWS_SP>(ConicCtx::ROWS_P, j, 0.0);
                    else
#endif
                    {
                        const unsigned a = (unsigned)cpgw::gld(C.Prp, j), e = (unsigned)cpgw::gld(C.Prp, j + 1u);
                        for (unsigned k = a; k < e; k++) acc = cpgw::dmax2(acc, fabs(B.P[(unsigned)cpgw::gld(C.Pent, k)]));
                    }
                    psum += acc;
                    qn = cpgw::dmax2(qn, fabs(B.q[j]));
                }
                psum = cpgw::wave_sum(psum); qn = cpgw::wave_max_nonneg(qn);
                const double pn = n ? psum / (double)n : 0.0;
                if (pn != 0.0 && qn != 0.0) {
                    double ct = 1.0 / cpgw::dmax2(pn, qn);
                    ct = ct < S.eq_min ? S.eq_min : (ct > S.eq_max ? S.eq_max : ct);
                    for (unsigned k = (unsigned)lane; k < (unsigned)C.nnzP; k += 64u) B.P[k] *= ct;
                    for (unsigned j = (unsigned)lane; j < n; j += 64u) B.q[j] *= ct;
                    cs *= ct;
                    cpgw::lds_order();
                }
            }
            // second-order cone rows must share one scale: mean of the cone
            for (int k = lane; k < C.n_soc; k += 64) {
                const unsigned st = (unsigned)cpgw::gld(C.soc_start, (unsigned)k), dm = (unsigned)cpgw::gld(C.soc_dim, (unsigned)k);
                double sum = 0.0;
                for (unsigned r = 0; r < dm; r++) sum += B.E[st + r];
                const double mean = sum / (double)dm;
                for (unsigned r = 0; r < dm; r++) B.tz[st + r] = mean / B.E[st + r];
            }
            if (NONSYM) {
                for (int k = lane; k < C.n_psd; k += 64) {        // PSD cones: one scale per cone as well
                    const unsigned st = (unsigned)cpgw::gld(C.psd_start, (unsigned)k);
                    const int pp = cpgw::gld(C.psd_dim, (unsigned)k), dm = pp * (pp + 1) / 2;
                    double sum = 0.0;
                    for (int r = 0; r < dm; r++) sum += B.E[st + (unsigned)r];
                    const double mean = sum / (double)dm;
                    for (int r = 0; r < dm; r++) B.tz[st + (unsigned)r] = mean / B.E[st + (unsigned)r];
                }
                // exponential / power cones admit no row scaling at all: back to 1
                for (unsigned i = cx.ns_first() + (unsigned)lane; i < m; i += 64u) B.tz[i] = 1.0 / B.E[i];
            }
            cpgw::lds_order();
            if (C.n_soc > 0 || (NONSYM && (C.n_ns > 0 || C.n_psd > 0))) {
                const unsigned first = (unsigned)(C.n_zero + C.n_nonneg);
                for (unsigned i = first + (unsigned)lane; i < m; i += 64u) {
                    const double ew = B.tz[i];
#ifdef CPG_GENC_ROWS
                    if (rows) {
#pragma unroll
                        for (int s_ = 0; s_ < CPG_GENC_ROWS_SA; s_++) {
                            const unsigned w = rows[ConicCtx::ROWS_A + s_ * 64 + (int)i];
                            if (CPG_GENC_ROW_VALID(w)) B.P[CPG_GENC_ROW_ENT(w)] *= ew;
                        }
                    } else
#endif
                    {
                        const unsigned a = (unsigned)cpgw::gld(C.Arp, i), e = (unsigned)cpgw::gld(C.Arp, i + 1u);
                        for (unsigned k = a; k < e; k++) B.A[(unsigned)cpgw::gld(C.Aent, k)] *= ew;
                    }
                    B.b[i] *= ew; B.E[i] *= ew;
                }
                cpgw::lds_order();
            }
        }
        const double cinv = 1.0 / cs;

        // ---- 3. initial point: identity scaling, one factorisation, shift into the cones
        if (nonsym) {
            // a nonsymmetric cone anywhere: x = 0 and every cone at its central point s = z
            cx.identity_scaling();           // (the zero-cone rows of the scaling vectors are written here and nowhere else)
            for (unsigned j = (unsigned)lane; j < n; j += 64u) B.x[j] = 0.0;
            for (unsigned i = (unsigned)lane; i < m; i += 64u) {
                double v = 0.0;
                if (i >= cx.ns_first()) {
                    const unsigned r = (i - cx.ns_first()) % 3u;
                    const double alpha = cpgw::gld(C.ns_alpha, (i - cx.ns_first()) / 3u);
                    if (alpha == 0.0) v = r == 0u ? -1.051383945322714 : (r == 1u ? 0.556409619469370 : 1.258967884768947);
                    else v = r == 0u ? sqrt(1.0 + alpha) : (r == 1u ? sqrt(2.0 - alpha) : 0.0);
                } else if (i >= (unsigned)C.n_zero) {
                    const int st = cpgw::gld(C.row_cone, i);
                    v = (st < 0 || (unsigned)st == i) ? 1.0 : 0.0;
                }
                B.s[i] = v; B.z[i] = v;
            }
        } else {
        cx.identity_scaling();
        cx.factor();
        {   // one solve (x, z) = K^{-1}(-q, b), s = -z; for P == 0 two: (x, -s) = K^{-1}(0, b), then (., z) = K^{-1}(-q, 0).  One
            // inlined copy of kkt_solve here as well.
            const int nsolve = C.p_is_zero ? 2 : 1;
            for (unsigned j = (unsigned)lane; j < n; j += 64u) B.rb[j] = C.p_is_zero ? 0.0 : -B.q[j];
            for (unsigned i = (unsigned)lane; i < m; i += 64u) B.rb[n + i] = B.b[i];
#pragma nounroll
            for (int k = 0; k < nsolve; k++) {
                cpgw::lds_order();
                cx.kkt_solve(SP);
                if (!C.p_is_zero) {
                    for (unsigned j = (unsigned)lane; j < n; j += 64u) B.x[j] = B.sol[j];
                    for (unsigned i = (unsigned)lane; i < m; i += 64u) { const double zv = B.sol[n + i]; B.z[i] = zv; B.s[i] = -zv; }
                } else if (k == 0) {
                    for (unsigned j = (unsigned)lane; j < n; j += 64u) { B.x[j] = B.sol[j]; B.rb[j] = -B.q[j]; }
                    for (unsigned i = (unsigned)lane; i < m; i += 64u) { B.s[i] = -B.sol[n + i]; B.rb[n + i] = 0.0; }
                } else {
                    for (unsigned i = (unsigned)lane; i < m; i += 64u) B.z[i] = B.sol[n + i];
                }
            }
        }
        cpgw::lds_order();
        cx.shift_to_cone(B.s, true);
        cx.shift_to_cone(B.z, false);
        }
        cpgw::lds_order();
        double tau = 1.0, kap = 1.0;
        bool dual_strategy = false;      // scaling strategy of the nonsymmetric cones: primal-dual until a checkpoint switches to dual

        // ---- 4. interior-point iterations
        int status = CPG_CL_UNSOLVED, iter = 0, almost = CPG_CL_UNSOLVED;
        double cost_p = 0.0, res_p = 0.0, res_d = 0.0;
        double prev_cost_p = CPG_INFTY, prev_res_p = CPG_INFTY, prev_res_d = CPG_INFTY, prev_gap_abs = CPG_INFTY, prev_gap_rel = CPG_INFTY;
        double prev_tau = 1.0, prev_kap = 1.0;
#pragma nounroll
        for (;;) {
            // residuals: tx = P x, rx = -P x - A'z - q tau, rz = A x + s - b tau
            for (unsigned j = (unsigned)lane; j < n; j += 64u) B.tx[j] = cx.row_P(j, B.x);
            cpgw::lds_order();
            double dqx = 0.0, dbz = 0.0, dsz = 0.0, xPx = 0.0;
            double n_x = 0.0, n_z = 0.0, n_s = 0.0, n_rxinf = 0.0, n_px = 0.0, n_rzinf = 0.0, n_rz = 0.0, n_rx = 0.0;
            for (unsigned j = (unsigned)lane; j < n; j += 64u) {
                const double xj = B.x[j], px = B.tx[j], di = 1.0 / B.D[j];
                const double rinf = -cx.col_At(j, B.z);
                const double r = rinf - px - B.q[j] * tau;
                B.rx[j] = r;
                dqx = fma(B.q[j], xj, dqx); xPx = fma(xj, px, xPx);
                n_x = cpgw::dmax2(n_x, fabs(B.D[j] * xj)); n_rxinf = cpgw::dmax2(n_rxinf, fabs(di * rinf));
                n_px = cpgw::dmax2(n_px, fabs(di * px)); n_rx = cpgw::dmax2(n_rx, fabs(di * r));
            }
            for (unsigned i = (unsigned)lane; i < m; i += 64u) {
                const double zi = B.z[i], si = B.s[i], ei = 1.0 / B.E[i];
                const double rinf = cx.row_A(i, B.x) + si;
                const double r = rinf - B.b[i] * tau;
                B.rz[i] = r;
                dbz = fma(B.b[i], zi, dbz); dsz = fma(si, zi, dsz);
                n_z = cpgw::dmax2(n_z, fabs(B.E[i] * zi)); n_s = cpgw::dmax2(n_s, fabs(ei * si));
                n_rzinf = cpgw::dmax2(n_rzinf, fabs(ei * rinf)); n_rz = cpgw::dmax2(n_rz, fabs(ei * r));
            }
            cpgw::lds_order();
            dqx = cpgw::wave_sum(dqx); dbz = cpgw::wave_sum(dbz); dsz = cpgw::wave_sum(dsz); xPx = cpgw::wave_sum(xPx);
            n_x = cpgw::wave_max_nonneg(n_x); n_z = cpgw::wave_max_nonneg(n_z) * cinv; n_s = cpgw::wave_max_nonneg(n_s);
            n_rxinf = cpgw::wave_max_nonneg(n_rxinf); n_px = cpgw::wave_max_nonneg(n_px);
            n_rzinf = cpgw::wave_max_nonneg(n_rzinf); n_rz = cpgw::wave_max_nonneg(n_rz); n_rx = cpgw::wave_max_nonneg(n_rx);
            const double rtau = dqx + dbz + kap + xPx / tau;
            const double mu = (dsz + tau * kap) / (double)(degree + 1);
            // termination quantities on the unscaled problem
            const double tinv = 1.0 / tau;
            cost_p = (dqx * tinv + 0.5 * xPx * tinv * tinv) * cinv;
            const double cost_d = (-dbz * tinv - 0.5 * xPx * tinv * tinv) * cinv;
            const double res_pinf = n_rxinf / cpgw::dmax2(1.0, n_z);
            const double res_dinf = cpgw::dmax2(n_px / cpgw::dmax2(1.0, n_x), n_rzinf / cpgw::dmax2(1.0, n_x + n_s));
            const double nx = n_x * tinv, nz = n_z * tinv, ns = n_s * tinv;
            res_p = n_rz * tinv / cpgw::dmax2(1.0, normb + nx + ns);
            res_d = n_rx * tinv * cinv / cpgw::dmax2(1.0, normq + nx + nz);
            double gap_abs = fabs(cost_p - cost_d);
            double gap_rel = gap_abs / cpgw::dmax2(1.0, cpgw::dmin2(fabs(cost_p), fabs(cost_d)));
            const double ktratio = kap / tau;
            const double bz = dbz * cinv, qx = dqx * cinv;
            // check_convergence: optimality at kappa/tau <= 1, certificates once kappa/tau > 1000 / tol_ktratio
            auto verdict = [&](double t_gap_abs, double t_gap_rel, double t_feas, double t_inf_abs, double t_inf_rel, double t_kt,
                               int solved, int pinf, int dinf) -> int {
                if (ktratio <= 1.0 && (gap_abs < t_gap_abs || gap_rel < t_gap_rel) && res_p < t_feas && res_d < t_feas) return solved;
                if (ktratio > 1000.0 / t_kt) {
                    if (bz < -t_inf_abs && res_pinf < -t_inf_rel * bz) return pinf;
                    if (qx < -t_inf_abs && res_dinf < -t_inf_rel * qx) return dinf;
                }
                return CPG_CL_UNSOLVED;
            };
            status = verdict(S.tol_gap_abs, S.tol_gap_rel, S.tol_feas, S.tol_infeas_abs, S.tol_infeas_rel, S.tol_ktratio,
                             CPG_CL_SOLVED, CPG_CL_PRIMAL_INFEASIBLE, CPG_CL_DUAL_INFEASIBLE);
            // poor progress: the residuals went up at round-off level with the previous gap inside its tolerance, or by a
            // factor 100 out of the feasibility tolerance -> stop, back on the previous iterate and its figures
            if (status == CPG_CL_UNSOLVED && iter > 1 && (res_d > prev_res_d || res_p > prev_res_p)) {
                if (ktratio < 100.0 * 2.220446049250313e-16 && (prev_gap_abs < S.tol_gap_abs || prev_gap_rel < S.tol_gap_rel))
                    status = CPG_CL_INSUFFICIENT_PROGRESS;
                if ((res_d > S.tol_feas && res_d > 100.0 * prev_res_d) || (res_p > S.tol_feas && res_p > 100.0 * prev_res_p))
                    status = CPG_CL_INSUFFICIENT_PROGRESS;
                if (nonsym && status == CPG_CL_INSUFFICIENT_PROGRESS && !dual_strategy) {
                    // strategy checkpoint: the primal-dual scaling gets a second chance as the dual scaling, from this iterate
                    status = CPG_CL_UNSOLVED; dual_strategy = true;
                    prev_cost_p = cost_p; prev_res_p = res_p; prev_res_d = res_d; prev_gap_abs = gap_abs; prev_gap_rel = gap_rel;
                    continue;
                }
                if (status == CPG_CL_INSUFFICIENT_PROGRESS) {
                    for (unsigned j = (unsigned)lane; j < n; j += 64u) B.x[j] = B.px[j];
                    for (unsigned i = (unsigned)lane; i < m; i += 64u) { B.z[i] = B.pz[i]; B.s[i] = B.ps[i]; }
                    cpgw::lds_order();
                    tau = prev_tau; kap = prev_kap;
                    cost_p = prev_cost_p; res_p = prev_res_p; res_d = prev_res_d; gap_abs = prev_gap_abs; gap_rel = prev_gap_rel;
                }
            }
            if (status == CPG_CL_UNSOLVED && iter >= S.max_iter) status = CPG_CL_MAX_ITERATIONS;
            // what these figures are worth at the reduced tolerances: the status of a solve that ends in an error or at
            // the iteration limit (post_process of the published solver)
            almost = verdict(S.red_gap_abs, S.red_gap_rel, S.red_feas, S.red_infeas_abs, S.red_infeas_rel, S.red_ktratio,
                             CPG_CL_ALMOST_SOLVED, CPG_CL_ALMOST_PRIMAL_INFEASIBLE, CPG_CL_ALMOST_DUAL_INFEASIBLE);
            if (status != CPG_CL_UNSOLVED) break;
            prev_cost_p = cost_p; prev_res_p = res_p; prev_res_d = res_d; prev_gap_abs = gap_abs; prev_gap_rel = gap_rel;
            iter++;

            // scaling, factorisation, constant part (x2, z2) = K^{-1}(-q, b)
            bool scaled = true;
            CPG_CONIC_REPEAT(8) scaled = cx.update_scaling(mu, dual_strategy);
            if (!scaled) { status = CPG_CL_NUMERICAL_ERROR; break; }
            CPG_CONIC_REPEAT(1) cx.factor();
            // The three solves of an iteration -- constant part (x2, z2) = K^{-1}(-q, b), affine step, combined step -- run
            // through ONE copy of kkt_solve (substitution sweeps + refinement): inlined three times, the loop body
            // outgrew the instruction cache two CUs share.  Same operations in the same order as the straight-line form.
            double den = 0.0, dtau = 0.0, dkap = 0.0, alpha = 1.0, sigma = 0.0, rk = 0.0;
            bool nonfinite = false;
#pragma nounroll
            for (int pass = 0; pass < 3; pass++) {
                if (pass == 0) {
                    for (unsigned j = (unsigned)lane; j < n; j += 64u) B.rb[j] = -B.q[j];
                    for (unsigned i = (unsigned)lane; i < m; i += 64u) B.rb[n + i] = B.b[i];
                } else if (pass == 1) {          // affine step: rhs (rx, rz, rtau, tau kappa), ds offset = s
                    for (unsigned j = (unsigned)lane; j < n; j += 64u) B.rb[j] = B.rx[j];
                    for (unsigned i = (unsigned)lane; i < m; i += 64u) { B.dsc[i] = B.s[i]; B.rb[n + i] = B.s[i] - B.rz[i]; }
                } else {                         // combined step
                    CPG_CONIC_REPEAT(32) cx.combined_ds_offset(sigma * mu);
                    rk = -sigma * mu + dtau * dkap + tau * kap;
                    for (unsigned j = (unsigned)lane; j < n; j += 64u) B.rb[j] = (1.0 - sigma) * B.rx[j];
                    for (unsigned i = (unsigned)lane; i < m; i += 64u) B.rb[n + i] = B.dsc[i] - (1.0 - sigma) * B.rz[i];
                }
                cpgw::lds_order();
                cx.kkt_solve(SP);
                if (pass == 0) {
                    for (unsigned j = (unsigned)lane; j < n; j += 64u) { B.x2[j] = B.sol[j]; B.cand[j] = B.x[j] / tau - B.sol[j]; }
                    for (unsigned i = (unsigned)lane; i < m; i += 64u) B.z2[i] = B.sol[n + i];
                    cpgw::lds_order();
                    double qx2 = 0.0, bz2 = 0.0, vPv = 0.0, x2Px2 = 0.0;
                    for (unsigned j = (unsigned)lane; j < n; j += 64u) {
                        qx2 = fma(B.q[j], B.x2[j], qx2);
                        vPv = fma(B.cand[j], cx.row_P(j, B.cand), vPv);
                        x2Px2 = fma(B.x2[j], cx.row_P(j, B.x2), x2Px2);
                    }
                    for (unsigned i = (unsigned)lane; i < m; i += 64u) bz2 = fma(B.b[i], B.z2[i], bz2);
                    qx2 = cpgw::wave_sum(qx2); bz2 = cpgw::wave_sum(bz2); vPv = cpgw::wave_sum(vPv); x2Px2 = cpgw::wave_sum(x2Px2);
                    den = kap / tau - qx2 - bz2 + vPv - x2Px2;
                } else {
                    const double rhs_tau = pass == 1 ? rtau : (1.0 - sigma) * rtau;
                    const double rhs_kap = pass == 1 ? tau * kap : rk;
                    CPG_CONIC_REPEAT(64) conic_step(cx, rhs_tau, rhs_kap, tau, kap, den, dtau, dkap);
                    CPG_CONIC_REPEAT(16) {
                        alpha = 1.0;
                        if (dtau < 0.0) alpha = cpgw::dmin2(alpha, -tau / dtau);
                        if (dkap < 0.0) alpha = cpgw::dmin2(alpha, -kap / dkap);
                        alpha = cx.step_length(B.z, B.dz, alpha);         // (symmetric cones first)
                        alpha = cx.step_length(B.s, B.ds, alpha);
                        if (NONSYM && C.n_psd > 0) alpha = cx.psd_step_length(alpha);
                        // back off from a full step so that the logarithms are not taken at the boundary, then backtrack
                        if (nonsym) alpha = cx.ns_step_length(cpgw::dmin2(alpha, S.max_step_fraction));
                    }
                    if (nonsym) {      // a step that is not finite: numerical-error checkpoint
                        bool bad = !(fabs(dtau) < CPG_NS_INF);
                        for (unsigned j = (unsigned)lane; j < n; j += 64u) bad = bad || !(fabs(B.dx[j]) < CPG_NS_INF);
                        for (unsigned i = (unsigned)lane; i < m; i += 64u) bad = bad || !(fabs(B.dz[i]) < CPG_NS_INF);
                        if (cpgw::wave_any(bad)) { nonfinite = true; break; }
                    }
                    if (pass == 1) sigma = (1.0 - alpha) * (1.0 - alpha) * (1.0 - alpha);
                }
            }
            if (nonsym && nonfinite) {
                if (!dual_strategy) { dual_strategy = true; continue; }
                status = CPG_CL_NUMERICAL_ERROR; break;
            }
            alpha *= S.max_step_fraction;
            if (nonsym && dual_strategy) {       // centrality: back to where the sum of the barriers is below 1
#pragma nounroll
                for (int t = 0; t < 50; t++) {
                    if (cx.barrier(alpha, tau, kap, dtau, dkap) < 1.0) break;
                    alpha *= S.ls_backtrack;
                }
            }
            if (nonsym && !dual_strategy && alpha < S.min_switch_step) { dual_strategy = true; continue; }   // small-step checkpoint
            if (alpha <= cpgw::dmax2(0.0, S.min_terminate_step)) { status = CPG_CL_INSUFFICIENT_PROGRESS; break; }   // undersized step
            for (unsigned j = (unsigned)lane; j < n; j += 64u) { const double v = B.x[j], d = B.dx[j]; B.px[j] = v; B.x[j] = v + alpha * d; }
            for (unsigned i = (unsigned)lane; i < m; i += 64u) {
                const double sv = B.s[i], zv = B.z[i], dsv = B.ds[i], dzv = B.dz[i];
                B.ps[i] = sv; B.pz[i] = zv;                 // (px / pz / ps ARE dx / dz / ds: read the step first)
                B.s[i] = sv + alpha * dsv; B.z[i] = zv + alpha * dzv;
            }
            prev_tau = tau; prev_kap = kap;
            tau += alpha * dtau; kap += alpha * dkap;
            cpgw::lds_order();
        }

        // ---- 5. retrieve: cpg_retrieve_prim / _dual / _info (utils.py:1040-1046; clarabel.py:37-46)
        if ((status == CPG_CL_MAX_ITERATIONS || status == CPG_CL_NUMERICAL_ERROR || status == CPG_CL_INSUFFICIENT_PROGRESS) &&
            almost != CPG_CL_UNSOLVED) status = almost;
        const bool infeasible = status == CPG_CL_PRIMAL_INFEASIBLE || status == CPG_CL_DUAL_INFEASIBLE ||
                                status == CPG_CL_ALMOST_PRIMAL_INFEASIBLE || status == CPG_CL_ALMOST_DUAL_INFEASIBLE;
        const double scale = infeasible ? 1.0 : 1.0 / tau;
        for (unsigned k = (unsigned)lane; k < (unsigned)C.n_prim; k += 64u) {
            const unsigned j = (unsigned)cpgw::gld(C.prim_idx, k);
            Bt.prim[(size_t)bi * C.n_prim + k] = B.D[j] * B.x[j] * scale;
        }
        for (unsigned k = (unsigned)lane; k < (unsigned)C.n_dual; k += 64u) {
            const unsigned i = (unsigned)cpgw::gld(C.dual_idx, k);
            Bt.dual[(size_t)bi * C.n_dual + k] = B.E[i] * B.z[i] * scale / cs;
        }
        if (lane == 0) {
            double ov = infeasible ? NAN : cost_p + dconst;
            if (C.is_max) ov = -ov;
            Bt.obj[bi] = ov; Bt.iter[bi] = iter; Bt.status[bi] = status; Bt.pri_res[bi] = res_p; Bt.dua_res[bi] = res_d;
        }
        cpgw::lds_order();
    }
}

}  // namespace cpg
