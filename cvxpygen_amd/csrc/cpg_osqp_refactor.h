// Per-instance refactorisation path of the batched OSQP backend: what the reference's generated
// cpg_solve() does when a parameter enters P or A --
//   cpg_canonicalize_P / _A            (cvxpygen/utils.py:279-294)
//   osqp_update_data_mat(...)          (cvxpygen/solvers/osqp.py:20-33; third-party OSQP: unscale,
//                                       overwrite values, Ruiz-equilibrate from scratch, numeric
//                                       LDL' on the fixed symbolic pattern)
//   osqp_update_data_vec, osqp_solve, cpg_retrieve_*   as in cpg_osqp_kernel.h
// -- for every instance of a batch, one wavefront per instance.  All structure (row / column views,
// KKT value sources, LDL' dot-product schedule, level-scheduled substitution program) is shared and
// prepared on the host (cvxpygen_amd/refactor_plan.py); per-instance numbers (scaled matrices,
// factor, substitution coefficients) live in a per-wavefront buffer in HBM / L2.
//
// This path favours generality over speed: every coefficient of the per-instance factor is
// streamed from memory in every ADMM iteration, and the substitution is level-scheduled without
// the partitioned-inverse merging of the shared-factor path.
#pragma once

#include "cpg_osqp_kernel.h"

#ifdef CPG_GENI_HEADER
// straight-line executor of the family's per-instance substitution program in shared-matrix mode, coefficients in
// registers (cvxpygen_amd/codegen.py::emit_instance_program)
#include CPG_GENI_HEADER
#endif
#ifdef CPG_GENS_HEADER
// straight-line executor of the family's per-instance substitution program on per-instance matrices: coefficients and tables
// from global memory at static addresses (cvxpygen_amd/codegen.py::stream_header)
#include CPG_GENS_HEADER
#define CPG_GENS_ACTIVE(geni, shared, crlds) (!(geni) && !(shared) && !(crlds))
#else
#define CPG_GENS_ACTIVE(geni, shared, crlds) false
#endif

namespace cpg {

#ifdef CPG_GENI_HEADER
// chunks {steps, first step} of the natural-layout row programs of the shared-matrix handle (canonical order) as
// literals of the generated family: the termination test's products request the loads of a whole chunk together
struct GeniRows {
    static constexpr int len(int which, int s) {
        constexpr int a[][2] = CPG_GENI_AROWS; constexpr int p[][2] = CPG_GENI_PROWS; constexpr int t[][2] = CPG_GENI_ATROWS;
        return which == 0 ? (s < CPG_GENI_AROWS_N ? a[s < CPG_GENI_AROWS_N ? s : 0][0] : 0)
             : which == 1 ? (s < CPG_GENI_PROWS_N ? p[s < CPG_GENI_PROWS_N ? s : 0][0] : 0)
                          : (s < CPG_GENI_ATROWS_N ? t[s < CPG_GENI_ATROWS_N ? s : 0][0] : 0);
    }
    static constexpr int off(int which, int s) {
        constexpr int a[][2] = CPG_GENI_AROWS; constexpr int p[][2] = CPG_GENI_PROWS; constexpr int t[][2] = CPG_GENI_ATROWS;
        return which == 0 ? a[s < CPG_GENI_AROWS_N ? s : 0][1] : which == 1 ? p[s < CPG_GENI_PROWS_N ? s : 0][1]
                                                                            : t[s < CPG_GENI_ATROWS_N ? s : 0][1];
    }
};
#define CPG_GENI_ROWS(P, which, s, w, lane) natural_chunk_lit(P, GeniRows::len(which, s), GeniRows::off(which, s), w, lane)
#endif

struct DevRefactor {
    int nnzP, nnzA, nnzL, n_eq, np_var, scaling_iters;
    // equilibration views
    const int *Ap, *Ai, *Arp, *Aent, *Acol, *Pp, *Pi, *Prp, *Pent, *Pcol;
    // factor
    const int *Lcol, *ksrc_kind, *ksrc_idx;
    const int *fac_ctab;
    const unsigned *fac_task, *fac_len, *fac_a, *fac_b, *fac_k;
    int fac_chunks;
    // substitution program (tables shared, values per instance) in the layout of run_program_stream
    const int *sol_kind, *sol_idx;
    const unsigned short *sol_fpos;
    int sol_nnz, sol_slots;
    const unsigned *sol_stab;     // pair table and entry table of run_program_stream; sol_nnz, sol_kind and
    const unsigned *sol_cr;       // sol_idx are in ITS entry layout (cpg_hip_set_refactor builds all of
    int sol_pairs;                // them from the plan's ctab / desc / cols / kind / idx)
    // canonicalisation of everything (UNSCALED): p = base + map @ theta_var
    const double *P_base, *A_base, *q_base, *u_base;
    const double *q_setup;   // unscaled q of the code-generation-time workspace (cost scaling sees this one)
    double d_base;
    DevCsr map_P, map_A, map_q, map_u, map_d;
    long long buf_doubles;   // per-wavefront buffer length
    // Shared-matrix mode (no varying parameter enters P or A; the instances own a factor only because their rho
    // differs -- OSQP's adapt_rho, or a row that changed class): the family's equilibrated matrices and scaling
    // vectors, canonical order; q_base / u_base / map_q / map_u are then pre-scaled (c D q, E u) like
    // DevUpdate's and steps 1 - 3 of the kernel (canonicalise P / A, Ruiz sweeps) are skipped.
    int shared_mats;
    const double *Ps, *As, *Ars, *Ds, *Dinvs, *Es, *Einvs;
    double cs;
    // Generated instance executor (family libraries, shared-matrix mode; cpg_hip_set_refactor checks that the
    // uploaded program is the one the library was generated for): operand byte offsets of all 64 lanes of every
    // step, four steps side by side ([step / 4][lane][4]; idle lanes: the zero slot), output slot | segment mask << 13
    // of every (chunk, lane), four chunks side by side, and per (step, lane) where the coefficient comes from
    // (kind << 28 | index; kind 1: 1.0, 2: -L[index], 3: 1 / d[index], 0: none).
    int gi_ok;
    const unsigned short *gi_cols, *gi_rows;
    const unsigned *gi_src;
    const unsigned short *gi_lcol;      // [step][lane] column of the L entry behind a kind-2 coefficient
    // generated factorisation of shared-matrix mode (numeric_ldl_gen of the family's cpg_instance_<name>.h): operand positions
    // a | b << 16 | k << 32 per (step, lane), destination | (rho row + 1) << 16 per (chunk, lane); see codegen.emit_factor_program
    const unsigned long long *gf_tri;
    const unsigned *gf_dk;
    // generated streaming executor (run_program_gens of cpg_stream_<name>.h, per-instance-matrix mode): operand offsets
    // [step / 4][lane][4] and output slots [chunk / 4][lane][4] in global memory; the coefficients stay in program-entry order
    const unsigned short *gs_cols, *gs_rows;
    const int *gs_kind, *gs_idx;        // value sources in program-entry order (+ 64 entries of kind 0)
    int gs_ok, gs_nnz;
    // shared-matrix mode: the KKT value of every destination of the factorisation is a family constant (fac_kc; sigma
    // included on the pivots) except the -1 / rho_vec of the (2,2) diagonal: fac_krow = its row, -1 elsewhere
    const double *fac_kc;
    const int *fac_krow;
    const double *fac_kc_cl;            // the same per (chunk, lane) of the factorisation schedule (no detour over the task number)
    const int *fac_krow_cl;
    const int *fac_kind_cl, *fac_idx_cl;   // per (chunk, lane): KKT source of the destination (per-instance matrices)
};

#define CPG_K_NONE 0
#define CPG_K_P 1
#define CPG_K_A 2
#define CPG_K_SIGMA 3
#define CPG_K_RHO 4

CPG_DEV double lim_scaling(double v) { v = v < 1e-4 ? 1.0 : v; return v > 1e4 ? 1e4 : v; }

// per-wavefront buffer layout (doubles)
struct InstBuf {
    double *P, *A, *D, *Dinv, *E, *Einv, *q, *u, *rinv, *Lx, *Dg, *Dginv, *sv;
    double *Ar;     // the entries of A once more in ROW order (row walks then need no entry-number indirection)
};
CPG_DEV InstBuf carve(double *b, const DevFamily &F, const DevRefactor &R) {
    InstBuf o;
    const size_t n = (size_t)F.n, m = (size_t)F.m, N = n + m;
    o.P = b; b += R.nnzP; o.A = b; b += R.nnzA;
    o.D = b; b += n; o.Dinv = b; b += n; o.E = b; b += m; o.Einv = b; b += m;
    o.q = b; b += n; o.u = b; b += m; o.rinv = b; b += m;
    o.Lx = b; b += R.nnzL; o.Dg = b; b += N; o.Dginv = b; b += N;
    o.sv = b; b += R.sol_nnz > R.gs_nnz ? R.sol_nnz : R.gs_nnz;
    o.Ar = b; b += R.nnzA;
    return o;
}

// CPG_REFACTOR_ROW_COPY (family libraries of families with long rows of A, codegen.family_library_defs): the row
// walks of the equilibration sweeps and of the termination test read a ROW-ordered copy of A's entries instead of
// going through the entry numbers -- one dependent memory round trip per four entries instead of two; a
// 101-entry row (portfolio family) is 26 such groups long and everything else of its 64-row slot waits for it.
// Portfolio: 105.2 -> 101.6 ms per 20 000; families with short rows only pay for the copy (MPC 12/4/10 with every
// parameter: 176 -> 182 ms), so the generic library and their family libraries leave it off.
// B.Ar[k] = B.A[entry number of the k-th entry in row order]: one coalesced pass per instance and scaling state
#ifndef CPG_ROW_COPY_BATCH
#define CPG_ROW_COPY_BATCH 8      // entries of a row requested together where the row-ordered copy is walked (4: 101.4, 8: 99.6 ms; 16 spills in the ADMM loop)
#endif
CPG_DEV void refresh_row_copy(const DevRefactor &R, const InstBuf &B, int lane) {
    cpgw::mem_order();
    for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzA; k += 64u)
        cpgw::gst(B.Ar, k, cpgw::gld((const double *)B.A, (unsigned)cpgw::gld(R.Aent, k)));
    cpgw::mem_order();
}

// Walks row r of a sparse pattern (ptr / optional entry numbers / columns) over the instance's values
// and calls f(value, column) for every entry in storage order.  An entry costs a chain of dependent
// loads (entry number -> value), so four entries are requested together before f consumes them.
#ifndef CPG_ROW_WALK_BATCH
#define CPG_ROW_WALK_BATCH 4       // entries requested together in every other row / column walk
#endif
template <bool ENT, int NB = CPG_ROW_WALK_BATCH, class Fn>
CPG_DEV void for_row_entries(const int *ptr, const int *ent, const int *col, const double *val, unsigned r, Fn f) {
    const unsigned a = (unsigned)cpgw::gld(ptr, r), e = (unsigned)cpgw::gld(ptr, r + 1u);
    for (unsigned k = a; k < e; k += (unsigned)NB) {
        unsigned en[NB], co[NB];
#pragma unroll
        for (int t = 0; t < NB; t++) {
            const unsigned kk = k + (unsigned)t < e ? k + (unsigned)t : a;
            en[t] = ENT ? (unsigned)cpgw::gld(ent, kk) : kk;
            co[t] = (unsigned)cpgw::gld(col, kk);
        }
        double av[NB];
#pragma unroll
        for (int t = 0; t < NB; t++) av[t] = cpgw::gld(val, en[t]);
#pragma unroll
        for (int t = 0; t < NB; t++)
            if (k + (unsigned)t < e) f(av[t], co[t]);
    }
}

// row products with the instance's own scaled matrices (values gathered from the buffer)
template <int NSX, int NSZ, bool QUMEM = false>
struct InstCtx {
    // check() runs the infeasibility tests itself, in OSQP's order, with the caller's lane id: on this kernel the
    // form that keeps the shared-factor kernel spill-free costs the ADMM loop its register allocation
    // (portfolio family: 0.334 instead of 0.293 ms per iteration of 20 000 instances, 30 instead of 2 scratch
    // instructions per lane and iteration; HISTORY.md 4.2)
    static constexpr bool kTestsFirst = false;
    // ... but the generated instance kernel (QUMEM) gets a per-call lane id in check(): with the kernel-wide one the
    // addresses of the per-row scaling reads (instance-invariant) were computed once outside the instance loop, kept
    // in scratch and reloaded one by one in front of their reads -- 379 "reload address, wait, load" sequences
    static constexpr bool kOpaqueLane = QUMEM;
    const DevFamily &F;
    const DevRefactor &R;
    const InstBuf &B;
    const double *w;
    int lane;
    // the instance's scaled q and u stay in registers for the whole ADMM loop (they are read in every iteration;
    // a global load there is a full memory latency on the critical path) -- or, QUMEM, in the wavefront's LDS slice
    // (the generated instance executor keeps its coefficients in registers and needs these 2 (NSX + NSZ) back)
    const double (&qr)[NSX];
    const double (&ur)[NSZ];
    const double *qm, *um;
    // shared-matrix mode: the termination test's products run through the family's natural-layout row programs
    // (coalesced, shared by every wavefront: cache hits) instead of per-lane walks over the instance's matrices --
    // a walk is a chain of dependent round trips per row (pointer -> entry number -> value)
    bool rowprog;
    // (QUMEM = the generated instance kernel: chunk tables of the row programs are literals, and the row programs are
    // what it always runs -- cpg_hip_set_refactor enables it only with them --, so the per-lane walks below are not
    // even compiled into it: their hoisted per-row addresses were half of its scratch)
    CPG_DEV double rows_gen(const DevProgram &P, int which, int s) const {
#ifdef CPG_GENI_HEADER
        return CPG_GENI_ROWS(P, which, s, w, lane);
#else
        return natural_chunk(P, s, w, lane);
#endif
    }
    CPG_DEV void products(int) const {}
    CPG_DEV void stage(int) const {}
    CPG_DEV double sE(unsigned i) const { return cpgw::gld(F.E, i); }
    CPG_DEV double sEinv(unsigned i) const { return cpgw::gld(F.Einv, i); }
    CPG_DEV double sD(unsigned i) const { return cpgw::gld(F.D, i); }
    CPG_DEV double sDinv(unsigned i) const { return cpgw::gld(F.Dinv, i); }
    CPG_DEV double q(int s, unsigned i) const { return QUMEM ? qm[i] : qr[s]; }
    CPG_DEV double u(int s, unsigned i) const { return QUMEM ? um[i] : ur[s]; }
    template <bool ENT, bool OFFS, int NB = CPG_ROW_WALK_BATCH>
    CPG_DEV double row_dot(const int *ptr, const int *ent, const int *col, const double *val, unsigned r) const {
        double acc = 0.0;
        const double *wv = w + (OFFS ? (unsigned)F.n : 0u);
        for_row_entries<ENT, NB>(ptr, ent, col, val, r, [&](double v, unsigned c) { acc = fma(v, wv[c], acc); });
        return acc;
    }
    CPG_DEV double ax(int s) const {
        if (QUMEM || rowprog) return QUMEM ? rows_gen(F.A_rows, 0, s) : natural_chunk(F.A_rows, s, w, lane);
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
#ifdef CPG_REFACTOR_ROW_COPY
        return i < (unsigned)F.m ? row_dot<false, false, CPG_ROW_COPY_BATCH>(R.Arp, nullptr, R.Acol, (const double *)B.Ar, i) : 0.0;
#else
        return i < (unsigned)F.m ? row_dot<true, false>(R.Arp, R.Aent, R.Acol, (const double *)B.A, i) : 0.0;
#endif
    }
    CPG_DEV double px(int s) const {
        if (QUMEM || rowprog) return QUMEM ? rows_gen(F.P_rows, 1, s) : natural_chunk(F.P_rows, s, w, lane);
        const unsigned j = (unsigned)lane + 64u * (unsigned)s;
        return j < (unsigned)F.n ? row_dot<true, false>(R.Prp, R.Pent, R.Pcol, (const double *)B.P, j) : 0.0;
    }
    CPG_DEV double atx(int s) const {
        if (QUMEM || rowprog) return QUMEM ? rows_gen(F.At_rows, 2, s) : natural_chunk(F.At_rows, s, w, lane);
        const unsigned j = (unsigned)lane + 64u * (unsigned)s;
        return j < (unsigned)F.n ? row_dot<false, true>(R.Ap, nullptr, R.Ai, (const double *)B.A, j) : 0.0;
    }
};

// (the streaming substitution executor, run_program_stream, lives in cpg_osqp_kernel.h: the shared-
// factor kernel uses it for programs that do not fit the LDS)

// Numeric LDL' of the instance's (permuted) KKT matrix through the dot-product schedule: chunk by
// chunk every lane (or group of 2^lg lanes, when a level has few destinations) accumulates
// sum_k L_ik d_k L_jk for its destination, subtracts it from the KKT
// entry and stores the pivot (and its reciprocal) or the unscaled column entry; when a level of the
// elimination tree is complete its columns are divided by their pivots.  `reg` is what the matrix
// carries on the (1,1) diagonal: sigma for the ADMM system, the adjoint's regularisation otherwise.
// The three index loads of a step do not depend on the factor, and the steps of a chunk are
// independent: CPG_LDL_BATCH steps are requested together (values of this batch, then the indices of
// the next one), accumulated in order.
#ifndef CPG_LDL_BATCH
#define CPG_LDL_BATCH 4
#endif
CPG_DEV void numeric_ldl(const DevRefactor &R, const InstBuf &B, double reg, int lane) {
    int level_start = 0;
#pragma nounroll
    for (int c = 0; c < R.fac_chunks; c++) {
        const int L = cpgw::read_first_lane(cpgw::gld(R.fac_ctab, 4u * (unsigned)c));
        const int last = cpgw::read_first_lane(cpgw::gld(R.fac_ctab, 4u * (unsigned)c + 1u));
        unsigned base = (unsigned)cpgw::read_first_lane(cpgw::gld(R.fac_ctab, 4u * (unsigned)c + 2u));
        const int lg = cpgw::read_first_lane(cpgw::gld(R.fac_ctab, 4u * (unsigned)c + 3u));
        const unsigned task = cpgw::gld(R.fac_task, (unsigned)c * 64u + (unsigned)lane);
        const unsigned lw = cpgw::gld(R.fac_len, (unsigned)c * 64u + (unsigned)lane);
        const int len = (int)(lw & 0xFFFFu), rlen = (int)(lw >> 16);     // addressing length | real terms of this lane
        double acc = 0.0;
        // NB steps per batch; the index triples of batch k + 1 are requested before the values of
        // batch k are consumed, so a batch exposes one memory round trip (its value gathers)
        constexpr int NB = CPG_LDL_BATCH;
        unsigned ia[NB], ib[NB], ik[NB];
        auto load_indices = [&](int s0, unsigned (&xa)[NB], unsigned (&xb)[NB], unsigned (&xk)[NB]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NB; t++) {
                const bool on = s0 + t < len;
                const unsigned e = on ? base + (unsigned)lane : 0u;
                base += cpgw::popc64(cpgw::ballot(on));
                xa[t] = cpgw::gld(R.fac_a, e); xb[t] = cpgw::gld(R.fac_b, e); xk[t] = cpgw::gld(R.fac_k, e);
            }
        };
        load_indices(0, ia, ib, ik);
#pragma nounroll
        for (int s = 0; s < L; s += NB) {
            double la[NB], lb[NB], dk[NB];
#pragma unroll
            for (int t = 0; t < NB; t++) {
                la[t] = cpgw::gld((const double *)B.Lx, ia[t]);
                lb[t] = cpgw::gld((const double *)B.Lx, ib[t]);
                dk[t] = cpgw::gld((const double *)B.Dg, ik[t]);
            }
            unsigned na[NB], nb[NB], nk[NB];
            if (s + NB < L) load_indices(s + NB, na, nb, nk);   // uniform
#pragma unroll
            for (int t = 0; t < NB; t++)
                if (s + t < rlen) acc = fma(la[t] * dk[t], lb[t], acc);
#pragma unroll
            for (int t = 0; t < NB; t++) { ia[t] = na[t]; ib[t] = nb[t]; ik[t] = nk[t]; }
        }
        acc = cpgw::group_sum_first_dyn(acc, lg);               // dot products split over 2^lg lanes (refactor_plan._pack_tasks)
        if (task != 0xFFFFFFFFu) {
            const int kind = cpgw::gld(R.ksrc_kind, task);
            const unsigned idx = (unsigned)cpgw::gld(R.ksrc_idx, task);
            const bool piv = task >= (unsigned)R.nnzL;
            double kv = 0.0;
            if (kind == CPG_K_P) kv = cpgw::gld((const double *)B.P, idx) + (piv ? reg : 0.0);
            else if (kind == CPG_K_A) kv = cpgw::gld((const double *)B.A, idx);
            else if (kind == CPG_K_SIGMA) kv = reg;
            else if (kind == CPG_K_RHO) kv = -cpgw::gld((const double *)B.rinv, idx);
            const double v = kv - acc;
            if (piv) { cpgw::gst(B.Dg, task - (unsigned)R.nnzL, v); cpgw::gst(B.Dginv, task - (unsigned)R.nnzL, 1.0 / v); }
            else cpgw::gst(B.Lx, task, v);
        }
        if (last) {   // level complete: divide the new columns by their pivots
            cpgw::mem_order();
#pragma nounroll
            for (int c2 = level_start; c2 <= c; c2++) {
                const unsigned t2 = cpgw::gld(R.fac_task, (unsigned)c2 * 64u + (unsigned)lane);
                if (t2 < (unsigned)R.nnzL)
                    cpgw::gst(B.Lx, t2, cpgw::gld((const double *)B.Lx, t2) * cpgw::gld((const double *)B.Dginv, (unsigned)cpgw::gld(R.Lcol, t2)));
            }
            cpgw::mem_order();
            level_start = c + 1;
        }
    }
}
// Coefficients of the substitution program from the factor (1, -L_ij or 1 / d_j per entry).  MFORM: B.Lx holds the
// undivided entries of numeric_ldl_m (l_ij = M_ij / d_j), B.Dginv the reciprocal pivots.
template <bool MFORM = false, bool GENS = false>
CPG_DEV void substitution_values(const DevRefactor &R, const InstBuf &B, int lane) {
    // (GENS: program-entry order for the generated streaming executor, else the streaming executor's own layout)
    const int *kinds = GENS ? R.gs_kind : R.sol_kind, *idxs = GENS ? R.gs_idx : R.sol_idx;
    const unsigned cnt = GENS ? (unsigned)R.gs_nnz : (unsigned)R.sol_nnz;
    for (unsigned e = (unsigned)lane; e < cnt; e += 64u) {
        const int kind = cpgw::gld(kinds, e);
        const unsigned idx = (unsigned)cpgw::gld(idxs, e);
        double v = 0.0;
        if (kind == 1) v = 1.0;
        else if (kind == 2) v = MFORM ? -(cpgw::gld((const double *)B.Lx, idx) * cpgw::gld((const double *)B.Dginv, (unsigned)cpgw::gld(R.Lcol, idx)))
                                      : -cpgw::gld((const double *)B.Lx, idx);
        else if (kind == 3) v = cpgw::gld((const double *)B.Dginv, idx);
        cpgw::gst(B.sv, e, v);
    }
    cpgw::mem_order();
}

// Numeric LDL' of shared-matrix mode: the same dot-product schedule as numeric_ldl, in the form that needs ONE
// dependent step per level of the elimination tree instead of two -- the entries of a column are kept UNDIVIDED
// (M_ij = l_ij d_j = K_ij - sum_k M_ik M_jk / d_k), so they do not wait for their column's pivot; what is stored
// per pivot is 1 / d_j.  The factor (Ml [nnzL], Dil [N]) lives in the wavefront's LDS slice (LDS: the generated
// instance kernel) or in its global buffer; the schedule's index tables (shared, L2), the chunk headers and the
// KKT values -- family constants in this mode except the -1 / rho_vec of the (2,2) diagonal -- do not depend on the
// factor and are requested one batch / one chunk ahead, so a level costs one round trip of its operands.  The
// instances of a shared-matrix batch differ in rho only: this is the whole per-instance cost of a rho change
// (config 2: most of the 14 ms the per-instance phase spent outside its iterations were the ~ 5 dependent
// global-memory round trips per chunk of the generic version).
// KCONST: the KKT values are family constants (shared-matrix mode); otherwise they come from the instance's own
// P / A (per (chunk, lane) source tables fac_kind_cl / fac_idx_cl) with `reg` on the (1,1) diagonal.
template <bool LDS, bool KCONST = true>
CPG_DEV void numeric_ldl_m(const DevRefactor &R, double *Ml, double *Dil, const double *rinv, int lane,
                           const double *Pv = nullptr, const double *Av = nullptr, double reg = 0.0) {
    constexpr int NB = CPG_LDL_BATCH;
    // Software pipeline over the chunks: everything a chunk needs that does not depend on the factor -- its header
    // (scalar loads), the lane's destination, term count and KKT constant (per (chunk, lane) tables), the index
    // triples of its first batch -- is requested while the chunk(s) in front of it run; a chunk of the (mostly one- or
    // two-step) schedule then waits for its operands only.
    struct Hdr { int L, last, lg; unsigned base; };
    struct LaneTab { unsigned task, lw; double kc; int krow; int kind; };
    struct Idx { unsigned a[NB], b[NB], k[NB]; unsigned base; };
    const unsigned *ctab = (const unsigned *)R.fac_ctab;
    const int nch = R.fac_chunks;
    if (nch <= 0) return;
    auto hdr = [&](int c) __attribute__((always_inline)) {
        const unsigned o = 4u * (unsigned)(c < nch ? c : nch - 1);
        return Hdr{(int)cpgw::sld(ctab, o), (int)cpgw::sld(ctab, o + 1u), (int)cpgw::sld(ctab, o + 3u), cpgw::sld(ctab, o + 2u)};
    };
    auto lanetab = [&](int c) __attribute__((always_inline)) {
        const unsigned e = (unsigned)(c < nch ? c : nch - 1) * 64u + (unsigned)lane;
        if (KCONST) return LaneTab{cpgw::gld(R.fac_task, e), cpgw::gld(R.fac_len, e), cpgw::gld(R.fac_kc_cl, e), cpgw::gld(R.fac_krow_cl, e), 0};
        return LaneTab{cpgw::gld(R.fac_task, e), cpgw::gld(R.fac_len, e), 0.0, cpgw::gld(R.fac_idx_cl, e), cpgw::gld(R.fac_kind_cl, e)};
    };
    auto indices = [&](unsigned base, int len, int s0) __attribute__((always_inline)) {
        Idx x;
#pragma unroll
        for (int t = 0; t < NB; t++) {
            const bool on = s0 + t < len;
            const unsigned e = on ? base + (unsigned)lane : 0u;
            base += cpgw::popc64(cpgw::ballot(on));
            x.a[t] = cpgw::gld(R.fac_a, e); x.b[t] = cpgw::gld(R.fac_b, e); x.k[t] = cpgw::gld(R.fac_k, e);
        }
        x.base = base;
        return x;
    };
    Hdr h0 = hdr(0), h1 = hdr(1);
    LaneTab t0 = lanetab(0), t1 = lanetab(1);
    Idx i0 = indices(h0.base, (int)(t0.lw & 0xFFFFu), 0);
#pragma nounroll
    for (int c = 0; c < nch; c++) {
        const Hdr h2 = hdr(c + 2);
        const LaneTab t2 = lanetab(c + 2);
        const Idx i1 = indices(h1.base, (int)(t1.lw & 0xFFFFu), 0);          // first batch of the next chunk
        const bool has = t0.task != 0xFFFFFFFFu;
        double kr = 0.0, kc = t0.kc;     // KKT value of the destination = kc - kr (kr: the instance's 1 / rho_vec where it enters)
        if (KCONST) { if (has && t0.krow >= 0) kr = cpgw::gld(rinv, (unsigned)t0.krow); }
        else if (has) {
            const bool piv = t0.task >= (unsigned)R.nnzL;
            if (t0.kind == CPG_K_P) kc = cpgw::gld(Pv, (unsigned)t0.krow) + (piv ? reg : 0.0);
            else if (t0.kind == CPG_K_A) kc = cpgw::gld(Av, (unsigned)t0.krow);
            else if (t0.kind == CPG_K_SIGMA) kc = reg;
            else if (t0.kind == CPG_K_RHO) kr = cpgw::gld(rinv, (unsigned)t0.krow);
        }
        const int L = cpgw::read_first_lane(h0.L), len = (int)(t0.lw & 0xFFFFu), rlen = (int)(t0.lw >> 16);
        double acc = 0.0;
        Idx cur = i0;
#pragma nounroll
        for (int s = 0; s < L; s += NB) {
            double la[NB], lb[NB], dk[NB];
#pragma unroll
            for (int t = 0; t < NB; t++) { la[t] = cpgw::gld((const double *)Ml, cur.a[t]); lb[t] = cpgw::gld((const double *)Ml, cur.b[t]); dk[t] = cpgw::gld((const double *)Dil, cur.k[t]); }
            Idx nxt = cur;
            if (s + NB < L) nxt = indices(cur.base, len, s + NB);   // uniform
#pragma unroll
            for (int t = 0; t < NB; t++)
                if (s + t < rlen) acc = fma(la[t] * dk[t], lb[t], acc);
            cur = nxt;
        }
        acc = cpgw::group_sum_first_dyn(acc, cpgw::read_first_lane(h0.lg));
        if (has) {
            const double v = (kc - kr) - acc;
            if (t0.task >= (unsigned)R.nnzL) cpgw::gst(Dil, t0.task - (unsigned)R.nnzL, 1.0 / v);
            else cpgw::gst(Ml, t0.task, v);
        }
        if (cpgw::read_first_lane(h0.last)) { if (LDS) cpgw::lds_order(); else cpgw::mem_order(); }     // level complete: the next one reads what this one stored
        h0 = h1; h1 = h2; t0 = t1; t1 = t2; i0 = i1;
    }
    if (LDS) cpgw::lds_order(); else cpgw::mem_order();
}

#ifdef CPG_GENI_HEADER
// the instance's coefficients of the generated executor, from its factor in LDS (one gather per step and lane, once
// per factorisation: the ADMM loop then reads none): -l_ij = -M_ij / d_j, 1 / d_i, or 1
CPG_DEV void load_instance_coefficients(const DevRefactor &R, const double *Ml, const double *Dil, double (&cf)[CPG_GENI_NREGS], int lane) {
    // (per coefficient REGISTER: narrow steps share one, each on its own lanes -- codegen.pack_step_registers)
    // (opaque lane: the table addresses below do not depend on the instance, and left alone the compiler computes all
    // 2 x NREGS of them once, outside the instance loop, keeps them alive through it -- i.e. in scratch -- and reloads
    // each one right in front of its table read: two dependent memory round trips per read)
    const unsigned ln = (unsigned)cpgw::opaque(lane);
#pragma unroll
    for (int t = 0; t < CPG_GENI_NREGS; t++) {
        const unsigned code = cpgw::gld(R.gi_src, (unsigned)t * 64u + ln);
        // (opaque: the column is read here, next to the source word -- left alone the compiler sinks the read into the
        // L-entry branch below, a second dependent L2 round trip per register)
        const unsigned col = (unsigned)cpgw::opaque((int)cpgw::gld(R.gi_lcol, (unsigned)t * 64u + ln));
        const unsigned kind = code >> 28, idx = code & 0x0FFFFFFFu;
        // (both reads UNCONDITIONAL, the kinds selected afterwards: as `if (kind == 2) v = -(Ml[idx] * Dil[col]); else if ...` every
        // register was a branch tree with an LDS round trip of its own)
        const double a = kind == 3u ? Dil[idx] : Ml[kind == 2u ? idx : 0u], b = Dil[kind == 2u ? col : 0u];
        cf[t] = kind == 2u ? -(a * b) : (kind == 3u ? a : (kind == 1u ? 1.0 : 0.0));
    }
}
#endif

// GENI: the substitution runs through the generated instance executor (register-resident coefficients) instead of
// the streaming one; the launch code selects it for shared-matrix handles of a family library (R.gi_ok)
// SHARED: shared-matrix mode as a compile-time fact (its own kernel instantiation): the per-instance-matrix kernel
// then carries none of that mode's code -- this body is inlined into kernels whose register allocation reacts to
// everything in it (config 3 lost 9 % when the shared-mode paths were merely present, profiles/r3_final1_*).
// CRLDS: the entry words of the streaming executor (operand offset | output row | segment mask per entry: shared by all
// instances, 4 bytes next to every 8-byte coefficient) get a block-shared LDS copy, one workgroup of eight wavefronts
// per CU.  With the per-instance coefficient streams of the resident wavefronts flowing through it the L2 does not keep
// that table: config 3 refetched it for every instance and iteration (a third of its FETCH_SIZE).
// a value every lane holds alike (rho: loaded per lane from the instance's workspace, or computed from wave-wide norms), moved
// to scalar registers: the step sizes derived from it then cost no VGPRs in the ADMM loop of the generated instance kernel
CPG_DEV double uniform_double(double v) {
    int w2[2];
    __builtin_memcpy(w2, &v, 8);
    w2[0] = cpgw::read_first_lane(w2[0]); w2[1] = cpgw::read_first_lane(w2[1]);
    __builtin_memcpy(&v, w2, 8);
    return v;
}

template <int NSX, int NSZ, bool GENI = false, bool SHARED = GENI, bool CRLDS = false>
CPG_DEV void osqp_refactor_body(const DevFamily &F0, const DevRefactor &R, const DevSettings &S,
                                const DevBatch &Bt, double *lds, int wave_global) {
    const int lane = cpgw::lane_id();
    // generated instance kernel: the family's dimensions are compile-time facts of its library (cpg_hip_set_refactor checks
    // them with the fingerprint) -- the bound tests of full 64-row slots, the class of the equality slots and every
    // address computation on n / m fold (the resident kernel's lesson)
#if defined(CPG_GENI_HEADER) && defined(CPG_GENI_N)
    const unsigned n = GENI ? (unsigned)CPG_GENI_N : (unsigned)F0.n, m = GENI ? (unsigned)CPG_GENI_M : (unsigned)F0.m, N = n + m;
    const unsigned n_eq = GENI ? (unsigned)CPG_GENI_NEQ : (unsigned)R.n_eq;
#else
    const unsigned n = (unsigned)F0.n, m = (unsigned)F0.m, N = n + m, n_eq = (unsigned)R.n_eq;
#endif
    const unsigned *cr_tab = R.sol_cr;
    if (CRLDS) {
        unsigned *lc = (unsigned *)lds;
        const unsigned ncr = (unsigned)R.sol_nnz;
        for (unsigned t = cpgw::thread_in_block(); t < ncr; t += cpgw::block_threads()) lc[t] = cpgw::gld(R.sol_cr, t);
        cpgw::block_sync();
        cr_tab = lc;
        lds += (ncr + 1u) / 2u;
    }
#ifdef CPG_GENI_HEADER
    const int ldw = GENI ? CPG_GENI_NSLOTS + CPG_GEN_EXTRA_SLOTS : (CPG_GENS_ACTIVE(GENI, SHARED, CRLDS) ? R.sol_slots + CPG_GEN_EXTRA_SLOTS : R.sol_slots);
    // block-shared copies of the executor's offset / output-slot tables in front of the work vectors
    constexpr unsigned gi_ncols = ((CPG_GENI_NSTEPS + 3u) / 4u) * 256u, gi_nrows = ((CPG_GENI_NCHUNKS + 3u) / 4u) * 256u;
    const unsigned short *gi_lc = nullptr, *gi_lr = nullptr;
    if (GENI) {
        unsigned short *lc = (unsigned short *)lds, *lr = lc + gi_ncols;
        for (unsigned t = cpgw::thread_in_block(); t < gi_ncols; t += cpgw::block_threads()) lc[t] = cpgw::gld(R.gi_cols, t);
        for (unsigned t = cpgw::thread_in_block(); t < gi_nrows; t += cpgw::block_threads()) lr[t] = cpgw::gld(R.gi_rows, t);
        cpgw::block_sync();
        gi_lc = lc; gi_lr = lr;
        lds += (gi_ncols + gi_nrows) / 4u;
    }
#else
    const int ldw = CPG_GENS_ACTIVE(GENI, SHARED, CRLDS) ? R.sol_slots + CPG_GEN_EXTRA_SLOTS : R.sol_slots;
#endif
    // per wavefront: the work vector, and with the generated executor the instance's q and u behind it; the same
    // slice holds the factor (M [nnzL] | 1 / d [N]) while numeric_ldl_lds runs -- nothing in it is live then
    size_t per_wave = (size_t)ldw + (GENI ? (size_t)(N + (N & 1u)) : 0u);
#ifdef CPG_GENI_FAC_NSTEPS
    const size_t fac_doubles = (size_t)R.nnzL + N + 1u;       // (+ the zero slot idle lanes of the generated factorisation read)
#else
    const size_t fac_doubles = (size_t)R.nnzL + N;
#endif
    if (GENI && per_wave < fac_doubles) per_wave = fac_doubles + (fac_doubles & 1u);
    double *w = lds + (size_t)cpgw::wave_in_block() * per_wave;
    double *qs = w + ldw, *us = qs + n;
    constexpr bool shared = SHARED || GENI;
#ifdef CPG_GENS_HEADER
    constexpr bool GENS = !shared && !CRLDS;          // (this library's per-instance-matrix program is the generated one: cpg_hip_set_refactor checked)
#else
    constexpr bool GENS = false;
#endif
    InstBuf B = carve(Bt.scratch + (size_t)wave_global * (size_t)R.buf_doubles, F0, R);
    const double rho_fr = CPG_RHO_MIN, ri_fr = 1.0 / rho_fr;
    const size_t state_len = (size_t)n + 2u * (size_t)m + 1u;
    unsigned short fpx[NSX], fpz[NSZ];
#pragma unroll
    for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; fpx[s] = i < n ? cpgw::gld(R.sol_fpos, i) : 0; }
#pragma unroll
    for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; fpz[s] = i < m ? cpgw::gld(R.sol_fpos, n + i) : 0; }

    const unsigned n_work = Bt.list_count ? cpgw::sld(Bt.list_count, 0u) : 0u;   // written by the kernel in front of this one
    if (shared) {     // the family's matrices serve every instance
        B.P = const_cast<double *>(R.Ps); B.A = const_cast<double *>(R.As); B.Ar = const_cast<double *>(R.Ars);
        B.D = const_cast<double *>(R.Ds); B.Dinv = const_cast<double *>(R.Dinvs);
        B.E = const_cast<double *>(R.Es); B.Einv = const_cast<double *>(R.Einvs);
    }

    for (;;) {
        unsigned ig = 0;
        if (lane == 0) ig = cpgw::atomic_next(Bt.counter);
        ig = (unsigned)cpgw::read_first_lane((int)ig);
        long long b = (long long)ig;
        if (Bt.list) {
            if (ig >= n_work) break;
            b = (long long)cpgw::read_first_lane(cpgw::gld(Bt.list, ig));
        } else if (b >= Bt.B) break;
        const double *theta = Bt.theta + (size_t)b * R.np_var;
        // rho of this instance's WORKSPACE (its factor): the family's, or what a sequential caller's previous solve
        // left (rho adaptation is workspace state in OSQP), or what the kernel in front handed over; clamped as
        // osqp_solve does.  rho of the SETTINGS (what compute_rho_estimate scales): every cpg_solve of the
        // reference resets it to the library default (osqp_set_default_settings, solvers/osqp.py:100-101) without
        // touching the workspace; adapt_rho (osqp_update_rho) then sets both.
        const double *state_in = (Bt.state_in && (S.warm_starting || Bt.resume)) ? Bt.state_in + (size_t)b * state_len : nullptr;
        double rho = Bt.state_in ? cpgw::gld(Bt.state_in + (size_t)b * state_len, n + 2u * m) : F0.rho;     // (osqp_cold_start resets the iterates only)
        rho = cpgw::dmin2(cpgw::dmax2(rho, CPG_RHO_MIN), CPG_RHO_MAX);
        if (GENI) rho = uniform_double(rho);
        double rho_stg = F0.rho;        // (a resumed instance: see where `iter` starts)
        double rho_eq = 1e3 * rho, rho_in = rho, ri_eq = 1.0 / rho_eq, ri_in = 1.0 / rho_in;

        // (experiments, generated instance kernel only, debug_stage 20: 100 MHz time stamps of the instance's stages replace
        // its first primal results -- scripts/gpu_probe_instance.py)
        const bool probe = GENI && __builtin_expect(S.debug_stage == 20, 0);
        unsigned long long ts[8];
        int n_ts = 0;
#define CPG_INST_PROBE() do { if (GENI && probe && n_ts < 8) ts[n_ts++] = cpgw::clock100(); } while (0)
        CPG_INST_PROBE();
        // ---- 1. canonicalise (unscaled; scaled in shared-matrix mode): P, A values, q, u, d
        if (!shared) {
            for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzA; k += 64u) cpgw::gst(B.A, k, csr_row(R.map_A, k, theta, cpgw::gld(R.A_base, k)));
            for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzP; k += 64u) cpgw::gst(B.P, k, csr_row(R.map_P, k, theta, cpgw::gld(R.P_base, k)));
        }
        for (unsigned i = (unsigned)lane; i < n; i += 64u) cpgw::gst(B.q, i, csr_row(R.map_q, i, theta, cpgw::gld(R.q_base, i)));
        for (unsigned i = (unsigned)lane; i < m; i += 64u) cpgw::gst(B.u, i, csr_row(R.map_u, i, theta, cpgw::gld(R.u_base, i)));
        const double dconst = csr_row(R.map_d, 0, theta, R.d_base);
        cpgw::mem_order();
        double cs = shared ? R.cs : 1.0;
        if (!shared) {
#ifdef CPG_REFACTOR_ROW_COPY
            refresh_row_copy(R, B, lane);            // unscaled A in row order: the row walks of the equilibration sweeps
#endif

            // ---- 2. Ruiz equilibration from scratch (D in w[0..n), E in w[n..N), cumulative form)
            for (unsigned i = (unsigned)lane; i < N; i += 64u) w[i] = 1.0;
            cpgw::lds_order();
#pragma nounroll
            for (int it = 0; it < R.scaling_iters; it++) {
                double dn[NSX], en[NSZ];
#pragma unroll
                for (int s = 0; s < NSX; s++) {
                    const unsigned j = (unsigned)lane + 64u * (unsigned)s;
                    double acc = 0.0;
                    if (j < n) {
                        const double dj = w[j];
                        for_row_entries<true>(R.Prp, R.Pent, R.Pcol, (const double *)B.P, j,
                                              [&](double v, unsigned c) { acc = cpgw::dmax2(acc, fabs(cs * dj * v * w[c])); });
                        for_row_entries<false>(R.Ap, nullptr, R.Ai, (const double *)B.A, j,
                                               [&](double v, unsigned c) { acc = cpgw::dmax2(acc, fabs(w[n + c] * v * dj)); });
                    }
                    dn[s] = acc;
                }
#pragma unroll
                for (int s = 0; s < NSZ; s++) {
                    const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                    double acc = 0.0;
                    if (i < m) {
                        const double ei = w[n + i];
#ifdef CPG_REFACTOR_ROW_COPY
                        for_row_entries<false, CPG_ROW_COPY_BATCH>(R.Arp, nullptr, R.Acol, (const double *)B.Ar, i,
#else
                        for_row_entries<true>(R.Arp, R.Aent, R.Acol, (const double *)B.A, i,
#endif
                                              [&](double v, unsigned c) { acc = cpgw::dmax2(acc, fabs(ei * v * w[c])); });
                    }
                    en[s] = acc;
                }
                cpgw::lds_order();
#pragma unroll
                for (int s = 0; s < NSX; s++) { const unsigned j = (unsigned)lane + 64u * (unsigned)s; if (j < n) w[j] = w[j] * (1.0 / sqrt(lim_scaling(dn[s]))); }
#pragma unroll
                for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < m) w[n + i] = w[n + i] * (1.0 / sqrt(lim_scaling(en[s]))); }
                cpgw::lds_order();
                // cost scaling: mean column norm of the scaled P against ||q||_inf
                double psum = 0.0, qn = 0.0;
#pragma unroll
                for (int s = 0; s < NSX; s++) {
                    const unsigned j = (unsigned)lane + 64u * (unsigned)s;
                    if (j < n) {
                        const double dj = w[j];
                        double acc = 0.0;
                        for_row_entries<true>(R.Prp, R.Pent, R.Pcol, (const double *)B.P, j,
                                              [&](double v, unsigned c) { acc = cpgw::dmax2(acc, fabs(cs * dj * v * w[c])); });
                        psum += acc;
                        qn = cpgw::dmax2(qn, fabs(cs * dj * cpgw::gld(R.q_setup, j)));   // update_mat runs before update_vec
                    }
                }
                psum = cpgw::wave_sum(psum);
                qn = lim_scaling(cpgw::wave_max_nonneg(qn));
                const double cm = n ? psum / (double)n : 0.0;
                cs = cs * (1.0 / lim_scaling(cpgw::dmax2(cm, qn)));
            }
            // ---- 3. scaled data, row classes, step sizes
            for (unsigned j = (unsigned)lane; j < n; j += 64u) {
                const double dj = w[j];
                cpgw::gst(B.D, j, dj); cpgw::gst(B.Dinv, j, 1.0 / dj);
                cpgw::gst(B.q, j, cs * dj * cpgw::gld((const double *)B.q, j));
                unsigned a = (unsigned)cpgw::gld(R.Ap, j), e = (unsigned)cpgw::gld(R.Ap, j + 1u);
                for (unsigned k = a; k < e; k++) cpgw::gst(B.A, k, w[n + (unsigned)cpgw::gld(R.Ai, k)] * cpgw::gld((const double *)B.A, k) * dj);
                a = (unsigned)cpgw::gld(R.Pp, j); e = (unsigned)cpgw::gld(R.Pp, j + 1u);
                for (unsigned k = a; k < e; k++) cpgw::gst(B.P, k, cs * w[(unsigned)cpgw::gld(R.Pi, k)] * cpgw::gld((const double *)B.P, k) * dj);
            }
#ifdef CPG_REFACTOR_ROW_COPY
            refresh_row_copy(R, B, lane);            // scaled A in row order: the termination test's A x
#endif
        }
        if (__builtin_expect(S.debug_stage == 1, 0)) { if (lane == 0) { Bt.status[b] = 11; Bt.iter[b] = 0; } continue; }     // (timing experiments: canonicalised)
        signed char ct[NSZ];
        const int lane_rc = GENI ? cpgw::opaque(lane) : lane;      // (generated instance kernel: addresses local to this block, see InstCtx)
#pragma unroll
        for (int s = 0; s < NSZ; s++) {
            const unsigned i = (unsigned)lane_rc + 64u * (unsigned)s;         // addresses
            const unsigned ic = (unsigned)lane + 64u * (unsigned)s;           // tests (range known: they fold on full slots when m, n_eq are constants)
            ct[s] = 0;
            if (ic < m) {
                double uu = cpgw::gld((const double *)B.u, i);      // shared-matrix mode: already E u
                if (!shared) {
                    const double ei = w[n + i];
                    uu = ei * uu;
                    cpgw::gst(B.E, i, ei); cpgw::gst(B.Einv, i, 1.0 / ei); cpgw::gst(B.u, i, uu);
                }
                // equality rows (l = u) are the first n_eq rows of the canonical form
                ct[s] = ic < n_eq ? 1 : (uu > CPG_INFTY * CPG_MIN_SCALING ? -1 : 0);
                cpgw::gst(B.rinv, i, ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr));
            }
        }
        cpgw::lds_order();
        cpgw::mem_order();
        // generated instance kernel: the step sizes of the slots whose class is not a compile-time fact (a slot below n_eq is
        // all equalities) as per-lane values, set here and after an adapt_rho -- the ADMM loop then selects nothing
        double rvv[NSZ], riv[NSZ];
        auto set_steps = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < NSZ; s++) {
                rvv[s] = ct[s] == 1 ? rho_eq : (ct[s] == 0 ? rho_in : rho_fr);
                riv[s] = ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr);
            }
        };
        if (GENI) set_steps();

        // ---- 4. numeric LDL' through the dot-product schedule, 5. coefficients of the substitution program
        auto factor_generic = [&]() __attribute__((always_inline)) {
            if (shared) {      // shared-matrix mode: one dependent step per level, KKT values from the family's table
                cpgw::mem_order();
                numeric_ldl_m<false>(R, B.Lx, B.Dginv, (const double *)B.rinv, lane);
                substitution_values<true>(R, B, lane);
            } else {
                // (per-instance matrices keep the two-step form: the one-step form with its KKT values gathered from the
                // instance's P / A measured SLOWER on this kernel -- config 3 161 -> 147 k/s, all parameters 200 -> 190 k/s,
                // profiles/r3_s13_*: its extra live values push the ADMM loop's allocation over the edge)
                numeric_ldl(R, B, F0.sigma, lane);
                substitution_values<false, GENS>(R, B, lane);
            }
        };
#ifdef CPG_GENI_HEADER
        double cf[CPG_GENI_NREGS];           // (dead, hence free, in the streaming instantiation)
        auto factor_in_lds = [&]() __attribute__((always_inline)) {
            cpgw::mem_order();                // B.rinv
#ifdef CPG_GENI_FAC_NSTEPS
            if (lane == 0) w[CPG_GENI_FAC_ZERO] = 0.0;
            cpgw::lds_order();
            numeric_ldl_gen(R.gf_tri, R.gf_dk, R.fac_kc_cl, (const double *)B.rinv, w, lane);
#else
            numeric_ldl_m<true>(R, w, w + R.nnzL, (const double *)B.rinv, lane);
#endif
            load_instance_coefficients(R, w, w + R.nnzL, cf, lane);
            cpgw::lds_order();
            // the slice goes back to its ADMM use: idle lanes of a step gather the zero slot, idle lanes of a chunk
            // store to the dummy slots behind the program's own (everything starts finite); q and u of the instance
            for (unsigned t = (unsigned)lane; t < (unsigned)ldw; t += 64u) w[t] = 0.0;
            for (unsigned i = (unsigned)lane; i < n; i += 64u) qs[i] = cpgw::gld((const double *)B.q, i);
            for (unsigned i = (unsigned)lane; i < m; i += 64u) us[i] = cpgw::gld((const double *)B.u, i);
            cpgw::lds_order();
        };
        if (__builtin_expect(S.debug_stage == 2, 0)) { if (lane == 0) { Bt.status[b] = 11; Bt.iter[b] = 0; } continue; }     // (row classes, rho_vec)
        CPG_INST_PROBE();          // 1: canonicalised, row classes
        if (GENI) factor_in_lds();
        else
#endif
        factor_generic();
        if (__builtin_expect(S.debug_stage == 3, 0)) { if (lane == 0) { Bt.status[b] = 11; Bt.iter[b] = 0; } continue; }     // (factorised, coefficients loaded)
        CPG_INST_PROBE();          // 2: factorised, coefficients in registers

#ifdef CPG_GENS_HEADER
        if (GENS) {    // the zero slot idle lanes of the generated executor gather (its dummy store targets sit next to it)
            if (lane == 0) w[CPG_GENS_NSLOTS + CPG_GEN_DUMMY_SLOTS] = 0.0;
            cpgw::lds_order();
        }
#endif
        // ---- 6. ADMM from cold start with the instance's own factor
        DevFamily F = F0;
        F.D = B.D; F.Dinv = B.Dinv; F.E = B.E; F.Einv = B.Einv; F.c = cs; F.cinv = 1.0 / cs;
        // (F.n / F.m stay run-time values: as literals they shrink the termination test and the retrieval, but the allocation of
        // the whole kernel shifts and the ADMM loop reloads six values from scratch per iteration -- tried, measured by
        // scripts/isa_hot_loops.py, dropped)
        StreamProg ST;
        ST.stab = R.sol_stab; ST.cr = cr_tab; ST.vals = B.sv;
        ST.n_pairs = R.sol_pairs; ST.dummy = (unsigned)R.sol_nnz / 2u - 1u;
        typedef InstCtx<NSX, NSZ, GENI> CtxT;
        double qr[NSX], ur[NSZ];
#pragma unroll
        for (int s = 0; s < NSX; s++) {
            const unsigned i = (unsigned)lane + 64u * (unsigned)s;
            qr[s] = i < n ? cpgw::gld((const double *)B.q, i) : 0.0;
        }
#pragma unroll
        for (int s = 0; s < NSZ; s++) {
            const unsigned i = (unsigned)lane + 64u * (unsigned)s;
            ur[s] = i < m ? cpgw::gld((const double *)B.u, i) : 0.0;
        }
        const CtxT cx{F, R, B, w, lane, qr, ur, qs, us, shared && F.A_rows.n_chunks > 0};
        double x[NSX], z[NSZ], y[NSZ];
#pragma unroll
        for (int s = 0; s < NSX; s++) x[s] = 0.0;
#pragma unroll
        for (int s = 0; s < NSZ; s++) { z[s] = 0.0; y[s] = 0.0; }
        if (state_in) load_state<NSX, NSZ>(F, state_in, x, z, y, lane);
        CheckOut o;
        o.prim_res = 0; o.dual_res = 0; o.obj = 0; o.status = 11;
        int iter = Bt.resume ? cpgw::read_first_lane(cpgw::gld((const int *)Bt.iter, (unsigned)b)) : 0;
        // handed over after an adapt_rho of this solve (osqp_update_rho wrote settings and workspace); at iteration 0
        // the workspace merely arrived with another rho than the family's
        if (iter > 0) rho_stg = rho;
        double dxr[NSX], dyr[NSZ];      // steps of the last checked iteration (infeasibility tests)
#pragma unroll
        for (int s = 0; s < NSX; s++) dxr[s] = 0.0;
#pragma unroll
        for (int s = 0; s < NSZ; s++) dyr[s] = 0.0;
        // One ADMM iteration; `chk` also keeps the steps delta x / delta y for the termination check.
        auto admm_iteration = [&](const bool chk) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < n) w[i] = F.sigma * x[s] - cx.q(s, i); }
#pragma unroll
            for (int s = 0; s < NSZ; s++) {
                const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                const bool ceq = GENI && 64u * (unsigned)(s + 1) <= n_eq;       // (a compile-time fact per unrolled slot)
                const double ri = !GENI ? (ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr)) : (ceq ? ri_eq : riv[s]);
                if (i < m) w[n + i] = z[s] - ri * y[s];
            }
            cpgw::lds_order();
#ifdef CPG_GENI_HEADER
            if (GENI) run_program_inst(cf, gi_lc, gi_lr, w, lane);
            else
#endif
#ifdef CPG_GENS_HEADER
            if (GENS) run_program_gens((const double *)B.sv, R.gs_cols, R.gs_rows, w, lane);
            else
#endif
            run_program_stream(ST, w, lane);
#pragma unroll
            for (int s = 0; s < NSX; s++) {
                const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                if (chk) dxr[s] = 0.0;           // (defined on every lane: nothing of the old steps stays live across the iterations)
                if (i < n) {
                    const double xn = F.alpha * w[GENI ? i : (unsigned)fpx[s]] + (1.0 - F.alpha) * x[s];     // (generated program: solved in place)
                    if (chk) dxr[s] = xn - x[s];
                    x[s] = xn;
                }
            }
#pragma unroll
            for (int s = 0; s < NSZ; s++) {
                const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                if (chk) dyr[s] = 0.0;
                if (i < m) {
                    const bool ceq = GENI && 64u * (unsigned)(s + 1) <= n_eq;
                    const double rv = !GENI ? (ct[s] == 1 ? rho_eq : (ct[s] == 0 ? rho_in : rho_fr)) : (ceq ? rho_eq : rvv[s]);
                    const double ri = !GENI ? (ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr)) : (ceq ? ri_eq : riv[s]);
                    const double zp = z[s], yp = y[s];
                    const double zt = (zp - ri * yp) + ri * w[GENI ? n + i : (unsigned)fpz[s]];
                    const double zr = F.alpha * zt + (1.0 - F.alpha) * zp;
                    const double uu = cx.u(s, i);
                    const double zc = cpgw::dmin2(zr + ri * yp, uu);        // (computed on every lane: a select, not a branch, in a slot of mixed classes)
                    const double zn = ceq ? uu : (ct[s] == 1 ? uu : zc);
                    const double dyv = rv * (zr - zn);
                    z[s] = zn; y[s] = yp + dyv;
                    if (chk) dyr[s] = dyv;
                }
            }
            cpgw::lds_order();
        };
        const int chk_int = S.check_termination, ad_int = S.adaptive_rho ? S.adaptive_rho_interval : 0;
        CPG_INST_PROBE();          // 3: workspace loaded
        // The iterations between two events (termination check, rho adaptation, max_iter) run in their own
        // inner loop: the check (row products, norms, infeasibility tests) needs many registers, and with
        // its code inside the hot loop the iterates were spilled and reloaded in every iteration.
#pragma nounroll
        while (o.status == 11) {
            if (iter < S.max_iter) {
                int next_ev = S.max_iter;
                if (chk_int > 0) { const int c = (iter / chk_int + 1) * chk_int; if (c < next_ev) next_ev = c; }
                if (ad_int > 0) { const int c = (iter / ad_int + 1) * ad_int; if (c < next_ev) next_ev = c; }
#pragma nounroll
                for (; iter < next_ev - 1; iter++) admm_iteration(false);
                iter++;
                admm_iteration(true);
            }
            CPG_INST_PROBE();      // 4 (6, ...): iterations up to the next event
            const bool can_check = chk_int > 0 && iter > 0 && iter % chk_int == 0;
            const bool adapt = ad_int > 0 && iter > 0 && iter % ad_int == 0;
            const bool last = iter >= S.max_iter;
            ScaledNorms sn;
            bool have_info = false;
            if (can_check) {
                o = check<NSX, NSZ, CtxT, RegDelta<NSX>, RegDelta<NSZ>>(F, cx, ct, S, x, z, y, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr}, InfeasVerdict{false, false}, w, lane, false, &sn);
                have_info = true;
                CPG_INST_PROBE();  // 5 (7, ...): termination test
                if (o.status != 11) break;
            }
            if (adapt) {
                // adapt_rho (OSQP paper sec. 5.2): rho <- rho sqrt(normalised primal / dual residual); a new
                // factorisation only when it changed by more than adaptive_rho_tolerance
                if (!have_info) (void)check<NSX, NSZ, CtxT, RegDelta<NSX>, RegDelta<NSZ>>(F, cx, ct, S, x, z, y, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr}, InfeasVerdict{false, false}, w, lane, false, &sn);
                const double rn = GENI ? uniform_double(rho_estimate(sn, rho_stg)) : rho_estimate(sn, rho_stg);
                if (rn > rho_stg * S.adaptive_rho_tolerance || rn < rho_stg / S.adaptive_rho_tolerance) {
                    rho = rn; rho_stg = rn; rho_eq = 1e3 * rho; rho_in = rho; ri_eq = 1.0 / rho_eq; ri_in = 1.0 / rho_in;
                    if (GENI) set_steps();
#pragma unroll
                    for (int s = 0; s < NSZ; s++) {
                        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                        if (i < m) cpgw::gst(B.rinv, i, ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr));
                    }
                    cpgw::mem_order();
#ifdef CPG_GENI_HEADER
                    if (GENI) factor_in_lds();
                    else
#endif
                    factor_generic();
                }
            }
            if (last) {
                if (!can_check) o = check<NSX, NSZ, CtxT, RegDelta<NSX>, RegDelta<NSZ>>(F, cx, ct, S, x, z, y, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr}, InfeasVerdict{false, false}, w, lane, false);
                if (o.status == 11) o = check<NSX, NSZ, CtxT, RegDelta<NSX>, RegDelta<NSZ>>(F, cx, ct, S, x, z, y, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr}, InfeasVerdict{false, false}, w, lane, true);
                if (o.status == 11) o.status = 7;
            }
        }
        finalize<NSX, NSZ, GENI>(F, Bt, x, z, y, dconst, b, w, lane, iter, o, rho);     // (per-call lane id in the generated instance kernel, see InstCtx)
        if (GENI && probe) {
            CPG_INST_PROBE();      // last: results written
            if (lane == 0) for (int k = 0; k < n_ts && k < F0.n_prim; k++) Bt.prim[(size_t)b * F0.n_prim + k] = (double)(ts[k] - ts[0]);
            if (lane == 0 && n_ts < F0.n_prim) Bt.prim[(size_t)b * F0.n_prim + n_ts] = -1.0;
        }
    }
}

}  // namespace cpg

// =================================================================================================
// QP adjoint for a batch: cpg_update_d<var> + cpg_gradient() + cpg_osqp_gradient()
// (cvxpygen/writer.py:222-312, cvxpygen/templates/cpg_osqp_grad_compute.c.jinja2:432-531).
//
// The reference keeps one factor of K = [[P + 1e-6 I, A'], [A, -1e-6 I]] and switches constraint
// rows on / off with rank-one up / down-dates (template :157-324) -- a sequential-reuse trick.  An
// inactive row leaves exactly "row and column zero, -1 on the diagonal" (template :196-214), so for
// independent instances the masked matrix is factored directly: its pattern is the pattern of the
// OSQP KKT matrix, hence the LDL' schedule and substitution program of the refactorisation path are
// reused with other value sources (1e-6 instead of sigma, -1e-6 / -1 instead of -1/rho, A masked).
namespace cpg {

struct DevGradient {
    const int *Pcolidx, *Acolidx;    // column of every stored entry of P / A
    // transposed canonical maps: for user-parameter column c the contributions (kind, index, coef)
    const int *tptr, *tkind, *tidx;
    const double *tcoef;
    int NP;
};
struct DevGradBatch {
    long long B;
    const double *theta, *sol_x, *sol_y, *dx;    // dx: upstream gradient scattered to canonical x [B][n]
    double *dtheta;
    unsigned *counter;
    double *scratch;
};
#define CPG_G_Q 0
#define CPG_G_L 1
#define CPG_G_U 2
#define CPG_G_P 3
#define CPG_G_A 4

template <int NSX, int NSZ>
CPG_DEV void osqp_gradient_body(const DevFamily &F, const DevRefactor &R, const DevGradient &Gd,
                                const DevGradBatch &Bt, double *lds, int wave_global) {
    const int lane = cpgw::lane_id();
    const unsigned n = (unsigned)F.n, m = (unsigned)F.m, N = n + m;
    const int ldw = R.sol_slots;
    // LDS per wavefront: w | r (N) | x (n) | y (m) | dx (n) | active flags (m)
    const size_t per_wave = (size_t)ldw + N + n + m + n + m;
    double *w = lds + (size_t)cpgw::wave_in_block() * per_wave;
    double *rr = w + ldw, *xs = rr + N, *ys = xs + n, *dxs = ys + m, *act = dxs + n;
    const InstBuf B = carve(Bt.scratch + (size_t)wave_global * (size_t)R.buf_doubles, F, R);
    const double eps = 1e-6;
    StreamProg ST;
    ST.stab = R.sol_stab; ST.cr = R.sol_cr; ST.vals = B.sv;
    ST.n_pairs = R.sol_pairs; ST.dummy = (unsigned)R.sol_nnz / 2u - 1u;

    for (;;) {
        unsigned ig = 0;
        if (lane == 0) ig = cpgw::atomic_next(Bt.counter);
        ig = (unsigned)cpgw::read_first_lane((int)ig);
        if ((long long)ig >= Bt.B) break;
        const size_t b = (size_t)ig;
        const double *theta = Bt.theta + b * (size_t)R.np_var;
        // ---- solution, active set, upstream gradient (cpg_update_d<var>: scatter to canonical x)
        for (unsigned i = (unsigned)lane; i < n; i += 64u) { xs[i] = cpgw::gld(Bt.sol_x + b * n, i); dxs[i] = cpgw::gld(Bt.dx + b * n, i); }
        for (unsigned i = (unsigned)lane; i < m; i += 64u) {
            const double yi = cpgw::gld(Bt.sol_y + b * m, i);
            ys[i] = yi;
            const double a = yi < -1e-12 ? -1.0 : (yi > 1e-12 ? 1.0 : 0.0);
            act[i] = a;
            cpgw::gst(B.rinv, i, a != 0.0 ? eps : 1.0);
        }
        cpgw::lds_order();
        // ---- canonical P, A of this instance (unscaled); inactive rows of A removed
        for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzA; k += 64u) {
            const double v = csr_row(R.map_A, k, theta, cpgw::gld(R.A_base, k));
            cpgw::gst(B.A, k, act[(unsigned)cpgw::gld(R.Ai, k)] != 0.0 ? v : 0.0);
        }
        for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzP; k += 64u) cpgw::gst(B.P, k, csr_row(R.map_P, k, theta, cpgw::gld(R.P_base, k)));
        cpgw::lds_order();
        cpgw::mem_order();
        // ---- numeric LDL' of the masked, regularised KKT matrix
        numeric_ldl(R, B, eps, lane);
        substitution_values(R, B, lane);
        // ---- r = K^-1 [dx; 0] and three sweeps of refinement against the exact masked KKT matrix
#pragma nounroll
        for (int sweep = 0; sweep < 4; sweep++) {
            if (sweep == 0) {
                for (unsigned i = (unsigned)lane; i < n; i += 64u) w[i] = dxs[i];
                for (unsigned i = (unsigned)lane; i < m; i += 64u) w[n + i] = 0.0;
            } else {
                // delta = rhs - K_true r, inactive rows / columns skipped (template :460-476)
                for (unsigned j = (unsigned)lane; j < n; j += 64u) {
                    double d = dxs[j];
                    for_row_entries<true>(R.Prp, R.Pent, R.Pcol, (const double *)B.P, j, [&](double v, unsigned c) { d -= v * rr[c]; });
                    for_row_entries<false>(R.Ap, nullptr, R.Ai, (const double *)B.A, j, [&](double v, unsigned c) { d -= v * rr[n + c]; });
                    w[j] = d;
                }
                for (unsigned i = (unsigned)lane; i < m; i += 64u) {
                    double d = 0.0;
                    if (act[i] != 0.0) {
                        for_row_entries<true>(R.Arp, R.Aent, R.Acol, (const double *)B.A, i, [&](double v, unsigned c) { d -= v * rr[c]; });
                    }
                    w[n + i] = d;
                }
            }
            cpgw::lds_order();
            run_program_stream(ST, w, lane);
            for (unsigned i = (unsigned)lane; i < N; i += 64u) {
                const double v = w[(unsigned)cpgw::gld(R.sol_fpos, i)];
                rr[i] = sweep == 0 ? v : rr[i] + v;
            }
            cpgw::lds_order();
        }
        // ---- canonical gradients and pull-back through the transposed maps (writer.py:268-311)
        double *out = Bt.dtheta + b * (size_t)Gd.NP;
        for (unsigned c = (unsigned)lane; c < (unsigned)Gd.NP; c += 64u) {
            const unsigned a0 = (unsigned)cpgw::gld(Gd.tptr, c), e0 = (unsigned)cpgw::gld(Gd.tptr, c + 1u);
            double acc = 0.0;
            for (unsigned t = a0; t < e0; t++) {
                const int kind = cpgw::gld(Gd.tkind, t);
                const unsigned idx = (unsigned)cpgw::gld(Gd.tidx, t);
                double d = 0.0;
                if (kind == CPG_G_Q) d = -rr[idx];
                else if (kind == CPG_G_L) d = act[idx] == -1.0 ? rr[n + idx] : 0.0;
                else if (kind == CPG_G_U) d = act[idx] == 1.0 ? rr[n + idx] : 0.0;
                else if (kind == CPG_G_P) {
                    const unsigned i = (unsigned)cpgw::gld(R.Pi, idx), j = (unsigned)cpgw::gld(Gd.Pcolidx, idx);
                    d = -0.5 * (rr[i] * xs[j] + xs[i] * rr[j]);
                } else {
                    const unsigned i = (unsigned)cpgw::gld(R.Ai, idx), j = (unsigned)cpgw::gld(Gd.Acolidx, idx);
                    d = act[i] != 0.0 ? -(rr[n + i] * xs[j] + ys[i] * rr[j]) : 0.0;
                }
                acc = fma(cpgw::gld(Gd.tcoef, t), d, acc);
            }
            cpgw::gst(out, c, acc);
        }
        cpgw::lds_order();
    }
}

}  // namespace cpg
