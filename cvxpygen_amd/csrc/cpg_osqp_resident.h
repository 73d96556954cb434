// RESIDENT per-instance factor kernel: families whose parameters enter P or A, i.e. per instance
//   cpg_canonicalize_P / _A / _q / _u   (cvxpygen/utils.py:279-294)
//   osqp_update_data_mat                (cvxpygen/solvers/osqp.py:20-33; third-party OSQP: overwrite the values,
//                                        Ruiz-equilibrate from scratch, numeric LDL' on the fixed pattern)
//   osqp_update_data_vec, osqp_solve, cpg_retrieve_*   (solvers/osqp.py:39-62, utils.py:950-985)
// with everything an ADMM iteration touches kept ON THE CU.  The streaming kernel of cpg_osqp_refactor.h reads the
// instance's substitution coefficients from HBM in every iteration (portfolio family: 87 KB per instance and
// iteration, 483 GB per launch of 20 000 instances, 49 % of the HBM peak -- profiles/r3_final7_pmc_config3.txt).
// Here:
//   * one wavefront per instance, ONE wavefront per SIMD: the unified register file of gfx950 gives it 512 registers
//     (256 VGPRs + 256 AGPRs), enough for the ~8 500 coefficients of the instance's substitution program
//     (2 registers per step and lane after sharing, codegen.pack_step_registers) next to the iterates;
//   * the program has MERGED levels (resident_plan.py): the diagonal blocks of merged groups are inverted
//     numerically by this kernel after every factorisation, 47 -> 13 dependent phases per KKT solve;
//   * set-up in the wavefront's LDS slice: theta staged once, canonicalisation through coalesced (ELL) maps,
//     equilibration sweeps entry-parallel with LDS max-atomics (the streaming kernel walks rows through three
//     dependent global-memory round trips per entry batch), numeric LDL' + inverses through one flat, prefetched
//     dot-product stream;
//   * the products of the termination test stream per-instance copies of A and P in program order (coalesced, no
//     index chase) through run_program_stream.
// HBM traffic per instance: theta in, results out, ~100 KB of set-up state once, ~60 KB per termination test.
#pragma once

#include "cpg_osqp_refactor.h"
#ifdef CPG_GENR_HEADER
// straight-line executor of the family's MERGED per-instance substitution program, coefficients in registers
// (cvxpygen_amd/codegen.py::resident_header)
#include CPG_GENR_HEADER
#endif

namespace cpg {

struct alignas(16) ResEntry { unsigned x, y, z, w; };

struct DevStreamTab {             // a ragged program in the layout of run_program_stream + where its values come from
    const unsigned *stab, *cr;
    const int *src;               // [n_entries] entry of the instance's matrix behind entry e (-1: padding)
    int n_pairs, n_entries;
    unsigned dummy;
};
struct DevEll {                   // out[k] = base[k] + sum_j coef[j * rows + k] * theta[idx[j * rows + k]]
    int J, rows;
    const int *idx;
    const double *coef;
};
#define CPG_RES_FAC_DEPTH 8       // steps of the factorisation stream in flight
struct DevResident {
    int ok;
    int nnzX, fac_len, fac_steps;          // fac = [M (nnzL) | 1/d (N) | X (nnzX) | 1.0 | 0.0]; fac_steps: multiple of the depth
    const unsigned *f_ctl;                 // [fac_steps + depth][2]: entry base | lanes << 24 ; flags (1 first step of a chunk,
                                           // 2 last, 4 level complete) | reduction stages << 4
    const ResEntry *f_ent;                   // per entry: position of l_ik (M), of 1/d_k, of l_jk / X_kj, destination of the
                                           // lane's task (first step of a chunk only; pivot flag in bit 31; none = ~0)
    unsigned f_dummy;                      // entry whose factors are the 0.0 slot
    const unsigned *k_src;                 // [nnzL + N] KKT source of a destination: kind << 28 | index; bit 31: a pivot without
                                           // dot product (store the reciprocal right away)
    const unsigned *g_src;                 // [NREGS][64] coefficient source of (register, lane): kind << 28 | index
    const unsigned short *g_lcol;          // ... and the column of the L entry behind a kind-2 coefficient
    const unsigned short *g_cols, *g_rows; // operand offsets / output slots of the generated executor (LDS tables)
    DevEll eP, eA, eq, eu;
    const unsigned *entA, *entP;           // row | column << 16 of every stored entry
    DevStreamTab pA, pP, pAt;              // A x, P x, A' y on the work vector [x | y | .. | A x | P x | A' y]
    int out_ax, out_px, out_aty;           // first slot of the products' results
    int slice_doubles;                     // LDS doubles per wavefront
    long long buf_doubles;                 // per-wavefront buffer in global memory
};

#ifdef CPG_GENR_HEADER
struct ResBuf { double *A, *P, *D, *Dinv, *E, *Einv, *q, *u, *rinv, *cA, *cP, *cAt; };
CPG_DEV ResBuf res_carve(double *b, const DevFamily &F, const DevRefactor &R, const DevResident &Rs) {
    ResBuf o;
    const size_t n = (size_t)F.n, m = (size_t)F.m;
    o.A = b; b += R.nnzA; o.P = b; b += R.nnzP;
    o.D = b; b += n; o.Dinv = b; b += n; o.E = b; b += m; o.Einv = b; b += m;
    o.q = b; b += n; o.u = b; b += m; o.rinv = b; b += m;
    o.cA = b; b += Rs.pA.n_entries; o.cP = b; b += Rs.pP.n_entries; o.cAt = b; b += Rs.pAt.n_entries;
    return o;
}

CPG_DEV double ell_row(const DevEll &E, unsigned k, const double *th, double v) {
    for (int j = 0; j < E.J; j++) {
        const unsigned e = (unsigned)j * (unsigned)E.rows + k;
        v = fma(cpgw::gld(E.coef, e), th[(unsigned)cpgw::gld(E.idx, e)], v);
    }
    return v;
}

// Numeric LDL' of the instance's KKT matrix in the M-form of numeric_ldl_m (undivided column entries, reciprocal
// pivots), followed by the inverses X = L_GG^-1 of the merged groups' diagonal blocks -- one flat stream of dot-product
// steps over `fac` (LDS), whose destinations were preloaded with their KKT values (zeros for X).  An entry is the
// index triple of one term plus, on a chunk's first step, the destination of the lane's task; the entries of the next
// CPG_RES_FAC_DEPTH steps are on their way while a step is consumed: the tables do not depend on the factor, and with
// one wavefront per SIMD nobody else hides a memory round trip.
CPG_DEV void resident_factor(const DevResident &Rs, double *fac, int lane) {
    constexpr int DP = CPG_RES_FAC_DEPTH;
    ResEntry ring[DP];
    unsigned c0[DP], c1[DP], n0[DP], n1[DP];
    auto request = [&](unsigned ctl) __attribute__((always_inline)) {
        const unsigned cnt = (ctl >> 24) & 0x7Fu;
        const unsigned e = (unsigned)lane < cnt ? (ctl & 0xFFFFFFu) + (unsigned)lane : Rs.f_dummy;
        return cpgw::gld(Rs.f_ent, e);
    };
#pragma unroll
    for (int u = 0; u < DP; u++) {
        c0[u] = cpgw::sld(Rs.f_ctl, 2u * (unsigned)u); c1[u] = cpgw::sld(Rs.f_ctl, 2u * (unsigned)u + 1u);
        n0[u] = cpgw::sld(Rs.f_ctl, 2u * (unsigned)(DP + u)); n1[u] = cpgw::sld(Rs.f_ctl, 2u * (unsigned)(DP + u) + 1u);
    }
#pragma unroll
    for (int u = 0; u < DP; u++) ring[u] = request(c0[u]);
    double acc = 0.0;
    unsigned dest = 0xFFFFFFFFu;
#pragma nounroll
    for (int t0 = 0; t0 < Rs.fac_steps; t0 += DP) {
#pragma unroll
        for (int u = 0; u < DP; u++) {
            const ResEntry en = ring[u];
            const unsigned fl = c1[u];
            ring[u] = request(n0[u]);                       // step t0 + u + DP
            if (fl & 1u) { dest = en.w; acc = 0.0; }
            const double la = fac[en.x], dk = fac[en.y], lb = fac[en.z];
            acc = fma(la * dk, lb, acc);
            if (fl & 2u) {
                const double r = cpgw::group_sum_first_dyn(acc, (int)(fl >> 4));
                if (dest != 0xFFFFFFFFu) {
                    const unsigned d = dest & 0x7FFFFFFFu;
                    const double v = fac[d] - r;
                    fac[d] = (dest & 0x80000000u) ? 1.0 / v : v;
                }
                if (fl & 4u) cpgw::lds_order();             // level complete: the next one reads what this one stored
            }
            c0[u] = n0[u]; c1[u] = n1[u];
            n0[u] = cpgw::sld(Rs.f_ctl, 2u * (unsigned)(t0 + 2 * DP + u)); n1[u] = cpgw::sld(Rs.f_ctl, 2u * (unsigned)(t0 + 2 * DP + u) + 1u);
        }
    }
    cpgw::lds_order();
}

// the instance's coefficients of the generated executor from `fac`: -l_ij = -M_ij / d_j, 1 / d_i, X_ij or 1
CPG_DEV void resident_coefficients(const DevRefactor &R, const DevResident &Rs, const double *fac, double (&cf)[CPG_GENR_NREGS], int lane) {
    const unsigned ln = (unsigned)cpgw::opaque(lane);            // (see load_instance_coefficients: addresses local to this block)
    const unsigned nnzL = (unsigned)R.nnzL, X0 = (unsigned)(Rs.fac_len - 2 - Rs.nnzX);
#pragma unroll
    for (int t = 0; t < CPG_GENR_NREGS; t++) {
        const unsigned code = cpgw::gld(Rs.g_src, (unsigned)t * 64u + ln);
        const unsigned col = (unsigned)cpgw::opaque((int)cpgw::gld(Rs.g_lcol, (unsigned)t * 64u + ln));
        const unsigned kind = code >> 28, idx = code & 0x0FFFFFFFu;
        double v = 0.0;
        if (kind == 1u) v = 1.0;
        else if (kind == 2u) v = -(fac[idx] * fac[nnzL + col]);
        else if (kind == 3u) v = fac[nnzL + idx];
        else if (kind == 4u) v = fac[X0 + idx];
        cf[t] = v;
    }
}

// q / u of the instance in the wavefront's LDS slice; the three products of the termination test through their row
// programs on per-instance copies of the scaled matrices in program order
template <int NSX, int NSZ>
struct ResidentCtx {
    static constexpr bool kTestsFirst = false;      // OSQP's own order: infeasibility tests inside check()
    static constexpr bool kOpaqueLane = true;
    const DevFamily &F;
    const DevResident &Rs;
    const ResBuf &B;
    double *w;
    const double *qm, *um;
    int lane;
    CPG_DEV double q(int, unsigned i) const { return qm[i]; }
    CPG_DEV double u(int, unsigned i) const { return um[i]; }
    CPG_DEV void run(const DevStreamTab &T, const double *vals) const {
        StreamProg ST;
        ST.stab = T.stab; ST.cr = T.cr; ST.vals = vals; ST.n_pairs = T.n_pairs; ST.dummy = T.dummy;
        run_program_stream(ST, w, lane);
    }
    CPG_DEV void products(int which) const {        // 1: A w[0..n)   2: P w[0..n)   4: A' w[n..n+m)
        if (which & 1) run(Rs.pA, B.cA);
        if (which & 2) run(Rs.pP, B.cP);
        if (which & 4) run(Rs.pAt, B.cAt);
        cpgw::lds_order();
    }
    CPG_DEV double ax(int s) const { const unsigned i = (unsigned)lane + 64u * (unsigned)s; return i < (unsigned)F.m ? w[(unsigned)Rs.out_ax + i] : 0.0; }
    CPG_DEV double px(int s) const { const unsigned i = (unsigned)lane + 64u * (unsigned)s; return i < (unsigned)F.n ? w[(unsigned)Rs.out_px + i] : 0.0; }
    CPG_DEV double atx(int s) const { const unsigned i = (unsigned)lane + 64u * (unsigned)s; return i < (unsigned)F.n ? w[(unsigned)Rs.out_aty + i] : 0.0; }
};

template <int NSX, int NSZ>
CPG_DEV void osqp_resident_body(const DevFamily &F0, const DevRefactor &R, const DevResident &Rs, const DevSettings &S,
                                const DevBatch &Bt, double *lds, int wave_global) {
    const int lane = cpgw::lane_id();
    const unsigned n = (unsigned)F0.n, m = (unsigned)F0.m, N = n + m;
    constexpr int ldw = CPG_GENR_NSLOTS + CPG_GEN_EXTRA_SLOTS;
    // block-shared copies of the executor's offset / output-slot tables in front of the wavefronts' slices
    constexpr unsigned t_ncols = ((CPG_GENR_NSTEPS + 3u) / 4u) * 256u, t_nrows = ((CPG_GENR_NCHUNKS + 3u) / 4u) * 256u;
    unsigned short *lc = (unsigned short *)lds, *lr = lc + t_ncols;
    for (unsigned t = cpgw::thread_in_block(); t < t_ncols; t += cpgw::block_threads()) lc[t] = cpgw::gld(Rs.g_cols, t);
    for (unsigned t = cpgw::thread_in_block(); t < t_nrows; t += cpgw::block_threads()) lr[t] = cpgw::gld(Rs.g_rows, t);
    cpgw::block_sync();
    lds += (t_ncols + t_nrows) / 4u;
    // The wavefront's slice, three lives:
    //   set-up    A (nnzA) | P (nnzP) | D (n) | E (m) | norms (max(n, m))      theta is staged where D starts
    //   factor    fac = M (nnzL) | 1/d (N) | X | 1.0 | 0.0
    //   ADMM      w (ldw) | q (n) | u (m) | A x (m) | P x (n) | A' y (n)
    double *sl = lds + (size_t)cpgw::wave_in_block() * (size_t)Rs.slice_doubles;
    double *w = sl, *qs = w + ldw, *us = qs + n;
    double *Al = sl, *Pl = Al + R.nnzA, *Dl = Pl + R.nnzP, *El = Dl + n;
    unsigned long long *nrm = (unsigned long long *)(El + m);
    const ResBuf B = res_carve(Bt.scratch + (size_t)wave_global * (size_t)Rs.buf_doubles, F0, R, Rs);
    const double rho_fr = CPG_RHO_MIN, ri_fr = 1.0 / rho_fr;
    const size_t state_len = (size_t)n + 2u * (size_t)m + 1u;
    const unsigned n_work = Bt.list_count ? cpgw::sld(Bt.list_count, 0u) : 0u;
    typedef ResidentCtx<NSX, NSZ> CtxT;

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
        // rho of the workspace / of the settings: see osqp_refactor_body
        const double *state_in = (Bt.state_in && (S.warm_starting || Bt.resume)) ? Bt.state_in + (size_t)b * state_len : nullptr;
        double rho = Bt.state_in ? cpgw::gld(Bt.state_in + (size_t)b * state_len, n + 2u * m) : F0.rho;
        rho = cpgw::dmin2(cpgw::dmax2(rho, CPG_RHO_MIN), CPG_RHO_MAX);
        double rho_stg = F0.rho;
        double rho_eq = 1e3 * rho, rho_in = rho, ri_eq = 1.0 / rho_eq, ri_in = 1.0 / rho_in;

        // ---- 1. theta -> LDS; canonicalise P, A (LDS), q, u (registers), d
        {
            double *th = Dl;
            for (unsigned t = (unsigned)lane; t < (unsigned)R.np_var; t += 64u) th[t] = cpgw::gld(theta, t);
            cpgw::lds_order();
            for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzA; k += 64u) Al[k] = ell_row(Rs.eA, k, th, cpgw::gld(R.A_base, k));
            for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzP; k += 64u) Pl[k] = ell_row(Rs.eP, k, th, cpgw::gld(R.P_base, k));
        }
        double qr[NSX], ur[NSZ];
        {
            const double *th = Dl;
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; qr[s] = i < n ? ell_row(Rs.eq, i, th, cpgw::gld(R.q_base, i)) : 0.0; }
#pragma unroll
            for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; ur[s] = i < m ? ell_row(Rs.eu, i, th, cpgw::gld(R.u_base, i)) : 0.0; }
        }
        const double dconst = csr_row(R.map_d, 0, theta, R.d_base);
        cpgw::lds_order();

        // ---- 2. Ruiz equilibration from scratch, cumulative form (D, E in LDS); entry-parallel sweeps: an entry's
        //         scaled magnitude goes to its column's / row's norm through an LDS max-atomic (non-negative doubles
        //         order like their bit patterns), the lane that owns a column / row then reads its norm
        for (unsigned i = (unsigned)lane; i < N; i += 64u) Dl[i] = 1.0;
        double cs = 1.0;
        cpgw::lds_order();
        auto p_norms = [&]() __attribute__((always_inline)) {       // column norms of c D P D (both triangles)
            for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzP; k += 64u) {
                const unsigned rc = cpgw::gld(Rs.entP, k), i = rc & 0xFFFFu, j = rc >> 16;
                const double p = Pl[k], di = Dl[i], dj = Dl[j];
                cpgw::lds_max_u64(nrm + j, fabs(cs * dj * p * di));
                if (i != j) cpgw::lds_max_u64(nrm + i, fabs(cs * di * p * dj));
            }
        };
#pragma nounroll
        for (int it = 0; it < R.scaling_iters; it++) {
            double dn[NSX], en[NSZ];
            for (unsigned i = (unsigned)lane; i < n; i += 64u) nrm[i] = 0ull;
            cpgw::lds_order();
            p_norms();
            for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzA; k += 64u) {
                const unsigned rc = cpgw::gld(Rs.entA, k), r = rc & 0xFFFFu, c = rc >> 16;
                cpgw::lds_max_u64(nrm + c, fabs(El[r] * Al[k] * Dl[c]));
            }
            cpgw::lds_order();
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned j = (unsigned)lane + 64u * (unsigned)s; dn[s] = j < n ? cpgw::u64_as_double(nrm[j]) : 0.0; }
            cpgw::lds_order();
            for (unsigned i = (unsigned)lane; i < m; i += 64u) nrm[i] = 0ull;
            cpgw::lds_order();
            for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzA; k += 64u) {
                const unsigned rc = cpgw::gld(Rs.entA, k), r = rc & 0xFFFFu, c = rc >> 16;
                cpgw::lds_max_u64(nrm + r, fabs(El[r] * Al[k] * Dl[c]));
            }
            cpgw::lds_order();
#pragma unroll
            for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; en[s] = i < m ? cpgw::u64_as_double(nrm[i]) : 0.0; }
            cpgw::lds_order();
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned j = (unsigned)lane + 64u * (unsigned)s; if (j < n) Dl[j] = Dl[j] * (1.0 / sqrt(lim_scaling(dn[s]))); }
#pragma unroll
            for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < m) El[i] = El[i] * (1.0 / sqrt(lim_scaling(en[s]))); }
            for (unsigned i = (unsigned)lane; i < n; i += 64u) nrm[i] = 0ull;
            cpgw::lds_order();
            // cost scaling: mean column norm of the scaled P against ||q||_inf of the workspace's q (update_mat runs
            // before update_vec, cvxpygen/solvers/osqp.py:20-59)
            p_norms();
            cpgw::lds_order();
            double psum = 0.0, qn = 0.0;
#pragma unroll
            for (int s = 0; s < NSX; s++) {
                const unsigned j = (unsigned)lane + 64u * (unsigned)s;
                if (j < n) {
                    psum += cpgw::u64_as_double(nrm[j]);
                    qn = cpgw::dmax2(qn, fabs(cs * Dl[j] * cpgw::gld(R.q_setup, j)));
                }
            }
            psum = cpgw::wave_sum(psum);
            qn = lim_scaling(cpgw::wave_max_nonneg(qn));
            const double cm = n ? psum / (double)n : 0.0;
            cs = cs * (1.0 / lim_scaling(cpgw::dmax2(cm, qn)));
            cpgw::lds_order();
        }
        // ---- 3. scaled data: matrices (LDS, then the wavefront's buffer: the factorisations read their KKT values
        //         there, the termination tests their program-order copies), scaling vectors, q, u, row classes
        for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzA; k += 64u) {
            const unsigned rc = cpgw::gld(Rs.entA, k), r = rc & 0xFFFFu, c = rc >> 16;
            const double v = El[r] * Al[k] * Dl[c];
            Al[k] = v; cpgw::gst(B.A, k, v);
        }
        for (unsigned k = (unsigned)lane; k < (unsigned)R.nnzP; k += 64u) {
            const unsigned rc = cpgw::gld(Rs.entP, k), i = rc & 0xFFFFu, j = rc >> 16;
            const double v = cs * Dl[i] * Pl[k] * Dl[j];
            Pl[k] = v; cpgw::gst(B.P, k, v);
        }
        signed char ct[NSZ];
#pragma unroll
        for (int s = 0; s < NSX; s++) {
            const unsigned j = (unsigned)lane + 64u * (unsigned)s;
            if (j < n) {
                const double dj = Dl[j];
                cpgw::gst(B.D, j, dj); cpgw::gst(B.Dinv, j, 1.0 / dj);
                qr[s] = cs * dj * qr[s];
                cpgw::gst(B.q, j, qr[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < NSZ; s++) {
            const unsigned i = (unsigned)lane + 64u * (unsigned)s;
            ct[s] = 0;
            if (i < m) {
                const double ei = El[i], uu = ei * ur[s];
                cpgw::gst(B.E, i, ei); cpgw::gst(B.Einv, i, 1.0 / ei); cpgw::gst(B.u, i, uu);
                ct[s] = i < (unsigned)R.n_eq ? 1 : (uu > CPG_INFTY * CPG_MIN_SCALING ? -1 : 0);
                cpgw::gst(B.rinv, i, ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr));
            }
        }
        cpgw::lds_order();
        auto copy_values = [&](const DevStreamTab &T, double *dst, const double *src) __attribute__((always_inline)) {
            for (unsigned e = (unsigned)lane; e < (unsigned)T.n_entries; e += 64u) {
                const int k = cpgw::gld(T.src, e);
                cpgw::gst(dst, e, k >= 0 ? src[(unsigned)k] : 0.0);
            }
        };
        copy_values(Rs.pA, B.cA, Al);
        copy_values(Rs.pAt, B.cAt, Al);
        copy_values(Rs.pP, B.cP, Pl);
        cpgw::lds_order();
        cpgw::mem_order();

        // ---- 4. numeric LDL' + inverses of the merged diagonal blocks in the slice, 5. coefficients -> registers
        double cf[CPG_GENR_NREGS];
        auto factorise = [&]() __attribute__((always_inline)) {
            const unsigned nd = (unsigned)R.nnzL + N;
            for (unsigned d = (unsigned)lane; d < nd; d += 64u) {
                const unsigned code = cpgw::gld(Rs.k_src, d), kind = (code >> 28) & 7u, idx = code & 0x0FFFFFFFu;
                double v = 0.0;
                if (kind == CPG_K_P) v = cpgw::gld((const double *)B.P, idx) + (d >= (unsigned)R.nnzL ? F0.sigma : 0.0);
                else if (kind == CPG_K_A) v = cpgw::gld((const double *)B.A, idx);
                else if (kind == CPG_K_SIGMA) v = F0.sigma;
                else if (kind == CPG_K_RHO) v = -cpgw::gld((const double *)B.rinv, idx);
                sl[d] = (code >> 31) ? 1.0 / v : v;
            }
            for (unsigned d = nd + (unsigned)lane; d < (unsigned)Rs.fac_len; d += 64u) sl[d] = d == (unsigned)Rs.fac_len - 2u ? 1.0 : 0.0;
            cpgw::lds_order();
            resident_factor(Rs, sl, lane);
            resident_coefficients(R, Rs, sl, cf, lane);
            cpgw::lds_order();
            // the slice goes back to its ADMM use (idle lanes of a step gather the zero slot, idle lanes of a chunk store
            // to the dummy slots: everything starts finite); q and u of the instance
            // (... and the results of the termination test's products: rows without an entry are never written)
            for (unsigned t = (unsigned)lane; t < (unsigned)ldw; t += 64u) w[t] = 0.0;
            for (unsigned t = (unsigned)ldw + N + (unsigned)lane; t < (unsigned)Rs.slice_doubles; t += 64u) w[t] = 0.0;
            for (unsigned i = (unsigned)lane; i < n; i += 64u) qs[i] = cpgw::gld((const double *)B.q, i);
            for (unsigned i = (unsigned)lane; i < m; i += 64u) us[i] = cpgw::gld((const double *)B.u, i);
            cpgw::lds_order();
        };
        factorise();

        // ---- 6. ADMM with the instance's own factor
        DevFamily F = F0;
        F.D = B.D; F.Dinv = B.Dinv; F.E = B.E; F.Einv = B.Einv; F.c = cs; F.cinv = 1.0 / cs;
        const CtxT cx{F, Rs, B, w, qs, us, lane};
        double x[NSX], z[NSZ], y[NSZ];
#pragma unroll
        for (int s = 0; s < NSX; s++) x[s] = 0.0;
#pragma unroll
        for (int s = 0; s < NSZ; s++) { z[s] = 0.0; y[s] = 0.0; }
        if (state_in) load_state<NSX, NSZ>(F, state_in, x, z, y, lane);
        CheckOut o;
        o.prim_res = 0; o.dual_res = 0; o.obj = 0; o.status = 11;
        int iter = Bt.resume ? cpgw::read_first_lane(cpgw::gld((const int *)Bt.iter, (unsigned)b)) : 0;
        if (iter > 0) rho_stg = rho;
        double dxr[NSX], dyr[NSZ];
#pragma unroll
        for (int s = 0; s < NSX; s++) dxr[s] = 0.0;
#pragma unroll
        for (int s = 0; s < NSZ; s++) dyr[s] = 0.0;
        auto admm_iteration = [&](const bool chk) __attribute__((always_inline)) {
            double qt[NSX];
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; qt[s] = i < n ? qs[i] : 0.0; }
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < n) w[i] = F.sigma * x[s] - qt[s]; }
#pragma unroll
            for (int s = 0; s < NSZ; s++) {
                const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                const double ri = ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr);
                if (i < m) w[n + i] = z[s] - ri * y[s];
            }
            cpgw::lds_order();
            run_program_res(cf, lc, lr, w, lane);
            double wt[NSZ], ut[NSZ];
#pragma unroll
            for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; wt[s] = i < m ? w[n + i] : 0.0; ut[s] = i < m ? us[i] : 0.0; }
#pragma unroll
            for (int s = 0; s < NSX; s++) {
                const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                if (i < n) {
                    const double xn = F.alpha * w[i] + (1.0 - F.alpha) * x[s];
                    if (chk) dxr[s] = xn - x[s];
                    x[s] = xn;
                }
            }
#pragma unroll
            for (int s = 0; s < NSZ; s++) {
                const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                if (i < m) {
                    const double rv = ct[s] == 1 ? rho_eq : (ct[s] == 0 ? rho_in : rho_fr);
                    const double ri = ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr);
                    const double zp = z[s], yp = y[s];
                    const double zt = (zp - ri * yp) + ri * wt[s];
                    const double zr = F.alpha * zt + (1.0 - F.alpha) * zp;
                    const double uu = ut[s];
                    const double zn = ct[s] == 1 ? uu : cpgw::dmin2(zr + ri * yp, uu);
                    const double dyv = rv * (zr - zn);
                    z[s] = zn; y[s] = yp + dyv;
                    if (chk) dyr[s] = dyv;
                }
            }
            cpgw::lds_order();
        };
        const int chk_int = S.check_termination, ad_int = S.adaptive_rho ? S.adaptive_rho_interval : 0;
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
            const bool can_check = chk_int > 0 && iter > 0 && iter % chk_int == 0;
            const bool adapt = ad_int > 0 && iter > 0 && iter % ad_int == 0;
            const bool last = iter >= S.max_iter;
            ScaledNorms sn;
            bool have_info = false;
            if (can_check) {
                o = check<NSX, NSZ, CtxT, RegDelta<NSX>, RegDelta<NSZ>>(F, cx, ct, S, x, z, y, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr}, InfeasVerdict{false, false}, w, lane, false, &sn);
                have_info = true;
                if (o.status != 11) break;
            }
            if (adapt) {
                if (!have_info) (void)check<NSX, NSZ, CtxT, RegDelta<NSX>, RegDelta<NSZ>>(F, cx, ct, S, x, z, y, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr}, InfeasVerdict{false, false}, w, lane, false, &sn);
                const double rn = rho_estimate(sn, rho_stg);
                if (rn > rho_stg * S.adaptive_rho_tolerance || rn < rho_stg / S.adaptive_rho_tolerance) {
                    rho = rn; rho_stg = rn; rho_eq = 1e3 * rho; rho_in = rho; ri_eq = 1.0 / rho_eq; ri_in = 1.0 / rho_in;
#pragma unroll
                    for (int s = 0; s < NSZ; s++) {
                        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                        if (i < m) cpgw::gst(B.rinv, i, ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr));
                    }
                    cpgw::mem_order();
                    factorise();
                }
            }
            if (last) {
                if (!can_check) o = check<NSX, NSZ, CtxT, RegDelta<NSX>, RegDelta<NSZ>>(F, cx, ct, S, x, z, y, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr}, InfeasVerdict{false, false}, w, lane, false);
                if (o.status == 11) o = check<NSX, NSZ, CtxT, RegDelta<NSX>, RegDelta<NSZ>>(F, cx, ct, S, x, z, y, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr}, InfeasVerdict{false, false}, w, lane, true);
                if (o.status == 11) o.status = 7;
            }
        }
        finalize<NSX, NSZ, true>(F, Bt, x, z, y, dconst, b, w, lane, iter, o, rho);
    }
}
#endif  // CPG_GENR_HEADER

}  // namespace cpg
