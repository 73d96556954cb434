// TEAM per-instance factor kernel: the path of cpg_osqp_resident.h --
//   cpg_canonicalize_P / _A / _q / _u   (cvxpygen/utils.py:279-294)
//   osqp_update_data_mat                (cvxpygen/solvers/osqp.py:20-33; third-party OSQP: overwrite the values,
//                                        Ruiz-equilibrate from scratch, numeric LDL' on the fixed pattern)
//   osqp_update_data_vec, osqp_solve, cpg_retrieve_*   (solvers/osqp.py:39-62, utils.py:950-985)
// per instance, everything an ADMM iteration touches on the CU -- for families whose merged substitution program does NOT
// fit one wavefront's 512 registers (MPC 12/4/10 with every parameter per instance: 401 coefficient register pairs;
// it ran on the streaming kernel of cpg_osqp_refactor.h, 484 dependent phases per iteration each waiting for its
// coefficients from HBM: 188 k instances/s, 364 GB of traffic per 20 000 instances).  Here:
//   * one WORKGROUP of W wavefronts (a team, W = 2 / 4 / 8) per instance; entry i of a vector lives on thread i % 64W;
//   * the merged program of resident_plan.py planned for the team (solve_program.pack_ragged(team=W)): the chunks of a
//     phase are spread over the wavefronts, one barrier per phase; a wavefront keeps the coefficients of ITS steps in
//     registers, and the operand offsets and output slots of its steps too -- no table in LDS or memory is read inside
//     the ADMM loop (codegen.emit_team_program);
//   * the products of merged groups with the inverses of their diagonal blocks write to slots of their own (no second
//     barrier between the gathers and the stores of an in-place phase);
//   * set-up (canonicalisation, ten equilibration sweeps), KKT assembly, coefficient extraction, the three products and
//     the norms of the termination test run on all 64 W threads; the numeric LDL' + block inverses -- a chain of levels
//     with one chunk each on this family -- on wavefront 0 (generated straight-line code);
//   * the stages are real calls as in the resident kernel; the instance's state between them lives in memory.
#pragma once

#include "cpg_osqp_resident.h"
#ifdef CPG_GENT_HEADER
#include CPG_GENT_HEADER

namespace cpg {

#define CPG_TEAM_W CPG_GENT_W
#define CPG_TEAM_T (64 * CPG_GENT_W)
// LDS of a team: [0, 8) broadcast words, [8, 8 + 2 * 8 W) two buffers of 8 partial results per wavefront, then the slice
#define CPG_TEAM_RED_OFF 8u
#define CPG_TEAM_SLICE_OFF (8u + 16u * (unsigned)CPG_GENT_W)

// Reduction over the team: every wavefront reduces its lanes, lane 0 leaves the partial result in LDS, after ONE barrier
// every thread combines the W partial results in wavefront order (the same value, bit for bit, on every thread: the
// team's control flow depends on these).  Two buffers alternate: a thread can only write buffer p again after the barrier
// of the reduction in between, which every thread reaches after its reads of buffer p.
struct TeamRed { unsigned par; };
template <int K, bool SUM>
CPG_DEV void team_reduce(double (&v)[K], TeamRed &tr, int lane, int wave) {
    static_assert(K <= 8, "at most 8 values per reduction");
    CPG_LDS double *buf = cpgw::lds_window3() + CPG_TEAM_RED_OFF + tr.par * (8u * (unsigned)CPG_TEAM_W);
#pragma unroll
    for (int k = 0; k < K; k++) {
        const double r = SUM ? cpgw::wave_sum(v[k]) : cpgw::wave_max_nonneg(v[k]);
        if (lane == 0) buf[(unsigned)wave * 8u + (unsigned)k] = r;
    }
    cpgw::block_sync();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double r = buf[k];
#pragma unroll
        for (int u = 1; u < CPG_TEAM_W; u++) r = SUM ? r + buf[(unsigned)u * 8u + (unsigned)k] : cpgw::dmax2(r, buf[(unsigned)u * 8u + (unsigned)k]);
        v[k] = r;
    }
    tr.par ^= 1u;
}
CPG_DEV bool team_any(bool p, TeamRed &tr, int lane, int wave) {
    double v[1] = {p ? 1.0 : 0.0};
    team_reduce<1, false>(v, tr, lane, wave);
    return v[0] != 0.0;
}

template <int NZ>
struct TeamSetupOut {
    double cs, dconst;
    unsigned free_rows;           // bit s: row tid + T s is free (infinite bound)
    signed char ct[NZ];           // row classes: 1 equality, 0 inequality, -1 free
};
template <int NX, int NZ>
struct TeamState { double x[NX], z[NZ], y[NZ], dx[NX], dy[NZ]; };

// ---- steps 1 - 3 of an instance: canonicalise, equilibrate, scaled data (resident_setup on 64 W threads)
template <int NX, int NZ>
CPG_DEV_NOINLINE void team_setup(const DevRefactor &R_, const DevResident &Rs_, const ResBuf &B_, const double *theta_v,
                                 double ri_eq, double ri_in, double ri_fr, TeamSetupOut<NZ> &out) {
    const int tid = (int)cpgw::thread_in_block(), lane = cpgw::lane_id(), wave = cpgw::read_first_lane(cpgw::wave_in_block());
    constexpr unsigned T = CPG_TEAM_T;
    const DevRefactor R = uniform_global_copy(R_); const DevResident Rs = uniform_global_copy(Rs_); const ResBuf B = uniform_global_copy(B_);
    const double *theta = (const double *)(((unsigned long long)(unsigned)cpgw::read_first_lane((int)((unsigned long long)theta_v >> 32)) << 32) |
                                           (unsigned)cpgw::read_first_lane((int)(unsigned long long)theta_v));
    theta = cpgw::as_global(theta);
    constexpr unsigned n = CPG_GENT_N, m = CPG_GENT_M, N = n + m, n_eq = CPG_GENT_NEQ;
    constexpr unsigned nnzA = CPG_GENT_NNZA, nnzP = CPG_GENT_NNZP;
    constexpr int KA = (int)((nnzA + T - 1) / T) > 0 ? (int)((nnzA + T - 1) / T) : 1, KP = (int)((nnzP + T - 1) / T) > 0 ? (int)((nnzP + T - 1) / T) : 1;
    TeamRed tr{0u};
    // (LDS pointers carry their address space in the type: ds_read / ds_write / ds_max instead of flat accesses)
    CPG_LDS double *sl = cpgw::pin_lds(cpgw::lds_window3() + CPG_TEAM_SLICE_OFF);
    // (the scaling vectors, the norms and theta use the space the scaled matrices take once D and E are dead: step 3 below writes
    // A, P behind a barrier, after every read of D and E)
    CPG_LDS double *Al = sl, *Pl = Al + nnzA, *Dl = sl, *El = Dl + n;
    CPG_LDS unsigned long long *nrm = (CPG_LDS unsigned long long *)(El + m);
    // ---- 1. theta -> LDS; canonicalise P, A, q, u into registers (entry k = tid + T t of a matrix, entry i = tid + T s of a vector)
    unsigned ea[KA], ep[KP];           // row | column << 16
    double av[KA], pv[KP];
    double qr[NX], ur[NZ];
    {
        CPG_LDS double *th = Dl;
        for (unsigned t0 = 0; t0 < (unsigned)R.np_var; t0 += 8u * T) {
            double tv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const unsigned t = t0 + T * (unsigned)u + (unsigned)tid; tv[u] = t < (unsigned)R.np_var ? cpgw::gld(theta, t) : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const unsigned t = t0 + T * (unsigned)u + (unsigned)tid; if (t < (unsigned)R.np_var) th[t] = tv[u]; }
        }
        cpgw::block_sync();
#pragma unroll
        for (int t = 0; t < KA; t++) { const unsigned k = (unsigned)tid + T * (unsigned)t; ea[t] = k < nnzA ? cpgw::gld(Rs.entA, k) : 0u; av[t] = k < nnzA ? cpgw::gld(R.A_base, k) : 0.0; }
#pragma unroll
        for (int t = 0; t < KP; t++) { const unsigned k = (unsigned)tid + T * (unsigned)t; ep[t] = k < nnzP ? cpgw::gld(Rs.entP, k) : 0u; pv[t] = k < nnzP ? cpgw::gld(R.P_base, k) : 0.0; }
#pragma unroll
        for (int s = 0; s < NX; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; qr[s] = i < n ? cpgw::gld(R.q_base, i) : 0.0; }
#pragma unroll
        for (int s = 0; s < NZ; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; ur[s] = i < m ? cpgw::gld(R.u_base, i) : 0.0; }
#pragma nounroll
        for (int j = 0; j < Rs.eA.J; j++) {
#pragma unroll
            for (int t = 0; t < KA; t++) {
                const unsigned k = (unsigned)tid + T * (unsigned)t, e = (unsigned)j * (unsigned)Rs.eA.rows + (k < nnzA ? k : 0u);
                av[t] = fma(cpgw::gld(Rs.eA.coef, e), th[(unsigned)cpgw::gld(Rs.eA.idx, e)], av[t]);
            }
        }
#pragma nounroll
        for (int j = 0; j < Rs.eP.J; j++) {
#pragma unroll
            for (int t = 0; t < KP; t++) {
                const unsigned k = (unsigned)tid + T * (unsigned)t, e = (unsigned)j * (unsigned)Rs.eP.rows + (k < nnzP ? k : 0u);
                pv[t] = fma(cpgw::gld(Rs.eP.coef, e), th[(unsigned)cpgw::gld(Rs.eP.idx, e)], pv[t]);
            }
        }
#pragma nounroll
        for (int j = 0; j < Rs.eq.J; j++) {
#pragma unroll
            for (int s = 0; s < NX; s++) {
                const unsigned i = (unsigned)tid + T * (unsigned)s, e = (unsigned)j * (unsigned)Rs.eq.rows + (i < n ? i : 0u);
                qr[s] = fma(cpgw::gld(Rs.eq.coef, e), th[(unsigned)cpgw::gld(Rs.eq.idx, e)], qr[s]);
            }
        }
#pragma nounroll
        for (int j = 0; j < Rs.eu.J; j++) {
#pragma unroll
            for (int s = 0; s < NZ; s++) {
                const unsigned i = (unsigned)tid + T * (unsigned)s, e = (unsigned)j * (unsigned)Rs.eu.rows + (i < m ? i : 0u);
                ur[s] = fma(cpgw::gld(Rs.eu.coef, e), th[(unsigned)cpgw::gld(Rs.eu.idx, e)], ur[s]);
            }
        }
    }
    const double dconst = csr_row(R.map_d, 0, theta, R.d_base);
    double qsu[NX];                        // q of the code-generation-time workspace (cost scaling)
#pragma unroll
    for (int s = 0; s < NX; s++) { const unsigned j = (unsigned)tid + T * (unsigned)s; qsu[s] = j < n ? cpgw::gld(R.q_setup, j) : 0.0; }
    cpgw::block_sync();                    // (theta is dead: D, E take its place)

    // ---- 2. Ruiz equilibration from scratch, cumulative form (D, E in LDS), entry-parallel sweeps with LDS max-atomics
    for (unsigned i = (unsigned)tid; i < N; i += T) Dl[i] = 1.0;
    double cs = 1.0;
    cpgw::block_sync();
    auto p_norms = [&]() __attribute__((always_inline)) {       // column norms of c D P D (both triangles)
#pragma unroll
        for (int t = 0; t < KP; t++) {
            const unsigned k = (unsigned)tid + T * (unsigned)t, i = ep[t] & 0xFFFFu, j = ep[t] >> 16;
            if (k < nnzP) {
                const double p = pv[t], di = Dl[i], dj = Dl[j];
                cpgw::lds_max_u64_l(nrm + j, fabs(cs * dj * p * di));
                if (i != j) cpgw::lds_max_u64_l(nrm + i, fabs(cs * di * p * dj));
            }
        }
    };
#pragma nounroll
    for (int it = 0; it < R.scaling_iters; it++) {
        double dn[NX], en[NZ];
        for (unsigned i = (unsigned)tid; i < n; i += T) nrm[i] = 0ull;
        cpgw::block_sync();
        p_norms();
        double mag[KA];
#pragma unroll
        for (int t = 0; t < KA; t++) { const unsigned r = ea[t] & 0xFFFFu, c = ea[t] >> 16; mag[t] = fabs(El[r] * av[t] * Dl[c]); }
#pragma unroll
        for (int t = 0; t < KA; t++) { const unsigned k = (unsigned)tid + T * (unsigned)t; if (k < nnzA) cpgw::lds_max_u64_l(nrm + (ea[t] >> 16), mag[t]); }
        cpgw::block_sync();
#pragma unroll
        for (int s = 0; s < NX; s++) { const unsigned j = (unsigned)tid + T * (unsigned)s; dn[s] = j < n ? cpgw::u64_as_double(nrm[j]) : 0.0; }
        cpgw::block_sync();
        for (unsigned i = (unsigned)tid; i < m; i += T) nrm[i] = 0ull;
        cpgw::block_sync();
#pragma unroll
        for (int t = 0; t < KA; t++) { const unsigned k = (unsigned)tid + T * (unsigned)t; if (k < nnzA) cpgw::lds_max_u64_l(nrm + (ea[t] & 0xFFFFu), mag[t]); }
        cpgw::block_sync();
#pragma unroll
        for (int s = 0; s < NZ; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; en[s] = i < m ? cpgw::u64_as_double(nrm[i]) : 0.0; }
        cpgw::block_sync();
#pragma unroll
        for (int s = 0; s < NX; s++) { const unsigned j = (unsigned)tid + T * (unsigned)s; if (j < n) Dl[j] = Dl[j] * (1.0 / sqrt(lim_scaling(dn[s]))); }
#pragma unroll
        for (int s = 0; s < NZ; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; if (i < m) El[i] = El[i] * (1.0 / sqrt(lim_scaling(en[s]))); }
        for (unsigned i = (unsigned)tid; i < n; i += T) nrm[i] = 0ull;
        cpgw::block_sync();
        // cost scaling: mean column norm of the scaled P against ||q||_inf of the workspace's q (update_mat runs
        // before update_vec, cvxpygen/solvers/osqp.py:20-59)
        p_norms();
        cpgw::block_sync();
        double psum[1] = {0.0}, qn[1] = {0.0};
#pragma unroll
        for (int s = 0; s < NX; s++) {
            const unsigned j = (unsigned)tid + T * (unsigned)s;
            if (j < n) {
                psum[0] += cpgw::u64_as_double(nrm[j]);
                qn[0] = cpgw::dmax2(qn[0], fabs(cs * Dl[j] * qsu[s]));
            }
        }
        team_reduce<1, true>(psum, tr, lane, wave);
        team_reduce<1, false>(qn, tr, lane, wave);
        const double qnl = lim_scaling(qn[0]);
        const double cm = n ? psum[0] / (double)n : 0.0;
        cs = cs * (1.0 / lim_scaling(cpgw::dmax2(cm, qnl)));
        cpgw::block_sync();
    }
    // ---- 3. scaled data: matrices (LDS for the copies below; the team's buffer: the factorisations read their KKT values
    //         there, the termination tests their program-order copies), scaling vectors, q, u, row classes
#pragma unroll
    for (int t = 0; t < KA; t++) { const unsigned r = ea[t] & 0xFFFFu, c = ea[t] >> 16; av[t] = El[r] * av[t] * Dl[c]; }
#pragma unroll
    for (int t = 0; t < KP; t++) { const unsigned i = ep[t] & 0xFFFFu, j = ep[t] >> 16; pv[t] = cs * Dl[i] * pv[t] * Dl[j]; }
    unsigned free_rows = 0u;
#pragma unroll
    for (int s = 0; s < NX; s++) {
        const unsigned j = (unsigned)tid + T * (unsigned)s;
        if (j < n) {
            const double dj = Dl[j];
            cpgw::gst(B.D, j, dj); cpgw::gst(B.Dinv, j, 1.0 / dj);
            qr[s] = cs * dj * qr[s];
            cpgw::gst(B.q, j, qr[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < NZ; s++) {
        const unsigned i = (unsigned)tid + T * (unsigned)s;
        out.ct[s] = 0;
        if (i < m) {
            const double ei = El[i], uu = ei * ur[s];
            cpgw::gst(B.E, i, ei); cpgw::gst(B.Einv, i, 1.0 / ei); cpgw::gst(B.u, i, uu);
            out.ct[s] = i < n_eq ? 1 : (uu > CPG_INFTY * CPG_MIN_SCALING ? -1 : 0);
            if (out.ct[s] == -1) free_rows |= 1u << s;
            cpgw::gst(B.rinv, i, out.ct[s] == 1 ? ri_eq : (out.ct[s] == 0 ? ri_in : ri_fr));
        }
    }
    cpgw::block_sync();                    // (D, E are dead: the scaled matrices take the front of the slice)
#pragma unroll
    for (int t = 0; t < KA; t++) { const unsigned k = (unsigned)tid + T * (unsigned)t; if (k < nnzA) { Al[k] = av[t]; cpgw::gst(B.A, k, av[t]); } }
#pragma unroll
    for (int t = 0; t < KP; t++) { const unsigned k = (unsigned)tid + T * (unsigned)t; if (k < nnzP) { Pl[k] = pv[t]; cpgw::gst(B.P, k, pv[t]); } }
    cpgw::block_sync();
    auto copy_values = [&](const DevStreamTab &Tb, double *dst, const CPG_LDS double *src) __attribute__((always_inline)) {
#pragma nounroll
        for (unsigned e0 = 0; e0 < (unsigned)Tb.n_entries; e0 += 8u * T) {
            int kk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const unsigned e = e0 + T * (unsigned)u + (unsigned)tid; kk[u] = e < (unsigned)Tb.n_entries ? cpgw::gld(Tb.src, e) : -1; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const unsigned e = e0 + T * (unsigned)u + (unsigned)tid;
                if (e < (unsigned)Tb.n_entries) cpgw::gst(dst, e, kk[u] >= 0 ? src[(unsigned)kk[u]] : 0.0);
            }
        }
    };
    copy_values(Rs.pA, B.cA, Al);
    copy_values(Rs.pAt, B.cAt, Al);
    copy_values(Rs.pP, B.cP, Pl);
    cpgw::mem_order();
    cpgw::block_sync();
    out.cs = cs; out.dconst = dconst; out.free_rows = free_rows;
}

// The combined factorisation + block-inverse schedule as a table walk in batches of CPG_TEAM_FAC_BATCH steps (DevResident::bf_*): the
// LDS reads of a batch are issued together, the operand words of the next CPG_TEAM_FAC_DEPTH batches are on their way from
// memory while a batch is consumed.  Wavefront 0 walks the LDL' part -- a chain of levels one chunk wide on the families this
// kernel exists for, nothing to share -- then the team meets, and every level of the block inverses is spread over the
// wavefronts, one barrier per level.  Same arithmetic, in the same order, as the straight-line form and resident_factor.
#ifndef CPG_TEAM_FAC_DEPTH
#define CPG_TEAM_FAC_DEPTH 4
#endif
#ifndef CPG_TEAM_FAC_BATCH
#ifdef CPG_GENT_FAC_BATCH
#define CPG_TEAM_FAC_BATCH CPG_GENT_FAC_BATCH      // chosen per family by codegen.team_fac_batch (padding steps against batches)
#else
#define CPG_TEAM_FAC_BATCH 8          // steps per batch: a chain level of 5 - 7 steps is ONE batch = one LDS round trip
#endif
#endif
struct alignas(16) TeamQuad { unsigned x, y, z, w; };
// (CPG_TEAM_FACTOR_PROBE: experiments only -- two more time stamps inside the factorisation; merely carrying the pointer cost the
// loop its register allocation: 194 -> 228 us, profiles/r5_s6_team_factor_probe.txt)
#ifdef CPG_TEAM_FACTOR_PROBE
#define CPG_TEAM_TS_PARAM , unsigned long long *ts = nullptr
#define CPG_TEAM_TS_ARG , ts
#else
#define CPG_TEAM_TS_PARAM
#define CPG_TEAM_TS_ARG
#endif
CPG_DEV void team_factor_batched(const DevResident &Rs, CPG_LDS double *fac_, int lane, int wave CPG_TEAM_TS_PARAM) {
    CPG_LDS double *fac = cpgw::pin_lds(fac_);
    constexpr int DP = CPG_TEAM_FAC_DEPTH, S = CPG_TEAM_FAC_BATCH, NQ = (3 * S / 2 + 3) / 4;
    static_assert(S % 2 == 0, "an even number of steps per batch: two steps share three words");
    const unsigned first = cpgw::sld(Rs.bf_hdr, 2u * (unsigned)wave), nb = cpgw::sld(Rs.bf_hdr, 2u * (unsigned)wave + 1u);
    // [batch][NQ][lane]: positions a, b, k of the batch's steps as 16-bit element numbers, three words per TWO steps -- 48 bytes per lane
    // and batch of eight steps; the tables of a long schedule (MPC 12/4/10 with every parameter: 770 batches) then stay inside an
    // XCD's 4 MB of L2 (as byte offsets, three words per step, they took 5.6 MB and 81 GB of fetches per 20 000 instances)
    const TeamQuad *tri = (const TeamQuad *)Rs.bf_tri;
    const CPG_LDS char *fb = (const CPG_LDS char *)fac;
    TeamQuad e[DP][NQ];
    unsigned dk[DP];
    // (the batch's flags ride in bits 22 - 29 of every lane's destination word: a scalar load per batch shares its counter with the
    // LDS reads -- the compiler drains it in front of the first ds_read of EVERY batch, a scalar-cache miss on the chain of each level)
    auto request = [&](int u, unsigned t) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NQ; k++) e[u][k] = cpgw::gld(tri, ((first + t) * (unsigned)NQ + (unsigned)k) * 64u + (unsigned)lane);
        dk[u] = cpgw::gld(Rs.bf_dk, (first + t) * 64u + (unsigned)lane);
    };
#pragma unroll
    for (int u = 0; u < DP; u++) request(u, (unsigned)u);
    double acc = 0.0;
#ifdef CPG_TEAM_FOLLOW_CHAIN
    // (experiments, with plans built under CPG_TEAM_GROUP_SECTIONS=1: the block inverses of a merged group need that group's rows of
    // the factor only -- wavefront 0 counts its complete LDL' levels in LDS, the wavefront that inverts a group follows it level by
    // level, no barrier.  Measured SLOWER, 215 against 204 us, and the mere presence of these tests in the loop costs 20 us:
    // profiles/r5_s8_*, r5_s9_team_with_follow_chain_code_in_loop.txt.  Off.)
    CPG_LDS unsigned *progress = (CPG_LDS unsigned *)cpgw::lds_window3() + 2;
    unsigned passed = 0u;
#endif
#pragma nounroll
    for (unsigned t0 = 0; t0 < nb; t0 += DP) {
#pragma unroll
        for (int u = 0; u < DP; u++) {
            unsigned o_[4 * NQ];
#pragma unroll
            for (int k = 0; k < NQ; k++) { o_[4 * k] = e[u][k].x; o_[4 * k + 1] = e[u][k].y; o_[4 * k + 2] = e[u][k].z; o_[4 * k + 3] = e[u][k].w; }
            const unsigned d = dk[u], du = (unsigned)cpgw::read_first_lane((int)d), fl = (du >> 22) & 0xFFu;
            request(u, t0 + (unsigned)(u + DP));
#ifdef CPG_TEAM_FOLLOW_CHAIN
            if (du & 0x200000u) cpgw::lds_spin_until_ge(progress, du & 0xFFFFFu);      // (a batch without work: its word carries the count)
#endif
            if (fl & 1u) acc = 0.0;
            double av[S], kv[S], bv[S];
            // (the destination's KKT value is requested with the operands: nothing of this chunk has stored yet, and the value is not
            // touched before the chunk's own store -- one LDS round trip less on the chain of a level; lanes without a task read the
            // zero slot their word points at)
#ifdef CPG_TEAM_FOLLOW_CHAIN
            const double dv = *(const CPG_LDS double *)(fb + ((du & 0x200000u) ? 0u : (d & 0xFFFFFu)));
#else
            const double dv = *(const CPG_LDS double *)(fb + (d & 0xFFFFFu));
#endif
#pragma unroll
            for (int k = 0; k < S; k += 2) {
                const unsigned w0 = o_[3 * (k / 2)], w1 = o_[3 * (k / 2) + 1], w2 = o_[3 * (k / 2) + 2];
#if defined(CPG_TEAM_FAC_EXPERIMENT) && CPG_TEAM_FAC_EXPERIMENT == 3
                av[k] = fac[w0 & 0xFFFFu]; bv[k] = 1.0; kv[k] = 1e-3; av[k + 1] = fac[w1 >> 16]; bv[k + 1] = 1.0; kv[k + 1] = 1e-3; (void)w2;
#else
                av[k] = *cpgw::lds_elem16<0>(fac, w0); bv[k] = *cpgw::lds_elem16<1>(fac, w0); kv[k] = *cpgw::lds_elem16<0>(fac, w1);
                av[k + 1] = *cpgw::lds_elem16<1>(fac, w1); bv[k + 1] = *cpgw::lds_elem16<0>(fac, w2); kv[k + 1] = *cpgw::lds_elem16<1>(fac, w2);
#endif
            }
#pragma unroll
            for (int k = 0; k < S; k++) acc = fma(av[k] * kv[k], bv[k], acc);
            if (fl & 2u) {
#if defined(CPG_TEAM_FAC_EXPERIMENT) && CPG_TEAM_FAC_EXPERIMENT == 2
                const double r = acc;
#elif defined(CPG_TEAM_FAC_FLAT_REDUCE)
                const double r = cpgw::group_sum_first_flat(acc, (int)((fl >> 5) & 7u));      // (experiments: six masked stages, no branch)
#else
                const double r = cpgw::group_sum_first_dyn(acc, (int)((fl >> 5) & 7u));
#endif
                const double v = dv - r;
                double st = v;
#if !defined(CPG_TEAM_FAC_EXPERIMENT) || CPG_TEAM_FAC_EXPERIMENT != 1      // (timing experiments: 1 no division, 2 no reduction, 3 one operand read per step)
                if (fl & 16u) st = (d >> 31) ? 1.0 / v : v;      // (a chunk without a pivot: no division at all)
#endif
                if (!(d & 0x40000000u)) *(CPG_LDS double *)((CPG_LDS char *)fac + (d & 0xFFFFFu)) = st;
            }
            if (fl & 4u) cpgw::lds_order();
#ifdef CPG_TEAM_FOLLOW_CHAIN
            if (du & 0x100000u) {            // one more level of the LDL' chain is complete (its stores are: the level end above)
                passed++;
                if (lane == 0) cpgw::lds_signal(progress, passed);
            }
#endif
            if (fl & 8u) {
                cpgw::block_sync();
#ifdef CPG_TEAM_FACTOR_PROBE
                if (__builtin_expect(ts != nullptr, 0) && !ts[1]) ts[1] = cpgw::clock100();      // (the LDL' part is done)
#endif
            }
        }
    }
}

// ---- step 4: KKT values into the slice (all threads), numeric LDL' + inverses of the merged diagonal blocks (wavefront 0:
//      the schedule is a chain of levels, most of them one chunk wide)
CPG_DEV_NOINLINE void team_factorise(const DevRefactor &R_, const DevResident &Rs_, const ResBuf &B_, double sigma CPG_TEAM_TS_PARAM) {
    const int tid = (int)cpgw::thread_in_block(), lane = cpgw::lane_id(), wave = cpgw::read_first_lane(cpgw::wave_in_block());
    constexpr unsigned T = CPG_TEAM_T;
    const DevResident Rs = uniform_global_copy(Rs_); const ResBuf B = uniform_global_copy(B_);
    (void)R_;
    CPG_LDS double *sl = cpgw::pin_lds(cpgw::lds_window3() + CPG_TEAM_SLICE_OFF);
    constexpr unsigned nd = CPG_GENT_NNZL + CPG_GENT_N + CPG_GENT_M;
    constexpr int KD = (int)((nd + T - 1) / T);
#ifndef CPG_TEAM_KKT_BATCH
#define CPG_TEAM_KKT_BATCH 32
#endif
    constexpr int KB = KD < CPG_TEAM_KKT_BATCH ? KD : CPG_TEAM_KKT_BATCH;      // (two dependent loads per destination: every batch is two round trips)
    const unsigned lk = (unsigned)cpgw::opaque(tid);
#pragma unroll
    for (int t0 = 0; t0 < KD; t0 += KB) {
        unsigned code[KB];
        double v[KB];
#pragma unroll
        for (int u = 0; u < KB; u++) { const unsigned d = lk + T * (unsigned)(t0 + u); code[u] = (t0 + u < KD && d < nd) ? cpgw::gld(Rs.k_src, d) : 0u; }
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const unsigned kind = (code[u] >> 28) & 7u, idx = code[u] & 0x0FFFFFFFu;
            v[u] = 0.0;
            if (kind == CPG_K_P) v[u] = cpgw::gld((const double *)B.P, idx);
            else if (kind == CPG_K_A) v[u] = cpgw::gld((const double *)B.A, idx);
            else if (kind == CPG_K_RHO) v[u] = -cpgw::gld((const double *)B.rinv, idx);
        }
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const unsigned d = lk + T * (unsigned)(t0 + u), kind = (code[u] >> 28) & 7u;
            double vv = v[u];
            if (kind == CPG_K_P) vv = vv + (d >= (unsigned)CPG_GENT_NNZL ? sigma : 0.0);
            else if (kind == CPG_K_SIGMA) vv = sigma;
            if (t0 + u < KD && d < nd) sl[d] = (code[u] >> 31) ? 1.0 / vv : vv;
        }
    }
    for (unsigned d = nd + (unsigned)tid; d < (unsigned)Rs.fac_len; d += T) sl[d] = d == (unsigned)Rs.fac_len - 2u ? 1.0 : 0.0;
#ifdef CPG_TEAM_FOLLOW_CHAIN
    if (tid == 0) cpgw::lds_signal((CPG_LDS unsigned *)cpgw::lds_window3() + 2, 0u);       // (progress count of team_factor_batched)
#endif
    cpgw::block_sync();
#ifdef CPG_TEAM_FACTOR_PROBE
    if (__builtin_expect(ts != nullptr, 0)) { ts[0] = cpgw::clock100(); ts[1] = 0ull; }       // (KKT values are in place)
#endif
#if defined(CPG_TEAM_TABLE_FACTOR)
    if (wave == 0) resident_factor(Rs, (double *)sl, lane);        // (experiments: the flat one-step-at-a-time stream of the resident kernel's fallback)
#elif defined(CPG_GENT_FAC_GENERATED) && !defined(CPG_TEAM_BATCHED_FACTOR)
    if (wave == 0) team_factor_gen(Rs.gf_tri, Rs.gf_dk, (double *)sl, lane);
#else
    team_factor_batched(Rs, sl, lane, wave CPG_TEAM_TS_ARG);     // (its last batch is a barrier of the team)
#endif
    cpgw::block_sync();
}

// ---- step 5: every wavefront's coefficients (-l_ij = -M_ij / d_j, 1 / d_i, X_ij or 1 per register and lane) to the team's
//      buffer in the layout the iteration function loads them in ([wave][register][lane]); the slice back to its ADMM use
//      (work vector | q | u | results of the termination test's products; 1 / D and 1 / E are read into registers by the test)
CPG_DEV_NOINLINE void team_store_coefficients(const DevRefactor &R_, const DevResident &Rs_, const ResBuf &B_) {
    const int tid = (int)cpgw::thread_in_block(), lane = cpgw::lane_id(), wave = cpgw::read_first_lane(cpgw::wave_in_block());
    constexpr unsigned T = CPG_TEAM_T;
    const DevResident Rs = uniform_global_copy(Rs_); const ResBuf B = uniform_global_copy(B_);
    (void)R_;
    CPG_LDS double *sl = cpgw::pin_lds(cpgw::lds_window3() + CPG_TEAM_SLICE_OFF);
    constexpr unsigned n = CPG_GENT_N, m = CPG_GENT_M;
    constexpr int ldw = CPG_GENT_NSLOTS + CPG_GEN_EXTRA_SLOTS;
    {
        const unsigned ln = (unsigned)cpgw::opaque(lane);
        const unsigned base = (unsigned)wave * (unsigned)CPG_GENT_NREGS * 64u;
        constexpr int NB = CPG_GENT_NREGS < 32 ? CPG_GENT_NREGS : 32;
#pragma unroll
        for (int t0 = 0; t0 < CPG_GENT_NREGS; t0 += NB) {
            // (two positions in the factor array per register and lane, precomputed on the host -- DevResident::g_pos --: the
            // coefficient is +-(sl[a] * sl[b]), both reads unconditional, no branch on its kind)
            unsigned long long pos[NB];
            double va[NB], vb[NB];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int t = t0 + u;
                pos[u] = t < CPG_GENT_NREGS ? cpgw::gld((const unsigned long long *)Rs.g_pos, base + (unsigned)t * 64u + ln) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int t = t0 + u;
                if (t >= CPG_GENT_NREGS) break;
                va[u] = sl[(unsigned)pos[u] & 0x7FFFFFFFu]; vb[u] = sl[(unsigned)(pos[u] >> 32)];
            }
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int t = t0 + u;
                if (t >= CPG_GENT_NREGS) break;
                const double pr = va[u] * vb[u];
                const double v = ((unsigned)pos[u] >> 31) ? -pr : pr;
#ifdef CPG_TEAM_STREAM_HINTS
                cpgw::gst_stream(B.cf, base + (unsigned)t * 64u + ln, v);
#else
                cpgw::gst(B.cf, base + (unsigned)t * 64u + ln, v);
#endif
            }
        }
    }
    cpgw::block_sync();                    // (every wavefront has read the factor)
    CPG_LDS double *w = sl, *qs = w + ldw;
    for (unsigned t = (unsigned)tid; t < (unsigned)Rs.slice_doubles; t += T) w[t] = 0.0;
    cpgw::block_sync();
    {
        constexpr int KV = (int)((n + m + T - 1) / T);
        double vq[KV];
#pragma unroll
        for (int u_ = 0; u_ < KV; u_++) {
            const unsigned i = T * (unsigned)u_ + (unsigned)tid;
            vq[u_] = i < n ? cpgw::gld((const double *)B.q, i) : (i < n + m ? cpgw::gld((const double *)B.u, i - n) : 0.0);
        }
#pragma unroll
        for (int u_ = 0; u_ < KV; u_++) {
            const unsigned i = T * (unsigned)u_ + (unsigned)tid;
            if (i < n + m) qs[i] = vq[u_];
        }
    }
    cpgw::mem_order();
    cpgw::block_sync();
}

// ---- One ADMM iteration: right-hand side to the work vector, the team's program, relaxation / projection / dual update on
//      the thread's own entries.  No barrier behind the read-out: a thread rewrites only the entries it alone reads, and the
//      program's last barrier is behind every gather of the iteration.
template <int NX, int NZ>
CPG_DEV void team_step(double (&x)[NX], double (&z)[NZ], double (&y)[NZ], const double (&cf)[CPG_GENT_NREGS],
                       const unsigned (&of)[CPG_GENT_NOFF], const unsigned (&rw)[CPG_GENT_NROW], CPG_LDS double *w,
                       const CPG_LDS double *qs, const CPG_LDS double *us, const ResRho rr, unsigned free_rows, int tid, int wave) {
    constexpr unsigned n = CPG_GENT_N, m = CPG_GENT_M, n_eq = CPG_GENT_NEQ, T = CPG_TEAM_T;
    double qt[NX];
#pragma unroll
    for (int s = 0; s < NX; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; qt[s] = i < n ? qs[i] : 0.0; }
#pragma unroll
    for (int s = 0; s < NX; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; if (i < n) w[i] = rr.sigma * x[s] - qt[s]; }
#pragma unroll
    for (int s = 0; s < NZ; s++) {
        const unsigned i = (unsigned)tid + T * (unsigned)s;
        const double ri = i < n_eq ? +rr.ri_eq : (((free_rows >> s) & 1u) ? +rr.ri_fr : +rr.ri_in);
        if (i < m) w[n + i] = z[s] - ri * y[s];
    }
    cpgw::block_sync();
    run_program_team(cf, of, rw, w, wave);
#pragma unroll
    for (int s = 0; s < NX; s++) {
        const unsigned i = (unsigned)tid + T * (unsigned)s;
        const double xn = i < n ? rr.alpha * w[i] + (1.0 - rr.alpha) * x[s] : 0.0;
        x[s] = xn;
    }
#pragma unroll
    for (int s = 0; s < NZ; s++) {
        const unsigned i = (unsigned)tid + T * (unsigned)s;
        const bool eq = i < n_eq;
        const bool fr = (free_rows >> s) & 1u;
        const double rv = eq ? +rr.rho_eq : (fr ? +rr.rho_fr : +rr.rho_in);
        const double ri = eq ? +rr.ri_eq : (fr ? +rr.ri_fr : +rr.ri_in);
        const double zp = z[s], yp = y[s];
        const double zt = (zp - ri * yp) + ri * (i < m ? w[n + i] : 0.0);
        const double zr = rr.alpha * zt + (1.0 - rr.alpha) * zp;
        const double uu = i < m ? us[i] : 0.0;
        const double zn = eq ? uu : cpgw::dmin2(zr + ri * yp, uu);
        const double dyv = rv * (zr - zn);
        z[s] = i < m ? zn : 0.0; y[s] = i < m ? yp + dyv : 0.0;
    }
}

// `count` ADMM iterations, the last one keeping its steps delta x / delta y for the termination test.  The only function
// that runs the generated executor: x, z, y, the wavefront's coefficients, operand offsets and output slots are loaded
// once per call (= per termination test).
template <int NX, int NZ>
CPG_DEV_NOINLINE void team_iterate(TeamState<NX, NZ> &st, const ResRho &rr_, const double *cfg, const unsigned *offg, const unsigned *rowg,
                                   unsigned free_rows, int count_v) {
    const int tid = (int)cpgw::thread_in_block(), lane = cpgw::lane_id(), wave = cpgw::read_first_lane(cpgw::wave_in_block());
    const int count = cpgw::read_first_lane(count_v);
    constexpr unsigned n = CPG_GENT_N;
    constexpr int ldw = CPG_GENT_NSLOTS + CPG_GEN_EXTRA_SLOTS;
    auto uniform_ptr = [](const void *p) __attribute__((always_inline)) {
        const unsigned long long a = (unsigned long long)p;
        return ((unsigned long long)(unsigned)cpgw::read_first_lane((int)(a >> 32)) << 32) | (unsigned)cpgw::read_first_lane((int)a);
    };
    const double *cfp = cpgw::as_global((const double *)uniform_ptr(cfg));
    const unsigned *offp = cpgw::as_global((const unsigned *)uniform_ptr(offg)), *rowp = cpgw::as_global((const unsigned *)uniform_ptr(rowg));
    CPG_LDS double *w = cpgw::pin_lds(cpgw::lds_window3() + CPG_TEAM_SLICE_OFF);
    const CPG_LDS double *qs = w + ldw, *us = qs + n;
    const ResRho rr{cpgw::sgpr_value(rr_.rho_eq), cpgw::sgpr_value(rr_.rho_in), cpgw::sgpr_value(rr_.rho_fr), cpgw::sgpr_value(rr_.ri_eq),
                    cpgw::sgpr_value(rr_.ri_in), cpgw::sgpr_value(rr_.ri_fr), cpgw::sgpr_value(rr_.sigma), cpgw::sgpr_value(rr_.alpha)};     // (values, not a struct in scratch behind the selects of the step)
    double cf[CPG_GENT_NREGS];
    unsigned of[CPG_GENT_NOFF], rw[CPG_GENT_NROW];
#pragma unroll
#ifdef CPG_TEAM_STREAM_HINTS
    for (int t = 0; t < CPG_GENT_NREGS; t++) cf[t] = cpgw::gld_stream(cfp, ((unsigned)wave * (unsigned)CPG_GENT_NREGS + (unsigned)t) * 64u + (unsigned)lane);
#else
    for (int t = 0; t < CPG_GENT_NREGS; t++) cf[t] = cpgw::gld(cfp, ((unsigned)wave * (unsigned)CPG_GENT_NREGS + (unsigned)t) * 64u + (unsigned)lane);
#endif
#pragma unroll
    for (int t = 0; t < CPG_GENT_NOFF; t++) of[t] = cpgw::gld(offp, ((unsigned)wave * (unsigned)CPG_GENT_NOFF + (unsigned)t) * 64u + (unsigned)lane);
#pragma unroll
    for (int t = 0; t < CPG_GENT_NROW; t++) rw[t] = cpgw::gld(rowp, ((unsigned)wave * (unsigned)CPG_GENT_NROW + (unsigned)t) * 64u + (unsigned)lane);
    double x[NX], z[NZ], y[NZ];
#pragma unroll
    for (int s = 0; s < NX; s++) x[s] = st.x[s];
#pragma unroll
    for (int s = 0; s < NZ; s++) { z[s] = st.z[s]; y[s] = st.y[s]; }
    // count - 1 iterations, then the checked one (ONE copy of the iteration's code); the steps of the checked iteration are
    // delta x = x(k+1) - x(k), delta y = y(k+1) - y(k), from the iterates saved in front of it
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {
        const int nk = pass == 0 ? count - 1 : (count > 0 ? 1 : 0);
#pragma nounroll
        for (int k = 0; k < nk; k++) team_step<NX, NZ>(x, z, y, cf, of, rw, w, qs, us, rr, free_rows, tid, wave);
        if (pass == 0) {
#pragma unroll
            for (int s = 0; s < NX; s++) st.dx[s] = x[s];
#pragma unroll
            for (int s = 0; s < NZ; s++) st.dy[s] = y[s];
        }
    }
    if (count > 0) {
#pragma unroll
        for (int s = 0; s < NX; s++) st.dx[s] = x[s] - st.dx[s];
#pragma unroll
        for (int s = 0; s < NZ; s++) st.dy[s] = y[s] - st.dy[s];
    }
#pragma unroll
    for (int s = 0; s < NX; s++) st.x[s] = x[s];
#pragma unroll
    for (int s = 0; s < NZ; s++) { st.z[s] = z[s]; st.y[s] = y[s]; }
    cpgw::block_sync();                    // (the work vector is the next stage's)
}

// ---- update_info + check_termination on 64 W threads: check() of cpg_osqp_kernel.h restated for a team (entry i on thread
//      i % T, reductions over the team, the three products through the team's row executors), OSQP's own order: the
//      infeasibility tests inside, on dx / dy, when the matching residual test has failed
template <int NX, int NZ>
struct TeamCheck {
    const DevFamily &F;
    const DevResident &Rs;
    const ResBuf &B;
    CPG_LDS double *w;
    const CPG_LDS double *qm, *um;
    const signed char (&ct)[NZ];
    TeamRed &tr;
    int tid, lane, wave;
    const double (&dinv_r)[NX];          // 1 / D, 1 / E of the thread's entries (loaded at the test's entry)
    const double (&einv_r)[NZ];
    static constexpr unsigned n = CPG_GENT_N, m = CPG_GENT_M, T = CPG_TEAM_T;
    CPG_DEV double sDinv(int s) const { return dinv_r[s]; }
    CPG_DEV double sEinv(int s) const { return einv_r[s]; }
    CPG_DEV double staged(unsigned i) const { return w[(unsigned)Rs.out_ax + i]; }
    CPG_DEV void stage(int which) const {          // D (2) or E (1) into the products' result slots (consumed before a product runs)
        const double *src = which == 1 ? (const double *)B.E : (const double *)B.D;
        const unsigned cnt = which == 1 ? m : n;
        cpgw::block_sync();
        for (unsigned i0 = 0; i0 < cnt; i0 += 4u * T) {
            double v[4];
#pragma unroll
            for (int u_ = 0; u_ < 4; u_++) { const unsigned i = i0 + T * (unsigned)u_ + (unsigned)tid; v[u_] = i < cnt ? cpgw::gld(src, i) : 0.0; }
#pragma unroll
            for (int u_ = 0; u_ < 4; u_++) { const unsigned i = i0 + T * (unsigned)u_ + (unsigned)tid; if (i < cnt) w[(unsigned)Rs.out_ax + i] = v[u_]; }
        }
        cpgw::block_sync();
    }
    CPG_DEV void products(int which) const {        // 1: A w[0..n)   2: P w[0..n)   4: A' w[n..n+m)
        // rows without an entry are never written by their program, and A x shares the slots of P x | A' y: clear first
        cpgw::block_sync();
        if (which & 1) for (unsigned i = (unsigned)tid; i < m; i += T) w[(unsigned)Rs.out_ax + i] = 0.0;
        if (which & 2) for (unsigned i = (unsigned)tid; i < n; i += T) w[(unsigned)Rs.out_px + i] = 0.0;
        if (which & 4) for (unsigned i = (unsigned)tid; i < n; i += T) w[(unsigned)Rs.out_aty + i] = 0.0;
        cpgw::block_sync();
        CPG_LDS double *wl = w;
        if (which & 1) run_rows_a_team(B.cA, (const unsigned *)Rs.pA.gcols, (const unsigned *)Rs.pA.grows, wl, lane, wave);
        if (which & 2) run_rows_p_team(B.cP, (const unsigned *)Rs.pP.gcols, (const unsigned *)Rs.pP.grows, wl, lane, wave);
        if (which & 4) run_rows_t_team(B.cAt, (const unsigned *)Rs.pAt.gcols, (const unsigned *)Rs.pAt.grows, wl, lane, wave);
        cpgw::block_sync();
    }
    CPG_DEV double ax(unsigned i) const { return i < m ? w[(unsigned)Rs.out_ax + i] : 0.0; }
    CPG_DEV double px(unsigned i) const { return i < n ? w[(unsigned)Rs.out_px + i] : 0.0; }
    CPG_DEV double atx(unsigned i) const { return i < n ? w[(unsigned)Rs.out_aty + i] : 0.0; }

    // is_primal_infeasible on delta_y (cpg_osqp_kernel.h primal_infeasible)
    CPG_DEV bool primal_infeasible(bool unsc, double eps, const double (&dy)[NZ]) const {
        stage(1);
        double nrm[1] = {0.0}, lhs[1] = {0.0};
        double dyp[NZ];
#pragma unroll
        for (int s = 0; s < NZ; s++) {
            const unsigned i = (unsigned)tid + T * (unsigned)s;
            dyp[s] = 0.0;
            if (i < m) {
                const double uu = um[i];
                const bool eq = ct[s] == 1;
                const double ll = eq ? uu : -CPG_INFTY;
                const bool iu = uu > CPG_INFTY * CPG_MIN_SCALING, il = !eq;
                double d = dy[s];
                if (iu && il) d = 0.0; else if (iu) d = cpgw::dmin2(d, 0.0); else if (il) d = cpgw::dmax2(d, 0.0);
                dyp[s] = d;
                nrm[0] = cpgw::dmax2(nrm[0], fabs(unsc ? staged(i) * d : d));
                lhs[0] += uu * cpgw::dmax2(d, 0.0) + ll * cpgw::dmin2(d, 0.0);
            }
        }
        team_reduce<1, false>(nrm, tr, lane, wave);
        team_reduce<1, true>(lhs, tr, lane, wave);
        if (!(nrm[0] > CPG_DIV_TOL)) return false;
        if (!(lhs[0] < eps * nrm[0])) return false;
#pragma unroll
        for (int s = 0; s < NZ; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; if (i < m) w[n + i] = dyp[s]; }
        products(4);
        double r[1] = {0.0};
#pragma unroll
        for (int s = 0; s < NX; s++) {
            const unsigned i = (unsigned)tid + T * (unsigned)s;
            const double t = atx(i);
            if (i < n) r[0] = cpgw::dmax2(r[0], fabs(unsc ? sDinv(s) * t : t));
        }
        team_reduce<1, false>(r, tr, lane, wave);
        return r[0] < eps * nrm[0];
    }
    // is_dual_infeasible on delta_x (cpg_osqp_kernel.h dual_infeasible)
    CPG_DEV bool dual_infeasible(bool unsc, double eps, const double (&dx)[NX]) const {
        stage(2);
        double nrm[1] = {0.0}, qdx[1] = {0.0};
#pragma unroll
        for (int s = 0; s < NX; s++) {
            const unsigned i = (unsigned)tid + T * (unsigned)s;
            if (i < n) {
                const double d = dx[s];
                nrm[0] = cpgw::dmax2(nrm[0], fabs(unsc ? staged(i) * d : d));
                qdx[0] += qm[i] * d;
            }
        }
        team_reduce<1, false>(nrm, tr, lane, wave);
        team_reduce<1, true>(qdx, tr, lane, wave);
        if (!(nrm[0] > CPG_DIV_TOL)) return false;
        const double cs = unsc ? F.c : 1.0;
        if (!(qdx[0] < -cs * eps * nrm[0])) return false;
#pragma unroll
        for (int s = 0; s < NX; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; if (i < n) w[i] = dx[s]; }
        products(2);
        double r[1] = {0.0};
#pragma unroll
        for (int s = 0; s < NX; s++) {
            const unsigned i = (unsigned)tid + T * (unsigned)s;
            const double t = px(i);
            if (i < n) r[0] = cpgw::dmax2(r[0], fabs(unsc ? sDinv(s) * t : t));
        }
        team_reduce<1, false>(r, tr, lane, wave);
        bool res = false;
        if (r[0] < cs * eps * nrm[0]) {
            products(1);
            bool viol = false;
#pragma unroll
            for (int s = 0; s < NZ; s++) {
                const unsigned i = (unsigned)tid + T * (unsigned)s;
                const double a = ax(i);
                if (i < m) {
                    const double av = unsc ? sEinv(s) * a : a;
                    if ((um[i] < CPG_INFTY * CPG_MIN_SCALING && av > eps * nrm[0]) || (ct[s] == 1 && av < -eps * nrm[0])) viol = true;
                }
            }
            res = !team_any(viol, tr, lane, wave);
        }
        return res;
    }
};

template <int NX, int NZ>
CPG_DEV_NOINLINE CheckOut team_check(const DevFamily &F_, const DevResident &Rs_, const ResBuf &B_, const signed char (&ct_)[NZ],
                                     const DevSettings &S_, const TeamState<NX, NZ> &st_, bool approximate_v, ScaledNorms *sn_) {
    const int tid = (int)cpgw::thread_in_block(), lane = cpgw::lane_id(), wave = cpgw::read_first_lane(cpgw::wave_in_block());
    constexpr unsigned n = CPG_GENT_N, m = CPG_GENT_M, T = CPG_TEAM_T;
    const bool approximate = cpgw::read_first_lane(approximate_v ? 1 : 0) != 0;
    const DevFamily F = uniform_global_copy(F_); const DevResident Rs = uniform_global_copy(Rs_); const ResBuf B = uniform_global_copy(B_); const DevSettings S = uniform_copy(S_);
    double Ix[NX], Iz[NZ], Iy[NZ], dxr[NX], dyr[NZ];
    signed char ct[NZ];
#pragma unroll
    for (int s = 0; s < NX; s++) { Ix[s] = st_.x[s]; dxr[s] = st_.dx[s]; }
#pragma unroll
    for (int s = 0; s < NZ; s++) { Iz[s] = st_.z[s]; Iy[s] = st_.y[s]; dyr[s] = st_.dy[s]; ct[s] = ct_[s]; }
    CPG_LDS double *w = cpgw::pin_lds(cpgw::lds_window3() + CPG_TEAM_SLICE_OFF);
    const CPG_LDS double *qs = w + (CPG_GENT_NSLOTS + CPG_GEN_EXTRA_SLOTS), *us = qs + n;
    TeamRed tr{0u};
    double dinv_r[NX], einv_r[NZ];
#pragma unroll
    for (int s = 0; s < NX; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; dinv_r[s] = i < n ? cpgw::gld((const double *)B.Dinv, i) : 0.0; }
#pragma unroll
    for (int s = 0; s < NZ; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; einv_r[s] = i < m ? cpgw::gld((const double *)B.Einv, i) : 0.0; }
    const TeamCheck<NX, NZ> cx{F, Rs, B, w, qs, us, ct, tr, tid, lane, wave, dinv_r, einv_r};
    const bool unsc = !S.scaled_termination;
    const double mult = approximate ? 10.0 : 1.0;
    const double ea = S.eps_abs * mult, er = S.eps_rel * mult;
    CheckOut o;
#pragma unroll
    for (int s = 0; s < NX; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; if (i < n) w[i] = Ix[s]; }
#pragma unroll
    for (int s = 0; s < NZ; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; if (i < m) w[n + i] = Iy[s]; }
    cx.products(1);
    // max: rp nz na | scaled rp nz na ; sums: sup
    double mx1[6] = {0, 0, 0, 0, 0, 0}, sm[3] = {0, 0, 0};      // sm: quad, lin, sup
#pragma unroll
    for (int s = 0; s < NZ; s++) {
        const unsigned i = (unsigned)tid + T * (unsigned)s;
        const double ax = cx.ax(i);
        if (i < m) {
            const double ei = unsc ? cx.sEinv(s) : 1.0;
            mx1[0] = cpgw::dmax2(mx1[0], fabs(ei * (ax - Iz[s])));
            mx1[1] = cpgw::dmax2(mx1[1], fabs(ei * Iz[s]));
            mx1[2] = cpgw::dmax2(mx1[2], fabs(ei * ax));
            mx1[3] = cpgw::dmax2(mx1[3], fabs(ax - Iz[s])); mx1[4] = cpgw::dmax2(mx1[4], fabs(Iz[s])); mx1[5] = cpgw::dmax2(mx1[5], fabs(ax));
            if (S.check_dualgap) {   // support function of [l, u] at y: u'y+ + l'y-  (l = u on equality rows, -inf otherwise)
                const double uu = us[i], yy = Iy[s];
                if (uu < CPG_INFTY * CPG_MIN_SCALING && yy > 0.0) sm[2] += uu * yy;
                if (ct[s] == 1 && uu > -CPG_INFTY * CPG_MIN_SCALING && yy < 0.0) sm[2] += uu * yy;
            }
        }
    }
    cx.products(6);
    // max: rd nq nat npx | scaled the same
    double mx2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NX; s++) {
        const unsigned i = (unsigned)tid + T * (unsigned)s;
        const double px = cx.px(i), aty = cx.atx(i);
        if (i < n) {
            const double di = unsc ? cx.sDinv(s) : 1.0;
            const double qq = qs[i];
            mx2[0] = cpgw::dmax2(mx2[0], fabs(di * (qq + px + aty)));
            mx2[1] = cpgw::dmax2(mx2[1], fabs(di * qq));
            mx2[2] = cpgw::dmax2(mx2[2], fabs(di * aty));
            mx2[3] = cpgw::dmax2(mx2[3], fabs(di * px));
            mx2[4] = cpgw::dmax2(mx2[4], fabs(qq + px + aty)); mx2[5] = cpgw::dmax2(mx2[5], fabs(qq));
            mx2[6] = cpgw::dmax2(mx2[6], fabs(aty)); mx2[7] = cpgw::dmax2(mx2[7], fabs(px));
            sm[0] += Ix[s] * px;
            sm[1] += qq * Ix[s];
        }
    }
    team_reduce<6, false>(mx1, tr, lane, wave);
    team_reduce<8, false>(mx2, tr, lane, wave);
    team_reduce<3, true>(sm, tr, lane, wave);
    const double cs = unsc ? F.cinv : 1.0;
    const double rp = mx1[0], nz = mx1[1], na = mx1[2];
    const double rd = cs * mx2[0];
    const double dn = cs * cpgw::dmax2(mx2[1], cpgw::dmax2(mx2[2], mx2[3]));
    const double quad = sm[0], lin = sm[1], sup = sm[2];
    o.prim_res = rp; o.dual_res = rd; o.obj = (0.5 * quad + lin) * F.cinv;
    if (sn_) {
        ScaledNorms sn;
        sn.prim_res = mx1[3]; sn.nz = mx1[4]; sn.nax = mx1[5];
        sn.dual_res = mx2[4]; sn.nq = mx2[5]; sn.naty = mx2[6]; sn.npx = mx2[7];
        *sn_ = sn;
    }
    o.status = 11;
    if (rp > CPG_INFTY || rd > CPG_INFTY) { o.status = 9; o.obj = NAN; cpgw::block_sync(); return o; }
    bool pc = false, dc = false, pic = false, dic = false, gc = true;
    if (m == 0) pc = true;
    else if (rp < ea + er * cpgw::dmax2(nz, na)) pc = true;
    else pic = cx.primal_infeasible(unsc, S.eps_prim_inf * mult, dyr);
    if (rd < ea + er * dn) dc = true;
    else dic = cx.dual_infeasible(unsc, S.eps_dual_inf * mult, dxr);
    if (S.check_dualgap) {   // OSQP >= 1.0: |primal - dual objective| against eps_abs + eps_rel max(|primal|, |dual|)
        const double dual_obj = (-0.5 * quad - sup) * F.cinv, gap = fabs(quad + lin + sup) * F.cinv;
        gc = gap < ea + er * cpgw::dmax2(fabs(o.obj), fabs(dual_obj));
    }
    if (pc && dc && gc) o.status = approximate ? 2 : 1;
    else if (pic) { o.status = approximate ? 4 : 3; o.obj = CPG_INFTY; }
    else if (dic) { o.status = approximate ? 6 : 5; o.obj = -CPG_INFTY; }
    cpgw::block_sync();
    return o;
}

// ---- store_solution + cpg_retrieve_* (finalize of cpg_osqp_kernel.h on 64 W threads)
template <int NX, int NZ>
CPG_DEV_NOINLINE void team_finalize(const DevFamily &F_, const DevBatch &Bt_, const TeamState<NX, NZ> &st_, double dconst, long long b_v,
                                    int iter, const CheckOut &o_, double rho) {
    const int tid = (int)cpgw::thread_in_block();
    constexpr unsigned n = CPG_GENT_N, m = CPG_GENT_M, T = CPG_TEAM_T;
    const long long b = ((long long)cpgw::read_first_lane((int)(b_v >> 32)) << 32) | (unsigned)cpgw::read_first_lane((int)b_v);
    const DevFamily F = uniform_global_copy(F_); const DevBatch Bt = uniform_global_copy(Bt_); const CheckOut o = o_;
    CPG_LDS double *w = cpgw::pin_lds(cpgw::lds_window3() + CPG_TEAM_SLICE_OFF);
    const bool has_sol = o.status == 1 || o.status == 2 || o.status == 7;
    if (Bt.state_out) {
        double *so = Bt.state_out + (size_t)b * (size_t)(n + 2u * m + 1u);
#pragma unroll
        for (int s = 0; s < NX; s++) {
            const unsigned i = (unsigned)tid + T * (unsigned)s;
            if (i < n) cpgw::gst(so, F.ord ? (unsigned)cpgw::gld(F.ord, i) : i, has_sol ? st_.x[s] : 0.0);
        }
#pragma unroll
        for (int s = 0; s < NZ; s++) {
            const unsigned i = (unsigned)tid + T * (unsigned)s;
            if (i < m) {
                const unsigned c = F.ord ? (unsigned)cpgw::gld(F.ord, n + i) : i;
                cpgw::gst(so, n + c, has_sol ? st_.z[s] : 0.0);
                cpgw::gst(so, n + m + c, has_sol ? st_.y[s] : 0.0);
            }
        }
        if (tid == 0) so[n + 2u * m] = rho;
    }
    cpgw::block_sync();
#pragma unroll
    for (int s = 0; s < NX; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; if (i < n) w[i] = has_sol ? cpgw::gld(F.D, i) * st_.x[s] : NAN; }
#pragma unroll
    for (int s = 0; s < NZ; s++) { const unsigned i = (unsigned)tid + T * (unsigned)s; if (i < m) w[n + i] = has_sol ? F.cinv * cpgw::gld(F.E, i) * st_.y[s] : NAN; }
    cpgw::block_sync();
    double *pp = Bt.prim + (size_t)b * F.n_prim, *dp = Bt.dual + (size_t)b * F.n_dual;
    for (unsigned k = (unsigned)tid; k < (unsigned)F.n_prim; k += T) cpgw::gst(pp, k, w[(unsigned)cpgw::gld(F.prim_idx, k)]);
    for (unsigned k = (unsigned)tid; k < (unsigned)F.n_dual; k += T) cpgw::gst(dp, k, w[n + (unsigned)cpgw::gld(F.dual_idx, k)]);
    if (tid == 0) {
        double ov = o.obj + dconst;
        if (F.is_max) ov = -ov;
        Bt.obj[b] = ov; Bt.iter[b] = iter; Bt.status[b] = o.status;
        Bt.pri_res[b] = o.prim_res; Bt.dua_res[b] = o.dual_res;
    }
    cpgw::block_sync();
}

// ---- the kernel body: one workgroup = one team = one instance at a time, instances pulled from a global counter
template <int NX, int NZ>
CPG_DEV void osqp_team_body(const DevFamily &F0, const DevRefactor &R, const DevResident &Rs, const DevSettings &S,
                            const DevBatch &Bt, double *lds, int team_global) {
    const int tid = (int)cpgw::thread_in_block();
    constexpr unsigned n = CPG_GENT_N, m = CPG_GENT_M, T = CPG_TEAM_T;
    const ResBuf B = res_carve(Bt.scratch + (size_t)team_global * (size_t)Rs.buf_doubles, F0, R, Rs, CPG_TEAM_W * CPG_GENT_NREGS);
    const double rho_fr = CPG_RHO_MIN, ri_fr = 1.0 / rho_fr;
    const size_t state_len = (size_t)n + 2u * (size_t)m + 1u;
    const unsigned n_work = Bt.list_count ? cpgw::sld(Bt.list_count, 0u) : 0u;
    unsigned *bc = (unsigned *)lds;          // broadcast word of the team

    for (;;) {
        cpgw::block_sync();                  // (every thread has read the previous instance's number)
        if (tid == 0) bc[0] = cpgw::atomic_next(Bt.counter);
        cpgw::block_sync();
        const unsigned ig = (unsigned)cpgw::read_first_lane((int)bc[0]);
        long long b = (long long)ig;
        if (Bt.list) {
            if (ig >= n_work) break;
            b = (long long)cpgw::read_first_lane(cpgw::gld(Bt.list, ig));
        } else if (b >= Bt.B) break;
        const double *theta = Bt.theta + (size_t)b * R.np_var;
        const double *state_in = (Bt.state_in && (S.warm_starting || Bt.resume)) ? Bt.state_in + (size_t)b * state_len : nullptr;
        double rho = Bt.state_in ? cpgw::gld(Bt.state_in + (size_t)b * state_len, n + 2u * m) : F0.rho;
        rho = cpgw::dmin2(cpgw::dmax2(rho, CPG_RHO_MIN), CPG_RHO_MAX);
        double rho_stg = F0.rho;
        double rho_eq = 1e3 * rho, rho_in = rho, ri_eq = 1.0 / rho_eq, ri_in = 1.0 / rho_in;
        // (experiments, debug_stage 20: the 100 MHz time stamps of the instance's stages replace its primal results)
        const bool probe = __builtin_expect(S.debug_stage == 20 || S.debug_stage == 23, 0);       // (23: two more stamps inside the factorisation)
        unsigned long long ts[8];
        int n_ts = 0;
#define CPG_TEAM_PROBE() do { if (probe && n_ts < 8) ts[n_ts++] = cpgw::clock100(); } while (0)
        CPG_TEAM_PROBE();
        TeamSetupOut<NZ> su;
        team_setup<NX, NZ>(R, Rs, B, theta, ri_eq, ri_in, ri_fr, su);
        const double cs = su.cs, dconst = su.dconst;
        CPG_TEAM_PROBE();
        DevFamily F = F0;
        F.D = B.D; F.Dinv = B.Dinv; F.E = B.E; F.Einv = B.Einv; F.c = cs; F.cinv = 1.0 / cs;
        TeamState<NX, NZ> st;
#pragma unroll
        for (int s = 0; s < NX; s++) { st.x[s] = 0.0; st.dx[s] = 0.0; }
#pragma unroll
        for (int s = 0; s < NZ; s++) { st.z[s] = 0.0; st.y[s] = 0.0; st.dy[s] = 0.0; }
        if (state_in) {
#pragma unroll
            for (int s = 0; s < NX; s++) {
                const unsigned i = (unsigned)tid + T * (unsigned)s;
                if (i < n) st.x[s] = cpgw::gld(state_in, F.ord ? (unsigned)cpgw::gld(F.ord, i) : i);
            }
#pragma unroll
            for (int s = 0; s < NZ; s++) {
                const unsigned i = (unsigned)tid + T * (unsigned)s;
                if (i < m) {
                    const unsigned c = F.ord ? (unsigned)cpgw::gld(F.ord, n + i) : i;
                    st.z[s] = cpgw::gld(state_in, n + c); st.y[s] = cpgw::gld(state_in, n + m + c);
                }
            }
        }
        CheckOut o;
        o.prim_res = 0; o.dual_res = 0; o.obj = 0; o.status = 11;
        int iter = Bt.resume ? cpgw::read_first_lane(cpgw::gld((const int *)Bt.iter, (unsigned)b)) : 0;
        if (iter > 0) rho_stg = rho;
        bool need_factor = true;
        const int chk_int = S.check_termination, ad_int = S.adaptive_rho ? S.adaptive_rho_interval : 0;
#pragma nounroll
        while (o.status == 11) {
            if (need_factor) {
#ifdef CPG_TEAM_FACTOR_PROBE
                unsigned long long fts[2] = {0ull, 0ull};
                team_factorise(R, Rs, B, F0.sigma, __builtin_expect(S.debug_stage == 23, 0) ? fts : nullptr);
                if (__builtin_expect(S.debug_stage == 23, 0) && n_ts < 7) { ts[n_ts++] = fts[0]; ts[n_ts++] = fts[1]; }
#else
                team_factorise(R, Rs, B, F0.sigma);
#endif
                CPG_TEAM_PROBE();
                team_store_coefficients(R, Rs, B);
                CPG_TEAM_PROBE();
                need_factor = false;
            }
            if (iter < S.max_iter) {
                int next_ev = S.max_iter;
                if (chk_int > 0) { const int c = (iter / chk_int + 1) * chk_int; if (c < next_ev) next_ev = c; }
                if (ad_int > 0) { const int c = (iter / ad_int + 1) * ad_int; if (c < next_ev) next_ev = c; }
                const ResRho rr{rho_eq, rho_in, rho_fr, ri_eq, ri_in, ri_fr, F0.sigma, F0.alpha};
                team_iterate<NX, NZ>(st, rr, B.cf, Rs.t_off, Rs.t_row, su.free_rows, next_ev - iter);
                iter = next_ev;
                CPG_TEAM_PROBE();
            }
            const bool can_check = chk_int > 0 && iter > 0 && iter % chk_int == 0;
            const bool adapt = ad_int > 0 && iter > 0 && iter % ad_int == 0;
            const bool last = iter >= S.max_iter;
            ScaledNorms sn;
            bool approx = false;
            for (;;) {
                const CheckOut oc = team_check<NX, NZ>(F, Rs, B, su.ct, S, st, approx, &sn);
                CPG_TEAM_PROBE();
                if (approx) { o = oc; break; }
                if (can_check) { o = oc; if (o.status != 11) break; }
                if (adapt) {
                    const double rn = rho_estimate(sn, rho_stg);
                    if (rn > rho_stg * S.adaptive_rho_tolerance || rn < rho_stg / S.adaptive_rho_tolerance) {
                        rho = rn; rho_stg = rn; rho_eq = 1e3 * rho; rho_in = rho; ri_eq = 1.0 / rho_eq; ri_in = 1.0 / rho_in;
#pragma unroll
                        for (int s = 0; s < NZ; s++) {
                            const unsigned i = (unsigned)tid + T * (unsigned)s;
                            if (i < m) cpgw::gst(B.rinv, i, su.ct[s] == 1 ? ri_eq : (su.ct[s] == 0 ? ri_in : ri_fr));
                        }
                        cpgw::mem_order();
                        cpgw::block_sync();
                        need_factor = true;
                    }
                }
                if (last) {
                    if (!can_check) o = oc;
                    if (o.status == 11) { approx = true; continue; }
                }
                break;
            }
            if (last && o.status == 11) o.status = 7;
        }
        team_finalize<NX, NZ>(F, Bt, st, dconst, b, iter, o, rho);
        if (probe) {
            CPG_TEAM_PROBE();
            if (tid == 0) for (int k = 0; k < n_ts && k < F0.n_prim; k++) Bt.prim[(size_t)b * F0.n_prim + k] = (double)(ts[k] - ts[0]);
        }
    }
}

}  // namespace cpg
#endif  // CPG_GENT_HEADER
