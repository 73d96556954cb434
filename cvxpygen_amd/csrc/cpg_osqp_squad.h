// SQUAD shared-factor kernel (round 6): the path of cpg_osqp_kernel.h --
//   cpg_canonicalize_q/l/u/d (cvxpygen/utils.py:279-294), osqp_update_data_vec (solvers/osqp.py:39-59),
//   osqp_solve (solvers/osqp.py:62), cpg_retrieve_* (utils.py:950-985)
// for batches whose KKT matrix is the family's -- with the family's solve program in REGISTERS instead of LDS.
//
// osqp_shared_kernel keeps the program (coefficients + operand offsets, 84 KB for MPC 12/4/10) in LDS and every multiply-add of
// every instance reads its coefficient, its offset and its operand from there: the LDS array is 85 % busy and the kernel sits on
// that roof (profiles/r5_final_pmc_config2.txt).  The coefficients are the same for every instance.  Here a workgroup of
// W = CPG_GENQ_W wavefronts (a squad) solves W instances at a time:
//   * the program is planned for a team (solve_program.pack_ragged(team=W)); wavefront v holds the coefficients, operand
//     offsets and output slots of ITS steps in registers (~36 steps: 72 + 18 + 6 VGPRs), loaded once per kernel;
//   * wavefront v runs its steps for ALL W instances: the work vectors of two instances are interleaved ([slot][2] doubles, a
//     "pair array"), one ds_read_b128 brings a step's operand of both, the coefficient goes into W accumulators; reduce, one
//     ds_write_b128 per pair, one barrier per phase (codegen.emit_squad_program);
//   * wavefront v OWNS instance v of the squad: iterates x, z, y in its registers, right-hand side / relaxation / projection,
//     termination test (check() of cpg_osqp_kernel.h on a private plain vector), retrieval, hand-over after a rho change;
//   * the squad steps in lock step: before a run of iterations every wavefront posts the distance to its next event
//     (termination test, rho adaptation, max_iter; "idle" when the batch is exhausted) and the squad runs the minimum.
//     Instances start at events of their predecessors, so with the reference's settings all four are aligned.
// Only the operand gathers and the reduce-stores reach the LDS: ~0.6 KB per multiply-add step of four instances instead of 4.6.
#pragma once

#include "cpg_osqp_kernel.h"

#ifdef CPG_GENQ_HEADER
// experiments only (-DCPG_SQUAD_PROBE): wavefront w of workgroup 0 leaves the shader clock behind every barrier of one run of
// iterations in a device array and prints it when the kernel ends
#ifdef CPG_SQUAD_PROBE
namespace cpg { __device__ unsigned long long squad_probe[8 * 128]; __device__ int squad_probe_on; }
#define CPG_SQUAD_TS(K) do { if (cpg::squad_probe_on && blockIdx.x == 0 && (threadIdx.x & 63) == 0) cpg::squad_probe[(threadIdx.x >> 6) * 128 + (K)] = __builtin_readcyclecounter(); } while (0)
#ifdef CPG_SQUAD_PROBE_FINE
#define CPG_SQUAD_TS2(K) CPG_SQUAD_TS(K)
#else
#define CPG_SQUAD_TS2(K) do { } while (0)
#endif
#else
#define CPG_SQUAD_TS2(K) do { } while (0)
#define CPG_SQUAD_TS(K) do { } while (0)
#endif
namespace cpg {
typedef double SquadPair __attribute__((vector_size(16)));       // two instances' entries of one slot: a 16-byte LDS access
}
#include CPG_GENQ_HEADER

namespace cpg {

#if CPG_GENQ_PARENT_FINGERPRINT != CPG_GEN_FINGERPRINT
#error "the squad executor was generated from another solve program than the library's LDS executor"
#endif

// bytes of LDS a squad needs: base vectors | control words | pair arrays | one plain vector per wavefront (termination test)
constexpr unsigned squad_lds_bytes(unsigned n, unsigned m) {
    return 8u * (n + m + ((n + m) & 1u)) + 8u * 16u + (CPG_GENQ_W / 2) * CPG_GENQ_PAIR_STRIDE + CPG_GENQ_W * 8u * (n + m + ((n + m) & 1u));
}

#define CPG_SQUAD_IDLE 0x7FFFFFFF

template <int NSX, int NSZ, int NV>
CPG_DEV void osqp_squad_body(const DevFamily &F, const DevUpdate &U, const DevSettings &S, const DevBatch &Bt, double *lds) {
    typedef Inst<NSX, NSZ, NV> InstT;
    constexpr int W = CPG_GENQ_W;
    constexpr unsigned n_c = GenFam::n, m_c = GenFam::m;
    constexpr unsigned N = n_c + m_c, NPAD = N + (N & 1u);
    static_assert((n_c + 63) / 64 == (unsigned)NSX && (m_c + 63) / 64 == (unsigned)NSZ, "slot class of the generated family");
    static_assert(CPG_GENQ_NSLOTS == CPG_GEN_NSLOTS, "the squad program and the LDS program number their slots alike");
    const int lane0 = cpgw::lane_id();
    const int wave = cpgw::read_first_lane(cpgw::wave_in_block());
    // ---- LDS
    double *sh = lds;                                                   // [q_base | u_base] of the family
    for (unsigned t = cpgw::thread_in_block(); t < N; t += cpgw::block_threads())
        sh[t] = t < n_c ? cpgw::gld(U.q_base, t) : cpgw::gld(U.u_base, t - n_c);
    const double *shu = sh + n_c;
    int *ctl = (int *)(lds + NPAD);                                     // [W] iterations to the next event of every wavefront
    CPG_LDS char *pairs = (CPG_LDS char *)(lds + NPAD + 16);            // W / 2 pair arrays
    double *wp = lds + NPAD + 16 + (W / 2) * (CPG_GENQ_PAIR_STRIDE / 8u) + (unsigned)wave * NPAD;    // this wavefront's plain vector
    // this wavefront's instance inside the pair arrays: slot i at mine + 16 i
    CPG_LDS char *mine = pairs + (unsigned)(wave >> 1) * CPG_GENQ_PAIR_STRIDE + (unsigned)(wave & 1) * 8u;
    // ---- the wavefront's share of the program, for the whole kernel
    double cf[CPG_GENQ_NREGS];
    unsigned of[CPG_GENQ_NOFF], rw[CPG_GENQ_NROW];
#pragma unroll
    for (int r = 0; r < CPG_GENQ_NREGS; r++) cf[r] = cpgw::gld(genq_cf, ((unsigned)wave * CPG_GENQ_NREGS + (unsigned)r) * 64u + (unsigned)lane0);
#pragma unroll
    for (int r = 0; r < CPG_GENQ_NOFF; r++) of[r] = cpgw::gld(genq_off, ((unsigned)wave * CPG_GENQ_NOFF + (unsigned)r) * 64u + (unsigned)lane0);
#pragma unroll
    for (int r = 0; r < CPG_GENQ_NROW; r++) rw[r] = cpgw::gld(genq_row, ((unsigned)wave * CPG_GENQ_NROW + (unsigned)r) * 64u + (unsigned)lane0);
    // the pair arrays start from zeros (idle lanes of partial steps gather with everybody else: every slot they can touch must
    // hold a finite number), the slot of zeros stays zero for the whole kernel
    for (unsigned t = cpgw::thread_in_block(); t < (W / 2) * (CPG_GENQ_PAIR_STRIDE / 8u); t += cpgw::block_threads()) ((CPG_LDS double *)pairs)[t] = 0.0;
    // (N odd: the pad entry behind the base vectors and behind every plain vector is read by no one's arithmetic, but must not be left to chance)
    // ... and so does the wavefront's plain vector: the first infeasibility test stages delta_y behind entry n and its row program's
    // idle lanes multiply a zero coefficient into entry 0, which nobody has written yet -- whatever the LDS held, times zero, must be
    // zero (found by the emulator's poisoned LDS, tests/sim/fake_hip: CPG_SIM_LDS_POISON)
    for (unsigned t = (unsigned)lane0; t < NPAD; t += 64u) wp[t] = 0.0;
    cpgw::block_sync();
    const double rho_eq = 1e3 * F.rho, rho_in = F.rho, rho_fr = 1e-6;
    const double ri_eq = 1.0 / rho_eq, ri_in = 1.0 / rho_in, ri_fr = 1.0 / rho_fr;
    signed char ct_reg[NSZ];
    // where the program leaves entry i of the solution: BYTE offsets into a pair array (slot * 16), two per register -- the read-out
    // adds a half-word to the (per-iteration opaque) base: one instruction per entry, nothing to hoist and keep alive
    unsigned fpp[(NSX + NSZ + 1) / 2];
#pragma unroll
    for (int s = 0; s < (NSX + NSZ + 1) / 2; s++) fpp[s] = 0u;
#pragma unroll
    for (int s = 0; s < NSX + NSZ; s++) {
        const unsigned i = s < NSX ? (unsigned)lane0 + 64u * (unsigned)s : n_c + (unsigned)lane0 + 64u * (unsigned)(s - NSX);
        const bool in = s < NSX ? i < n_c : i < N;
        const unsigned fp = in ? 16u * (unsigned)cpgw::gld(F.fpos, i) : 0u;
        fpp[s / 2] |= (s & 1) ? fp << 16 : fp;
    }
#pragma unroll
    for (int s = 0; s < NSZ; s++) {
        const unsigned i = (unsigned)lane0 + 64u * (unsigned)s;
        ct_reg[s] = (i < m_c) ? cpgw::gld(F.ctype, i) : 0;
    }
    const int chk_int = S.check_termination;
    const int ad_int = (S.adaptive_rho && S.adaptive_rho_interval > 0) ? S.adaptive_rho_interval : 0;

    InstT I;
    CheckOut co;
    bool active = false, exhausted = false;
    int iter = 0;
    co.prim_res = 0; co.dual_res = 0; co.obj = 0; co.status = 11;
#pragma unroll
    for (int s = 0; s < NSX; s++) I.x[s] = 0.0;
#pragma unroll
    for (int s = 0; s < NSZ; s++) { I.z[s] = 0.0; I.y[s] = 0.0; }
    I.b = -1; I.done = 1; I.dconst = 0.0;

    // OSQP's events of this wavefront's instance at iteration `iter` (the checked iteration's steps in dxr / dyr): termination
    // test, rho adaptation, max_iter -- the per-instance part of osqp_shared_body, one instance
    auto event = [&](const double (&dxr)[NSX], const double (&dyr)[NSZ]) __attribute__((always_inline)) {
        const int lane = cpgw::opaque(lane0);
        cpgw::assume((unsigned)lane < 64u);
        const bool last = iter >= S.max_iter;
        const bool can_check = last || (chk_int > 0 && iter % chk_int == 0);
        const bool adapt = ad_int > 0 && iter > 0 && iter % ad_int == 0;
        InfeasVerdict iv_exact = InfeasVerdict{false, false}, iv_approx = InfeasVerdict{false, false};
        if (can_check) {
#pragma nounroll
            for (int pass = 0; pass < (last ? 2 : 1); pass++) {
                const InfeasVerdict v = infeasibility_tests<NSX, NSZ, SharedCtx<NSX, NSZ, NV>, RegDelta<NSX>, RegDelta<NSZ>>(
                    F, SharedCtx<NSX, NSZ, NV>{F, sh, shu, I, wp, lane}, ct_reg, S, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr}, wp, lane, pass == 1);
                if (pass == 0) iv_exact = v; else iv_approx = v;
            }
        }
        CheckOut o = co;
        ScaledNorms sn;
        double rho_ws = F.rho;
        bool rho_changed = false;
#pragma nounroll
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1 && !(o.status == 11 && last)) break;
            const CheckOut oc = check<NSX, NSZ, SharedCtx<NSX, NSZ, NV>, NoDelta, NoDelta>(
                F, SharedCtx<NSX, NSZ, NV>{F, sh, shu, I, wp, lane}, ct_reg, S, I.x, I.z, I.y, NoDelta{}, NoDelta{},
                pass == 0 ? iv_exact : iv_approx, wp, lane, pass == 1, (pass == 0 && adapt) ? &sn : nullptr);
            if (can_check) o = oc;
            if (pass == 0 && adapt && o.status == 11) {
                const double rn = rho_estimate(sn, F.rho);
                if (rn > F.rho * S.adaptive_rho_tolerance || rn < F.rho / S.adaptive_rho_tolerance) { rho_ws = rn; rho_changed = true; }
            }
        }
        if (o.status == 11 && last) o.status = 7;
        co = o;
        if (o.status != 11) {
            finalize<NSX, NSZ>(F, Bt, I.x, I.z, I.y, I.dconst, I.b, wp, lane, iter, o, rho_ws);
            active = false;
        } else if (rho_changed) {
            hand_over<NSX, NSZ>(F, Bt, I.x, I.z, I.y, I.b, lane, iter, rho_ws);
            active = false;
        }
    };

    for (;;) {
        const int lane = cpgw::opaque(lane0);
        cpgw::assume((unsigned)lane < 64u);
        // ---- an idle wavefront takes the next instance of the batch
        while (!active && !exhausted) {
            unsigned ig = 0;
            if (lane == 0) ig = cpgw::atomic_next(Bt.counter);
            ig = (unsigned)cpgw::read_first_lane((int)ig);
            if ((long long)ig >= Bt.B) { exhausted = true; break; }
            const long long b = (long long)ig;
            I.b = b;
            co.prim_res = 0; co.dual_res = 0; co.obj = 0; co.status = 11;
            iter = 0;
            active = true;
            const double *theta = Bt.theta + (size_t)b * U.np_var;
            const bool bad = canonicalise<NSX, NSZ, NV>(F, U, theta, I, lane);
            if (Bt.state_in && S.warm_starting)
                load_state<NSX, NSZ>(F, Bt.state_in + (size_t)b * (size_t)(F.n + 2 * F.m + 1), I.x, I.z, I.y, lane);
            if (Bt.state_in && !bad) {
                const double rho_st = cpgw::gld(Bt.state_in + (size_t)b * (size_t)(F.n + 2 * F.m + 1), (unsigned)(F.n + 2 * F.m));
                if (__builtin_expect(cpgw::dmin2(cpgw::dmax2(rho_st, CPG_RHO_MIN), CPG_RHO_MAX) != F.rho, 0)) {
                    hand_over<NSX, NSZ>(F, Bt, I.x, I.z, I.y, b, lane, 0, rho_st);
                    active = false;
                }
            }
            if (__builtin_expect(bad && active, 0)) {
                // a row changed class: flagged for the per-instance factor path (cvxpygen_amd/runtime.py), as osqp_shared_body does
                if (lane == 0) { Bt.obj[b] = NAN; Bt.iter[b] = 0; Bt.status[b] = -2; Bt.pri_res[b] = 0.0; Bt.dua_res[b] = 0.0; }
                active = false;
            }
            if (active && iter >= S.max_iter) {          // max_iter <= 0: the test runs on the initial iterates
                double dx0[NSX], dy0[NSZ];
#pragma unroll
                for (int s = 0; s < NSX; s++) dx0[s] = 0.0;
#pragma unroll
                for (int s = 0; s < NSZ; s++) dy0[s] = 0.0;
                event(dx0, dy0);
                active = false;                          // (status 7 at the latest: check() leaves nothing unsolved at max_iter)
            }
            if (active) {
                // its half of the pair slots starts from zeros: no stale NaN / Inf of the instance before
                for (unsigned t = (unsigned)lane; t < (unsigned)(CPG_GENQ_NSLOTS + CPG_GEN_DUMMY_SLOTS); t += 64u) *(CPG_LDS double *)(mine + 16u * t) = 0.0;
            }
        }
        // ---- the squad agrees on a run of iterations: the shortest distance to an event
        int my_k = CPG_SQUAD_IDLE;
        if (active) {
            int next_ev = S.max_iter;
            if (chk_int > 0) { const int c = (iter / chk_int + 1) * chk_int; if (c < next_ev) next_ev = c; }
            if (ad_int > 0) { const int c = (iter / ad_int + 1) * ad_int; if (c < next_ev) next_ev = c; }
            my_k = next_ev - iter;
        }
        if (lane == 0) ctl[wave] = my_k;
        cpgw::block_sync();
        int k = CPG_SQUAD_IDLE;
#pragma unroll
        for (int v = 0; v < W; v++) { const int kv = ctl[v]; k = kv < k ? kv : k; }
        k = cpgw::read_first_lane(k);
        if (k == CPG_SQUAD_IDLE) break;                  // nobody has an instance and the batch is exhausted
        // (ctl is written again after the barriers of the run below)

        // ---- k iterations in lock step
        auto rhs = [&]() __attribute__((always_inline)) {
            const int ln = cpgw::opaque(lane);
            cpgw::assume((unsigned)ln < 64u);
            signed char ct[NSZ];
#pragma unroll
            for (int s = 0; s < NSZ; s++) ct[s] = (signed char)cpgw::opaque((int)ct_reg[s]);
            if (active) {
#pragma unroll
                for (int s = 0; s < NSX; s++) {
                    const unsigned i = (unsigned)ln + 64u * (unsigned)s;
                    if (i < n_c) *(CPG_LDS double *)(mine + 16u * i) = F.sigma * I.x[s] - SharedCtx<NSX, NSZ, NV>{F, sh, shu, I, wp, ln}.q(s, i);
                    CPG_FENCE_EVERY(s);
                }
#pragma unroll
                for (int s = 0; s < NSZ; s++) {
                    const unsigned i = (unsigned)ln + 64u * (unsigned)s;
                    const int cts = CPG_ROW_CLASS(s);
                    const double ri = cts == 1 ? ri_eq : (cts == 0 ? ri_in : ri_fr);
                    if (i < m_c) *(CPG_LDS double *)(mine + 16u * (n_c + i)) = I.z[s] - ri * I.y[s];
                    CPG_FENCE_EVERY(s);
                }
            }
        };
        auto update = [&](auto stash_c, double (&dxr)[NSX], double (&dyr)[NSZ]) __attribute__((always_inline)) {
            constexpr bool STASH = decltype(stash_c)::value;
            const int ln = cpgw::opaque(lane);
            cpgw::assume((unsigned)ln < 64u);
            const CPG_LDS char *mine_o = cpgw::pin_lds(mine);
            signed char ct[NSZ];
#pragma unroll
            for (int s = 0; s < NSZ; s++) ct[s] = (signed char)cpgw::opaque((int)ct_reg[s]);
            if (STASH) {
#pragma unroll
                for (int s = 0; s < NSX; s++) dxr[s] = 0.0;
#pragma unroll
                for (int s = 0; s < NSZ; s++) dyr[s] = 0.0;
            }
            if (active) {
#pragma unroll
                for (int s = 0; s < NSX; s++) {
                    const unsigned i = (unsigned)ln + 64u * (unsigned)s;
                    if (i < n_c) {
                        const double xn = F.alpha * *(const CPG_LDS double *)(mine_o + ((s & 1) ? fpp[s / 2] >> 16 : fpp[s / 2] & 0xFFFFu)) + (1.0 - F.alpha) * I.x[s];
                        if (STASH) dxr[s] = xn - I.x[s];
                        I.x[s] = xn;
                    }
                    CPG_FENCE_EVERY(s);
                }
#pragma unroll
                for (int s = 0; s < NSZ; s++) {
                    const unsigned i = (unsigned)ln + 64u * (unsigned)s;
                    if (i < m_c) {
                        const int cts = CPG_ROW_CLASS(s);
                        const double rv = cts == 1 ? rho_eq : (cts == 0 ? rho_in : rho_fr);
                        const double ri = cts == 1 ? ri_eq : (cts == 0 ? ri_in : ri_fr);
                        const double zp = I.z[s], yp = I.y[s];
                        const double zt = (zp - ri * yp) + ri * *(const CPG_LDS double *)(mine_o + (((NSX + s) & 1) ? fpp[(NSX + s) / 2] >> 16 : fpp[(NSX + s) / 2] & 0xFFFFu));
                        const double zr = F.alpha * zt + (1.0 - F.alpha) * zp;
                        const double uu = SharedCtx<NSX, NSZ, NV>{F, sh, shu, I, wp, ln}.u(s, i);
                        const double zn = cts == 1 ? uu : cpgw::dmin2(zr + ri * yp, uu);
                        const double dyv = rv * (zr - zn);
                        I.z[s] = zn; I.y[s] = yp + dyv;
                        if (STASH) dyr[s] = dyv;
                    }
                    CPG_FENCE_EVERY(s);
                }
            }
            cpgw::lds_order();          // the read-out above before the next right-hand side (other lanes' slots)
        };
        double dxr[NSX], dyr[NSZ];
        int it = 0;
#pragma nounroll
        for (;;) {
            it++;
#ifdef CPG_SQUAD_PROBE
            if (blockIdx.x == 0 && lane == 0) squad_probe_on = (it == 7 && iter == 25) ? 1 : 0;
#endif
            CPG_SQUAD_TS(30);
            rhs();
            CPG_SQUAD_TS(31);
            cpgw::block_sync();
            CPG_SQUAD_TS(0);
            run_program_squad(cf, of, rw, pairs, wave);      // (ends with a barrier: every result is in place)
            if (it >= k) break;
            update(std::false_type{}, dxr, dyr);
            CPG_SQUAD_TS(32);
        }
        update(std::true_type{}, dxr, dyr);
        if (active) {
            iter += k;
            int next_ev = S.max_iter;
            if (chk_int > 0) { const int c = ((iter - 1) / chk_int + 1) * chk_int; if (c < next_ev) next_ev = c; }
            if (ad_int > 0) { const int c = ((iter - 1) / ad_int + 1) * ad_int; if (c < next_ev) next_ev = c; }
            if (iter >= next_ev) event(dxr, dyr);
        }
    }
#ifdef CPG_SQUAD_PROBE
    if (blockIdx.x == 0 && lane0 == 0) {
        const unsigned long long *t = squad_probe + wave * 128;
        for (int v = 0; v < wave; v++) __builtin_amdgcn_s_sleep(127);
        printf("squad probe wave %d: rhs %llu sync %llu\n", wave, t[31] - t[30], t[0] - t[31]);
        for (int p = 1; p <= 20 && t[p]; p++)
            printf("  wave %d phase %d: %llu  (gathers back %lld, multiply-adds issued %lld, reduce + stores done %lld, barrier %lld)\n", wave, p - 1, t[p] - t[p - 1],
                   t[40 + 4 * (p - 1)] ? (long long)(t[40 + 4 * (p - 1)] - t[p - 1]) : -1ll, t[41 + 4 * (p - 1)] ? (long long)(t[41 + 4 * (p - 1)] - t[40 + 4 * (p - 1)]) : -1ll,
                   t[41 + 4 * (p - 1)] ? (long long)(t[42 + 4 * (p - 1)] - t[41 + 4 * (p - 1)]) : -1ll, (long long)(t[p] - t[42 + 4 * (p - 1)]));
    }
#endif
}

}  // namespace cpg
#endif  // CPG_GENQ_HEADER
