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
    // generated row executors (run_rows_a / _p / _t of the family's cpg_resident_<name>.h): the values sit in program-entry
    // order (n_entries = nnz + 64 zeros of padding), operand offsets / output slots in the layout of run_program_res's tables
    const unsigned short *gcols, *grows;
};
struct DevEll {                   // out[k] = base[k] + sum_j coef[j * rows + k] * theta[idx[j * rows + k]]
    int J, rows;
    const int *idx;
    const double *coef;
};
#define CPG_RES_FAC_DEPTH 8       // steps of the factorisation stream in flight (each is a copy of the step's code)
#define CPG_RES_PRODUCT_DEPTH 16  // steps of a product of the termination test in flight (run_program_stream)
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
    const unsigned *g_pos;                 // the same as two positions in the factor array per (register, lane): a | negate << 31, b  (+-(fac[a] * fac[b]))
    const unsigned short *g_cols, *g_rows; // operand offsets / output slots of the generated executor (LDS tables)
    DevEll eP, eA, eq, eu;
    const unsigned *entA, *entP;           // row | column << 16 of every stored entry
    // generated factorisation (resident_factor_gen of the family's header): operand positions a | b << 16 | k << 32 per
    // (step, lane), destination | pivot << 16 per (chunk, lane) (0xFFFF: none)
    const unsigned long long *gf_tri;
    const unsigned *gf_dk;
    DevStreamTab pA, pP, pAt;              // A x, P x, A' y on the work vector [x | y | .. | A x | P x | A' y]
    int out_ax, out_px, out_aty;           // first slot of the products' results (A x shares the slots of P x | A' y)
    int out_sc;                            // ... and of 1 / D (n) | 1 / E (m): the residuals of every termination test unscale with them
    int slice_doubles;                     // LDS doubles per wavefront
    long long buf_doubles;                 // per-wavefront buffer in global memory
    // team kernel (cpg_osqp_team.h; ok == 2): per wavefront of the team the operand offsets of its steps and the output slots
    // of its chunks as the 32-bit register images the generated executor keeps (two 16-bit words each), [wave][NOFF | NROW][lane];
    // g_src / g_lcol are [wave][NREGS][lane] then, and pA / pP / pAt carry the same two tables of the row executors (gcols / grows
    // reinterpreted as 32-bit words)
    const unsigned *t_off, *t_row;
    // ... and its BATCHED factorisation (team_factor_batched): the combined schedule as batches of CPG_TEAM_FAC_BATCH steps, one list per wavefront
    // -- the LDL' part (a chain of levels, one chunk wide) on wavefront 0, the chunks of every level of the block inverses spread over
    // the team.  bf_hdr [wave][2] first batch | batches; bf_ctl [batch] 1 first of a chunk | 2 last | 4 level end inside a
    // wavefront's own section | 8 level end the team meets at | 16 the chunk holds a pivot | reduction stages << 5; bf_tri
    // [batch][quad][lane][4] positions in the factor of the operands a, b, k of the batch's steps, 16 bits each, three words per two
    // steps (idle: the zero slot); bf_dk [batch][lane] byte offset of the destination | the batch's bf_ctl flags << 22 | none << 30 (then: the zero slot) | pivot << 31
    const unsigned *bf_hdr, *bf_ctl, *bf_dk, *bf_tri;
};

#if defined(CPG_GENR_HEADER) || defined(CPG_GENT_HEADER)
// A wave-uniform struct that arrives by reference (the caller's stack): one batch of loads, and every word through
// v_readfirstlane -- the compiler then knows the pointers in it are uniform (scalar base addresses, s_load for the tables)
// instead of reloading a field from the stack, as a per-lane value, in front of each use.
template <class T>
CPG_DEV T uniform_copy(const T &src) {
    static_assert(sizeof(T) % 4 == 0, "word-sized structs only");
    T dst;
    int words[sizeof(T) / 4];
    __builtin_memcpy(words, &src, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) words[i] = cpgw::read_first_lane(words[i]);
    __builtin_memcpy(&dst, words, sizeof(T));
    return dst;
}
struct ResBuf { double *A, *P, *D, *Dinv, *E, *Einv, *q, *u, *rinv, *cA, *cP, *cAt, *cf; };
// every pointer of a struct that arrived through a call, marked as global memory (cpgw::as_global)
#define CPG_G(x) x = cpgw::as_global(x)
CPG_DEV void globalise(ResBuf &B) { CPG_G(B.A); CPG_G(B.P); CPG_G(B.D); CPG_G(B.Dinv); CPG_G(B.E); CPG_G(B.Einv); CPG_G(B.q); CPG_G(B.u); CPG_G(B.rinv); CPG_G(B.cA); CPG_G(B.cP); CPG_G(B.cAt); CPG_G(B.cf); }
CPG_DEV void globalise(DevStreamTab &T) { CPG_G(T.stab); CPG_G(T.cr); CPG_G(T.src); CPG_G(T.gcols); CPG_G(T.grows); }
CPG_DEV void globalise(DevEll &E) { CPG_G(E.idx); CPG_G(E.coef); }
CPG_DEV void globalise(DevResident &Rs) {
    CPG_G(Rs.f_ctl); CPG_G(Rs.f_ent); CPG_G(Rs.k_src); CPG_G(Rs.g_src); CPG_G(Rs.g_lcol); CPG_G(Rs.g_pos); CPG_G(Rs.g_cols); CPG_G(Rs.g_rows);
    globalise(Rs.eP); globalise(Rs.eA); globalise(Rs.eq); globalise(Rs.eu); CPG_G(Rs.entA); CPG_G(Rs.entP); CPG_G(Rs.gf_tri); CPG_G(Rs.gf_dk);
    globalise(Rs.pA); globalise(Rs.pP); globalise(Rs.pAt);
    CPG_G(Rs.t_off); CPG_G(Rs.t_row); CPG_G(Rs.bf_hdr); CPG_G(Rs.bf_ctl); CPG_G(Rs.bf_dk); CPG_G(Rs.bf_tri);
}
CPG_DEV void globalise(DevRefactor &R) {     // (the members the resident path reads)
    CPG_G(R.P_base); CPG_G(R.A_base); CPG_G(R.q_base); CPG_G(R.u_base); CPG_G(R.q_setup);
    CPG_G(R.map_d.ptr); CPG_G(R.map_d.idx); CPG_G(R.map_d.val);
}
CPG_DEV void globalise(DevFamily &F) { CPG_G(F.D); CPG_G(F.Dinv); CPG_G(F.E); CPG_G(F.Einv); CPG_G(F.prim_idx); CPG_G(F.dual_idx); CPG_G(F.ord); CPG_G(F.ctype); }
CPG_DEV void globalise(DevBatch &Bt) {
    CPG_G(Bt.theta); CPG_G(Bt.prim); CPG_G(Bt.dual); CPG_G(Bt.obj); CPG_G(Bt.pri_res); CPG_G(Bt.dua_res); CPG_G(Bt.iter); CPG_G(Bt.status);
    CPG_G(Bt.state_in); CPG_G(Bt.state_out);
}
#undef CPG_G
template <class T>
CPG_DEV T uniform_global_copy(const T &src) { T d = uniform_copy(src); globalise(d); return d; }
CPG_DEV ResBuf res_carve(double *b, const DevFamily &F, const DevRefactor &R, const DevResident &Rs, const int n_coef_regs) {
    ResBuf o;
    const size_t n = (size_t)F.n, m = (size_t)F.m;
    o.A = b; b += R.nnzA; o.P = b; b += R.nnzP;
    o.D = b; b += n; o.Dinv = b; b += n; o.E = b; b += m; o.Einv = b; b += m;
    o.q = b; b += n; o.u = b; b += m; o.rinv = b; b += m;
    o.cA = b; b += Rs.pA.n_entries; o.cP = b; b += Rs.pP.n_entries; o.cAt = b; b += Rs.pAt.n_entries;
    o.cf = b; b += 64 * n_coef_regs;
    return o;
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
                const double r = cpgw::group_sum_first_flat(acc, (int)(fl >> 4));
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

// the iterates of an instance between two calls / the step sizes of an ADMM iteration (both kernels)
struct ResRho { double rho_eq, rho_in, rho_fr, ri_eq, ri_in, ri_fr, sigma, alpha; };
// KKT value of a destination from its source word (kind << 28 | index): P, A and 1 / rho_vec live in ONE per-wavefront buffer
// (res_carve: A first), so the source is one element offset from B.A and the load is UNCONDITIONAL -- written as `if (kind == P) v =
// P[idx]; else if (kind == A) ...` every destination was a branch tree with a wait of its own: a batch's loads went out one memory
// round trip at a time (round 6: config 3 393.6 -> 399.6 k/s with this and unconditional coefficient reads, config 2's instance kernel
// 10.2 -> 10.1 ms with the same rewrite of load_instance_coefficients -- A/B on one box, profiles/r6_t7_*; in the team kernel the
// unconditional KKT loads changed nothing and are not in).  With the coefficient sources as POSITIONS precomputed on the host
// (DevResident::g_pos, below) the selects between source kinds are gone too: config 3 394 -> 406 k/s, the team kernel's store stage
// 25.9 -> 12.9 us and 312 -> 322 k/s (profiles/r6_t8_*).
CPG_DEV unsigned kkt_source_element(const ResBuf &B, unsigned code) {
    const unsigned kind = (code >> 28) & 7u, idx = code & 0x0FFFFFFFu;
    const unsigned oP = (unsigned)(B.P - B.A), oR = (unsigned)(B.rinv - B.A);
    return kind == CPG_K_P ? oP + idx : (kind == CPG_K_A ? idx : (kind == CPG_K_RHO ? oR + idx : 0u));
}
CPG_DEV double kkt_source_value(unsigned code, double raw) {
    const unsigned kind = (code >> 28) & 7u;
    return (kind == CPG_K_P || kind == CPG_K_A) ? raw : (kind == CPG_K_RHO ? -raw : 0.0);
}
// A substitution coefficient -- 1 | -M_ij / d_j | 1 / d_i | X_ij -- is +-(fac[a] * fac[b]) with the two positions precomputed on the host
// (DevResident::g_pos; the array's slots 1.0 and 0.0 stand in where a factor is missing: x * 1.0 and 0.0 * 1.0 are exact): two
// UNCONDITIONAL reads per register, no branch on the kind.
#endif  // CPG_GENR_HEADER || CPG_GENT_HEADER

#ifdef CPG_GENR_HEADER
// the instance's coefficients of the generated executor from `fac`: -l_ij = -M_ij / d_j, 1 / d_i, X_ij or 1 per (coefficient
// register, lane), to the wavefront's buffer in the layout the iteration function loads them in ([register][lane])
CPG_DEV void resident_coefficients(const DevRefactor &R, const DevResident &Rs, const ResBuf &B, const double *fac, int lane) {
    const unsigned ln = (unsigned)cpgw::opaque(lane);            // (see load_instance_coefficients: addresses local to this block)
    (void)R;
    // (the source words of 48 registers requested together: with one wavefront per SIMD every batch is an exposed round trip)
    constexpr int NB = 48;
#pragma unroll
    for (int t0 = 0; t0 < CPG_GENR_NREGS; t0 += NB) {
        unsigned long long pos[NB];
        double va[NB], vb[NB];
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int t = t0 + u;
            pos[u] = t < CPG_GENR_NREGS ? cpgw::gld((const unsigned long long *)Rs.g_pos, (unsigned)t * 64u + ln) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int t = t0 + u;
            if (t >= CPG_GENR_NREGS) break;
            va[u] = fac[(unsigned)pos[u] & 0x7FFFFFFFu]; vb[u] = fac[(unsigned)(pos[u] >> 32)];
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int t = t0 + u;
            if (t >= CPG_GENR_NREGS) break;
            const double pr = va[u] * vb[u];
            cpgw::gst(B.cf, (unsigned)t * 64u + ln, ((unsigned)pos[u] >> 31) ? -pr : pr);
        }
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
    unsigned w_off;               // ... as an offset into the workgroup's LDS window (for the calls)
    const double *qm, *um;
    int lane;
    int skip_products;            // (experiments, debug_stage 21: the termination test without its three products)
    CPG_DEV double q(int, unsigned i) const { return qm[i]; }
    CPG_DEV double u(int, unsigned i) const { return um[i]; }
    // Scaling vectors: with one wavefront per SIMD a global load per 64-entry slot inside the test's loops is a memory
    // round trip each (~60 of them per test, most of its 96 us: profiles/r4_s3_*).  1 / D and 1 / E sit in the slice; D / E,
    // which only the infeasibility tests read, are staged into the products' result slots when such a test starts (one
    // batch of loads) and consumed before the test's first product overwrites them.
#ifdef CPG_GENR_TABLES_GLOBAL
    // (four wavefronts per CU: 1 / E alone stays in the slice; 1 / D sits in registers of the test, loaded at its entry --
    // i = lane + 64 s with lane < 64, so i >> 6 is the literal s after unrolling)
    const double (&dinv_r)[NSX];
    CPG_DEV double sDinv(unsigned i) const { return dinv_r[i >> 6]; }
    CPG_DEV double sEinv(unsigned i) const { return w[(unsigned)Rs.out_sc + i]; }
#else
    CPG_DEV double sDinv(unsigned i) const { return w[(unsigned)Rs.out_sc + i]; }
    CPG_DEV double sEinv(unsigned i) const { return w[(unsigned)Rs.out_sc + (unsigned)F.n + i]; }
#endif
    CPG_DEV double sE(unsigned i) const { return w[(unsigned)Rs.out_ax + i]; }
    CPG_DEV double sD(unsigned i) const { return w[(unsigned)Rs.out_ax + i]; }
    CPG_DEV void stage(int which) const {
        const double *src = which == 1 ? (const double *)B.E : (const double *)B.D;
        const unsigned cnt = which == 1 ? (unsigned)F.m : (unsigned)F.n;
        cpgw::lds_order();
        for (unsigned i0 = 0; i0 < cnt; i0 += 1024u) {        // (16 loads in flight: one round trip for up to 1 024 rows)
            double v[16];
#pragma unroll
            for (int u_ = 0; u_ < 16; u_++) { const unsigned i = i0 + 64u * (unsigned)u_ + (unsigned)lane; v[u_] = i < cnt ? cpgw::gld(src, i) : 0.0; }
#pragma unroll
            for (int u_ = 0; u_ < 16; u_++) { const unsigned i = i0 + 64u * (unsigned)u_ + (unsigned)lane; if (i < cnt) w[(unsigned)Rs.out_ax + i] = v[u_]; }
        }
        cpgw::lds_order();
    }
    CPG_DEV void run(const DevStreamTab &T, const double *vals, int k) const {
#ifdef CPG_GENRA_NSTEPS
        CPG_LDS double *wl = cpgw::lds_window3() + w_off;
        if (k == 0) run_rows_a(vals, T.gcols, T.grows, wl, lane);
        else if (k == 1) run_rows_p(vals, T.gcols, T.grows, wl, lane);
        else run_rows_t(vals, T.gcols, T.grows, wl, lane);
        return;
#endif
        StreamProg ST;
        ST.stab = T.stab; ST.cr = T.cr; ST.vals = vals; ST.n_pairs = T.n_pairs; ST.dummy = T.dummy;
        run_program_stream<CPG_RES_PRODUCT_DEPTH>(ST, w, lane);
    }
    CPG_DEV void products(int which) const {        // 1: A w[0..n)   2: P w[0..n)   4: A' w[n..n+m)
        // rows without an entry are never written by their program, and A x shares the slots of P x | A' y: clear first
        // (behind every lane's last read of the previous product).  (As a real call -- one copy of the executor instead of
        // one per call site -- a test took 2.27 instead of 1.69 ms per 20 000 instances: profiles/r4_s4c_*.)
        if (__builtin_expect(skip_products, 0)) return;
        cpgw::lds_order();
        if (which & 1) for (unsigned i = (unsigned)lane; i < (unsigned)F.m; i += 64u) w[(unsigned)Rs.out_ax + i] = 0.0;
        if (which & 2) for (unsigned i = (unsigned)lane; i < (unsigned)F.n; i += 64u) w[(unsigned)Rs.out_px + i] = 0.0;
        if (which & 4) for (unsigned i = (unsigned)lane; i < (unsigned)F.n; i += 64u) w[(unsigned)Rs.out_aty + i] = 0.0;
        cpgw::lds_order();
#pragma nounroll
        for (int k = 0; k < 3; k++)
            if ((which >> k) & 1) run(k == 0 ? Rs.pA : (k == 1 ? Rs.pP : Rs.pAt), k == 0 ? B.cA : (k == 1 ? B.cP : B.cAt), k);
        cpgw::lds_order();
    }
    CPG_DEV double ax(int s) const { const unsigned i = (unsigned)lane + 64u * (unsigned)s; return i < (unsigned)F.m ? w[(unsigned)Rs.out_ax + i] : 0.0; }
    CPG_DEV double px(int s) const { const unsigned i = (unsigned)lane + 64u * (unsigned)s; return i < (unsigned)F.n ? w[(unsigned)Rs.out_px + i] : 0.0; }
    CPG_DEV double atx(int s) const { const unsigned i = (unsigned)lane + 64u * (unsigned)s; return i < (unsigned)F.n ? w[(unsigned)Rs.out_aty + i] : 0.0; }
};

// The pieces of an instance's life that are NOT its ADMM iterations are real function calls (noinline), each with a
// register allocation of its own: inlined into one body, the set-up (~230 live registers of matrix entries), the
// termination test (four copies of the streaming executor) and the factorisation decided where the allocator put the
// ADMM loop's coefficients -- in scratch memory, ~100 reloads per iteration (profiles/r4_s2_isa_*).  The AMDGPU calling
// convention keeps a32 - a255 and half of the VGPRs across a call: the coefficients stay where they are.

template <int NSZ>
struct ResSetupOut {
    double cs, dconst;
    unsigned free_rows;           // bit s: row lane + 64 s is free (infinite bound)
    signed char ct[NSZ];          // row classes for check(): 1 equality, 0 inequality, -1 free
    unsigned long long ts[3];     // (experiments, debug_stage 22: time stamps after canonicalisation, equilibration, scaled data)
};

// steps 1 - 3 of an instance: canonicalise, equilibrate, scaled data (the wavefront's buffer B, program-order copies)
template <int NSX, int NSZ>
CPG_DEV_NOINLINE void resident_setup(const DevRefactor &R_, const DevResident &Rs_, const ResBuf &B_, unsigned sl_off_v, const double *theta_v,
                                     double ri_eq, double ri_in, double ri_fr, int probe_v, ResSetupOut<NSZ> &out) {
    const int lane = cpgw::lane_id();      // (the compiler knows this one's range, and folds the bounds tests of full slots)
    const bool probe = cpgw::read_first_lane(probe_v) != 0;
    const DevRefactor R = uniform_global_copy(R_); const DevResident Rs = uniform_global_copy(Rs_); const ResBuf B = uniform_global_copy(B_);
    const unsigned sl_off = (unsigned)cpgw::read_first_lane((int)sl_off_v);
    const double *theta = nullptr;
    theta = (const double *)(((unsigned long long)(unsigned)cpgw::read_first_lane((int)((unsigned long long)theta_v >> 32)) << 32) |
                                           (unsigned)cpgw::read_first_lane((int)(unsigned long long)theta_v));
    theta = cpgw::as_global(theta);
    constexpr unsigned n = CPG_GENR_N, m = CPG_GENR_M, N = n + m, n_eq = CPG_GENR_NEQ;
    double *sl = cpgw::lds_window() + sl_off;
#ifdef CPG_GENR_TABLES_GLOBAL
    // (four wavefronts per CU: the scaling vectors, the norms and theta use the space the scaled matrices take once D and E are
    // dead -- step 3 below writes A, P behind a fence, after every read of D and E)
    double *Al = sl, *Pl = Al + CPG_GENR_NNZA, *Dl = sl, *El = Dl + n;
#else
    double *Al = sl, *Pl = Al + CPG_GENR_NNZA, *Dl = Pl + CPG_GENR_NNZP, *El = Dl + n;
#endif
    unsigned long long *nrm = (unsigned long long *)(El + m);
    {
        // ---- 1. theta -> LDS; canonicalise P, A, q, u (registers: entry k = lane + 64 t of a matrix, entry i = lane + 64 s
        //         of a vector), d.  The family's dimensions are compile-time constants: every table read below is an
        //         independent, unrolled load -- with one wavefront per SIMD a loop that waits for one global load per
        //         trip is a chain of memory latencies.
        constexpr int KA = (CPG_GENR_NNZA + 63) / 64 > 0 ? (CPG_GENR_NNZA + 63) / 64 : 1, KP = (CPG_GENR_NNZP + 63) / 64 > 0 ? (CPG_GENR_NNZP + 63) / 64 : 1;
        constexpr unsigned nnzA = CPG_GENR_NNZA, nnzP = CPG_GENR_NNZP;
        unsigned ea[KA], ep[KP];           // row | column << 16
        double av[KA], pv[KP];
        double qr[NSX], ur[NSZ];
        {
            double *th = Dl;
            for (unsigned t0 = 0; t0 < (unsigned)R.np_var; t0 += 2048u) {      // (32 loads in flight: one round trip for up to 2 048 parameters)
                double tv[32];
#pragma unroll
                for (int u = 0; u < 32; u++) { const unsigned t = t0 + 64u * (unsigned)u + (unsigned)lane; tv[u] = t < (unsigned)R.np_var ? cpgw::gld(theta, t) : 0.0; }
#pragma unroll
                for (int u = 0; u < 32; u++) { const unsigned t = t0 + 64u * (unsigned)u + (unsigned)lane; if (t < (unsigned)R.np_var) th[t] = tv[u]; }
            }
            cpgw::lds_order();
#pragma unroll
            for (int t = 0; t < KA; t++) { const unsigned k = (unsigned)lane + 64u * (unsigned)t; ea[t] = k < nnzA ? cpgw::gld(Rs.entA, k) : 0u; av[t] = k < nnzA ? cpgw::gld(R.A_base, k) : 0.0; }
#pragma unroll
            for (int t = 0; t < KP; t++) { const unsigned k = (unsigned)lane + 64u * (unsigned)t; ep[t] = k < nnzP ? cpgw::gld(Rs.entP, k) : 0u; pv[t] = k < nnzP ? cpgw::gld(R.P_base, k) : 0.0; }
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; qr[s] = i < n ? cpgw::gld(R.q_base, i) : 0.0; }
#pragma unroll
            for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; ur[s] = i < m ? cpgw::gld(R.u_base, i) : 0.0; }
#pragma nounroll
            for (int j = 0; j < Rs.eA.J; j++) {
#pragma unroll
                for (int t = 0; t < KA; t++) {
                    const unsigned k = (unsigned)lane + 64u * (unsigned)t, e = (unsigned)j * (unsigned)Rs.eA.rows + (k < nnzA ? k : 0u);
                    av[t] = fma(cpgw::gld(Rs.eA.coef, e), th[(unsigned)cpgw::gld(Rs.eA.idx, e)], av[t]);
                }
            }
#pragma nounroll
            for (int j = 0; j < Rs.eP.J; j++) {
#pragma unroll
                for (int t = 0; t < KP; t++) {
                    const unsigned k = (unsigned)lane + 64u * (unsigned)t, e = (unsigned)j * (unsigned)Rs.eP.rows + (k < nnzP ? k : 0u);
                    pv[t] = fma(cpgw::gld(Rs.eP.coef, e), th[(unsigned)cpgw::gld(Rs.eP.idx, e)], pv[t]);
                }
            }
#pragma nounroll
            for (int j = 0; j < Rs.eq.J; j++) {
#pragma unroll
                for (int s = 0; s < NSX; s++) {
                    const unsigned i = (unsigned)lane + 64u * (unsigned)s, e = (unsigned)j * (unsigned)Rs.eq.rows + (i < n ? i : 0u);
                    qr[s] = fma(cpgw::gld(Rs.eq.coef, e), th[(unsigned)cpgw::gld(Rs.eq.idx, e)], qr[s]);
                }
            }
#pragma nounroll
            for (int j = 0; j < Rs.eu.J; j++) {
#pragma unroll
                for (int s = 0; s < NSZ; s++) {
                    const unsigned i = (unsigned)lane + 64u * (unsigned)s, e = (unsigned)j * (unsigned)Rs.eu.rows + (i < m ? i : 0u);
                    ur[s] = fma(cpgw::gld(Rs.eu.coef, e), th[(unsigned)cpgw::gld(Rs.eu.idx, e)], ur[s]);
                }
            }
        }
        const double dconst = csr_row(R.map_d, 0, theta, R.d_base);
        if (probe) out.ts[0] = cpgw::clock100();
        double qsu[NSX];                       // q of the code-generation-time workspace (cost scaling)
#pragma unroll
        for (int s = 0; s < NSX; s++) { const unsigned j = (unsigned)lane + 64u * (unsigned)s; qsu[s] = j < n ? cpgw::gld(R.q_setup, j) : 0.0; }
        cpgw::lds_order();

        // ---- 2. Ruiz equilibration from scratch, cumulative form (D, E in LDS); entry-parallel sweeps: an entry's
        //         scaled magnitude goes to its column's / row's norm through an LDS max-atomic (non-negative doubles
        //         order like their bit patterns), the lane that owns a column / row then reads its norm
        for (unsigned i = (unsigned)lane; i < N; i += 64u) Dl[i] = 1.0;
        double cs = 1.0;
        cpgw::lds_order();
        auto p_norms = [&]() __attribute__((always_inline)) {       // column norms of c D P D (both triangles)
#pragma unroll
            for (int t = 0; t < KP; t++) {
                const unsigned k = (unsigned)lane + 64u * (unsigned)t, i = ep[t] & 0xFFFFu, j = ep[t] >> 16;
                if (k < nnzP) {
                    const double p = pv[t], di = Dl[i], dj = Dl[j];
                    cpgw::lds_max_u64(nrm + j, fabs(cs * dj * p * di));
                    if (i != j) cpgw::lds_max_u64(nrm + i, fabs(cs * di * p * dj));
                }
            }
        };
#pragma nounroll
        for (int it = 0; it < R.scaling_iters; it++) {
            double dn[NSX], en[NSZ];
            for (unsigned i = (unsigned)lane; i < n; i += 64u) nrm[i] = 0ull;
            cpgw::lds_order();
            p_norms();
            double mag[KA];
#pragma unroll
            for (int t = 0; t < KA; t++) { const unsigned r = ea[t] & 0xFFFFu, c = ea[t] >> 16; mag[t] = fabs(El[r] * av[t] * Dl[c]); }
#pragma unroll
            for (int t = 0; t < KA; t++) { const unsigned k = (unsigned)lane + 64u * (unsigned)t; if (k < nnzA) cpgw::lds_max_u64(nrm + (ea[t] >> 16), mag[t]); }
            cpgw::lds_order();
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned j = (unsigned)lane + 64u * (unsigned)s; dn[s] = j < n ? cpgw::u64_as_double(nrm[j]) : 0.0; }
            cpgw::lds_order();
            for (unsigned i = (unsigned)lane; i < m; i += 64u) nrm[i] = 0ull;
            cpgw::lds_order();
#pragma unroll
            for (int t = 0; t < KA; t++) { const unsigned k = (unsigned)lane + 64u * (unsigned)t; if (k < nnzA) cpgw::lds_max_u64(nrm + (ea[t] & 0xFFFFu), mag[t]); }
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
                    qn = cpgw::dmax2(qn, fabs(cs * Dl[j] * qsu[s]));
                }
            }
            psum = cpgw::wave_sum(psum);
            qn = lim_scaling(cpgw::wave_max_nonneg(qn));
            const double cm = n ? psum / (double)n : 0.0;
            cs = cs * (1.0 / lim_scaling(cpgw::dmax2(cm, qn)));
            cpgw::lds_order();
        }
        if (probe) out.ts[1] = cpgw::clock100();
        // ---- 3. scaled data: matrices (LDS for the copies below; the wavefront's buffer: the factorisations read their
        //         KKT values there, the termination tests their program-order copies), scaling vectors, q, u, row classes
#pragma unroll
        for (int t = 0; t < KA; t++) { const unsigned r = ea[t] & 0xFFFFu, c = ea[t] >> 16; av[t] = El[r] * av[t] * Dl[c]; }
#pragma unroll
        for (int t = 0; t < KP; t++) { const unsigned i = ep[t] & 0xFFFFu, j = ep[t] >> 16; pv[t] = cs * Dl[i] * pv[t] * Dl[j]; }
        signed char (&ct)[NSZ] = out.ct;
        unsigned free_rows = 0u;
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
                ct[s] = i < n_eq ? 1 : (uu > CPG_INFTY * CPG_MIN_SCALING ? -1 : 0);
                if (ct[s] == -1) free_rows |= 1u << s;
                cpgw::gst(B.rinv, i, ct[s] == 1 ? ri_eq : (ct[s] == 0 ? ri_in : ri_fr));
            }
        }
        cpgw::lds_order();                 // (D, E are dead: the scaled matrices take the front of the slice)
#pragma unroll
        for (int t = 0; t < KA; t++) { const unsigned k = (unsigned)lane + 64u * (unsigned)t; if (k < nnzA) { Al[k] = av[t]; cpgw::gst(B.A, k, av[t]); } }
#pragma unroll
        for (int t = 0; t < KP; t++) { const unsigned k = (unsigned)lane + 64u * (unsigned)t; if (k < nnzP) { Pl[k] = pv[t]; cpgw::gst(B.P, k, pv[t]); } }
        cpgw::lds_order();
        if (probe) out.ts[2] = cpgw::clock100();
        auto copy_values = [&](const DevStreamTab &T, double *dst, const double *src) __attribute__((always_inline)) {
#pragma nounroll
            for (unsigned e0 = 0; e0 < (unsigned)T.n_entries; e0 += 1536u) {      // (24 source words in flight)
                int kk[24];
#pragma unroll
                for (int u = 0; u < 24; u++) { const unsigned e = e0 + 64u * (unsigned)u + (unsigned)lane; kk[u] = e < (unsigned)T.n_entries ? cpgw::gld(T.src, e) : -1; }
#pragma unroll
                for (int u = 0; u < 24; u++) {
                    const unsigned e = e0 + 64u * (unsigned)u + (unsigned)lane;
                    if (e < (unsigned)T.n_entries) cpgw::gst(dst, e, kk[u] >= 0 ? src[(unsigned)kk[u]] : 0.0);
                }
            }
        };
        copy_values(Rs.pA, B.cA, Al);
        copy_values(Rs.pAt, B.cAt, Al);
        copy_values(Rs.pP, B.cP, Pl);
        cpgw::lds_order();
        cpgw::mem_order();

        out.cs = cs; out.dconst = dconst; out.free_rows = free_rows;
    }
}

// step 4: KKT values into the slice, numeric LDL' + inverses of the merged diagonal blocks
CPG_DEV_NOINLINE void resident_factorise(const DevRefactor &R_, const DevResident &Rs_, const ResBuf &B_, unsigned sl_off_v, double sigma, int) {
    const int lane = cpgw::lane_id();
    const DevRefactor R = uniform_global_copy(R_); const DevResident Rs = uniform_global_copy(Rs_); const ResBuf B = uniform_global_copy(B_);
    const unsigned sl_off = (unsigned)cpgw::read_first_lane((int)sl_off_v);
    double *sl = cpgw::lds_window() + sl_off;
    {
            constexpr unsigned nd = CPG_GENR_NNZL + CPG_GENR_N + CPG_GENR_M;
            constexpr int KD = (int)((nd + 63u) / 64u);
            constexpr int KB = 40;           // (two dependent loads per destination: every batch is two exposed round trips)
            const unsigned lk = (unsigned)cpgw::opaque(lane);
#pragma unroll
            for (int t0 = 0; t0 < KD; t0 += KB) {            // KKT values of the destinations: KB independent sources at a time
                unsigned code[KB];
                double v[KB];
#pragma unroll
                for (int u = 0; u < KB; u++) { const unsigned d = lk + 64u * (unsigned)(t0 + u); code[u] = (t0 + u < KD && d < nd) ? cpgw::gld(Rs.k_src, d) : 0u; }
#pragma unroll
                for (int u = 0; u < KB; u++) {
                    v[u] = cpgw::gld((const double *)B.A, kkt_source_element(B, code[u]));
                }
#pragma unroll
                for (int u = 0; u < KB; u++) v[u] = kkt_source_value(code[u], v[u]);
#pragma unroll
                for (int u = 0; u < KB; u++) {
                    const unsigned d = lk + 64u * (unsigned)(t0 + u), kind = (code[u] >> 28) & 7u;
                    double vv = v[u];
                    if (kind == CPG_K_P) vv = vv + (d >= (unsigned)CPG_GENR_NNZL ? sigma : 0.0);
                    else if (kind == CPG_K_SIGMA) vv = sigma;
                    if (t0 + u < KD && d < nd) sl[d] = (code[u] >> 31) ? 1.0 / vv : vv;
                }
            }
            for (unsigned d = nd + (unsigned)lane; d < (unsigned)Rs.fac_len; d += 64u) sl[d] = d == (unsigned)Rs.fac_len - 2u ? 1.0 : 0.0;
            cpgw::lds_order();
#ifdef CPG_GENR_FAC_NSTEPS
            resident_factor_gen(Rs.gf_tri, Rs.gf_dk, sl, lane);
#else
            resident_factor(Rs, sl, lane);
#endif
    }
}

// step 5: the coefficients of the generated executor to the wavefront's buffer, and the slice back to its ADMM use: idle
// lanes of a step gather the zero slot, idle lanes of a chunk store to the dummy slots (everything starts finite); the
// results of the termination test's products; q and u of the instance
CPG_DEV_NOINLINE void resident_store_coefficients(const DevRefactor &R_, const DevResident &Rs_, const ResBuf &B_, unsigned sl_off_v, int) {
    const int lane = cpgw::lane_id();
    const DevRefactor R = uniform_global_copy(R_); const DevResident Rs = uniform_global_copy(Rs_); const ResBuf B = uniform_global_copy(B_);
    const unsigned sl_off = (unsigned)cpgw::read_first_lane((int)sl_off_v);
    double *sl = cpgw::lds_window() + sl_off;
    constexpr unsigned n = CPG_GENR_N, m = CPG_GENR_M;
    constexpr int ldw = CPG_GENR_NSLOTS + CPG_GEN_EXTRA_SLOTS;
    resident_coefficients(R, Rs, B, sl, lane);
    cpgw::lds_order();
    double *w = sl, *qs = w + ldw, *us = qs + n;
    for (unsigned t = (unsigned)lane; t < (unsigned)Rs.slice_doubles; t += 64u) w[t] = 0.0;
    cpgw::lds_order();
    // q | u and 1 / D | 1 / E of the instance: all loads of both in flight together (the dimensions are literals)
    {
        constexpr int KV = (int)((n + m + 63u) / 64u);
        double vq[KV], vs[KV];
#pragma unroll
        for (int u_ = 0; u_ < KV; u_++) {
            const unsigned i = 64u * (unsigned)u_ + (unsigned)lane;
            vq[u_] = i < n ? cpgw::gld((const double *)B.q, i) : (i < n + m ? cpgw::gld((const double *)B.u, i - n) : 0.0);
            vs[u_] = i < n ? cpgw::gld((const double *)B.Dinv, i) : (i < n + m ? cpgw::gld((const double *)B.Einv, i - n) : 0.0);
        }
        double *sc = w + (unsigned)Rs.out_sc;
#pragma unroll
        for (int u_ = 0; u_ < KV; u_++) {
            const unsigned i = 64u * (unsigned)u_ + (unsigned)lane;
            if (i < n + m) {
                qs[i] = vq[u_];
#ifdef CPG_GENR_TABLES_GLOBAL
                if (i >= n) sc[i - n] = vs[u_];             // (1 / E only: 1 / D is read into registers by the test)
#else
                sc[i] = vs[u_];
#endif
            }
        }
    }
    cpgw::lds_order();
    cpgw::mem_order();
}

// The iterates of an instance between two calls (in memory: what a call takes by reference lives there), the steps of
// the last checked iteration.
template <int NSX, int NSZ>
struct ResState { double x[NSX], z[NSZ], y[NSZ], dx[NSX], dy[NSZ]; };

#ifdef CPG_GENR_TABLES_GLOBAL
#define CPG_RES_TAB                 // the executor's tables in global memory
#else
#define CPG_RES_TAB CPG_LDS         // ... in a block-shared LDS copy
#endif
// One ADMM iteration on the instance's registers.
// CPG_RES_STEP_VALUES 1: the step sizes of a slot as per-lane values selected once per call instead of selected from `rr` in every
// iteration (what removes 18 of the loop's 22 scratch reloads -- the compiler turns the select between members of `rr` into a
// select between their stack addresses and a load).  Measured A/B on one box (profiles/r4_s14_ab_step_values.txt): 121 instead
// of 118 us per 25 iterations, and the OTHER stages 3 - 8 % slower with it too; the reloads are overlapped.  Off.
#ifndef CPG_RES_STEP_VALUES
#define CPG_RES_STEP_VALUES 0
#endif
template <int NSX, int NSZ>
CPG_DEV void resident_step(double (&x)[NSX], double (&z)[NSZ], double (&y)[NSZ], const double (&cf)[CPG_GENR_NREGS],
                           const CPG_RES_TAB unsigned short *lc, const CPG_RES_TAB unsigned short *lr, CPG_LDS double *w, const CPG_LDS double *qs,
                           const CPG_LDS double *us, const ResRho &rr, const double (&riv)[NSZ], const double (&rvv)[NSZ], unsigned free_rows, int lane) {
    constexpr unsigned n = CPG_GENR_N, m = CPG_GENR_M, n_eq = CPG_GENR_NEQ;
    double qt[NSX];
#pragma unroll
    for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; qt[s] = i < n ? qs[i] : 0.0; }
#pragma unroll
    for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < n) w[i] = rr.sigma * x[s] - qt[s]; }
#pragma unroll
    for (int s = 0; s < NSZ; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
#if CPG_RES_STEP_VALUES
        const double ri = riv[s];
#else
        const double ri = i < n_eq ? +rr.ri_eq : (((free_rows >> s) & 1u) ? +rr.ri_fr : +rr.ri_in);
#endif
        if (i < m) w[n + i] = z[s] - ri * y[s];
    }
    cpgw::lds_order();
    run_program_res(cf, lc, lr, w, lane);
#pragma unroll
    for (int s = 0; s < NSX; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        const double xn = i < n ? rr.alpha * w[i] + (1.0 - rr.alpha) * x[s] : 0.0;
        x[s] = xn;
    }
#pragma unroll
    for (int s = 0; s < NSZ; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        const bool eq = i < n_eq;
#if CPG_RES_STEP_VALUES
        const double rv = rvv[s], ri = riv[s];
#else
        const bool fr = (free_rows >> s) & 1u;
        const double rv = eq ? +rr.rho_eq : (fr ? +rr.rho_fr : +rr.rho_in);
        const double ri = eq ? +rr.ri_eq : (fr ? +rr.ri_fr : +rr.ri_in);
#endif
        const double zp = z[s], yp = y[s];
        const double zt = (zp - ri * yp) + ri * (i < m ? w[n + i] : 0.0);
        const double zr = rr.alpha * zt + (1.0 - rr.alpha) * zp;
        const double uu = i < m ? us[i] : 0.0;
        const double zn = eq ? uu : cpgw::dmin2(zr + ri * yp, uu);
        const double dyv = rv * (zr - zn);
        z[s] = i < m ? zn : 0.0; y[s] = i < m ? yp + dyv : 0.0;
    }
    cpgw::lds_order();
}

// (a real call, ~7 us each -- one per termination test: inlined into the kernel the same loop carried 29 scratch loads and
// 23 stores per iteration and the default mode ran 95 instead of 65 ms per 20 000 instances, profiles/r4_s5_*)
#ifndef CPG_RES_ITERATE_LINKAGE
#define CPG_RES_ITERATE_LINKAGE CPG_DEV_NOINLINE
#endif
// `count` ADMM iterations, the last one keeping its steps delta x / delta y for the termination test.  The only function
// that runs the generated executor: x, z, y and the VGPR coefficients are loaded once, nothing in here is a call, and the
// loop holds no scratch access (scripts/isa_resident.py checks it).
template <int NSX, int NSZ>
CPG_RES_ITERATE_LINKAGE void resident_iterate(ResState<NSX, NSZ> &st, const ResRho &rr_, const double *cfg, unsigned free_rows, unsigned sl_off_v,
                                       int count_v, int, const unsigned short *gcols_v = nullptr, const unsigned short *grows_v = nullptr) {
    const int lane = cpgw::lane_id();      // (range known: bounds tests of full slots fold away)
    // (arguments arrive in VGPRs: tell the compiler which of them are wave-uniform)
    const unsigned sl_off = (unsigned)cpgw::read_first_lane((int)sl_off_v);
    const int count = cpgw::read_first_lane(count_v);
    constexpr unsigned n = CPG_GENR_N, m = CPG_GENR_M, n_eq = CPG_GENR_NEQ;
    constexpr int ldw = CPG_GENR_NSLOTS + CPG_GEN_EXTRA_SLOTS;
    constexpr unsigned t_ncols = ((CPG_GENR_NSTEPS + 3u) / 4u) * 256u;
#ifdef CPG_GENR_TABLES_GLOBAL
    auto uniform_ptr = [](const unsigned short *p) __attribute__((always_inline)) {
        const unsigned long long a = (unsigned long long)p;
        return cpgw::as_global((const unsigned short *)(((unsigned long long)(unsigned)cpgw::read_first_lane((int)(a >> 32)) << 32) | (unsigned)cpgw::read_first_lane((int)a)));
    };
    const unsigned short *lc = uniform_ptr(gcols_v), *lr = uniform_ptr(grows_v);
    (void)t_ncols;
#else
    const CPG_LDS unsigned short *lc = (const CPG_LDS unsigned short *)cpgw::lds_window3(), *lr = lc + t_ncols;
#endif
    CPG_LDS double *w = cpgw::lds_window3() + sl_off;
    const CPG_LDS double *qs = w + ldw, *us = qs + n;
    const ResRho rr = uniform_copy(rr_);
    // (round 6: the members pinned in scalar registers as values -- cpgw::sgpr_value, what the team kernel does -- take the loop's
    // scratch loads from 22 to 4 and make it SLOWER here, 51.5 against 50.8 ms per 20 000 instances in an A/B on one box
    // (profiles/r6_t6_*): 30 more coefficient reads through v_accvgpr_read; the reloads are overlapped, as round 4 measured)
    // (step sizes per slot as per-lane values: only read with CPG_RES_STEP_VALUES, see resident_step)
    double riv[NSZ], rvv[NSZ];
#pragma unroll
    for (int s = 0; s < NSZ; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        const bool eq = i < n_eq, fr = (free_rows >> s) & 1u;
        riv[s] = eq ? +rr.ri_eq : (fr ? +rr.ri_fr : +rr.ri_in);
        rvv[s] = eq ? +rr.rho_eq : (fr ? +rr.rho_fr : +rr.rho_in);
    }
    // The coefficients: ~2 registers per step and lane, loaded once per call (one call = the iterations between two
    // termination tests; 74 KB per call on the portfolio family) and held in the wavefront's 512 registers -- which of them
    // are VGPRs and which AGPR copies read through v_accvgpr_read at their use is the compiler's business, in a function
    // that contains nothing but this loop (hand-named AGPRs were tried and dropped: DESIGN.md 4.6)
    double cf[CPG_GENR_NREGS];
#pragma unroll
    for (int t = 0; t < CPG_GENR_NREGS; t++) cf[t] = cpgw::gld(cpgw::as_global(cfg), (unsigned)t * 64u + (unsigned)lane);
    double x[NSX], z[NSZ], y[NSZ];
#pragma unroll
    for (int s = 0; s < NSX; s++) x[s] = st.x[s];
#pragma unroll
    for (int s = 0; s < NSZ; s++) { z[s] = st.z[s]; y[s] = st.y[s]; }
    // count - 1 iterations, then the checked one: ONE copy of the iteration's code (a second copy, or store blocks inside
    // the loop, cost the loop's register allocation: 22 scratch reloads per iteration); the steps of the checked iteration
    // are delta x = x(k+1) - x(k), delta y = y(k+1) - y(k), from the iterates saved in front of it
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {
        const int nk = pass == 0 ? count - 1 : (count > 0 ? 1 : 0);
#pragma nounroll
        for (int k = 0; k < nk; k++) resident_step<NSX, NSZ>(x, z, y, cf, lc, lr, w, qs, us, rr, riv, rvv, free_rows, lane);
        if (pass == 0) {
#pragma unroll
            for (int s = 0; s < NSX; s++) st.dx[s] = x[s];
#pragma unroll
            for (int s = 0; s < NSZ; s++) st.dy[s] = y[s];
        }
    }
    if (count > 0) {
#pragma unroll
        for (int s = 0; s < NSX; s++) st.dx[s] = x[s] - st.dx[s];
#pragma unroll
        for (int s = 0; s < NSZ; s++) st.dy[s] = y[s] - st.dy[s];
    }
#pragma unroll
    for (int s = 0; s < NSX; s++) st.x[s] = x[s];
#pragma unroll
    for (int s = 0; s < NSZ; s++) { st.z[s] = z[s]; st.y[s] = y[s]; }
}

template <int NSX, int NSZ>
CPG_DEV_NOINLINE CheckOut resident_check(const DevFamily &F_, const DevResident &Rs_, const ResBuf &B_, const signed char (&ct_)[NSZ],
                                                const DevSettings &S_, const double (&x_)[NSX], const double (&z_)[NSZ], const double (&y_)[NSZ],
                                                const double (&dxr_)[NSX], const double (&dyr_)[NSZ], unsigned sl_off_v, int,
                                                bool approximate_v, ScaledNorms *sn_) {
    typedef ResidentCtx<NSX, NSZ> CtxT;
    const int lane = cpgw::lane_id();
    const unsigned sl_off = (unsigned)cpgw::read_first_lane((int)sl_off_v);
    const bool approximate = cpgw::read_first_lane(approximate_v ? 1 : 0) != 0;
    // Everything that arrives by reference is copied into locals once, with all loads in flight together: read where it is
    // used, every entry was a flat load from the caller's stack in front of its use -- ~100 memory round trips per test
    // with nobody to hide them (one wavefront per SIMD): 96 us per test (profiles/r4_s3_*).
    const DevFamily F = uniform_global_copy(F_); const DevResident Rs = uniform_global_copy(Rs_); const ResBuf B = uniform_global_copy(B_); const DevSettings S = uniform_copy(S_);
    double x[NSX], z[NSZ], y[NSZ], dxr[NSX], dyr[NSZ];
    signed char ct[NSZ];
#pragma unroll
    for (int s = 0; s < NSX; s++) { x[s] = x_[s]; dxr[s] = dxr_[s]; }
#pragma unroll
    for (int s = 0; s < NSZ; s++) { z[s] = z_[s]; y[s] = y_[s]; dyr[s] = dyr_[s]; ct[s] = ct_[s]; }
    ScaledNorms sn_local;
    ScaledNorms *sn = &sn_local;
    double *w = cpgw::lds_window() + sl_off;
    const double *qs = w + (CPG_GENR_NSLOTS + CPG_GEN_EXTRA_SLOTS), *us = qs + CPG_GENR_N;
#ifdef CPG_GENR_TABLES_GLOBAL
    double dinv_r[NSX];
#pragma unroll
    for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; dinv_r[s] = i < (unsigned)CPG_GENR_N ? cpgw::gld((const double *)B.Dinv, i) : 0.0; }
    const CtxT cx{F, Rs, B, w, sl_off, qs, us, lane, S.debug_stage == 21, dinv_r};
#else
    const CtxT cx{F, Rs, B, w, sl_off, qs, us, lane, S.debug_stage == 21};
#endif
    const CheckOut o = check<NSX, NSZ, CtxT, RegDelta<NSX>, RegDelta<NSZ>>(F, cx, ct, S, x, z, y, RegDelta<NSX>{dxr}, RegDelta<NSZ>{dyr},
                                                                           InfeasVerdict{false, false}, w, lane, approximate, sn);
    if (sn_) *sn_ = sn_local;
    return o;
}
template <int NSX, int NSZ>
CPG_DEV_NOINLINE void resident_finalize(const DevFamily &F_, const DevBatch &Bt_, const double (&x_)[NSX], const double (&z_)[NSZ],
                                               const double (&y_)[NSZ], double dconst, long long b_v, unsigned sl_off_v, int, int iter,
                                               const CheckOut &o_, double rho) {
    const int lane = cpgw::lane_id();
    const unsigned sl_off = (unsigned)cpgw::read_first_lane((int)sl_off_v);
    const long long b = ((long long)cpgw::read_first_lane((int)(b_v >> 32)) << 32) | (unsigned)cpgw::read_first_lane((int)b_v);
    const DevFamily F = uniform_global_copy(F_); const DevBatch Bt = uniform_global_copy(Bt_); const CheckOut o = o_;      // (see resident_check)
    double x[NSX], z[NSZ], y[NSZ];
#pragma unroll
    for (int s = 0; s < NSX; s++) x[s] = x_[s];
#pragma unroll
    for (int s = 0; s < NSZ; s++) { z[s] = z_[s]; y[s] = y_[s]; }
    double *w = cpgw::lds_window() + sl_off;
    finalize<NSX, NSZ, true>(F, Bt, x, z, y, dconst, b, w, lane, iter, o, rho);
}

template <int NSX, int NSZ>
CPG_DEV void osqp_resident_body(const DevFamily &F0, const DevRefactor &R, const DevResident &Rs, const DevSettings &S,
                                const DevBatch &Bt, double *lds, int wave_global) {
    const int lane = cpgw::lane_id();
    // the family's dimensions are compile-time constants of its library (cpg_hip_set_resident checks them): bounds tests
    // of full 64-entry slots fold away -- as run-time values they cost the ADMM loop ~130 branches and ~290 exec-mask
    // reloads (v_readlane of spilled SGPR pairs) per iteration
    constexpr unsigned n = CPG_GENR_N, m = CPG_GENR_M, N = n + m, n_eq = CPG_GENR_NEQ;
    constexpr int ldw = CPG_GENR_NSLOTS + CPG_GEN_EXTRA_SLOTS;
    // block-shared copies of the executor's offset / output-slot tables in front of the wavefronts' slices
    constexpr unsigned t_ncols = ((CPG_GENR_NSTEPS + 3u) / 4u) * 256u, t_nrows = ((CPG_GENR_NCHUNKS + 3u) / 4u) * 256u;
#ifdef CPG_GENR_TABLES_GLOBAL
    constexpr unsigned tab_doubles = 0u;             // (four wavefronts per CU: the tables are read from global memory)
    (void)t_ncols; (void)t_nrows; (void)lds;
#else
    constexpr unsigned tab_doubles = (t_ncols + t_nrows) / 4u;
    unsigned short *lc = (unsigned short *)lds, *lr = lc + t_ncols;
    for (unsigned t = cpgw::thread_in_block(); t < t_ncols; t += cpgw::block_threads()) lc[t] = cpgw::gld(Rs.g_cols, t);
    for (unsigned t = cpgw::thread_in_block(); t < t_nrows; t += cpgw::block_threads()) lr[t] = cpgw::gld(Rs.g_rows, t);
    cpgw::block_sync();
    lds += tab_doubles;
#endif
    // The wavefront's slice, three lives:
    //   set-up    A (nnzA) | P (nnzP) | D (n) | E (m) | norms (max(n, m))      theta is staged where D starts
    //   factor    fac = M (nnzL) | 1/d (N) | X | 1.0 | 0.0
    //   ADMM      w (ldw) | q (n) | u (m) | A x (m)  or  P x (n) | A' y (n) | 1 / D (n) | 1 / E (m)
    const unsigned sl_off = tab_doubles + (unsigned)cpgw::wave_in_block() * (unsigned)Rs.slice_doubles;
    const ResBuf B = res_carve(Bt.scratch + (size_t)wave_global * (size_t)Rs.buf_doubles, F0, R, Rs, CPG_GENR_NREGS);
    const double rho_fr = CPG_RHO_MIN, ri_fr = 1.0 / rho_fr;
    const size_t state_len = (size_t)n + 2u * (size_t)m + 1u;
    const unsigned n_work = Bt.list_count ? cpgw::sld(Bt.list_count, 0u) : 0u;

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

        // (experiments, debug_stage 20: the 100 MHz time stamps of the instance's stages replace its primal results)
        const bool probe = __builtin_expect(S.debug_stage >= 20 && S.debug_stage <= 22, 0);
        unsigned long long ts[8];
        int n_ts = 0;
#define CPG_RES_PROBE() do { if (probe && n_ts < 8) ts[n_ts++] = cpgw::clock100(); } while (0)
        CPG_RES_PROBE();
        ResSetupOut<NSZ> su;
        resident_setup<NSX, NSZ>(R, Rs, B, sl_off, theta, ri_eq, ri_in, ri_fr, S.debug_stage == 22 ? 1 : 0, su);
        const double cs = su.cs, dconst = su.dconst;
        if (__builtin_expect(S.debug_stage == 22, 0)) { for (int k = 0; k < 3 && n_ts < 8; k++) ts[n_ts++] = su.ts[k]; }
        // (experiments: leave the instance after stage k of its life: 1 set-up, 2 factorisation, 3 coefficients, 4 first iterations)
#define CPG_RES_STOP_AFTER(k) if (__builtin_expect(S.debug_stage == (k), 0)) { if (lane == 0) { Bt.status[b] = 11; Bt.iter[b] = 0; } continue; }
        CPG_RES_STOP_AFTER(1)
        CPG_RES_PROBE();

        // ---- 4. - 6.  factorise (call), iterate to the next event (call), test / adapt (call), in osqp_solve's order.  The
        //      instance's state between the calls lives in memory (st); the AGPR-held coefficients survive them by name.
        DevFamily F = F0;
        F.D = B.D; F.Dinv = B.Dinv; F.E = B.E; F.Einv = B.Einv; F.c = cs; F.cinv = 1.0 / cs;
        ResState<NSX, NSZ> st;
#pragma unroll
        for (int s = 0; s < NSX; s++) { st.x[s] = 0.0; st.dx[s] = 0.0; }
#pragma unroll
        for (int s = 0; s < NSZ; s++) { st.z[s] = 0.0; st.y[s] = 0.0; st.dy[s] = 0.0; }
        if (state_in) load_state<NSX, NSZ>(F, state_in, st.x, st.z, st.y, lane);
        CheckOut o;
        o.prim_res = 0; o.dual_res = 0; o.obj = 0; o.status = 11;
        int iter = Bt.resume ? cpgw::read_first_lane(cpgw::gld((const int *)Bt.iter, (unsigned)b)) : 0;
        if (iter > 0) rho_stg = rho;
        bool need_factor = true;
        const int chk_int = S.check_termination, ad_int = S.adaptive_rho ? S.adaptive_rho_interval : 0;
#pragma nounroll
        while (o.status == 11) {
            if (need_factor) {
                resident_factorise(R, Rs, B, sl_off, F0.sigma, lane);
                CPG_RES_PROBE();
                if (__builtin_expect(S.debug_stage == 2, 0)) break;
                resident_store_coefficients(R, Rs, B, sl_off, lane);
                CPG_RES_PROBE();
                if (__builtin_expect(S.debug_stage == 3, 0)) break;
                need_factor = false;
            }
            if (iter < S.max_iter) {
                int next_ev = S.max_iter;
                if (chk_int > 0) { const int c = (iter / chk_int + 1) * chk_int; if (c < next_ev) next_ev = c; }
                if (ad_int > 0) { const int c = (iter / ad_int + 1) * ad_int; if (c < next_ev) next_ev = c; }
                const ResRho rr{rho_eq, rho_in, rho_fr, ri_eq, ri_in, ri_fr, F0.sigma, F0.alpha};
                if (__builtin_expect(S.debug_stage == 7, 0)) {        // (experiments: one call per iteration)
                    for (int k = iter; k < next_ev; k++) resident_iterate<NSX, NSZ>(st, rr, B.cf, su.free_rows, sl_off, 1, lane, Rs.g_cols, Rs.g_rows);
                } else
                resident_iterate<NSX, NSZ>(st, rr, B.cf, su.free_rows, sl_off, next_ev - iter, lane, Rs.g_cols, Rs.g_rows);
                iter = next_ev;
                CPG_RES_PROBE();
                if (__builtin_expect(S.debug_stage == 4, 0)) break;
            }
            const bool can_check = chk_int > 0 && iter > 0 && iter % chk_int == 0;
            const bool adapt = ad_int > 0 && iter > 0 && iter % ad_int == 0;
            const bool last = iter >= S.max_iter;
            ScaledNorms sn;
            bool approx = false;
            if (__builtin_expect(S.debug_stage == 9 && !last, 0)) continue;      // (experiments: no test before max_iter)
            for (;;) {
                const CheckOut oc = resident_check<NSX, NSZ>(F, Rs, B, su.ct, S, st.x, st.z, st.y, st.dx, st.dy, sl_off, lane, approx, &sn);
                CPG_RES_PROBE();
                if (approx) { o = oc; break; }
                if (can_check) { o = oc; if (o.status != 11) break; }
                if (adapt) {
                    const double rn = rho_estimate(sn, rho_stg);
                    if (rn > rho_stg * S.adaptive_rho_tolerance || rn < rho_stg / S.adaptive_rho_tolerance) {
                        rho = rn; rho_stg = rn; rho_eq = 1e3 * rho; rho_in = rho; ri_eq = 1.0 / rho_eq; ri_in = 1.0 / rho_in;
#pragma unroll
                        for (int s = 0; s < NSZ; s++) {
                            const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                            if (i < m) cpgw::gst(B.rinv, i, su.ct[s] == 1 ? ri_eq : (su.ct[s] == 0 ? ri_in : ri_fr));
                        }
                        cpgw::mem_order();
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
            if (__builtin_expect(S.debug_stage == 5, 0)) break;
        }
        if (__builtin_expect(S.debug_stage >= 2 && S.debug_stage <= 5, 0)) { if (lane == 0) { Bt.status[b] = 11; Bt.iter[b] = iter; } continue; }
        resident_finalize<NSX, NSZ>(F, Bt, st.x, st.z, st.y, dconst, b, sl_off, lane, iter, o, rho);
        if (probe) {
            CPG_RES_PROBE();
            if (lane == 0) for (int k = 0; k < n_ts && k < F0.n_prim; k++) Bt.prim[(size_t)b * F0.n_prim + k] = (double)(ts[k] - ts[0]);
        }
    }
}
#endif  // CPG_GENR_HEADER

}  // namespace cpg
