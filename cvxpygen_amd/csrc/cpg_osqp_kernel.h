// Batched embedded-OSQP ADMM kernel for gfx950: one wavefront solves G problem instances of ONE
// problem family from cold start to termination without leaving the CU.
//
// Replaces, per instance, the reference's generated cpg_solve() (cvxpygen/utils.py:1008-1052):
//   cpg_canonicalize_q/l/u/d (utils.py:279-294)         -> canonicalise()
//   osqp_update_data_vec     (solvers/osqp.py:39-59)    -> folded: maps are pre-scaled by D, E, c
//   osqp_solve               (solvers/osqp.py:62)       -> the iteration loop below
//   cpg_retrieve_prim/dual/info (utils.py:950-985)      -> finalize()
//
// Data placement:
//   registers : iterates x, z, y -- element i lives on lane i % 64, slot i / 64
//               (NSX = ceil(n/64), NSZ = ceil(m/64) slots, compile-time); the instance's own
//               q, l, u only for the first NV slots: the device ordering places every entry that
//               depends on a user parameter first, all other entries are read from the family's
//               shared base vectors (cache resident)
//   LDS       : one work vector w[n_slots] per instance: right-hand side / solution of the KKT
//               system and staging area for the sparse products of the termination test
//   L2 / HBM  : the family's solve program (shared, read-only, streamed coalesced), theta in,
//               solutions out, delta_x / delta_y stash written at check iterations only
#pragma once

#include <type_traits>
#include "cpg_wave.h"

namespace cpg {

#define CPG_INFTY 1e30
#define CPG_MIN_SCALING 1e-4
#define CPG_DIV_TOL 1e-30
#define CPG_RHO_MIN 1e-6
#define CPG_RHO_MAX 1e6
#define CPG_NO_ROW 0xFFFF
#ifndef CPG_ILP
#define CPG_ILP 2   // unrolled slot iterations the scheduler may interleave in the hot loops
#endif
#define CPG_FENCE_EVERY(s) do { if (((s) + 1) % CPG_ILP == 0) cpgw::sched_fence(); } while (0)
#ifndef CPG_LDS_UNROLL
#define CPG_LDS_UNROLL 4
#endif
#ifndef CPG_CHUNK_UNROLL
#define CPG_CHUNK_UNROLL 4
#endif

struct DevProgram {
    const int *hdr;
    const unsigned short *rows;
    const double *vals;
    const unsigned short *cols;
    int n_chunks;
};
// compact program (solve_program.RaggedProgram); staged into LDS once per block
struct DevRagged {
    const int *ctab;            // [n_chunks][4]: max len, log2 g, first entry, 0
    const unsigned *desc;       // [n_chunks][64]: out slot | len << 16
    const double *vals;         // [nnz]
    const unsigned short *cols; // [nnz]
    int n_chunks, nnz;
    // generated executor with CPG_GEN_PADDED_OFFSETS: operand offsets of all 64 lanes of every step, in
    // execution order (idle lanes: the zero slot); built by cpg_hip_create_osqp from the header's step table
    const unsigned short *cols_padded;
    // generated executor: output slot | segment mask << 13 of every (chunk, lane), four chunks of a lane side by
    // side ([chunk / 4][lane][chunk % 4]); lanes without a row get a dummy slot whose bank pair no row of their
    // 16-lane store group uses (cpg_hip_create_osqp builds it)
    const unsigned short *rows_gen;
    // dictionary-compressed form (family libraries built with CPG_GEN_COMPRESSED): the distinct
    // coefficients and, per entry, operand byte offset | dictionary byte offset << 16
    const double *dict;
    const unsigned *words;
    int n_dict;
};
struct DevCsr {
    const int *ptr;
    const int *idx;
    const double *val;
    int nnz;
};
#ifndef CPG_STREAM_DEPTH
#define CPG_STREAM_DEPTH 8
#endif
struct StreamProg {
    const unsigned *stab;
    const unsigned *cr;
    const double *vals;
    int n_pairs;                  // multiple of CPG_STREAM_DEPTH / 2
    unsigned dummy;               // pair index of the trailing zero pair
};
struct DevFamily {
    int n, m, n_eq, is_max, n_slots;
    double sigma, alpha, rho;
    const double *D, *Dinv, *E, *Einv;
    double c, cinv;
    const signed char *ctype;
    const unsigned short *fpos;   // [n + m] LDS slot holding entry i after the KKT program
    DevProgram kkt, A_rows, P_rows, At_rows;
    DevRagged kkt_ragged;
    StreamProg kkt_stream;        // the same program in the layout of run_program_stream (n_pairs == 0: none)
    int n_prim, n_dual;
    const int *prim_idx, *dual_idx;
    const int *ord;               // [n + m] canonical index of the entry at device position i (state I/O); null: identity
};
struct DevUpdate {
    int np_var;
    // l is implied by the row class: equality rows have l = u, inequality rows l = -inf
    // (the only two kinds of rows the reference's OSQP canonical form contains,
    // cvxpygen/solvers/_interface.py:62-79); the binding verifies this when it builds the plan.
    const double *q_base, *u_base;
    double d_base;
    DevCsr map_q, map_u, map_d;
};
struct DevSettings {
    int max_iter, check_termination, scaled_termination;
    double eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
    // OSQP library defaults that `osqp_set_default_settings(solver.settings)` restores on every cpg_solve of the
    // reference (cvxpygen/solvers/osqp.py:100-101, utils.py:1071-1073) and the generated shim offers no setter for:
    // OSQP >= 1.0 (what the reference's API requires, pyproject.toml:26): adaptive_rho 1, interval 50, tolerance 5,
    // check_dualgap 1.  cpg_hip_set_default_settings restores the same; cpg_hip_set_build_option overrides.
    int check_dualgap;            // duality-gap term of OSQP >= 1.0's termination test
    int adaptive_rho, adaptive_rho_interval;   // rho adaptation every `interval` iterations
    double adaptive_rho_tolerance;
    int warm_starting;            // 1: start from DevBatch::state_in when it is given
    int debug_stage;              // measurements only (0 = off): the per-instance factor kernel leaves an instance after stage k of its set-up
};
struct DevBatch {
    long long B;
    const double *theta;
    double *prim, *dual, *obj, *pri_res, *dua_res;
    int *iter, *status;
    unsigned *counter;
    double *scratch;   // per-wavefront buffers of the refactorisation / adjoint kernels
    // Sequential use (the reference's static workspace keeps its iterates between solves, warm_starting = 1):
    // [B][n + 2 m + 1] = scaled iterates x | z | y in CANONICAL order, then rho.  Null: cold start / not wanted.
    const double *state_in;
    double *state_out;
    // ---- hybrid execution of rho adaptation on a shared-factor batch ----
    // The shared-factor kernel serves an instance until its rho changes (OSQP's adapt_rho gives it a KKT
    // matrix of its own); it then writes the instance's workspace to ho_state (layout of state_out, rho = the
    // new value), its iteration count to iter[], and appends its number to ho_list.  A per-instance factor
    // kernel launched behind it on the same stream continues those instances (`list`, `resume`).
    int *ho_list;
    unsigned *ho_count;
    double *ho_state;
    const int *list;              // instances to process (null: 0 .. B - 1) ...
    const unsigned *list_count;   // ... and how many (device memory, written by the kernel in front)
    int resume;                   // 1: continue at iteration iter[b] from state_in[b] (x | z | y | rho), whatever warm_starting says
};

template <int A, int B> struct MinI { static const int v = A < B ? A : B; };

// ------------------------------------------------------------------------------------ executor
// One chunk: `len` multiply-add steps per lane, then the sum over groups of 2^lg lanes.
template <int G, bool REDUCE>
CPG_DEV void do_chunk(const DevProgram &P, int c, const double *w, int ldw, int lane,
                      double (&out)[G]) {
    const int len = cpgw::read_first_lane(cpgw::gld(P.hdr, 4u * (unsigned)c + 0u));
    const int off = cpgw::read_first_lane(cpgw::gld(P.hdr, 4u * (unsigned)c + 3u));
    const unsigned e0 = (unsigned)off * 64u + (unsigned)lane;
    double acc[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc[g] = 0.0;
#pragma unroll CPG_CHUNK_UNROLL
    for (int s = 0; s < len; s++) {
        const double v = cpgw::gld(P.vals, e0 + 64u * (unsigned)s);
        const unsigned ci = cpgw::gld(P.cols, e0 + 64u * (unsigned)s);
#pragma unroll
        for (int g = 0; g < G; g++) acc[g] = fma(v, w[(unsigned)(g * ldw) + ci], acc[g]);
    }
    if (REDUCE) {
        const int lg = cpgw::read_first_lane(cpgw::gld(P.hdr, 4u * (unsigned)c + 1u));
#pragma unroll
        for (int g = 0; g < G; g++) out[g] = cpgw::group_sum_first_dyn(acc[g], lg);
    } else {
#pragma unroll
        for (int g = 0; g < G; g++) out[g] = acc[g];
    }
}

// w <- program(w).  Results are stored right away: the host-side slot allocation guarantees that a
// chunk never overwrites a slot that a later chunk of the same phase still reads.
template <int G>
CPG_DEV void run_program(const DevProgram &P, double *w, int ldw, int lane) {
#pragma nounroll
    for (int c = 0; c < P.n_chunks; c++) {
        double r[G];
        do_chunk<G, true>(P, c, w, ldw, lane, r);
        const unsigned row = cpgw::gld(P.rows, (unsigned)c * 64u + (unsigned)lane);
        cpgw::lds_order();
        if (row != CPG_NO_ROW) {
#pragma unroll
            for (int g = 0; g < G; g++) w[(unsigned)(g * ldw) + row] = r[g];
        }
        cpgw::lds_order();
    }
}

// run_program_lds<1> for coefficients that are not LDS resident: the per-instance substitution values
// of the refactorisation / adjoint kernels (HBM: 8 bytes per entry and iteration, far more than the
// caches hold across the resident waves) and shared programs too large for the LDS (L2).  What bounds this executor is the number of vector memory instructions (each
// occupies the CU's address unit for ~16 cycles) and how many of them a wave keeps in flight, so:
//   * the walk over (chunk, step) is flattened into one stream and consecutive steps are PAIRED: a
//     lane's two entries are adjacent, one 16-byte load brings both coefficients and one 8-byte load
//     both entry words -- one vector load per step instead of two;
//   * the operands of pair p + DP (DP = CPG_STREAM_DEPTH / 2) are requested when pair p is consumed
//     -- they do not depend on the work vector -- so a wave keeps CPG_STREAM_DEPTH steps in flight
//     across chunk boundaries; no load sits under a branch and all of them are issued from the loop
//     body in one fixed order, so every wait names exactly the loads issued after its operands;
//   * the per-step control word comes through the scalar cache (s_load), one block of pairs ahead.
// Tables (cpg_hip_set_refactor builds them):
//   stab[2p]   entry-pair base | lanes << 18 | control of the first step << 25;  stab[2p+1] control
//              of the second step.  Control: reduction stages | segmented (balanced) chunk << 3 |
//              first step of its chunk << 4 | last step << 5 | rows accumulate into their slot << 6
//              (forward sweep: w[r] += -sum L_rk w[k], no unit-diagonal entry to stream)
//   cr[e]      per entry: byte offset of the operand in the work vector | output row << 16 | segment
//              mask << 29 (row and mask are picked up at the first step of a chunk, where every lane
//              that writes or belongs to a multi-lane row is active; no row = 0x1FFF)
//   vals[e]    per instance, entry e = 2 * (pair base + lane) + step of the pair
// The pair table ends with 2 DP empty pairs and the pair count is a multiple of DP.  Same accumulation
// order as run_program_lds<1>; idle lanes read the trailing zero pair.
struct StreamPairD { double a, b; };
struct StreamPairU { unsigned a, b; };
struct alignas(8) OffsetQuad { unsigned a, b; };     // operand offsets of four steps (generated executor)
CPG_DEV OffsetQuad quad_from(unsigned long long v) { OffsetQuad q; q.a = (unsigned)v; q.b = (unsigned)(v >> 32); return q; }
CPG_DEV void stream_chunk_end(unsigned f, unsigned rowmask, double &acc, double *w) {
    const int stages = (int)(f & 7u);
    // segmented chunks: all three stages, branch-free (the mask of an unused stage is zero)
    const double r = (f & 8u) ? cpgw::seg_sum_first<3>(acc, rowmask >> 13) : cpgw::group_sum_first_dyn(acc, stages);
    cpgw::lds_order();
    if ((rowmask & 0x1FFFu) != 0x1FFFu) w[rowmask & 0x1FFFu] = (f & 64u) ? w[rowmask & 0x1FFFu] + r : r;
    cpgw::lds_order();
    acc = 0.0;
}
template <int DEPTH = CPG_STREAM_DEPTH>          // (tables built for a depth serve every smaller power of two)
CPG_DEV void run_program_stream(const StreamProg &P, double *w, int lane) {
    constexpr int DP = DEPTH / 2;
    const char *wb = (const char *)w;
    // The ring starts with DP empty pairs (zero coefficients) and the loop runs DP pairs past the end
    // of the stream: see above (one fixed order of loads).
    StreamPairD v[DP];
    StreamPairU cr[DP];
    unsigned fa[DP], fb[DP], na[DP], nb[DP];
#pragma unroll
    for (int u = 0; u < DP; u++) {
        v[u].a = 0.0; v[u].b = 0.0; cr[u].a = 0x1FFF0000u; cr[u].b = 0x1FFF0000u; fa[u] = 0u; fb[u] = 0u;
        na[u] = cpgw::sld(P.stab, 2u * (unsigned)u); nb[u] = cpgw::sld(P.stab, 2u * (unsigned)u + 1u);
    }
    unsigned row = 0x1FFFu;                                     // output row | segment mask << 13
    double acc = 0.0;
    double wv = *(const double *)wb;                            // operand of the step about to be consumed
#pragma nounroll
    for (int p0 = 0; p0 < P.n_pairs + DP; p0 += DP) {
        unsigned ca[DP], cb[DP];
#pragma unroll
        for (int u = 0; u < DP; u++) {
            ca[u] = na[u]; cb[u] = nb[u];
            na[u] = cpgw::sld(P.stab, 2u * (unsigned)(p0 + DP + u)); nb[u] = cpgw::sld(P.stab, 2u * (unsigned)(p0 + DP + u) + 1u);
        }
#pragma unroll
        for (int u = 0; u < DP; u++) {
            const unsigned f0 = fa[u] >> 25, f1 = fb[u];
            acc = fma(v[u].a, wv, acc);
            if (f0 & 16u) row = cr[u].a >> 16;
            if (f0 & 32u) stream_chunk_end(f0, row, acc, w);
            wv = *(const double *)(wb + (cr[u].b & 0xFFFFu));   // gathers come after the store of a chunk end
            cpgw::sched_fence();
            acc = fma(v[u].b, wv, acc);
            if (f1 & 16u) row = cr[u].b >> 16;
            if (f1 & 32u) stream_chunk_end(f1, row, acc, w);
            wv = *(const double *)(wb + (cr[(u + 1) % DP].a & 0xFFFFu));
            cpgw::sched_fence();                                // ... and before the requests below, not next to their use
            const unsigned st = ca[u];
            const unsigned e = (unsigned)lane < ((st >> 18) & 0x7Fu) ? (st & 0x3FFFFu) + (unsigned)lane : P.dummy;
            v[u] = cpgw::gld((const StreamPairD *)P.vals, e);
            cr[u] = cpgw::gld((const StreamPairU *)P.cr, e);
            fa[u] = st; fb[u] = cb[u];
        }
    }
}

// Same, with the program resident in LDS in its compact form (solve_program.RaggedProgram): lanes
// of a chunk are ordered by non-increasing entry count, so at step s the lanes that still have an
// entry are a prefix and lane t's entry sits at  first + (entries of earlier steps) + t.
struct LdsProg {
    const int *ctab;
    const unsigned *desc;
    const double *vals;
    const unsigned short *cols;   // BYTE offsets into the work vector
    int n_chunks;
    unsigned dummy;               // index of the trailing zero entry
    const unsigned short *rows16; // generated executor: output slot per (chunk, lane)
};
template <int G>
CPG_DEV void run_program_lds(const LdsProg &P, double *w, int ldw, int lane) {
    const char *wb = (const char *)w;
    const unsigned ldwb = (unsigned)ldw * 8u;
    // descriptors of chunk c + 1 are fetched while chunk c runs
    int t0 = P.ctab[0], t1 = P.ctab[1], t2 = P.ctab[2], t3 = P.ctab[3];
    unsigned dn = P.desc[(unsigned)lane];
#pragma nounroll
    for (int c = 0; c < P.n_chunks; c++) {
        const int L = cpgw::read_first_lane(t0);
        const int lg = cpgw::read_first_lane(t1);
        unsigned base = (unsigned)cpgw::read_first_lane(t2);
        const int balanced = cpgw::read_first_lane(t3);      // rows over a variable number of lanes
        const unsigned d = dn;
        if (c + 1 < P.n_chunks) {
            t0 = P.ctab[4 * c + 4]; t1 = P.ctab[4 * c + 5]; t2 = P.ctab[4 * c + 6]; t3 = P.ctab[4 * c + 7];
            dn = P.desc[(unsigned)(c + 1) * 64u + (unsigned)lane];
        }
        const unsigned row = d & 0xFFFFu;
        const int len = balanced ? (int)((d >> 16) & 0xFFFu) : (int)(d >> 16);
        double acc[G];
#pragma unroll
        for (int g = 0; g < G; g++) acc[g] = 0.0;
        // lanes without an entry at step s read the trailing dummy entry (0 * w[0]) instead of being
        // masked off: no exec-mask traffic, and the loads of CPG_LDS_UNROLL steps are issued
        // back to back before the first multiply-add needs them
        int s = 0;
#pragma nounroll
        for (; s + CPG_LDS_UNROLL <= L; s += CPG_LDS_UNROLL) {
            unsigned e[CPG_LDS_UNROLL];
#pragma unroll
            for (int u = 0; u < CPG_LDS_UNROLL; u++) {
                const bool act = s + u < len;
                e[u] = act ? base + (unsigned)lane : P.dummy;
                base += cpgw::popc64(cpgw::ballot(act));
            }
            double v[CPG_LDS_UNROLL];
            unsigned co[CPG_LDS_UNROLL];
#pragma unroll
            for (int u = 0; u < CPG_LDS_UNROLL; u++) { v[u] = P.vals[e[u]]; co[u] = P.cols[e[u]]; }
            double x[CPG_LDS_UNROLL][G];
#pragma unroll
            for (int u = 0; u < CPG_LDS_UNROLL; u++)
#pragma unroll
                for (int g = 0; g < G; g++) x[u][g] = *(const double *)(wb + (unsigned)g * ldwb + co[u]);
#pragma unroll
            for (int u = 0; u < CPG_LDS_UNROLL; u++)
#pragma unroll
                for (int g = 0; g < G; g++) acc[g] = fma(v[u], x[u][g], acc[g]);
        }
#pragma nounroll
        for (; s < L; s++) {
            const bool act = s < len;
            const unsigned e = act ? base + (unsigned)lane : P.dummy;
            base += cpgw::popc64(cpgw::ballot(act));
            const double v = P.vals[e];
            const unsigned co = P.cols[e];
#pragma unroll
            for (int g = 0; g < G; g++)
                acc[g] = fma(v, *(const double *)(wb + (unsigned)g * ldwb + co), acc[g]);
        }
        double r[G];
        if (balanced) {
#pragma unroll
            for (int g = 0; g < G; g++) r[g] = cpgw::seg_sum_first<3>(acc[g], d >> 28);   // branch-free: the mask of an unused stage is zero
        } else {
#pragma unroll
            for (int g = 0; g < G; g++) r[g] = cpgw::group_sum_first_dyn(acc[g], lg);
        }
        cpgw::lds_order();
        if (row != CPG_NO_ROW) {
#pragma unroll
            for (int g = 0; g < G; g++) w[(unsigned)(g * ldw) + row] = r[g];
        }
        cpgw::lds_order();
    }
}

// work-vector slots behind the program's own that the generated executors use (see the table macros below):
// 16 dummy store targets and one slot that always holds 0.0
#define CPG_GEN_SLOT_MASK 0x1FFFu
#define CPG_GEN_DUMMY_SLOTS 16
#define CPG_GEN_EXTRA_SLOTS (CPG_GEN_DUMMY_SLOTS + 1)
#ifdef CPG_GEN_HEADER
// ---- family-specialised executor generated by cvxpygen_amd/codegen.py ---------------------------
#define CPG_GEN_ZERO(A) _Pragma("unroll") for (int g_ = 0; g_ < G; g_++) A[g_] = 0.0
// One multiply-add step in three separately scheduled parts (software pipeline, see codegen.py):
// offset + coefficient loads with literal LDS offsets, the gather from the work vector, the FMA.
// In a partial step lanes >= CNT load like everybody else (they hit later entries of the program, or
// the zero padding behind it: no address arithmetic) and get their coefficient replaced by zero -- or, when the
// offsets are stored for all 64 lanes of a step, find 0.0 as their operand and need no special treatment.
#define CPG_GEN_PAD 64
#ifdef CPG_GEN_COMPRESSED
// Dictionary-compressed program: structured families repeat their coefficients (MPC 12/4/10: 1 739
// distinct values among 8 523 entries), so an entry is ONE 32-bit word -- operand byte offset | byte
// offset of its coefficient in a dictionary << 16 -- read with a literal offset; the coefficient and the
// operand are then two independent gathers.  Same number of LDS operations per step as the plain form
// (8-byte coefficient + 2-byte offset + gather), 4 instead of 10 bytes per entry: the program of that
// family shrinks from 88 to 52 KB and a third wavefront per SIMD fits next to it.
#define CPG_GEN_VB(vals, lane) ((const char *)(vals))
#define CPG_GEN_CB(cols, lane) ((const char *)(cols) + (unsigned)(lane) * 4u)
#define CPG_GEN_LOAD_CV(ID, E) const unsigned wd##ID = *(const unsigned *)(cb + (E) * 4u);
#define CPG_GEN_LOAD_W(ID)                                                                         \
    const double vl##ID = *(const double *)(vb + (wd##ID >> 16));                                  \
    double x##ID[G];                                                                               \
    _Pragma("unroll") for (int g_ = 0; g_ < G; g_++) x##ID[g_] = *(const double *)(wb + (unsigned)g_ * ldwb + (wd##ID & 0xFFFFu));
#else
#define CPG_GEN_VB(vals, lane) ((const char *)(vals) + (unsigned)(lane) * 8u)
#define CPG_GEN_CB(cols, lane) ((const char *)(cols) + (unsigned)(lane) * 2u)
#define CPG_GEN_LOAD_CV(ID, E)                                                                     \
    const double vl##ID = *(const double *)(vb + (E) * 8u);                                        \
    const unsigned co##ID = *(const unsigned short *)(cb + (E) * 2u);
// offsets stored for all 64 lanes of every step (the header defines CPG_GEN_PADDED_OFFSETS): step K of the
// execution order finds them at [K][lane]
#define CPG_GEN_LOAD_CVP(ID, E, K)                                                                 \
    const double vl##ID = *(const double *)(vb + (E) * 8u);                                        \
    const unsigned co##ID = *(const unsigned short *)(cb + (K) * 128u);
// ... and with CPG_GEN_GROUPED_OFFSETS 2 the offsets of steps 2k and 2k + 1 share one 32-bit word at [k][lane]:
// one LDS read serves two steps
#define CPG_GEN_LOAD_CVP2A(ID, E, K)                                                               \
    const double vl##ID = *(const double *)(vb + (E) * 8u);                                        \
    const unsigned cp##ID = *(const unsigned *)(cb + (unsigned)lane * 2u + ((K) >> 1) * 256u);     \
    const unsigned co##ID = cp##ID & 0xFFFFu;
#define CPG_GEN_LOAD_CVP2B(ID, E, PREV)                                                            \
    const double vl##ID = *(const double *)(vb + (E) * 8u);                                        \
    const unsigned co##ID = cp##PREV >> 16;
// ... and with CPG_GEN_GROUPED_OFFSETS 4 the offsets of steps 4k .. 4k + 3 share one 64-bit word at [k][lane]
#define CPG_GEN_LOAD_CVP4A(ID, E, K)                                                               \
    const double vl##ID = *(const double *)(vb + (E) * 8u);                                        \
    const OffsetQuad cq##ID = *(const OffsetQuad *)(cb + (unsigned)lane * 6u + ((K) >> 2) * 512u);          \
    const unsigned co##ID = cq##ID.a & 0xFFFFu;
#define CPG_GEN_LOAD_CVP4B(ID, E, FIRST, J)                                                        \
    const double vl##ID = *(const double *)(vb + (E) * 8u);                                        \
    const unsigned co##ID = (J) == 1 ? cq##FIRST.a >> 16 : (J) == 2 ? (cq##FIRST.b & 0xFFFFu) : cq##FIRST.b >> 16;
#define CPG_GEN_LOAD_W(ID)                                                                         \
    double x##ID[G];                                                                               \
    _Pragma("unroll") for (int g_ = 0; g_ < G; g_++) x##ID[g_] = *(const double *)(wb + (unsigned)g_ * ldwb + co##ID);
#endif
#define CPG_GEN_FMA_FULL(A, ID)                                                                    \
    _Pragma("unroll") for (int g_ = 0; g_ < G; g_++) A[g_] = fma(vl##ID, x##ID[g_], A[g_]);
#define CPG_GEN_FMA_PART(A, ID, CNT)                                                               \
    {                                                                                              \
        const double v_ = lane < (CNT) ? vl##ID : 0.0;                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < G; g_++) A[g_] = fma(v_, x##ID[g_], A[g_]);        \
    }
// per (chunk, lane) table entry: output slot (13 bits) | segmented-reduction mask << 13.  Lanes that
// own no row store to one of 16 dummy slots behind the work vector (a ds_write_b64 is served in 16-lane groups
// whose 8-byte slots collide modulo 16: every idle lane of a group gets a dummy no row of the group collides
// with): the store is unconditional and the whole
// program stays one basic block.  One more slot behind them always holds 0.0: the operand of idle lanes
// when the offsets are stored for all 64 lanes of a step (CPG_GEN_PADDED_OFFSETS).
// the table entries of four chunks come with one LDS read (CPG_GEN_LOAD_ROWS, issued when the phase of the first
// of them opens); CPG_GEN_ROW picks chunk C's
#define CPG_GEN_LOAD_ROWS(Q) const OffsetQuad rq##Q = *(const OffsetQuad *)((const char *)rows + (unsigned)lane * 8u + (Q) * 512u);
#define CPG_GEN_ROW(Q, J) ((J) == 0 ? (rq##Q.a & 0xFFFFu) : (J) == 1 ? (rq##Q.a >> 16) : (J) == 2 ? (rq##Q.b & 0xFFFFu) : (rq##Q.b >> 16))
#define CPG_GEN_REDUCE_STORE(A, LG, Q, J)                                                          \
    {                                                                                              \
        const unsigned row_ = CPG_GEN_ROW(Q, J) & CPG_GEN_SLOT_MASK;                               \
        _Pragma("unroll") for (int g_ = 0; g_ < G; g_++)                                           \
            w[(unsigned)(g_ * ldw) + row_] = cpgw::group_sum_first<LG>(A[g_]);                     \
    }
// ... with the stage masks as 64-bit lane-mask literals (codegen.stage_masks): no shift / and / compare per stage
#define CPG_GEN_SEGREDUCE_STORE_LIT(A, S, Q, J, M0, M1, M2)                                       \
    {                                                                                              \
        const unsigned row_ = CPG_GEN_ROW(Q, J) & CPG_GEN_SLOT_MASK;                               \
        _Pragma("unroll") for (int g_ = 0; g_ < G; g_++)                                           \
            w[(unsigned)(g_ * ldw) + row_] = cpgw::seg_sum_first_lit<S>(A[g_], M0, M1, M2);        \
    }
#define CPG_GEN_SEGREDUCE_STORE(A, S, Q, J)                                                        \
    {                                                                                              \
        const unsigned e_ = CPG_GEN_ROW(Q, J);                                                     \
        _Pragma("unroll") for (int g_ = 0; g_ < G; g_++)                                           \
            w[(unsigned)(g_ * ldw) + (e_ & CPG_GEN_SLOT_MASK)] = cpgw::seg_sum_first<S>(A[g_], e_ >> 13); \
    }
}  // namespace cpg
#include CPG_GEN_HEADER
namespace cpg {
#ifndef CPG_GEN_N
#error "generated program headers must carry the family dimensions (codegen.write_program_header(plan=...))"
#endif
#ifdef CPG_GEN_N
// dimensions and per-slot row classes of the family this build was generated for
struct GenFam {
    static constexpr unsigned n = CPG_GEN_N, m = CPG_GEN_M;
    static constexpr int n_slots = CPG_GEN_NSLOTS;
    static constexpr int ct(int s) { constexpr signed char t[] = CPG_GEN_CT_INIT; return t[s]; }
    // chunks of the natural-layout row programs of the termination test: {steps, first step}
    static constexpr int rows_len(int which, int s) {
        constexpr int a[][2] = CPG_GEN_AROWS; constexpr int p[][2] = CPG_GEN_PROWS; constexpr int t[][2] = CPG_GEN_ATROWS;
        return which == 0 ? (s < CPG_GEN_AROWS_N ? a[s < CPG_GEN_AROWS_N ? s : 0][0] : 0)
             : which == 1 ? (s < CPG_GEN_PROWS_N ? p[s < CPG_GEN_PROWS_N ? s : 0][0] : 0)
                          : (s < CPG_GEN_ATROWS_N ? t[s < CPG_GEN_ATROWS_N ? s : 0][0] : 0);
    }
    static constexpr int rows_off(int which, int s) {
        constexpr int a[][2] = CPG_GEN_AROWS; constexpr int p[][2] = CPG_GEN_PROWS; constexpr int t[][2] = CPG_GEN_ATROWS;
        return which == 0 ? a[s < CPG_GEN_AROWS_N ? s : 0][1] : which == 1 ? p[s < CPG_GEN_PROWS_N ? s : 0][1]
                                                                            : t[s < CPG_GEN_ATROWS_N ? s : 0][1];
    }
};
#endif
#endif

// natural-layout product: chunk s delivers element lane + 64 s of the result to this lane
CPG_DEV double natural_chunk(const DevProgram &P, int s, const double *w, int lane) {
    double o[1];
    o[0] = 0.0;
    if (s < P.n_chunks) do_chunk<1, false>(P, s, w, 0, lane, o);
    return o[0];
}

// The same with the chunk's step count and first step as literals of the generated family (the loops
// unroll): the coefficient / offset loads of up to CPG_ROWS_BATCH steps are requested together before the
// first multiply-add consumes one, so a chunk costs about one L2 round trip instead of one per four steps
// plus one for its header.  The termination test's three products walk 19 chunks (MPC 12/4/10).
#ifndef CPG_ROWS_BATCH
#define CPG_ROWS_BATCH 8
#endif
CPG_DEV double natural_chunk_lit(const DevProgram &P, const int len, const int off, const double *w, int lane_in) {
    // per-call copy of the lane: keeps the optimiser from hoisting the ~330 per-lane load offsets of the
    // unrolled products out of the enclosing loops (and spilling them), see cpgw::opaque
    const int lane = cpgw::opaque(lane_in);
    const unsigned e0 = (unsigned)off * 64u + (unsigned)lane;
    double acc = 0.0;
#pragma unroll
    for (int t0 = 0; t0 < len; t0 += CPG_ROWS_BATCH) {
        double v[CPG_ROWS_BATCH];
        unsigned c[CPG_ROWS_BATCH];
#pragma unroll
        for (int t = 0; t < CPG_ROWS_BATCH; t++)
            if (t0 + t < len) { v[t] = cpgw::gld(P.vals, e0 + 64u * (unsigned)(t0 + t)); c[t] = cpgw::gld(P.cols, e0 + 64u * (unsigned)(t0 + t)); }
#pragma unroll
        for (int t = 0; t < CPG_ROWS_BATCH; t++)
            if (t0 + t < len) acc = fma(v[t], w[c[t]], acc);
    }
    return acc;
}
#ifdef CPG_GEN_N
#define CPG_NATURAL_ROWS(P, which, s, w, lane) natural_chunk_lit(P, GenFam::rows_len(which, s), GenFam::rows_off(which, s), w, lane)
#else
#define CPG_NATURAL_ROWS(P, which, s, w, lane) natural_chunk(P, s, w, lane)
#endif

// ------------------------------------------------------------------------------------ per-instance state
template <int NSX, int NSZ, int NV>
struct Inst {
    static const int NVX = MinI<NV, NSX>::v, NVZ = MinI<NV, NSZ>::v;
    double x[NSX], z[NSZ], y[NSZ];
    double qv[NVX], uv[NVZ];
    double dconst;
    long long b;
    int done;
};

struct CheckOut {
    double prim_res, dual_res, obj;
    int status;
};

CPG_DEV double csr_row(const DevCsr &mp, unsigned row, const double *theta, double v) {
    if (mp.nnz > 0) {
        const unsigned s = (unsigned)cpgw::gld(mp.ptr, row), e = (unsigned)cpgw::gld(mp.ptr, row + 1u);
        for (unsigned k = s; k < e; k++) v = fma(cpgw::gld(mp.val, k), cpgw::gld(theta, (unsigned)cpgw::gld(mp.idx, k)), v);
    }
    return v;
}

// How the termination test obtains the instance's q / u and the sparse row products of the
// staged vector w.  Shared-factor kernel: q / u of entry i (slot s) = per-instance register for the
// leading NV slots, otherwise the family's base vectors staged in LDS (`sh` = [q_base | u_base]);
// products through the family's natural-layout row programs.
template <int NSX, int NSZ, int NV>
struct SharedCtx {
    static constexpr bool kTestsFirst = true;      // check(): verdicts of infeasibility_tests() are passed in
    static constexpr bool kOpaqueLane = true;
    const DevFamily &F;
    const double *sh, *shu;       // base q / base u (block-shared LDS copy, or the global arrays)
    const Inst<NSX, NSZ, NV> &I;
    const double *w;
    int lane;
    CPG_DEV double q(int s, unsigned i) const {
        return s < Inst<NSX, NSZ, NV>::NVX ? I.qv[s < Inst<NSX, NSZ, NV>::NVX ? s : 0] : sh[i];
    }
    CPG_DEV double u(int s, unsigned i) const {
        return s < Inst<NSX, NSZ, NV>::NVZ ? I.uv[s < Inst<NSX, NSZ, NV>::NVZ ? s : 0] : shu[i];
    }
    CPG_DEV void products(int) const {}             // (contexts that compute ALL rows of a product at once do it here)
    // entries of the scaling vectors as the termination test reads them (a context may keep them closer than global memory)
    CPG_DEV void stage(int) const {}
    CPG_DEV double sE(unsigned i) const { return cpgw::gld(F.E, i); }
    CPG_DEV double sEinv(unsigned i) const { return cpgw::gld(F.Einv, i); }
    CPG_DEV double sD(unsigned i) const { return cpgw::gld(F.D, i); }
    CPG_DEV double sDinv(unsigned i) const { return cpgw::gld(F.Dinv, i); }
    CPG_DEV double ax(int s) const { return CPG_NATURAL_ROWS(F.A_rows, 0, s, w, lane); }     // (A v)_i, v = w[0..n)
    CPG_DEV double px(int s) const { return CPG_NATURAL_ROWS(F.P_rows, 1, s, w, lane); }     // (P v)_j
    CPG_DEV double atx(int s) const { return CPG_NATURAL_ROWS(F.At_rows, 2, s, w, lane); }   // (A' v)_j, v = w[n..)
};

// cpg_canonicalize_q/l/u/d + osqp_update_data_vec for the parameter-dependent entries; returns
// (wave-uniform) whether a row changed class w.r.t. the family's factor.
template <int NSX, int NSZ, int NV>
CPG_DEV bool canonicalise(const DevFamily &F, const DevUpdate &U, const double *theta,
                          Inst<NSX, NSZ, NV> &I, int lane) {
    bool bad = false;
#pragma unroll
    for (int s = 0; s < NSX; s++) I.x[s] = 0.0;
#pragma unroll
    for (int s = 0; s < NSZ; s++) { I.z[s] = 0.0; I.y[s] = 0.0; }
#pragma unroll
    for (int s = 0; s < Inst<NSX, NSZ, NV>::NVX; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        I.qv[s] = (i < (unsigned)F.n) ? csr_row(U.map_q, i, theta, cpgw::gld(U.q_base, i)) : 0.0;
    }
#pragma unroll
    for (int s = 0; s < Inst<NSX, NSZ, NV>::NVZ; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        I.uv[s] = 0.0;
        if (i < (unsigned)F.m) {
            const double uv = csr_row(U.map_u, i, theta, cpgw::gld(U.u_base, i));
            I.uv[s] = uv;
            // an inequality row whose bound became infinite is a free row for OSQP (rho_min)
            const int ctp = (int)cpgw::gld(F.ctype, i);
            bad |= (ctp == 0 && uv > CPG_INFTY * CPG_MIN_SCALING) || (ctp == -1 && !(uv > CPG_INFTY * CPG_MIN_SCALING));
        }
    }
    I.dconst = csr_row(U.map_d, 0, theta, U.d_base);
    return cpgw::wave_any(bad);
}

// Where the steps delta_x / delta_y of the checked iteration are kept for OSQP's infeasibility tests:
// registers (element i on lane i % 64, slot i / 64; the shared-factor kernel: nothing leaves the CU) or the
// wavefront's buffer in global memory (the per-instance factor kernels, whose iterates, q and u already
// take most of the register budget).
template <int NS>
struct RegDelta {
    const double (&a)[NS];
    CPG_DEV double operator()(int s, unsigned) const { return a[s]; }
};
struct MemDelta {
    const double *p;
    CPG_DEV double operator()(int, unsigned i) const { return cpgw::gld(p, i); }
};
struct NoDelta {      // check() of a kernel that ran the infeasibility tests itself
    CPG_DEV double operator()(int, unsigned) const { return 0.0; }
};

// is_primal_infeasible on delta_y (OSQP paper sec. 3.4); wave-uniform result.
template <int NSX, int NSZ, typename Ctx, typename DY>
CPG_DEV bool primal_infeasible(const DevFamily &F, const Ctx &cx, const signed char (&ct)[NSZ],
                               bool unsc, double eps, double *w, const DY &dy, int lane) {
    cx.stage(1);               // (contexts that keep E closer than global memory for this test)
    double nrm = 0.0, lhs = 0.0;
    double dyp[NSZ];                                   // delta_y projected on the polar of the recession cone of [l, u]
#pragma unroll
    for (int s = 0; s < NSZ; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        dyp[s] = 0.0;
        if (i < (unsigned)F.m) {
            const double uu = cx.u(s, i);
            const bool eq = ct[s] == 1;
            const double ll = eq ? uu : -CPG_INFTY;
            const bool iu = uu > CPG_INFTY * CPG_MIN_SCALING, il = !eq;
            double d = dy(s, i);
            if (iu && il) d = 0.0; else if (iu) d = cpgw::dmin2(d, 0.0); else if (il) d = cpgw::dmax2(d, 0.0);
            dyp[s] = d;
            nrm = cpgw::dmax2(nrm, fabs(unsc ? cx.sE(i) * d : d));
            lhs += uu * cpgw::dmax2(d, 0.0) + ll * cpgw::dmin2(d, 0.0);
        }
        cpgw::sched_fence();
    }
    nrm = cpgw::wave_max_nonneg(nrm);
    if (!(nrm > CPG_DIV_TOL)) return false;
    lhs = cpgw::wave_sum(lhs);
    if (!(lhs < eps * nrm)) return false;
#pragma unroll
    for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < (unsigned)F.m) w[(unsigned)F.n + i] = dyp[s]; }
    cpgw::lds_order();
    cx.products(4);
    double r = 0.0;
#pragma unroll
    for (int s = 0; s < NSX; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        const double t = cx.atx(s);
        if (i < (unsigned)F.n) r = cpgw::dmax2(r, fabs(unsc ? cx.sDinv(i) * t : t));
        cpgw::sched_fence();
    }
    r = cpgw::wave_max_nonneg(r);
    cpgw::lds_order();
    return r < eps * nrm;
}

// is_dual_infeasible on delta_x; wave-uniform result
template <int NSX, int NSZ, typename Ctx, typename DX>
CPG_DEV bool dual_infeasible(const DevFamily &F, const Ctx &cx, const signed char (&ct)[NSZ],
                             bool unsc, double eps, double *w, const DX &dx, int lane) {
    cx.stage(2);               // (... D)
    double nrm = 0.0, qdx = 0.0;
#pragma unroll
    for (int s = 0; s < NSX; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        if (i < (unsigned)F.n) {
            const double d = dx(s, i);
            nrm = cpgw::dmax2(nrm, fabs(unsc ? cx.sD(i) * d : d));
            qdx += cx.q(s, i) * d;
        }
    }
    nrm = cpgw::wave_max_nonneg(nrm);
    if (!(nrm > CPG_DIV_TOL)) return false;
    const double cs = unsc ? F.c : 1.0;
    qdx = cpgw::wave_sum(qdx);
    if (!(qdx < -cs * eps * nrm)) return false;
#pragma unroll
    for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < (unsigned)F.n) w[i] = dx(s, i); }
    cpgw::lds_order();
    cx.products(2);
    double r = 0.0;
#pragma unroll
    for (int s = 0; s < NSX; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        const double t = cx.px(s);
        if (i < (unsigned)F.n) r = cpgw::dmax2(r, fabs(unsc ? cx.sDinv(i) * t : t));
        cpgw::sched_fence();
    }
    r = cpgw::wave_max_nonneg(r);
    bool res = false;
    if (r < cs * eps * nrm) {
        cx.products(1);
        bool viol = false;
#pragma unroll
        for (int s = 0; s < NSZ; s++) {
            const unsigned i = (unsigned)lane + 64u * (unsigned)s;
            const double a = cx.ax(s);
            if (i < (unsigned)F.m) {
                const double av = unsc ? cx.sEinv(i) * a : a;
                if ((cx.u(s, i) < CPG_INFTY * CPG_MIN_SCALING && av > eps * nrm) ||
                    (ct[s] == 1 && av < -eps * nrm)) viol = true;
            }
            cpgw::sched_fence();
        }
        res = !cpgw::wave_any(viol);
    }
    cpgw::lds_order();
    return res;
}

// OSQP runs is_primal_infeasible / is_dual_infeasible inside check_termination, and only when the matching
// residual test has failed.  They are pure functions of (delta_y, delta_x, problem data, eps), so the kernels
// evaluate them right after the checked iteration -- while the steps are still in the registers that
// iteration produced them in -- and hand check() the verdicts, which it uses under OSQP's conditions.  The
// steps (13 registers per lane on the MPC 12/4/10 family) are then dead before the register-hungry residual
// products of check() start; kept alive across them (for the tests, and for the second, approximate pass at
// max_iter) they were spilled to scratch at every check: 2.9 GB of writes per 100 000 instances.
struct InfeasVerdict { bool primal, dual; };

template <int NSX, int NSZ, typename Ctx, typename DX, typename DY>
CPG_DEV InfeasVerdict infeasibility_tests(const DevFamily &F, const Ctx &cx, const signed char (&ct)[NSZ],
                                          const DevSettings &S, const DX &dx, const DY &dy,
                                          double *w, int lane_in, bool approximate) {
    const int lane = cpgw::opaque(lane_in);                 // per-call copy of the lane id, see check()
    cpgw::assume((unsigned)lane < 64u);
    const bool unsc = !S.scaled_termination;
    const double mult = approximate ? 10.0 : 1.0;
    InfeasVerdict v;
    v.primal = F.m != 0 && primal_infeasible<NSX, NSZ, Ctx, DY>(F, cx, ct, unsc, S.eps_prim_inf * mult, w, dy, lane);
    v.dual = dual_infeasible<NSX, NSZ, Ctx, DX>(F, cx, ct, unsc, S.eps_dual_inf * mult, w, dx, lane);
    return v;
}

// The three sparse products of update_info on the staged iterates (w = [x | y]) and the SCALED norms
// OSQP's compute_rho_estimate needs; shared by the termination test and the rho adaptation.
struct ScaledNorms {
    double prim_res, dual_res;    // ||Ax - z||, ||Px + q + A'y||   (scaled space)
    double nz, nax, nq, naty, npx;
};

// update_info + check_termination: residuals in the unscaled space (scaled_termination = 0),
// optimality / infeasibility decisions.  status stays 11 (unsolved) when nothing triggers.
// `sn` (optional) receives the scaled norms of the same products.
//
// Two forms, chosen by the context type, because this code is inlined into kernels whose register allocation
// around their hot loops reacts to it (both measured, HISTORY.md 4.1c / 4.2):
//   Ctx::kTestsFirst  (shared-factor kernel)  the caller has run infeasibility_tests() right after the iteration
//                     and passes the verdicts `iv`; the lane id is re-derived per call (cpgw::opaque) so that the
//                     dozens of per-lane addresses of the test are computed here and die here;
//   otherwise         (per-instance factor kernels)  OSQP's own order: the tests run in here, on dx / dy, when
//                     the matching residual test has failed; the caller's lane id is used as it is.
template <int NSX, int NSZ, typename Ctx, typename DX, typename DY>
CPG_DEV CheckOut check(const DevFamily &F, const Ctx &cx, const signed char (&ct)[NSZ],
                       const DevSettings &S, const double (&Ix)[NSX], const double (&Iz)[NSZ],
                       const double (&Iy)[NSZ], const DX &dx, const DY &dy, InfeasVerdict iv,
                       double *w, int lane_in, bool approximate, ScaledNorms *sn = nullptr) {
    const int lane = (Ctx::kTestsFirst || Ctx::kOpaqueLane) ? cpgw::opaque(lane_in) : lane_in;
    cpgw::assume((unsigned)lane < 64u);
    const bool unsc = !S.scaled_termination;
    const double mult = approximate ? 10.0 : 1.0;
    const double ea = S.eps_abs * mult, er = S.eps_rel * mult;
    CheckOut o;
#pragma unroll
    for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < (unsigned)F.n) w[i] = Ix[s]; }
#pragma unroll
    for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < (unsigned)F.m) w[(unsigned)F.n + i] = Iy[s]; }
    cpgw::lds_order();
    cx.products(1);          // (contexts whose products share result slots: A x is consumed before P x and A' y are formed)
    double rp = 0.0, nz = 0.0, na = 0.0, sup = 0.0;
    double s_rp = 0.0, s_nz = 0.0, s_na = 0.0;
#pragma unroll
    for (int s = 0; s < NSZ; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        const double ax = cx.ax(s);
        if (i < (unsigned)F.m) {
            const double ei = unsc ? cx.sEinv(i) : 1.0;
            rp = cpgw::dmax2(rp, fabs(ei * (ax - Iz[s])));
            nz = cpgw::dmax2(nz, fabs(ei * Iz[s]));
            na = cpgw::dmax2(na, fabs(ei * ax));
            if (sn) { s_rp = cpgw::dmax2(s_rp, fabs(ax - Iz[s])); s_nz = cpgw::dmax2(s_nz, fabs(Iz[s])); s_na = cpgw::dmax2(s_na, fabs(ax)); }
            if (S.check_dualgap) {   // support function of [l, u] at y: u'y+ + l'y-  (l = u on equality rows, -inf otherwise)
                const double uu = cx.u(s, i), yy = Iy[s];
                if (uu < CPG_INFTY * CPG_MIN_SCALING && yy > 0.0) sup += uu * yy;
                if (ct[s] == 1 && uu > -CPG_INFTY * CPG_MIN_SCALING && yy < 0.0) sup += uu * yy;
            }
        }
        cpgw::sched_fence();
    }
    cx.products(6);
    double rd = 0.0, nq = 0.0, nat = 0.0, npx = 0.0, quad = 0.0, lin = 0.0;
    double s_rd = 0.0, s_nq = 0.0, s_nat = 0.0, s_npx = 0.0;
#pragma unroll
    for (int s = 0; s < NSX; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        const double px = cx.px(s);
        const double aty = cx.atx(s);
        if (i < (unsigned)F.n) {
            const double di = unsc ? cx.sDinv(i) : 1.0;
            const double qq = cx.q(s, i);
            rd = cpgw::dmax2(rd, fabs(di * (qq + px + aty)));
            nq = cpgw::dmax2(nq, fabs(di * qq));
            nat = cpgw::dmax2(nat, fabs(di * aty));
            npx = cpgw::dmax2(npx, fabs(di * px));
            if (sn) { s_rd = cpgw::dmax2(s_rd, fabs(qq + px + aty)); s_nq = cpgw::dmax2(s_nq, fabs(qq));
                      s_nat = cpgw::dmax2(s_nat, fabs(aty)); s_npx = cpgw::dmax2(s_npx, fabs(px)); }
            quad += Ix[s] * px;
            lin += qq * Ix[s];
        }
        cpgw::sched_fence();
    }
    cpgw::lds_order();
    const double cs = unsc ? F.cinv : 1.0;
    rp = cpgw::wave_max_nonneg(rp); nz = cpgw::wave_max_nonneg(nz); na = cpgw::wave_max_nonneg(na);
    rd = cs * cpgw::wave_max_nonneg(rd);
    const double dn = cs * cpgw::dmax2(cpgw::wave_max_nonneg(nq),
                                       cpgw::dmax2(cpgw::wave_max_nonneg(nat), cpgw::wave_max_nonneg(npx)));
    quad = cpgw::wave_sum(quad); lin = cpgw::wave_sum(lin);
    o.prim_res = rp; o.dual_res = rd; o.obj = (0.5 * quad + lin) * F.cinv;
    if (sn) {
        sn->prim_res = cpgw::wave_max_nonneg(s_rp); sn->nz = cpgw::wave_max_nonneg(s_nz); sn->nax = cpgw::wave_max_nonneg(s_na);
        sn->dual_res = cpgw::wave_max_nonneg(s_rd); sn->nq = cpgw::wave_max_nonneg(s_nq);
        sn->naty = cpgw::wave_max_nonneg(s_nat); sn->npx = cpgw::wave_max_nonneg(s_npx);
    }

    o.status = 11;
    if (rp > CPG_INFTY || rd > CPG_INFTY) { o.status = 9; o.obj = NAN; return o; }
    bool pc = false, dc = false, pic = false, dic = false, gc = true;
    if (F.m == 0) pc = true;
    else if (rp < ea + er * cpgw::dmax2(nz, na)) pc = true;
    else pic = Ctx::kTestsFirst ? iv.primal : primal_infeasible<NSX, NSZ, Ctx, DY>(F, cx, ct, unsc, S.eps_prim_inf * mult, w, dy, lane);
    if (rd < ea + er * dn) dc = true;
    else dic = Ctx::kTestsFirst ? iv.dual : dual_infeasible<NSX, NSZ, Ctx, DX>(F, cx, ct, unsc, S.eps_dual_inf * mult, w, dx, lane);
    if (S.check_dualgap) {   // OSQP >= 1.0: |primal - dual objective| against eps_abs + eps_rel max(|primal|, |dual|)
        sup = cpgw::wave_sum(sup);
        const double dual_obj = (-0.5 * quad - sup) * F.cinv, gap = fabs(quad + lin + sup) * F.cinv;
        gc = gap < ea + er * cpgw::dmax2(fabs(o.obj), fabs(dual_obj));
    }
    if (pc && dc && gc) o.status = approximate ? 2 : 1;
    else if (pic) { o.status = approximate ? 4 : 3; o.obj = CPG_INFTY; }
    else if (dic) { o.status = approximate ? 6 : 5; o.obj = -CPG_INFTY; }
    return o;
}

// The workspace of an instance in the layout of DevBatch::state_out: scaled iterates x | z | y in canonical
// order (zeros when `keep` is false), then rho.
template <int NSX, int NSZ>
CPG_DEV void store_state(const DevFamily &F, double *so, const double (&Ix)[NSX], const double (&Iz)[NSZ],
                         const double (&Iy)[NSZ], bool keep, double rho, int lane) {
#pragma unroll
    for (int s = 0; s < NSX; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        if (i < (unsigned)F.n) cpgw::gst(so, F.ord ? (unsigned)cpgw::gld(F.ord, i) : i, keep ? Ix[s] : 0.0);
    }
#pragma unroll
    for (int s = 0; s < NSZ; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        if (i < (unsigned)F.m) {
            const unsigned c = F.ord ? (unsigned)cpgw::gld(F.ord, (unsigned)F.n + i) : i;
            cpgw::gst(so, (unsigned)F.n + c, keep ? Iz[s] : 0.0);
            cpgw::gst(so, (unsigned)(F.n + F.m) + c, keep ? Iy[s] : 0.0);
        }
    }
    if (lane == 0) so[F.n + 2 * F.m] = rho;
}

// OSQP's compute_rho_estimate on the scaled norms of update_info: rho sqrt(normalised primal / dual residual)
CPG_DEV double rho_estimate(const ScaledNorms &sn, double rho_settings) {
    const double pr = sn.prim_res / (cpgw::dmax2(sn.nz, sn.nax) + CPG_DIV_TOL);
    const double dr = sn.dual_res / (cpgw::dmax2(sn.nq, cpgw::dmax2(sn.naty, sn.npx)) + CPG_DIV_TOL);
    return cpgw::dmin2(cpgw::dmax2(rho_settings * sqrt(pr / dr), CPG_RHO_MIN), CPG_RHO_MAX);
}

// store_solution + cpg_retrieve_*: unscale, gather the user-facing entries, write the info scalars;
// with Bt.state_out also the workspace a sequential caller carries to its next solve: the scaled iterates
// (reset to zero when there is no solution, as osqp_solve does) and rho.
template <int NSX, int NSZ, bool OPAQUE_LANE = true>
CPG_DEV void finalize(const DevFamily &F, const DevBatch &Bt, const double (&Ix)[NSX], const double (&Iz)[NSZ],
                      const double (&Iy)[NSZ], double dconst, long long b, double *w, int lane_in, int iter,
                      const CheckOut &o, double rho) {
    const int lane = OPAQUE_LANE ? cpgw::opaque(lane_in) : lane_in;     // see check(): addresses computed here die here
    cpgw::assume((unsigned)lane < 64u);
    const bool has_sol = o.status == 1 || o.status == 2 || o.status == 7;
    if (Bt.state_out) store_state<NSX, NSZ>(F, Bt.state_out + (size_t)b * (size_t)(F.n + 2 * F.m + 1), Ix, Iz, Iy, has_sol, rho, lane);
#pragma unroll
    for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < (unsigned)F.n) w[i] = has_sol ? cpgw::gld(F.D, i) * Ix[s] : NAN; }
#pragma unroll
    for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; if (i < (unsigned)F.m) w[(unsigned)F.n + i] = has_sol ? F.cinv * cpgw::gld(F.E, i) * Iy[s] : NAN; }
    cpgw::lds_order();
    double *pp = Bt.prim + (size_t)b * F.n_prim, *dp = Bt.dual + (size_t)b * F.n_dual;
    for (unsigned k = (unsigned)lane; k < (unsigned)F.n_prim; k += 64u) cpgw::gst(pp, k, w[(unsigned)cpgw::gld(F.prim_idx, k)]);
    for (unsigned k = (unsigned)lane; k < (unsigned)F.n_dual; k += 64u) cpgw::gst(dp, k, w[(unsigned)F.n + (unsigned)cpgw::gld(F.dual_idx, k)]);
    if (lane == 0) {
        double ov = o.obj + dconst;
        if (F.is_max) ov = -ov;
        Bt.obj[b] = ov; Bt.iter[b] = iter; Bt.status[b] = o.status;
        Bt.pri_res[b] = o.prim_res; Bt.dua_res[b] = o.dual_res;
    }
    cpgw::lds_order();
}

// osqp_warm_start semantics of a sequential caller: the scaled iterates of the previous solve
template <int NSX, int NSZ>
CPG_DEV void load_state(const DevFamily &F, const double *si, double (&Ix)[NSX], double (&Iz)[NSZ],
                        double (&Iy)[NSZ], int lane) {
#pragma unroll
    for (int s = 0; s < NSX; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        if (i < (unsigned)F.n) Ix[s] = cpgw::gld(si, F.ord ? (unsigned)cpgw::gld(F.ord, i) : i);
    }
#pragma unroll
    for (int s = 0; s < NSZ; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        if (i < (unsigned)F.m) {
            const unsigned c = F.ord ? (unsigned)cpgw::gld(F.ord, (unsigned)F.n + i) : i;
            Iz[s] = cpgw::gld(si, (unsigned)F.n + c); Iy[s] = cpgw::gld(si, (unsigned)(F.n + F.m) + c);
        }
    }
}

// Hand-over of an instance whose rho changed to the per-instance factor kernel behind this one (DevBatch):
// workspace (iterates + the new rho), iteration count, and its number in the list; status -3 marks the row
// until that kernel has written its results.
#define CPG_STATUS_HANDED_OVER (-3)
template <int NSX, int NSZ>
CPG_DEV void hand_over(const DevFamily &F, const DevBatch &Bt, const double (&Ix)[NSX], const double (&Iz)[NSZ],
                       const double (&Iy)[NSZ], long long b, int lane_in, int iter, double rho_new) {
    const int lane = cpgw::opaque(lane_in);
    cpgw::assume((unsigned)lane < 64u);
    if (!Bt.ho_state || !Bt.ho_list) {      // no kernel behind this one: report it (the host layer re-solves the row)
        if (lane == 0) { Bt.obj[b] = NAN; Bt.iter[b] = iter; Bt.status[b] = -2; Bt.pri_res[b] = 0.0; Bt.dua_res[b] = 0.0; }
        return;
    }
    store_state<NSX, NSZ>(F, Bt.ho_state + (size_t)b * (size_t)(F.n + 2 * F.m + 1), Ix, Iz, Iy, true, rho_new, lane);
    if (lane == 0) {
        Bt.iter[b] = iter; Bt.status[b] = CPG_STATUS_HANDED_OVER;
        Bt.ho_list[cpgw::atomic_next(Bt.ho_count)] = (int)b;
    }
}

// row class of slot s in the hot loop: a literal where the generated family has a uniform slot
#if defined(CPG_GEN_N)
#define CPG_ROW_CLASS(s) (GenFam::ct(s) != 2 ? GenFam::ct(s) : (int)ct[s])
#else
#define CPG_ROW_CLASS(s) ((int)ct[s])
#endif

// ------------------------------------------------------------------------------------ the kernel body
template <int NSX, int NSZ, int NV, int G, bool LDSPROG>
CPG_DEV void osqp_shared_body(const DevFamily &F, const DevUpdate &U, const DevSettings &S,
                              const DevBatch &Bt, double *lds, int wave_global) {
    typedef Inst<NSX, NSZ, NV> InstT;
    const int lane0 = cpgw::lane_id();
    const int lane = lane0;
#if defined(CPG_GEN_N)
    // family-specialised build: dimensions are literals, so the bounds checks of full slots fold away
    constexpr unsigned n_c = GenFam::n, m_c = GenFam::m;
    constexpr int ldw = GenFam::n_slots + CPG_GEN_EXTRA_SLOTS;   // + dummy store targets and the zero slot
    static_assert(!LDSPROG || ((n_c + 63) / 64 == (unsigned)NSX && (m_c + 63) / 64 == (unsigned)NSZ), "slot class of the generated family");
#else
    const unsigned n_c = (unsigned)F.n, m_c = (unsigned)F.m;
    const int ldw = F.n_slots;
#endif
    const int N = F.n + F.m;
    // block-shared copy of the family's base vectors, then one work vector per instance
    double *sh = lds;
    for (unsigned t = cpgw::thread_in_block(); t < (unsigned)N; t += cpgw::block_threads())
        sh[t] = t < (unsigned)F.n ? cpgw::gld(U.q_base, t) : cpgw::gld(U.u_base, t - (unsigned)F.n);
    const double *shu = sh + n_c;
    // LDS-resident program: [vals | cols | desc | ctab] right after the base vectors
    LdsProg LP;
    size_t lds_off = (size_t)N;
    if (LDSPROG) {
        const DevRagged &R = F.kkt_ragged;
#ifdef CPG_GEN_HEADER
        const unsigned nnzp = (unsigned)R.nnz + CPG_GEN_PAD;   // zero padding, see CPG_GEN_STEP_PART
#else
        const unsigned nnzp = (unsigned)R.nnz;
#endif
        const unsigned nt = cpgw::block_threads(), t0 = cpgw::thread_in_block();
#ifdef CPG_GEN_COMPRESSED
        double *lv = lds + lds_off;                         lds_off += (size_t)R.n_dict;
        unsigned *lw = (unsigned *)(lds + lds_off);         lds_off += (size_t)((nnzp + 1) / 2);
        for (unsigned t = t0; t < (unsigned)R.n_dict; t += nt) lv[t] = cpgw::gld(R.dict, t);
        for (unsigned t = t0; t < nnzp; t += nt) lw[t] = t < (unsigned)R.nnz ? cpgw::gld(R.words, t) : 0u;   // padding: 0 * w[0]
        const unsigned short *lc = (const unsigned short *)lw;
#else
        double *lv = lds + lds_off;                         lds_off += (size_t)nnzp;
#ifdef CPG_GEN_PADDED_OFFSETS
        constexpr unsigned n_off = 64u * ((CPG_GEN_PADDED_OFFSETS + 3u) & ~3u);
        unsigned short *lc = (unsigned short *)(lds + lds_off); lds_off += (size_t)((n_off + 3) / 4);
        for (unsigned t = t0; t < nnzp; t += nt) lv[t] = t < (unsigned)R.nnz ? cpgw::gld(R.vals, t) : 0.0;
        for (unsigned t = t0; t < n_off; t += nt) lc[t] = cpgw::gld(R.cols_padded, t);
#else
        unsigned short *lc = (unsigned short *)(lds + lds_off); lds_off += (size_t)((nnzp + 3) / 4);
        for (unsigned t = t0; t < nnzp; t += nt) {
            const bool in = t < (unsigned)R.nnz;
            lv[t] = in ? cpgw::gld(R.vals, t) : 0.0; lc[t] = in ? cpgw::gld(R.cols, t) : (unsigned short)0;
        }
#endif
#endif
        LP.vals = lv; LP.cols = lc; LP.n_chunks = R.n_chunks; LP.dummy = (unsigned)R.nnz - 1u;
#ifdef CPG_GEN_HEADER
        // the generated executor has every count / offset baked in; it only needs the per-lane
        // output slots as a 16-bit table
        const unsigned n_rows16 = (((unsigned)R.n_chunks + 3u) & ~3u) * 64u;
        unsigned short *lr = (unsigned short *)(lds + lds_off); lds_off += (size_t)(n_rows16 / 4u);
        for (unsigned t = t0; t < n_rows16; t += nt) lr[t] = cpgw::gld(R.rows_gen, t);
        LP.rows16 = lr; LP.ctab = nullptr; LP.desc = nullptr;
#else
        unsigned *ld = (unsigned *)(lds + lds_off);         lds_off += (size_t)R.n_chunks * 32;
        int *lt = (int *)(lds + lds_off);                   lds_off += (size_t)R.n_chunks * 2;
        for (unsigned t = t0; t < (unsigned)R.n_chunks * 64u; t += nt) ld[t] = cpgw::gld(R.desc, t);
        for (unsigned t = t0; t < (unsigned)R.n_chunks * 4u; t += nt) lt[t] = cpgw::gld(R.ctab, t);
        LP.ctab = lt; LP.desc = ld; LP.rows16 = nullptr;
#endif
    }
    cpgw::block_sync();
    double *w = lds + lds_off + (size_t)cpgw::wave_in_block() * G * ldw;
    (void)wave_global;
    const long long ngroups = (Bt.B + G - 1) / G;
    const double rho_eq = 1e3 * F.rho, rho_in = F.rho, rho_fr = 1e-6;
    const double ri_eq = 1.0 / rho_eq, ri_in = 1.0 / rho_in, ri_fr = 1.0 / rho_fr;
    // family constants kept in registers: row class and final LDS slot of every owned entry
    signed char ct_reg[NSZ];
    unsigned short fpx[NSX], fpz[NSZ];
#pragma unroll
    for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * (unsigned)s; fpx[s] = (i < (unsigned)F.n) ? cpgw::gld(F.fpos, i) : 0; }
#pragma unroll
    for (int s = 0; s < NSZ; s++) {
        const unsigned i = (unsigned)lane + 64u * (unsigned)s;
        ct_reg[s] = (i < (unsigned)F.m) ? cpgw::gld(F.ctype, i) : 0;
        fpz[s] = (i < (unsigned)F.m) ? cpgw::gld(F.fpos, (unsigned)F.n + i) : 0;
    }

    for (;;) {
        // per-instance copy of the lane id (see cpgw::opaque): nothing derived from it is invariant across
        // instances, so the optimiser cannot hoist the dozens of per-lane addresses of the set-up, update and
        // retrieval code into the kernel prologue and keep them alive (spilled) across the hot loop
        const int lane = cpgw::opaque(lane0);
        cpgw::assume((unsigned)lane < 64u);
        unsigned ig = 0;
        if (lane == 0) ig = cpgw::atomic_next(Bt.counter);
        ig = (unsigned)cpgw::read_first_lane((int)ig);
        if ((long long)ig >= ngroups) break;

        InstT I[G];
        CheckOut co[G];
        int n_open = 0;
#ifdef CPG_GEN_HEADER
        // The generated executor gathers in partial steps with ALL lanes (the coefficient of the idle
        // lanes is forced to zero): every slot they can touch must hold a finite number, so the work
        // vector starts from zeros for every instance (no stale NaN / Inf from LDS or from a
        // diverged previous instance).
        for (unsigned t = (unsigned)lane; t < (unsigned)(G * ldw); t += 64u) w[t] = 0.0;
        cpgw::lds_order();
#endif
#pragma unroll
        for (int g = 0; g < G; g++) {
            const long long b = (long long)ig * G + g;
            I[g].b = b < Bt.B ? b : -1;
            I[g].done = b < Bt.B ? 0 : 1;
            co[g].prim_res = 0; co[g].dual_res = 0; co[g].obj = 0; co[g].status = 11;
            const double *theta = Bt.theta + (size_t)(b < Bt.B ? b : 0) * U.np_var;
            const bool bad = canonicalise<NSX, NSZ, NV>(F, U, theta, I[g], lane);
            if (Bt.state_in && S.warm_starting && b < Bt.B)
                load_state<NSX, NSZ>(F, Bt.state_in + (size_t)b * (size_t)(F.n + 2 * F.m + 1), I[g].x, I[g].z, I[g].y, lane);
            if (Bt.state_in && b < Bt.B && !bad) {
                // the workspace of a sequential caller keeps the rho its last adapt_rho left (and the factor that
                // goes with it); this kernel's factor only serves the family's rho
                const double rho_in = cpgw::gld(Bt.state_in + (size_t)b * (size_t)(F.n + 2 * F.m + 1), (unsigned)(F.n + 2 * F.m));
                if (__builtin_expect(cpgw::dmin2(cpgw::dmax2(rho_in, CPG_RHO_MIN), CPG_RHO_MAX) != F.rho, 0)) {
                    hand_over<NSX, NSZ>(F, Bt, I[g].x, I[g].z, I[g].y, b, lane, 0, rho_in);
                    I[g].done = 1;
                }
            }
            if (__builtin_expect(bad && !I[g].done, 0)) {
                // a row changed class: the family's factor does not serve this instance.  It is only flagged
                // here (its solution rows stay unwritten); the host layer sends it through the per-instance
                // factor path and overwrites every output of the row (cvxpygen_amd/runtime.py)
                if (lane == 0) { Bt.obj[b] = NAN; Bt.iter[b] = 0; Bt.status[b] = -2; Bt.pri_res[b] = 0.0; Bt.dua_res[b] = 0.0; }
                I[g].done = 1;
            }
            n_open += I[g].done ? 0 : 1;
        }

        int iter = 0;
        // right-hand side of the KKT system and its solution (the hot code: no termination test here)
        auto rhs_and_solve = [&]() __attribute__((always_inline)) {
            const int lane_outer = lane;
            const int lane = cpgw::opaque(lane_outer);   // per-iteration copy, see cpgw::opaque
            cpgw::assume((unsigned)lane < 64u);
            signed char ct[NSZ];
#pragma unroll
            for (int s = 0; s < NSZ; s++) ct[s] = (signed char)cpgw::opaque((int)ct_reg[s]);
#pragma unroll
            for (int g = 0; g < G; g++) {
                double *wg = w + g * ldw;
#pragma unroll
                for (int s = 0; s < NSX; s++) {
                    const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                    if (i < n_c) wg[i] = F.sigma * I[g].x[s] - SharedCtx<NSX, NSZ, NV>{F, sh, shu, I[g], wg, lane}.q(s, i);
                    CPG_FENCE_EVERY(s);
                }
#pragma unroll
                for (int s = 0; s < NSZ; s++) {
                    const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                    const int cts = CPG_ROW_CLASS(s);
                    const double ri = cts == 1 ? ri_eq : (cts == 0 ? ri_in : ri_fr);
                    if (i < m_c) wg[n_c + i] = I[g].z[s] - ri * I[g].y[s];
                    CPG_FENCE_EVERY(s);
                }
            }
            cpgw::lds_order();
#ifdef CPG_GEN_HEADER
            if (LDSPROG) run_program_gen<G>(LP.vals, LP.cols, LP.rows16, w, ldw, lane);
#else
            if (LDSPROG) run_program_lds<G>(LP, w, ldw, lane);
#endif
            else if (G == 1 && F.kkt_stream.n_pairs > 0) run_program_stream(F.kkt_stream, w, lane);
            else run_program<G>(F.kkt, w, ldw, lane);
        };
        // relaxation, projection on [l, u], dual update; the instantiation of a checked iteration also
        // keeps the steps delta_x / delta_y (registers) for OSQP's infeasibility tests
        auto update = [&](auto stash_c, double (&dxr)[G][NSX], double (&dyr)[G][NSZ]) __attribute__((always_inline)) {
            constexpr bool STASH = decltype(stash_c)::value;
            const int lane_outer = lane;
            const int lane = cpgw::opaque(lane_outer);
            cpgw::assume((unsigned)lane < 64u);
            signed char ct[NSZ];
#pragma unroll
            for (int s = 0; s < NSZ; s++) ct[s] = (signed char)cpgw::opaque((int)ct_reg[s]);
#pragma unroll
            for (int g = 0; g < G; g++) {
                const double *wg = w + g * ldw;
#pragma unroll
                for (int s = 0; s < NSX; s++) {
                    const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                    if (STASH) dxr[g][s] = 0.0;
                    if (i < n_c) {
                        const double xn = F.alpha * wg[fpx[s]] + (1.0 - F.alpha) * I[g].x[s];
                        if (STASH) dxr[g][s] = xn - I[g].x[s];
                        I[g].x[s] = xn;
                    }
                    CPG_FENCE_EVERY(s);
                }
#pragma unroll
                for (int s = 0; s < NSZ; s++) {
                    const unsigned i = (unsigned)lane + 64u * (unsigned)s;
                    if (STASH) dyr[g][s] = 0.0;
                    if (i < m_c) {
                        const int cts = CPG_ROW_CLASS(s);
                        const double rv = cts == 1 ? rho_eq : (cts == 0 ? rho_in : rho_fr);
                        const double ri = cts == 1 ? ri_eq : (cts == 0 ? ri_in : ri_fr);
                        const double zp = I[g].z[s], yp = I[g].y[s];
                        const double zt = (zp - ri * yp) + ri * wg[fpz[s]];
                        const double zr = F.alpha * zt + (1.0 - F.alpha) * zp;
                        // projection on [l, u]: equality rows have l = u, all others l = -inf
                        const double uu = SharedCtx<NSX, NSZ, NV>{F, sh, shu, I[g], wg, lane}.u(s, i);
                        const double zn = cts == 1 ? uu : cpgw::dmin2(zr + ri * yp, uu);
                        const double dyv = rv * (zr - zn);
                        I[g].z[s] = zn; I[g].y[s] = yp + dyv;
                        if (STASH) dyr[g][s] = dyv;
                    }
                    CPG_FENCE_EVERY(s);
                }
            }
            cpgw::lds_order();
        };
        // events: the multiples of check_termination (termination test), the multiples of adaptive_rho_interval
        // (OSQP >= 1.0's adapt_rho) and max_iter (osqp_solve)
        const int chk_int = S.check_termination;
        const int ad_int = (S.adaptive_rho && S.adaptive_rho_interval > 0) ? S.adaptive_rho_interval : 0;
#pragma nounroll
        while (n_open > 0) {
            double dxr[G][NSX], dyr[G][NSZ];
            if (iter < S.max_iter) {
                int next_ev = S.max_iter;
                if (chk_int > 0) { const int c = (iter / chk_int + 1) * chk_int; if (c < next_ev) next_ev = c; }
                if (ad_int > 0) { const int c = (iter / ad_int + 1) * ad_int; if (c < next_ev) next_ev = c; }
                // plain iterations in their own loop: the termination test, its products and their
                // register demand stay outside the hot code
#pragma nounroll
                for (;;) {
                    iter++;
                    rhs_and_solve();
                    if (iter >= next_ev) break;
                    update(std::false_type{}, dxr, dyr);
                }
                update(std::true_type{}, dxr, dyr);
            } else {   // max_iter <= 0: the test runs on the initial iterates
#pragma unroll
                for (int g = 0; g < G; g++) {
#pragma unroll
                    for (int s = 0; s < NSX; s++) dxr[g][s] = 0.0;
#pragma unroll
                    for (int s = 0; s < NSZ; s++) dyr[g][s] = 0.0;
                }
            }
            const bool last = iter >= S.max_iter;
            const bool can_check = last || (chk_int > 0 && iter % chk_int == 0);
            const bool adapt = ad_int > 0 && iter > 0 && iter % ad_int == 0;
            // ---- termination test / bookkeeping
            // (verdicts in named variables selected by value: a local array indexed by `pass` would live in scratch)
            InfeasVerdict iv_exact[G], iv_approx[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                iv_exact[g] = iv_approx[g] = InfeasVerdict{false, false};
                if (I[g].done || !can_check) continue;
                double *wg = w + g * ldw;
#pragma nounroll
                for (int pass = 0; pass < (last ? 2 : 1); pass++) {
                    const InfeasVerdict v = infeasibility_tests<NSX, NSZ, SharedCtx<NSX, NSZ, NV>, RegDelta<NSX>, RegDelta<NSZ>>(
                        F, SharedCtx<NSX, NSZ, NV>{F, sh, shu, I[g], wg, lane}, ct_reg, S,
                        RegDelta<NSX>{dxr[g]}, RegDelta<NSZ>{dyr[g]}, wg, lane, pass == 1);
                    if (pass == 0) iv_exact[g] = v; else iv_approx[g] = v;
                }
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (I[g].done) continue;
                CheckOut o = co[g];
                double *wg = w + g * ldw;
                ScaledNorms sn;
                double rho_ws = F.rho;          // rho of the workspace this instance leaves behind
                bool rho_changed = false;
#pragma nounroll
                for (int pass = 0; pass < 2; pass++) {
                    if (pass == 1 && !(o.status == 11 && last)) break;
                    // update_info (+ check_termination): at an adaptation point that is not a checked iteration
                    // only the norms are used (osqp_solve updates the info for adapt_rho, no test)
                    const CheckOut oc = check<NSX, NSZ, SharedCtx<NSX, NSZ, NV>, NoDelta, NoDelta>(
                        F, SharedCtx<NSX, NSZ, NV>{F, sh, shu, I[g], wg, lane}, ct_reg, S, I[g].x, I[g].z, I[g].y,
                        NoDelta{}, NoDelta{}, pass == 0 ? iv_exact[g] : iv_approx[g], wg, lane, pass == 1,
                        (pass == 0 && adapt) ? &sn : nullptr);
                    if (can_check) o = oc;
                    if (pass == 0 && adapt && o.status == 11) {
                        // adapt_rho: this kernel's factor belongs to the family's rho -- an instance whose estimate
                        // leaves [rho / tolerance, rho * tolerance] continues on a factor of its own (hand-over)
                        const double rn = rho_estimate(sn, F.rho);
                        if (rn > F.rho * S.adaptive_rho_tolerance || rn < F.rho / S.adaptive_rho_tolerance) { rho_ws = rn; rho_changed = true; }
                    }
                }
                if (o.status == 11 && last) o.status = 7;
                co[g] = o;
                if (o.status != 11) {
                    finalize<NSX, NSZ>(F, Bt, I[g].x, I[g].z, I[g].y, I[g].dconst, I[g].b, wg, lane, iter, o, rho_ws);
                    I[g].done = 1; n_open--;
                } else if (rho_changed) {
                    hand_over<NSX, NSZ>(F, Bt, I[g].x, I[g].z, I[g].y, I[g].b, lane, iter, rho_ws);
                    I[g].done = 1; n_open--;
                }
            }
        }
    }
}

}  // namespace cpg
