/*
 * cpg_hip.h -- C-ABI of the MI355X batched-solve backend for cvxpygen-generated solvers.
 *
 * This shared library (libcpg_hip.so, built by hipcc for gfx950) is what a binding loads in place
 * of the reference's pybind11 extension `cpg_module` (emitted by cvxpygen/utils.py:1163-1412,
 * declared in cvxpygen/templates/cpg_module.hpp.jinja2:1-102).  Plain pointers and sizes only; no
 * C++/torch types.  All functions return 0 on success or a negative CPG_E_* code; the message of
 * the last failure on the calling thread is available from cpg_hip_last_error().
 *
 * Mapping to the reference interface (one instance per call there, a batch here):
 *   cpg_module.solve(upd, par)               utils.py:1194-1270 -> cpg_hip_solve_batch[_device]
 *     cpg_update_<param>(idx, val)           utils.py:904-935   -> theta_var rows (coalesced load)
 *     cpg_canonicalize_<p>()                 utils.py:279-294   -> in-kernel CSR product, see
 *                                                                  cpg_osqp_update_t
 *     osqp_update_data_vec(q, l, u)          solvers/osqp.py:39-59  -> in-kernel rescale with D,E,c
 *     osqp_solve(&solver)                    solvers/osqp.py:62     -> ADMM kernel
 *     cpg_retrieve_prim/dual/info            utils.py:950-985   -> prim / dual / obj / iter / ...
 *   cpg_module.set_solver_default_settings() utils.py:1070-1076 -> cpg_hip_set_default_settings
 *   cpg_module.set_solver_<name>(v)          utils.py:1077-1084,1407-1410 -> cpg_hip_set_setting
 *   static workspace of the extension        utils.py:470-689   -> opaque cpg_handle_t (no globals)
 *   CPG_Info.status (OSQP string)            utils.py:982       -> int32 code, cpg_hip_status_string
 */
#ifndef CPG_HIP_H
#define CPG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPG_OK 0
#define CPG_E_BADARG (-1)
#define CPG_E_HIP (-2)
#define CPG_E_NOMEM (-3)
#define CPG_E_UNSUPPORTED (-4)

/* per-instance status codes: OSQP 1.0 numbering, plus one backend-specific code */
#define CPG_STATUS_SOLVED 1
#define CPG_STATUS_SOLVED_INACCURATE 2
#define CPG_STATUS_PRIMAL_INFEASIBLE 3
#define CPG_STATUS_PRIMAL_INFEASIBLE_INACCURATE 4
#define CPG_STATUS_DUAL_INFEASIBLE 5
#define CPG_STATUS_DUAL_INFEASIBLE_INACCURATE 6
#define CPG_STATUS_MAX_ITER_REACHED 7
#define CPG_STATUS_NON_CVX 9
#define CPG_STATUS_UNSOLVED 11
/* INTERNAL: set by the shared-factor kernel on an instance it handed to the per-instance factor kernel queued
 * behind it (cpg_hip_set_handover); that kernel overwrites it before the stream is idle again */
#define CPG_STATUS_HANDED_OVER (-3)
/* a constraint row changed class (equality <-> inequality <-> free) w.r.t. code generation time:
 * the shared KKT factor is not valid for this instance (the reference refactors here,
 * osqp_update_data_vec -> update_rho_vec).  INTERNAL to the two-handle protocol: the host layer
 * (cvxpygen_amd/runtime.py, BatchSolver._resolve_class_changes) re-solves such instances through the
 * refactor-mode handle, whose kernel classifies rows per instance; users never see this status. */
#define CPG_STATUS_NEEDS_REFACTOR (-2)

typedef struct cpg_solver_s *cpg_handle_t;

/* A "solve program": sequence of sparse phases  w[r] <- sum_k val_k * w[col_k]  cut into chunks
 * of 64 lane-tasks (see cvxpygen_amd/solve_program.py for the layout). */
typedef struct {
    int32_t n_chunks;
    int32_t n_steps;
    const int32_t *hdr;   /* [n_chunks][4]: len, log2(lanes per row), flags, first step */
    const uint16_t *rows; /* [n_chunks][64] */
    const double *vals;   /* [n_steps][64] */
    const uint16_t *cols; /* [n_steps][64] */
} cpg_program_t;

/* The same program without padding (cvxpygen_amd/solve_program.py::RaggedProgram): small enough
 * to be kept resident in LDS, one copy per workgroup.  Optional (n_chunks == 0: not provided). */
typedef struct {
    int32_t n_chunks;
    int32_t nnz;
    const int32_t *ctab;  /* [n_chunks][4]: max len, reduction stages, first entry, kind (bit 0: rows occupy a
                           * variable number of adjacent lanes, segmented reduction; bit 1: rows accumulate) */
    const uint32_t *desc; /* [n_chunks][64]: output slot | (entries of the lane << 16) [| segment mask << 28] */
    const double *vals;   /* [nnz] */
    const uint16_t *cols; /* [nnz] */
} cpg_ragged_t;

typedef struct {
    int32_t rows;
    int32_t nnz;
    const int32_t *ptr; /* [rows + 1] */
    const int32_t *idx; /* [nnz] column = position in theta_var */
    const double *val;  /* [nnz] */
} cpg_csr_t;

/* Everything fixed at code-generation time for one OSQP problem family (host pointers; copied).
 * All vectors / index lists are in DEVICE ORDER: the binding permutes the canonical x entries and
 * constraint rows so that those depending on user parameters come first (cvxpygen_amd/runtime.py). */
typedef struct {
    int32_t n;    /* canonical variables */
    int32_t m;    /* canonical constraints = n_eq + n_ineq */
    int32_t n_eq;
    int32_t is_maximization;
    double sigma, alpha, rho;
    const double *D; /* [n] Ruiz column scaling at code-generation parameters */
    const double *E; /* [m] */
    double c;        /* cost scaling */
    const int8_t *ctype; /* [m] row class at code-generation time: -1 free, 0 inequality, 1 equality */
    int32_t n_slots;       /* LDS work-vector length (>= n + m) required by `kkt` */
    const uint16_t *fpos;  /* [n + m] slot that holds entry i after `kkt` ran */
    int32_t n_vary_x;      /* entries [0, n_vary_x) of q may depend on user parameters */
    int32_t n_vary_z;      /* rows [0, n_vary_z) of l / u may depend on user parameters */
    cpg_program_t kkt;     /* w <- K^-1 w on w = [x-part (n); z-part (m)] */
    cpg_program_t A_rows;  /* natural layout: (A v)_i,  v = w[0..n)   */
    cpg_program_t P_rows;  /* natural layout: (P v)_j,  v = w[0..n)   */
    cpg_program_t At_rows; /* natural layout: (A' v)_j, v = w[n..n+m) */
    cpg_ragged_t kkt_ragged; /* compact form of `kkt` for the LDS-resident path (optional) */
    int32_t n_prim;        /* user primal entries */
    const int32_t *prim_idx; /* [n_prim] indices into x */
    int32_t n_dual;
    const int32_t *dual_idx; /* [n_dual] indices into y */
    const int32_t *ord;      /* [n + m] canonical index of the entry at device position i (x entries, then
                              * rows); NULL = identity.  Only the state buffers of the *_state entry points
                              * (canonical order) go through it. */
} cpg_osqp_family_t;

/* Which user parameters vary across the batch and how the canonical VECTORS depend on them
 * (cpg_update_<param> + cpg_canonicalize_q/l/u/d + osqp_update_data_vec in one description):
 *     q_scaled = q_base + map_q @ theta_var      (rows pre-multiplied by c * D)
 *     u_scaled = u_base + map_u @ theta_var      (rows pre-multiplied by E; +1e30 rows included)
 *     d        = d_base + map_d @ theta_var
 * The lower bound is implied by the row class: equality rows (ctype 1) have l = u, every other row
 * has l = -infinity -- the only two kinds of rows the reference's OSQP canonical form contains
 * (cvxpygen/solvers/_interface.py:62-79); the binding refuses families that violate this. */
typedef struct {
    int32_t np_var; /* doubles per instance in theta_var */
    const double *q_base; /* [n] */
    const double *u_base; /* [m] */
    double d_base;
    cpg_csr_t map_q, map_u, map_d;
} cpg_osqp_update_t;

/* Per-instance refactorisation path (parameters entering P or A): shared structural tables built by
 * cvxpygen_amd/refactor_plan.py + the UNSCALED canonicalisation of every canonical parameter over
 * the updated user parameters.  Replaces, for a batch, the reference's
 * cpg_canonicalize_P/_A + osqp_update_data_mat + osqp_update_data_vec (solvers/osqp.py:20-61).
 * Canonical (not device) ordering throughout. */
typedef struct {
    int32_t nnzP, nnzA, nnzL, scaling_iters;
    /* equilibration views */
    const int32_t *Ap, *Ai;             /* A in CSC */
    const int32_t *Arp, *Aent, *Acol;   /* row view of A: entry index into CSC order, column */
    const int32_t *Pp, *Pi;             /* upper-triangular P in CSC */
    const int32_t *Prp, *Pent, *Pcol;   /* full symmetric row view of P */
    /* factor */
    const int32_t *Lcol;                /* [nnzL] column of every entry of L */
    const int32_t *ksrc_kind, *ksrc_idx;/* [nnzL + n + m] where the KKT value of a destination comes from */
    int32_t fac_chunks, fac_triples;
    const int32_t *fac_ctab;            /* [fac_chunks][4]: max len, last-of-level, first triple, #tasks */
    const uint32_t *fac_task, *fac_len; /* [fac_chunks][64] */
    const uint32_t *fac_a, *fac_b, *fac_k; /* [fac_triples]: positions of L_ik, L_jk and column k */
    /* substitution program (ragged layout), value sources instead of values */
    int32_t sol_chunks, sol_nnz, sol_slots;
    const int32_t *sol_ctab; const uint32_t *sol_desc; const uint16_t *sol_cols;
    const int32_t *sol_kind, *sol_idx;  /* [sol_nnz]: 0 zero, 1 one, 2 -L[idx], 3 1/d[idx] */
    const uint16_t *sol_fpos;           /* [n + m] */
    /* canonicalisation over theta_var (unscaled) */
    int32_t np_var;
    const double *P_base, *A_base, *q_base, *u_base; double d_base;
    cpg_csr_t map_P, map_A, map_q, map_u, map_d;
    /* [n] unscaled q of the code-generation-time workspace: the reference applies
     * osqp_update_data_mat BEFORE osqp_update_data_vec (cvxpygen/solvers/osqp.py:20-59), so the
     * cost scaling of the re-equilibration sees the old q, never the instance's new one */
    const double *q_setup;
    /* Shared-matrix mode (no varying parameter enters P or A; instances own a factor only because their rho differs:
     * OSQP's adapt_rho, or a row that changed class): the workspace's equilibrated matrices and scaling -- Ps [nnzP],
     * As [nnzA] (c D P D, E A D), D [n], E [m], c -- in canonical order; q_base / u_base / map_q / map_u are then
     * PRE-SCALED (c D q, E u) as in cpg_osqp_update_t and the kernel skips canonicalise-P/A and the re-equilibration
     * (there was no osqp_update_data_mat). */
    int32_t shared_mats;
    const double *Ps, *As, *D, *E;
    double c;
} cpg_osqp_refactor_t;

/* RESIDENT per-instance factor kernel (cvxpygen_amd/csrc/cpg_osqp_resident.h, cvxpygen_amd/resident_plan.py): the same
 * path as cpg_osqp_refactor_t -- osqp_update_data_mat + osqp_solve per instance (cvxpygen/solvers/osqp.py:20-62) -- with the
 * instance's factor kept on the CU.  Tables next to those of cpg_osqp_refactor_t:
 *   f_*     the numeric LDL' schedule of the refactor tables FOLLOWED BY the schedule that inverts the diagonal blocks of
 *           merged level groups; all positions are absolute in  fac = [M (nnzL) | 1/d (n + m) | X (nnzX) | 1.0 | 0.0];
 *           f_task: destination, bit 31 = pivot (store the reciprocal)
 *   sol_*   the MERGED substitution program; sol_kind 0 zero, 1 one, 2 -L[idx] (column sol_lcol), 3 1/d[idx], 4 X[idx]
 *   rows_*  ragged row programs of the termination test's products on the work vector; `ent`: entry of A (of upper-
 *           triangular P) behind every coefficient, -1 padding */
typedef struct {
    int32_t n_chunks, nnz;
    const int32_t *ctab; const uint32_t *desc; const uint16_t *cols;
    const int32_t *ent;
} cpg_rows_program_t;
typedef struct {
    int32_t nnzX, fac_chunks, fac_triples;
    const int32_t *f_ctab;                  /* [fac_chunks][4]: steps, level complete, first triple, log2 lanes per task */
    const uint32_t *f_task, *f_len;         /* [fac_chunks][64] */
    const uint32_t *f_a, *f_b, *f_k;        /* [fac_triples] */
    int32_t sol_chunks, sol_nnz, sol_slots;
    const int32_t *sol_ctab; const uint32_t *sol_desc; const uint16_t *sol_cols;
    const int32_t *sol_kind, *sol_idx, *sol_lcol;
    cpg_rows_program_t rows_A, rows_P, rows_At;
    int32_t out_ax, out_px, out_aty;        /* first work-vector slot of A x, P x, A' y */
} cpg_osqp_resident_t;

/* QP adjoint (gradient=True in the reference): transposed canonical maps over ALL user parameters.
 * For parameter column c, entries tptr[c] .. tptr[c+1]: kind 0 q[idx], 1 l[idx], 2 u[idx],
 * 3 P entry idx, 4 A entry idx, with coefficient tcoef (rows of the reference's canon_<p>_map,
 * cvxpygen/writer.py:268-311). */
typedef struct {
    int32_t NP;
    const int32_t *Pcolidx;  /* [nnzP] column of every stored P entry */
    const int32_t *Acolidx;  /* [nnzA] */
    const int32_t *tptr;     /* [NP + 1] */
    const int32_t *tkind, *tidx;
    const double *tcoef;
} cpg_osqp_gradient_t;

/* Everything fixed at code-generation time for one CONIC problem family solved by the
 * interior-point kernel (reference: the Clarabel path, cvxpygen/solvers/clarabel.py:19-46, 133-204):
 *   minimise 1/2 x'Px + q'x + d   s.t.   Ax + s = b,  s in K,
 * rows ordered zero cone, nonnegative cone, second-order cones, PSD cones, exponential cones, three-dimensional power
 * cones -- every cone type the reference's `cones` array can hold (clarabel.py:133-155, 308-323), in the
 * order in which cvxpy stacks the rows for this solver.  (The reference lists the exponential cones AHEAD of the
 * second-order cones, clarabel.py:316-319: with both kinds present its cones do not match its rows; with one
 * kind the orders coincide.)  The reference builds a new solver per solve
 * (clarabel_DefaultSolver_new, clarabel.py:201-204), so canonicalisation, equilibration and every
 * factorisation happen per instance inside the kernel; the tables below are the family's fixed
 * patterns and schedules (cvxpygen_amd/conic_plan.py).  Natural order (no device permutation). */
typedef struct {
    int32_t n, m, is_maximization;
    int32_t n_zero, n_nonneg, n_soc;
    const int32_t *soc_dims;            /* [n_soc] */
    int32_t nnzP, nnzA, nnzL;
    const int32_t *Ap, *Ai;             /* A in CSC (row indices) */
    const int32_t *Arp, *Aent, *Acol;   /* A by rows: entry number in CSC order, column */
    const int32_t *Pp, *Pi;             /* upper-triangular P in CSC */
    const int32_t *Prp, *Pent, *Pcol;   /* full symmetric row view of P */
    /* factor of K = [[P + eps I, A'], [A, -W'W - eps I]] (same table layout as cpg_osqp_refactor_t);
     * ksrc_kind: 1 P entry idx, 2 A entry idx, 3 eps only, 5 -(W'W)_ii - eps of row idx,
     * 6 off-diagonal entry of a second-order-cone block, idx = row_i | row_j << 16,
     * 7 off-diagonal entry of an exponential / power cone's 3 x 3 block: idx = first row of the cone + (0 for (0,1),
     *   1 for (0,2), 2 for (1,2)),
     * 8 off-diagonal entry of a PSD cone's block between svec rows (i, j) and (k, l): idx = offset of the cone in the PSD store
     *   (sum over the cones before it of 11 p^2 + 3 p: NT point, factors, workspace) | p << 12 | i << 16 | j << 19 | k << 22 | l << 25 */
    const int32_t *Lcol, *ksrc_kind, *ksrc_idx;
    int32_t fac_chunks, fac_triples;
    const int32_t *fac_ctab;
    const uint32_t *fac_task, *fac_len, *fac_a, *fac_b, *fac_k;
    int32_t sol_chunks, sol_nnz, sol_slots;
    const int32_t *sol_ctab; const uint32_t *sol_desc; const uint16_t *sol_cols;
    const int32_t *sol_kind, *sol_idx;
    const uint16_t *sol_fpos;
    /* canonicalisation over theta_var: p = base + map @ theta_var  (cpg_canonicalize_<p>, utils.py:279-294) */
    int32_t np_var;
    const double *P_base, *A_base, *q_base, *b_base; double d_base;
    cpg_csr_t map_P, map_A, map_q, map_b, map_d;
    int32_t n_prim; const int32_t *prim_idx;   /* user primal entries: indices into x */
    int32_t n_dual; const int32_t *dual_idx;   /* user dual entries: indices into z */
    /* exponential cones {(x, y, z): y exp(x / y) <= z, y > 0} and power cones {x^a y^(1 - a) >= |z|, x, y >= 0}: three rows
     * each, behind the second-order cones (ClarabelExponentialConeT / ClarabelPowerConeT(a), clarabel.py:136-147) */
    int32_t n_exp, n_pow;
    const double *pow_alpha;            /* [n_pow], each in (0, 1) */
    /* PSD cones of matrix order psd_dims[k] <= 8 (ClarabelPSDTriangleConeT, clarabel.py:138, 146): psd_dims[k] (psd_dims[k] + 1) / 2
     * rows each -- the upper triangle column by column, off-diagonal entries times sqrt 2 --, between the second-order and the
     * exponential cones */
    int32_t n_psd;
    const int32_t *psd_dims;            /* [n_psd] */
} cpg_conic_family_t;

/* ---- lifecycle ---------------------------------------------------------------------------- */
int cpg_hip_device_count(int *count);
int cpg_hip_create_osqp(const cpg_osqp_family_t *family, int device, cpg_handle_t *out);
/* conic interior-point handle; solve with cpg_hip_solve_batch[_device]; `status` then carries
 * Clarabel's SolverStatus integers (1 solved, 2 primal infeasible, 3 dual infeasible, 4 / 5 / 6 their
 * "almost" forms at the reduced tolerances after an error or at the iteration limit, 7 maximum
 * iterations, 9 numerical error, 10 insufficient progress) and the setting names are those of
 * cvxpygen/solvers/clarabel.py:63-119.  In a library compiled for one conic family
 * (cvxpygen_amd.codegen.build_conic_library) a handle of exactly that family runs the generated executor of
 * its substitution program with the family's dimensions compiled in, and walks the patterns of P and A through the
 * library's generated row words (cvxpygen_amd.codegen.conic_row_tables; guarded by a hash of the words rebuilt from
 * `family`); cpg_hip_get_setting reports it as "generated_executor" / "specialised_kernel" (1.0 / 0.0); any other family
 * runs the table-driven path.  Settings of the reference's Clarabel interface this backend cannot honour are REFUSED at
 * any value but their default -- cpg_hip_set_setting returns CPG_E_UNSUPPORTED for time_limit (finite), direct_kkt_solver
 * (0), presolve_enable (0) -- instead of being accepted and ignored. */
int cpg_hip_create_clarabel(const cpg_conic_family_t *family, int device, cpg_handle_t *out);
int cpg_hip_destroy(cpg_handle_t h);
const char *cpg_hip_last_error(void);
const char *cpg_hip_status_string(int32_t status);

/* ---- settings (reference: reset to defaults, then apply kwargs, on every solve) -------------- */
int cpg_hip_set_default_settings(cpg_handle_t h);
int cpg_hip_set_setting(cpg_handle_t h, const char *name, double value);
int cpg_hip_get_setting(cpg_handle_t h, const char *name, double *value);
/* OSQP settings the generated shim offers no setter for (they are not among cvxpygen/solvers/osqp.py:102-115):
 * "adaptive_rho" (0/1), "adaptive_rho_interval" (iterations), "adaptive_rho_tolerance", "check_dualgap" (0/1).
 * Their values are the defaults of the OSQP library the generated code is linked with.  Defaults = OSQP >= 1.0 (1, 50, 5.0, 1): like every other
 * OSQP setting they are what osqp_set_default_settings restores on each cpg_solve of the reference
 * (solvers/osqp.py:100-101), so cpg_hip_set_default_settings restores the values given here. */
int cpg_hip_set_build_option(cpg_handle_t h, const char *name, double value);
/* Hybrid execution of rho adaptation for a family whose matrices do not vary across the batch.  `h` is a
 * shared-factor handle (cpg_hip_set_update), `per_instance` a handle of the same family carrying per-instance
 * factor tables in shared-matrix mode (cpg_hip_set_refactor).  A solve on `h` with adaptive_rho on then runs
 * TWO kernels back to back on h's stream: the shared-factor kernel serves every instance until OSQP's adapt_rho
 * changes its rho (iterations 1 .. interval always; for as long as the estimate stays inside [rho / tolerance,
 * rho * tolerance] after that) and hands the others -- workspace, iteration count -- to the per-instance factor
 * kernel, which refactors K for the new rho and continues.  per_instance == NULL unlinks.
 *
 * !! A handle from cpg_hip_create_osqp has adaptive_rho ON (the library default the reference links).  Solving on a
 * !! shared-factor handle (cpg_hip_set_update) with it on and NO handle linked here is refused with CPG_E_UNSUPPORTED by
 * !! every solve entry point (cpg_hip_solve_batch, _device, the pipelined one): a shared factor cannot follow a rho change,
 * !! so those instances could only come back unsolved.  Either link a per-instance factor handle, or switch adaptation
 * !! off (cpg_hip_set_build_option(h, "adaptive_rho", 0)), or opt in to the flagging protocol with
 * !! cpg_hip_set_build_option(h, "flag_rho_changes", 1): the solve then returns CPG_OK and every instance whose rho would
 * !! have changed carries status CPG_STATUS_NEEDS_REFACTOR (-2) and NO solution -- the caller must re-solve those. */
int cpg_hip_set_handover(cpg_handle_t h, cpg_handle_t per_instance);
/* split of the most recent solve on `h`: kernel time of the two phases and the number of instances handed over
 * (0 / 0 when the solve ran one kernel) */
int cpg_hip_last_phase_ms(cpg_handle_t h, float *ms_shared, float *ms_per_instance, int64_t *n_handed_over);

/* ---- which parameters are updated (sticky until changed) -------------------------------------- */
int cpg_hip_set_update(cpg_handle_t h, const cpg_osqp_update_t *upd);

/* per-instance refactorisation path; after this call solves go through it until cpg_hip_set_update
 * is called again */
int cpg_hip_set_refactor(cpg_handle_t h, const cpg_osqp_refactor_t *rf);

/* cpg_hip_set_refactor(h, rf), and -- when this library carries the generated resident executor of exactly this
 * family (cvxpygen_amd.codegen.resident_header; the merged program's fingerprint decides) -- the resident kernel's
 * tables: solves then run cpg_osqp_resident.h instead of the streaming kernel.  cpg_hip_get_setting(h,
 * "resident_executor") reports which (1.0 / 0.0; 0.0 for a team library); any other library keeps the streaming kernel and returns CPG_OK.
 * A library generated with the TEAM executor (cvxpygen_amd.codegen.team_header: families whose merged program does not
 * fit the registers of one wavefront, e.g. an MPC with every parameter per instance) takes the same tables and runs
 * cpg_osqp_team.h -- one workgroup of W wavefronts per instance, the program split over them; "team_executor" reports the
 * W in use (0.0: not -- no team plan, the streaming placement was asked for, or the team's LDS slice exceeds the device limit).  The tables must be those of a plan built for that W and group limit (the header's
 * `// CPG_GENT_GROUPS` line; cvxpygen_amd.runtime reads it back): another plan is refused by fingerprint and the handle
 * keeps the streaming kernel. */
int cpg_hip_set_resident(cpg_handle_t h, const cpg_osqp_refactor_t *rf, const cpg_osqp_resident_t *rs);

/* adjoint tables; requires cpg_hip_set_refactor on the same handle (canonical ordering) */
int cpg_hip_set_gradient(cpg_handle_t h, const cpg_osqp_gradient_t *g);
/* Batched cpg_gradient(): for every instance, canonical solution sol_x [B][n], sol_y [B][m] of the
 * forward solve, upstream gradient dx [B][n] on the canonical variables (the user-variable gradients
 * scattered to their canonical positions, the reference's cpg_update_d<var>) -> dtheta [B][NP],
 * gradient w.r.t. every user parameter.  Host buffers. */
int cpg_hip_gradient_batch(cpg_handle_t h, int64_t B, const double *theta_var, const double *sol_x,
                           const double *sol_y, const double *dx, double *dtheta);

/* ---- solve ---------------------------------------------------------------------------------- */
/* Host buffers: theta_var [B][np_var]; outputs prim [B][n_prim], dual [B][n_dual], obj/pri_res/
 * dua_res [B], iter/status [B].  Copies in, solves, copies out, synchronises. */
int cpg_hip_solve_batch(cpg_handle_t h, int64_t B, const double *theta_var, double *prim, double *dual,
                        double *obj, int32_t *iter, int32_t *status, double *pri_res, double *dua_res);

/* Device-resident variant: every pointer is device memory obtained from cpg_hip_malloc on the
 * handle's device.  Asynchronous on the handle's stream; pair with cpg_hip_synchronize. */
int cpg_hip_solve_batch_device(cpg_handle_t h, int64_t B, const double *d_theta_var, double *d_prim,
                               double *d_dual, double *d_obj, int32_t *d_iter, int32_t *d_status,
                               double *d_pri_res, double *d_dua_res);
/* Sequential use of one workspace (the reference's static OSQP workspace keeps its iterates and rho
 * between cpg_solve() calls; warm_starting = 1 is its default, cvxpygen/solvers/osqp.py:110):
 * state [B][n + 2 m + 1] = scaled iterates x | z | y in canonical order, then rho.  state_in NULL (or
 * warm_starting = 0) = cold start from the family's rho; state_out NULL = not wanted.  After a solve
 * without a solution the iterates in state_out are zero (osqp_solve resets them). */
int cpg_hip_solve_batch_state(cpg_handle_t h, int64_t B, const double *theta_var, const double *state_in,
                              double *state_out, double *prim, double *dual, double *obj, int32_t *iter,
                              int32_t *status, double *pri_res, double *dua_res);
int cpg_hip_solve_batch_device_state(cpg_handle_t h, int64_t B, const double *d_theta_var, const double *d_state_in,
                                     double *d_state_out, double *d_prim, double *d_dual, double *d_obj,
                                     int32_t *d_iter, int32_t *d_status, double *d_pri_res, double *d_dua_res);
/* n_batches consecutive batches of B instances each, all in HOST memory (theta_var [n_batches][B][np_var],
 * outputs likewise): H2D of batch i + 1, the solve of batch i and D2H of batch i - 1 overlap on three HIP
 * streams over two sets of device buffers, so that in steady state the PCIe transfers hide behind the
 * kernel.  Buffers from cpg_hip_host_malloc (page-locked) make the copies truly asynchronous. */
int cpg_hip_solve_batches_pipelined(cpg_handle_t h, int64_t B, int32_t n_batches, const double *theta_var,
                                    double *prim, double *dual, double *obj, int32_t *iter, int32_t *status,
                                    double *pri_res, double *dua_res);
int cpg_hip_host_malloc(cpg_handle_t h, size_t bytes, void **hptr);
int cpg_hip_host_free(cpg_handle_t h, void *hptr);
int cpg_hip_synchronize(cpg_handle_t h);
/* the handle's HIP stream (hipStream_t), for work that must be ordered behind its solves -- the RCCL
 * sends / receives of the multi-GPU result gather (cvxpygen_amd/sharding.py) */
int cpg_hip_get_stream(cpg_handle_t h, void **stream);
/* duration of the most recent solve kernel on this handle, from HIP events on its stream */
int cpg_hip_last_kernel_ms(cpg_handle_t h, float *ms);
/* launch geometry: waves per block (1..16), instances per wave (1 or 2), blocks per CU; 0 = auto */
int cpg_hip_set_launch(cpg_handle_t h, int waves_per_block, int inst_per_wave, int blocks_per_cu);
/* where the solve program lives: 0 = streamed through L2, 1 = resident in LDS (one workgroup per
 * CU; fails if it does not fit), -1 = the library's choice (family library with a generated executor:
 * LDS resident; table-driven kernels: streamed for one instance per wave, which measures faster);
 * 2 = per-instance factor handles only: the streaming executor with its shared entry words in LDS even when the
 * library carries a generated (instance / resident) executor for the family -- the kernel those replaced, kept
 * selectable for comparison;
 * 3 = shared-factor handles of a family library that carries the squad executor (cvxpygen_amd.codegen.squad_header;
 * cpg_hip_get_setting(h, "squad_executor") reports 1.0 once selected): the solve program in the REGISTERS of a workgroup of
 * four wavefronts that solves four instances at a time (csrc/cpg_osqp_squad.h) -- same results; measured slower than 1 on
 * MI355X, kept selectable for comparison; a solve fails with CPG_E_UNSUPPORTED where the library has none.
 * Replaces nothing of the reference's interface: the generated C has one placement, the CPU's (cvxpygen/solvers/osqp.py:62). */
int cpg_hip_set_program_placement(cpg_handle_t h, int in_lds);

/* ---- device memory helpers for the device-resident variant ---------------------------------------- */
int cpg_hip_malloc(cpg_handle_t h, size_t bytes, void **dptr);
int cpg_hip_free(cpg_handle_t h, void *dptr);
int cpg_hip_memcpy_h2d(cpg_handle_t h, void *dst, const void *src, size_t bytes);
int cpg_hip_memcpy_d2h(cpg_handle_t h, void *dst, const void *src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* CPG_HIP_H */
