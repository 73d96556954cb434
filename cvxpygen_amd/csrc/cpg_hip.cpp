// C-ABI of the MI355X batched-solve backend (declared in include/cpg_hip.h) and the launch code
// of the OSQP (shared factor, refactorisation, adjoint) and conic interior-point kernels.  Built by hipcc for gfx950 into libcpg_hip.so (see csrc/build.py).
#include "../../include/cpg_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "cpg_osqp_kernel.h"
#include "cpg_osqp_refactor.h"
#include "cpg_osqp_resident.h"
#include "cpg_osqp_team.h"
#include "cpg_osqp_squad.h"
#include "cpg_clarabel_kernel.h"

// ------------------------------------------------------------------------------------ runtime layer
#include <hip/hip_runtime.h>
typedef hipStream_t rt_stream_t;
typedef hipEvent_t rt_event_t;
#define RT_CHECK(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return CPG_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

static thread_local std::string g_err;
static void set_error(const std::string &s) { g_err = s; }

struct DevBuf {   // one device allocation
    void *p = nullptr;
    size_t bytes = 0;
};

struct cpg_solver_s {
    int device = 0;
    rt_stream_t stream{};
    rt_event_t ev0{}, ev1{};
    bool have_events = false;
    float last_ms = 0.f;
    std::vector<void *> owned;          // family / update buffers
    std::vector<void *> update_owned;
    cpg::DevFamily F{};
    cpg::DevUpdate U{};
    bool have_update = false;
    cpg::DevRefactor R{};
    bool refactor_mode = false;
    std::vector<void *> refactor_owned;
    cpg::DevResident Rs{};               // resident per-instance factor kernel (cpg_hip_set_resident); Rs.ok: in use
    std::vector<void *> resident_owned;
    cpg::DevGradient Gd{};
    bool have_gradient = false;
    std::vector<void *> gradient_owned;
    DevBuf g_theta, g_x, g_y, g_dprim, g_dtheta;
    cpg::DevSettings S{};
    int waves_per_block = 0, inst_per_wave = 1, blocks_per_cu = 0;
    int program_in_lds = -1;            // -1 auto, 0 stream from L2/HBM, 1 resident in LDS, 3 squad executor (program in registers)
    bool squad_ok = false;              // family library with this family's squad executor (cpg_osqp_squad.h)
    int num_cu = 256;
    size_t lds_limit = 160 * 1024;
    unsigned *d_counter = nullptr;
    int n_vary_x = 0, n_vary_z = 0;
    bool conic = false;                 // interior-point handle (cpg_hip_create_clarabel)
    bool conic_specialised = false;     // ... of the family its library was generated for (dimensions compiled in)
    cpg::DevConic C{};
    cpg::DevConicSettings CS{};
    double time_limit = 1e10; int verbose = 1, direct_kkt_solver = 1, presolve_enable = 1;   // accepted, unused
    DevBuf scratch;                     // delta_x / delta_y stash, [waves][G][n + m]
    // staging for the host-pointer entry point
    DevBuf s_theta, s_prim, s_dual, s_obj, s_pri, s_dua, s_iter, s_status, s_state_in, s_state_out;
    struct cpg_pipe_s *pipe = nullptr;  // cpg_hip_solve_batches_pipelined
    std::vector<int> rows_hdr[3];       // host copies of the chunk tables of A_rows / P_rows / At_rows (literal checks of generated kernels)
    // OSQP library defaults the generated shim has no setter for (cpg_hip_set_build_option); restored, like the
    // others, by cpg_hip_set_default_settings
    int opt_adaptive_rho = 1, opt_adaptive_rho_interval = 50, opt_check_dualgap = 1;
    double opt_adaptive_rho_tolerance = 5.0;
    // hybrid execution of rho adaptation (cpg_hip_set_handover): instances of this shared-factor handle whose
    // rho changes continue on `linked`'s per-instance factor kernel, launched behind on this handle's stream
    cpg_solver_s *linked = nullptr;
    bool flag_rho_changes = false;    // build option: a solve WITHOUT a linked handle may flag rho changes as status -2 instead of refusing
    DevBuf ho_state, ho_list;
    rt_event_t ev_mid{};
    bool two_phase_last = false;
};

// ---- runtime primitives -------------------------------------------------------------------------
static int rt_malloc(void **p, size_t bytes) {
    if (bytes == 0) bytes = 8;
    RT_CHECK(hipMalloc(p, bytes));
    return CPG_OK;
}
static int rt_free(void *p) {
    if (p) RT_CHECK(hipFree(p));
    return CPG_OK;
}
static int rt_h2d(cpg_handle_t h, void *dst, const void *src, size_t bytes) {
    if (!bytes) return CPG_OK;
    RT_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    return CPG_OK;
}
static int rt_d2h(cpg_handle_t h, void *dst, const void *src, size_t bytes) {
    if (!bytes) return CPG_OK;
    RT_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    return CPG_OK;
}
static int rt_sync(cpg_handle_t h) {
    RT_CHECK(hipStreamSynchronize(h->stream));
    return CPG_OK;
}
static int rt_set_device(int dev) {
    RT_CHECK(hipSetDevice(dev));
    return CPG_OK;
}

template <typename T>
static int upload(cpg_handle_t h, std::vector<void *> &own, const T *src, size_t count, const T **dst) {
    void *p = nullptr;
    int rc = rt_malloc(&p, count * sizeof(T));
    if (rc) return rc;
    own.push_back(p);
    if (count) { rc = rt_h2d(h, p, src, count * sizeof(T)); if (rc) return rc; }
    *dst = (const T *)p;
    return CPG_OK;
}

static int upload_program(cpg_handle_t h, const cpg_program_t &src, cpg::DevProgram *dst) {
    int rc;
    dst->n_chunks = src.n_chunks;
    if ((rc = upload<int>(h, h->owned, src.hdr, (size_t)src.n_chunks * 4, &dst->hdr))) return rc;
    if ((rc = upload<unsigned short>(h, h->owned, src.rows, (size_t)src.n_chunks * 64, &dst->rows))) return rc;
    if ((rc = upload<double>(h, h->owned, src.vals, (size_t)src.n_steps * 64, &dst->vals))) return rc;
    if ((rc = upload<unsigned short>(h, h->owned, src.cols, (size_t)src.n_steps * 64, &dst->cols))) return rc;
    return CPG_OK;
}
static int upload_csr(cpg_handle_t h, std::vector<void *> &own, const cpg_csr_t &src, cpg::DevCsr *dst) {
    int rc;
    dst->nnz = src.nnz;
    if (src.nnz == 0) { dst->ptr = nullptr; dst->idx = nullptr; dst->val = nullptr; return CPG_OK; }
    if ((rc = upload<int>(h, own, src.ptr, (size_t)src.rows + 1, &dst->ptr))) return rc;
    if ((rc = upload<int>(h, own, src.idx, (size_t)src.nnz, &dst->idx))) return rc;
    if ((rc = upload<double>(h, own, src.val, (size_t)src.nnz, &dst->val))) return rc;
    return CPG_OK;
}

// A ragged program (solve_program.RaggedProgram: ctab / desc / cols) in the layout of the streaming
// executor (run_program_stream in cpg_osqp_kernel.h documents the encoding): the (chunk, step) walk
// is flattened, consecutive steps are PAIRED -- a lane's two entries sit next to each other, so that
// one 16-byte load brings the coefficients and one 8-byte load the operand offsets of two steps.
// Lanes that are active in only one step of a pair get a zero entry.  src[e]: entry of the ragged
// program behind entry e of the new layout (-1: padding).
struct StreamTables {
    std::vector<unsigned> st, cr;
    std::vector<int> src;
    int n_pairs = 0;
};
static int build_stream_tables(const int *ctab, const unsigned *desc, const unsigned short *cols, int n_chunks,
                               int n_slots, StreamTables &T, const int DP = CPG_STREAM_DEPTH / 2) {
    if (n_slots >= 0x1FFF) { set_error("substitution program: work vector too large for the packed entry table"); return CPG_E_BADARG; }
    struct Step { unsigned base, cnt, flags; int chunk; };
    std::vector<Step> steps;
    for (int c = 0; c < n_chunks; c++) {
        const int L = ctab[4 * c], stages = ctab[4 * c + 1], kind = ctab[4 * c + 3];
        unsigned base = (unsigned)ctab[4 * c + 2];
        // kind: bit 0 segmented (balanced) chunk, bit 1 rows accumulate into their slot
        if (kind < 0 || kind > 3 || L < 1 || stages > 6) { set_error("substitution program: unsupported chunk kind"); return CPG_E_BADARG; }
        for (int s = 0; s < L; s++) {
            unsigned cnt = 0;
            for (int l = 0; l < 64; l++) {
                const unsigned d = desc[(size_t)c * 64 + l];
                const int len = (kind & 1) ? (int)((d >> 16) & 0xFFFu) : (int)(d >> 16);
                const unsigned row = d & 0xFFFFu, mask = (kind & 1) ? d >> 28 : 0u;
                const bool act = len > s;
                if (act && (unsigned)l != cnt) { set_error("substitution program: active lanes are not a prefix"); return CPG_E_BADARG; }
                if (!act && s == 0 && (row != 0xFFFFu || mask)) { set_error("substitution program: empty output row"); return CPG_E_BADARG; }
                if (act && s == 0 && (mask > 7u || (row != 0xFFFFu && row >= 0x1FFFu))) { set_error("substitution program: row / mask out of range"); return CPG_E_BADARG; }
                cnt += act;
            }
            steps.push_back({base, cnt, (unsigned)stages | ((unsigned)(kind & 1) << 3) | (s == 0 ? 16u : 0u) | (s == L - 1 ? 32u : 0u) |
                             ((kind & 2) ? 64u : 0u), c});
            base += cnt;
        }
    }
    if (steps.size() % 2) steps.push_back({0u, 0u, 0u, 0});
    unsigned pb = 0;                                          // pair base, in pairs of entries
    for (size_t p = 0; p < steps.size(); p += 2) {
        const Step &A = steps[p], &Bs = steps[p + 1];
        const unsigned cnt = A.cnt > Bs.cnt ? A.cnt : Bs.cnt;
        T.st.push_back(pb | (cnt << 18) | (A.flags << 25));
        T.st.push_back(Bs.flags);
        for (unsigned l = 0; l < cnt; l++)
            for (int t = 0; t < 2; t++) {
                const Step &S = t ? Bs : A;
                int from = -1;
                unsigned x = 0x1FFFu << 16;
                if (l < S.cnt) {
                    const unsigned eo = S.base + l;
                    from = (int)eo;
                    x = (unsigned)cols[eo] | (0x1FFFu << 16);
                    if (S.flags & 16u) {
                        const unsigned d = desc[(size_t)S.chunk * 64 + l];
                        const unsigned row = d & 0xFFFFu, mask = (S.flags & 8u) ? d >> 28 : 0u;
                        x = (unsigned)cols[eo] | ((row == 0xFFFFu ? 0x1FFFu : row) << 16) | (mask << 29);
                    }
                }
                T.src.push_back(from); T.cr.push_back(x);
            }
        pb += cnt;
    }
    if (pb >= 0x3FFFFu) { set_error("substitution program: too many entries for the packed step table"); return CPG_E_BADARG; }
    for (int t = 0; t < 2; t++) { T.src.push_back(-1); T.cr.push_back(0x1FFFu << 16); }   // the idle pair
    while ((T.st.size() / 2) % DP) { T.st.push_back(0u); T.st.push_back(0u); }
    T.n_pairs = (int)(T.st.size() / 2);
    T.st.resize(T.st.size() + 4 * DP, 0u);
    return CPG_OK;
}

// ------------------------------------------------------------------------------------ kernels
#define CPG_BLOCK_MAX 1024
// the refactorisation kernel streams its per-instance factor and waits on memory: a third wavefront
// per SIMD (168 VGPRs) pays on the portfolio family (10 + 13 slots); family libraries with few slots
// build it with four (128 VGPRs suffice: codegen.build_family_library)
#ifndef CPG_REFACTOR_WAVES_PER_SIMD
#define CPG_REFACTOR_WAVES_PER_SIMD 3
#endif
// (the adjoint kernel needs ~72 VGPRs; its residency is bound by the LDS vectors of a wavefront, the
// launch picks the workgroup shape that keeps the most wavefronts resident: cpg_hip_gradient_batch)
#ifndef CPG_GRADIENT_WAVES_PER_SIMD
#define CPG_GRADIENT_WAVES_PER_SIMD 3
#endif
#ifndef CPG_MIN_WAVES_PER_SIMD
#define CPG_MIN_WAVES_PER_SIMD 4   // 16 waves per CU: <= 128 VGPRs
#endif

// WMAX = waves per workgroup the kernel may be launched with; it fixes the register budget:
// streaming kernels run several 4-wave workgroups per CU (CPG_MIN_WAVES_PER_SIMD), LDS-resident
// kernels run ONE workgroup of up to WMAX waves per CU (WMAX / 4 waves per SIMD).
template <int NSX, int NSZ, int NV, int G, bool LDSPROG, int WMAX>
__global__ void __launch_bounds__(WMAX * 64, LDSPROG ? (WMAX + 3) / 4 : CPG_MIN_WAVES_PER_SIMD)
osqp_shared_kernel(cpg::DevFamily F, cpg::DevUpdate U, cpg::DevSettings S, cpg::DevBatch Bt) {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    const int wave_global = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    cpg::osqp_shared_body<NSX, NSZ, NV, G, LDSPROG>(F, U, S, Bt, cpg_lds, wave_global);
}
template <int NSX, int NSZ, int NV, int G, bool LDSPROG, int WMAX>
static int launch_t(cpg_handle_t h, const cpg::DevBatch &Bt, int blocks, int waves, size_t lds) {
    auto kern = osqp_shared_kernel<NSX, NSZ, NV, G, LDSPROG, WMAX>;
    if (lds > 48 * 1024)
        RT_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, h->stream, h->F, h->U, h->S, Bt);
    RT_CHECK(hipGetLastError());
    return CPG_OK;
}

template <int NSX, int NSZ, bool SHARED>
__global__ void __launch_bounds__(256, CPG_REFACTOR_WAVES_PER_SIMD)
osqp_refactor_kernel(cpg::DevFamily F, cpg::DevRefactor R, cpg::DevSettings S, cpg::DevBatch Bt) {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    const int wave_global = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    cpg::osqp_refactor_body<NSX, NSZ, false, SHARED>(F, R, S, Bt, cpg_lds, wave_global);
}
template <int NSX, int NSZ>
static int launch_refactor_t(cpg_handle_t h, rt_stream_t stream, const cpg::DevSettings &S, const cpg::DevBatch &Bt, int blocks, int waves, size_t lds) {
    auto kern = h->R.shared_mats ? osqp_refactor_kernel<NSX, NSZ, true> : osqp_refactor_kernel<NSX, NSZ, false>;
    if (lds > 48 * 1024)
        RT_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, stream, h->F, h->R, S, Bt);
    RT_CHECK(hipGetLastError());
    return CPG_OK;
}
#ifdef CPG_REFACTOR_CR_LDS
// per-instance-matrix kernel with the streaming executor's entry words in LDS (osqp_refactor_body<.., CRLDS>): one
// workgroup of eight wavefronts per CU shares the copy; compiled for families whose kernel runs two wavefronts per SIMD anyway
template <int NSX, int NSZ>
__global__ void __launch_bounds__(512, 2)
osqp_refactor_crlds_kernel(cpg::DevFamily F, cpg::DevRefactor R, cpg::DevSettings S, cpg::DevBatch Bt) {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    const int wave_global = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    cpg::osqp_refactor_body<NSX, NSZ, false, false, true>(F, R, S, Bt, cpg_lds, wave_global);
}
template <int NSX, int NSZ>
static int launch_refactor_crlds_t(cpg_handle_t h, rt_stream_t stream, const cpg::DevSettings &S, const cpg::DevBatch &Bt, int blocks, int waves, size_t lds) {
    auto kern = osqp_refactor_crlds_kernel<NSX, NSZ>;
    if (lds > 48 * 1024)
        RT_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, stream, h->F, h->R, S, Bt);
    RT_CHECK(hipGetLastError());
    return CPG_OK;
}
#endif
#ifdef CPG_GENI_HEADER
// the same body with the generated instance executor: a lane keeps its CPG_GENI_NSTEPS coefficients in registers
// for the whole ADMM loop, so the budget is the 256 VGPRs of two wavefronts per SIMD
template <int NSX, int NSZ>
__global__ void __launch_bounds__(512, 2)
osqp_instance_kernel(cpg::DevFamily F, cpg::DevRefactor R, cpg::DevSettings S, cpg::DevBatch Bt) {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    const int wave_global = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    cpg::osqp_refactor_body<NSX, NSZ, true>(F, R, S, Bt, cpg_lds, wave_global);
}
template <int NSX, int NSZ>
static int launch_instance_t(cpg_handle_t h, rt_stream_t stream, const cpg::DevSettings &S, const cpg::DevBatch &Bt, int blocks, int waves, size_t lds) {
    auto kern = osqp_instance_kernel<NSX, NSZ>;
    if (lds > 48 * 1024)
        RT_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, stream, h->F, h->R, S, Bt);
    RT_CHECK(hipGetLastError());
    return CPG_OK;
}
#endif
#ifdef CPG_GENR_HEADER
// resident per-instance factor kernel (cpg_osqp_resident.h): at most four wavefronts per workgroup and ONE workgroup per
// CU -- one wavefront per SIMD, which owns the SIMD's whole unified register file (256 VGPRs + 256 AGPRs)
template <int NSX, int NSZ>
__global__ void __launch_bounds__(256)
osqp_resident_kernel(cpg::DevFamily F, cpg::DevRefactor R, cpg::DevResident Rs, cpg::DevSettings S, cpg::DevBatch Bt) {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    const int wave_global = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    cpg::osqp_resident_body<NSX, NSZ>(F, R, Rs, S, Bt, cpg_lds, wave_global);
}
template <int NSX, int NSZ>
static int launch_resident_t(cpg_handle_t h, rt_stream_t stream, const cpg::DevSettings &S, const cpg::DevBatch &Bt, int blocks, int waves, size_t lds) {
    auto kern = osqp_resident_kernel<NSX, NSZ>;
    if (lds > 48 * 1024)
        RT_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, stream, h->F, h->R, h->Rs, S, Bt);
    RT_CHECK(hipGetLastError());
    return CPG_OK;
}
#endif
#ifdef CPG_GENT_HEADER
// team per-instance factor kernel (cpg_osqp_team.h): one workgroup of CPG_GENT_W wavefronts per instance.  Up to four
// wavefronts: one per SIMD, each with the SIMD's whole unified register file (512 registers); eight: two per SIMD (256).
__global__ void __launch_bounds__(CPG_GENT_W * 64)
osqp_team_kernel(cpg::DevFamily F, cpg::DevRefactor R, cpg::DevResident Rs, cpg::DevSettings S, cpg::DevBatch Bt) {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    constexpr int NX = (CPG_GENT_N + CPG_GENT_W * 64 - 1) / (CPG_GENT_W * 64), NZ = (CPG_GENT_M + CPG_GENT_W * 64 - 1) / (CPG_GENT_W * 64);
    cpg::osqp_team_body<(NX > 0 ? NX : 1), (NZ > 0 ? NZ : 1)>(F, R, Rs, S, Bt, cpg_lds, (int)blockIdx.x);
}
static int launch_team(cpg_handle_t h, rt_stream_t stream, const cpg::DevSettings &S, const cpg::DevBatch &Bt, int blocks, size_t lds) {
    auto kern = osqp_team_kernel;
    if (lds > 48 * 1024)
        RT_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(CPG_GENT_W * 64), lds, stream, h->F, h->R, h->Rs, S, Bt);
    RT_CHECK(hipGetLastError());
    return CPG_OK;
}
#endif
#ifdef CPG_GENQ_HEADER
// squad shared-factor kernel (cpg_osqp_squad.h): a workgroup of CPG_GENQ_W wavefronts solves CPG_GENQ_W instances at a time with the
// family's solve program in its registers; two wavefronts per SIMD (256 VGPRs)
template <int NSX, int NSZ, int NV>
__global__ void __launch_bounds__(CPG_GENQ_W * 64, 2)
osqp_squad_kernel(cpg::DevFamily F, cpg::DevUpdate U, cpg::DevSettings S, cpg::DevBatch Bt) {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    cpg::osqp_squad_body<NSX, NSZ, NV>(F, U, S, Bt, cpg_lds);
}
template <int NSX, int NSZ, int NV>
static int launch_squad_t(cpg_handle_t h, const cpg::DevBatch &Bt, int blocks, size_t lds) {
    auto kern = osqp_squad_kernel<NSX, NSZ, NV>;
    if (lds > 48 * 1024)
        RT_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(CPG_GENQ_W * 64), lds, h->stream, h->F, h->U, h->S, Bt);
    RT_CHECK(hipGetLastError());
    return CPG_OK;
}
#endif
#ifndef CPG_KERNELS_REFACTOR
#define CPG_KERNELS_REFACTOR(Z) Z(1, 1) Z(4, 4) Z(8, 8) Z(16, 16)
#endif
template <int NSX, int NSZ>
__global__ void __launch_bounds__(512, CPG_GRADIENT_WAVES_PER_SIMD)
osqp_gradient_kernel(cpg::DevFamily F, cpg::DevRefactor R, cpg::DevGradient Gd, cpg::DevGradBatch Bt) {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    const int wave_global = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    cpg::osqp_gradient_body<NSX, NSZ>(F, R, Gd, Bt, cpg_lds, wave_global);
}
template <int NSX, int NSZ>
static int launch_gradient_t(cpg_handle_t h, const cpg::DevGradBatch &Bt, int blocks, int waves, size_t lds) {
    auto kern = osqp_gradient_kernel<NSX, NSZ>;
    if (lds > 48 * 1024)
        RT_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, h->stream, h->F, h->R, h->Gd, Bt);
    RT_CHECK(hipGetLastError());
    return CPG_OK;
}
static int launch_gradient(cpg_handle_t h, const cpg::DevGradBatch &Bt, int blocks, int waves, size_t lds) {
    const int nsx = (h->F.n + 63) / 64, nsz = (h->F.m + 63) / 64;
#define Z(a, b) if (nsx <= a && nsz <= b) return launch_gradient_t<a, b>(h, Bt, blocks, waves, lds);
    CPG_KERNELS_REFACTOR(Z)
#undef Z
    set_error("problem family larger than the largest compiled slot class");
    return CPG_E_UNSUPPORTED;
}
// (stream and settings are the caller's: the hand-over launch of a linked handle runs on the shared-factor
// handle's stream with that handle's settings)
static int launch_refactor(cpg_handle_t h, rt_stream_t stream, const cpg::DevSettings &S, const cpg::DevBatch &Bt, int blocks, int waves, size_t lds) {
    const int nsx = (h->F.n + 63) / 64, nsz = (h->F.m + 63) / 64;
#define Z(a, b) if (nsx <= a && nsz <= b) return launch_refactor_t<a, b>(h, stream, S, Bt, blocks, waves, lds);
    CPG_KERNELS_REFACTOR(Z)
#undef Z
    set_error("problem family larger than the largest compiled slot class");
    return CPG_E_UNSUPPORTED;
}

// ---- conic interior-point kernel: no slot classes, every vector lives in LDS ---------------------
#ifndef CPG_CONIC_WAVES_PER_SIMD
#define CPG_CONIC_WAVES_PER_SIMD 4   // <= 128 VGPRs
#endif
template <bool TABLES_IN_LDS, bool SPECIALISED, bool NONSYM>
__global__ void __launch_bounds__(512, CPG_CONIC_WAVES_PER_SIMD)
clarabel_kernel(cpg::DevConic C, cpg::DevConicSettings S, cpg::DevBatch Bt) {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    const int wave_global = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    cpg::clarabel_body<TABLES_IN_LDS, SPECIALISED, NONSYM>(C, S, Bt, cpg_lds, wave_global);
}
// NONSYM: families with exponential / power cones run an instantiation of their own -- the symmetric kernel carries none of that code
template <bool TABLES_IN_LDS, bool SPECIALISED, bool NONSYM = false>
static int launch_conic_t(cpg_handle_t h, const cpg::DevBatch &Bt, int blocks, int waves, size_t lds) {
    auto kern = clarabel_kernel<TABLES_IN_LDS, SPECIALISED, NONSYM>;
    if (lds > 48 * 1024)
        RT_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, h->stream, h->C, h->CS, Bt);
    RT_CHECK(hipGetLastError());
    return CPG_OK;
}
static int launch_conic(cpg_handle_t h, const cpg::DevBatch &Bt, int blocks, int waves, size_t lds, bool tables_in_lds) {
    const bool extended = h->C.n_ns > 0 || h->C.n_psd > 0;       // exponential / power / PSD cones: the instantiation that carries their code
#ifdef CPG_GENC_HEADER
    // the library's own family: dimensions as compile-time constants (clarabel_body<., true>)
    if (h->conic_specialised && tables_in_lds)
        return extended ? launch_conic_t<true, true, true>(h, Bt, blocks, waves, lds) : launch_conic_t<true, true>(h, Bt, blocks, waves, lds);
#endif
    if (extended)
        return tables_in_lds ? launch_conic_t<true, false, true>(h, Bt, blocks, waves, lds) : launch_conic_t<false, false, true>(h, Bt, blocks, waves, lds);
    return tables_in_lds ? launch_conic_t<true, false>(h, Bt, blocks, waves, lds) : launch_conic_t<false, false>(h, Bt, blocks, waves, lds);
}


// Instantiated kernels.  (NSX, NSZ): slot class (ceil(n/64), ceil(m/64)); NV: leading slots with
// per-instance q / u (1: at most 64 parameter-dependent entries; NS: all); G instances per wave.
// The smallest class that covers the family is used.
//   streaming kernels  X(NSX, NSZ, NV, G)       4 waves per workgroup, several workgroups per CU
//   LDS-resident       Y(NSX, NSZ, NV, G, WMAX) one workgroup of <= WMAX waves per CU
#ifndef CPG_KERNELS
#define CPG_KERNELS(X)                                                                              \
    X(1, 1, 1, 1) X(1, 1, 1, 2) X(4, 4, 1, 1) X(4, 4, 1, 2) X(4, 4, 4, 1) X(8, 8, 1, 1) X(8, 8, 1, 2)   \
    X(8, 8, 8, 1) X(16, 16, 16, 1)
#endif
#ifndef CPG_KERNELS_LDS
#define CPG_KERNELS_LDS(Y)                                                                          \
    Y(1, 1, 1, 1, 8) Y(1, 1, 1, 1, 16) Y(1, 1, 1, 2, 8) Y(4, 4, 1, 1, 8) Y(4, 4, 1, 1, 16) Y(4, 4, 1, 2, 8)    \
    Y(4, 4, 4, 1, 8) Y(8, 8, 1, 1, 8) Y(8, 8, 1, 1, 12) Y(8, 8, 1, 2, 4) Y(8, 8, 1, 2, 8) Y(8, 8, 8, 1, 8) \
    Y(16, 16, 16, 1, 8)
#endif

static int launch(cpg_handle_t h, const cpg::DevBatch &Bt, int blocks, int waves, int G, size_t lds, bool in_lds) {
    const int nsx = (h->F.n + 63) / 64, nsz = (h->F.m + 63) / 64;
    const int nvx = (h->n_vary_x + 63) / 64, nvz = (h->n_vary_z + 63) / 64;
    const int nv = nvx > nvz ? nvx : nvz;
    if (!in_lds) {
#define X(a, b, v, g)                                                                             \
        if (nsx <= a && nsz <= b && (nv <= v || (v >= a && v >= b)) && G == g && waves <= 4)      \
            return launch_t<a, b, v, g, false, 4>(h, Bt, blocks, waves, lds);
        CPG_KERNELS(X)
#undef X
    } else {
#define Y(a, b, v, g, wm)                                                                         \
        if (nsx <= a && nsz <= b && (nv <= v || (v >= a && v >= b)) && G == g && waves <= wm)     \
            return launch_t<a, b, v, g, true, wm>(h, Bt, blocks, waves, lds);
        CPG_KERNELS_LDS(Y)
#undef Y
    }
    set_error("no compiled kernel for this family size / launch geometry");
    return CPG_E_UNSUPPORTED;
}

#ifdef CPG_GENQ_HEADER
static int launch_squad(cpg_handle_t h, const cpg::DevBatch &Bt, int blocks, size_t lds) {
    const int nsx = (h->F.n + 63) / 64, nsz = (h->F.m + 63) / 64;
    const int nvx = (h->n_vary_x + 63) / 64, nvz = (h->n_vary_z + 63) / 64;
    const int nv = nvx > nvz ? nvx : nvz;
#define Y(a, b, v, g, wm)                                                                         \
    if (nsx == a && nsz == b && (nv <= v || (v >= a && v >= b)) && g == 1)                        \
        return launch_squad_t<a, b, v>(h, Bt, blocks, lds);
    CPG_KERNELS_LDS(Y)
#undef Y
    set_error("no compiled squad kernel for this family size");
    return CPG_E_UNSUPPORTED;
}
#endif

// ------------------------------------------------------------------------------------ C-ABI
extern "C" {

const char *cpg_hip_last_error(void) { return g_err.c_str(); }

const char *cpg_hip_status_string(int32_t s) {
    switch (s) {   // OSQP's status strings (what CPG_Info.status holds in the reference)
        case 1: return "solved";
        case 2: return "solved inaccurate";
        case 3: return "primal infeasible";
        case 4: return "primal infeasible inaccurate";
        case 5: return "dual infeasible";
        case 6: return "dual infeasible inaccurate";
        case 7: return "maximum iterations reached";
        case 9: return "problem non convex";
        case 11: return "unsolved";
        case -2: return "needs refactorization";
        default: return "unknown";
    }
}

int cpg_hip_device_count(int *count) {
    if (!count) { set_error("count is NULL"); return CPG_E_BADARG; }
    RT_CHECK(hipGetDeviceCount(count));
    return CPG_OK;
}

// conic settings by name (cvxpygen/solvers/clarabel.py:63-119); returns the slot or nullptr
static double *conic_double_setting(cpg_handle_t h, const std::string &s) {
    cpg::DevConicSettings &c = h->CS;
    if (s == "max_step_fraction") return &c.max_step_fraction;
    if (s == "tol_gap_abs") return &c.tol_gap_abs;
    if (s == "tol_gap_rel") return &c.tol_gap_rel;
    if (s == "tol_feas") return &c.tol_feas;
    if (s == "tol_infeas_abs") return &c.tol_infeas_abs;
    if (s == "tol_infeas_rel") return &c.tol_infeas_rel;
    if (s == "tol_ktratio") return &c.tol_ktratio;
    if (s == "reduced_tol_gap_abs") return &c.red_gap_abs;
    if (s == "reduced_tol_gap_rel") return &c.red_gap_rel;
    if (s == "reduced_tol_feas") return &c.red_feas;
    if (s == "reduced_tol_infeas_abs") return &c.red_infeas_abs;
    if (s == "reduced_tol_infeas_rel") return &c.red_infeas_rel;
    if (s == "reduced_tol_ktratio") return &c.red_ktratio;
    if (s == "equilibrate_min_scaling") return &c.eq_min;
    if (s == "equilibrate_max_scaling") return &c.eq_max;
    if (s == "linesearch_backtrack_step") return &c.ls_backtrack;          // (exponential / power cones only)
    if (s == "min_switch_step_length") return &c.min_switch_step;
    if (s == "min_terminate_step_length") return &c.min_terminate_step;
    if (s == "static_regularization_constant") return &c.static_const;
    if (s == "static_regularization_proportional") return &c.static_prop;
    if (s == "dynamic_regularization_eps") return &c.dyn_eps;
    if (s == "dynamic_regularization_delta") return &c.dyn_delta;
    if (s == "iterative_refinement_reltol") return &c.ir_reltol;
    if (s == "iterative_refinement_abstol") return &c.ir_abstol;
    if (s == "iterative_refinement_stop_ratio") return &c.ir_stop_ratio;
    if (s == "time_limit") return &h->time_limit;
    return nullptr;
}
static int *conic_int_setting(cpg_handle_t h, const std::string &s) {
    cpg::DevConicSettings &c = h->CS;
    if (s == "max_iter") return &c.max_iter;
    if (s == "equilibrate_enable") return &c.equilibrate_enable;
    if (s == "equilibrate_max_iter") return &c.equilibrate_max_iter;
    if (s == "static_regularization_enable") return &c.static_reg_enable;
    if (s == "dynamic_regularization_enable") return &c.dynamic_reg_enable;
    if (s == "iterative_refinement_enable") return &c.ir_enable;
    if (s == "iterative_refinement_max_iter") return &c.ir_max_iter;
    if (s == "verbose") return &h->verbose;
    if (s == "direct_kkt_solver") return &h->direct_kkt_solver;
    if (s == "presolve_enable") return &h->presolve_enable;
    return nullptr;
}

int cpg_hip_set_default_settings(cpg_handle_t h) {
    if (!h) { set_error("null handle"); return CPG_E_BADARG; }
    if (h->conic) {   // cvxpygen/solvers/clarabel.py:63-119
        cpg::DevConicSettings &c = h->CS;
        c.max_iter = 200; c.max_step_fraction = 0.99;
        c.tol_gap_abs = 1e-8; c.tol_gap_rel = 1e-8; c.tol_feas = 1e-8; c.tol_infeas_abs = 1e-8; c.tol_infeas_rel = 1e-8;
        c.equilibrate_enable = 1; c.equilibrate_max_iter = 10; c.eq_min = 1e-4; c.eq_max = 1e4;
        c.min_terminate_step = 1e-4;
        c.static_reg_enable = 1; c.static_const = 1e-8; c.static_prop = 2.2e-16;
        c.dynamic_reg_enable = 1; c.dyn_eps = 1e-13; c.dyn_delta = 2e-7;
        c.ir_enable = 1; c.ir_reltol = 1e-13; c.ir_abstol = 1e-12; c.ir_max_iter = 10; c.ir_stop_ratio = 5.0;
        h->time_limit = 1e10; h->verbose = 1; h->direct_kkt_solver = 1; h->presolve_enable = 1;
        c.red_gap_abs = 5e-5; c.red_gap_rel = 5e-5; c.red_feas = 1e-4; c.red_infeas_abs = 5e-5; c.red_infeas_rel = 5e-5;
        c.red_ktratio = 1e-4; c.tol_ktratio = 1e-6; c.ls_backtrack = 0.8; c.min_switch_step = 0.1;
        return CPG_OK;
    }
    // defaults of the generated solver, cvxpygen/solvers/osqp.py:102-115
    h->S.max_iter = 4000; h->S.eps_abs = 1e-3; h->S.eps_rel = 1e-3; h->S.eps_prim_inf = 1e-4;
    h->S.eps_dual_inf = 1e-4; h->S.scaled_termination = 0; h->S.check_termination = 25;
    h->S.warm_starting = 1; h->S.debug_stage = 0;
    // ... and every other OSQP setting goes back to the linked library's default as well: the generated
    // cpg_set_solver_default_settings IS osqp_set_default_settings(solver.settings) (solvers/osqp.py:101,
    // utils.py:1071-1073).  For OSQP >= 1.0 -- the only API the reference's emitted calls compile against
    // (osqp_update_data_mat / _vec, OSQPSolver; pyproject.toml:26) -- that means rho adaptation every 50 iterations
    // with tolerance 5 and the duality-gap term in the termination test.  cpg_hip_set_build_option replaces these
    // four (e.g. adaptive_rho 0, check_dualgap 0 for a solver generated against an OSQP that never adapts).
    h->S.adaptive_rho = h->opt_adaptive_rho; h->S.adaptive_rho_interval = h->opt_adaptive_rho_interval;
    h->S.adaptive_rho_tolerance = h->opt_adaptive_rho_tolerance; h->S.check_dualgap = h->opt_check_dualgap;
    return CPG_OK;
}

int cpg_hip_set_setting(cpg_handle_t h, const char *name, double v) {
    if (!h || !name) { set_error("null argument"); return CPG_E_BADARG; }
    std::string s(name);
    if (h->conic) {
        // settings of the reference's Clarabel interface this backend has no counterpart for are accepted at their DEFAULTS only
        // (cvxpygen/solvers/clarabel.py:63-119): a batch kernel has no per-instance clock, solves every KKT system directly and runs no
        // presolve pass -- a value that would change what the reference's solver does is refused, not silently ignored
        if ((s == "time_limit" && v < 1e10) || (s == "direct_kkt_solver" && (int)v != 1) || (s == "presolve_enable" && (int)v != 1)) {
            set_error("Solver setting \"" + s + "\" is accepted at its default only (the batched interior-point kernel has no per-instance "
                      "time limit, no indirect KKT solver and no presolve pass)"); return CPG_E_UNSUPPORTED; }
        if (double *d = conic_double_setting(h, s)) { *d = v; return CPG_OK; }
        if (int *i = conic_int_setting(h, s)) { *i = (int)v; return CPG_OK; }
        set_error("Solver setting \"" + s + "\" not available."); return CPG_E_BADARG;
    }
    if (s == "max_iter") h->S.max_iter = (int)v;
    else if (s == "eps_abs") h->S.eps_abs = v;
    else if (s == "eps_rel") h->S.eps_rel = v;
    else if (s == "eps_prim_inf") h->S.eps_prim_inf = v;
    else if (s == "eps_dual_inf") h->S.eps_dual_inf = v;
    else if (s == "scaled_termination") h->S.scaled_termination = (int)v;
    else if (s == "check_termination") h->S.check_termination = (int)v;
    else if (s == "warm_starting") h->S.warm_starting = (int)v;
    else if (s == "debug_stage") h->S.debug_stage = (int)v;          // measurements only, see DevSettings
    else { set_error("Solver setting \"" + s + "\" not available."); return CPG_E_BADARG; }
    return CPG_OK;
}

int cpg_hip_set_build_option(cpg_handle_t h, const char *name, double v) {
    if (!h || !name) { set_error("null argument"); return CPG_E_BADARG; }
    if (h->conic) { set_error("not available for a conic (interior-point) handle"); return CPG_E_BADARG; }
    std::string s(name);
    if (s == "adaptive_rho") h->S.adaptive_rho = h->opt_adaptive_rho = (int)v;
    else if (s == "adaptive_rho_interval") h->S.adaptive_rho_interval = h->opt_adaptive_rho_interval = (int)v;
    else if (s == "adaptive_rho_tolerance") h->S.adaptive_rho_tolerance = h->opt_adaptive_rho_tolerance = v;
    else if (s == "check_dualgap") h->S.check_dualgap = h->opt_check_dualgap = (int)v;
    else if (s == "flag_rho_changes") h->flag_rho_changes = v != 0.0;
    else { set_error("Build option \"" + s + "\" not available."); return CPG_E_BADARG; }
    return CPG_OK;
}

int cpg_hip_get_setting(cpg_handle_t h, const char *name, double *v) {
    if (!h || !name || !v) { set_error("null argument"); return CPG_E_BADARG; }
    std::string s(name);
    if (h->conic) {
        // (read-only fact about the handle) 1: the substitution program runs on the library's generated executor
        if (s == "generated_executor") { *v = h->C.gc_ok ? 1.0 : 0.0; return CPG_OK; }
        // ... 1: its factorisations run the library's generated straight-line schedule (codegen.emit_conic_factor)
        if (s == "generated_factorisation") { *v = h->C.gf_ok ? 1.0 : 0.0; return CPG_OK; }
        // ... 1: the kernel instantiation with this family's dimensions compiled in is the one launched
        if (s == "specialised_kernel") { *v = h->conic_specialised ? 1.0 : 0.0; return CPG_OK; }
        if (double *d = conic_double_setting(h, s)) { *v = *d; return CPG_OK; }
        if (int *i = conic_int_setting(h, s)) { *v = *i; return CPG_OK; }
        set_error("Solver setting \"" + s + "\" not available."); return CPG_E_BADARG;
    }
    if (s == "max_iter") *v = h->S.max_iter;
    else if (s == "eps_abs") *v = h->S.eps_abs;
    else if (s == "eps_rel") *v = h->S.eps_rel;
    else if (s == "eps_prim_inf") *v = h->S.eps_prim_inf;
    else if (s == "eps_dual_inf") *v = h->S.eps_dual_inf;
    else if (s == "scaled_termination") *v = h->S.scaled_termination;
    else if (s == "check_termination") *v = h->S.check_termination;
    else if (s == "warm_starting") *v = h->S.warm_starting;
    else if (s == "adaptive_rho") *v = h->S.adaptive_rho;
    else if (s == "adaptive_rho_interval") *v = h->S.adaptive_rho_interval;
    else if (s == "adaptive_rho_tolerance") *v = h->S.adaptive_rho_tolerance;
    else if (s == "check_dualgap") *v = h->S.check_dualgap;
    // (read-only facts about the handle) 1: per-instance solves of this handle run the generated instance executor
    else if (s == "squad_executor") *v = (!h->refactor_mode && !h->conic && h->squad_ok && h->program_in_lds == 3 && h->inst_per_wave == 1) ? 1.0 : 0.0;
    else if (s == "team_executor") {
        // wavefronts per instance of the team kernel this handle's per-instance solves run on; 0: another kernel (no team plan, the
        // streaming placement was asked for, or the team's LDS need exceeds the device limit -- launch_per_instance's own test)
        *v = 0.0;
#ifdef CPG_GENT_HEADER
        if (h->refactor_mode && h->Rs.ok == 2 && !h->R.shared_mats && h->program_in_lds != 0 && h->program_in_lds != 2 &&
            ((size_t)CPG_TEAM_SLICE_OFF + (size_t)h->Rs.slice_doubles) * sizeof(double) <= h->lds_limit) *v = (double)CPG_GENT_W;
#endif
    }
    else if (s == "resident_executor") *v = (h->refactor_mode && h->Rs.ok == 1 && !h->R.shared_mats && h->program_in_lds != 0 && h->program_in_lds != 2) ? 1.0 : 0.0;
    else if (s == "generated_instance_executor") *v = (h->refactor_mode && h->R.gi_ok && h->program_in_lds != 0 && h->program_in_lds != 2) ? 1.0 : 0.0;
    else { set_error("Solver setting \"" + s + "\" not available."); return CPG_E_BADARG; }
    return CPG_OK;
}

int cpg_hip_create_osqp(const cpg_osqp_family_t *f, int device, cpg_handle_t *out) {
    if (!f || !out) { set_error("null argument"); return CPG_E_BADARG; }
    if (f->n <= 0 || f->m < 0 || f->n + f->m >= 0xFFFF) { set_error("bad family dimensions"); return CPG_E_BADARG; }
    int rc = rt_set_device(device);
    if (rc) return rc;
    cpg_handle_t h = new cpg_solver_s();
    h->device = device;
    {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess) { set_error(std::string("hipGetDeviceProperties: ") + hipGetErrorString(e)); delete h; return CPG_E_HIP; }
        h->num_cu = prop.multiProcessorCount;
        h->lds_limit = prop.sharedMemPerBlock;
        if (h->lds_limit < 160 * 1024 && strstr(prop.gcnArchName, "gfx950")) h->lds_limit = 160 * 1024;
        e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e)); delete h; return CPG_E_HIP; }
        hipEventCreate(&h->ev0); hipEventCreate(&h->ev1); hipEventCreate(&h->ev_mid); h->have_events = true;
    }
    cpg::DevFamily &F = h->F;
    F.n = f->n; F.m = f->m; F.n_eq = f->n_eq; F.is_max = f->is_maximization;
    F.sigma = f->sigma; F.alpha = f->alpha; F.rho = f->rho; F.c = f->c; F.cinv = 1.0 / f->c;
    std::vector<double> Dinv(f->n), Einv(f->m);
    for (int i = 0; i < f->n; i++) Dinv[i] = 1.0 / f->D[i];
    for (int i = 0; i < f->m; i++) Einv[i] = 1.0 / f->E[i];
#define TRY(x) do { rc = (x); if (rc) { cpg_hip_destroy(h); return rc; } } while (0)
    TRY(upload<double>(h, h->owned, f->D, f->n, &F.D));
    TRY(upload<double>(h, h->owned, Dinv.data(), f->n, &F.Dinv));
    TRY(upload<double>(h, h->owned, f->E, f->m, &F.E));
    TRY(upload<double>(h, h->owned, Einv.data(), f->m, &F.Einv));
    TRY(upload<signed char>(h, h->owned, (const signed char *)f->ctype, f->m, &F.ctype));
    TRY(upload_program(h, f->kkt, &F.kkt));
    TRY(upload_program(h, f->A_rows, &F.A_rows));
    TRY(upload_program(h, f->P_rows, &F.P_rows));
    TRY(upload_program(h, f->At_rows, &F.At_rows));
    { const cpg_program_t *pr[3] = {&f->A_rows, &f->P_rows, &f->At_rows};
      for (int k = 0; k < 3; k++) h->rows_hdr[k].assign(pr[k]->hdr, pr[k]->hdr + (size_t)pr[k]->n_chunks * 4); }
    F.kkt_ragged.n_chunks = f->kkt_ragged.n_chunks; F.kkt_ragged.nnz = f->kkt_ragged.nnz;
    F.kkt_stream = cpg::StreamProg{nullptr, nullptr, nullptr, 0, 0u};
    if (f->kkt_ragged.n_chunks > 0) {
        TRY(upload<int>(h, h->owned, f->kkt_ragged.ctab, (size_t)f->kkt_ragged.n_chunks * 4, &F.kkt_ragged.ctab));
        TRY(upload<unsigned>(h, h->owned, f->kkt_ragged.desc, (size_t)f->kkt_ragged.n_chunks * 64, &F.kkt_ragged.desc));
        TRY(upload<double>(h, h->owned, f->kkt_ragged.vals, (size_t)f->kkt_ragged.nnz, &F.kkt_ragged.vals));
        TRY(upload<unsigned short>(h, h->owned, f->kkt_ragged.cols, (size_t)f->kkt_ragged.nnz, &F.kkt_ragged.cols));
        // the same program for the streaming executor (used when the program is not LDS resident)
        if (f->n_slots < 0x1FFF) {
            StreamTables stt;
            std::vector<double> v2;
            int rcs = build_stream_tables(f->kkt_ragged.ctab, f->kkt_ragged.desc, f->kkt_ragged.cols, f->kkt_ragged.n_chunks, f->n_slots, stt);
            if (rcs == CPG_OK) {
                v2.resize(stt.src.size());
                for (size_t e = 0; e < stt.src.size(); e++) v2[e] = stt.src[e] >= 0 ? f->kkt_ragged.vals[stt.src[e]] : 0.0;
                TRY(upload<unsigned>(h, h->owned, stt.st.data(), stt.st.size(), &F.kkt_stream.stab));
                TRY(upload<unsigned>(h, h->owned, stt.cr.data(), stt.cr.size(), &F.kkt_stream.cr));
                TRY(upload<double>(h, h->owned, v2.data(), v2.size(), &F.kkt_stream.vals));
                TRY(rt_sync(h));
                F.kkt_stream.n_pairs = stt.n_pairs;
                F.kkt_stream.dummy = (unsigned)(stt.cr.size() / 2 - 1);
            }
        }
    }
    F.kkt_ragged.dict = nullptr; F.kkt_ragged.words = nullptr; F.kkt_ragged.n_dict = 0; F.kkt_ragged.cols_padded = nullptr; F.kkt_ragged.rows_gen = nullptr;
#ifdef CPG_GEN_COMPRESSED
    if (f->kkt_ragged.n_chunks > 0) {
        // dictionary of the distinct coefficients (bit patterns, sorted) and one word per entry
        const cpg_ragged_t &rg = f->kkt_ragged;
        std::vector<unsigned long long> bits((size_t)rg.nnz);
        memcpy(bits.data(), rg.vals, (size_t)rg.nnz * 8);
        std::vector<unsigned long long> uniq(bits);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        if (uniq.size() != (size_t)CPG_GEN_NDICT || uniq.size() > 8192) {
            set_error("this library was generated for a different problem family (coefficient dictionary)");
            cpg_hip_destroy(h); return CPG_E_BADARG; }
        std::vector<double> dict(uniq.size());
        memcpy(dict.data(), uniq.data(), uniq.size() * 8);
        std::vector<unsigned> words((size_t)rg.nnz);
        for (int e = 0; e < rg.nnz; e++) {
            const unsigned di = (unsigned)(std::lower_bound(uniq.begin(), uniq.end(), bits[e]) - uniq.begin());
            words[e] = (unsigned)rg.cols[e] | ((di * 8u) << 16);
        }
        TRY(upload<double>(h, h->owned, dict.data(), dict.size(), &F.kkt_ragged.dict));
        TRY(upload<unsigned>(h, h->owned, words.data(), words.size(), &F.kkt_ragged.words));
        TRY(rt_sync(h));
        F.kkt_ragged.n_dict = (int)dict.size();
    }
#endif
#ifdef CPG_GEN_HEADER
    if (f->kkt_ragged.n_chunks > 0) {   // (handles without a shared program only serve the refactorisation / adjoint kernels)
        // this build contains an executor generated for one specific family: refuse any other
        unsigned hsh = 0x811C9DC5u;
        auto mix = [&](const void *p, size_t nbytes) { const unsigned char *b = (const unsigned char *)p;
            for (size_t i = 0; i < nbytes; i++) hsh = (hsh ^ b[i]) * 0x01000193u; };
        const cpg_ragged_t &rg = f->kkt_ragged;
        if (rg.n_chunks > 0) { mix(rg.ctab, (size_t)rg.n_chunks * 16); mix(rg.desc, (size_t)rg.n_chunks * 256); mix(rg.cols, (size_t)rg.nnz * 2); }
        if (rg.n_chunks != CPG_GEN_NCHUNKS || rg.nnz != CPG_GEN_NNZ || hsh != CPG_GEN_FINGERPRINT) {
            set_error("this library was generated for a different problem family (solve-program fingerprint mismatch)");
            cpg_hip_destroy(h); return CPG_E_BADARG; }
#ifdef CPG_GEN_N
        {   // dimensions and uniform row classes are literals in the generated kernel
            bool ok = f->n == (int)cpg::GenFam::n && f->m == (int)cpg::GenFam::m && f->n_slots == cpg::GenFam::n_slots;
            for (int i = 0; ok && i < f->m; i++) { const int c = cpg::GenFam::ct(i / 64); if (c != 2 && c != (int)f->ctype[i]) ok = false; }
            {   // the termination test's row programs carry their chunk table as literals too
                const cpg_program_t *pr[3] = {&f->A_rows, &f->P_rows, &f->At_rows};
                const int nn[3] = {CPG_GEN_AROWS_N, CPG_GEN_PROWS_N, CPG_GEN_ATROWS_N};
                for (int k = 0; ok && k < 3; k++) {
                    if (pr[k]->n_chunks != nn[k]) ok = false;
                    for (int c2 = 0; ok && c2 < pr[k]->n_chunks; c2++)
                        if (pr[k]->hdr[4 * c2] != cpg::GenFam::rows_len(k, c2) || pr[k]->hdr[4 * c2 + 3] != cpg::GenFam::rows_off(k, c2)) ok = false;
                }
            }
            if (!ok) { set_error("this library was generated for a different problem family (dimensions / row classes)");
                       cpg_hip_destroy(h); return CPG_E_BADARG; }
        }
#endif
        {   // output-slot table of the generated executor: [chunk / 4][lane][chunk % 4], slot | segment mask << 13
            const int nch4 = (rg.n_chunks + 3) & ~3;
            std::vector<unsigned short> rt((size_t)nch4 * 64, (unsigned short)f->n_slots);
            for (int c = 0; c < rg.n_chunks; c++)
                for (int g0 = 0; g0 < 64; g0 += 16) {
                    // a ds_write_b64 is served in 16-lane groups whose 8-byte slots collide modulo 16: idle lanes
                    // take, in order, the dummy slots whose residue no row of the group uses
                    bool used[16] = {false};
                    for (int t = g0; t < g0 + 16; t++) { const unsigned d = rg.desc[(size_t)c * 64 + t]; if ((d & 0xFFFFu) != 0xFFFFu) used[(d & 0xFFFFu) % 16u] = true; }
                    int nxt = 0;
                    for (int t = g0; t < g0 + 16; t++) {
                        const unsigned d = rg.desc[(size_t)c * 64 + t];
                        unsigned slot = d & 0xFFFFu;
                        if (slot == 0xFFFFu) {
                            while (nxt < CPG_GEN_DUMMY_SLOTS && used[(unsigned)(f->n_slots + nxt) % 16u]) nxt++;
                            const int j = nxt < CPG_GEN_DUMMY_SLOTS ? nxt++ : (t & (CPG_GEN_DUMMY_SLOTS - 1));
                            slot = (unsigned)(f->n_slots + j);
                        }
                        rt[((size_t)(c >> 2) * 64 + t) * 4 + (c & 3)] = (unsigned short)(slot | ((d >> 28) << 13));
                    }
                }
            TRY(upload<unsigned short>(h, h->owned, rt.data(), rt.size(), &F.kkt_ragged.rows_gen));
            TRY(rt_sync(h));      // `rt` is a stack-lifetime buffer
        }
#ifdef CPG_GEN_PADDED_OFFSETS
        {   // operand offsets of all 64 lanes of every step, in the executor's order; idle lanes read the zero slot
            static const int steps[][2] = CPG_GEN_STEPS;        // {first entry, active lanes}
            const unsigned zero_off = (unsigned)(f->n_slots + CPG_GEN_DUMMY_SLOTS) * 8u;
            std::vector<unsigned short> cp((size_t)((CPG_GEN_PADDED_OFFSETS + 3) & ~3) * 64, (unsigned short)zero_off);
            bool ok = zero_off <= 0xFFFFu;
            for (int k = 0; ok && k < CPG_GEN_PADDED_OFFSETS; k++) {
                const int e = steps[k][0], cnt = steps[k][1];
                if (e < 0 || cnt < 0 || cnt > 64 || e + cnt > rg.nnz) { ok = false; break; }
                for (int t = 0; t < 64; t++) {
                    const unsigned short v = t < cnt ? rg.cols[e + t] : (unsigned short)zero_off;
#ifdef CPG_GEN_GROUPED_OFFSETS      // 2 or 4 consecutive steps of a lane side by side: one LDS read serves them all
                    cp[((size_t)(k / CPG_GEN_GROUPED_OFFSETS) * 64 + t) * CPG_GEN_GROUPED_OFFSETS + (k % CPG_GEN_GROUPED_OFFSETS)] = v;
#else
                    cp[(size_t)k * 64 + t] = v;
#endif
                }
            }
            if (!ok) { set_error("generated step table does not match the solve program"); cpg_hip_destroy(h); return CPG_E_BADARG; }
            TRY(upload<unsigned short>(h, h->owned, cp.data(), cp.size(), &F.kkt_ragged.cols_padded));
            TRY(rt_sync(h));      // `cp` is a stack-lifetime buffer
        }
#endif
        h->program_in_lds = 1;
#ifdef CPG_GENQ_HEADER
        h->squad_ok = true;          // (generated from the same plan as the LDS executor: CPG_GENQ_PARENT_FINGERPRINT, checked at compile time)
#endif
    }
#endif
    F.n_slots = f->n_slots;
    if (f->n_slots < f->n + f->m || f->n_slots >= 0xFFFF) { set_error("bad n_slots"); cpg_hip_destroy(h); return CPG_E_BADARG; }
    TRY(upload<unsigned short>(h, h->owned, f->fpos, (size_t)f->n + f->m, &F.fpos));
    h->n_vary_x = f->n_vary_x; h->n_vary_z = f->n_vary_z;
    F.n_prim = f->n_prim; F.n_dual = f->n_dual;
    TRY(upload<int>(h, h->owned, f->prim_idx, f->n_prim, &F.prim_idx));
    TRY(upload<int>(h, h->owned, f->dual_idx, f->n_dual, &F.dual_idx));
    F.ord = nullptr;
    if (f->ord) TRY(upload<int>(h, h->owned, f->ord, (size_t)f->n + f->m, &F.ord));
    { void *p = nullptr; TRY(rt_malloc(&p, 64)); h->d_counter = (unsigned *)p; }
    TRY(rt_sync(h));   // Dinv / Einv are stack-lifetime buffers
#undef TRY
    cpg_hip_set_default_settings(h);
    *out = h;
    return CPG_OK;
}

int cpg_hip_destroy(cpg_handle_t h);
static int open_device(cpg_handle_t h, int device) {
    h->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { set_error(std::string("hipGetDeviceProperties: ") + hipGetErrorString(e)); return CPG_E_HIP; }
    h->num_cu = prop.multiProcessorCount;
    h->lds_limit = prop.sharedMemPerBlock;
    if (h->lds_limit < 160 * 1024 && strstr(prop.gcnArchName, "gfx950")) h->lds_limit = 160 * 1024;
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e)); return CPG_E_HIP; }
    hipEventCreate(&h->ev0); hipEventCreate(&h->ev1); hipEventCreate(&h->ev_mid); h->have_events = true;
    return CPG_OK;
}

#ifdef CPG_GENC_FACTOR_HEADER
// The per-lane words of the library's generated factorisation (codegen.conic_factor_words) rebuilt from the schedule handed in and
// hashed (FNV-1a over the words as 32-bit little-endian): conic_factor_gen runs only the schedule it was emitted from.
static unsigned conic_factor_words_hash(const cpg_conic_family_t *f) {
    const int N = f->n + f->m;
    if (f->nnzL > 0xFF || N > 0xFF || f->fac_chunks != CPG_GENC_FAC_NCHUNKS) return 0;
    unsigned hsh = 2166136261u;
    auto put = [&](unsigned w) { for (int b = 0; b < 4; b++) hsh = (hsh ^ ((w >> (8 * b)) & 0xFFu)) * 16777619u; };
    std::vector<unsigned> tw;
    for (int c = 0; c < f->fac_chunks; c++) {
        const int L = f->fac_ctab[4 * c];
        unsigned base = (unsigned)f->fac_ctab[4 * c + 2];
        for (int s = 0; s < L; s++) {
            unsigned cnt = 0;
            for (int t = 0; t < 64; t++) {
                const unsigned lw = f->fac_len[(size_t)c * 64 + t];
                const int len = (int)(lw & 0xFFFFu), rlen = (int)(lw >> 16);
                unsigned w = 0;
                if (s < len) {
                    if (s < rlen) { const unsigned e = base + cnt; w = f->fac_a[e] | (f->fac_b[e] << 8) | (f->fac_k[e] << 16) | (1u << 24); }
                    cnt++;
                }
                put(w);
            }
            base += cnt;
        }
        for (int half = 0; half < 2; half++)
            for (int t = 0; t < 64; t++) {
                const unsigned tk = f->fac_task[(size_t)c * 64 + t];
                unsigned w = half == 0 ? 0xFFFFu : 0u;
                if (tk != 0xFFFFFFFFu) {
                    const bool piv = tk >= (unsigned)f->nnzL;
                    w = half == 0 ? (tk | ((unsigned)f->ksrc_kind[tk] << 16) | ((piv ? 1u : 0u) << 20) | ((piv ? 0u : (unsigned)f->Lcol[tk]) << 24))
                                  : (unsigned)f->ksrc_idx[tk];
                }
                tw.push_back(w);
            }
    }
    for (unsigned w : tw) put(w);
    return hsh;
}
#endif
#ifdef CPG_GENC_ROWS
// The generated row words of the library's family (codegen.conic_row_tables) rebuilt from the patterns of the family handed
// in -- rows of P (full symmetric view), columns of A, rows of A, [step][lane], same bit layout -- and hashed (FNV-1a over the
// words as 32-bit little-endian): the specialised kernel walks the two patterns through the compiled-in words only.
static unsigned conic_row_words_hash(const cpg_conic_family_t *f) {
    const int n = f->n, m = f->m;
    if (n > 64 || m > 64) return 0;
    const bool shortw = f->nnzP + f->nnzA <= 0xFF;
    const int ent_bits = shortw ? 8 : 16, top = shortw ? 15 : 31;
    unsigned hsh = 2166136261u;
    auto put = [&](unsigned w) { for (int b = 0; b < 4; b++) hsh = (hsh ^ ((w >> (8 * b)) & 0xFFu)) * 16777619u; };
    auto pass = [&](int rows, const int *ptr, auto entry, auto operand) {
        int S = 0;
        for (int r = 0; r < rows; r++) S = std::max(S, ptr[r + 1] - ptr[r]);
        for (int s = 0; s < S; s++)
            for (int lane = 0; lane < 64; lane++) {
                unsigned w = 0;
                if (lane < rows && s < ptr[lane + 1] - ptr[lane]) {
                    const int k = ptr[lane] + s;
                    w = (1u << top) | ((unsigned)operand(k) << ent_bits) | (unsigned)entry(k);
                }
                put(w);
            }
    };
    pass(n, f->Prp, [&](int k) { return f->Pent[k]; }, [&](int k) { return f->Pcol[k]; });
    pass(n, f->Ap, [&](int k) { return f->nnzP + k; }, [&](int k) { return f->Ai[k]; });
    pass(m, f->Arp, [&](int k) { return f->nnzP + f->Aent[k]; }, [&](int k) { return f->Acol[k]; });
    return hsh;
}
#endif

static bool conic_dims_equal(const cpg::DevConic &a, const cpg::DevConic &b) {
    return a.n == b.n && a.m == b.m && a.nnzP == b.nnzP && a.nnzA == b.nnzA && a.nnzL == b.nnzL && a.n_zero == b.n_zero &&
           a.n_nonneg == b.n_nonneg && a.n_soc == b.n_soc && a.n_ns == b.n_ns && a.n_psd == b.n_psd && a.psd_first == b.psd_first &&
           a.psd_doubles == b.psd_doubles && a.psd_degree == b.psd_degree && a.is_max == b.is_max && a.p_is_zero == b.p_is_zero &&
           a.fac_chunks == b.fac_chunks && a.sol_chunks == b.sol_chunks && a.sol_nnz == b.sol_nnz && a.sol_slots == b.sol_slots &&
           a.fac_triples == b.fac_triples && a.n_pfull == b.n_pfull && a.sv_pad == b.sv_pad && a.w_extra == b.w_extra &&
           a.gc_ncols == b.gc_ncols && a.gc_nrows == b.gc_nrows;
}
// LDS tables of a generated per-instance executor (codegen.emit_instance_program): operand byte offsets
// [step / 4][lane][4] (lanes without an entry read the zero slot) and output slots [chunk / 4][lane][4] (lanes
// without a row store to dummy slots chosen per 16-lane store group so that they add no bank conflict; segmented
// chunks carry their segment mask above bit 13).  `steps`: {first entry, active lanes} in execution order.
static bool generated_tables(const int *ctab, const unsigned *desc, const unsigned short *cols, int n_chunks, int nnz, int n_slots,
                             const int (*steps)[4], int T, std::vector<unsigned short> &gcols, std::vector<unsigned short> &grows,
                             const int *chunk_shift = nullptr) {
    // steps carry {first entry, active lanes, coefficient register, lane shift of the chunk}: narrow chunks sit at a
    // lane offset so that several steps share one coefficient register (codegen.pack_step_registers); `chunk_shift` is the
    // same shift per chunk, for the output-slot table
    const unsigned zero_off = (unsigned)(n_slots + CPG_GEN_DUMMY_SLOTS) * 8u;
    if (zero_off > 0xFFFFu || n_slots + CPG_GEN_EXTRA_SLOTS > 0x1FFF) return false;
    const int T4 = (T + 3) & ~3, C4 = (n_chunks + 3) & ~3;
    gcols.assign((size_t)T4 * 64, (unsigned short)zero_off);
    for (int t = 0; t < T; t++) {
        const int e = steps[t][0], cnt = steps[t][1], sh = steps[t][3];
        if (e < 0 || cnt < 0 || sh < 0 || sh + cnt > 64 || e + cnt > nnz) return false;
        for (int l = 0; l < cnt; l++) gcols[((size_t)(t / 4) * 64 + (l + sh)) * 4 + (t % 4)] = cols[e + l];
    }
    grows.assign((size_t)C4 * 64, (unsigned short)n_slots);
    for (int c = 0; c < n_chunks; c++) {
        const bool seg = ctab[4 * c + 3] & 1;
        const int sh = chunk_shift ? chunk_shift[c] : 0;
        if (sh < 0 || sh > 56 || (sh & 7)) return false;          // (multiples of 16; of 8 for chunks of at most 8 lanes)
        unsigned dsh[64];                                      // the chunk's lane descriptors at their shifted lanes
        for (int t = 0; t < 64; t++) dsh[t] = 0xFFFFu;
        for (int t = 0; t + sh < 64; t++) dsh[t + sh] = desc[(size_t)c * 64 + t];
        for (int t = 64 - sh; t < 64; t++) if ((desc[(size_t)c * 64 + t] & 0xFFFFu) != 0xFFFFu) return false;    // (a row would fall off)
        for (int g0 = 0; g0 < 64; g0 += 16) {
            bool used[16] = {false};
            for (int t = g0; t < g0 + 16; t++) { const unsigned d = dsh[t]; if ((d & 0xFFFFu) != 0xFFFFu) used[(d & 0xFFFFu) % 16u] = true; }
            int nxt = 0;
            for (int t = g0; t < g0 + 16; t++) {
                const unsigned d = dsh[t];
                unsigned slot = d & 0xFFFFu;
                if (slot == 0xFFFFu) {
                    while (nxt < CPG_GEN_DUMMY_SLOTS && used[(unsigned)(n_slots + nxt) % 16u]) nxt++;
                    const int j = nxt < CPG_GEN_DUMMY_SLOTS ? nxt++ : (t & (CPG_GEN_DUMMY_SLOTS - 1));
                    slot = (unsigned)(n_slots + j);
                }
                grows[((size_t)(c >> 2) * 64 + t) * 4 + (c & 3)] = (unsigned short)(slot | ((seg ? (d >> 28) : 0u) << 13));       // (every lane of a row carries its segment mask)
            }
        }
    }
    return true;
}
static unsigned program_fingerprint(const int *ctab, const unsigned *desc, const unsigned short *cols, int n_chunks, int nnz) {
    unsigned hsh = 0x811C9DC5u;
    auto mix = [&](const void *p, size_t nbytes) { const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < nbytes; i++) hsh = (hsh ^ b[i]) * 0x01000193u; };
    mix(ctab, (size_t)n_chunks * 16); mix(desc, (size_t)n_chunks * 256); mix(cols, (size_t)nnz * 2);
    return hsh;
}

int cpg_hip_create_clarabel(const cpg_conic_family_t *f, int device, cpg_handle_t *out) {
    if (!f || !out) { set_error("null argument"); return CPG_E_BADARG; }
    if (f->n <= 0 || f->m < 0 || f->n + f->m >= 0xFFFF) { set_error("bad family dimensions"); return CPG_E_BADARG; }
    long long rows = (long long)f->n_zero + f->n_nonneg;
    for (int k = 0; k < f->n_soc; k++) { if (f->soc_dims[k] < 1) { set_error("second-order cone of dimension < 1"); return CPG_E_BADARG; } rows += f->soc_dims[k]; }
    if (f->n_exp < 0 || f->n_pow < 0 || (f->n_pow > 0 && !f->pow_alpha)) { set_error("bad exponential / power cone counts"); return CPG_E_BADARG; }
    for (int k = 0; k < f->n_pow; k++)
        if (!(f->pow_alpha[k] > 0.0 && f->pow_alpha[k] < 1.0)) { set_error("power cone exponent outside (0, 1)"); return CPG_E_BADARG; }
    rows += 3LL * ((long long)f->n_exp + f->n_pow);
    if (f->n_psd < 0 || (f->n_psd > 0 && !f->psd_dims)) { set_error("bad PSD cone count"); return CPG_E_BADARG; }
    for (int k = 0; k < f->n_psd; k++) {
        if (f->psd_dims[k] < 1 || f->psd_dims[k] > CPG_PSD_MAX) { set_error("PSD cone order outside 1 .. 8"); return CPG_E_UNSUPPORTED; }
        rows += (long long)f->psd_dims[k] * (f->psd_dims[k] + 1) / 2;
    }
    if (rows != f->m) { set_error("cone dimensions do not add up to m"); return CPG_E_BADARG; }
    if (f->sol_slots < f->n + f->m || f->sol_slots > 8191) { set_error("bad sol_slots"); return CPG_E_BADARG; }
    int rc = rt_set_device(device);
    if (rc) return rc;
    cpg_handle_t h = new cpg_solver_s();
    h->conic = true;
    if ((rc = open_device(h, device))) { delete h; return rc; }
    cpg::DevConic &C = h->C;
    const int n = f->n, m = f->m, N = n + m;
    C.n = n; C.m = m; C.nnzP = f->nnzP; C.nnzA = f->nnzA; C.nnzL = f->nnzL; C.n_zero = f->n_zero;
    C.n_nonneg = f->n_nonneg; C.n_soc = f->n_soc; C.n_ns = f->n_exp + f->n_pow; C.ns_alpha = nullptr; C.is_max = f->is_maximization; C.p_is_zero = f->nnzP == 0;
    C.fac_chunks = f->fac_chunks; C.sol_chunks = f->sol_chunks; C.sol_nnz = f->sol_nnz; C.sol_slots = f->sol_slots;
    C.np_var = f->np_var; C.d_base = f->d_base; C.n_prim = f->n_prim; C.n_dual = f->n_dual;
    h->F.n = n; h->F.m = m; h->F.n_prim = f->n_prim; h->F.n_dual = f->n_dual;   // staging sizes of the host entry point
    std::vector<int> soc_start(f->n_soc > 0 ? f->n_soc : 1), row_cone(m > 0 ? m : 1, -1);
    std::vector<int> psd_start((size_t)(f->n_psd > 0 ? f->n_psd : 1)), psd_off((size_t)(f->n_psd > 0 ? f->n_psd : 1));
    C.n_psd = f->n_psd; C.psd_doubles = 0; C.psd_degree = 0;
    C.psd_start = C.psd_dim = C.psd_off = nullptr;
    {
        int o = f->n_zero + f->n_nonneg;
        for (int k = 0; k < f->n_soc; k++) { soc_start[k] = o; for (int r = 0; r < f->soc_dims[k]; r++) row_cone[o + r] = o; o += f->soc_dims[k]; }
        C.psd_first = o;
        for (int k = 0; k < f->n_psd; k++) {
            // row_cone of a PSD row: itself on the diagonal of the matrix (where a unit shift lands), the cone's first row elsewhere
            const int pp = f->psd_dims[k];
            psd_start[k] = o; psd_off[k] = C.psd_doubles;
            C.psd_doubles += 3 * pp * pp + pp + CPG_PSD_WORK(pp); C.psd_degree += pp;      // (NT point, factors, lambda | workspace)
            int a = 0;
            for (int j = 0; j < pp; j++) for (int i = 0; i <= j; i++, a++) row_cone[o + a] = i == j ? o + a : o;
            o += pp * (pp + 1) / 2;
        }
        if (C.psd_doubles > 0xFFF) { set_error("PSD cones too large for the 12-bit offsets of their KKT sources"); cpg_hip_destroy(h); return CPG_E_UNSUPPORTED; }
    }
    std::vector<void *> &own = h->owned;
#define TRY(x) do { rc = (x); if (rc) { cpg_hip_destroy(h); return rc; } } while (0)
#define UP(T, field, count) TRY(upload<T>(h, own, (const T *)f->field, (size_t)(count), (const T **)&C.field))
    TRY(upload<int>(h, own, soc_start.data(), (size_t)f->n_soc, &C.soc_start));
    TRY(upload<int>(h, own, f->soc_dims, (size_t)f->n_soc, &C.soc_dim));
    TRY(upload<int>(h, own, row_cone.data(), (size_t)m, &C.row_cone));
    if (f->n_psd > 0) {
        TRY(upload<int>(h, own, psd_start.data(), (size_t)f->n_psd, &C.psd_start));
        TRY(upload<int>(h, own, f->psd_dims, (size_t)f->n_psd, &C.psd_dim));
        TRY(upload<int>(h, own, psd_off.data(), (size_t)f->n_psd, &C.psd_off));
    }
    if (C.n_ns > 0) {      // exponent per nonsymmetric cone, 0 = exponential cone (rows: exponential cones first)
        std::vector<double> ns_alpha((size_t)C.n_ns, 0.0);
        for (int k = 0; k < f->n_pow; k++) ns_alpha[(size_t)f->n_exp + k] = f->pow_alpha[k];
        TRY(upload<double>(h, own, ns_alpha.data(), (size_t)C.n_ns, &C.ns_alpha));
    }
    UP(int, Ap, n + 1); UP(int, Ai, f->nnzA); UP(int, Arp, m + 1); UP(int, Aent, f->nnzA); UP(int, Acol, f->nnzA);
    UP(int, Pp, n + 1); UP(int, Pi, f->nnzP); UP(int, Prp, n + 1);
    { const int npf = f->Prp[n]; UP(int, Pent, npf); UP(int, Pcol, npf); }
    UP(int, Lcol, f->nnzL); UP(int, ksrc_kind, f->nnzL + N); UP(int, ksrc_idx, f->nnzL + N);
    UP(int, fac_ctab, (size_t)f->fac_chunks * 4); UP(unsigned, fac_task, (size_t)f->fac_chunks * 64);
    UP(unsigned, fac_len, (size_t)f->fac_chunks * 64);
    UP(unsigned, fac_a, f->fac_triples); UP(unsigned, fac_b, f->fac_triples); UP(unsigned, fac_k, f->fac_triples);
    UP(int, sol_ctab, (size_t)f->sol_chunks * 4); UP(unsigned, sol_desc, (size_t)f->sol_chunks * 64);
    UP(unsigned short, sol_cols, f->sol_nnz); UP(int, sol_kind, f->sol_nnz); UP(int, sol_idx, f->sol_nnz);
    UP(unsigned short, sol_fpos, N);
    UP(double, P_base, f->nnzP); UP(double, A_base, f->nnzA); UP(double, q_base, n); UP(double, b_base, m);
    UP(int, prim_idx, f->n_prim); UP(int, dual_idx, f->n_dual);
#undef UP
    TRY(upload_csr(h, own, f->map_P, &C.map_P)); TRY(upload_csr(h, own, f->map_A, &C.map_A));
    TRY(upload_csr(h, own, f->map_q, &C.map_q)); TRY(upload_csr(h, own, f->map_b, &C.map_b));
    TRY(upload_csr(h, own, f->map_d, &C.map_d));
    C.gc_ok = 0; C.sv_pad = 0; C.w_extra = 0; C.gc_ncols = 0; C.gc_nrows = 0; C.gc_cols = nullptr; C.gc_rows = nullptr;
    C.gf_ok = 0;
#ifdef CPG_GENC_FACTOR_HEADER
    C.gf_ok = (conic_factor_words_hash(f) == CPG_GENC_FAC_HASH && !(getenv("CPG_CONIC_FACTOR") && atoi(getenv("CPG_CONIC_FACTOR")) == 0)) ? 1 : 0;
#endif
#ifdef CPG_GENC_HEADER
    // a family library: its generated executor replaces the table-driven one when the substitution program handed in is
    // the one the executor was generated from
    if (f->sol_chunks == CPG_GENC_NCHUNKS && f->sol_nnz == CPG_GENC_NNZ && f->sol_slots == CPG_GENC_NSLOTS &&
        program_fingerprint(f->sol_ctab, f->sol_desc, f->sol_cols, f->sol_chunks, f->sol_nnz) == CPG_GENC_FINGERPRINT &&
        !(getenv("CPG_CONIC_GENERATED") && atoi(getenv("CPG_CONIC_GENERATED")) == 0)) {
        static const int steps[][4] = CPG_GENC_STEPS;               // {first entry, active lanes, step, 0}
        std::vector<unsigned short> gcols, grows;
        if (generated_tables(f->sol_ctab, f->sol_desc, f->sol_cols, f->sol_chunks, f->sol_nnz, f->sol_slots, steps, CPG_GENC_NSTEPS, gcols, grows)) {
            TRY(upload<unsigned short>(h, own, gcols.data(), gcols.size(), &C.gc_cols));
            TRY(upload<unsigned short>(h, own, grows.data(), grows.size(), &C.gc_rows));
            C.gc_ncols = (int)gcols.size(); C.gc_nrows = (int)grows.size();
            C.gc_ok = 1; C.sv_pad = 64; C.w_extra = CPG_GEN_EXTRA_SLOTS;
            // every dimension the generated header compiled in must be this family's (CPG_GENC_SPECIALISE assigns the
            // same fields of the kernel's copy of C)
            cpg::DevConic Cs = C;
            Cs.n_pfull = f->Prp[n]; Cs.fac_triples = f->fac_triples;
            cpg::DevConic Ck = Cs;
            { cpg::DevConic &Cref = Ck; CPG_GENC_SPECIALISE(Cref) }
            h->conic_specialised = conic_dims_equal(Cs, Ck) && !(getenv("CPG_CONIC_SPECIALISED") && atoi(getenv("CPG_CONIC_SPECIALISED")) == 0);
#ifdef CPG_GENC_ROWS
            if (h->conic_specialised && conic_row_words_hash(f) != CPG_GENC_ROWS_HASH) h->conic_specialised = false;   // same dimensions, other patterns
#endif
        }
    }
#endif
    {   // LDS doubles of the block-shared copy of the index tables, same order and rounding as
        // clarabel_body<true> (conic_stage)
        C.fac_triples = f->fac_triples; C.n_pfull = f->Prp[n];
        auto dbl = [](size_t count, size_t elem) { return (count * elem + 7) / 8; };
        size_t t = 0;
        t += 2 * dbl((size_t)f->n_soc, 4) + dbl((size_t)m, 4);
        bool row_words = false;
#ifdef CPG_GENC_ROWS
        // the library's own family: the generated row words instead of the CSR / CSC arrays of the two patterns
        row_words = h->conic_specialised;
        if (row_words) t += dbl((size_t)CPG_GENC_ROWS_WORDS, sizeof(cpg::genc_row_word));
#endif
        if (!row_words) {
            t += dbl((size_t)n + 1, 4) + dbl((size_t)f->nnzA, 4) + dbl((size_t)m + 1, 4) + 2 * dbl((size_t)f->nnzA, 4);
            t += dbl((size_t)n + 1, 4) + dbl((size_t)f->nnzP, 4) + dbl((size_t)n + 1, 4) + 2 * dbl((size_t)C.n_pfull, 4);
        }
        t += dbl((size_t)f->nnzL, 4) + 2 * dbl((size_t)f->nnzL + N, 4);
        t += dbl((size_t)f->fac_chunks * 4, 4) + 2 * dbl((size_t)f->fac_chunks * 64, 4) + 3 * dbl((size_t)f->fac_triples, 4);
#ifdef CPG_GENC_HEADER
        t += dbl((size_t)C.gc_ncols, 2) + dbl((size_t)C.gc_nrows, 2);
#else
        t += dbl((size_t)f->sol_chunks * 4, 4) + dbl((size_t)f->sol_chunks * 64, 4) + dbl((size_t)f->sol_nnz, 2);
#endif
        t += 2 * dbl((size_t)f->sol_nnz, 4) + dbl((size_t)N, 2);
        C.tab_doubles = (int)t;
    }
    {   // per-wavefront LDS slice, see conic_carve()
        const long long d = (long long)f->nnzP + f->nnzA + 7LL * n + 14LL * m + 6LL * N + f->nnzL + f->sol_nnz + C.sv_pad + f->sol_slots + C.w_extra + C.psd_doubles;
        C.lds_doubles = (int)((d + 1) & ~1LL);
        if ((size_t)C.lds_doubles * 8 > h->lds_limit) {
            set_error("conic family too large: the interior-point state of one instance does not fit the LDS");
            cpg_hip_destroy(h); return CPG_E_UNSUPPORTED; }
    }
    { void *p = nullptr; TRY(rt_malloc(&p, 64)); h->d_counter = (unsigned *)p; }
    TRY(rt_sync(h));
#undef TRY
    h->have_update = true;
    cpg_hip_set_default_settings(h);
    *out = h;
    return CPG_OK;
}

static int ensure(DevBuf &b, size_t bytes);
struct cpg_pipe_s;
static void free_pipe(cpg_pipe_s *p);
static void free_list(std::vector<void *> &v) { for (void *p : v) rt_free(p); v.clear(); }
static void free_buf(DevBuf &b) { if (b.p) rt_free(b.p); b.p = nullptr; b.bytes = 0; }

int cpg_hip_destroy(cpg_handle_t h) {
    if (!h) return CPG_OK;
    rt_set_device(h->device);
    rt_sync(h);
    free_list(h->owned); free_list(h->update_owned); free_list(h->refactor_owned); free_list(h->resident_owned); free_list(h->gradient_owned);
    free_buf(h->g_theta); free_buf(h->g_x); free_buf(h->g_y); free_buf(h->g_dprim); free_buf(h->g_dtheta);
    if (h->d_counter) rt_free(h->d_counter);
    free_buf(h->scratch);
    free_buf(h->s_theta); free_buf(h->s_prim); free_buf(h->s_dual); free_buf(h->s_obj);
    free_buf(h->s_pri); free_buf(h->s_dua); free_buf(h->s_iter); free_buf(h->s_status);
    free_buf(h->s_state_in); free_buf(h->s_state_out);
    free_buf(h->ho_state); free_buf(h->ho_list);
    free_pipe(h->pipe); h->pipe = nullptr;
    if (h->have_events) { hipEventDestroy(h->ev0); hipEventDestroy(h->ev1); hipEventDestroy(h->ev_mid); }
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return CPG_OK;
}

int cpg_hip_set_update(cpg_handle_t h, const cpg_osqp_update_t *u) {
    if (!h || !u) { set_error("null argument"); return CPG_E_BADARG; }
    if (h && h->conic) { set_error("not available for a conic (interior-point) handle"); return CPG_E_BADARG; }
    if (u->map_q.nnz && u->map_q.rows != h->F.n) { set_error("map_q rows != n"); return CPG_E_BADARG; }
    if (u->map_u.nnz && u->map_u.rows != h->F.m) { set_error("map_u rows != m"); return CPG_E_BADARG; }
    {   // parameter-dependent entries must sit inside the leading "varying" prefix
        const cpg_csr_t *mp[2] = {&u->map_q, &u->map_u};
        const int lim[2] = {h->n_vary_x, h->n_vary_z};
        for (int k = 0; k < 2; k++)
            if (mp[k]->nnz > 0 && mp[k]->ptr[mp[k]->rows] != mp[k]->ptr[lim[k] < mp[k]->rows ? lim[k] : mp[k]->rows]) {
                set_error("update map touches entries outside the parameter-dependent prefix"); return CPG_E_BADARG; }
    }
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    if ((rc = rt_sync(h))) return rc;
    free_list(h->update_owned);
    cpg::DevUpdate &U = h->U;
    U.np_var = u->np_var; U.d_base = u->d_base;
    if ((rc = upload<double>(h, h->update_owned, u->q_base, h->F.n, &U.q_base))) return rc;
    if ((rc = upload<double>(h, h->update_owned, u->u_base, h->F.m, &U.u_base))) return rc;
    if ((rc = upload_csr(h, h->update_owned, u->map_q, &U.map_q))) return rc;
    if ((rc = upload_csr(h, h->update_owned, u->map_u, &U.map_u))) return rc;
    if ((rc = upload_csr(h, h->update_owned, u->map_d, &U.map_d))) return rc;
    if ((rc = rt_sync(h))) return rc;
    h->have_update = true;
    h->refactor_mode = false;
    return CPG_OK;
}

int cpg_hip_set_refactor(cpg_handle_t h, const cpg_osqp_refactor_t *r) {
    if (h && h->conic) { set_error("not available for a conic (interior-point) handle"); return CPG_E_BADARG; }
    if (!h || !r) { set_error("null argument"); return CPG_E_BADARG; }
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    if ((rc = rt_sync(h))) return rc;
    free_list(h->refactor_owned);
    free_list(h->resident_owned); h->Rs = cpg::DevResident{};      // (cpg_hip_set_resident installs its tables behind this call)
    std::vector<void *> &own = h->refactor_owned;
    cpg::DevRefactor &R = h->R;
    const size_t n = h->F.n, m = h->F.m, N = n + m;
    R.nnzP = r->nnzP; R.nnzA = r->nnzA; R.nnzL = r->nnzL; R.n_eq = h->F.n_eq; R.np_var = r->np_var;
    R.scaling_iters = r->scaling_iters; R.d_base = r->d_base;
    R.fac_chunks = r->fac_chunks; R.sol_nnz = r->sol_nnz; R.sol_slots = r->sol_slots;
#define UP(T, field, count) if ((rc = upload<T>(h, own, (const T *)r->field, (size_t)(count), (const T **)&R.field))) return rc
    UP(int, Ap, n + 1); UP(int, Ai, r->nnzA); UP(int, Arp, m + 1); UP(int, Aent, r->nnzA); UP(int, Acol, r->nnzA);
    UP(int, Pp, n + 1); UP(int, Pi, r->nnzP); UP(int, Prp, n + 1);
    { const int npf = r->Prp[n]; UP(int, Pent, npf); UP(int, Pcol, npf); }
    UP(int, Lcol, r->nnzL); UP(int, ksrc_kind, r->nnzL + N); UP(int, ksrc_idx, r->nnzL + N);
    UP(int, fac_ctab, (size_t)r->fac_chunks * 4); UP(unsigned, fac_task, (size_t)r->fac_chunks * 64);
    UP(unsigned, fac_len, (size_t)r->fac_chunks * 64);
    UP(unsigned, fac_a, r->fac_triples); UP(unsigned, fac_b, r->fac_triples); UP(unsigned, fac_k, r->fac_triples);
    // Substitution program in the layout of the streaming executor; the value sources are permuted
    // accordingly (build_stream_tables above).
    StreamTables stt;                                             // alive until the sync below
    std::vector<int> kind2, idx2;
    R.gs_cols = nullptr; R.gs_rows = nullptr; R.gs_kind = nullptr; R.gs_idx = nullptr; R.gs_ok = 0; R.gs_nnz = 0;
#ifdef CPG_GENS_HEADER
    std::vector<unsigned short> gscols, gsrows;
    std::vector<int> gskind, gsidx;
    if (!r->shared_mats && r->sol_chunks == CPG_GENS_NCHUNKS && r->sol_nnz == CPG_GENS_NNZ && r->sol_slots == CPG_GENS_NSLOTS &&
        program_fingerprint(r->sol_ctab, r->sol_desc, r->sol_cols, r->sol_chunks, r->sol_nnz) == CPG_GENS_FINGERPRINT) {
        // this library runs THIS per-instance-matrix program as generated straight-line code: next to the streaming layout
        // (the adjoint kernel reads that one) the value sources in program-entry order, 64 zeros behind them (a step's idle
        // lanes read the entries that follow), and the executor's offset / slot tables
        static const int gsteps[][4] = CPG_GENS_STEPS;
        if (generated_tables(r->sol_ctab, r->sol_desc, r->sol_cols, r->sol_chunks, r->sol_nnz, r->sol_slots, gsteps, CPG_GENS_NSTEPS, gscols, gsrows)) {
            gskind.assign((size_t)r->sol_nnz + 64, 0); gsidx.assign((size_t)r->sol_nnz + 64, 0);
            for (int e = 0; e < r->sol_nnz; e++) { gskind[e] = r->sol_kind[e]; gsidx[e] = r->sol_idx[e]; }
            R.gs_nnz = r->sol_nnz + 64;
            if ((rc = upload<unsigned short>(h, own, gscols.data(), gscols.size(), &R.gs_cols))) return rc;
            if ((rc = upload<unsigned short>(h, own, gsrows.data(), gsrows.size(), &R.gs_rows))) return rc;
            if ((rc = upload<int>(h, own, gskind.data(), gskind.size(), &R.gs_kind))) return rc;
            if ((rc = upload<int>(h, own, gsidx.data(), gsidx.size(), &R.gs_idx))) return rc;
            R.gs_ok = 1;
        }
    }
#endif
    {
        if ((rc = build_stream_tables(r->sol_ctab, r->sol_desc, r->sol_cols, r->sol_chunks, r->sol_slots, stt))) return rc;
        kind2.resize(stt.src.size()); idx2.resize(stt.src.size());
        for (size_t e = 0; e < stt.src.size(); e++) {
            kind2[e] = stt.src[e] >= 0 ? r->sol_kind[stt.src[e]] : 0;
            idx2[e] = stt.src[e] >= 0 ? r->sol_idx[stt.src[e]] : 0;
        }
        R.sol_pairs = stt.n_pairs;
        R.sol_nnz = (int)stt.cr.size();
        if ((rc = upload<unsigned>(h, own, stt.st.data(), stt.st.size(), &R.sol_stab))) return rc;
        if ((rc = upload<unsigned>(h, own, stt.cr.data(), stt.cr.size(), &R.sol_cr))) return rc;
        if ((rc = upload<int>(h, own, kind2.data(), kind2.size(), &R.sol_kind))) return rc;
        if ((rc = upload<int>(h, own, idx2.data(), idx2.size(), &R.sol_idx))) return rc;
    }
    UP(unsigned short, sol_fpos, N);
    UP(double, P_base, r->nnzP); UP(double, A_base, r->nnzA); UP(double, q_base, n); UP(double, u_base, m);
    UP(double, q_setup, n);
#undef UP
    if ((rc = upload_csr(h, own, r->map_P, &R.map_P))) return rc;
    if ((rc = upload_csr(h, own, r->map_A, &R.map_A))) return rc;
    if ((rc = upload_csr(h, own, r->map_q, &R.map_q))) return rc;
    if ((rc = upload_csr(h, own, r->map_u, &R.map_u))) return rc;
    if ((rc = upload_csr(h, own, r->map_d, &R.map_d))) return rc;
    R.buf_doubles = (long long)(r->nnzP + 2 * r->nnzA + 3 * n + 4 * m + r->nnzL + 2 * N + std::max(R.sol_nnz, R.gs_nnz) + 64);   // see carve()
    R.shared_mats = r->shared_mats ? 1 : 0; R.cs = 1.0;
    R.Ps = R.As = R.Ars = R.Ds = R.Dinvs = R.Es = R.Einvs = nullptr;
    std::vector<double> ars, dinv, einv;                          // alive until the sync below
    if (r->shared_mats) {
        if (!r->Ps || !r->As || !r->D || !r->E || !(r->c > 0.0)) { set_error("shared-matrix mode needs Ps, As, D, E, c"); return CPG_E_BADARG; }
        ars.resize((size_t)r->nnzA); dinv.resize(n); einv.resize(m);
        for (int k = 0; k < r->nnzA; k++) ars[k] = r->As[r->Aent[k]];               // row-ordered copy (CPG_REFACTOR_ROW_COPY builds)
        for (size_t i = 0; i < n; i++) dinv[i] = 1.0 / r->D[i];
        for (size_t i = 0; i < m; i++) einv[i] = 1.0 / r->E[i];
        R.cs = r->c;
        if ((rc = upload<double>(h, own, r->Ps, (size_t)r->nnzP, &R.Ps))) return rc;
        if ((rc = upload<double>(h, own, r->As, (size_t)r->nnzA, &R.As))) return rc;
        if ((rc = upload<double>(h, own, ars.data(), ars.size(), &R.Ars))) return rc;
        if ((rc = upload<double>(h, own, r->D, n, &R.Ds))) return rc;
        if ((rc = upload<double>(h, own, dinv.data(), n, &R.Dinvs))) return rc;
        if ((rc = upload<double>(h, own, r->E, m, &R.Es))) return rc;
        if ((rc = upload<double>(h, own, einv.data(), m, &R.Einvs))) return rc;
    }
    R.gf_tri = nullptr; R.gf_dk = nullptr;
    R.gi_ok = 0; R.gi_cols = R.gi_rows = nullptr; R.gi_src = nullptr; R.gi_lcol = nullptr; R.fac_kc = nullptr; R.fac_krow = nullptr; R.fac_kc_cl = nullptr; R.fac_krow_cl = nullptr; R.fac_kind_cl = nullptr; R.fac_idx_cl = nullptr;
    std::vector<double> fkc, fkc_cl;                              // alive until the sync below
    std::vector<int> fkrow, fkrow_cl;

    if (r->shared_mats) {   // KKT values of the factorisation's destinations: constants of the family, except -1 / rho_vec
        bool ok = true;
        const size_t nd = (size_t)r->nnzL + N;
        fkc.assign(nd, 0.0); fkrow.assign(nd, -1);
        for (size_t d = 0; ok && d < nd; d++) {
            const int kind = r->ksrc_kind[d], idx = r->ksrc_idx[d];
            const bool piv = d >= (size_t)r->nnzL;
            if (kind == CPG_K_P) { if (idx < 0 || idx >= r->nnzP) ok = false; else fkc[d] = r->Ps[idx] + (piv ? h->F.sigma : 0.0); }
            else if (kind == CPG_K_A) { if (idx < 0 || idx >= r->nnzA) ok = false; else fkc[d] = r->As[idx]; }
            else if (kind == CPG_K_SIGMA) fkc[d] = h->F.sigma;
            else if (kind == CPG_K_RHO) { if (idx < 0 || (size_t)idx >= m) ok = false; else fkrow[d] = idx; }
            else if (kind != CPG_K_NONE) ok = false;
        }
        if (!ok) { set_error("cpg_hip_set_refactor: KKT source table out of range"); return CPG_E_BADARG; }
        fkc_cl.assign((size_t)r->fac_chunks * 64, 0.0); fkrow_cl.assign((size_t)r->fac_chunks * 64, -1);
        for (size_t e = 0; e < fkc_cl.size(); e++) {
            const unsigned t = r->fac_task[e];
            if (t == 0xFFFFFFFFu) continue;
            if (t >= nd) { set_error("cpg_hip_set_refactor: factorisation task out of range"); return CPG_E_BADARG; }
            fkc_cl[e] = fkc[t]; fkrow_cl[e] = fkrow[t];
        }
        if ((rc = upload<double>(h, own, fkc.data(), fkc.size(), &R.fac_kc))) return rc;
        if ((rc = upload<int>(h, own, fkrow.data(), fkrow.size(), &R.fac_krow))) return rc;
        if ((rc = upload<double>(h, own, fkc_cl.data(), fkc_cl.size(), &R.fac_kc_cl))) return rc;
        if ((rc = upload<int>(h, own, fkrow_cl.data(), fkrow_cl.size(), &R.fac_krow_cl))) return rc;
    }
#ifdef CPG_GENI_HEADER
    std::vector<unsigned short> gcols, grows, glcol;
    std::vector<unsigned> gsrc, gdk;
    std::vector<unsigned long long> gtri;
#ifdef CPG_GENI_N
    const bool geni_dims = h->F.n == CPG_GENI_N && h->F.m == CPG_GENI_M && h->F.n_eq == CPG_GENI_NEQ;
#else
    const bool geni_dims = true;
#endif
    if (geni_dims && r->shared_mats && r->sol_chunks == CPG_GENI_NCHUNKS && r->sol_nnz == CPG_GENI_NNZ && r->sol_slots == CPG_GENI_NSLOTS) {
        const unsigned hsh = program_fingerprint(r->sol_ctab, r->sol_desc, r->sol_cols, r->sol_chunks, r->sol_nnz);
        static const int steps[][4] = CPG_GENI_STEPS;             // {first entry, active lanes, coefficient register, lane shift} in execution order
        static const int chunk_shift[] = CPG_GENI_CHUNK_SHIFT;
        bool ok = hsh == CPG_GENI_FINGERPRINT;
        {   // the row programs of the termination test carry their chunk tables as literals
            const int nn[3] = {CPG_GENI_AROWS_N, CPG_GENI_PROWS_N, CPG_GENI_ATROWS_N};
            for (int k = 0; ok && k < 3; k++) {
                if ((int)h->rows_hdr[k].size() != 4 * nn[k]) ok = false;
                for (int c2 = 0; ok && c2 < nn[k]; c2++)
                    if (h->rows_hdr[k][4 * c2] != cpg::GeniRows::len(k, c2) || h->rows_hdr[k][4 * c2 + 3] != cpg::GeniRows::off(k, c2)) ok = false;
            }
        }
        if (ok) ok = generated_tables(r->sol_ctab, r->sol_desc, r->sol_cols, r->sol_chunks, r->sol_nnz, r->sol_slots, steps, CPG_GENI_NSTEPS, gcols, grows, chunk_shift);
        if (ok) {
            // coefficient sources per (register, lane): the steps that share a register occupy disjoint lane ranges
            gsrc.assign((size_t)CPG_GENI_NREGS * 64, 0u);
            glcol.assign((size_t)CPG_GENI_NREGS * 64, (unsigned short)0);
            std::vector<char> taken((size_t)CPG_GENI_NREGS * 64, 0);
            for (int t = 0; ok && t < CPG_GENI_NSTEPS; t++) {
                const int e = steps[t][0], cnt = steps[t][1], reg = steps[t][2], sh = steps[t][3];
                if (reg < 0 || reg >= CPG_GENI_NREGS) { ok = false; break; }
                for (int l = 0; l < cnt; l++) {
                    const int kind = r->sol_kind[e + l], idx = r->sol_idx[e + l];
                    if (kind < 0 || kind > 3 || idx < 0 || idx >= (1 << 28)) { ok = false; break; }
                    const size_t at = (size_t)reg * 64 + (size_t)(l + sh);
                    if (taken[at]) { ok = false; break; }                // (two steps on one lane of a register)
                    taken[at] = 1;
                    gsrc[at] = ((unsigned)kind << 28) | (unsigned)idx;
                    if (kind == 2) {
                        if (idx >= r->nnzL || r->Lcol[idx] < 0 || r->Lcol[idx] > 0xFFFF) { ok = false; break; }
                        glcol[at] = (unsigned short)r->Lcol[idx];
                    }
                }
            }
        }
#ifdef CPG_GENI_FAC_NSTEPS
        if (ok) {
            // tables of the generated factorisation: its header fixes (steps, level end, group width) per chunk and the term
            // counts per lane (fingerprint); the operand positions and destinations are packed from the plan handed in here
            static const int fch[][3] = CPG_GENI_FAC_CHUNKS;
            const size_t nd = (size_t)r->nnzL + N;
            unsigned hsh = 0x811C9DC5u;
            auto mix = [&](unsigned v) { for (int k = 0; k < 4; k++) { hsh = (hsh ^ ((v >> (8 * k)) & 0xFFu)) * 0x01000193u; } };
            ok = r->fac_chunks == CPG_GENI_FAC_NCHUNKS && nd == (size_t)CPG_GENI_FAC_ZERO && nd < 0xFFFFu;
            for (int c = 0; ok && c < r->fac_chunks; c++) { mix((unsigned)r->fac_ctab[4 * c]); mix((unsigned)r->fac_ctab[4 * c + 1]); mix((unsigned)r->fac_ctab[4 * c + 3]); }
            for (size_t e = 0; ok && e < (size_t)r->fac_chunks * 64; e++) mix(r->fac_len[e]);
            ok = ok && hsh == CPG_GENI_FAC_FINGERPRINT;
            if (ok) {
                const unsigned long long Z = (unsigned long long)nd;
                gtri.assign((size_t)CPG_GENI_FAC_NSTEPS * 64, Z | (Z << 16) | (Z << 32));
                gdk.assign((size_t)r->fac_chunks * 64, 0xFFFFu);
                size_t step = 0;
                for (int c = 0; ok && c < r->fac_chunks; c++) {
                    const int L = r->fac_ctab[4 * c];
                    size_t base = (size_t)r->fac_ctab[4 * c + 2];
                    if (L != fch[c][0]) { ok = false; break; }
                    for (int s = 0; ok && s < L; s++, step++) {
                        int cnt = 0;
                        for (int l = 0; l < 64; l++) {
                            const unsigned ln = r->fac_len[(size_t)c * 64 + l];
                            if ((int)(ln & 0xFFFFu) <= s) continue;
                            const size_t e = base + (size_t)l;        // (lanes with entries at step s form a prefix of the chunk)
                            if (l != cnt || e >= (size_t)r->fac_triples) { ok = false; break; }
                            cnt++;
                            if ((int)(ln >> 16) > s) {
                                const unsigned a = r->fac_a[e], b = r->fac_b[e], k = r->fac_k[e];
                                if (a >= (unsigned)r->nnzL || b >= (unsigned)r->nnzL || k >= (unsigned)N) { ok = false; break; }
                                gtri[step * 64 + (size_t)l] = (unsigned long long)a | ((unsigned long long)b << 16) | ((unsigned long long)((unsigned)r->nnzL + k) << 32);
                            }
                        }
                        base += (size_t)cnt;
                    }
                    for (int l = 0; ok && l < 64; l++) {
                        const unsigned t = r->fac_task[(size_t)c * 64 + l];
                        if (t == 0xFFFFFFFFu) continue;
                        const int kr = fkrow_cl[(size_t)c * 64 + l];
                        gdk[(size_t)c * 64 + l] = t | ((unsigned)(kr + 1) << 16);
                    }
                }
                if (ok && step != (size_t)CPG_GENI_FAC_NSTEPS) ok = false;
            }
            if (ok) {
                if ((rc = upload<unsigned long long>(h, own, gtri.data(), gtri.size(), &R.gf_tri))) return rc;
                if ((rc = upload<unsigned>(h, own, gdk.data(), gdk.size(), &R.gf_dk))) return rc;
            }
        }
#endif
        if (ok) {
            if ((rc = upload<unsigned short>(h, own, glcol.data(), glcol.size(), &R.gi_lcol))) return rc;
            if ((rc = upload<unsigned short>(h, own, gcols.data(), gcols.size(), &R.gi_cols))) return rc;
            if ((rc = upload<unsigned short>(h, own, grows.data(), grows.size(), &R.gi_rows))) return rc;
            if ((rc = upload<unsigned>(h, own, gsrc.data(), gsrc.size(), &R.gi_src))) return rc;
            R.gi_ok = 1;
        }
    }
#endif
    if ((rc = rt_sync(h))) return rc;
    h->refactor_mode = true;
    h->have_update = true;
    return CPG_OK;
}

#if defined(CPG_GENT_HEADER)
#define CPG_RX(x) CPG_GENT_##x
#define CPG_RX_PRESENT 1
#define CPG_RX_FAC 1
#elif defined(CPG_GENR_HEADER)
#define CPG_RX(x) CPG_GENR_##x
#define CPG_RX_PRESENT 1
#ifdef CPG_GENR_FAC_NSTEPS
#define CPG_RX_FAC 1
#endif
#endif
#ifdef CPG_GENT_HEADER
// Register images of a team executor (codegen.emit_team_program): per wavefront of the team the operand offsets of ITS steps
// (two 16-bit byte offsets per word, step j of the wavefront in half j & 1 of word j / 2) and the output slots of ITS
// chunks, from the [step / 4][lane][4] / [chunk / 4][lane][4] tables generated_tables builds; unused halves: the zero slot /
// a dummy slot.
static bool team_tables(const std::vector<unsigned short> &gcols, const std::vector<unsigned short> &grows, int T, int n_chunks,
                        const int *step_wave, const int *step_local, const int *chunk_wave, const int *chunk_local, int W, int n_off, int n_row,
                        int n_slots, std::vector<unsigned> &toff, std::vector<unsigned> &trow) {
    const unsigned zero_off = (unsigned)(n_slots + CPG_GEN_DUMMY_SLOTS) * 8u, dummy = (unsigned)n_slots;
    toff.assign((size_t)W * n_off * 64, zero_off | (zero_off << 16));
    trow.assign((size_t)W * n_row * 64, dummy | (dummy << 16));
    for (int t = 0; t < T; t++) {
        const int wv = step_wave[t], j = step_local[t];
        if (wv < 0 || wv >= W || j < 0 || j / 2 >= n_off) return false;
        for (int l = 0; l < 64; l++) {
            unsigned &wd = toff[((size_t)wv * n_off + (size_t)(j / 2)) * 64 + l];
            const unsigned v = gcols[((size_t)(t / 4) * 64 + l) * 4 + (t % 4)];
            wd = (j & 1) ? ((wd & 0xFFFFu) | (v << 16)) : ((wd & 0xFFFF0000u) | v);
        }
    }
    for (int c = 0; c < n_chunks; c++) {
        const int wv = chunk_wave[c], j = chunk_local[c];
        if (wv < 0 || wv >= W || j < 0 || j / 2 >= n_row) return false;
        for (int l = 0; l < 64; l++) {
            unsigned &wd = trow[((size_t)wv * n_row + (size_t)(j / 2)) * 64 + l];
            const unsigned v = grows[((size_t)(c >> 2) * 64 + l) * 4 + (c & 3)];
            wd = (j & 1) ? ((wd & 0xFFFFu) | (v << 16)) : ((wd & 0xFFFF0000u) | v);
        }
    }
    return true;
}
#endif
// Tables of the resident per-instance factor kernel (cpg_osqp_resident.h); see include/cpg_hip.h.
int cpg_hip_set_resident(cpg_handle_t h, const cpg_osqp_refactor_t *r, const cpg_osqp_resident_t *rs) {
    int rc = cpg_hip_set_refactor(h, r);
    if (rc) return rc;
    free_list(h->resident_owned);
    h->Rs = cpg::DevResident{};
    if (!rs) { set_error("null argument"); return CPG_E_BADARG; }
#ifdef CPG_RX_PRESENT
    if (r->shared_mats) return CPG_OK;
    std::vector<void *> &own = h->resident_owned;
    cpg::DevResident &Rs = h->Rs;
    const int n = h->F.n, m = h->F.m, N = n + m, nnzL = r->nnzL;
    if (n != CPG_RX(N) || m != CPG_RX(M) || h->F.n_eq != CPG_RX(NEQ) || r->nnzA != CPG_RX(NNZA) || r->nnzP != CPG_RX(NNZP) ||
        nnzL != CPG_RX(NNZL) || rs->nnzX != CPG_RX(NNZX) ||
        rs->sol_chunks != CPG_RX(NCHUNKS) || rs->sol_nnz != CPG_RX(NNZ) || rs->sol_slots != CPG_RX(NSLOTS) ||
        program_fingerprint(rs->sol_ctab, rs->sol_desc, rs->sol_cols, rs->sol_chunks, rs->sol_nnz) != CPG_RX(FINGERPRINT))
        return CPG_OK;                                    // another family's library: the streaming kernel serves this handle
    static const int steps[][4] = CPG_RX(STEPS);         // {first entry, active lanes, coefficient register, lane shift}
    static const int chunk_shift[] = CPG_RX(CHUNK_SHIFT);
    std::vector<unsigned short> gcols, grows, glcol;
    std::vector<unsigned> gsrc;
    if (!generated_tables(rs->sol_ctab, rs->sol_desc, rs->sol_cols, rs->sol_chunks, rs->sol_nnz, rs->sol_slots, steps, CPG_RX(NSTEPS), gcols, grows, chunk_shift)) {
        set_error("cpg_hip_set_resident: the merged program does not fit the generated executor's tables"); return CPG_E_BADARG; }
#ifdef CPG_GENT_HEADER
    // team kernel: coefficient registers, operand offsets and output slots per WAVEFRONT of the team
    static const int step_wave[] = CPG_GENT_STEP_WAVE, step_local[] = CPG_GENT_STEP_LOCAL, chunk_wave[] = CPG_GENT_CHUNK_WAVE, chunk_local[] = CPG_GENT_CHUNK_LOCAL;
    constexpr int TW = CPG_GENT_W;
    std::vector<unsigned> toff, trow;
    if (!team_tables(gcols, grows, CPG_GENT_NSTEPS, CPG_GENT_NCHUNKS, step_wave, step_local, chunk_wave, chunk_local, TW, CPG_GENT_NOFF, CPG_GENT_NROW,
                     CPG_GENT_NSLOTS, toff, trow)) { set_error("cpg_hip_set_resident: bad team tables"); return CPG_E_BADARG; }
#else
    constexpr int TW = 1;
    static const int step_wave[1] = {0};
#endif
    gsrc.assign((size_t)TW * CPG_RX(NREGS) * 64, 0u); glcol.assign((size_t)TW * CPG_RX(NREGS) * 64, (unsigned short)0);
    {
        std::vector<char> taken((size_t)TW * CPG_RX(NREGS) * 64, 0);
        for (int t = 0; t < CPG_RX(NSTEPS); t++) {
            const int e = steps[t][0], cnt = steps[t][1], reg = steps[t][2], sh = steps[t][3];
            const int wv = TW > 1 ? step_wave[TW > 1 ? t : 0] : 0;
            if (reg < 0 || reg >= CPG_RX(NREGS) || wv < 0 || wv >= TW) { set_error("cpg_hip_set_resident: coefficient register out of range"); return CPG_E_BADARG; }
            for (int l = 0; l < cnt; l++) {
                const int kind = rs->sol_kind[e + l], idx = rs->sol_idx[e + l], lc = rs->sol_lcol[e + l];
                const size_t at = ((size_t)wv * CPG_RX(NREGS) + (size_t)reg) * 64 + (size_t)(l + sh);
                const bool ok = kind >= 0 && kind <= 4 && idx >= 0 && idx < (1 << 28) && !taken[at] &&
                                (kind != 2 || (idx < nnzL && lc >= 0 && lc < N)) && (kind != 3 || idx < N) && (kind != 4 || idx < rs->nnzX);
                if (!ok) { set_error("cpg_hip_set_resident: bad coefficient source"); return CPG_E_BADARG; }
                taken[at] = 1;
                gsrc[at] = ((unsigned)kind << 28) | (unsigned)idx;
                glcol[at] = (unsigned short)(kind == 2 ? lc : 0);
            }
        }
    }
    // ---- flat factorisation + inverse stream
    const int X0 = nnzL + N, ONE = X0 + rs->nnzX, ZERO = ONE + 1;
    Rs.nnzX = rs->nnzX; Rs.fac_len = ZERO + 1;
    std::vector<unsigned> ksrc((size_t)nnzL + N, 0u), ctl;
    std::vector<cpg::ResEntry> ent;
    for (int d = 0; d < nnzL + N; d++) {
        const int kind = r->ksrc_kind[d];
        const int idx = (kind == CPG_K_P || kind == CPG_K_A || kind == CPG_K_RHO) ? r->ksrc_idx[d] : 0;      // (sigma / none: no operand)
        const int lim = kind == CPG_K_P ? r->nnzP : kind == CPG_K_A ? r->nnzA : kind == CPG_K_RHO ? m : 1;
        if (kind < 0 || kind > 4 || idx < 0 || idx >= (lim > 0 ? lim : 1)) { set_error("cpg_hip_set_resident: KKT source table out of range"); return CPG_E_BADARG; }
        ksrc[d] = ((unsigned)kind << 28) | (unsigned)idx;
    }
    {
        constexpr int DP = CPG_RES_FAC_DEPTH;
        std::vector<char> has_dot((size_t)Rs.fac_len, 0);
        size_t level_first = 0;
        for (int c = 0; c < rs->fac_chunks; c++) {
            const int L = rs->f_ctab[4 * c], last = rs->f_ctab[4 * c + 1] & 1, lg = rs->f_ctab[4 * c + 3];      // (bit 0: level complete; a team plan keeps marks above it)
            unsigned base = (unsigned)rs->f_ctab[4 * c + 2];
            if (L < 0 || lg < 0 || lg > 6) { set_error("cpg_hip_set_resident: bad factorisation chunk"); return CPG_E_BADARG; }
            for (int s = 0; s < L; s++) {
                unsigned cnt = 0;
                const unsigned ebase = (unsigned)ent.size();
                for (int l = 0; l < 64; l++) {
                    const unsigned lw = rs->f_len[(size_t)c * 64 + l];
                    const int al = (int)(lw & 0xFFFFu), rl = (int)(lw >> 16);
                    if (al <= s) continue;
                    if ((unsigned)l != cnt) { set_error("cpg_hip_set_resident: active lanes are not a prefix"); return CPG_E_BADARG; }
                    cpg::ResEntry e{(unsigned)ZERO, (unsigned)ZERO, (unsigned)ZERO, 0xFFFFFFFFu};
                    const unsigned src = base + cnt;
                    if (s < rl) {
                        if (src >= (unsigned)rs->fac_triples) { set_error("cpg_hip_set_resident: triple out of range"); return CPG_E_BADARG; }
                        e.x = rs->f_a[src]; e.y = rs->f_k[src]; e.z = rs->f_b[src];
                        if (e.x >= (unsigned)Rs.fac_len || e.y >= (unsigned)Rs.fac_len || e.z >= (unsigned)Rs.fac_len) { set_error("cpg_hip_set_resident: factor position out of range"); return CPG_E_BADARG; }
                    }
                    if (s == 0) {
                        const unsigned t = rs->f_task[(size_t)c * 64 + l];
                        if (t != 0xFFFFFFFFu) {
                            if ((t & 0x7FFFFFFFu) >= (unsigned)ONE) { set_error("cpg_hip_set_resident: destination out of range"); return CPG_E_BADARG; }
                            e.w = t; has_dot[t & 0x7FFFFFFFu] = 1;
                        }
                    }
                    ent.push_back(e);
                    cnt++;
                }
                base += cnt;
                ctl.push_back(ebase | (cnt << 24));
                ctl.push_back((s == 0 ? 1u : 0u) | (s == L - 1 ? 2u : 0u) | ((unsigned)lg << 4));
            }
            // tasks without a single term: their destination already holds its final value (a pivot: its reciprocal)
            for (int l = 0; l < 64; l++) {
                const unsigned t = rs->f_task[(size_t)c * 64 + l];
                if (t == 0xFFFFFFFFu || (rs->f_len[(size_t)c * 64 + l] & 0xFFFFu)) continue;
                if ((t & 0x7FFFFFFFu) >= (unsigned)(nnzL + N)) { set_error("cpg_hip_set_resident: empty inverse task"); return CPG_E_BADARG; }
                if (t & 0x80000000u) ksrc[t & 0x7FFFFFFFu] |= 0x80000000u;
            }
            if (last) {
                if (ctl.size() > level_first) ctl[ctl.size() - 1] |= 4u;       // level complete behind the last step emitted in it
                level_first = ctl.size();
            }
        }
        if (ent.size() >= (1u << 24)) { set_error("cpg_hip_set_resident: schedule too long"); return CPG_E_BADARG; }
        Rs.f_dummy = (unsigned)ent.size();
        ent.push_back(cpg::ResEntry{(unsigned)ZERO, (unsigned)ZERO, (unsigned)ZERO, 0xFFFFFFFFu});
        while ((ctl.size() / 2) % DP) { ctl.push_back(Rs.f_dummy); ctl.push_back(0u); }
        Rs.fac_steps = (int)(ctl.size() / 2);
        for (int t = 0; t < 2 * DP; t++) { ctl.push_back(Rs.f_dummy); ctl.push_back(0u); }
    }
#ifdef CPG_RX_FAC
    // ---- tables of the generated factorisation: the header fixes (steps, level end, group width) per chunk and the per-lane
    //      term counts (fingerprint); positions and destinations are packed from the plan handed in here
    std::vector<unsigned long long> gftri;
    std::vector<unsigned> gfdk;
    {
        static const int fch[][3] = CPG_RX(FAC_CHUNKS);
        unsigned hsh = 0x811C9DC5u;
        auto mix = [&](unsigned v) { for (int k = 0; k < 4; k++) { hsh = (hsh ^ ((v >> (8 * k)) & 0xFFu)) * 0x01000193u; } };
        bool ok = rs->fac_chunks == CPG_RX(FAC_NCHUNKS) && ZERO == CPG_RX(FAC_ZERO) && Rs.fac_len < 0xFFFF;
        for (int c = 0; ok && c < rs->fac_chunks; c++) { mix((unsigned)rs->f_ctab[4 * c]); mix((unsigned)rs->f_ctab[4 * c + 1]); mix((unsigned)rs->f_ctab[4 * c + 3]); }
        for (size_t e = 0; ok && e < (size_t)rs->fac_chunks * 64; e++) mix(rs->f_len[e]);
        if (!ok || hsh != CPG_RX(FAC_FINGERPRINT)) { set_error("cpg_hip_set_resident: this library's factorisation was generated for a different family"); return CPG_E_BADARG; }
        const unsigned long long Z = (unsigned long long)ZERO;
        gftri.assign((size_t)CPG_RX(FAC_NSTEPS) * 64, Z | (Z << 16) | (Z << 32));
        gfdk.assign((size_t)rs->fac_chunks * 64, 0xFFFFu);
        size_t step = 0;
        for (int c = 0; c < rs->fac_chunks; c++) {
            const int L = rs->f_ctab[4 * c];
            size_t base = (size_t)rs->f_ctab[4 * c + 2];
            if (L != fch[c][0]) { set_error("cpg_hip_set_resident: factorisation chunk differs from the generated one"); return CPG_E_BADARG; }
            for (int s = 0; s < L; s++, step++) {
                int cnt = 0;
                for (int l = 0; l < 64; l++) {
                    const unsigned lw = rs->f_len[(size_t)c * 64 + l];
                    if ((int)(lw & 0xFFFFu) <= s) continue;
                    const size_t e = base + (size_t)l;            // (validated above: active lanes are a prefix, positions in range)
                    cnt++;
                    if ((int)(lw >> 16) > s)
                        gftri[step * 64 + (size_t)l] = (unsigned long long)rs->f_a[e] | ((unsigned long long)rs->f_b[e] << 16) | ((unsigned long long)rs->f_k[e] << 32);
                }
                base += (size_t)cnt;
            }
            for (int l = 0; l < 64; l++) {
                const unsigned t = rs->f_task[(size_t)c * 64 + l];
                if (t == 0xFFFFFFFFu || !(rs->f_len[(size_t)c * 64 + l] & 0xFFFFu)) continue;       // (no term: the destination is final)
                gfdk[(size_t)c * 64 + l] = (t & 0xFFFFu) | ((t & 0x80000000u) ? 0x10000u : 0u);
            }
        }
        if (step != (size_t)CPG_RX(FAC_NSTEPS)) { set_error("cpg_hip_set_resident: factorisation steps differ from the generated ones"); return CPG_E_BADARG; }
    }
#endif
#ifdef CPG_GENT_HEADER
    // ---- the same schedule for the team's batched table walk (team_factor_batched): batches of CPG_TEAM_FAC_BATCH steps, one list per wavefront;
    //      the LDL' part (the first r->fac_chunks chunks) on wavefront 0, every level of the block inverses spread over the team
    std::vector<unsigned> bfhdr((size_t)2 * TW, 0u), bfctl, bfdk, bftri;
    {
        constexpr int DPF = CPG_TEAM_FAC_DEPTH, SB = CPG_TEAM_FAC_BATCH;
        const unsigned long long Zp = (unsigned long long)ZERO;
        constexpr int NQ = (3 * SB / 2 + 3) / 4, NWD = 3 * SB / 2;
        struct WaveList { std::vector<unsigned> ctl, dk, tri; };
        std::vector<WaveList> wl((size_t)TW);
        const unsigned Zb = (unsigned)ZERO * 8u;
        // tri of a batch: [quad][lane][4] words; three words per two steps: a | b << 16, k | a' << 16, b' | k' << 16 (element numbers)
        const unsigned Zh = (unsigned)ZERO | ((unsigned)ZERO << 16);
        auto put_batch = [&](WaveList &L, unsigned fl, const unsigned (*words)[3 * SB / 2], const unsigned *dks) {
            L.ctl.push_back(fl);
            for (int l = 0; l < 64; l++) L.dk.push_back(dks ? dks[l] : (Zb | 0x40000000u));       // (flags: or-ed in when the lists are final)
            for (int q = 0; q < NQ; q++) for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) {
                const int wd = 4 * q + k;
                L.tri.push_back((words && wd < NWD) ? words[l][wd] : Zh);
            }
        };
        auto null_batch = [&](WaveList &L, unsigned fl) { put_batch(L, fl, nullptr, nullptr); };
        auto add_chunk = [&](WaveList &L, int c, size_t step0) {
            const int Ls = rs->f_ctab[4 * c], lg = rs->f_ctab[4 * c + 3], nbat = (Ls + SB - 1) / SB;
            unsigned dks[64];
            bool pivot = false;
            for (int l = 0; l < 64; l++) {
                const unsigned g = gfdk[(size_t)c * 64 + l];
                if ((g & 0xFFFFu) == 0xFFFFu) dks[l] = Zb | 0x40000000u;
                else { dks[l] = ((g & 0xFFFFu) * 8u) | ((g >> 16) ? 0x80000000u : 0u); pivot |= (g >> 16) != 0; }
            }
            static unsigned words[64][3 * SB / 2];
            for (int bt = 0; bt < nbat; bt++) {
                for (int l = 0; l < 64; l++) for (int k = 0; k < SB; k += 2) {
                    unsigned f[6];
                    for (int q = 0; q < 2; q++) {
                        const unsigned long long t = SB * bt + k + q < Ls ? gftri[(step0 + (size_t)(SB * bt + k + q)) * 64 + l] : (Zp | (Zp << 16) | (Zp << 32));
                        f[3 * q] = (unsigned)(t & 0xFFFFu); f[3 * q + 1] = (unsigned)((t >> 16) & 0xFFFFu); f[3 * q + 2] = (unsigned)((t >> 32) & 0xFFFFu);
                    }
                    words[l][3 * (k / 2)] = f[0] | (f[1] << 16); words[l][3 * (k / 2) + 1] = f[2] | (f[3] << 16); words[l][3 * (k / 2) + 2] = f[4] | (f[5] << 16);
                }
                put_batch(L, (bt == 0 ? 1u : 0u) | (bt == nbat - 1 ? 2u : 0u) | (pivot ? 16u : 0u) | ((unsigned)lg << 5), words, dks);
            }
        };
        const int ldl = std::min(r->fac_chunks, rs->fac_chunks);
        size_t step = 0;
        // extra bits of a batch's destination words: 0x100000 one more level of the LDL' chain is complete (signal), 0x200000 wait until
        // the number of complete levels in the word's low bits has been reached
        auto mark = [&](WaveList &L, unsigned bits) { for (int l = 0; l < 64; l++) L.dk[L.dk.size() - 64 + l] |= bits; };
        bool marked = false;               // (a plan whose inverse sections follow the chain: resident_plan, CPG_TEAM_GROUP_SECTIONS)
        for (int c = ldl; c < rs->fac_chunks; c++) if (rs->f_ctab[4 * c + 1] & 2) marked = true;
#ifndef CPG_TEAM_FOLLOW_CHAIN
        if (marked) { set_error("cpg_hip_set_resident: a plan whose block inverses follow the chain needs a library built with CPG_TEAM_FOLLOW_CHAIN"); return CPG_E_UNSUPPORTED; }
#endif
        for (int c = 0; c < ldl; c++) {
            const int Ls = rs->f_ctab[4 * c];
            if (Ls > 0) add_chunk(wl[0], c, step);
            step += (size_t)Ls;
            if (rs->f_ctab[4 * c + 1] & 1) {
                if (wl[0].ctl.empty() || (wl[0].ctl.back() & 4u)) null_batch(wl[0], 0u);     // (a level without a term, or two ends in a row)
                wl[0].ctl.back() |= 4u;
                if (TW > 1 && marked) mark(wl[0], 0x100000u);
            }
        }
        // the sections of the block inverses (one per merged group, ends marked by bit 1 of the chunk table's second column; a plan
        // without marks: one section that the team shares level by level behind the chain).  A marked section runs on ONE wavefront,
        // 1 .. W - 1 in turn, and FOLLOWS the chain: in front of a chunk that needs more complete LDL' levels than the wavefront has
        // waited for so far (bits 8 .. of the column) sits a batch without work that waits for that count.
        std::vector<std::pair<int, int>> sections;
        for (int c = ldl, c0 = ldl; c < rs->fac_chunks; c++)
            if ((rs->f_ctab[4 * c + 1] & 2) || c == rs->fac_chunks - 1) { sections.push_back({c0, c + 1}); c0 = c + 1; }
        std::vector<size_t> sec_step(sections.size(), 0);
        { size_t st = step; for (size_t k = 0; k < sections.size(); k++) { sec_step[k] = st; for (int c = sections[k].first; c < sections[k].second; c++) st += (size_t)rs->f_ctab[4 * c]; } }
        std::vector<unsigned> waited((size_t)TW, 0u);
        for (size_t k = 0; marked && k < sections.size(); k++) {
            const int wv = TW > 1 ? 1 + (int)(k % (size_t)(TW - 1)) : 0;
            size_t st = sec_step[k];
            for (int c = sections[k].first; c < sections[k].second; c++) {
                const int Ls = rs->f_ctab[4 * c];
                const unsigned need = (unsigned)rs->f_ctab[4 * c + 1] >> 8;
                if (Ls > 0 && TW > 1 && need > waited[wv]) {
                    if (need >= (1u << 20)) { set_error("cpg_hip_set_resident: level count out of range"); return CPG_E_BADARG; }
                    null_batch(wl[wv], 0u);
                    for (int l = 0; l < 64; l++) wl[wv].dk[wl[wv].dk.size() - 64 + l] = need | 0x200000u | 0x40000000u;
                    waited[wv] = need;
                }
                if (Ls > 0) add_chunk(wl[wv], c, st);
                st += (size_t)Ls;
                if ((rs->f_ctab[4 * c + 1] & 1) && !wl[wv].ctl.empty() && !(wl[wv].dk.back() & 0x200000u)) wl[wv].ctl.back() |= 4u;
            }
        }
        for (int wv = 0; wv < TW; wv++) { if (wl[wv].ctl.empty() || (wl[wv].dk.back() & 0x200000u)) null_batch(wl[wv], 0u); wl[wv].ctl.back() |= 8u; }      // the team meets
        std::vector<char> has((size_t)TW, 0);
        int in_level = 0, level = 0;
        auto close_level = [&]() {
            for (int wv = 0; wv < TW; wv++) { if (has[wv]) wl[wv].ctl.back() |= 8u; else null_batch(wl[wv], 8u); has[wv] = 0; }
            in_level = 0; level++;
        };
        if (!marked && !sections.empty()) {
            size_t st = sec_step[0];
            for (int c = sections.front().first; c < sections.back().second; c++) {
                const int Ls = rs->f_ctab[4 * c];
                if (Ls > 0) { const int wv = (level + in_level) % TW; add_chunk(wl[wv], c, st); has[wv] = 1; in_level++; }
                st += (size_t)Ls;
                if (rs->f_ctab[4 * c + 1] & 1) close_level();
            }
            if (in_level) close_level();
        }
        for (int wv = 0; wv < TW; wv++) {
            while (wl[wv].ctl.size() % DPF) null_batch(wl[wv], 0u);
            bfhdr[2 * wv] = (unsigned)bfctl.size(); bfhdr[2 * wv + 1] = (unsigned)wl[wv].ctl.size();
            for (int k = 0; k < 2 * DPF; k++) null_batch(wl[wv], 0u);       // (what the prefetch ring reads past the end)
            // the flags of a batch into bits 22 - 29 of every lane's destination word (what the kernel reads them from)
            for (size_t bt = 0; bt < wl[wv].ctl.size(); bt++) {
                const unsigned f = wl[wv].ctl[bt];
                if (f > 0xFFu || Zb >= (1u << 20)) { set_error("cpg_hip_set_resident: factorisation batch flags / offsets out of range"); return CPG_E_BADARG; }
                for (int l = 0; l < 64; l++) wl[wv].dk[bt * 64 + l] |= f << 22;
            }
            bfctl.insert(bfctl.end(), wl[wv].ctl.begin(), wl[wv].ctl.end());
            bfdk.insert(bfdk.end(), wl[wv].dk.begin(), wl[wv].dk.end());
            bftri.insert(bftri.end(), wl[wv].tri.begin(), wl[wv].tri.end());
        }
    }
#endif
    // ---- coalesced canonicalisation maps, entry tables
    struct EllHost { std::vector<int> idx; std::vector<double> coef; int J = 0, rows = 0; };
    auto make_ell = [&](const cpg_csr_t &M, int rows, EllHost &E) {
        E.rows = rows > 0 ? rows : 1; E.J = 0;
        if (M.nnz > 0) for (int i = 0; i < rows; i++) E.J = std::max(E.J, M.ptr[i + 1] - M.ptr[i]);
        E.idx.assign((size_t)std::max(E.J, 1) * E.rows, 0); E.coef.assign((size_t)std::max(E.J, 1) * E.rows, 0.0);
        if (M.nnz > 0) for (int i = 0; i < rows; i++)
            for (int k = M.ptr[i]; k < M.ptr[i + 1]; k++) { E.idx[(size_t)(k - M.ptr[i]) * E.rows + i] = M.idx[k]; E.coef[(size_t)(k - M.ptr[i]) * E.rows + i] = M.val[k]; }
    };
    EllHost eP, eA, eq, eu;
    make_ell(r->map_P, r->nnzP, eP); make_ell(r->map_A, r->nnzA, eA); make_ell(r->map_q, n, eq); make_ell(r->map_u, m, eu);
    auto up_ell = [&](const EllHost &E, cpg::DevEll &D) {
        D.J = E.J; D.rows = E.rows;
        int rc2;
        if ((rc2 = upload<int>(h, own, E.idx.data(), E.idx.size(), &D.idx))) return rc2;
        return upload<double>(h, own, E.coef.data(), E.coef.size(), &D.coef);
    };
    if ((rc = up_ell(eP, Rs.eP)) || (rc = up_ell(eA, Rs.eA)) || (rc = up_ell(eq, Rs.eq)) || (rc = up_ell(eu, Rs.eu))) return rc;
    std::vector<unsigned> entA((size_t)std::max(r->nnzA, 1)), entP((size_t)std::max(r->nnzP, 1));
    if (n > 0xFFFF || m > 0xFFFF) { set_error("cpg_hip_set_resident: family too large"); return CPG_E_UNSUPPORTED; }
    for (int j = 0; j < n; j++) {
        for (int k = r->Ap[j]; k < r->Ap[j + 1]; k++) entA[k] = (unsigned)r->Ai[k] | ((unsigned)j << 16);
        for (int k = r->Pp[j]; k < r->Pp[j + 1]; k++) entP[k] = (unsigned)r->Pi[k] | ((unsigned)j << 16);
    }
    // ---- products of the termination test: generated row executors of the family's header, or the streaming executor's layout
    StreamTables st3[3];
    std::vector<int> src3[3];
    std::vector<unsigned short> rcols3[3], rrows3[3];
    const cpg_rows_program_t *rows3[3] = {&rs->rows_A, &rs->rows_P, &rs->rows_At};
    cpg::DevStreamTab *dst3[3] = {&Rs.pA, &Rs.pP, &Rs.pAt};
    const int w_slots = std::max(rs->out_ax + m, rs->out_aty + n);
    for (int k = 0; k < 3; k++) {
        const cpg_rows_program_t &p = *rows3[k];
        for (int c = 0; c < p.n_chunks; c++) if (p.ctab[4 * c + 3] & ~1) { set_error("cpg_hip_set_resident: row program with an unsupported chunk kind"); return CPG_E_BADARG; }
#if defined(CPG_GENRA_NSTEPS) || defined(CPG_GENT_HEADER)
        {
#ifdef CPG_GENT_HEADER
#define CPG_RXR(which, x) CPG_GENT##which##_##x
#else
#define CPG_RXR(which, x) CPG_GENR##which##_##x
#endif
            static const int stA[][4] = CPG_RXR(A, STEPS), stP[][4] = CPG_RXR(P, STEPS), stT[][4] = CPG_RXR(T, STEPS);
            const int (*steps)[4] = k == 0 ? stA : (k == 1 ? stP : stT);
            const int T = k == 0 ? CPG_RXR(A, NSTEPS) : (k == 1 ? CPG_RXR(P, NSTEPS) : CPG_RXR(T, NSTEPS));
            const int nch = k == 0 ? CPG_RXR(A, NCHUNKS) : (k == 1 ? CPG_RXR(P, NCHUNKS) : CPG_RXR(T, NCHUNKS));
            const int nz = k == 0 ? CPG_RXR(A, NNZ) : (k == 1 ? CPG_RXR(P, NNZ) : CPG_RXR(T, NNZ));
            const unsigned fp = k == 0 ? CPG_RXR(A, FINGERPRINT) : (k == 1 ? CPG_RXR(P, FINGERPRINT) : CPG_RXR(T, FINGERPRINT));
            if (p.n_chunks != nch || p.nnz != nz || program_fingerprint(p.ctab, p.desc, p.cols, p.n_chunks, p.nnz) != fp) {
                set_error("cpg_hip_set_resident: this library's row executors were generated for a different family"); return CPG_E_BADARG; }
            // idle lanes gather the zero slot of the substitution program's work vector and store to its dummy slots
            if (!generated_tables(p.ctab, p.desc, p.cols, p.n_chunks, p.nnz, CPG_RX(NSLOTS), steps, T, rcols3[k], rrows3[k])) {
                set_error("cpg_hip_set_resident: row program does not fit the generated executor's tables"); return CPG_E_BADARG; }
            const int lim = k == 1 ? r->nnzP : r->nnzA;
            src3[k].assign((size_t)p.nnz + 64, -1);               // (a step's idle lanes read the entries behind it: finite padding)
            for (int e = 0; e < p.nnz; e++) { const int v = p.ent[e]; if (v >= lim) { set_error("cpg_hip_set_resident: matrix entry out of range"); return CPG_E_BADARG; } src3[k][e] = v; }
            cpg::DevStreamTab &D = *dst3[k];
            D.n_pairs = 0; D.n_entries = (int)src3[k].size(); D.dummy = 0; D.stab = nullptr; D.cr = nullptr;
            if ((rc = upload<int>(h, own, src3[k].data(), src3[k].size(), &D.src))) return rc;
#ifdef CPG_GENT_HEADER
            {
                // the team's row executors read 32-bit register images per wavefront, like the substitution executor
                static const int swA[] = CPG_GENTA_STEP_WAVE, slA[] = CPG_GENTA_STEP_LOCAL, cwA[] = CPG_GENTA_CHUNK_WAVE, clA[] = CPG_GENTA_CHUNK_LOCAL;
                static const int swP[] = CPG_GENTP_STEP_WAVE, slP[] = CPG_GENTP_STEP_LOCAL, cwP[] = CPG_GENTP_CHUNK_WAVE, clP[] = CPG_GENTP_CHUNK_LOCAL;
                static const int swT[] = CPG_GENTT_STEP_WAVE, slT[] = CPG_GENTT_STEP_LOCAL, cwT[] = CPG_GENTT_CHUNK_WAVE, clT[] = CPG_GENTT_CHUNK_LOCAL;
                const int noff = k == 0 ? CPG_GENTA_NOFF : (k == 1 ? CPG_GENTP_NOFF : CPG_GENTT_NOFF), nrow = k == 0 ? CPG_GENTA_NROW : (k == 1 ? CPG_GENTP_NROW : CPG_GENTT_NROW);
                std::vector<unsigned> to, tr_;
                if (!team_tables(rcols3[k], rrows3[k], T, nch, k == 0 ? swA : (k == 1 ? swP : swT), k == 0 ? slA : (k == 1 ? slP : slT),
                                 k == 0 ? cwA : (k == 1 ? cwP : cwT), k == 0 ? clA : (k == 1 ? clP : clT), CPG_GENT_W, noff, nrow, CPG_GENT_NSLOTS, to, tr_)) {
                    set_error("cpg_hip_set_resident: bad team tables of a row program"); return CPG_E_BADARG; }
                const unsigned *d_to = nullptr, *d_tr = nullptr;
                if ((rc = upload<unsigned>(h, own, to.data(), to.size(), &d_to))) return rc;
                if ((rc = upload<unsigned>(h, own, tr_.data(), tr_.size(), &d_tr))) return rc;
                D.gcols = (const unsigned short *)d_to; D.grows = (const unsigned short *)d_tr;
            }
#else
            if ((rc = upload<unsigned short>(h, own, rcols3[k].data(), rcols3[k].size(), &D.gcols))) return rc;
            if ((rc = upload<unsigned short>(h, own, rrows3[k].data(), rrows3[k].size(), &D.grows))) return rc;
#endif
            continue;
        }
#endif
        if ((rc = build_stream_tables(p.ctab, p.desc, p.cols, p.n_chunks, w_slots, st3[k], CPG_RES_PRODUCT_DEPTH / 2))) return rc;
        src3[k].resize(st3[k].src.size());
        const int lim = k == 1 ? r->nnzP : r->nnzA;
        for (size_t e = 0; e < src3[k].size(); e++) {
            const int from = st3[k].src[e];
            int v = -1;
            if (from >= 0) { if (from >= p.nnz) { set_error("cpg_hip_set_resident: row program entry out of range"); return CPG_E_BADARG; } v = p.ent[from]; }
            if (v >= lim) { set_error("cpg_hip_set_resident: matrix entry out of range"); return CPG_E_BADARG; }
            src3[k][e] = v;
        }
        cpg::DevStreamTab &D = *dst3[k];
        D.gcols = nullptr; D.grows = nullptr;
        D.n_pairs = st3[k].n_pairs; D.n_entries = (int)st3[k].cr.size(); D.dummy = (unsigned)st3[k].cr.size() / 2u - 1u;
        if ((rc = upload<unsigned>(h, own, st3[k].st.data(), st3[k].st.size(), &D.stab))) return rc;
        if ((rc = upload<unsigned>(h, own, st3[k].cr.data(), st3[k].cr.size(), &D.cr))) return rc;
        if ((rc = upload<int>(h, own, src3[k].data(), src3[k].size(), &D.src))) return rc;
    }
    Rs.out_ax = rs->out_ax; Rs.out_px = rs->out_px; Rs.out_aty = rs->out_aty;
    const int ldw = CPG_RX(NSLOTS) + CPG_GEN_EXTRA_SLOTS;
    if (rs->out_ax < ldw + N || rs->out_px < ldw + N || rs->out_aty < rs->out_px + n) { set_error("cpg_hip_set_resident: result slots overlap the work vector"); return CPG_E_BADARG; }
    Rs.out_sc = w_slots;                                  // 1 / D | 1 / E of the instance behind the products' results
#ifdef CPG_GENR_TABLES_GLOBAL
    // four wavefronts per CU: 1 / E alone behind the products' results; the set-up's D, E, norms and theta alias the space of the
    // scaled matrices (cpg_osqp_resident.h)
    long long slice = std::max<long long>(Rs.fac_len, (long long)w_slots + m);
    slice = std::max<long long>(slice, std::max<long long>((long long)r->nnzA + r->nnzP, std::max<long long>(r->np_var, (long long)N + std::max(n, m))));
#elif defined(CPG_GENT_HEADER)
    // team kernel: 1 / D | 1 / E are not kept in the slice (the termination test reads them into registers), the set-up's scaling
    // vectors, norms and theta alias the space of the scaled matrices (cpg_osqp_team.h)
    long long slice = std::max<long long>(Rs.fac_len, (long long)w_slots);
    slice = std::max<long long>(slice, std::max<long long>((long long)r->nnzA + r->nnzP, std::max<long long>(r->np_var, (long long)N + std::max(n, m))));
#else
    long long slice = std::max<long long>(Rs.fac_len, (long long)w_slots + N);
    slice = std::max<long long>(slice, (long long)r->nnzA + r->nnzP + std::max<long long>(r->np_var, (long long)N + std::max(n, m)));
#endif
    slice += slice & 1;
#ifdef CPG_GENT_HEADER
    // (one slice per workgroup behind the team's scratch; the factor positions are 16-bit ELEMENT numbers, the executor's
    // operand offsets 16-bit byte offsets into the work vector at the front of the slice)
    if (Rs.fac_len >= 0xFFFF || ((long long)w_slots + N) * 8 > 0x7FFFFFFFLL) { set_error("cpg_hip_set_resident: team slice too large"); return CPG_E_UNSUPPORTED; }
#else
    if (slice * 8 > 0xFFFF) { set_error("cpg_hip_set_resident: LDS slice beyond 16-bit offsets"); return CPG_E_UNSUPPORTED; }
#endif
    Rs.slice_doubles = (int)slice;
    Rs.buf_doubles = (long long)r->nnzA + r->nnzP + 3LL * n + 4LL * m + Rs.pA.n_entries + Rs.pP.n_entries + Rs.pAt.n_entries + 64LL * TW * CPG_RX(NREGS) + 64;
    Rs.gf_tri = nullptr; Rs.gf_dk = nullptr;
#ifdef CPG_RX_FAC
    if ((rc = upload<unsigned long long>(h, own, gftri.data(), gftri.size(), &Rs.gf_tri))) return rc;
    if ((rc = upload<unsigned>(h, own, gfdk.data(), gfdk.size(), &Rs.gf_dk))) return rc;
#endif
    if ((rc = upload<unsigned>(h, own, ctl.data(), ctl.size(), &Rs.f_ctl))) return rc;
    if ((rc = upload<cpg::ResEntry>(h, own, ent.data(), ent.size(), &Rs.f_ent))) return rc;
    if ((rc = upload<unsigned>(h, own, ksrc.data(), ksrc.size(), &Rs.k_src))) return rc;
    if ((rc = upload<unsigned>(h, own, gsrc.data(), gsrc.size(), &Rs.g_src))) return rc;
    if ((rc = upload<unsigned short>(h, own, glcol.data(), glcol.size(), &Rs.g_lcol))) return rc;
    {
        // the same sources as POSITIONS in the factor array [M | 1 / d | X | 1.0 | 0.0], two words per (register, lane): a coefficient is
        // +-(fac[a] * fac[b]) -- -M_ij * (1 / d_j), (1 / d_i) * 1.0, X_ij * 1.0, 1.0 * 1.0, 0.0 * 1.0 -- read without a branch on its kind
        std::vector<unsigned> gpos(2 * gsrc.size());
        for (size_t at = 0; at < gsrc.size(); at++) {
            const unsigned kind = gsrc[at] >> 28, idx = gsrc[at] & 0x0FFFFFFFu;
            const unsigned a = kind == 2u ? idx : kind == 3u ? (unsigned)nnzL + idx : kind == 4u ? (unsigned)X0 + idx : kind == 1u ? (unsigned)ONE : (unsigned)ZERO;
            const unsigned b = kind == 2u ? (unsigned)nnzL + glcol[at] : (unsigned)ONE;
            gpos[2 * at] = a | (kind == 2u ? 0x80000000u : 0u); gpos[2 * at + 1] = b;
        }
        if ((rc = upload<unsigned>(h, own, gpos.data(), gpos.size(), &Rs.g_pos))) return rc;
    }
    if ((rc = upload<unsigned short>(h, own, gcols.data(), gcols.size(), &Rs.g_cols))) return rc;
    if ((rc = upload<unsigned short>(h, own, grows.data(), grows.size(), &Rs.g_rows))) return rc;
    if ((rc = upload<unsigned>(h, own, entA.data(), entA.size(), &Rs.entA))) return rc;
    if ((rc = upload<unsigned>(h, own, entP.data(), entP.size(), &Rs.entP))) return rc;
#ifdef CPG_GENT_HEADER
    if ((rc = upload<unsigned>(h, own, bfhdr.data(), bfhdr.size(), &Rs.bf_hdr))) return rc;
    if ((rc = upload<unsigned>(h, own, bfctl.data(), bfctl.size(), &Rs.bf_ctl))) return rc;
    if ((rc = upload<unsigned>(h, own, bfdk.data(), bfdk.size(), &Rs.bf_dk))) return rc;
    if ((rc = upload<unsigned>(h, own, bftri.data(), bftri.size(), &Rs.bf_tri))) return rc;
    if ((rc = upload<unsigned>(h, own, toff.data(), toff.size(), &Rs.t_off))) return rc;
    if ((rc = upload<unsigned>(h, own, trow.data(), trow.size(), &Rs.t_row))) return rc;
#endif
    if ((rc = rt_sync(h))) return rc;
#ifdef CPG_GENT_HEADER
    Rs.ok = 2;          // the team kernel's tables
#else
    Rs.ok = 1;
#endif
#endif
    return CPG_OK;
}

int cpg_hip_set_handover(cpg_handle_t h, cpg_handle_t per_instance) {
    if (!h || h->conic || (per_instance && per_instance->conic)) { set_error("cpg_hip_set_handover: OSQP handles only"); return CPG_E_BADARG; }
    if (per_instance && (per_instance->device != h->device || per_instance->F.n != h->F.n || per_instance->F.m != h->F.m ||
                         per_instance->F.n_prim != h->F.n_prim || per_instance->F.n_dual != h->F.n_dual)) {
        set_error("cpg_hip_set_handover: the two handles must describe the same family on the same device"); return CPG_E_BADARG; }
    h->linked = per_instance;
    return CPG_OK;
}

int cpg_hip_last_phase_ms(cpg_handle_t h, float *ms_shared, float *ms_per_instance, int64_t *n_handed_over) {
    if (!h || !ms_shared || !ms_per_instance || !n_handed_over) { set_error("null argument"); return CPG_E_BADARG; }
    RT_CHECK(hipEventSynchronize(h->ev1));
    *ms_per_instance = 0.f; *n_handed_over = 0;
    if (!h->two_phase_last) { RT_CHECK(hipEventElapsedTime(ms_shared, h->ev0, h->ev1)); return CPG_OK; }
    RT_CHECK(hipEventElapsedTime(ms_shared, h->ev0, h->ev_mid));
    RT_CHECK(hipEventElapsedTime(ms_per_instance, h->ev_mid, h->ev1));
    unsigned cnt = 0;
    RT_CHECK(hipMemcpyAsync(&cnt, h->d_counter + 1, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    RT_CHECK(hipStreamSynchronize(h->stream));
    *n_handed_over = (int64_t)cnt;
    return CPG_OK;
}

int cpg_hip_set_gradient(cpg_handle_t h, const cpg_osqp_gradient_t *g) {
    if (h && h->conic) { set_error("not available for a conic (interior-point) handle"); return CPG_E_BADARG; }
    if (!h || !g) { set_error("null argument"); return CPG_E_BADARG; }
    if (!h->refactor_mode && h->refactor_owned.empty()) { set_error("cpg_hip_set_refactor must be called first"); return CPG_E_BADARG; }
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    if ((rc = rt_sync(h))) return rc;
    free_list(h->gradient_owned);
    std::vector<void *> &own = h->gradient_owned;
    h->Gd.NP = g->NP;
    const size_t nt = (size_t)g->tptr[g->NP];
    if ((rc = upload<int>(h, own, g->Pcolidx, h->R.nnzP, &h->Gd.Pcolidx))) return rc;
    if ((rc = upload<int>(h, own, g->Acolidx, h->R.nnzA, &h->Gd.Acolidx))) return rc;
    if ((rc = upload<int>(h, own, g->tptr, (size_t)g->NP + 1, &h->Gd.tptr))) return rc;
    if ((rc = upload<int>(h, own, g->tkind, nt, &h->Gd.tkind))) return rc;
    if ((rc = upload<int>(h, own, g->tidx, nt, &h->Gd.tidx))) return rc;
    if ((rc = upload<double>(h, own, g->tcoef, nt, &h->Gd.tcoef))) return rc;
    if ((rc = rt_sync(h))) return rc;
    h->have_gradient = true;
    return CPG_OK;
}

int cpg_hip_gradient_batch(cpg_handle_t h, int64_t B, const double *theta, const double *sol_x, const double *sol_y,
                           const double *dx, double *dtheta) {
    if (!h || !h->have_gradient) { set_error("cpg_hip_set_gradient has not been called"); return CPG_E_BADARG; }
    if (B < 0 || !sol_x || !sol_y || !dx || !dtheta || (h->R.np_var > 0 && !theta)) { set_error("null buffer"); return CPG_E_BADARG; }
    if (B == 0) return CPG_OK;
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    const size_t b = (size_t)B, n = h->F.n, m = h->F.m, N = n + m;
    if ((rc = ensure(h->g_theta, b * h->R.np_var * sizeof(double)))) return rc;
    if ((rc = ensure(h->g_x, b * n * sizeof(double)))) return rc;
    if ((rc = ensure(h->g_y, b * m * sizeof(double)))) return rc;
    if ((rc = ensure(h->g_dprim, b * n * sizeof(double)))) return rc;
    if ((rc = ensure(h->g_dtheta, b * h->Gd.NP * sizeof(double)))) return rc;
    if ((rc = rt_h2d(h, h->g_theta.p, theta, b * h->R.np_var * sizeof(double)))) return rc;
    if ((rc = rt_h2d(h, h->g_x.p, sol_x, b * n * sizeof(double)))) return rc;
    if ((rc = rt_h2d(h, h->g_y.p, sol_y, b * m * sizeof(double)))) return rc;
    if ((rc = rt_h2d(h, h->g_dprim.p, dx, b * n * sizeof(double)))) return rc;
    // workgroup shape: the LDS vectors of a wavefront bound the residency; W waves per workgroup and
    // per_cu workgroups are chosen to keep as many wavefronts resident as LDS and the register budget
    // (4 x CPG_GRADIENT_WAVES_PER_SIMD per CU) allow -- the kernel waits on dependent loads
    const size_t per_wave = (size_t)h->R.sol_slots + N + n + m + n + m;
    if (4 * per_wave * sizeof(double) > h->lds_limit) { set_error("adjoint work vectors do not fit the LDS"); return CPG_E_UNSUPPORTED; }
    int W = 4, per_cu = 1;
    for (int w = 4; w <= 8; w++) {
        int pc = (int)(h->lds_limit / ((size_t)w * per_wave * sizeof(double)));
        if (pc * w > 4 * CPG_GRADIENT_WAVES_PER_SIMD) pc = 4 * CPG_GRADIENT_WAVES_PER_SIMD / w;
        if (pc >= 1 && pc * w > per_cu * W) { W = w; per_cu = pc; }
    }
    const size_t lds = (size_t)W * per_wave * sizeof(double);
    long long blocks = (B + W - 1) / W;
    const long long cap = (long long)h->num_cu * per_cu;
    if (blocks > cap) blocks = cap;
    if ((rc = ensure(h->scratch, (size_t)blocks * W * (size_t)h->R.buf_doubles * sizeof(double)))) return rc;
    cpg::DevGradBatch Bt;
    Bt.B = B; Bt.theta = (const double *)h->g_theta.p; Bt.sol_x = (const double *)h->g_x.p;
    Bt.sol_y = (const double *)h->g_y.p; Bt.dx = (const double *)h->g_dprim.p; Bt.dtheta = (double *)h->g_dtheta.p;
    Bt.counter = h->d_counter; Bt.scratch = (double *)h->scratch.p;
    RT_CHECK(hipMemsetAsync(h->d_counter, 0, sizeof(unsigned), h->stream));
    RT_CHECK(hipEventRecord(h->ev0, h->stream));
    rc = launch_gradient(h, Bt, (int)blocks, W, lds);
    if (rc) return rc;
    RT_CHECK(hipEventRecord(h->ev1, h->stream));
    if ((rc = rt_d2h(h, dtheta, h->g_dtheta.p, b * h->Gd.NP * sizeof(double)))) return rc;
    return rt_sync(h);
}

int cpg_hip_set_launch(cpg_handle_t h, int waves_per_block, int inst_per_wave, int blocks_per_cu) {
    if (!h) { set_error("null handle"); return CPG_E_BADARG; }
    if (waves_per_block < 0 || waves_per_block > 16 || (inst_per_wave != 0 && inst_per_wave != 1 && inst_per_wave != 2)) {
        set_error("waves_per_block in 0..16, inst_per_wave in {0,1,2}"); return CPG_E_BADARG; }
    h->waves_per_block = waves_per_block;
    h->inst_per_wave = inst_per_wave ? inst_per_wave : 1;
    h->blocks_per_cu = blocks_per_cu;
    return CPG_OK;
}

static int ensure(DevBuf &b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return CPG_OK;
    if (b.p) rt_free(b.p);
    b.p = nullptr; b.bytes = 0;
    int rc = rt_malloc(&b.p, bytes);
    if (rc) return rc;
    b.bytes = bytes;
    return CPG_OK;
}

int cpg_hip_set_program_placement(cpg_handle_t h, int in_lds) {
    if (!h || in_lds < -1 || in_lds > 3) { set_error("in_lds must be -1, 0, 1, 2 or 3"); return CPG_E_BADARG; }
    h->program_in_lds = in_lds;
    return CPG_OK;
}

static cpg::DevBatch make_batch(int64_t B, const double *d_theta, const double *d_state_in, double *d_state_out, double *d_prim,
                                double *d_dual, double *d_obj, int32_t *d_iter, int32_t *d_status, double *d_pri, double *d_dua) {
    cpg::DevBatch Bt;
    Bt.scratch = nullptr; Bt.state_in = d_state_in; Bt.state_out = d_state_out;
    Bt.B = B; Bt.theta = d_theta; Bt.prim = d_prim; Bt.dual = d_dual; Bt.obj = d_obj; Bt.pri_res = d_pri;
    Bt.dua_res = d_dua; Bt.iter = d_iter; Bt.status = d_status; Bt.counter = nullptr;
    Bt.ho_list = nullptr; Bt.ho_count = nullptr; Bt.ho_state = nullptr; Bt.list = nullptr; Bt.list_count = nullptr; Bt.resume = 0;
    return Bt;
}

// per-instance factor kernel of handle `h` (its tables, its scratch) on `stream` with settings `S`
static int launch_per_instance(cpg_handle_t h, rt_stream_t stream, const cpg::DevSettings &S, cpg::DevBatch &Bt) {
    const int W = 4;
#ifdef CPG_GENT_HEADER
    if (h->Rs.ok == 2 && !h->R.shared_mats && h->program_in_lds != 0 && h->program_in_lds != 2) {
        // team kernel: one instance per workgroup; as many workgroups per CU as their LDS (scratch + slice) allows
        const size_t lds = ((size_t)CPG_TEAM_SLICE_OFF + (size_t)h->Rs.slice_doubles) * sizeof(double);
        if (lds <= h->lds_limit) {
            long long per_cu = (long long)(h->lds_limit / lds);
            if (per_cu * CPG_GENT_W > 16) per_cu = 16 / CPG_GENT_W;
            if (CPG_GENT_W <= 4 && per_cu * CPG_GENT_W > 4) per_cu = 4 / CPG_GENT_W;       // (kernels built for one wavefront per SIMD)
            if (h->blocks_per_cu > 0 && per_cu > h->blocks_per_cu) per_cu = h->blocks_per_cu;
            long long blocks = Bt.B;
            if (blocks > (long long)h->num_cu * per_cu) blocks = (long long)h->num_cu * per_cu;
            if (blocks < 1) blocks = 1;
            int rc;
            if ((rc = ensure(h->scratch, (size_t)blocks * (size_t)h->Rs.buf_doubles * sizeof(double)))) return rc;
            Bt.scratch = (double *)h->scratch.p;
            return launch_team(h, stream, S, Bt, (int)blocks, lds);
        }
    }
#endif
#ifdef CPG_GENR_HEADER
    if (h->Rs.ok && !h->R.shared_mats && h->program_in_lds != 0 && h->program_in_lds != 2) {
        // resident kernel: one workgroup per CU, as many wavefronts (<= 4: one per SIMD) as slices fit the LDS
#ifdef CPG_GENR_TABLES_GLOBAL
        const size_t tab = 0;                          // (the executor's tables stay in global memory)
#else
        const size_t tab = (size_t)(((CPG_GENR_NSTEPS + 3) / 4) * 256 + ((CPG_GENR_NCHUNKS + 3) / 4) * 256) * sizeof(unsigned short);
#endif
        const size_t slice = (size_t)h->Rs.slice_doubles * sizeof(double);
        int NW = h->waves_per_block > 0 ? h->waves_per_block : 4;
        if (NW > 4) NW = 4;
        while (NW > 1 && tab + (size_t)NW * slice > h->lds_limit) NW--;
        if (tab + (size_t)NW * slice <= h->lds_limit) {
            long long blocks = (Bt.B + NW - 1) / NW;
            if (blocks > (long long)h->num_cu) blocks = h->num_cu;
            int rc;
            if ((rc = ensure(h->scratch, (size_t)blocks * NW * (size_t)h->Rs.buf_doubles * sizeof(double)))) return rc;
            Bt.scratch = (double *)h->scratch.p;
            const int nsx = (h->F.n + 63) / 64, nsz = (h->F.m + 63) / 64;
#define Z(a, b) if (nsx <= a && nsz <= b) return launch_resident_t<a, b>(h, stream, S, Bt, (int)blocks, NW, tab + (size_t)NW * slice);
            CPG_KERNELS_REFACTOR(Z)
#undef Z
        }
    }
#endif
#ifdef CPG_GENI_HEADER
    if (h->R.gi_ok && h->program_in_lds != 0 && h->program_in_lds != 2) {
        const int W = 8;                               // one workgroup of eight wavefronts per CU shares the tables       // generated instance executor (cpg_hip_set_program_placement(0): the streaming one)
        const size_t tab = (size_t)(((CPG_GENI_NSTEPS + 3) / 4) * 256 + ((CPG_GENI_NCHUNKS + 3) / 4) * 256) * sizeof(unsigned short);
        const size_t nq = (size_t)(h->F.n + h->F.m);
        size_t per_wave = (size_t)(CPG_GENI_NSLOTS + CPG_GEN_EXTRA_SLOTS) + nq + (nq & 1);   // work vector | q | u ...
#ifdef CPG_GENI_FAC_NSTEPS
        const size_t fac = (size_t)h->R.nnzL + nq + 1;                                        // ... or the factor (+ a zero slot) while it is computed
#else
        const size_t fac = (size_t)h->R.nnzL + nq;                                            // ... or the factor while it is computed
#endif
        if (per_wave < fac) per_wave = fac + (fac & 1);
        const size_t lds = tab + (size_t)W * per_wave * sizeof(double);
        if (lds <= h->lds_limit) {
            const int per_cu = 1;                      // 8 wavefronts per CU: the register budget of the kernel
            long long blocks = (Bt.B + W - 1) / W;
            const long long cap = (long long)h->num_cu * per_cu;
            if (blocks > cap) blocks = cap;
            int rc;
            if ((rc = ensure(h->scratch, (size_t)blocks * W * (size_t)h->R.buf_doubles * sizeof(double)))) return rc;
            Bt.scratch = (double *)h->scratch.p;
            const int nsx = (h->F.n + 63) / 64, nsz = (h->F.m + 63) / 64;
#define Z(a, b) if (nsx <= a && nsz <= b) return launch_instance_t<a, b>(h, stream, S, Bt, (int)blocks, W, lds);
            CPG_KERNELS_REFACTOR(Z)
#undef Z
        }
    }
#endif
#ifdef CPG_REFACTOR_CR_LDS
    if (!h->R.shared_mats && h->program_in_lds != 0) {       // (cpg_hip_set_program_placement(0): entry words through L2, as before)
        const int W8 = 8;
        const size_t tab = (((size_t)h->R.sol_nnz + 1) / 2) * sizeof(double);
        const size_t lds8 = tab + (size_t)W8 * h->R.sol_slots * sizeof(double);
        if (lds8 <= h->lds_limit) {
            long long blocks = (Bt.B + W8 - 1) / W8;
            if (blocks > (long long)h->num_cu) blocks = h->num_cu;
            int rc;
            if ((rc = ensure(h->scratch, (size_t)blocks * W8 * (size_t)h->R.buf_doubles * sizeof(double)))) return rc;
            Bt.scratch = (double *)h->scratch.p;
            const int nsx = (h->F.n + 63) / 64, nsz = (h->F.m + 63) / 64;
#define Z(a, b) if (nsx <= a && nsz <= b) return launch_refactor_crlds_t<a, b>(h, stream, S, Bt, (int)blocks, W8, lds8);
            CPG_KERNELS_REFACTOR(Z)
#undef Z
        }
    }
#endif
#ifdef CPG_GENS_HEADER
    if (!h->R.shared_mats && !h->R.gs_ok) {
        set_error("this library's per-instance substitution program was generated for a different family"); return CPG_E_BADARG; }
    const size_t lds = (size_t)W * (size_t)(h->R.shared_mats ? h->R.sol_slots : h->R.sol_slots + CPG_GEN_EXTRA_SLOTS) * sizeof(double);
#else
    const size_t lds = (size_t)W * h->R.sol_slots * sizeof(double);
#endif
    if (lds > h->lds_limit) { set_error("work vectors do not fit the LDS"); return CPG_E_UNSUPPORTED; }
    int per_cu = h->blocks_per_cu > 0 ? h->blocks_per_cu : CPG_REFACTOR_WAVES_PER_SIMD;    // workgroups of 4 waves
    if (per_cu > CPG_REFACTOR_WAVES_PER_SIMD) per_cu = CPG_REFACTOR_WAVES_PER_SIMD;
    if ((long long)per_cu * (long long)lds > (long long)h->lds_limit) per_cu = (int)(h->lds_limit / lds);
    long long blocks = (Bt.B + W - 1) / W;
    const long long cap = (long long)h->num_cu * per_cu;
    if (blocks > cap) blocks = cap;
    int rc;
    if ((rc = ensure(h->scratch, (size_t)blocks * W * (size_t)h->R.buf_doubles * sizeof(double)))) return rc;
    Bt.scratch = (double *)h->scratch.p;
    return launch_refactor(h, stream, S, Bt, (int)blocks, W, lds);
}

int cpg_hip_solve_batch_device_state(cpg_handle_t h, int64_t B, const double *d_theta, const double *d_state_in,
                                     double *d_state_out, double *d_prim, double *d_dual, double *d_obj,
                                     int32_t *d_iter, int32_t *d_status, double *d_pri, double *d_dua) {
    if (!h) { set_error("null handle"); return CPG_E_BADARG; }
    if (h->conic && (d_state_in || d_state_out)) {
        // the reference builds a new Clarabel solver per solve (solvers/clarabel.py:201-204): nothing carries over
        set_error("a conic (interior-point) handle has no state between solves"); return CPG_E_BADARG; }
    if (!h->have_update) { set_error("cpg_hip_set_update has not been called"); return CPG_E_BADARG; }
    if (B < 0 || !d_prim || !d_dual || !d_obj || !d_iter || !d_status || !d_pri || !d_dua || ((h->conic ? h->C.np_var : h->refactor_mode ? h->R.np_var : h->U.np_var) > 0 && !d_theta)) {
        set_error("null buffer"); return CPG_E_BADARG; }
    if (B == 0) return CPG_OK;
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    if (h->conic) {
        const size_t per_wave = (size_t)h->C.lds_doubles * sizeof(double);
        // the kernel is compiled for CPG_CONIC_WAVES_PER_SIMD waves per SIMD (register budget).  The
        // family's index tables get a block-shared LDS copy whenever a workgroup still fits; measured
        // best on MI355X (ADP): workgroups of 8 waves (15.8 ms; 17.3 with 7, 19.4 with 6).
        int W = h->waves_per_block > 0 ? (h->waves_per_block > 8 ? 8 : h->waves_per_block) : 8;
        const size_t tab = (size_t)h->C.tab_doubles * sizeof(double);
        const bool tables_in_lds = h->program_in_lds != 0 && tab + per_wave <= h->lds_limit;
        const size_t fixed = tables_in_lds ? tab : 0;
        while (W > 1 && fixed + (size_t)W * per_wave > h->lds_limit) W--;
        if (h->waves_per_block <= 0) {
            // what counts is the number of resident waves per CU (the kernel is VALU-issue bound from ~12 on): when a
            // second 8-wave workgroup just misses the LDS, two smaller ones beat one (ADP with the previous-iterate
            // copy: 8 + 0 waves at 83.8 KB per workgroup, 7 + 7 at 75.1 KB)
            int best = W, best_res = 0;
            for (int w2 = W; w2 >= 4; w2--) {
                int pc = (int)(h->lds_limit / (fixed + (size_t)w2 * per_wave));
                if (h->blocks_per_cu > 0 && pc > h->blocks_per_cu) pc = h->blocks_per_cu;
                int res = pc * w2; if (res > 4 * CPG_CONIC_WAVES_PER_SIMD) res = 4 * CPG_CONIC_WAVES_PER_SIMD;
                if (res > best_res) { best_res = res; best = w2; }
            }
            W = best;
        }
        // (+ 16 doubles behind the last wavefront's slice: the specialised kernel's per-cone loops are unrolled to the family's largest
        // cone and load past the end of a shorter trailing cone before they mask the use -- inside the allocation with this pad)
        const size_t lds = fixed + (size_t)W * per_wave + (fixed + (size_t)W * per_wave + 128 <= h->lds_limit ? 128 : 0);
        long long blocks = (B + W - 1) / W;
        int per_cu = (int)(h->lds_limit / lds); if (per_cu < 1) per_cu = 1;
        if (h->blocks_per_cu > 0 && per_cu > h->blocks_per_cu) per_cu = h->blocks_per_cu;
        if (per_cu * W > 4 * CPG_CONIC_WAVES_PER_SIMD) per_cu = (4 * CPG_CONIC_WAVES_PER_SIMD) / W;
        if (per_cu < 1) per_cu = 1;
        const long long cap = (long long)h->num_cu * per_cu;
        if (blocks > cap) blocks = cap;
        cpg::DevBatch Bt = make_batch(B, d_theta, nullptr, nullptr, d_prim, d_dual, d_obj, d_iter, d_status, d_pri, d_dua);
        Bt.counter = h->d_counter;
        RT_CHECK(hipMemsetAsync(h->d_counter, 0, sizeof(unsigned), h->stream));
        RT_CHECK(hipEventRecord(h->ev0, h->stream));
        rc = launch_conic(h, Bt, (int)blocks, W, lds, tables_in_lds);
        if (rc) return rc;
        RT_CHECK(hipEventRecord(h->ev1, h->stream));
        return CPG_OK;
    }
    if (h->refactor_mode) {
        cpg::DevBatch Bt = make_batch(B, d_theta, d_state_in, d_state_out, d_prim, d_dual, d_obj, d_iter, d_status, d_pri, d_dua);
        Bt.counter = h->d_counter;
        RT_CHECK(hipMemsetAsync(h->d_counter, 0, 4 * sizeof(unsigned), h->stream));
        RT_CHECK(hipEventRecord(h->ev0, h->stream));
        rc = launch_per_instance(h, h->stream, h->S, Bt);
        if (rc) return rc;
        RT_CHECK(hipEventRecord(h->ev1, h->stream));
        h->two_phase_last = false;
        return CPG_OK;
    }
    // rho adaptation on the shared factor: hybrid execution when a per-instance factor handle is linked
    // (cpg_hip_set_handover); without one the kernel flags the instances whose rho changes (status -2) and the
    // host layer re-solves them through the per-instance factor path
    const bool two_phase = h->S.adaptive_rho && h->S.adaptive_rho_interval > 0 && h->linked != nullptr;
    if (h->S.adaptive_rho && h->S.adaptive_rho_interval > 0 && h->linked == nullptr && !h->flag_rho_changes) {
        // a shared factor cannot follow a rho change: without a linked per-instance factor handle every instance whose rho
        // estimate leaves the tolerance band would come back UNSOLVED under the internal status -2 -- refuse instead
        set_error("rho adaptation is on and this shared-factor handle has no per-instance factor handle linked (cpg_hip_set_handover): "
                  "link one, or turn it off (cpg_hip_set_build_option(h, \"adaptive_rho\", 0)), or accept instances flagged "
                  "CPG_STATUS_NEEDS_REFACTOR (-2) with cpg_hip_set_build_option(h, \"flag_rho_changes\", 1)");
        return CPG_E_UNSUPPORTED;
    }
    if (two_phase && !h->linked->refactor_mode) { set_error("linked handle has no per-instance factor tables (cpg_hip_set_refactor)"); return CPG_E_BADARG; }
    const int G = h->inst_per_wave;
    const size_t N = (size_t)(h->F.n + h->F.m);
#ifdef CPG_GENQ_HEADER
    if (h->squad_ok && G == 1 && h->program_in_lds == 3) {
        // squad executor (on request: it lost the A/B against the LDS-resident program on MI355X, HISTORY.md round 6): CPG_GENQ_W
        // instances per workgroup of CPG_GENQ_W wavefronts, the program in their registers
        const size_t lds_q = cpg::squad_lds_bytes((unsigned)h->F.n, (unsigned)h->F.m);
        if (lds_q <= h->lds_limit) {
            int per_cu = (int)(h->lds_limit / lds_q);
            const int by_regs = 8 / CPG_GENQ_W > 0 ? 8 / CPG_GENQ_W : 1;     // two wavefronts per SIMD
            if (per_cu > by_regs) per_cu = by_regs;
            if (h->blocks_per_cu > 0 && per_cu > h->blocks_per_cu) per_cu = h->blocks_per_cu;
            long long blocks = (B + CPG_GENQ_W - 1) / CPG_GENQ_W;
            const long long cap = (long long)h->num_cu * per_cu;
            if (blocks > cap) blocks = cap;
            cpg::DevBatch Bt = make_batch(B, d_theta, d_state_in, d_state_out, d_prim, d_dual, d_obj, d_iter, d_status, d_pri, d_dua);
            Bt.counter = h->d_counter;
            const size_t state_bytes = (size_t)B * ((size_t)h->F.n + 2 * (size_t)h->F.m + 1) * sizeof(double);
            if (two_phase) {
                if (!d_state_out) { if ((rc = ensure(h->ho_state, state_bytes))) return rc; }
                if ((rc = ensure(h->ho_list, (size_t)B * sizeof(int)))) return rc;
                Bt.ho_state = d_state_out ? d_state_out : (double *)h->ho_state.p;
                Bt.ho_list = (int *)h->ho_list.p; Bt.ho_count = h->d_counter + 1;
            }
            RT_CHECK(hipMemsetAsync(h->d_counter, 0, 4 * sizeof(unsigned), h->stream));
            RT_CHECK(hipEventRecord(h->ev0, h->stream));
            rc = launch_squad(h, Bt, (int)blocks, lds_q);
            if (rc) return rc;
            h->two_phase_last = two_phase;
            if (two_phase) {
                RT_CHECK(hipEventRecord(h->ev_mid, h->stream));
                cpg::DevBatch B2 = make_batch(B, d_theta, Bt.ho_state, d_state_out, d_prim, d_dual, d_obj, d_iter, d_status, d_pri, d_dua);
                B2.counter = h->d_counter + 2; B2.list = Bt.ho_list; B2.list_count = h->d_counter + 1; B2.resume = 1;
                rc = launch_per_instance(h->linked, h->stream, h->S, B2);
                if (rc) return rc;
            }
            RT_CHECK(hipEventRecord(h->ev1, h->stream));
            return CPG_OK;
        }
        if (h->program_in_lds == 3) { set_error("the squad executor's LDS need exceeds the device limit"); return CPG_E_UNSUPPORTED; }
    }
#endif
    if (h->program_in_lds == 3) { set_error("this library carries no squad executor for the family (cpg_hip_set_program_placement(3))"); return CPG_E_UNSUPPORTED; }
#if defined(CPG_GEN_HEADER) && defined(CPG_GEN_N)
    const size_t per_wave = (size_t)G * (h->F.n_slots + CPG_GEN_EXTRA_SLOTS) * sizeof(double);
#else
    const size_t per_wave = (size_t)G * h->F.n_slots * sizeof(double);
#endif
    // LDS-resident program: one workgroup per CU, as many waves as fit next to the program
    const cpg::DevRagged &R = h->F.kkt_ragged;
#ifdef CPG_GEN_HEADER
    const size_t tab_doubles = (size_t)((R.n_chunks + 3) & ~3) * 16;     // 16-bit output-slot table, four chunks per entry group
#else
    const size_t tab_doubles = (size_t)R.n_chunks * 34;     // desc (u32 x 64) + ctab (int x 4)
#endif
#ifdef CPG_GEN_HEADER
    const size_t nnzp = (size_t)R.nnz + CPG_GEN_PAD;
#else
    const size_t nnzp = (size_t)R.nnz;
#endif
#ifdef CPG_GEN_COMPRESSED
    const size_t prog_bytes = R.n_chunks > 0 ? ((size_t)R.n_dict + (nnzp + 1) / 2 + tab_doubles) * 8 : 0;
#else
#ifdef CPG_GEN_PADDED_OFFSETS
    const size_t n_off = (size_t)64 * ((CPG_GEN_PADDED_OFFSETS + 3) & ~3);    // operand offsets of all 64 lanes of every step
#else
    const size_t n_off = nnzp;
#endif
    const size_t prog_bytes = R.n_chunks > 0 ? (nnzp + (n_off + 3) / 4 + tab_doubles) * 8 : 0;
#endif
    bool in_lds = false;
    int W = h->waves_per_block;
    // table-driven kernels, automatic placement: the streaming executor (program through L2, operands
    // of eight steps in flight, more resident waves) beats the LDS-resident table walk -- 1.44 M vs
    // 1.10 M instances/s on MPC 12/4/10; the LDS-resident form remains for G = 2 and on request
    const int placement = h->program_in_lds == 2 ? -1 : h->program_in_lds;     // (2 concerns per-instance factor handles only)
#ifdef CPG_GEN_HEADER
    const bool prefer_stream = false;   // family library: the generated executor works on the LDS-resident program
#else
    const bool prefer_stream = placement == -1 && G == 1 && h->F.kkt_stream.n_pairs > 0;
#endif
    if (placement != 0 && R.n_chunks > 0 && !prefer_stream) {
        const size_t fixed = N * 8 + prog_bytes;
        int wfit = fixed < h->lds_limit ? (int)((h->lds_limit - fixed) / per_wave) : 0;
        // every slot class has an LDS kernel for <= 8 waves (<= 4 for G = 2 on the larger classes);
        // more waves only on explicit request (cpg_hip_set_launch) where such a kernel exists
        int wcap = h->waves_per_block > 0 ? 16 : (G == 2 ? 4 : 8);
#ifdef CPG_GEN_HEADER
        if (h->waves_per_block <= 0) {   // family library: as many waves as its widest compiled kernel admits
            const int nsx_ = (h->F.n + 63) / 64, nsz_ = (h->F.m + 63) / 64;
#define Y(a, b, v, g, wm) if (nsx_ <= a && nsz_ <= b && G == g && wm > wcap) wcap = wm;
            CPG_KERNELS_LDS(Y)
#undef Y
        }
#endif
        if (wfit > wcap) wfit = wcap;
        if (wfit >= 4 || (placement == 1 && wfit >= 1)) {
            in_lds = true;
            if (W <= 0 || W > wfit) W = wfit;
        } else if (placement == 1) {
            set_error("solve program does not fit into LDS next to the work vectors"); return CPG_E_UNSUPPORTED;
        }
    }
    if (!in_lds && (W <= 0 || W > 4)) W = 4;
    const size_t lds = N * 8 + (in_lds ? prog_bytes : 0) + (size_t)W * per_wave;
    if (lds > h->lds_limit) { set_error("work vectors do not fit the 160 KiB LDS; lower waves_per_block / inst_per_wave"); return CPG_E_UNSUPPORTED; }
    const long long ngroups = (B + G - 1) / G;
    int per_cu = h->blocks_per_cu;
    if (in_lds) per_cu = 1;
    else if (per_cu <= 0) {   // as many blocks as LDS and the register budget (CPG_MIN_WAVES_PER_SIMD) admit
        per_cu = (int)(h->lds_limit / (lds ? lds : 1));
        const int by_regs = (CPG_MIN_WAVES_PER_SIMD * 4) / W;
        if (per_cu > by_regs) per_cu = by_regs;
        if (per_cu < 1) per_cu = 1;
    }
    long long blocks = (ngroups + W - 1) / W;
    const long long cap = (long long)h->num_cu * per_cu;
    if (blocks > cap) blocks = cap;
    cpg::DevBatch Bt = make_batch(B, d_theta, d_state_in, d_state_out, d_prim, d_dual, d_obj, d_iter, d_status, d_pri, d_dua);
    Bt.counter = h->d_counter;
    const size_t state_bytes = (size_t)B * ((size_t)h->F.n + 2 * (size_t)h->F.m + 1) * sizeof(double);
    if (two_phase) {
        // hand-over buffers: the workspace of every instance whose rho changes (the caller's state_out rows serve
        // when it gave a buffer: the continuing kernel overwrites them with the final workspace) and their numbers
        if (!d_state_out) { if ((rc = ensure(h->ho_state, state_bytes))) return rc; }
        if ((rc = ensure(h->ho_list, (size_t)B * sizeof(int)))) return rc;
        Bt.ho_state = d_state_out ? d_state_out : (double *)h->ho_state.p;
        Bt.ho_list = (int *)h->ho_list.p; Bt.ho_count = h->d_counter + 1;
    }
    RT_CHECK(hipMemsetAsync(h->d_counter, 0, 4 * sizeof(unsigned), h->stream));
    RT_CHECK(hipEventRecord(h->ev0, h->stream));
    rc = launch(h, Bt, (int)blocks, W, G, lds, in_lds);
    if (rc) return rc;
    h->two_phase_last = two_phase;
    if (two_phase) {
        RT_CHECK(hipEventRecord(h->ev_mid, h->stream));
        cpg::DevBatch B2 = make_batch(B, d_theta, Bt.ho_state, d_state_out, d_prim, d_dual, d_obj, d_iter, d_status, d_pri, d_dua);
        B2.counter = h->d_counter + 2; B2.list = Bt.ho_list; B2.list_count = h->d_counter + 1; B2.resume = 1;
        rc = launch_per_instance(h->linked, h->stream, h->S, B2);
        if (rc) return rc;
    }
    RT_CHECK(hipEventRecord(h->ev1, h->stream));
    return CPG_OK;
}

int cpg_hip_solve_batch_device(cpg_handle_t h, int64_t B, const double *d_theta, double *d_prim, double *d_dual,
                               double *d_obj, int32_t *d_iter, int32_t *d_status, double *d_pri, double *d_dua) {
    return cpg_hip_solve_batch_device_state(h, B, d_theta, nullptr, nullptr, d_prim, d_dual, d_obj, d_iter, d_status, d_pri, d_dua);
}

int cpg_hip_synchronize(cpg_handle_t h) {
    if (!h) { set_error("null handle"); return CPG_E_BADARG; }
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    return rt_sync(h);
}

int cpg_hip_get_stream(cpg_handle_t h, void **stream) {
    if (!h || !stream) { set_error("null argument"); return CPG_E_BADARG; }
    *stream = (void *)h->stream;
    return CPG_OK;
}

int cpg_hip_last_kernel_ms(cpg_handle_t h, float *ms) {
    if (!h || !ms) { set_error("null argument"); return CPG_E_BADARG; }
    RT_CHECK(hipEventSynchronize(h->ev1));
    RT_CHECK(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return CPG_OK;
}

int cpg_hip_solve_batch(cpg_handle_t h, int64_t B, const double *theta, double *prim, double *dual, double *obj,
                        int32_t *iter, int32_t *status, double *pri_res, double *dua_res) {
    return cpg_hip_solve_batch_state(h, B, theta, nullptr, nullptr, prim, dual, obj, iter, status, pri_res, dua_res);
}

int cpg_hip_solve_batch_state(cpg_handle_t h, int64_t B, const double *theta, const double *state_in, double *state_out,
                              double *prim, double *dual, double *obj, int32_t *iter, int32_t *status,
                              double *pri_res, double *dua_res) {
    if (!h) { set_error("null handle"); return CPG_E_BADARG; }
    if (!h->have_update) { set_error("cpg_hip_set_update has not been called"); return CPG_E_BADARG; }
    if (B < 0 || !prim || !dual || !obj || !iter || !status || !pri_res || !dua_res || ((h->conic ? h->C.np_var : h->refactor_mode ? h->R.np_var : h->U.np_var) > 0 && !theta)) {
        set_error("null buffer"); return CPG_E_BADARG; }
    if (B == 0) return CPG_OK;
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    const size_t b = (size_t)B;
    const size_t npv = (size_t)(h->conic ? h->C.np_var : h->refactor_mode ? h->R.np_var : h->U.np_var);
    if ((rc = ensure(h->s_theta, b * npv * sizeof(double)))) return rc;
    if ((rc = ensure(h->s_prim, b * h->F.n_prim * sizeof(double)))) return rc;
    if ((rc = ensure(h->s_dual, b * h->F.n_dual * sizeof(double)))) return rc;
    if ((rc = ensure(h->s_obj, b * sizeof(double)))) return rc;
    if ((rc = ensure(h->s_pri, b * sizeof(double)))) return rc;
    if ((rc = ensure(h->s_dua, b * sizeof(double)))) return rc;
    if ((rc = ensure(h->s_iter, b * sizeof(int32_t)))) return rc;
    if ((rc = ensure(h->s_status, b * sizeof(int32_t)))) return rc;
    if ((rc = rt_h2d(h, h->s_theta.p, theta, b * npv * sizeof(double)))) return rc;
    const size_t state_bytes = b * ((size_t)h->F.n + 2 * (size_t)h->F.m + 1) * sizeof(double);
    if (state_in) { if ((rc = ensure(h->s_state_in, state_bytes))) return rc; if ((rc = rt_h2d(h, h->s_state_in.p, state_in, state_bytes))) return rc; }
    if (state_out) { if ((rc = ensure(h->s_state_out, state_bytes))) return rc; }
    rc = cpg_hip_solve_batch_device_state(h, B, (const double *)h->s_theta.p, state_in ? (const double *)h->s_state_in.p : nullptr,
                                          state_out ? (double *)h->s_state_out.p : nullptr, (double *)h->s_prim.p,
                                          (double *)h->s_dual.p, (double *)h->s_obj.p, (int32_t *)h->s_iter.p,
                                          (int32_t *)h->s_status.p, (double *)h->s_pri.p, (double *)h->s_dua.p);
    if (rc) return rc;
    if (state_out) { if ((rc = rt_d2h(h, state_out, h->s_state_out.p, state_bytes))) return rc; }
    if ((rc = rt_d2h(h, prim, h->s_prim.p, b * h->F.n_prim * sizeof(double)))) return rc;
    if ((rc = rt_d2h(h, dual, h->s_dual.p, b * h->F.n_dual * sizeof(double)))) return rc;
    if ((rc = rt_d2h(h, obj, h->s_obj.p, b * sizeof(double)))) return rc;
    if ((rc = rt_d2h(h, pri_res, h->s_pri.p, b * sizeof(double)))) return rc;
    if ((rc = rt_d2h(h, dua_res, h->s_dua.p, b * sizeof(double)))) return rc;
    if ((rc = rt_d2h(h, iter, h->s_iter.p, b * sizeof(int32_t)))) return rc;
    if ((rc = rt_d2h(h, status, h->s_status.p, b * sizeof(int32_t)))) return rc;
    return rt_sync(h);
}

// ---- streaming many batches from host memory: transfers hidden behind the solve kernel ----------------
// Two sets of device buffers; batch i is copied in on `copy_in` while batch i - 1 is being solved on the
// handle's stream and batch i - 2 is copied out on `copy_out`; events chain the three stages.
struct PipeSet { DevBuf theta, prim, dual, obj, pri, dua, iter, status; hipEvent_t in_done{}, k_done{}, out_done{}; };
struct cpg_pipe_s { hipStream_t copy_in{}, copy_out{}; PipeSet set[2]; bool ready = false; };
static void free_pipe(cpg_pipe_s *p) {
    if (!p) return;
    for (auto &s : p->set) {
        free_buf(s.theta); free_buf(s.prim); free_buf(s.dual); free_buf(s.obj); free_buf(s.pri); free_buf(s.dua);
        free_buf(s.iter); free_buf(s.status);
        if (p->ready) { hipEventDestroy(s.in_done); hipEventDestroy(s.k_done); hipEventDestroy(s.out_done); }
    }
    if (p->ready) { hipStreamDestroy(p->copy_in); hipStreamDestroy(p->copy_out); }
    delete p;
}

int cpg_hip_solve_batches_pipelined(cpg_handle_t h, int64_t B, int32_t n_batches, const double *theta, double *prim,
                                    double *dual, double *obj, int32_t *iter, int32_t *status, double *pri_res,
                                    double *dua_res) {
    if (!h) { set_error("null handle"); return CPG_E_BADARG; }
    if (!h->have_update) { set_error("cpg_hip_set_update has not been called"); return CPG_E_BADARG; }
    const size_t npv = (size_t)(h->conic ? h->C.np_var : h->refactor_mode ? h->R.np_var : h->U.np_var);
    if (B < 0 || n_batches < 0 || !prim || !dual || !obj || !iter || !status || !pri_res || !dua_res || (npv > 0 && !theta)) {
        set_error("null buffer"); return CPG_E_BADARG; }
    if (B == 0 || n_batches == 0) return CPG_OK;
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    if (!h->pipe) {
        // built aside and published only when every stream / event exists: a failed create must not leave a
        // half-initialised pipe behind for the next call
        cpg_pipe_s *np_ = new cpg_pipe_s();
        bool ok = hipStreamCreateWithFlags(&np_->copy_in, hipStreamNonBlocking) == hipSuccess;
        const bool in_ok = ok;
        const bool out_ok = ok && hipStreamCreateWithFlags(&np_->copy_out, hipStreamNonBlocking) == hipSuccess;
        ok = out_ok;
        int n_ev = 0;
        hipEvent_t *evs[6] = {&np_->set[0].in_done, &np_->set[0].k_done, &np_->set[0].out_done,
                              &np_->set[1].in_done, &np_->set[1].k_done, &np_->set[1].out_done};
        for (; ok && n_ev < 6; n_ev++) ok = hipEventCreate(evs[n_ev]) == hipSuccess;
        if (!ok) {
            for (int k = 0; k < n_ev - 1; k++) hipEventDestroy(*evs[k]);
            if (out_ok) hipStreamDestroy(np_->copy_out);
            if (in_ok) hipStreamDestroy(np_->copy_in);
            delete np_;
            set_error("cpg_hip_solve_batches_pipelined: could not create the copy streams / events");
            return CPG_E_HIP;
        }
        np_->ready = true;
        h->pipe = np_;
    }
    cpg_pipe_s &P = *h->pipe;
    const size_t b = (size_t)B, np_ = (size_t)h->F.n_prim, nd = (size_t)h->F.n_dual;
    for (auto &s : P.set) {
        if ((rc = ensure(s.theta, b * npv * 8)) || (rc = ensure(s.prim, b * np_ * 8)) || (rc = ensure(s.dual, b * nd * 8)) ||
            (rc = ensure(s.obj, b * 8)) || (rc = ensure(s.pri, b * 8)) || (rc = ensure(s.dua, b * 8)) ||
            (rc = ensure(s.iter, b * 4)) || (rc = ensure(s.status, b * 4))) return rc;
    }
    for (int i = 0; i < n_batches; i++) {
        PipeSet &s = P.set[i & 1];
        const size_t o = (size_t)i * b;
        if (i >= 2) RT_CHECK(hipStreamWaitEvent(P.copy_in, s.out_done, 0));     // the set's previous results left the device
        if (npv) RT_CHECK(hipMemcpyAsync(s.theta.p, theta + o * npv, b * npv * 8, hipMemcpyHostToDevice, P.copy_in));
        RT_CHECK(hipEventRecord(s.in_done, P.copy_in));
        RT_CHECK(hipStreamWaitEvent(h->stream, s.in_done, 0));
        rc = cpg_hip_solve_batch_device_state(h, B, (const double *)s.theta.p, nullptr, nullptr, (double *)s.prim.p, (double *)s.dual.p,
                                              (double *)s.obj.p, (int32_t *)s.iter.p, (int32_t *)s.status.p, (double *)s.pri.p,
                                              (double *)s.dua.p);
        if (rc) {   // batches already queued copy into the caller's buffers: let them finish before handing the error back
            hipStreamSynchronize(P.copy_in); hipStreamSynchronize(h->stream); hipStreamSynchronize(P.copy_out);
            return rc;
        }
        RT_CHECK(hipEventRecord(s.k_done, h->stream));
        RT_CHECK(hipStreamWaitEvent(P.copy_out, s.k_done, 0));
        RT_CHECK(hipMemcpyAsync(prim + o * np_, s.prim.p, b * np_ * 8, hipMemcpyDeviceToHost, P.copy_out));
        RT_CHECK(hipMemcpyAsync(dual + o * nd, s.dual.p, b * nd * 8, hipMemcpyDeviceToHost, P.copy_out));
        RT_CHECK(hipMemcpyAsync(obj + o, s.obj.p, b * 8, hipMemcpyDeviceToHost, P.copy_out));
        RT_CHECK(hipMemcpyAsync(pri_res + o, s.pri.p, b * 8, hipMemcpyDeviceToHost, P.copy_out));
        RT_CHECK(hipMemcpyAsync(dua_res + o, s.dua.p, b * 8, hipMemcpyDeviceToHost, P.copy_out));
        RT_CHECK(hipMemcpyAsync(iter + o, s.iter.p, b * 4, hipMemcpyDeviceToHost, P.copy_out));
        RT_CHECK(hipMemcpyAsync(status + o, s.status.p, b * 4, hipMemcpyDeviceToHost, P.copy_out));
        RT_CHECK(hipEventRecord(s.out_done, P.copy_out));
    }
    RT_CHECK(hipStreamSynchronize(P.copy_out));
    return rt_sync(h);
}

// page-locked host memory: lets the copies of the pipelined entry point run asynchronously at full PCIe rate
int cpg_hip_host_malloc(cpg_handle_t h, size_t bytes, void **hptr) {
    if (!h || !hptr) { set_error("null argument"); return CPG_E_BADARG; }
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    RT_CHECK(hipHostMalloc(hptr, bytes ? bytes : 8, 0));
    return CPG_OK;
}
int cpg_hip_host_free(cpg_handle_t h, void *hptr) {
    if (!h) { set_error("null handle"); return CPG_E_BADARG; }
    if (hptr) RT_CHECK(hipHostFree(hptr));
    return CPG_OK;
}

int cpg_hip_malloc(cpg_handle_t h, size_t bytes, void **dptr) {
    if (!h || !dptr) { set_error("null argument"); return CPG_E_BADARG; }
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    return rt_malloc(dptr, bytes);
}
int cpg_hip_free(cpg_handle_t h, void *dptr) {
    if (!h) { set_error("null handle"); return CPG_E_BADARG; }
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    return rt_free(dptr);
}
int cpg_hip_memcpy_h2d(cpg_handle_t h, void *dst, const void *src, size_t bytes) {
    if (!h || !dst || !src) { set_error("null argument"); return CPG_E_BADARG; }
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    if ((rc = rt_h2d(h, dst, src, bytes))) return rc;
    return rt_sync(h);
}
int cpg_hip_memcpy_d2h(cpg_handle_t h, void *dst, const void *src, size_t bytes) {
    if (!h || !dst || !src) { set_error("null argument"); return CPG_E_BADARG; }
    int rc = rt_set_device(h->device);
    if (rc) return rc;
    if ((rc = rt_d2h(h, dst, src, bytes))) return rc;
    return rt_sync(h);
}

}  // extern "C"
