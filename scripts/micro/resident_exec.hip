// Micro-benchmark (round 4): can the per-instance substitution program of a family with matrix
// parameters (portfolio: 172 steps, 144 coefficient register pairs) run with its coefficients in REGISTERS at one
// wavefront per SIMD (unified 512-register file of gfx950: VGPRs + AGPRs), and how long is an ADMM iteration then?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cvxpygen_amd/csrc -DNW=3 scripts/micro/resident_exec.hip -o /tmp/resident_exec
// Build the inputs first: python scripts/micro/resident_gen.py
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include "cpg_osqp_kernel.h"
#include "out/cpg_instance_micro.h"
#include "out/micro_tables.h"

#ifndef NW
#define NW 3
#endif
#ifndef QU_LDS
#define QU_LDS 1
#endif
constexpr int N_ = MICRO_N, M_ = MICRO_M;
constexpr int NSX = (N_ + 63) / 64, NSZ = (M_ + 63) / 64;
constexpr int LDW = CPG_GENI_NSLOTS + CPG_GEN_EXTRA_SLOTS;
constexpr unsigned NCOLS = ((CPG_GENI_NSTEPS + 3u) / 4u) * 256u, NROWS = ((CPG_GENI_NCHUNKS + 3u) / 4u) * 256u;

__global__ void __launch_bounds__(NW * 64)
resident_kernel(const unsigned short *gcols, const unsigned short *grows, const double *cfsrc, const double *qu,
                double *out, int iters, int n_inst, unsigned *counter) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = cpgw::lane_id();
    unsigned short *lc = (unsigned short *)lds, *lr = lc + NCOLS;
    for (unsigned t = cpgw::thread_in_block(); t < NCOLS; t += cpgw::block_threads()) lc[t] = gcols[t];
    for (unsigned t = cpgw::thread_in_block(); t < NROWS; t += cpgw::block_threads()) lr[t] = grows[t];
    cpgw::block_sync();
    double *base = lds + (NCOLS + NROWS) / 4u;
    constexpr unsigned per_wave = LDW + (QU_LDS ? N_ + M_ + ((N_ + M_) & 1) : 0);
    double *w = base + (size_t)cpgw::wave_in_block() * per_wave;
    double *qs = w + LDW, *us = qs + N_;
    const unsigned n = N_, m = M_;
    const double sigma = 1e-6, alpha = 1.6, rho_eq = 100.0, rho_in = 0.1, ri_eq = 0.01, ri_in = 10.0;
    for (;;) {
        unsigned ig = 0;
        if (lane == 0) ig = cpgw::atomic_next(counter);
        ig = (unsigned)cpgw::read_first_lane((int)ig);
        if ((int)ig >= n_inst) break;
        double cf[CPG_GENI_NREGS];
#pragma unroll
        for (int t = 0; t < CPG_GENI_NREGS; t++) cf[t] = cpgw::gld(cfsrc, (unsigned)t * 64u + (unsigned)lane);
        for (unsigned t = (unsigned)lane; t < (unsigned)LDW; t += 64u) w[t] = 0.0;
        double qr[NSX], ur[NSZ];
#pragma unroll
        for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * s; qr[s] = i < n ? cpgw::gld(qu, i) : 0.0; if (QU_LDS && i < n) qs[i] = qr[s]; }
#pragma unroll
        for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * s; ur[s] = i < m ? cpgw::gld(qu, n + i) : 0.0; if (QU_LDS && i < m) us[i] = ur[s]; }
        cpgw::lds_order();
        double x[NSX], z[NSZ], y[NSZ];
#pragma unroll
        for (int s = 0; s < NSX; s++) x[s] = 0.0;
#pragma unroll
        for (int s = 0; s < NSZ; s++) { z[s] = 0.0; y[s] = 0.0; }
#pragma nounroll
        for (int it = 0; it < iters; it++) {
            double qt[NSX];
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * s; qt[s] = QU_LDS ? (i < n ? qs[i] : 0.0) : qr[s]; }
#pragma unroll
            for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * s; if (i < n) w[i] = sigma * x[s] - qt[s]; }
#pragma unroll
            for (int s = 0; s < NSZ; s++) {
                const unsigned i = (unsigned)lane + 64u * s;
                const double ri = i < (unsigned)MICRO_NEQ ? ri_eq : ri_in;
                if (i < m) w[n + i] = z[s] - ri * y[s];
            }
            cpgw::lds_order();
            cpg::run_program_inst(cf, lc, lr, w, lane);
#pragma unroll
            for (int s = 0; s < NSX; s++) {
                const unsigned i = (unsigned)lane + 64u * s;
                if (i < n) x[s] = alpha * w[i] + (1.0 - alpha) * x[s];
            }
#pragma unroll
            for (int s = 0; s < NSZ; s++) {
                const unsigned i = (unsigned)lane + 64u * s;
                if (i < m) {
                    const bool eq = i < (unsigned)MICRO_NEQ;
                    const double rv = eq ? rho_eq : rho_in, ri = eq ? ri_eq : ri_in;
                    const double zp = z[s], yp = y[s];
                    const double zt = (zp - ri * yp) + ri * w[n + i];
                    const double zr = alpha * zt + (1.0 - alpha) * zp;
                    const double uu = QU_LDS ? us[i] : ur[s];
                    const double zn = eq ? uu : cpgw::dmin2(zr + ri * yp, uu);
                    z[s] = zn; y[s] = yp + rv * (zr - zn);
                }
            }
            cpgw::lds_order();
        }
#pragma unroll
        for (int s = 0; s < NSX; s++) { const unsigned i = (unsigned)lane + 64u * s; if (i < n) out[(size_t)ig * (n + 2 * m) + i] = x[s]; }
#pragma unroll
        for (int s = 0; s < NSZ; s++) { const unsigned i = (unsigned)lane + 64u * s; if (i < m) { out[(size_t)ig * (n + 2 * m) + n + i] = z[s]; out[(size_t)ig * (n + 2 * m) + n + m + i] = y[s]; } }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200, n_inst = argc > 2 ? atoi(argv[2]) : 20000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned short *dc, *dr; double *dcf, *dqu, *dout; unsigned *dctr;
    CK(hipMalloc(&dc, sizeof(MICRO_GCOLS))); CK(hipMalloc(&dr, sizeof(MICRO_GROWS)));
    CK(hipMemcpy(dc, MICRO_GCOLS, sizeof(MICRO_GCOLS), hipMemcpyHostToDevice));
    CK(hipMemcpy(dr, MICRO_GROWS, sizeof(MICRO_GROWS), hipMemcpyHostToDevice));
    std::vector<double> cf((size_t)CPG_GENI_NREGS * 64), qu(N_ + M_);
    srand(1);
    for (auto &v : cf) v = 0.02 * ((double)rand() / RAND_MAX - 0.5);
    for (auto &v : qu) v = (double)rand() / RAND_MAX;
    CK(hipMalloc(&dcf, cf.size() * 8)); CK(hipMemcpy(dcf, cf.data(), cf.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dqu, qu.size() * 8)); CK(hipMemcpy(dqu, qu.data(), qu.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, (size_t)n_inst * (N_ + 2 * M_) * 8)); CK(hipMalloc(&dctr, 4));
    const size_t per_wave = LDW + (QU_LDS ? N_ + M_ + ((N_ + M_) & 1) : 0);
    const size_t ldsb = (NCOLS + NROWS) * 2 + (size_t)NW * per_wave * 8;
    printf("NW %d  QU_LDS %d  lds %zu B  regs %d  steps %d  CUs %d\n", NW, QU_LDS, ldsb, CPG_GENI_NREGS, CPG_GENI_NSTEPS, cus);
    CK(hipFuncSetAttribute((const void *)resident_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(dctr, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(resident_kernel, dim3(cus), dim3(NW * 64), ldsb, 0, dc, dr, dcf, dqu, dout, iters, n_inst, dctr);
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double inst_iter = (double)n_inst * iters;
        printf("  %.3f ms  -> %.1f M instance-iterations/s, %.2f us per wave-iteration\n", ms, inst_iter / ms * 1e-3,
               ms * 1e3 * cus * NW / inst_iter);
    }
    std::vector<double> o(8); CK(hipMemcpy(o.data(), dout, 64, hipMemcpyDeviceToHost));
    printf("  out %g %g %g\n", o[0], o[1], o[2]);
    return 0;
}
