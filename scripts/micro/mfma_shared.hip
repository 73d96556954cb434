// Micro-benchmark for the north_star's MFMA clause: the 13-phase shared-factor KKT solve of MPC 12/4/10 as 16 x 4 tiles on
// v_mfma_f64_16x16x4_f64, 16 instances of one wavefront as the N dimension (scripts/micro/mfma_gen.py writes the tile program).
//   A operand: a dense 16 x 4 tile of the phase's matrix, 64 doubles, one per lane (lane i + 16 k holds A[i][k]) -- 591 KiB of
//              tiles for a 73 KiB sparse program, so they are streamed from global memory / L2 (they do not fit the LDS);
//   B operand: 4 slots x 16 instances of the wavefront's work vectors in LDS, [slot][instance] (lane j + 16 k reads slot k, instance j);
//   D        : 16 rows x 16 instances, 4 doubles per lane (lane j + 16 (i / 4), register i % 4), stored to the output slots.
// The work vectors of 16 instances take 117 KiB of the CU's 160 KiB: ONE wavefront per CU.
// Prints: layout self-test of one MFMA, max error of the solve against the host reference, microseconds per solve of 16
// instances (tiles from memory; tiles not loaded at all = the MFMA + LDS bound), on every CU at once.
//   hipcc --offload-arch=gfx950 -O3 mfma_shared.hip -o out/mfma_shared && out/mfma_shared out/mfma_program.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

struct Prog {
    int n_phases, n_tiles, n_rb, n_slots, ni;
    const int *phase_rb;            // [n_phases][2] first row block, count
    const unsigned short *tcols;    // [n_tiles][4] slots of the tile's four columns
    const double *tvals;            // [n_tiles][64]
    const int *rb_tiles;            // [n_rb][2] first tile, count
    const unsigned short *rb_out;   // [n_rb][16] output slots
};

// one instruction on arbitrary per-lane operands: the host works out which matrix element every lane / register holds
__global__ void __launch_bounds__(64) mfma_selftest(const double *a_lane, const double *b_lane, double *d_out) {
    const int l = threadIdx.x;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lane[l], b_lane[l], acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) d_out[l * 4 + v] = acc[v];
}

// layout of the operands, found by the self-test: lane l holds A[ai(l)][ak(l)], B[bk(l)][bj(l)]; register v of lane l holds D[di(l, v)][dj(l)]
struct Layout { int a_mode, b_mode, d_mode; };
__device__ __host__ inline int lay_ai(int m, int l) { return m == 0 ? l % 16 : l / 4; }
__device__ __host__ inline int lay_ak(int m, int l) { return m == 0 ? l / 16 : l % 4; }
__device__ __host__ inline int lay_bk(int m, int l) { return m == 0 ? l / 16 : l % 4; }
__device__ __host__ inline int lay_bj(int m, int l) { return m == 0 ? l % 16 : l / 4; }
__device__ __host__ inline int lay_di(int m, int l, int v) { return m == 0 ? 4 * (l / 16) + v : (m == 1 ? (l / 16) + 4 * v : l % 16); }
__device__ __host__ inline int lay_dj(int m, int l, int v) { return m == 2 ? 4 * (l / 16) + v : l % 16; }

template <bool LOAD_TILES>
__global__ void __launch_bounds__(64) mfma_solve(Prog P, Layout L, const double *w0, double *w_out, int reps, unsigned long long *cyc) {
    extern __shared__ double w[];                 // [n_slots][16] | tile column slots [n_tiles][4] | row block tables
    unsigned short *tc = (unsigned short *)(w + (size_t)P.n_slots * 16);
    unsigned short *ro = tc + 4 * (size_t)P.n_tiles;
    int *rt = (int *)(ro + 16 * (size_t)P.n_rb + 8);
    const int l = threadIdx.x;
    for (int t = l; t < P.n_slots * 16; t += 64) w[t] = w0[t];
    for (int t = l; t < 4 * P.n_tiles; t += 64) tc[t] = P.tcols[t];
    for (int t = l; t < 16 * P.n_rb; t += 64) ro[t] = P.rb_out[t];
    for (int t = l; t < 2 * P.n_rb; t += 64) rt[t] = P.rb_tiles[t];
    __syncthreads();
    const int ai = lay_ai(L.a_mode, l), ak = lay_ak(L.a_mode, l), bk = lay_bk(L.b_mode, l), bj = lay_bj(L.b_mode, l);
    const int a_src = ak * 16 + ai;               // the tile file stores A[i][k] at i + 16 k
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        for (int p = 0; p < P.n_phases; p++) {
            const int rb0 = P.phase_rb[2 * p], nrb = P.phase_rb[2 * p + 1];
            for (int rb = rb0; rb < rb0 + nrb; rb++) {
                const int tl0 = rt[2 * rb], nt = rt[2 * rb + 1];
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                // A tiles two ahead (global memory / L2), B operands from LDS
                double a0 = LOAD_TILES && nt > 0 ? P.tvals[(size_t)tl0 * 64 + a_src] : 1.0;
                double a1 = LOAD_TILES && nt > 1 ? P.tvals[(size_t)(tl0 + 1) * 64 + a_src] : 1.0;
                for (int t = tl0; t < tl0 + nt; t++) {
                    const double a = a0;
                    a0 = a1;
                    if (LOAD_TILES && t + 2 < tl0 + nt) a1 = P.tvals[(size_t)(t + 2) * 64 + a_src];
                    const double b = w[(unsigned)tc[4 * t + bk] * 16u + (unsigned)bj];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
                for (int v = 0; v < 4; v++) w[(unsigned)ro[16 * rb + lay_di(L.d_mode, l, v)] * 16u + (unsigned)lay_dj(L.d_mode, l, v)] = acc[v];
            }
            __syncthreads();
        }
        if (r + 1 < reps) { for (int t = l; t < P.n_slots * 16; t += 64) w[t] = w0[t]; __syncthreads(); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) for (int t = l; t < P.n_slots * 16; t += 64) w_out[t] = w[t];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class T> static T *up(const void *p, size_t n) { T *d; hipMalloc(&d, n * sizeof(T)); hipMemcpy(d, p, n * sizeof(T), hipMemcpyHostToDevice); return d; }

int main(int argc, char **argv) {
    FILE *f = fopen(argc > 1 ? argv[1] : "out/mfma_program.bin", "rb");
    if (!f) { printf("cannot open the tile program (python scripts/micro/mfma_gen.py)\n"); return 1; }
    int hdr[8];
    if (fread(hdr, 4, 8, f) != 8) return 1;
    const int np = hdr[0], T = hdr[1], nrb = hdr[2], ns = hdr[3], ni = hdr[4];
    std::vector<int> phase_rb(2 * np), rb_tiles(2 * nrb);
    std::vector<unsigned short> tcols(4 * (size_t)T), rb_out(16 * (size_t)nrb);
    std::vector<double> tvals(64 * (size_t)T), W0((size_t)ns * ni), Wr((size_t)ns * ni);
    size_t ok = fread(phase_rb.data(), 4, phase_rb.size(), f) + fread(tcols.data(), 2, tcols.size(), f) + fread(tvals.data(), 8, tvals.size(), f) +
                fread(rb_tiles.data(), 4, rb_tiles.size(), f) + fread(rb_out.data(), 2, rb_out.size(), f) + fread(W0.data(), 8, W0.size(), f) + fread(Wr.data(), 8, Wr.size(), f);
    (void)ok; fclose(f);
    // ---- layout self-test of one instruction: which matrix element does every lane / register hold?
    Layout L{0, 0, 0};
    {
        double Am[16][4], Bm[4][16], Dr[16][16], al[64], bl[64], dg[256];
        for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) Am[i][k] = sin(1.0 + 7 * i + 3 * k);
        for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) Bm[k][j] = cos(2.0 + 5 * k + 11 * j);
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += Am[i][k] * Bm[k][j]; Dr[i][j] = s; }
        double best = 1e300;
        double *dA, *dB, *dD; hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 256 * 8);
        for (int am = 0; am < 2; am++) for (int bm = 0; bm < 2; bm++) {
            for (int l = 0; l < 64; l++) { al[l] = Am[lay_ai(am, l)][lay_ak(am, l)]; bl[l] = Bm[lay_bk(bm, l)][lay_bj(bm, l)]; }
            hipMemcpy(dA, al, 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, bl, 64 * 8, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mfma_selftest, dim3(1), dim3(64), 0, 0, dA, dB, dD);
            CK(hipMemcpy(dg, dD, 256 * 8, hipMemcpyDeviceToHost));
            for (int dm = 0; dm < 3; dm++) {
                double e = 0;
                for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) e = fmax(e, fabs(dg[l * 4 + v] - Dr[lay_di(dm, l, v)][lay_dj(dm, l, v)]));
                if (e < best) { best = e; L = Layout{am, bm, dm}; }
            }
        }
        printf("v_mfma_f64_16x16x4_f64 operand layout self-test: A mode %d, B mode %d, D mode %d, max |D - A B| = %.3e\n", L.a_mode, L.b_mode, L.d_mode, best);
    }
    Prog P{np, T, nrb, ns, ni, up<int>(phase_rb.data(), phase_rb.size()), up<unsigned short>(tcols.data(), tcols.size()),
           up<double>(tvals.data(), tvals.size()), up<int>(rb_tiles.data(), rb_tiles.size()), up<unsigned short>(rb_out.data(), rb_out.size())};
    double *dW0 = up<double>(W0.data(), W0.size()), *dWo; hipMalloc(&dWo, W0.size() * 8);
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    unsigned long long *dc; hipMalloc(&dc, 8 * (size_t)cus * 4);
    const size_t lds = (size_t)ns * ni * 8 + 8 * (size_t)T + 32 * (size_t)nrb + 16 + 8 * (size_t)nrb + 64;
    CK(hipFuncSetAttribute((const void *)mfma_solve<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)mfma_solve<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("%d tiles, %d row blocks, %d phases; LDS per wavefront %.1f KiB (16 instances); %d CUs, %.0f MHz\n", T, nrb, np, lds / 1024.0, cus, pr.clockRate / 1e3);
    // ---- correctness: one solve against the host reference
    hipLaunchKernelGGL(mfma_solve<true>, dim3(1), dim3(64), lds, 0, P, L, dW0, dWo, 1, dc);
    std::vector<double> Wo(W0.size());
    CK(hipMemcpy(Wo.data(), dWo, Wo.size() * 8, hipMemcpyDeviceToHost));
    double e = 0, sc = 0; for (size_t i = 0; i < Wo.size(); i++) { e = fmax(e, fabs(Wo[i] - Wr[i])); sc = fmax(sc, fabs(Wr[i])); }
    printf("solve of 16 right-hand sides against the host reference: max error %.3e (scale %.3e)\n", e, sc);
    // ---- timing: every CU runs `blocks_per_cu` wavefronts (one fits the LDS)
    const int reps = 50;
    for (int load = 1; load >= 0; load--) {
        for (int it = 0; it < 2; it++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (load) hipLaunchKernelGGL(mfma_solve<true>, dim3(cus), dim3(64), lds, 0, P, L, dW0, dWo, reps, dc);
            else hipLaunchKernelGGL(mfma_solve<false>, dim3(cus), dim3(64), lds, 0, P, L, dW0, dWo, reps, dc);
            hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (it == 1)
                printf("%s: %.2f us per solve of 16 instances per wavefront (one wavefront per CU, %d CUs busy) = %.3f instance-solves per us and CU\n",
                       load ? "tiles streamed from memory / L2" : "tiles NOT loaded (MFMA + LDS bound)", 1e3 * ms / reps, cus, 16.0 / (1e3 * ms / reps));
        }
    }
    return 0;
}
