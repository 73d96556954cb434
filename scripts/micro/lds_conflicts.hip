// Micro-benchmark: cost of ds_read_b64 gathers as a function of the per-lane slot pattern (one wavefront per
// CU, eight wavefronts issuing the same pattern; patterns from the host).  Prints cycles per instruction for every pattern, to calibrate the
// bank-conflict model of cvxpygen_amd/slot_layout.py.   hipcc --offload-arch=gfx950 -O3 lds_conflicts.hip -o lds_conflicts
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void __launch_bounds__(512) probe(const unsigned short *pat, int npat, int reps, unsigned long long *cyc, double *sink, int write_mode) {
    __shared__ double w[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 512) w[i] = (double)i;
    __syncthreads();
    double acc = 0.0;
    for (int p = 0; p < npat; p++) {
        const unsigned off = (unsigned)pat[p * 64 + lane] * 8u;
        const char *wb = (const char *)w;
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; r++) {
            if (write_mode) {
                const unsigned aw = (unsigned)(unsigned long long)wb + off;
                const double d = acc + (double)r;
                asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1\n\tds_write_b64 %0, %1\n\tds_write_b64 %0, %1\n\t"
                             "ds_write_b64 %0, %1\n\tds_write_b64 %0, %1\n\tds_write_b64 %0, %1\n\tds_write_b64 %0, %1\n\t"
                             "s_waitcnt lgkmcnt(0)" :: "v"(aw), "v"(d) : "memory");
                continue;
            }
            double v0, v1, v2, v3, v4, v5, v6, v7;
            const unsigned a = (unsigned)(unsigned long long)wb + off;     // LDS byte address
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8\n\tds_read_b64 %2, %8\n\tds_read_b64 %3, %8\n\t"
                         "ds_read_b64 %4, %8\n\tds_read_b64 %5, %8\n\tds_read_b64 %6, %8\n\tds_read_b64 %7, %8\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a));
            acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
        }
        __syncthreads();
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[p] = t1 - t0;
    }
    sink[blockIdx.x * 64 + lane] = acc;
}

int main(int argc, char **argv) {
    // patterns from stdin: npat lines of 64 slot numbers
    std::vector<unsigned short> pat;
    int v;
    while (scanf("%d", &v) == 1) pat.push_back((unsigned short)v);
    const int npat = (int)pat.size() / 64, reps = 256;
    unsigned short *dp; unsigned long long *dc; double *ds;
    hipMalloc(&dp, pat.size() * 2); hipMalloc(&dc, npat * 8); hipMalloc(&ds, 64 * 8);
    hipMemcpy(dp, pat.data(), pat.size() * 2, hipMemcpyHostToDevice);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, dp, npat, reps, dc, ds, argc > 1 && argv[1][0] == 'w');
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(npat);
    hipMemcpy(c.data(), dc, npat * 8, hipMemcpyDeviceToHost);
    for (int p = 0; p < npat; p++) printf("%d %.3f\n", p, (double)c[p] / (reps * 8.0 * 8.0));   // clock ticks per wave-instruction with 8 waves issuing
    return 0;
}
