// TEST INFRASTRUCTURE ONLY -- stand-in for <hip/hip_runtime.h> used by the lock-step emulator build
// (tests/sim/build_sim.py: g++ -I tests/sim/fake_hip -include tests/sim/cpg_wave_sim.h).  It provides
// just the slice of the HIP runtime API that cvxpygen_amd/csrc/cpg_hip.cpp calls, on host memory:
// device buffers are malloc'ed, copies are memcpy, a kernel launch runs the workgroups one after the
// other, each as blockDim.x host threads (64 per emulated wavefront, see cpg_wave_sim.h).
#pragma once

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

#include "../../cpg_wave_sim.h"

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
typedef void *hipStream_t;
typedef double *hipEvent_t;
enum { hipStreamNonBlocking = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
    int multiProcessorCount;
    size_t sharedMemPerBlock;
    char gcnArchName[64];
};
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__
#define __launch_bounds__(...)

// per-thread launch coordinates and the (single, blocks run one at a time) dynamic LDS window
struct SimDim { unsigned x, y, z; };
inline thread_local SimDim threadIdx, blockIdx, blockDim, gridDim;
alignas(16) inline double cpg_lds[(160 * 1024) / 8 + 16];

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulator error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    p->multiProcessorCount = 2;                      // keeps the emulated grids small
    p->sharedMemPerBlock = 160 * 1024;
    strcpy(p->gcnArchName, "gfx950-emulated");
    return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 8); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t) {
    memcpy(dst, src, bytes); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t) { memset(dst, v, bytes); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (void *)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new double(0.0); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    *e = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*b - *a); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned = 0) { return hipMalloc(p, bytes); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }

template <class Kern, class... Args>
inline void hipLaunchKernelGGL(Kern kern, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t, Args... args) {
    const int waves = (int)(block.x / 64);
    for (unsigned b = 0; b < grid.x; b++) {
        memset(cpg_lds, 0, lds_bytes + 64 <= sizeof(cpg_lds) ? lds_bytes + 64 : sizeof(cpg_lds));
        std::vector<cpgw::SimWave> wv(waves);
        for (auto &w : wv) pthread_barrier_init(&w.bar, nullptr, 64);
        pthread_barrier_t block_bar;
        pthread_barrier_init(&block_bar, nullptr, block.x);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < block.x; t++)
            th.emplace_back([&, t]() {
                threadIdx = {t, 0, 0}; blockIdx = {b, 0, 0}; blockDim = {block.x, 1, 1}; gridDim = {grid.x, 1, 1};
                cpgw::tls.lane = (int)(t & 63); cpgw::tls.wave = (int)(t >> 6); cpgw::tls.block = (int)b;
                cpgw::tls.nblocks = (int)grid.x; cpgw::tls.waves_per_block = waves;
                cpgw::tls.wv = &wv[t >> 6]; cpgw::tls.lds = (char *)cpg_lds; cpgw::tls.block_bar = &block_bar;
                kern(args...);
            });
        for (auto &t : th) t.join();
        for (auto &w : wv) pthread_barrier_destroy(&w.bar);
        pthread_barrier_destroy(&block_bar);
    }
}
