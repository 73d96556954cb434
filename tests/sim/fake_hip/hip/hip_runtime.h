// TEST INFRASTRUCTURE ONLY -- stand-in for <hip/hip_runtime.h> used by the lock-step emulator build
// (tests/sim/build_sim.py: g++ -I tests/sim/fake_hip -include tests/sim/cpg_wave_sim.h).  It provides
// just the slice of the HIP runtime API that cvxpygen_amd/csrc/cpg_hip.cpp calls, on host memory:
// device buffers are malloc'ed, copies are memcpy, a kernel launch runs every workgroup on a host thread
// as blockDim.x cooperatively scheduled fibers (64 per emulated wavefront, see cpg_wave_sim.h).
#pragma once

#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <functional>
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
#define __shared__ __thread
#define __launch_bounds__(...)

// launch coordinates of the running fiber; the dynamic LDS window is per host thread = per running workgroup
#define threadIdx (cpgw::cur->tidx)
#define blockIdx (cpgw::cur->bidx)
#define blockDim (cpgw::cur->bdim)
#define gridDim (cpgw::cur->gdim)
alignas(16) inline __thread double cpg_lds[(160 * 1024) / 8 + 16];

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulator error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    p->multiProcessorCount = 4;                      // keeps the emulated grids small
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

namespace cpgsim {
constexpr size_t STACK_BYTES = 2u << 20;          // per fiber; mmap'ed, touched pages only
struct Block {                                    // the fibers of one workgroup, run by one host thread
    std::vector<cpgw::SimThread> th;
    std::vector<cpgw::SimWave> wv;
    cpgw::SimBarrier block_bar;
    std::function<void()> body;
    void *main_sp = nullptr;
    int live = 0;
};
inline thread_local Block *running;

// first frame of every fiber: run the kernel, then leave the ring for good
inline void fiber_main() {
    Block *blk = running;
    blk->body();
    cpgw::SimThread *me = cpgw::cur;
    me->done = true;
    void *dummy;
    if (--blk->live == 0) { cpgw::cpg_sim_switch(&dummy, blk->main_sp); }
    cpgw::SimThread *nx = me->next;
    while (nx->done) nx = nx->next;
    cpgw::cur = nx;
    cpgw::cpg_sim_switch(&dummy, nx->sp);
    abort();
}

inline void run_block(unsigned b, unsigned nblocks, unsigned nthreads, size_t lds_bytes, std::function<void()> body) {
    Block blk;
    const int waves = (int)(nthreads / 64);
    blk.th.resize(nthreads);
    blk.wv.resize(waves);
    for (auto &w : blk.wv) w.bar.n = 64;
    blk.block_bar.n = (int)nthreads;
    blk.body = std::move(body);
    blk.live = (int)nthreads;
    // (the GPU's LDS holds whatever the previous workgroup left; CPG_SIM_LDS_POISON=1 fills it with NaN bit patterns so that a
    // read of a slot nobody wrote shows on the CPU tier instead of on the GPU)
    static const int poison = getenv("CPG_SIM_LDS_POISON") ? atoi(getenv("CPG_SIM_LDS_POISON")) : 0;
    memset(cpg_lds, poison ? 0xFF : 0, lds_bytes + 64 <= sizeof(cpg_lds) ? lds_bytes + 64 : sizeof(cpg_lds));
    char *stacks = (char *)mmap(nullptr, STACK_BYTES * nthreads, PROT_READ | PROT_WRITE,
                                MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == (char *)MAP_FAILED) abort();
    for (unsigned t = 0; t < nthreads; t++) {
        cpgw::SimThread &f = blk.th[t];
        f.lane = (int)(t & 63); f.wave = (int)(t >> 6); f.block = (int)b; f.nblocks = (int)nblocks;
        f.waves_per_block = waves;
        f.tidx = {t, 0, 0}; f.bidx = {b, 0, 0}; f.bdim = {nthreads, 1, 1}; f.gdim = {nblocks, 1, 1};
        f.wv = &blk.wv[t >> 6]; f.block_bar = &blk.block_bar; f.done = false;
        f.next = &blk.th[(t + 1) % nthreads];
        // initial frame for cpg_sim_switch: six callee-saved registers, then the entry point as return address
        uintptr_t top = ((uintptr_t)(stacks + STACK_BYTES * (t + 1))) & ~(uintptr_t)15;
        void **sp = (void **)top;
        *--sp = nullptr;                           // fake return address of fiber_main (never returns)
        *--sp = (void *)&fiber_main;
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        f.sp = (void *)sp;
    }
    running = &blk;
    cpgw::cur = &blk.th[0];
    cpgw::cpg_sim_switch(&blk.main_sp, blk.th[0].sp);
    cpgw::cur = nullptr;
    running = nullptr;
    munmap(stacks, STACK_BYTES * nthreads);
}
}  // namespace cpgsim

template <class Kern, class... Args>
inline void hipLaunchKernelGGL(Kern kern, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t, Args... args) {
    // workgroups are independent: one host thread each, at most the host's core count at a time
    unsigned par = std::thread::hardware_concurrency();
    if (par == 0) par = 1;
    std::atomic<unsigned> next_block{0};
    auto worker = [&]() {
        for (;;) {
            unsigned b = next_block.fetch_add(1);
            if (b >= grid.x) return;
            cpgsim::run_block(b, grid.x, block.x, lds_bytes, [&]() { kern(args...); });
        }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < par && i < grid.x; i++) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
}
