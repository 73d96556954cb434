// TEST INFRASTRUCTURE ONLY -- host emulation of the wavefront primitives of
// cvxpygen_amd/csrc/cpg_wave_gfx950.h: the 64 lanes of a wavefront run lock-stepped and meet at a barrier
// in every cross-lane primitive.  Every thread of a workgroup is a FIBER (own stack, cooperative switch)
// and all fibers of one workgroup run on one host thread, so a barrier costs 64 user-space context
// switches instead of 64 futex waits; different workgroups run on different host threads
// (fake_hip/hip/hip_runtime.h).  tests/sim/build_sim.py force-includes this header in front of the
// product's cpg_hip.cpp (g++ -include) together with the stand-in <hip/hip_runtime.h> of
// tests/sim/fake_hip, so the CPU-only test tier executes the product's kernel SOURCES through the real
// C-ABI.  Defining the include guard of the gfx950 header makes the product sources pick up these
// definitions; nothing under cvxpygen_amd/ knows about the emulator.
#ifndef CPG_WAVE_PRIMITIVES_H
#define CPG_WAVE_PRIMITIVES_H

#include <stdint.h>
#include <atomic>
#include <cmath>
#include <cstring>

#define CPG_DEV inline
#define CPG_DEV_DATA
#define CPG_DEV_NOINLINE inline
#define CPG_LANES 64

extern __thread double cpg_lds[];     // the running workgroup's LDS window (fake_hip/hip/hip_runtime.h)

namespace cpgw {

#define CPG_LDS
inline double *lds_window() { return ::cpg_lds; }
inline double *lds_window3() { return ::cpg_lds; }
template <class T> inline T *as_global(T *p) { return p; }
struct SimBarrier { int count = 0, gen = 0, n = 0; };
struct SimWave {                 // shared by the 64 fibers of one emulated wavefront
    SimBarrier bar;
    double xch[64];
    int ixch[64];
};
struct SimDim3 { unsigned x, y, z; };
struct SimThread {               // one fiber = one GPU thread
    void *sp;                    // saved stack pointer while switched out
    int lane, wave, block, nblocks, waves_per_block;
    SimDim3 tidx, bidx, bdim, gdim;
    SimWave *wv;
    SimBarrier *block_bar;
    bool done;
    SimThread *next;             // ring of the workgroup's fibers
};
inline thread_local SimThread *cur;

#if defined(__x86_64__)
// minimal System-V context switch: callee-saved registers + stack pointer
extern "C" void cpg_sim_switch(void **save_sp, void *load_sp);
asm(".text\n.globl cpg_sim_switch\n.type cpg_sim_switch,@function\ncpg_sim_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size cpg_sim_switch,.-cpg_sim_switch\n");
#else
#error "the lock-step emulator's fiber switch is written for x86-64 hosts"
#endif

// hand the host thread to the next live fiber of the workgroup
inline void fiber_yield() {
    SimThread *me = cur, *nx = me->next;
    while (nx->done) nx = nx->next;
    if (nx == me) return;
    cur = nx;
    cpg_sim_switch(&me->sp, nx->sp);
}
inline void bar_wait(SimBarrier *b) {
    const int g = b->gen;
    if (++b->count == b->n) { b->count = 0; b->gen = g + 1; return; }
    while (*(volatile int *)&b->gen == g) fiber_yield();
}

inline void wave_sync() { bar_wait(&cur->wv->bar); }

inline int lane_id() { return cur->lane; }
inline int wave_in_block() { return cur->wave; }
inline unsigned thread_in_block() { return (unsigned)(cur->wave * 64 + cur->lane); }
inline unsigned block_threads() { return (unsigned)(cur->waves_per_block * 64); }
inline void block_sync() { bar_wait(cur->block_bar); }
inline void lds_order() { wave_sync(); }

template <int N>
inline double row_shl(double v) {
    SimWave *w = cur->wv;
    w->xch[cur->lane] = v;
    wave_sync();
    int src = cur->lane + N;
    double r = ((src >> 4) == (cur->lane >> 4)) ? w->xch[src] : 0.0;
    wave_sync();
    return r;
}
inline double lane_select(unsigned long long mask, double v) { return ((mask >> cur->lane) & 1ull) ? v : 0.0; }
inline double read_lane(double v, int lane) {
    SimWave *w = cur->wv;
    w->xch[cur->lane] = v;
    wave_sync();
    double r = w->xch[lane];
    wave_sync();
    return r;
}
inline double shfl_down(double v, int delta) {
    SimWave *w = cur->wv;
    w->xch[cur->lane] = v;
    wave_sync();
    int src = cur->lane + delta;
    double r = src < 64 ? w->xch[src] : v;
    wave_sync();
    return r;
}
inline double up16(double v) { return shfl_down(v, 16); }
inline double up32(double v) { return shfl_down(v, 32); }
inline int read_first_lane(int v) {
    SimWave *w = cur->wv;
    w->ixch[cur->lane] = v;
    wave_sync();
    int r = w->ixch[0];
    wave_sync();
    return r;
}
inline bool wave_any(bool p) {
    SimWave *w = cur->wv;
    w->ixch[cur->lane] = p ? 1 : 0;
    wave_sync();
    int r = 0;
    for (int i = 0; i < 64; i++) r |= w->ixch[i];
    wave_sync();
    return r != 0;
}
inline unsigned atomic_next(unsigned *ctr) {
    return __atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED);
}
inline void lds_max_u64(unsigned long long *p, double v) {       // (fibers switch at the primitives only: a plain update is atomic)
    unsigned long long b; std::memcpy(&b, &v, 8);
    if (b > *p) *p = b;
}
inline void lds_max_u64_l(unsigned long long *p, double v) { lds_max_u64(p, v); }
inline double u64_as_double(unsigned long long v) { double d; std::memcpy(&d, &v, 8); return d; }
inline unsigned long long clock100() { return 0ull; }
inline void mem_order() { wave_sync(); }
inline unsigned long long ballot(bool p) {
    SimWave *w = cur->wv;
    w->ixch[cur->lane] = p ? 1 : 0;
    wave_sync();
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) if (w->ixch[i]) m |= 1ULL << i;
    wave_sync();
    return m;
}
inline unsigned mbcnt(unsigned long long mask) {
    return (unsigned)__builtin_popcountll(mask & ((1ULL << cur->lane) - 1ULL));
}
inline unsigned popc64(unsigned long long m) { return (unsigned)__builtin_popcountll(m); }
inline void sched_fence() {}
inline int opaque(int v) { return v; }
template <class T> inline T gld_stream(const T *base, unsigned idx) { return base[idx]; }
template <class T> inline void gst_stream(T *base, unsigned idx, T v) { base[idx] = v; }
template <class T> inline T *pin_lds(T *p) { return p; }
inline double sgpr_value(double v) { return v; }
// element `field HI of the packed word` of an LDS array of doubles (cpg_wave_gfx950.h: one v_mad_u32_u16)
template <int HI> inline const double *lds_elem16(const double *base, unsigned w) { return base + ((w >> (16 * HI)) & 0xFFFFu); }
inline void lds_signal(unsigned *p, unsigned v) { *(volatile unsigned *)p = v; }
inline void lds_spin_until_ge(unsigned *p, unsigned v) { while (*(volatile unsigned *)p < v) fiber_yield(); }
inline void assume(bool) {}
inline unsigned sld(const unsigned *base, unsigned idx) { return base[idx]; }

}  // namespace cpgw
#endif  // CPG_WAVE_PRIMITIVES_H
