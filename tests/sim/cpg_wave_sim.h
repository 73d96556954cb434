// TEST INFRASTRUCTURE ONLY -- host emulation of the wavefront primitives of
// cvxpygen_amd/csrc/cpg_wave_gfx950.h: the 64 lanes of a wavefront run as lock-stepped host threads
// that meet at a barrier in every cross-lane primitive.  tests/sim/build_sim.py force-includes this
// header in front of the product's cpg_hip.cpp (g++ -include) together with the stand-in
// <hip/hip_runtime.h> of tests/sim/fake_hip, so the CPU-only test tier executes the product's kernel
// SOURCES through the real C-ABI.  Defining the include guard of the gfx950 header makes the product
// sources pick up these definitions; nothing under cvxpygen_amd/ knows about the emulator.
#ifndef CPG_WAVE_PRIMITIVES_H
#define CPG_WAVE_PRIMITIVES_H

#include <stdint.h>
#include <atomic>
#include <cmath>
#include <cstring>
#include <pthread.h>

#define CPG_DEV inline
#define CPG_LANES 64

namespace cpgw {

struct SimWave {                 // shared by the 64 threads of one emulated wavefront
    pthread_barrier_t bar;
    double xch[64];
    int ixch[64];
};
struct SimThread {
    int lane, wave, block, nblocks, waves_per_block;
    SimWave *wv;
    pthread_barrier_t *block_bar;
    char *lds;                   // block-wide dynamic LDS
};
inline thread_local SimThread tls;

inline void wave_sync() { pthread_barrier_wait(&tls.wv->bar); }

inline int lane_id() { return tls.lane; }
inline int wave_in_block() { return tls.wave; }
inline unsigned thread_in_block() { return (unsigned)(tls.wave * 64 + tls.lane); }
inline unsigned block_threads() { return (unsigned)(tls.waves_per_block * 64); }
inline void block_sync() { pthread_barrier_wait(tls.block_bar); }
inline void lds_order() { wave_sync(); }

template <int N>
inline double row_shl(double v) {
    SimWave *w = tls.wv;
    w->xch[tls.lane] = v;
    wave_sync();
    int src = tls.lane + N;
    double r = ((src >> 4) == (tls.lane >> 4)) ? w->xch[src] : 0.0;
    wave_sync();
    return r;
}
inline double read_lane(double v, int lane) {
    SimWave *w = tls.wv;
    w->xch[tls.lane] = v;
    wave_sync();
    double r = w->xch[lane];
    wave_sync();
    return r;
}
inline double shfl_down(double v, int delta) {
    SimWave *w = tls.wv;
    w->xch[tls.lane] = v;
    wave_sync();
    int src = tls.lane + delta;
    double r = src < 64 ? w->xch[src] : v;
    wave_sync();
    return r;
}
inline double up16(double v) { return shfl_down(v, 16); }
inline double up32(double v) { return shfl_down(v, 32); }
inline int read_first_lane(int v) {
    SimWave *w = tls.wv;
    w->ixch[tls.lane] = v;
    wave_sync();
    int r = w->ixch[0];
    wave_sync();
    return r;
}
inline bool wave_any(bool p) {
    SimWave *w = tls.wv;
    w->ixch[tls.lane] = p ? 1 : 0;
    wave_sync();
    int r = 0;
    for (int i = 0; i < 64; i++) r |= w->ixch[i];
    wave_sync();
    return r != 0;
}
inline unsigned atomic_next(unsigned *ctr) {
    return __atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED);
}
inline void mem_order() { wave_sync(); }
inline unsigned long long ballot(bool p) {
    SimWave *w = tls.wv;
    w->ixch[tls.lane] = p ? 1 : 0;
    wave_sync();
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) if (w->ixch[i]) m |= 1ULL << i;
    wave_sync();
    return m;
}
inline unsigned mbcnt(unsigned long long mask) {
    return (unsigned)__builtin_popcountll(mask & ((1ULL << tls.lane) - 1ULL));
}
inline unsigned popc64(unsigned long long m) { return (unsigned)__builtin_popcountll(m); }
inline void sched_fence() {}
inline int opaque(int v) { return v; }
inline void assume(bool) {}
inline unsigned sld(const unsigned *base, unsigned idx) { return base[idx]; }

}  // namespace cpgw
#endif  // CPG_WAVE_PRIMITIVES_H
