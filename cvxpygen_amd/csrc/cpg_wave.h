// Wavefront-level primitives for gfx950 (CDNA4): 64-lane cross-lane moves via DPP / readlane,
// wave-uniform reductions, and LDS ordering inside one wavefront.
//
// The kernels in this directory are written against these few functions only.  Besides the real
// gfx950 implementation there is a second, host-side implementation (CPG_HOST_SIM) that runs the 64
// lanes of a wavefront as lock-stepped threads; it exists solely so that tests/ can execute the
// very same kernel source on a machine without a GPU (tests/sim/).  The product library
// (libcpg_hip.so) is never built with CPG_HOST_SIM.
#pragma once

#include <stdint.h>

#ifndef CPG_HOST_SIM
// =================================================================================== gfx950
#include <hip/hip_runtime.h>

#define CPG_DEV __device__ __forceinline__
#define CPG_LANES 64

namespace cpgw {

CPG_DEV int lane_id() { return (int)(threadIdx.x & 63); }
CPG_DEV int wave_in_block() { return (int)(threadIdx.x >> 6); }
CPG_DEV unsigned thread_in_block() { return threadIdx.x; }
CPG_DEV unsigned block_threads() { return blockDim.x; }
CPG_DEV void block_sync() { __syncthreads(); }

// Orders the LDS traffic of ONE wavefront: DS operations of a wave are executed in program order
// by the hardware; this only stops the compiler from moving loads above earlier stores.
CPG_DEV void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int CTRL>
CPG_DEV double dpp_move_zero(double v) {   // invalid source lanes deliver 0 (bound_ctrl)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// lane i receives the value of lane i + N of the same 16-lane row, 0 when i + N leaves the row
template <int N>
CPG_DEV double row_shl(double v) { return dpp_move_zero<0x100 + N>(v); }

CPG_DEV double read_lane(double v, int lane) {   // `lane` must be wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
CPG_DEV double shfl_down(double v, int delta) { return __shfl_down(v, delta, 64); }
// Value of lane + 16 (lane + 32) on the lanes of the even 16-lane rows (of the lower half); other
// lanes receive values that must not be used.  gfx950 row / half swaps: plain VALU, no LDS crossbar
// and no index register, unlike the ds_bpermute behind __shfl_down.
CPG_DEV double up16(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}
CPG_DEV double up32(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}
CPG_DEV int read_first_lane(int v) { return __builtin_amdgcn_readfirstlane(v); }
CPG_DEV bool wave_any(bool p) { return __any(p) != 0; }
// orders GLOBAL stores and loads of one wavefront among its own lanes (per-wavefront buffers:
// written by some lanes, read by others later); the CU's L1 is coherent for its own traffic
CPG_DEV void mem_order() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
CPG_DEV unsigned long long ballot(bool p) { return __ballot(p); }
// number of set bits of `mask` below this lane
CPG_DEV unsigned mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
CPG_DEV unsigned popc64(unsigned long long m) { return (unsigned)__popcll(m); }

CPG_DEV unsigned atomic_next(unsigned *ctr) { return atomicAdd(ctr, 1u); }
// keeps the instruction scheduler from interleaving unrolled loop bodies (register pressure)
CPG_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// value the optimiser must treat as unknown: stops loop-invariant code motion from hoisting (and
// keeping alive) everything derived from it
CPG_DEV int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// tells the optimiser a fact it lost (e.g. the lane range after opaque()): bounds checks fold again
CPG_DEV void assume(bool c) { __builtin_assume(c); }
// Word of a read-only table at a wave-uniform index through the SCALAR cache (s_load): the constant
// address space tells the compiler that no store of the kernel can alias it, which it cannot prove
// for a plain global pointer in kernels that also write global memory.
CPG_DEV unsigned sld(const unsigned *base, unsigned idx) {
    typedef const unsigned __attribute__((address_space(4))) *cptr_t;
    return ((cptr_t)(unsigned long long)base)[idx];
}

}  // namespace cpgw

#else
// =================================================================================== host emulation
#include <atomic>
#include <cmath>
#include <cstring>
#include <pthread.h>

#define CPG_DEV inline
#define CPG_LANES 64
#define __global__
#define __restrict__

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
extern thread_local SimThread tls;

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
#endif

namespace cpgw {

// Global loads / stores with a wave-uniform base and a 32-bit BYTE offset per lane: lets the
// compiler use the `global_load v, v_off, s[base:base+1]` form (one VGPR per access) instead of
// materialising -- and keeping alive -- a 64-bit address pair per array.
template <typename T>
CPG_DEV T gld(const T *base, unsigned idx) {
    return *(const T *)((const char *)base + (size_t)(idx * (unsigned)sizeof(T)));
}
template <typename T>
CPG_DEV void gst(T *base, unsigned idx, T v) {
    *(T *)((char *)base + (size_t)(idx * (unsigned)sizeof(T))) = v;
}

// ---- reductions built on the primitives (identical code on both back ends) ----------------------

// Sum over groups of G = 2^LG consecutive lanes; the FIRST lane of every group holds the group sum
// afterwards (other lanes hold partial sums that must not be used).
template <int LG>
CPG_DEV double group_sum_first(double v) {
    if (LG >= 1) v += row_shl<1>(v);
    if (LG >= 2) v += row_shl<2>(v);
    if (LG >= 3) v += row_shl<4>(v);
    if (LG >= 4) v += row_shl<8>(v);
    if (LG >= 5) v += up16(v);
    if (LG >= 6) v += up32(v);
    return v;
}
CPG_DEV double group_sum_first_dyn(double v, int lg) {   // lg wave-uniform
    switch (lg) {
        case 0: return v;
        case 1: return group_sum_first<1>(v);
        case 2: return group_sum_first<2>(v);
        case 3: return group_sum_first<3>(v);
        case 4: return group_sum_first<4>(v);
        case 5: return group_sum_first<5>(v);
        default: return group_sum_first<6>(v);
    }
}
// Segmented sum for rows that occupy a variable number (<= 8) of ADJACENT lanes inside one 16-lane
// DPP row: in stage j lane t adds lane t + 2^j iff bit j of its mask is set (the source lane belongs
// to the same row); after S stages the first lane of every row holds the row sum.
template <int S>
CPG_DEV double seg_sum_first(double v, unsigned mask) {
    if (S >= 1) { const double t = row_shl<1>(v); v += (mask & 1u) ? t : 0.0; }
    if (S >= 2) { const double t = row_shl<2>(v); v += (mask & 2u) ? t : 0.0; }
    if (S >= 3) { const double t = row_shl<4>(v); v += (mask & 4u) ? t : 0.0; }
    return v;
}
CPG_DEV double seg_sum_first_dyn(double v, unsigned mask, int stages) {   // stages wave-uniform
    switch (stages) {
        case 0: return v;
        case 1: return seg_sum_first<1>(v, mask);
        case 2: return seg_sum_first<2>(v, mask);
        default: return seg_sum_first<3>(v, mask);
    }
}
CPG_DEV double dmax2(double a, double b) { return a > b ? a : b; }
CPG_DEV double dmin2(double a, double b) { return a < b ? a : b; }

// Wave-wide sum / max delivered to every lane (wave-uniform result).
CPG_DEV double wave_sum(double v) {
    v += row_shl<1>(v); v += row_shl<2>(v); v += row_shl<4>(v); v += row_shl<8>(v);
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
CPG_DEV double wave_max_nonneg(double v) {   // v >= 0 on every lane (0 is the neutral element)
    v = dmax2(v, row_shl<1>(v)); v = dmax2(v, row_shl<2>(v));
    v = dmax2(v, row_shl<4>(v)); v = dmax2(v, row_shl<8>(v));
    return dmax2(dmax2(read_lane(v, 0), read_lane(v, 16)), dmax2(read_lane(v, 32), read_lane(v, 48)));
}

// general wave-wide min / max (any sign), wave-uniform result
CPG_DEV double wave_min(double v) {
    v = dmin2(v, shfl_down(v, 1)); v = dmin2(v, shfl_down(v, 2)); v = dmin2(v, shfl_down(v, 4));
    v = dmin2(v, shfl_down(v, 8)); v = dmin2(v, shfl_down(v, 16)); v = dmin2(v, shfl_down(v, 32));
    return read_lane(v, 0);
}
CPG_DEV double wave_max(double v) {
    v = dmax2(v, shfl_down(v, 1)); v = dmax2(v, shfl_down(v, 2)); v = dmax2(v, shfl_down(v, 4));
    v = dmax2(v, shfl_down(v, 8)); v = dmax2(v, shfl_down(v, 16)); v = dmax2(v, shfl_down(v, 32));
    return read_lane(v, 0);
}

}  // namespace cpgw
