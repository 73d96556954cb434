// Wavefront-level primitives for gfx950 (CDNA4): 64-lane cross-lane moves via DPP / readlane /
// permlane swaps, wave-uniform broadcasts, and LDS / memory ordering inside one wavefront.
// The kernels in this directory are written against these few functions (and the reductions that
// cpg_wave.h builds from them) only.
#ifndef CPG_WAVE_PRIMITIVES_H
#define CPG_WAVE_PRIMITIVES_H

#include <stdint.h>
#include <hip/hip_runtime.h>

#define CPG_DEV __device__ __forceinline__
#define CPG_DEV_DATA __device__        // a read-only table in global memory (generated headers)
#define CPG_DEV_NOINLINE __device__ __attribute__((noinline))      // a real call: a register allocation of its own
#define CPG_LANES 64

namespace cpgw {

// the workgroup's dynamic LDS window, for functions that are real calls (CPG_DEV_NOINLINE): a pointer handed through a
// call is a generic pointer (flat_load / flat_store), an offset into this array keeps the accesses LDS instructions
// (pointers of functions that are real calls carry their address space in the TYPE: the compiler does not infer it
// through the dynamic-LDS table a non-kernel function reads its window from, and a generic pointer costs a 64-bit address
// register per access and flat_load instead of ds_read)
#define CPG_LDS __attribute__((address_space(3)))
CPG_DEV double *lds_window() {
    extern __shared__ __attribute__((aligned(16))) double cpg_lds[];
    return cpg_lds;
}
// A pointer that is known to point to global memory, said so in a way the optimiser keeps: the address-space inference
// rewrites the accesses through a generic pointer that was cast from a global one to global_load / global_store.  Pointers that reach a called function
// inside a struct are generic to the compiler: their accesses become flat_load, which counts on the LDS counter too --
// every wait for an LDS read then also waits for the table prefetches in flight.
template <class T>
CPG_DEV T *as_global(T *p) {      // (through an integer: a flat address of global memory IS its global address)
    return (T *)(__attribute__((address_space(1))) T *)(unsigned long long)p;
}
CPG_DEV CPG_LDS double *lds_window3() { return (CPG_LDS double *)lds_window(); }
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

// v on the lanes whose bit of the (wave-uniform, here: literal) 64-bit mask is set, 0.0 on the others: v_cndmask with the mask in a
// scalar register pair
CPG_DEV double lane_select(unsigned long long mask, double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(lo) : "v"(lo), "s"(mask));
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(hi) : "v"(hi), "s"(mask));
    return __hiloint2double(hi, lo);
}
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
CPG_DEV unsigned long long clock100() { return __builtin_amdgcn_s_memrealtime(); }      // 100 MHz (timing experiments)
// max of non-negative doubles in LDS (they order like their bit patterns): ds_max_u64, no return value
CPG_DEV void lds_max_u64(unsigned long long *p, double v) { atomicMax(p, (unsigned long long)__double_as_longlong(v)); }
// ... the same through an LDS pointer by type (ds_max_u64 instead of a flat atomic)
CPG_DEV void lds_max_u64_l(CPG_LDS unsigned long long *p, double v) {
    __hip_atomic_fetch_max(p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
CPG_DEV double u64_as_double(unsigned long long v) { return __longlong_as_double((long long)v); }
// keeps the instruction scheduler from interleaving unrolled loop bodies (register pressure)
CPG_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// value the optimiser must treat as unknown: stops loop-invariant code motion from hoisting (and
// keeping alive) everything derived from it
CPG_DEV int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// Global loads / stores of per-instance STREAMS (written once, read once or a few times, by one workgroup: coefficient images,
// program-order copies) with a 32-bit byte offset per lane, marked non-temporal: they pass the L2 without pushing out the tables every
// workgroup of the XCD shares.
template <typename T>
CPG_DEV T gld_stream(const T *base, unsigned idx) {
    return __builtin_nontemporal_load((const T *)((const char *)base + (size_t)(idx * (unsigned)sizeof(T))));
}
template <typename T>
CPG_DEV void gst_stream(T *base, unsigned idx, T v) {
    __builtin_nontemporal_store(v, (T *)((char *)base + (size_t)(idx * (unsigned)sizeof(T))));
}
// An LDS pointer as an opaque value held in a register: inside a function that is a real call the address of
// the workgroup's dynamic LDS window comes from a table in memory, and under register pressure the compiler re-reads it (s_load)
// wherever it is used instead of keeping it -- a scalar load shares its counter with the LDS reads, so every such reload is drained
// in front of the next ds_read.  Laundered through a register constraint it cannot be rematerialised.
template <class T>
CPG_DEV CPG_LDS T *pin_lds(CPG_LDS T *p) {
    unsigned a = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)p;
    asm volatile("" : "+v"(a));          // (a vector register: a scalar one was refused inside loops with barriers, "illegal VGPR to SGPR copy")
    return (CPG_LDS T *)(__attribute__((address_space(3))) void *)(unsigned long long)a;
}
// A wave-uniform double pinned in a scalar register pair as a VALUE the optimiser cannot trace back to the memory it was loaded from:
// a select between two members of a struct that arrived by reference is otherwise rewritten into a select between their ADDRESSES and
// a load -- of a struct that then has to live in scratch, one memory round trip per use (the resident kernel's ADMM loop: 22 per iteration).
CPG_DEV double sgpr_value(double v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(v)), hi = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(v));
    unsigned long long u = ((unsigned long long)hi << 32) | lo;
    asm("" : "+s"(u));
    return __longlong_as_double((long long)u);
}
// Address of element number (16-bit field HI of the packed word w) of an LDS array of doubles: ONE vector instruction
// (v_mad_u32_u16 with op_sel picking the half: field * 8 + base) where `base[w & 0xFFFF]` / `base[w >> 16]` compile to an extraction
// and a shift-add.  The table walks of the factorisations run on one wavefront per SIMD, i.e. at one instruction per ~5 cycles whatever
// its kind: two of a step's thirteen instructions per operand were address arithmetic (profiles/r6_t2_team_factor_knockouts.txt).
template <int HI>
CPG_DEV const CPG_LDS double *lds_elem16(const CPG_LDS double *base, unsigned w) {
    const unsigned b = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)base;
    unsigned a;
    if (HI) asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[1,0,0,0]" : "=v"(a) : "v"(w), "v"(b));
    else asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[0,0,0,0]" : "=v"(a) : "v"(w), "v"(b));
    return (const CPG_LDS double *)(__attribute__((address_space(3))) const void *)(unsigned long long)a;
}
// A flag in LDS between two wavefronts of one workgroup that do NOT meet at a barrier: the producer has finished a piece of work
// (its LDS stores are complete: lds_order() in front), the consumer polls.  Release / acquire at workgroup scope.
CPG_DEV void lds_signal(CPG_LDS unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
CPG_DEV void lds_spin_until_ge(CPG_LDS unsigned *p, unsigned v) {
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
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

#endif  // CPG_WAVE_PRIMITIVES_H
