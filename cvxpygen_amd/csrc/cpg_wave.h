// Wavefront-level building blocks of the kernels: the gfx950 primitives (cpg_wave_gfx950.h) and the
// reductions / addressing helpers built on them.
#pragma once

#include <stdint.h>

#include "cpg_wave_gfx950.h"

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

// ---- reductions built on the primitives ------------------------------------------------

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
// The same without a branch per depth (lg wave-uniform, 0 .. 6): all six stages, a stage beyond lg adds zero.  For cold
// code whose size matters more than six moves (the resident kernel's factorisation stream: the unrolled steps of its
// prefetch ring each carry a copy, and one wavefront per SIMD executes straight-line code at the speed of its
// instruction fetch).
CPG_DEV double group_sum_first_flat(double v, int lg) {
    { const double t = row_shl<1>(v); v += lg >= 1 ? t : 0.0; }
    { const double t = row_shl<2>(v); v += lg >= 2 ? t : 0.0; }
    { const double t = row_shl<4>(v); v += lg >= 3 ? t : 0.0; }
    { const double t = row_shl<8>(v); v += lg >= 4 ? t : 0.0; }
    { const double t = up16(v); v += lg >= 5 ? t : 0.0; }
    { const double t = up32(v); v += lg >= 6 ? t : 0.0; }
    return v;
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
// The same with the stage masks as 64-bit LANE masks known at code-generation time (lane t adds lane t + 2^j iff bit t of Mj):
// two scalar moves and a select per stage -- no per-lane mask register, no compare
template <int S>
CPG_DEV double seg_sum_first_lit(double v, unsigned long long m0, unsigned long long m1, unsigned long long m2) {
    if (S >= 1) v += lane_select(m0, row_shl<1>(v));
    if (S >= 2) v += lane_select(m1, row_shl<2>(v));
    if (S >= 3) v += lane_select(m2, row_shl<4>(v));
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
