// 16-lane row reductions for device code: a wavefront is treated as four rows of 16 lanes and
// every row reduces on its own with DPP rotates (no LDS traffic).
#pragma once

#include <hip/hip_runtime.h>

namespace ssmap {

// numpy argmax order: NaN beats everything, ties go to the lower index.  Written without
// short-circuit operators so that it compiles to compares and selects, not branches.
__device__ inline bool arg_better(float a, int ia, float b, int ib) {
    const bool an = a != a, bn = b != b;
    const bool lower = ia < ib;
    const bool num = (a > b) | ((a == b) & lower);
    return an ? (!bn | lower) : (!bn & num);
}
// After the four rotate steps every lane of a row holds the reduction over the row.
template <int N>
__device__ inline float row_ror(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xF, 0xF, false));
}
template <int N>
__device__ inline int row_ror(int x) {
    return __builtin_amdgcn_update_dpp(0, x, 0x120 + N, 0xF, 0xF, false);
}
__device__ inline float row_sum(float x) {
    x += row_ror<8>(x); x += row_ror<4>(x); x += row_ror<2>(x); x += row_ror<1>(x);
    return x;
}
template <int N>
__device__ inline void row_argmax_step(float &val, int &idx) {
    const float ov = row_ror<N>(val);
    const int oi = row_ror<N>(idx);
    const bool take = arg_better(ov, oi, val, idx);
    val = take ? ov : val;
    idx = take ? oi : idx;
}
__device__ inline void row_argmax(float &val, int &idx) {
    row_argmax_step<8>(val, idx); row_argmax_step<4>(val, idx);
    row_argmax_step<2>(val, idx); row_argmax_step<1>(val, idx);
}

__device__ inline float row_max(float x) {
    x = fmaxf(x, row_ror<8>(x)); x = fmaxf(x, row_ror<4>(x)); x = fmaxf(x, row_ror<2>(x)); x = fmaxf(x, row_ror<1>(x));
    return x;
}

}  // namespace ssmap
