// dpp.h -- lane exchanges of the 8- / 16-lane problem groups (solve.hip, cgmm.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace setk {

#ifndef SD
#define SD __device__ __forceinline__
#endif

// Exchange with lane j ^ M of an 8- or 16-lane group as DPP moves (VALU, a few
// cycles) instead of ds_bpermute (LDS pipe, ~100 cycles): the perfect matchings
// j <-> j ^ M, M = 1..7 (..15), visit every column pair once per sweep.
//   M = 1, 2, 3: quad_perm;  M = 7: row_half_mirror (j -> 7 - j = j ^ 7);
//   M = 4, 5, 6: half_mirror followed by the quad_perm of M ^ 7;
//   M = 15: row_mirror;  M = 12, 13, 14: mirror + quad_perm of M ^ 15;
//   M = 8: row_ror:8;  M = 9, 10, 11: ror:8 + quad_perm of M ^ 8.
template <int M>
SD int dpp_xor(int v) {
    constexpr int qp[4] = {0, 0xB1, 0x4E, 0x1B};  // quad_perm of j ^ 1, j ^ 2, j ^ 3
    if constexpr (M < 4) {
        return __builtin_amdgcn_update_dpp(0, v, qp[M], 0xf, 0xf, true);
    } else if constexpr (M == 7) {
        return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
    } else if constexpr (M < 8) {
        const int h = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
        return __builtin_amdgcn_update_dpp(0, h, qp[M ^ 7], 0xf, 0xf, true);
    } else if constexpr (M == 8) {
        return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);
    } else if constexpr (M < 12) {
        const int h = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);
        return __builtin_amdgcn_update_dpp(0, h, qp[M ^ 8], 0xf, 0xf, true);
    } else if constexpr (M == 15) {
        return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    } else {
        const int h = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
        return __builtin_amdgcn_update_dpp(0, h, qp[M ^ 15], 0xf, 0xf, true);
    }
}
template <int M>
SD double dshfl_xor(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = dpp_xor<M>((int)(b & 0xffffffffll));
    const int hi = dpp_xor<M>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

}  // namespace setk
