// covar_fold.h -- the masked outer-product fold shared by the two pass-1 kernels
// (pass1.hip: fp32 butterflies; pass1_mc.hip: matrix-core transforms).
// Replaces compute_covar (funcwj/setk libs/beamformer.py:87-103).
#pragma once
#include "common.h"
#include "fft512.h"

namespace setk {

// ---- Hermitian pairs split over the two covariance halves -------------------
// Half H owns the diagonals (i, i) with i % 2 == H (real: one accumulator per
// mask) and every other off-diagonal pair of the (i < j) enumeration.  Keeping
// the diagonals apart saves their imaginary accumulators (8 VGPRs at C = 8).
// Planes in the partial slab stay in the global (i <= j) order: pair_index()
// of common.h.
template <int C>
struct PairSplit {
    static constexpr int ND = (C + 1) / 2;                 // diagonals per half (max)
    static constexpr int NO = (C * (C - 1) / 2 + 1) / 2;   // off-diagonal pairs per half (max)
};

// x_i conj(x_j) with the SAME expression for its real part as the diagonal's |x_i|^2 below:
// for equal operands (a duplicated channel) the off-diagonal sum is then bit for bit the
// diagonal one, and SETK_FLAG_STRICT_REFERENCE's elimination (solve.hip, lu_refusal_kernel)
// cancels exactly where the reference's does.  Same two instructions per part as the
// contracted a.x b.x + a.y b.y.
SETK_DEV cf cmulc_cov(cf a, cf b) {
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
}

template <int C, int H>
SETK_DEV void accumulate_half(const cf (&x)[C], float ws, float wn, float* ds, float* dn, cf* os,
                              cf* on) {
#pragma unroll
    for (int i = H; i < C; i += 2) {
        const float p = fmaf(x[i].x, x[i].x, x[i].y * x[i].y);
        ds[i / 2] = fmaf(ws, p, ds[i / 2]);
        dn[i / 2] = fmaf(wn, p, dn[i / 2]);
    }
    int k = 0;
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
        for (int j = i + 1; j < C; ++j) {
            if ((k & 1) == H) {
                const cf p = cmulc_cov(x[i], x[j]);
                os[k / 2].x = fmaf(ws, p.x, os[k / 2].x);
                on[k / 2].x = fmaf(wn, p.x, on[k / 2].x);
                os[k / 2].y = fmaf(ws, p.y, os[k / 2].y);
                on[k / 2].y = fmaf(wn, p.y, on[k / 2].y);
            }
            ++k;
        }
}

template <int C, int H>
SETK_DEV void store_half(float* P, int f, const float* ds, const float* dn, const cf* os,
                         const cf* on) {
    constexpr int NP = npairs(C);
    constexpr int FP = kBinsPad;
#pragma unroll
    for (int i = H; i < C; i += 2) {
        const int e = pair_index(i, i, C);
        P[(0 * NP + e) * FP + f] = ds[i / 2];
        P[(1 * NP + e) * FP + f] = 0.f;
        P[(2 * NP + e) * FP + f] = dn[i / 2];
        P[(3 * NP + e) * FP + f] = 0.f;
    }
    int k = 0;
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
        for (int j = i + 1; j < C; ++j) {
            if ((k & 1) == H) {
                const int e = pair_index(i, j, C);
                P[(0 * NP + e) * FP + f] = os[k / 2].x;
                P[(1 * NP + e) * FP + f] = os[k / 2].y;
                P[(2 * NP + e) * FP + f] = on[k / 2].x;
                P[(3 * NP + e) * FP + f] = on[k / 2].y;
            }
            ++k;
        }
}

// max(m, |a|, |b|) in one VALU instruction
SETK_DEV float max3_abs(float m, float a, float b) {
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}

SETK_DEV void wg_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace setk
