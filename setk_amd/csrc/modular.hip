// modular.hip -- stand-alone operators behind the python API mirror
// (compute_covar on a stored spectrogram, generic power-of-two STFT/iSTFT for
// n_fft != 512).  The fused hot path does not use these.
#include "common.h"
#include "fft512.h"

namespace setk {

// ---------------------------------------------------------------------------
// compute_covar (libs/beamformer.py:87-103) on spec[C][T][F] (f fastest).
// Thread = bin, blockIdx.y = slice of the frame axis; partial planes
// [split][2*NP + 1][pitch] = (re | im | sum m).
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void covar_spec_kernel(const cf* __restrict__ spec,
                                                         const float* __restrict__ mask, int T,
                                                         int F, int pitch, int t_per_split,
                                                         float* __restrict__ partials) {
    constexpr int NP = npairs(C);
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int split = blockIdx.y;
    if (f >= F) return;
    const int t0 = split * t_per_split;
    const int t1 = min(T, t0 + t_per_split);
    cf acc[NP];
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < NP; ++e) acc[e] = make_float2(0.f, 0.f);
    for (int t = t0; t < t1; ++t) {
        const float m = mask[(size_t)t * F + f];
        cf x[C];
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = spec[((size_t)c * T + t) * F + f];
        sum += m;
        int e = 0;
#pragma unroll
        for (int i = 0; i < C; ++i)
#pragma unroll
            for (int j = i; j < C; ++j) {
                const cf p = cmulc(x[i], x[j]);
                acc[e].x = fmaf(m, p.x, acc[e].x);
                if (i != j) acc[e].y = fmaf(m, p.y, acc[e].y);
                ++e;
            }
    }
    float* P = partials + (size_t)split * (2 * NP + 1) * pitch;
#pragma unroll
    for (int e = 0; e < NP; ++e) {
        P[(size_t)e * pitch + f] = acc[e].x;
        P[(size_t)(NP + e) * pitch + f] = acc[e].y;
    }
    P[(size_t)(2 * NP) * pitch + f] = sum;
}

hipError_t launch_covar_spec(int C, const float* spec, const float* mask, int T, int F,
                             float* partials, int t_split, hipStream_t s) {
    const int pitch = ((F + 7) / 8) * 8;
    const int per = (T + t_split - 1) / t_split;
    dim3 grid((F + 255) / 256, t_split);
#define SETK_CASE(c)                                                                         \
    case c:                                                                                  \
        hipLaunchKernelGGL(covar_spec_kernel<c>, grid, dim3(256), 0, s,                      \
                           reinterpret_cast<const cf*>(spec), mask, T, F, pitch, per, partials); \
        break;
    switch (C) {
        SETK_CASE(1)
        SETK_CASE(2)
        SETK_CASE(3)
        SETK_CASE(4)
        SETK_CASE(5)
        SETK_CASE(6)
        SETK_CASE(7)
        SETK_CASE(8)
        default:
            return hipErrorInvalidValue;
    }
#undef SETK_CASE
    return hipGetLastError();
}

__global__ void covar_spec_finalize_kernel(const float* __restrict__ partials, int nparts, int F,
                                           int C, int pitch, cf* __restrict__ out) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const int NP = npairs(C);
    const size_t slab = (size_t)(2 * NP + 1) * pitch;
    float den = 0.f;
    for (int p = 0; p < nparts; ++p) den += partials[p * slab + (size_t)(2 * NP) * pitch + f];
    den = fmaxf(den, 1e-6f);
    for (int i = 0; i < C; ++i)
        for (int j = i; j < C; ++j) {
            const int e = pair_index(i, j, C);
            float re = 0.f, im = 0.f;
            for (int p = 0; p < nparts; ++p) {
                re += partials[p * slab + (size_t)e * pitch + f];
                im += partials[p * slab + (size_t)(NP + e) * pitch + f];
            }
            re /= den;
            im /= den;
            out[((size_t)f * C + i) * C + j] = make_float2(re, im);
            if (i != j) out[((size_t)f * C + j) * C + i] = make_float2(re, -im);
        }
}

hipError_t launch_covar_spec_finalize(int C, const float* partials, int nparts, int F,
                                      float* covar_fcc, hipStream_t s) {
    const int pitch = ((F + 7) / 8) * 8;
    hipLaunchKernelGGL(covar_spec_finalize_kernel, dim3((F + 255) / 256), dim3(256), 0, s,
                       partials, nparts, F, C, pitch, reinterpret_cast<cf*>(covar_fcc));
    return hipGetLastError();
}

}  // namespace setk
