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

// 8 < C <= 16: one workgroup row (blockIdx.z = i) accumulates the pairs (i, j >= i)
// of the upper triangle; the channels are re-read per row (the fused kernels stop
// at C = 8, this keeps the operator whole for larger arrays).
__global__ __launch_bounds__(256) void covar_spec_wide_kernel(const cf* __restrict__ spec,
                                                              const float* __restrict__ mask, int C,
                                                              int T, int F, int pitch,
                                                              int t_per_split,
                                                              float* __restrict__ partials) {
    const int NP = npairs(C);
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int split = blockIdx.y, i = blockIdx.z;
    if (f >= F) return;
    const int t0 = split * t_per_split;
    const int t1 = min(T, t0 + t_per_split);
    cf acc[kMaxChannels16];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxChannels16; ++j) acc[j] = make_float2(0.f, 0.f);
    for (int t = t0; t < t1; ++t) {
        const float m = mask[(size_t)t * F + f];
        const cf xi = spec[((size_t)i * T + t) * F + f];
        const cf mxi = make_float2(m * xi.x, m * xi.y);
        sum += m;
#pragma unroll
        for (int j = 0; j < kMaxChannels16; ++j) {
            if (j >= i && j < C) {
                const cf xj = spec[((size_t)j * T + t) * F + f];
                const cf p = cmulc(mxi, xj);
                acc[j].x += p.x;
                acc[j].y += p.y;
            }
        }
    }
    float* P = partials + (size_t)split * (2 * NP + 1) * pitch;
#pragma unroll
    for (int j = 0; j < kMaxChannels16; ++j) {
        if (j >= i && j < C) {
            const int e = pair_index(i, j, C);
            P[(size_t)e * pitch + f] = acc[j].x;
            P[(size_t)(NP + e) * pitch + f] = (i == j) ? 0.f : acc[j].y;
        }
    }
    if (i == 0) P[(size_t)(2 * NP) * pitch + f] = sum;
}

hipError_t launch_covar_spec(int C, const float* spec, const float* mask, int T, int F,
                             float* partials, int t_split, hipStream_t s) {
    const int pitch = ((F + 7) / 8) * 8;
    const int per = (T + t_split - 1) / t_split;
    if (C > kMaxChannels && C <= kMaxChannels16) {
        hipLaunchKernelGGL(covar_spec_wide_kernel, dim3((F + 255) / 256, t_split, C), dim3(256), 0,
                           s, reinterpret_cast<const cf*>(spec), mask, C, T, F, pitch, per,
                           partials);
        return hipGetLastError();
    }
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

// thread = (bin, entry e of the upper triangle): the sums over the slabs run in slab order (the
// result does not depend on the grid), the entries of a bin in parallel -- one thread per bin
// walking all 2 NP + 1 planes was 420 dependent loads and two workgroups (62 us at 4 channels
// and 20 slabs, four times the accumulation it finishes)
__global__ __launch_bounds__(256) void covar_spec_finalize_kernel(const float* __restrict__ partials,
                                                                  int nparts, int F, int C, int pitch,
                                                                  cf* __restrict__ out) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const int NP = npairs(C);
    const int e = blockIdx.y;
    // (i, j) of entry e in the row-major upper triangle
    int i = 0, rem = e;
    while (rem >= C - i) {
        rem -= C - i;
        ++i;
    }
    const int j = i + rem;
    const size_t slab = (size_t)(2 * NP + 1) * pitch;
    const float* pd = partials + (size_t)(2 * NP) * pitch + f;
    const float* pr = partials + (size_t)e * pitch + f;
    const float* pi = partials + (size_t)(NP + e) * pitch + f;
    float den = 0.f, re = 0.f, im = 0.f;
    for (int p = 0; p < nparts; ++p) {
        den += pd[p * slab];
        re += pr[p * slab];
        im += pi[p * slab];
    }
    den = fmaxf(den, 1e-6f);
    re /= den;
    im /= den;
    out[((size_t)f * C + i) * C + j] = make_float2(re, im);
    if (i != j) out[((size_t)f * C + j) * C + i] = make_float2(re, -im);
}

hipError_t launch_covar_spec_finalize(int C, const float* partials, int nparts, int F,
                                      float* covar_fcc, hipStream_t s) {
    const int pitch = ((F + 7) / 8) * 8;
    hipLaunchKernelGGL(covar_spec_finalize_kernel, dim3((F + 255) / 256, npairs(C)), dim3(256), 0, s,
                       partials, nparts, F, C, pitch, reinterpret_cast<cf*>(covar_fcc));
    return hipGetLastError();
}

}  // namespace setk

// ---------------------------------------------------------------------------
// Generic power-of-two STFT / iSTFT (n_fft in [64, 4096], != 512): one
// workgroup per (channel | batch item, frame), iterative radix-2 in LDS.
// Simple and bandwidth-unfriendly on purpose: the n_fft = 512 kernels are the
// hot path, these keep the python API (library default frame_len = 1024) whole.
// ---------------------------------------------------------------------------
namespace setk {

SETK_DEV int reflect_idx_g(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// in-place radix-2 DIT on bit-reversed input; tw[k] = exp(-2 pi i k / n), k < n/2
template <int DIR>
SETK_DEV void fft_radix2_lds(cf* buf, const cf* tw, int n, int logn) {
    for (int s = 1; s <= logn; ++s) {
        const int half = 1 << (s - 1);
        const int stride = n >> s;  // twiddle step
        __syncthreads();
        for (int b = threadIdx.x; b < n / 2; b += blockDim.x) {
            const int grp = b / half, pos = b - grp * half;
            const int i0 = grp * 2 * half + pos, i1 = i0 + half;
            cf w = tw[pos * stride];
            if (DIR > 0) w.y = -w.y;
            const cf u = buf[i0];
            const cf t = cmul(buf[i1], w);
            buf[i0] = cadd(u, t);
            buf[i1] = csub(u, t);
        }
    }
    __syncthreads();
}

// natural order in -> bit-reversed order out (decimation in frequency)
template <int DIR>
SETK_DEV void fft_radix2_dif_lds(cf* buf, const cf* tw, int n, int logn) {
    for (int s = logn; s >= 1; --s) {
        const int half = 1 << (s - 1);
        const int stride = n >> s;
        __syncthreads();
        for (int b = threadIdx.x; b < n / 2; b += blockDim.x) {
            const int grp = b / half, pos = b - grp * half;
            const int i0 = grp * 2 * half + pos, i1 = i0 + half;
            cf w = tw[pos * stride];
            if (DIR > 0) w.y = -w.y;
            const cf u = buf[i0], v = buf[i1];
            buf[i0] = cadd(u, v);
            buf[i1] = cmul(csub(u, v), w);
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// Transform sizes that are not a power of two (--round-power-of-two false with
// e.g. frame_len 400; libs/utils.py:115 hands n_fft = frame_len to librosa):
// Bluestein's chirp-z form of the length-n DFT on top of the radix-2 routines,
//   X[k] = c[k] * sum_j (x[j] c[j]) conj(c[k - j]),  c[k] = exp(-i pi k^2 / n),
// i.e. one circular convolution of length M = 2^m >= 2n - 1: forward DIF ->
// multiply by the (host-computed, bit-reversed) spectrum of conj(c) -> inverse
// DIT.  On entry buf[0..n) holds x[j] c[j] (natural order), buf[n..M) zeros; on
// return buf[k], k < n, holds M * sum_j ... (the caller scales by c[k] / M).
// ---------------------------------------------------------------------------
SETK_DEV void bluestein_core(cf* buf, const cf* tw, const cf* __restrict__ bhat_br, int M,
                             int logM) {
    fft_radix2_dif_lds<-1>(buf, tw, M, logM);
    for (int i = threadIdx.x; i < M; i += blockDim.x) buf[i] = cmul(buf[i], bhat_br[i]);
    fft_radix2_lds<+1>(buf, tw, M, logM);
}

__global__ __launch_bounds__(256) void stft_bluestein_kernel(
    const float* __restrict__ audio, int n_samp, int T, int n_fft, int M, int logM, int hop, int pad,
    const float* __restrict__ window, const cf* __restrict__ twg, const cf* __restrict__ chirp,
    const cf* __restrict__ bhat_br, cf* __restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    cf* buf = reinterpret_cast<cf*>(gsm);
    cf* tw = buf + M;
    const int t = blockIdx.x, c = blockIdx.y;
    const int F = n_fft / 2 + 1;
    const float* x = audio + (size_t)c * n_samp;
    const int s0 = t * hop - pad;
    for (int i = threadIdx.x; i < M / 2; i += blockDim.x) tw[i] = twg[i];
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        cf v = make_float2(0.f, 0.f);
        if (i < n_fft) {
            const float xv = 2.f * window[i] * x[reflect_idx_g(s0 + i, n_samp)];
            v = make_float2(xv * chirp[i].x, xv * chirp[i].y);
        }
        buf[i] = v;
    }
    bluestein_core(buf, tw, bhat_br, M, logM);
    cf* out = spec + ((size_t)c * T + t) * F;
    const float sc = 1.f / (float)M;
    for (int k = threadIdx.x; k < F; k += blockDim.x) out[k] = cscale(cmul(buf[k], chirp[k]), sc);
}

// irfft through the same forward core: x = conj(DFT(conj(Y))) / n
__global__ __launch_bounds__(256) void istft_frames_bluestein_kernel(
    const cf* __restrict__ spec, int T, int n_fft, int M, int logM,
    const float* __restrict__ window, const cf* __restrict__ twg, const cf* __restrict__ chirp,
    const cf* __restrict__ bhat_br, float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    cf* buf = reinterpret_cast<cf*>(gsm);
    cf* tw = buf + M;
    const int t = blockIdx.x, b = blockIdx.y;
    const int F = n_fft / 2 + 1;
    const cf* in = spec + ((size_t)b * T + t) * F;
    for (int i = threadIdx.x; i < M / 2; i += blockDim.x) tw[i] = twg[i];
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        cf v = make_float2(0.f, 0.f);
        if (i < n_fft) {
            cf y;
            if (i < F) {
                y = in[i];
                if (i == 0 || i == n_fft / 2) y.y = 0.f;  // numpy irfft drops these
            } else {
                const cf m = in[n_fft - i];
                y = make_float2(m.x, -m.y);
            }
            v = cmul(make_float2(y.x, -y.y), chirp[i]);  // conj(Y) c
        }
        buf[i] = v;
    }
    bluestein_core(buf, tw, bhat_br, M, logM);
    float* out = frames + ((size_t)b * T + t) * n_fft;
    // Re(conj(z)) = Re(z); window table is 0.5-scaled
    const float sc = 2.f / ((float)M * (float)n_fft);
    for (int i = threadIdx.x; i < n_fft; i += blockDim.x)
        out[i] = cmul(buf[i], chirp[i]).x * sc * window[i];
}

__global__ __launch_bounds__(256) void stft_generic_kernel(const float* __restrict__ audio,
                                                           int n_samp, int T, int n_fft, int logn,
                                                           int hop, int pad,
                                                           const float* __restrict__ window,
                                                           const cf* __restrict__ twg,
                                                           cf* __restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    cf* buf = reinterpret_cast<cf*>(gsm);  // [n_fft]
    cf* tw = buf + n_fft;                  // [n_fft / 2]
    const int t = blockIdx.x, c = blockIdx.y;
    const int F = n_fft / 2 + 1;
    const float* x = audio + (size_t)c * n_samp;
    const int s0 = t * hop - pad;
    for (int i = threadIdx.x; i < n_fft / 2; i += blockDim.x) tw[i] = twg[i];
    for (int i = threadIdx.x; i < n_fft; i += blockDim.x) {
        const int r = (int)(__brev((unsigned)i) >> (32 - logn));
        // window[i] is the 0.5-scaled table (see capi.hip): undo the 1/2 here
        buf[r] = make_float2(2.f * window[i] * x[reflect_idx_g(s0 + i, n_samp)], 0.f);
    }
    fft_radix2_lds<-1>(buf, tw, n_fft, logn);
    cf* out = spec + ((size_t)c * T + t) * F;
    for (int k = threadIdx.x; k < F; k += blockDim.x) out[k] = buf[k];
}

// spec[b][t][F] -> windowed time frames[b][t][n_fft]
__global__ __launch_bounds__(256) void istft_frames_kernel(const cf* __restrict__ spec, int T,
                                                           int n_fft, int logn,
                                                           const float* __restrict__ window,
                                                           const cf* __restrict__ twg,
                                                           float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    cf* buf = reinterpret_cast<cf*>(gsm);
    cf* tw = buf + n_fft;
    const int t = blockIdx.x, b = blockIdx.y;
    const int F = n_fft / 2 + 1;
    const cf* in = spec + ((size_t)b * T + t) * F;
    for (int i = threadIdx.x; i < n_fft / 2; i += blockDim.x) tw[i] = twg[i];
    for (int i = threadIdx.x; i < n_fft; i += blockDim.x) {
        const int r = (int)(__brev((unsigned)i) >> (32 - logn));
        cf v;
        if (i < F) {
            v = in[i];
            if (i == 0 || i == n_fft / 2) v.y = 0.f;  // numpy irfft drops these
        } else {
            const cf m = in[n_fft - i];
            v = make_float2(m.x, -m.y);
        }
        buf[r] = v;
    }
    fft_radix2_lds<+1>(buf, tw, n_fft, logn);
    float* out = frames + ((size_t)b * T + t) * n_fft;
    const float sc = 2.f / (float)n_fft;  // window table is 0.5-scaled
    for (int i = threadIdx.x; i < n_fft; i += blockDim.x) out[i] = buf[i].x * sc * window[i];
}

// overlap-add, / sum(window^2) where > tiny, centre trim, max |y|
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames,
                                                        int t_stride, int T, int n_fft, int hop,
                                                        int pad, int out_len,
                                                        const float* __restrict__ winsq,
                                                        float* __restrict__ wave,
                                                        unsigned* __restrict__ outmax) {
    const int b = blockIdx.y;
    float mx = 0.f;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < out_len; o += gridDim.x * 256) {
        const int n = o + pad;
        float v = 0.f, wss = 0.f;
        if (n < n_fft + hop * (T - 1)) {
            const int t_hi = min(n / hop, T - 1);
            const int t_lo = (n < n_fft) ? 0 : (n - n_fft) / hop + 1;
            for (int t = t_lo; t <= t_hi; ++t) {
                const int off = n - t * hop;
                v += frames[((size_t)b * t_stride + t) * n_fft + off];
                wss += winsq[off];
            }
            if (wss > 1.17549435e-38f) v /= wss;
        }
        wave[(size_t)b * out_len + o] = v;
        mx = fmaxf(mx, fabsf(v));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
    if ((threadIdx.x & 63) == 0) atomicMax(outmax + b, __float_as_uint(mx));
}

// wave[b] *= norm[b] / (max|wave[b]| + eps) where norm[b] > 0
__global__ void istft_scale_kernel(float* __restrict__ wave, int out_len,
                                   const float* __restrict__ norm,
                                   const unsigned* __restrict__ outmax) {
    const int b = blockIdx.y;
    const float nv = norm[b];
    if (!(nv > 0.f)) return;
    const float sc = nv / (__uint_as_float(outmax[b]) + 1.1920928955078125e-07f);
    for (int o = blockIdx.x * 256 + threadIdx.x; o < out_len; o += gridDim.x * 256)
        wave[(size_t)b * out_len + o] *= sc;
}

static hipError_t allow_lds(const void* fn, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t launch_stft_generic(const float* audio, int C, int n_samp, int T, int n_fft, int hop,
                               int pad, const float* window, const float* tw, float* spec,
                               const BluesteinPlan* bp, hipStream_t s) {
    if (bp && bp->M) {
        int logM = 0;
        while ((1 << logM) < bp->M) ++logM;
        const size_t lds = (size_t)bp->M * sizeof(cf) + (size_t)(bp->M / 2) * sizeof(cf);
        hipError_t e = allow_lds(reinterpret_cast<const void*>(stft_bluestein_kernel), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(stft_bluestein_kernel, dim3(T, C), dim3(256), lds, s, audio, n_samp, T,
                           n_fft, bp->M, logM, hop, pad, window, reinterpret_cast<const cf*>(tw),
                           reinterpret_cast<const cf*>(bp->chirp),
                           reinterpret_cast<const cf*>(bp->bhat_br), reinterpret_cast<cf*>(spec));
        return hipGetLastError();
    }
    int logn = 0;
    while ((1 << logn) < n_fft) ++logn;
    const size_t lds = (size_t)n_fft * sizeof(cf) + (size_t)(n_fft / 2) * sizeof(cf);
    hipLaunchKernelGGL(stft_generic_kernel, dim3(T, C), dim3(256), lds, s, audio, n_samp, T, n_fft,
                       logn, hop, pad, window, reinterpret_cast<const cf*>(tw),
                       reinterpret_cast<cf*>(spec));
    return hipGetLastError();
}

hipError_t launch_istft_generic(const float* spec, int B, int T, int n_fft, int hop, int pad,
                                int out_len, const float* window, const float* winsq,
                                const float* tw, float* frames, float* wave, unsigned* outmax,
                                const float* norm, int T_eff, const BluesteinPlan* bp,
                                hipStream_t s) {
    if (bp && bp->M) {
        int logM = 0;
        while ((1 << logM) < bp->M) ++logM;
        const size_t lds = (size_t)bp->M * sizeof(cf) + (size_t)(bp->M / 2) * sizeof(cf);
        hipError_t e = allow_lds(reinterpret_cast<const void*>(istft_frames_bluestein_kernel), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(istft_frames_bluestein_kernel, dim3(T_eff, B), dim3(256), lds, s,
                           reinterpret_cast<const cf*>(spec), T, n_fft, bp->M, logM, window,
                           reinterpret_cast<const cf*>(tw), reinterpret_cast<const cf*>(bp->chirp),
                           reinterpret_cast<const cf*>(bp->bhat_br), frames);
    } else {
        int logn = 0;
        while ((1 << logn) < n_fft) ++logn;
        const size_t lds = (size_t)n_fft * sizeof(cf) + (size_t)(n_fft / 2) * sizeof(cf);
        hipLaunchKernelGGL(istft_frames_kernel, dim3(T_eff, B), dim3(256), lds, s,
                           reinterpret_cast<const cf*>(spec), T, n_fft, logn, window,
                           reinterpret_cast<const cf*>(tw), frames);
    }
    int bx = (out_len + 256 * 4 - 1) / (256 * 4);
    bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
    hipLaunchKernelGGL(istft_ola_kernel, dim3(bx, B), dim3(256), 0, s, frames, T, T_eff, n_fft, hop, pad,
                       out_len, winsq, wave, outmax);
    if (norm)
        hipLaunchKernelGGL(istft_scale_kernel, dim3(bx, B), dim3(256), 0, s, wave, out_len, norm,
                           outmax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// PCM16 ingest: interleaved wav frames pcm[n][c] (int16) -> audio[c][n] float32
// with soundfile's dtype="float32" scaling (x / 32768, exact), i.e. read_wav
// (libs/utils.py:65-92) + the transpose to C x N on the device: half the PCIe
// bytes of a float upload and no host conversion.  One thread per frame n reads
// its C samples (contiguous) and scatters them to the C rows (coalesced per row).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcm16_to_float_kernel(const int16_t* __restrict__ pcm, int C,
                                                             int N, float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int16_t* src = pcm + (size_t)n * C;
    for (int c = 0; c < C; ++c) out[(size_t)c * N + n] = (float)src[c] * (1.0f / 32768.0f);
}

hipError_t launch_pcm16_to_float(const int16_t* pcm, int C, int N, float* out, hipStream_t s) {
    hipLaunchKernelGGL(pcm16_to_float_kernel, dim3((N + 255) / 256), dim3(256), 0, s, pcm, C, N,
                       out);
    return hipGetLastError();
}

// The way back for a multi-channel result: float rows [C][N] -> interleaved 16-bit frames
// [N][C], quantised by libsndfile's float -> short rule as the host writer applies it
// (wavio.float_to_pcm16: rint(x * 32767) in double, no clipping -- out-of-range values wrap).
__global__ __launch_bounds__(256) void float_to_pcm16_kernel(const float* __restrict__ in, int C, int N,
                                                             int16_t* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    for (int c = 0; c < C; ++c) {
        const double v = rint((double)in[(size_t)c * N + n] * 32767.0);
        out[(size_t)n * C + c] = (int16_t)(long long)v;
    }
}

hipError_t launch_float_to_pcm16(const float* in, int C, int N, int16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(float_to_pcm16_kernel, dim3((N + 255) / 256), dim3(256), 0, s, in, C, N, out);
    return hipGetLastError();
}

// Batched form: one launch for a whole staging slab (blockIdx.y = utterance).
// Optionally sums x^2 of channel 0 (SpectrogramReader.power, only used for the
// CLI's log line) into power0[u] (double, atomics: log precision only).
struct PcmItem {
    const int16_t* pcm;
    float* out;
    int n;
    int pad_;
};

// CT > 0: channel count known at compile time -- a thread takes FOUR frames (4 CT int16 = 64
// bytes at CT = 8, the compiler merges the loads) and stores a float4 per channel; CT = 0: any C, one frame
// and 2-byte loads per thread (also taken by a CT build when an item's pointers are not 4- / 8-byte
// aligned or its frame count is not a multiple of four).  32 x (8 ch x 30 s): 0.34 ms (one frame per thread) -> 0.23 ms, 3.2 TB/s.
template <int CT>
__global__ __launch_bounds__(256) void pcm16_to_float_batch_kernel(const PcmItem* __restrict__ items,
                                                                   int C, double* power0) {
    const PcmItem it = items[blockIdx.y];
    const float k = 1.0f / 32768.0f;
    float acc = 0.f;
    constexpr int CV = CT > 0 ? CT : 1;
    constexpr int FR = 4;  // frames per thread: FR * CV int16 in, one float4 per channel out
    const bool fast = CT > 0 && (reinterpret_cast<uintptr_t>(it.pcm) & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(it.out) & 15) == 0 && (it.n % FR) == 0;
    if (fast) {
        const int groups = it.n / FR;
        for (int p = blockIdx.x * 256 + threadIdx.x; p < groups; p += gridDim.x * 256) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(it.pcm + (size_t)FR * p * CV);
            int16_t v[FR * CV];
#pragma unroll
            for (int w = 0; w < FR * CV / 2; ++w) {
                const uint32_t u = src[w];
                v[2 * w] = (int16_t)(u & 0xffffu);
                v[2 * w + 1] = (int16_t)(u >> 16);
            }
#pragma unroll
            for (int c = 0; c < CV; ++c) {
                const float4 o = make_float4((float)v[c] * k, (float)v[CV + c] * k,
                                             (float)v[2 * CV + c] * k, (float)v[3 * CV + c] * k);
                *reinterpret_cast<float4*>(it.out + (size_t)c * it.n + FR * p) = o;
                if (c == 0) acc = fmaf(o.w, o.w, fmaf(o.z, o.z, fmaf(o.y, o.y, fmaf(o.x, o.x, acc))));
            }
        }
    } else {
        for (int n = blockIdx.x * 256 + threadIdx.x; n < it.n; n += gridDim.x * 256) {
            const int16_t* src = it.pcm + (size_t)n * C;
            for (int c = 0; c < C; ++c) {
                const float v = (float)src[c] * k;
                it.out[(size_t)c * it.n + n] = v;
                if (c == 0) acc = fmaf(v, v, acc);
            }
        }
    }
    if (power0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(power0 + blockIdx.y, (double)acc);
    }
}

size_t pcm_item_bytes() { return sizeof(PcmItem); }
void pcm_item_fill(void* dst, int i, const int16_t* pcm, float* out, int n) {
    PcmItem* it = static_cast<PcmItem*>(dst) + i;
    it->pcm = pcm;
    it->out = out;
    it->n = n;
    it->pad_ = 0;
}

hipError_t launch_pcm16_to_float_batch(const void* d_items, int n_utts, int C, int max_n,
                                       double* power0, hipStream_t s) {
    int bx = (max_n + 256 * 8 - 1) / (256 * 8);
    bx = bx < 1 ? 1 : bx;
    const PcmItem* items = static_cast<const PcmItem*>(d_items);
    const dim3 grid(bx, n_utts), block(256);
    switch (C) {
        case 2: hipLaunchKernelGGL(pcm16_to_float_batch_kernel<2>, grid, block, 0, s, items, C, power0); break;
        case 4: hipLaunchKernelGGL(pcm16_to_float_batch_kernel<4>, grid, block, 0, s, items, C, power0); break;
        case 6: hipLaunchKernelGGL(pcm16_to_float_batch_kernel<6>, grid, block, 0, s, items, C, power0); break;
        case 8: hipLaunchKernelGGL(pcm16_to_float_batch_kernel<8>, grid, block, 0, s, items, C, power0); break;
        default: hipLaunchKernelGGL(pcm16_to_float_batch_kernel<0>, grid, block, 0, s, items, C, power0);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// 16-bit PCM ingest without a float32 twin: the wave file's interleaved frames pcm[n][c] ->
// planar int16 out[c][stride] (stride even, >= n: UttDesc::ch_stride), which the fused
// kernels read directly (SETK_FLAG_IN_PCM16: 2 bytes per sample in both streaming passes
// instead of 4, 4 C N bytes moved here instead of 6 C N).  read_wav's int16 / 32768
// (libs/utils.py:80-90) happens inside the transforms, folded into their window tables.
// A thread takes FR = 4 frames: CT dwords in, one 8-byte store per channel.  power0 as above
// (the CLI's log line), from the dequantised samples of channel 0.
// ---------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256) void pcm16_deinterleave_batch_kernel(const PcmItem* __restrict__ items,
                                                                       int C, double* power0) {
    const PcmItem it = items[blockIdx.y];
    int16_t* out = reinterpret_cast<int16_t*>(it.out);
    const int stride = it.pad_;
    const float k = 1.0f / 32768.0f;
    float acc = 0.f;
    constexpr int CV = CT > 0 ? CT : 1;
    constexpr int FR = 4;
    const bool fast = CT > 0 && (reinterpret_cast<uintptr_t>(it.pcm) & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(out) & 7) == 0 && (stride % FR) == 0;
    const int groups = fast ? it.n / FR : 0;
    if (fast) {
        for (int p = blockIdx.x * 256 + threadIdx.x; p < groups; p += gridDim.x * 256) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(it.pcm + (size_t)FR * p * CV);
            uint32_t w[FR * CV / 2];
#pragma unroll
            for (int i = 0; i < FR * CV / 2; ++i) w[i] = src[i];
            auto at = [&](int fr, int c) -> uint32_t {  // sample c of frame fr as 16 bits
                const int e = fr * CV + c;
                return (e & 1) ? (w[e / 2] >> 16) : (w[e / 2] & 0xffffu);
            };
#pragma unroll
            for (int c = 0; c < CV; ++c) {
                uint2 o;
                o.x = at(0, c) | (at(1, c) << 16);
                o.y = at(2, c) | (at(3, c) << 16);
                *reinterpret_cast<uint2*>(out + (size_t)c * stride + FR * p) = o;
                if (c == 0) {
#pragma unroll
                    for (int fr = 0; fr < FR; ++fr) {
                        const float v = (float)(int16_t)at(fr, 0) * k;
                        acc = fmaf(v, v, acc);
                    }
                }
            }
        }
    }
    // the frames the fast path leaves (n % 4), or everything
    for (int n = FR * groups + blockIdx.x * 256 + threadIdx.x; n < it.n; n += gridDim.x * 256) {
        const int16_t* src = it.pcm + (size_t)n * C;
        for (int c = 0; c < C; ++c) {
            out[(size_t)c * stride + n] = src[c];
            if (c == 0) {
                const float v = (float)src[0] * k;
                acc = fmaf(v, v, acc);
            }
        }
    }
    // the padding up to the stride reads as silence (never addressed by the transforms, which
    // reflect at num_samples; zeroed so that a dump of the buffer is deterministic)
    for (int n = it.n + blockIdx.x * 256 + threadIdx.x; n < stride; n += gridDim.x * 256)
        for (int c = 0; c < C; ++c) out[(size_t)c * stride + n] = 0;
    if (power0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(power0 + blockIdx.y, (double)acc);
    }
}

void pcm_item_fill_planar(void* dst, int i, const int16_t* pcm, int16_t* out, int n, int stride) {
    PcmItem* it = static_cast<PcmItem*>(dst) + i;
    it->pcm = pcm;
    it->out = reinterpret_cast<float*>(out);
    it->n = n;
    it->pad_ = stride;
}

hipError_t launch_pcm16_deinterleave_batch(const void* d_items, int n_utts, int C, int max_n,
                                           double* power0, hipStream_t s) {
    int bx = (max_n + 256 * 8 - 1) / (256 * 8);
    bx = bx < 1 ? 1 : bx;
    const PcmItem* items = static_cast<const PcmItem*>(d_items);
    const dim3 grid(bx, n_utts), block(256);
    switch (C) {
        case 2: hipLaunchKernelGGL(pcm16_deinterleave_batch_kernel<2>, grid, block, 0, s, items, C, power0); break;
        case 4: hipLaunchKernelGGL(pcm16_deinterleave_batch_kernel<4>, grid, block, 0, s, items, C, power0); break;
        case 6: hipLaunchKernelGGL(pcm16_deinterleave_batch_kernel<6>, grid, block, 0, s, items, C, power0); break;
        case 8: hipLaunchKernelGGL(pcm16_deinterleave_batch_kernel<8>, grid, block, 0, s, items, C, power0); break;
        default: hipLaunchKernelGGL(pcm16_deinterleave_batch_kernel<0>, grid, block, 0, s, items, C, power0);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Kaldi CompressedMatrix bodies -> float32 masks [T][F], a batch per launch (the streaming CLI
// ships a compressed mask as its 1 - 2 bytes per element instead of decoding it on the host to 4).
// Replaces (funcwj/setk): `uncompress`, scripts/sptk/libs/kaldi_io.py:248-292 -- the same
// float32 operations in the same order (no fused multiply-adds, IEEE division), so the result
// equals numpy's bit for bit:
//   CM  (kOneByteWithColHeaders): per column four uint16 percentiles p = u16 * range / 65535 + min,
//        then one byte q per element, COLUMN-major: q <= 64: q (p25 - p0) / 64 + p0;
//        q >= 193: (q - 192)(p100 - p75) / 63 + p75; else (q - 64)(p75 - p25) / 128 + p25
//   CM2 (kTwoByte): min + u16 * float32(range / 65535.0), row-major;  CM3 (kOneByte): min + u8 * float32(range / 255.0)
// transpose: the stored matrix is F x T (apply_adaptive_beamformer.py:146-151 turns it): the
// output is its transpose.  One thread per output element; the bodies are 0.5 - 1 MB, L2 serves the
// strided byte reads of the column-major form.
// ---------------------------------------------------------------------------
struct CmItem {
    const unsigned char* src;
    float* dst;
    float vmin, vrange;
    int rows, cols, kind, transpose;
};
size_t cm_item_bytes() { return sizeof(CmItem); }
void cm_item_fill(void* tbl, int i, const void* src, float* dst, float vmin, float vrange, int rows, int cols,
                  int kind, int transpose) {
    CmItem* it = static_cast<CmItem*>(tbl) + i;
    it->src = static_cast<const unsigned char*>(src);
    it->dst = dst;
    it->vmin = vmin;
    it->vrange = vrange;
    it->rows = rows;
    it->cols = cols;
    it->kind = kind;
    it->transpose = transpose;
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void kaldi_cm_decode_batch_kernel(const CmItem* __restrict__ items) {
    const CmItem it = items[blockIdx.y];
    const long n = (long)it.rows * it.cols;
    const int ocols = it.transpose ? it.rows : it.cols;
    const unsigned short* hdr = reinterpret_cast<const unsigned short*>(it.src);   // CM: [cols][4]
    const unsigned char* body = it.src + (it.kind == 1 ? (size_t)8 * it.cols : 0);
    const float step2 = (float)((double)it.vrange / 65535.0), step3 = (float)((double)it.vrange / 255.0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int orow = (int)(i / ocols), ocol = (int)(i - (long)orow * ocols);
        const int r = it.transpose ? ocol : orow, c = it.transpose ? orow : ocol;
        float v;
        if (it.kind == 1) {
            const float p0 = (float)hdr[4 * c + 0] * it.vrange / 65535.0f + it.vmin;
            const float p25 = (float)hdr[4 * c + 1] * it.vrange / 65535.0f + it.vmin;
            const float p75 = (float)hdr[4 * c + 2] * it.vrange / 65535.0f + it.vmin;
            const float p100 = (float)hdr[4 * c + 3] * it.vrange / 65535.0f + it.vmin;
            const float q = (float)body[(size_t)c * it.rows + r];
            if (q <= 64.f) v = q * (p25 - p0) / 64.0f + p0;
            else if (q >= 193.f) v = (q - 192.f) * (p100 - p75) / 63.0f + p75;
            else v = (q - 64.f) * (p75 - p25) / 128.0f + p25;
        } else if (it.kind == 2) {
            const float q = (float)reinterpret_cast<const unsigned short*>(body)[(size_t)r * it.cols + c];
            v = it.vmin + q * step2;
        } else {
            const float q = (float)body[(size_t)r * it.cols + c];
            v = it.vmin + q * step3;
        }
        it.dst[i] = v;
    }
}
#pragma clang fp contract(fast)

hipError_t launch_kaldi_cm_decode_batch(const void* d_items, int n, long max_elems, hipStream_t s) {
    long bx = (max_elems + 256 * 4 - 1) / (256 * 4);
    bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
    hipLaunchKernelGGL(kaldi_cm_decode_batch_kernel, dim3((unsigned)bx, n), dim3(256), 0, s,
                       static_cast<const CmItem*>(d_items));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Helpers of the fixed-weight path (apply_fixed_beamformer.py:38-48):
// max |audio| per utterance (SpectrogramReader.maxabs, the renorm target) and
// the reference's F x M weight sets -> the planar [C][264] layout of pass 2.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxabs_kernel(const UttDesc* utts, int C,
                                                     unsigned* norm_bits) {
    const UttDesc ud = utts[blockIdx.y];
    const size_t n = (size_t)C * ud.num_samples;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        m = fmaxf(m, fabsf(ud.audio[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(norm_bits + blockIdx.y, __float_as_uint(m));
}

hipError_t launch_maxabs(const UttDesc* utts, int C, unsigned* norm_bits, int n_utts, int max_samples,
                         hipStream_t s) {
    int bx = (int)(((size_t)C * max_samples + 256 * 16 - 1) / (256 * 16));
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(maxabs_kernel, dim3(bx, n_utts), dim3(256), 0, s, utts, C, norm_bits);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void pack_fixed_weights_kernel(const float2* __restrict__ sets,
                                                                 const int* __restrict__ index,
                                                                 int C, float2* __restrict__ out) {
    const int u = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= kBinsPad) return;
    const float2* w = sets + (size_t)(index ? index[u] : 0) * kBins * C;
    for (int c = 0; c < C; ++c)
        out[((size_t)u * C + c) * kBinsPad + f] = (f < kBins) ? w[(size_t)f * C + c] : make_float2(0.f, 0.f);
}

hipError_t launch_pack_fixed_weights(const float* sets, const int* index, int n_utts, int C,
                                     float* out, hipStream_t s) {
    hipLaunchKernelGGL(pack_fixed_weights_kernel, dim3((kBinsPad + 255) / 256, n_utts), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(sets), index, C,
                       reinterpret_cast<float2*>(out));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// directional_feats (libs/spatial.py:184-208):
//   out[t][f] = mean_p cos((arg X_i - arg X_j)[t][f] - (arg v_i - arg v_j)[f])
// over the microphone pairs p = (i, j).  spec [C][T][F], sv [F][C] complex64.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void directional_feats_kernel(
    const float2* __restrict__ spec, const float2* __restrict__ sv, const int* __restrict__ pairs,
    int n_pairs, int C, int T, int F, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)T * F;
    if (i >= n) return;
    const int f = (int)(i % F);
    float acc = 0.f;
    for (int p = 0; p < n_pairs; ++p) {
        const int a = pairs[2 * p], b = pairs[2 * p + 1];
        const float2 xa = spec[(size_t)a * n + i], xb = spec[(size_t)b * n + i];
        const float2 va = sv[(size_t)f * C + a], vb = sv[(size_t)f * C + b];
        const float ds = atan2f(xa.y, xa.x) - atan2f(xb.y, xb.x);
        const float dt = atan2f(va.y, va.x) - atan2f(vb.y, vb.x);
        acc += cosf(ds - dt);
    }
    out[i] = acc / (float)n_pairs;
}

hipError_t launch_directional_feats(const float* spec, const float* sv, const int* pairs, int n_pairs,
                                    int C, int T, int F, float* out, hipStream_t s) {
    const size_t n = (size_t)T * F;
    hipLaunchKernelGGL(directional_feats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(spec), reinterpret_cast<const float2*>(sv),
                       pairs, n_pairs, C, T, F, out);
    return hipGetLastError();
}

}  // namespace setk
