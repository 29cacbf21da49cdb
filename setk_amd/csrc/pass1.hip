// pass1.hip -- fused windowed rFFT-512 + masked spatial-covariance accumulation.
//
// Replaces (funcwj/setk): SpectrogramReader._load (libs/data_handler.py:492-503)
// -> forward_stft (libs/utils.py:96-138) -> compute_covar x2
// (libs/beamformer.py:87-103, 279-281).  X is never written to HBM.
//
// A 512-thread workgroup walks a contiguous frame range of one utterance in
// tiles of TB = 32/C frames:
//   produce  each of the 32 quad-rows (16 lanes) transforms one (frame,
//            channel) pair on its own -- global load + window, radix-16,
//            LDS transpose, radix-16, Hermitian split -- with no workgroup
//            barrier inside (a quad-row lives in one wavefront);
//   consume  thread (f, h) = (tid & 255, tid >> 8) folds the tile's outer
//            products x x^H of bin f, weighted by the speech and noise masks,
//            into register accumulators; the Hermitian pairs are split between
//            the two halves h so that a thread carries <= 72 accumulators and
//            two workgroups (4 waves/SIMD) fit a CU.
// The per-range sums leave the chip once, as a partial slab reduced by
// covar_finalize.  Roofline: HBM by construction (4*C*N + 4*T*F bytes per
// utterance); at C = 8 the VALU work is of the same order -- see DESIGN.md.
#include "common.h"
#include "fft512.h"
#include <cstdio>
#include <cstdlib>

namespace setk {

// numpy "reflect" index (no edge repeat); valid for -N < i < 2N-1
SETK_DEV int reflect_index(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// Load one windowed frame as 16 packed complex points per lane:
// v[j] = (x[s+2n] w[2n], x[s+2n+1] w[2n+1]),  n = la + 16 j.
// mx accumulates max |x| over the raw samples.
SETK_DEV void load_frame(cf (&v)[16], const float* __restrict__ x, int n_samp, int s,
                         int la, const float* win, bool valid, float& mx) {
    const float2* w2 = reinterpret_cast<const float2*>(win);
    if (!valid) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = make_float2(0.f, 0.f);
        return;
    }
    const bool interior = (s >= 0) && (s + kNfft <= n_samp) &&
                          ((reinterpret_cast<uintptr_t>(x + s) & 7) == 0);
    if (interior) {
        const float2* p = reinterpret_cast<const float2*>(x + s);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = la + 16 * j;
            const float2 d = p[n];
            const float2 w = w2[n];
            mx = fmaxf(mx, fmaxf(fabsf(d.x), fabsf(d.y)));
            v[j] = make_float2(d.x * w.x, d.y * w.y);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = la + 16 * j;
            const float d0 = x[reflect_index(s + 2 * n, n_samp)];
            const float d1 = x[reflect_index(s + 2 * n + 1, n_samp)];
            const float2 w = w2[n];
            mx = fmaxf(mx, fmaxf(fabsf(d0), fabsf(d1)));
            v[j] = make_float2(d0 * w.x, d1 * w.y);
        }
    }
}

// One quad-row: full forward real transform of the frame in v; on return the
// slot holds X[0..255] and *nyq = X[256].  No workgroup barrier: the 16 lanes
// share a wavefront and LDS operations of a wavefront complete in order.
SETK_DEV void quadrow_rfft(cf (&v)[16], cf* slot, float* nyq, const cf* tw, const cf* tw5,
                           int la) {
    fft256_stage_a<-1>(v, slot, tw, la);
    __builtin_amdgcn_wave_barrier();
    fft256_stage_b<-1>(v, slot, la);
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) slot[la + 16 * kb] = v[dft16_pos(kb)];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int k = la + 16 * m;
        const cf Zk = slot[k];
        const cf Zm = slot[(256 - k) & 255];
        cf Xk, Xm;
        rfft_split(Zk, Zm, tw5[k], Xk, Xm);
        if (k == 0) {
            const cf Z128 = slot[128];
            slot[0] = make_float2(Xk.x, 0.f);
            *nyq = Xm.x;
            slot[128] = make_float2(Z128.x, -Z128.y);
        } else {
            slot[k] = Xk;
            slot[256 - k] = Xm;
        }
    }
}

// outer products of the pairs [LO, HI) of the (i <= j) enumeration
template <int C, int LO, int HI>
SETK_DEV void accumulate_pairs(const cf (&x)[C], float ws, float wn, cf* acc_s, cf* acc_n) {
    int e = 0;
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
        for (int j = i; j < C; ++j) {
            if (e >= LO && e < HI) {
                const cf p = cmulc(x[i], x[j]);
                acc_s[e - LO].x = fmaf(ws, p.x, acc_s[e - LO].x);
                acc_n[e - LO].x = fmaf(wn, p.x, acc_n[e - LO].x);
                if (i != j) {
                    acc_s[e - LO].y = fmaf(ws, p.y, acc_s[e - LO].y);
                    acc_n[e - LO].y = fmaf(wn, p.y, acc_n[e - LO].y);
                }
            }
            ++e;
        }
}

template <int C, int LO, int HI>
SETK_DEV void store_pairs(float* P, int f, const cf* acc_s, const cf* acc_n) {
    constexpr int NP = npairs(C);
    constexpr int FP = kBinsPad;
#pragma unroll
    for (int e = LO; e < HI; ++e) {
        P[(0 * NP + e) * FP + f] = acc_s[e - LO].x;
        P[(1 * NP + e) * FP + f] = acc_s[e - LO].y;
        P[(2 * NP + e) * FP + f] = acc_n[e - LO].x;
        P[(3 * NP + e) * FP + f] = acc_n[e - LO].y;
    }
}

constexpr int kP1Threads = 512;

template <int C, bool DUMP>
__global__ __launch_bounds__(kP1Threads, 2) void stft_covar_kernel(Pass1Args a) {
    constexpr int TB = 32 / C;  // frames per tile
    constexpr int NF = TB * C;  // transforms per tile (<= 32)
    constexpr int NP = npairs(C);
    constexpr int NPH = (NP + 1) / 2;  // pairs of half 0; half 1 takes the rest
    constexpr int F = kBins, FP = kBinsPad;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* xt = reinterpret_cast<cf*>(smem);             // [NF][256]
    cf* tw = xt + NF * 256;                           // [16][16]
    cf* tw5 = tw + 256;                               // [128]
    float* win = reinterpret_cast<float*>(tw5 + 128);  // [512]
    float* xn = win + kNfft;                          // [32] nyquist bins (real)
    float* red = xn + 32;                             // [8]

    const int tid = threadIdx.x;
    const int la = tid & 15, grp = tid >> 4;
    const int f = tid & 255, h = tid >> 8;
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples;
    const int T = ud.num_frames;

    if (tid < 256) tw[tid] = a.tw256[tid];
    if (tid < 128) tw5[tid] = a.tw512[tid];
    win[tid] = a.window[tid];

    cf acc_s[NPH], acc_n[NPH];
    float sum_s = 0.f, sum_n = 0.f;
#pragma unroll
    for (int e = 0; e < NPH; ++e) {
        acc_s[e] = make_float2(0.f, 0.f);
        acc_n[e] = make_float2(0.f, 0.f);
    }
    // nyquist-bin items (bin 256 is purely real): threads [0, 2*NP+2)
    //   item < NP        : speech pair item
    //   item < 2 NP      : noise  pair item - NP
    //   2NP, 2NP+1       : mask sums (speech, noise)
    int ny_i = 0, ny_j = 0;
    const int ny_item = tid;
    const bool ny_active = !DUMP && (tid < 2 * NP + 2);
    {
        const int e = (tid < NP) ? tid : tid - NP;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < C; ++i)
#pragma unroll
            for (int j = i; j < C; ++j) {
                if (cnt == e) { ny_i = i; ny_j = j; }
                ++cnt;
            }
    }
    float ny_acc = 0.f;
    float mx = 0.f;

    const bool clamp = (a.flags & 0x2) != 0;
    const bool has_mn = ud.mask_n != nullptr;
    const int my_tt = grp / C, my_c = grp - my_tt * C;  // this quad-row's transform

    for (int tb = wi.t0; tb < wi.t1; tb += TB) {
        // ---- mask prefetch (bin f, nyquist column for the ny threads) ----
        float ms[TB], mn[TB], nyw[TB];
        if (!DUMP) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                const int t = tb + tt;
                const bool valid = t < wi.t1;
                float s = 0.f, n = 0.f, w = 0.f;
                if (valid) {
                    s = ud.mask_s[(size_t)t * F + f];
                    if (clamp) s = fminf(s, 1.f);
                    n = has_mn ? ud.mask_n[(size_t)t * F + f] : 1.f - s;
                    if (ny_active) {
                        float s256 = ud.mask_s[(size_t)t * F + 256];
                        if (clamp) s256 = fminf(s256, 1.f);
                        const float n256 = has_mn ? ud.mask_n[(size_t)t * F + 256] : 1.f - s256;
                        const bool speech = (ny_item < NP) || (ny_item == 2 * NP);
                        w = speech ? s256 : n256;
                    }
                }
                ms[tt] = s;
                mn[tt] = n;
                nyw[tt] = w;
            }
        }
        __syncthreads();  // tables ready / previous tile fully consumed

        // ---- produce: one transform per quad-row ----
        if (grp < NF) {
            const int t = tb + my_tt;
            cf v[16];
            load_frame(v, ud.audio + (size_t)my_c * n_samp, n_samp, t * a.g.hop - a.g.pad, la, win,
                       t < wi.t1, mx);
            quadrow_rfft(v, xt + grp * 256, xn + grp, tw, tw5, la);
        }
        __syncthreads();

        if (DUMP) {
            // spec[c][t][f], f fastest
            for (int i = h; i < NF; i += 2) {
                const int tt = i / C, c = i - tt * C;
                const int t = tb + tt;
                if (t < wi.t1) {
                    float2* dst = reinterpret_cast<float2*>(a.spec_dump) + ((size_t)c * T + t) * F;
                    dst[f] = xt[i * 256 + f];
                    if (f == 0) dst[256] = make_float2(xn[i], 0.f);
                }
            }
        } else {
            // ---- consume: masked outer products of bin f, pair half h ----
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                cf x[C];
#pragma unroll
                for (int c = 0; c < C; ++c) x[c] = xt[(tt * C + c) * 256 + f];
                const float ws = ms[tt], wn = mn[tt];
                if (h == 0) {
                    sum_s += ws;
                    sum_n += wn;
                    accumulate_pairs<C, 0, NPH>(x, ws, wn, acc_s, acc_n);
                } else {
                    accumulate_pairs<C, NPH, NP>(x, ws, wn, acc_s, acc_n);
                }
                if (ny_active) {
                    const float prod = (ny_item < 2 * NP)
                                           ? xn[tt * C + ny_i] * xn[tt * C + ny_j]
                                           : 1.f;
                    ny_acc = fmaf(nyw[tt], prod, ny_acc);
                }
            }
        }
    }

    if (!DUMP) {
        // ---- max |audio| (the renorm target, WaveReader.maxabs) ----
        if (wi.last) {
            // samples after the last frame's span are never loaded above
            const int covered = (T - 1) * a.g.hop - a.g.pad + kNfft;
            for (int c = 0; c < C; ++c)
                for (int i = covered + tid; i < n_samp; i += kP1Threads)
                    mx = fmaxf(mx, fabsf(ud.audio[(size_t)c * n_samp + i]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            float bm = red[0];
#pragma unroll
            for (int w = 1; w < kP1Threads / 64; ++w) bm = fmaxf(bm, red[w]);
            atomicMax(a.norm_bits + wi.utt, __float_as_uint(bm));
        }

        // ---- partial slab: planes [s.re | s.im | n.re | n.im | sum_s sum_n] ----
        float* P = a.partials + (size_t)wi.part * nplanes_partial(C) * FP;
        if (h == 0) {
            store_pairs<C, 0, NPH>(P, f, acc_s, acc_n);
            P[(4 * NP + 0) * FP + f] = sum_s;
            P[(4 * NP + 1) * FP + f] = sum_n;
        } else {
            store_pairs<C, NPH, NP>(P, f, acc_s, acc_n);
        }
        if (ny_active) {
            if (ny_item < NP) {
                P[(0 * NP + ny_item) * FP + 256] = ny_acc;
                P[(1 * NP + ny_item) * FP + 256] = 0.f;
            } else if (ny_item < 2 * NP) {
                P[(2 * NP + ny_item - NP) * FP + 256] = ny_acc;
                P[(3 * NP + ny_item - NP) * FP + 256] = 0.f;
            } else {
                P[(4 * NP + ny_item - 2 * NP) * FP + 256] = ny_acc;
            }
        }
    }
}

template <int C, bool DUMP>
static hipError_t launch_pass1_t(const Pass1Args& a, int n_items, hipStream_t s) {
    constexpr int TB = 32 / C, NF = TB * C;
    const size_t lds = (size_t)NF * 256 * sizeof(cf) + 256 * sizeof(cf) + 128 * sizeof(cf) +
                       kNfft * sizeof(float) + 32 * sizeof(float) + 8 * sizeof(float);
    auto k = stft_covar_kernel<C, DUMP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (getenv("SETK_DEBUG")) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k),
                                                           kP1Threads, lds);
        fprintf(stderr, "[setk] pass1<%d,%d> lds=%zu items=%d blocks/CU=%d\n", C, (int)DUMP, lds,
                n_items, nb);
    }
    hipLaunchKernelGGL(k, dim3(n_items), dim3(kP1Threads), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pass1(int C, bool dump, const Pass1Args& a, int n_items, hipStream_t s) {
#define SETK_CASE(c)                                                        \
    case c:                                                                 \
        return dump ? launch_pass1_t<c, true>(a, n_items, s)                \
                    : launch_pass1_t<c, false>(a, n_items, s);
    switch (C) {
        SETK_CASE(1)
        SETK_CASE(2)
        SETK_CASE(3)
        SETK_CASE(4)
        SETK_CASE(5)
        SETK_CASE(6)
        SETK_CASE(7)
        SETK_CASE(8)
    }
#undef SETK_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// covar_finalize: sum the partial slabs of each utterance and normalise by
// max(sum_t m, 1e-6) (libs/beamformer.py:99-102).  Output planes per utterance:
//   [Rs.re NP | Rs.im NP | Rn.re NP | Rn.im NP | (Ry.re NP | Ry.im NP)]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void covar_finalize_kernel(FinalizeArgs a) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int u = blockIdx.z;
    const int C = a.num_channels;
    const int NP = npairs(C);
    const int planes_in = 4 * NP + 2;
    const int planes_out = a.with_ry ? 6 * NP : 4 * NP;
    const int e = blockIdx.y;  // output plane
    if (f >= kBinsPad) return;
    const UttDesc ud = a.utts[u];
    float* out = a.covar + ((size_t)u * planes_out + e) * kBinsPad;
    if (f >= kBins) {
        out[f] = 0.f;
        return;
    }
    const float* P = a.partials + (size_t)ud.part0 * planes_in * kBinsPad;
    const size_t slab = (size_t)planes_in * kBinsPad;
    if (e < 4 * NP) {
        const int sel = e / (2 * NP);  // 0 speech, 1 noise
        float acc = 0.f, den = 0.f;
        for (int p = 0; p < ud.nparts; ++p) {
            acc += P[p * slab + (size_t)e * kBinsPad + f];
            den += P[p * slab + (size_t)(4 * NP + sel) * kBinsPad + f];
        }
        out[f] = acc / fmaxf(den, 1e-6f);
    } else {
        // Ry: all-ones mask == speech + noise numerators when mask_n = 1 - mask_s
        const int r = e - 4 * NP;  // [0, 2NP)
        float acc = 0.f;
        for (int p = 0; p < ud.nparts; ++p)
            acc += P[p * slab + (size_t)r * kBinsPad + f] +
                   P[p * slab + (size_t)(2 * NP + r) * kBinsPad + f];
        out[f] = acc / fmaxf((float)ud.num_frames, 1e-6f);
    }
}

hipError_t launch_finalize(const FinalizeArgs& a, int n_utts, hipStream_t s) {
    const int NP = npairs(a.num_channels);
    const int planes_out = a.with_ry ? 6 * NP : 4 * NP;
    dim3 grid((kBinsPad + 255) / 256, planes_out, n_utts);
    hipLaunchKernelGGL(covar_finalize_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace setk
