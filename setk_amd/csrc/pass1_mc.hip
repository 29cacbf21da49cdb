// pass1_mc.hip -- pass 1 with the transforms on the matrix cores (mcdft.h).
//
// Replaces (funcwj/setk): SpectrogramReader._load (libs/data_handler.py:492-503)
// -> forward_stft (libs/utils.py:96-138) -> compute_covar x2
// (libs/beamformer.py:87-103, 279-281).  X is never written to HBM.
//
// Same frame of work as pass1.hip -- a 1024-thread workgroup walks a frame range of one
// utterance in tiles of TB = 32/C frames through a double-buffered LDS spectrum tile, ONE
// s_barrier per tile, waves 8-15 fold the masked outer products -- but the 32 transforms of a
// tile are dense contractions on the fp16 matrix pipe now (one wavefront = one transform:
// 8 samples per lane in, 4 bins per lane out, ~60 VALU + 12 MFMA wave-instructions against 161
// VALU), which runs BESIDE the vector ALUs: the transform waves leave two thirds of the issue
// slots they used to take to the covariance waves.  The spectra are scaled by 2^10 / peak
// (fp16 operand range, mcdft.h); the covariance numerators carry that factor squared and
// covar_finalize removes it (a power of two: exact).
#include "common.h"
#include "fft512.h"
#include "covar_fold.h"
#include "mcdft.h"
#include <cstdio>
#include <cstdlib>

namespace setk {

constexpr int kMcSlot = 272;  // complex entries per spectrum slot (257 used; 16-byte multiple)

// raw frame samples in the operand layout of mcdft.h: v[e] = x[s + mc::sample_of(lane, e)]
template <class FloatPtr>
SETK_DEV void load_raw_mc(float (&v)[8], FloatPtr x, int n_samp, int s, int lane, bool valid) {
    if (!valid) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        return;
    }
    const int o = 64 * (lane >> 4) + (lane & 15);  // mc::sample_of(lane, e) = o + 16 e (+ 192 for e >= 4)
    if (s >= 0 && s + kFrame <= n_samp) {
        FloatPtr p = x + s + o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = p[16 * e];
            v[4 + e] = p[256 + 16 * e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = x[reflect_index(s + o + 16 * e, n_samp)];
            v[4 + e] = x[reflect_index(s + o + 256 + 16 * e, n_samp)];
        }
    }
}

template <int C>
__global__ __launch_bounds__(1024, 4) void stft_covar_mc_kernel(Pass1Args a) {
    constexpr int NT = 1024;
    constexpr int TB = pass1_tile_frames(C);
    constexpr int NF = TB * C;            // transforms per tile (<= 32)
    constexpr int NP = npairs(C);
    constexpr int ND = PairSplit<C>::ND, NO = PairSplit<C>::NO;
    constexpr int F = kBins, FP = kBinsPad;
    constexpr int SL = kMcSlot;
    constexpr int PER = (NF + 7) / 8;     // transforms per transform wave and tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* xt0 = reinterpret_cast<cf*>(smem);                              // [2][NF][SL]
    float* a16s = reinterpret_cast<float*>(xt0 + 2 * NF * SL);          // [8][16][kOddPitch]
    float* nym = a16s + 8 * 16 * mc::kOddPitch;                         // [2][2][8] bin-256 weights
    float* red = nym + 32;                                              // [16]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples;
    const int T = ud.num_frames;
    const bool clamp = (a.flags & 0x2) != 0;
    const bool has_mn = ud.mask_n != nullptr;

    float mx = 0.f;
    const int ct = tid - 512;
    const int f = ct & 255, q = (ct >> 8) & 1;
    float dg_s[ND], dg_n[ND];
    cf of_s[NO > 0 ? NO : 1], of_n[NO > 0 ? NO : 1];
    float aux0 = 0.f, aux1 = 0.f;  // see pass1.hip: mask sums (q == 0) / Nyquist items (q == 1)
    const int ny_item = ct - 256;
    const bool ny_active = ny_item >= 0 && ny_item < 2 * NP + 2;

    if (wave < 8) {
#ifndef SETK_ONLY_CONS
        // ================= transform waves =================
        const int lane = tid & 63;
        const int c16 = lane & 15, g = lane >> 4;
        const int nv = (NF - wave + 7) / 8;  // transforms of this wave per tile (wave-uniform)
        mc::Fwd K;
        mc::load_fwd(K, a.mc_tab, lane);
        const mc::h8 ot_h = mc::tab_h8(a.mc_tab, mc::kW_OT_H, lane);
        const mc::h8 ot_l = mc::tab_h8(a.mc_tab, mc::kW_OT_L, lane);
        float win[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) win[e] = gptr(a.mc_win)[e * 64 + lane];
        float* a16w = a16s + wave * 16 * mc::kOddPitch;
        const int lane_bin = mc::bin_of(c16, g, 0);
        const bool st_ok = !(c16 == 0 && g >= 2);
        const bool st_256 = (c16 == 0 && g == 2);

        // samples (and the bin-256 mask entries of channel 0) travel SETK_P1MC_PF transforms
        // ahead of their use: a wave issues its VALU instructions ~8 cycles apart whatever its
        // neighbours do, so one transform lasts ~1000 cycles -- less than a loaded HBM round trip
#ifndef SETK_P1MC_PF
#define SETK_P1MC_PF 2
#endif
        struct Stage {
            float raw[8];
            float ms, mn;
            bool ok;
        };
        Stage st[SETK_P1MC_PF];
        // transform number jt (>= nv: of a later tile) counted from tile tb_tile
        auto fetch = [&](Stage& S, int tb_tile, int jt) {
            while (jt >= nv) {
                jt -= nv;
                tb_tile += TB;
            }
            S.ok = false;
            S.ms = S.mn = 0.f;
            if (tb_tile >= wi.t1) return;  // past the range: never consumed
            const int i = wave + 8 * jt;
            const int tt = i / C, c = i - tt * C;
            const int t = tb_tile + tt;
            S.ok = t < wi.t1;
            load_raw_mc(S.raw, gptr(ud.audio) + (size_t)c * n_samp, n_samp, t * a.g.hop - a.g.pad, lane, S.ok);
            if (c == 0 && lane == 0 && S.ok) {
                S.ms = gptr(ud.mask_s)[(size_t)t * F + 256];
                if (has_mn) S.mn = gptr(ud.mask_n)[(size_t)t * F + 256];
            }
        };
        auto produce = [&](int b, int tb_tile) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (j < nv) {
                    const int i = wave + 8 * j;
                    const int tt = i / C, c = i - tt * C;
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = st[0].raw[e];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) mx = max3_abs(mx, x[e], x[e + 1]);
                    if (c == 0 && lane == 0) {
                        const float s = clamp ? fminf(st[0].ms, 1.f) : st[0].ms;
                        nym[(b * 2 + 0) * 8 + tt] = s;
                        nym[(b * 2 + 1) * 8 + tt] = st[0].ok ? (has_mn ? st[0].mn : 1.f - s) : 0.f;
                    }
#pragma unroll
                    for (int d = 0; d + 1 < SETK_P1MC_PF; ++d) st[d] = st[d + 1];
                    fetch(st[SETK_P1MC_PF - 1], tb_tile, j + SETK_P1MC_PF);
                    mc::f4 zr, zi, a16;
                    mc::forward(x, win, K, zr, zi, a16);
                    cf* slot = xt0 + (b * NF + i) * SL;
                    mc::lds_fp re = mc::to_lds(reinterpret_cast<float*>(slot + lane_bin));
                    mc::lds_fp im = mc::opaque_next(re);
                    if (st_ok) mc::store_bins(re, im, zr, zi);
                    if (st_256) slot[256] = make_float2(zr[3], 0.f);
                    mc::store_a16(a16w, j, lane, a16);
                }
            }
            // odd family X[16 + 32 q] of this wave's transforms (columns j < nv of the tile)
            const mc::f4 d = mc::odd_tile(a16w, ot_h, ot_l, lane);
            if (c16 < nv) {
                cf* sj = xt0 + (b * NF + wave + 8 * c16) * SL;
                sj[16 + 64 * g] = make_float2(d[0], d[1]);
                sj[48 + 64 * g] = make_float2(d[2], d[3]);
            }
        };

#pragma unroll
        for (int d = 0; d < SETK_P1MC_PF; ++d) fetch(st[d], wi.t0, d);
        produce(0, wi.t0);
        wg_barrier();
        int buf = 0;
#pragma unroll 1
        for (int tb = wi.t0; tb < wi.t1; tb += TB, buf ^= 1) {
            if (tb + TB < wi.t1) produce(buf ^ 1, tb + TB);
            wg_barrier();
        }
#endif
    } else {
#ifndef SETK_ONLY_PROD
        // ================= covariance waves (as pass1.hip) =================
#pragma unroll
        for (int e = 0; e < ND; ++e) dg_s[e] = dg_n[e] = 0.f;
#pragma unroll
        for (int e = 0; e < NO; ++e) {
            of_s[e] = make_float2(0.f, 0.f);
            of_n[e] = make_float2(0.f, 0.f);
        }
        int ny_i = 0, ny_j = 0;
        {
            const int e = (ny_item < NP) ? ny_item : ny_item - NP;
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < C; ++i)
#pragma unroll
                for (int j = i; j < C; ++j) {
                    if (cnt == e) { ny_i = i; ny_j = j; }
                    ++cnt;
                }
        }
        float cur_s[TB], cur_n[TB], nxt_s[TB], nxt_n[TB];
        auto fetch_masks = [&](int tb_tile, float (&ms)[TB], float (&mn)[TB]) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                const int t = tb_tile + tt;
                float vs = 0.f, vn = 0.f;
                if (t < wi.t1) {
                    vs = gptr(ud.mask_s)[(size_t)t * F + f];
                    if (has_mn) vn = gptr(ud.mask_n)[(size_t)t * F + f];
                }
                ms[tt] = vs;
                mn[tt] = vn;
            }
        };
        fetch_masks(wi.t0, cur_s, cur_n);
        wg_barrier();  // tile 0 transformed
        int buf = 0;
#pragma unroll 1
        for (int tb = wi.t0; tb < wi.t1; tb += TB, buf ^= 1) {
            const cf* xt = xt0 + buf * NF * SL;
            if (tb + TB < wi.t1) fetch_masks(tb + TB, nxt_s, nxt_n);
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                cf x[C];
#pragma unroll
                for (int c = 0; c < C; ++c) x[c] = xt[(tt * C + c) * SL + f];
                const bool fvalid = tb + tt < wi.t1;
                const float ws = clamp ? fminf(cur_s[tt], 1.f) : cur_s[tt];
                const float wn = fvalid ? (has_mn ? cur_n[tt] : 1.f - ws) : 0.f;
                if (q == 0) {
                    aux0 += ws;
                    aux1 += wn;
                    accumulate_half<C, 0>(x, ws, wn, dg_s, dg_n, of_s, of_n);
                } else {
                    accumulate_half<C, 1>(x, ws, wn, dg_s, dg_n, of_s, of_n);
                }
                if (ny_active) {
                    const float prod_ny = (ny_item < 2 * NP)
                                              ? xt[(tt * C + ny_i) * SL + 256].x * xt[(tt * C + ny_j) * SL + 256].x
                                              : 1.f;
                    const bool speech = (ny_item < NP) || (ny_item == 2 * NP);
                    const float w256 = nym[(buf * 2 + (speech ? 0 : 1)) * 8 + tt];
                    aux0 = fmaf(w256, prod_ny, aux0);
                }
            }
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                cur_s[tt] = nxt_s[tt];
                cur_n[tt] = nxt_n[tt];
            }
            wg_barrier();
        }
#endif
    }

    // ---- max |audio| (the renorm target, WaveReader.maxabs) ----
    if (wi.last) {
        const int covered = (T - 1) * a.g.hop - a.g.pad + kNfft;
        for (int c = 0; c < C; ++c)
            for (int i = covered + tid; i < n_samp; i += NT)
                mx = fmaxf(mx, fabsf(gptr(ud.audio)[(size_t)c * n_samp + i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        float bm = red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) bm = fmaxf(bm, red[w]);
        atomicMax(a.norm_bits + wi.utt, __float_as_uint(bm));
    }
    // ---- partial slab: planes [s.re | s.im | n.re | n.im | sum_s sum_n] ----
    if (wave >= 8) {
        float* P = a.partials + (size_t)wi.part * nplanes_partial(C) * FP;
        if (q == 0) {
            store_half<C, 0>(P, f, dg_s, dg_n, of_s, of_n);
            P[(4 * NP + 0) * FP + f] = aux0;
            P[(4 * NP + 1) * FP + f] = aux1;
        } else {
            store_half<C, 1>(P, f, dg_s, dg_n, of_s, of_n);
        }
        if (ny_active) {
            if (ny_item < NP) {
                P[(0 * NP + ny_item) * FP + 256] = aux0;
                P[(1 * NP + ny_item) * FP + 256] = 0.f;
            } else if (ny_item < 2 * NP) {
                P[(2 * NP + ny_item - NP) * FP + 256] = aux0;
                P[(3 * NP + ny_item - NP) * FP + 256] = 0.f;
            } else {
                P[(4 * NP + ny_item - 2 * NP) * FP + 256] = aux0;
            }
        }
    }
}

template <int C>
static hipError_t launch_pass1_mc_t(const Pass1Args& a, int n_items, hipStream_t s) {
    constexpr int NF = pass1_tile_frames(C) * C;
    const size_t lds = (size_t)2 * NF * kMcSlot * sizeof(cf) +
                       (8 * 16 * mc::kOddPitch + 32 + 16) * sizeof(float);
    auto k = stft_covar_mc_kernel<C>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_items), dim3(1024), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pass1_mc(int C, const Pass1Args& a, int n_items, hipStream_t s) {
    switch (C) {
        case 1: return launch_pass1_mc_t<1>(a, n_items, s);
        case 2: return launch_pass1_mc_t<2>(a, n_items, s);
        case 3: return launch_pass1_mc_t<3>(a, n_items, s);
        case 4: return launch_pass1_mc_t<4>(a, n_items, s);
        case 5: return launch_pass1_mc_t<5>(a, n_items, s);
        case 6: return launch_pass1_mc_t<6>(a, n_items, s);
        case 7: return launch_pass1_mc_t<7>(a, n_items, s);
        case 8: return launch_pass1_mc_t<8>(a, n_items, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace setk
