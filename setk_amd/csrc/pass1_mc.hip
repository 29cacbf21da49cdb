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

    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* xt0 = reinterpret_cast<cf*>(smem);                              // [2][NF][SL]
    mc::u4* ktiles = reinterpret_cast<mc::u4*>(xt0 + 2 * NF * SL);      // [10][64]: the forward's 8 operand tiles, OT_H, OT_L
    float* a16s = reinterpret_cast<float*>(ktiles + 10 * 64);           // [8][4][kOddPitch]
    float* nym = a16s + 8 * 4 * mc::kOddPitch;                          // [2][2][8] bin-256 weights
    float* red = nym + 32;                                              // [16]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples;
    const int T = ud.num_frames;
    const bool clamp = (a.flags & 0x2) != 0;
    const bool has_mn = ud.mask_n != nullptr;

    mc::stage_tiles(ktiles, a.mc_tab, mc::kW_MC_H, 8, tid, NT);
    mc::stage_tiles(ktiles + 8 * 64, a.mc_tab, mc::kW_OT_H, 2, tid, NT);
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
        // C divides 8: G = 8 / C waves per channel, each owns FPW = TB / G CONSECUTIVE frames of
        // a tile.  With hop = 256 consecutive frames share half their samples in the same lane's
        // registers (mc::sample_of), so the wave works on a stream of half-frames: NH per tile,
        // held one tile ahead in H (16 - 20 registers; a half is requested for the NEXT tile the
        // moment its slot is consumed, a whole tile period before its use).
        constexpr int G = 8 / C;
        constexpr int FPW = TB / G;
        constexpr int NH = FPW + (G > 1 ? 1 : 0);  // halves a tile consumes (G == 1: the first one is carried over)
        const int lane = tid & 63;
        const int c16 = lane & 15, g = lane >> 4;
        const int ch = wave % C, gi = wave / C;   // wave-uniform
        float tr[4], ti[4], tri[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tr[r] = mc::tab_f(a.mc_tab, mc::kW_TR + r, lane);
            ti[r] = mc::tab_f(a.mc_tab, mc::kW_TI + r, lane);
            tri[r] = mc::tab_f(a.mc_tab, mc::kW_TRI + r, lane);
        }
        float win[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) win[e] = gptr(a.mc_win)[e * 64 + lane];
        float* a16w = a16s + wave * 4 * mc::kOddPitch;
        const int lane_bin = mc::bin_of(c16, g, 0);
        const bool st_ok = !(c16 == 0 && g >= 2);
        const bool st_256 = (c16 == 0 && g == 2);
        gcfloat_p xa = gptr(ud.audio) + (size_t)ch * n_samp;
        const int lo = 64 * g + c16;
        const int pad = a.g.pad;

        // half-frame u = padded positions [256 u, 256 u + 256): frame t = halves (t, t + 1);
        // halves past the last frame repeat the last one (frames beyond t1 carry zero weights)
        auto load_half = [&](float (&h)[4], int u) {
            u = min(u, T);
            const int s0 = 256 * u - pad;
            if (s0 >= 0 && s0 + 256 <= n_samp) {
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = xa[s0 + lo + 16 * e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = xa[reflect_index(s0 + lo + 16 * e, n_samp)];
            }
        };
        // half number i (< NH) of the tile that starts at frame tb
        auto half_of = [&](int tb, int i) { return tb + gi * FPW + i + (G > 1 ? 0 : 1); };
        float H[NH][4];
        float prev[4];  // first half of the next transform (G == 1: across tiles too)
        float ms[FPW], mn[FPW];  // bin-256 mask entries of the next tile (channel-0 waves, lane 0)
        auto fetch_ny = [&](int tb) {
#pragma unroll
            for (int j = 0; j < FPW; ++j) {
                const int t = tb + gi * FPW + j;
                ms[j] = mn[j] = 0.f;
                if (ch == 0 && lane == 0 && t < wi.t1) {
                    ms[j] = gptr(ud.mask_s)[(size_t)t * F + 256];
                    if (has_mn) mn[j] = gptr(ud.mask_n)[(size_t)t * F + 256];
                }
            }
        };
        if (G == 1) load_half(prev, wi.t0);
#pragma unroll
        for (int i = 0; i < NH; ++i) load_half(H[i], half_of(wi.t0, i));
        fetch_ny(wi.t0);

        auto produce = [&](int b, int tb) {
#pragma unroll
            for (int j = 0; j < FPW; ++j) {
                const int i = (gi * FPW + j) * C + ch;  // slot of (frame tb + gi FPW + j, channel ch)
                float x[8];
                if (G > 1 && j == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) prev[e] = H[0][e];
                    load_half(H[0], half_of(tb + TB, 0));
                }
                const int hs = j + (G > 1 ? 1 : 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = prev[e];
                    x[4 + e] = H[hs][e];
                    prev[e] = H[hs][e];
                }
                mx = max3_abs(mx, x[4], x[5]);
                mx = max3_abs(mx, x[6], x[7]);
                if (G > 1 && j == 0) {
                    mx = max3_abs(mx, x[0], x[1]);
                    mx = max3_abs(mx, x[2], x[3]);
                }
                load_half(H[hs], half_of(tb + TB, hs));  // the same slot, one tile ahead
                if (ch == 0 && lane == 0) {
                    const int tt = gi * FPW + j;
                    const float sv = clamp ? fminf(ms[j], 1.f) : ms[j];
                    nym[(b * 2 + 0) * 8 + tt] = sv;
                    nym[(b * 2 + 1) * 8 + tt] = (tb + tt < wi.t1) ? (has_mn ? mn[j] : 1.f - sv) : 0.f;
                }
                mc::f4 zr, zi, a16;
                asm volatile("" ::: "memory");  // the operand tiles are re-read per transform, not kept
                mc::forward_t(x, win, [&](int k) { return mc::lds_h8(ktiles, k, lane); }, tr, ti, tri, zr, zi, a16);
                cf* slot = xt0 + (b * NF + i) * SL;
                mc::lds_fp re = mc::to_lds(reinterpret_cast<float*>(slot + lane_bin));
                mc::lds_fp im = mc::opaque_next(re);
                if (st_ok) mc::store_bins(re, im, zr, zi);
                if (st_256) slot[256] = make_float2(zr[3], 0.f);
                mc::store_a16(a16w, j, lane, a16);
            }
            if (G == 1 && wi.t0 == tb) {
                // (the very first half of the range was outside H: nothing else to do)
            }
            fetch_ny(tb + TB);
            // odd family X[16 + 32 q] of this wave's transforms (columns j < FPW of the tile)
            const mc::f4 d = mc::odd_tile(a16w, mc::lds_h8(ktiles, 8, lane), mc::lds_h8(ktiles, 9, lane), lane, FPW);
            if (c16 < FPW) {
                cf* sj = xt0 + (b * NF + (gi * FPW + c16) * C + ch) * SL;
                sj[16 + 64 * g] = make_float2(d[0], d[1]);
                sj[48 + 64 * g] = make_float2(d[2], d[3]);
            }
        };
        // G == 1: the first half of the range's first frame also counts for max |x|
        if (G == 1) {
            mx = max3_abs(mx, prev[0], prev[1]);
            mx = max3_abs(mx, prev[2], prev[3]);
        }
        wg_barrier();  // operand tiles staged
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
        wg_barrier();  // operand tiles staged
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
    const size_t lds = (size_t)2 * NF * kMcSlot * sizeof(cf) + 10 * 1024 +
                       (8 * 4 * mc::kOddPitch + 32 + 16) * sizeof(float);
    auto k = stft_covar_mc_kernel<C>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_items), dim3(1024), lds, s, a);
    return hipGetLastError();
}

// channel counts that divide 8 and hop = n_fft / 2 (the half-frame stream of the transform
// waves); everything else keeps pass1.hip
bool pass1_mc_supported(int C, int hop) { return hop == kNfft / 2 && (C == 1 || C == 2 || C == 4 || C == 8); }

hipError_t launch_pass1_mc(int C, const Pass1Args& a, int n_items, hipStream_t s) {
    if (!pass1_mc_supported(C, a.g.hop)) return hipErrorInvalidValue;
    switch (C) {
        case 1: return launch_pass1_mc_t<1>(a, n_items, s);
        case 2: return launch_pass1_mc_t<2>(a, n_items, s);
        case 4: return launch_pass1_mc_t<4>(a, n_items, s);
        case 8: return launch_pass1_mc_t<8>(a, n_items, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace setk
