// pass1.hip -- fused windowed rFFT-512 + masked spatial-covariance accumulation.
//
// Replaces (funcwj/setk): SpectrogramReader._load (libs/data_handler.py:492-503)
// -> forward_stft (libs/utils.py:96-138) -> compute_covar x2
// (libs/beamformer.py:87-103, 279-281).  X is never written to HBM.
//
// A 1024-thread workgroup walks a contiguous frame range of one utterance in
// tiles of TB = 32/C frames (<= 8), its 16 waves split in two roles (4 waves per
// SIMD at <= 128 VGPRs -- each role alone fits the budget, merged they need 256):
//   transform waves (0-7)   each of the 32 quad-rows (16 lanes) turns one
//            (frame, channel) pair into its spectrum on its own -- global load,
//            window, radix-16, 16x16 LDS transpose, radix-16, Hermitian split --
//            with no workgroup barrier inside (a quad-row lives in one wave);
//   covariance waves (8-15) thread (f, h) folds the outer products x x^H of bin f
//            for the tile's frames, weighted by the speech and noise masks, into
//            register accumulators; the Hermitian pairs are split between the two
//            halves h (real-only diagonals apart) so a thread carries 64 sums.
// Tile k+1 is transformed into one half of a double-buffered LDS tile while tile
// k is folded from the other: ONE s_barrier per tile.  The per-range sums leave
// the chip once, as a partial slab reduced by covar_finalize.
// Roofline: HBM by construction (4*C*N + 4*T*F bytes per utterance); at C = 8
// the VALU and LDS work is what is actually felt -- see DESIGN.md section 4/5.
//
// Experiment hooks (tools/mk_abl.sh): -DSETK_ONLY_PROD / -DSETK_ONLY_CONS run one
// role alone, -DSETK_NO_GLOAD skips the audio loads (timing only, wrong results).
#include "common.h"
#include "fft512.h"
#include "covar_fold.h"
#include <cstdio>
#include <type_traits>
#include <cstdlib>

namespace setk {

template <int C, bool DUMP, bool PCM = false>
__global__ __launch_bounds__(1024, 4) void stft_covar_kernel(Pass1Args a) {
    constexpr int NT = 1024;
    constexpr int TB = pass1_tile_frames(C);
    constexpr int NF = TB * C;            // transforms per tile (<= 32)
    constexpr int NP = npairs(C);
    constexpr int ND = PairSplit<C>::ND, NO = PairSplit<C>::NO;
    constexpr int NS = 32 / NF;           // transform sets (small C: tiles alternate sets)
    constexpr int F = kBins, FP = kBinsPad;
    constexpr int ROW = 17;               // transpose / table row stride (fft512.h)
    constexpr int SL = slot_entries(ROW); // slot stride (padded 16x16 transpose)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* xt0 = reinterpret_cast<cf*>(smem);            // [2][NF][SL]
    cf* win_l = xt0 + 2 * NF * SL;                    // per-lane table rows, see fft512.h
    cf* tw_l = win_l + LaneTab<ROW>::size;
    cf* tw5_l = tw_l + LaneTab<ROW>::size;
    float* xn0 = reinterpret_cast<float*>(win_l + table_entries(ROW));  // [2][32] nyquist bins (real)
    float* nym = xn0 + 64;                            // [2][2][8] bin-256 weights (speech|noise)
    float* red = nym + 32;                            // [16]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples;
    const int T = ud.num_frames;
    const bool clamp = (a.flags & 0x2) != 0;
    const bool has_mn = ud.mask_n != nullptr;

    fill_lane_tables<ROW>(win_l, tw_l, tw5_l, a.window, a.tw256, a.tw512, tid, NT);

    float mx = 0.f;
    // covariance-role state (declared here: the epilogue stores it)
    const int ct = tid - 512;
    const int f = ct & 255, q = (ct >> 8) & 1;
    float dg_s[ND], dg_n[ND];  // diagonal (real) sums of this half
    cf of_s[NO > 0 ? NO : 1], of_n[NO > 0 ? NO : 1];
    // aux0/aux1: mask sums in the q == 0 half; the q == 1 half has them free and
    // carries the Nyquist-bin items (bin 256 is purely real) in aux0:
    //   item < NP: speech pair | item < 2 NP: noise pair | 2NP, 2NP+1: mask sums
    float aux0 = 0.f, aux1 = 0.f;
    const int ny_item = ct - 256;
    const bool ny_active = !DUMP && ny_item >= 0 && ny_item < 2 * NP + 2;

    if (wave < 8) {
#ifndef SETK_ONLY_CONS
        // ================= transform waves =================
        const int la = tid & 15, grp = tid >> 4;
        const int my_set = grp / NF;
        const int my_i = grp - my_set * NF;
        const int my_tt = my_i / C, my_c = my_i - my_tt * C;
        const bool producer = my_set < NS;
        gcfloat_p my_audio = gptr(ud.audio) + (size_t)my_c * n_samp;
        gcshort_p my_pcm = (gcshort_p)gptr(ud.audio) + (size_t)my_c * ud.ch_stride;
        const cf* win_row = win_l + la * LaneTab<ROW>::lstride;
        const cf* tw_row = tw_l + la * LaneTab<ROW>::lstride;
        const cf* tw5_row = tw5_l + la * LaneTab<ROW>::lstride5;
        const bool ny_lane = !DUMP && producer && my_c == 0 && la == 0;

        typename std::conditional<PCM, int, cf>::type raw[16];  // PCM: packed pairs of samples
        float raw_ms = 0.f, raw_mn = 0.f;
        bool raw_ok = false, raw_last = false;
        auto fetch = [&](int tb_tile) {
            const int t = tb_tile + my_tt;
            raw_ok = t < wi.t1;
            raw_last = t == T - 1;
#ifdef SETK_NO_GLOAD
            const bool raw_go = raw_ok && n_samp < 0;
#else
            const bool raw_go = raw_ok;
#endif
            if constexpr (PCM) load_raw_pcm(raw, my_pcm, n_samp, t * a.g.hop - a.g.pad, la, raw_go);
            else load_raw(raw, my_audio, n_samp, t * a.g.hop - a.g.pad, la, raw_go);
            if (ny_lane) {
                raw_ms = 0.f;
                raw_mn = 0.f;
                if (raw_ok) {
                    raw_ms = gptr(ud.mask_s)[(size_t)t * F + 256];
                    if (has_mn) raw_mn = gptr(ud.mask_n)[(size_t)t * F + 256];
                }
            }
        };
        // max |x|: with hop <= 256 the first halves of consecutive frames tile the
        // signal, so only the last frame of the utterance needs its second half
        const bool half_max = a.g.hop <= kNfft / 2;
        auto produce = [&](int b, int tb_next_own) {
            cf* slot = xt0 + (b * NF + my_i) * SL;
            cf v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if constexpr (PCM) v[j] = unpack_pcm(raw[j]);  // (integers as floats; the window carries 2^-15)
                else v[j] = raw[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = max3_abs(mx, v[j].x, v[j].y);
            if (!half_max || raw_last) {
#pragma unroll
                for (int j = 8; j < 16; ++j) mx = max3_abs(mx, v[j].x, v[j].y);
            }
            apply_window<ROW>(v, win_row);
            if (ny_lane) {
                const float s = clamp ? fminf(raw_ms, 1.f) : raw_ms;
                nym[(b * 2 + 0) * 8 + my_tt] = s;
                nym[(b * 2 + 1) * 8 + my_tt] = raw_ok ? (has_mn ? raw_mn : 1.f - s) : 0.f;
            }
            fft256_stage_a_pad<-1, ROW>(v, slot, tw_row, la);
            __builtin_amdgcn_wave_barrier();
            // the next frame is requested between the two radix-16 stages: the
            // first stage's working set is dead, so the 32 registers in flight fit
            // the 128-VGPR budget without spills (requested before stage one:
            // 1.07 ms, here: 0.915, after the split: 0.935)
            if (tb_next_own < wi.t1) fetch(tb_next_own);
            qr_stage23<ROW>(slot, xn0 + b * 32 + my_i, tw5_row, la);
        };

        wg_barrier();  // tables ready
        if (producer) fetch(wi.t0 + my_set * TB);
        if (producer && my_set == 0) produce(0, wi.t0 + NS * TB);
        wg_barrier();
        int buf = 0, next_set = 1 % NS;
#pragma unroll 1
        for (int tb = wi.t0; tb < wi.t1; tb += TB, buf ^= 1) {
            const bool prod = producer && (my_set == next_set) && (tb + TB < wi.t1);
            next_set = (next_set + 1 == NS) ? 0 : next_set + 1;
            if (prod) produce(buf ^ 1, tb + TB + NS * TB);
            wg_barrier();
        }
#endif
    } else {
#ifndef SETK_ONLY_PROD
        // ================= covariance waves =================
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
        // mask rows: global -> registers, one tile ahead
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
        if (!DUMP) fetch_masks(wi.t0, cur_s, cur_n);
        wg_barrier();  // tables ready
        wg_barrier();  // tile 0 transformed
        int buf = 0;
#pragma unroll 1
        for (int tb = wi.t0; tb < wi.t1; tb += TB, buf ^= 1) {
            const cf* xt = xt0 + buf * NF * SL;
            const float* xn = xn0 + buf * 32;
            if (DUMP) {
                // spec[c][t][f], f fastest
                for (int i = q; i < NF; i += 2) {
                    const int tt = i / C, c = i - tt * C;
                    const int t = tb + tt;
                    if (t < wi.t1) {
                        // one spectrogram (setk_stft) or one per utterance (setk_stft_batch:
                        // the descriptor's output pointer)
                        float2* base = a.spec_dump ? reinterpret_cast<float2*>(a.spec_dump)
                                                   : reinterpret_cast<float2*>(ud.wave_out);
                        float2* dst = base + ((size_t)c * T + t) * (a.dump_pitch ? a.dump_pitch : F);
                        dst[f] = xt[i * SL + f];
                        if (f == 0) dst[256] = make_float2(xn[i], 0.f);
                    }
                }
            } else {
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
                        const float prod_ny =
                            (ny_item < 2 * NP) ? xn[tt * C + ny_i] * xn[tt * C + ny_j] : 1.f;
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
            }
            wg_barrier();
        }
#endif
    }

    if (!DUMP) {
        // ---- max |audio| (the renorm target, WaveReader.maxabs) ----
        if (wi.last) {
            // samples after the last frame's span are never loaded above
            const int covered = (T - 1) * a.g.hop - a.g.pad + kNfft;
            for (int c = 0; c < C; ++c)
                for (int i = covered + tid; i < n_samp; i += NT) {
                    if constexpr (PCM)
                        mx = fmaxf(mx, fabsf((float)((gcshort_p)gptr(ud.audio))[(size_t)c * ud.ch_stride + i]));
                    else
                        mx = fmaxf(mx, fabsf(gptr(ud.audio)[(size_t)c * n_samp + i]));
                }
        }
        if constexpr (PCM) mx *= 3.0517578125e-05f;  // 2^-15: max |int16| -> max |x| (exact)
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
}

// ---- STFT straight into the bin-major layout of the bin-resident CGMM (cgmm_bin.hip) ----
// out[f][c][t] (frames contiguous, pitch Tp = (T + 3) & ~3) per utterance, no [C][T][F]
// intermediate and no transpose pass.  A 1024-thread workgroup owns 64 consecutive frames
// of one utterance and walks the channels: its 64 quad-rows transform the 64 frames of
// channel c into 64 LDS slots (the spectrum ends up as slot[bin]); then every wave writes
// rows of 64 frames = 512 contiguous bytes per (bin, channel) with 16-byte stores (a lane
// carries two consecutive frames, a wave two bins); the slot stride is odd (273 entries),
// which spreads the strided LDS reads of a bin column over the banks.
constexpr int kBmFrames = 64;
constexpr int kBmSlot = slot_entries(17) + 1;

__global__ __launch_bounds__(1024, 4) void stft_binmajor_kernel(Pass1Args a, int C) {
    constexpr int ROW = 17;
    constexpr int NT = 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* xt = reinterpret_cast<cf*>(smem);                      // [64][kBmSlot]
    cf* win_l = xt + kBmFrames * kBmSlot;
    cf* tw_l = win_l + LaneTab<ROW>::size;
    cf* tw5_l = tw_l + LaneTab<ROW>::size;
    float* xn = reinterpret_cast<float*>(win_l + table_entries(ROW));  // [64] bin 256 (real)

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int la = tid & 15, grp = tid >> 4;
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples, T = ud.num_frames;
    const int Tp = (T + 3) & ~3;
    cf* out = reinterpret_cast<cf*>(ud.wave_out);
    fill_lane_tables<ROW>(win_l, tw_l, tw5_l, a.window, a.tw256, a.tw512, tid, NT);
    const cf* win_row = win_l + la * LaneTab<ROW>::lstride;
    const cf* tw_row = tw_l + la * LaneTab<ROW>::lstride;
    const cf* tw5_row = tw5_l + la * LaneTab<ROW>::lstride5;
    cf* slot = xt + grp * kBmSlot;
    const int t = wi.t0 + grp;
    const bool valid = t < wi.t1;

    const int s0 = t * a.g.hop - a.g.pad;
    __syncthreads();  // tables ready
    // (a prefetch of the next channel's frame across the write-out keeps 32 registers live
    // over the loop edge and spills ~60 at the 128-VGPR budget of a 1024-thread workgroup;
    // the loads are issued at the top of each channel step instead)
    for (int c = 0; c < C; ++c) {
        cf v[16];
        load_raw(v, gptr(ud.audio) + (size_t)c * n_samp, n_samp, s0, la, valid);
        apply_window<ROW>(v, win_row);
        fft256_stage_a_pad<-1, ROW>(v, slot, tw_row, la);
        __builtin_amdgcn_wave_barrier();
        qr_stage23<ROW>(slot, xn + grp, tw5_row, la);
        __syncthreads();
        // 16-byte stores: a lane writes two consecutive frames of one bin, a wave two bins
        {
            const int fr = 2 * (lane & 31), half = lane >> 5;
            const int t2 = wi.t0 + fr;
            for (int f = 2 * wave + half; f < kBins; f += 2 * (NT / 64)) {
                cf v0, v1;
                if (f < 256) {
                    v0 = xt[fr * kBmSlot + f];
                    v1 = xt[(fr + 1) * kBmSlot + f];
                } else {
                    v0 = make_float2(xn[fr], 0.f);
                    v1 = make_float2(xn[fr + 1], 0.f);
                }
                cf* dst = out + ((size_t)f * C + c) * Tp + t2;
                if (t2 + 1 < wi.t1)  // Tp and the block start are even: 16-byte aligned
                    *reinterpret_cast<float4*>(dst) = make_float4(v0.x, v0.y, v1.x, v1.y);
                else if (t2 < wi.t1)
                    *dst = v0;
            }
        }
        __syncthreads();
    }
}

hipError_t launch_stft_binmajor(int C, const Pass1Args& a, int n_items, hipStream_t s) {
    const size_t lds = (size_t)kBmFrames * kBmSlot * sizeof(cf) + table_entries(17) * sizeof(cf) +
                       kBmFrames * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stft_binmajor_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(stft_binmajor_kernel, dim3(n_items), dim3(1024), lds, s, a, C);
    return hipGetLastError();
}

template <int C, bool DUMP, bool PCM = false>
static hipError_t launch_pass1_t(const Pass1Args& a, int n_items, hipStream_t s) {
    constexpr int NF = pass1_tile_frames(C) * C;
    const size_t lds = (size_t)2 * NF * slot_entries(17) * sizeof(cf) + table_entries(17) * sizeof(cf) +
                       (64 + 32 + 16) * sizeof(float);
    auto k = stft_covar_kernel<C, DUMP, PCM>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_items), dim3(1024), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pass1(int C, bool dump, const Pass1Args& a, int n_items, hipStream_t s, bool pcm16) {
    if (pcm16 && dump) return hipErrorInvalidValue;  // (the spectrogram dump takes float32 samples)
#define SETK_CASE(c)                                                \
    case c:                                                         \
        if (dump) return launch_pass1_t<c, true>(a, n_items, s);     \
        if (pcm16) return launch_pass1_t<c, false, true>(a, n_items, s); \
        return launch_pass1_t<c, false>(a, n_items, s);
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
__global__ __launch_bounds__(320) void covar_finalize_kernel(FinalizeArgs a) {
    const int f = threadIdx.x;  // one workgroup = one (plane, utterance) row of 264 bins
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
        // (one scale per row, then a product: the fused form in solve.hip evaluates the same
        //  two expressions, so both routes give the same bits)
        const float sc = a.num_scale / fmaxf(den, 1e-6f);
        out[f] = acc * sc;
    } else {
        // Ry: all-ones mask == speech + noise numerators when mask_n = 1 - mask_s
        const int r = e - 4 * NP;  // [0, 2NP)
        float acc = 0.f;
        for (int p = 0; p < ud.nparts; ++p)
            acc += P[p * slab + (size_t)r * kBinsPad + f] +
                   P[p * slab + (size_t)(2 * NP + r) * kBinsPad + f];
        const float sc = a.num_scale / fmaxf((float)ud.num_frames, 1e-6f);
        out[f] = acc * sc;
    }
}

hipError_t launch_finalize(const FinalizeArgs& a, int n_utts, hipStream_t s) {
    const int NP = npairs(a.num_channels);
    const int planes_out = a.with_ry ? 6 * NP : 4 * NP;
    dim3 grid(1, planes_out, n_utts);
    hipLaunchKernelGGL(covar_finalize_kernel, grid, dim3(320), 0, s, a);
    return hipGetLastError();
}

}  // namespace setk
