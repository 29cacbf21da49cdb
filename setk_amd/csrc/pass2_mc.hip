// pass2_mc.hip -- pass 2 with the transforms on the matrix cores (mcdft.h):
// rDFT recompute + w^H x + inverse rDFT + window + overlap-add.
//
// Replaces (funcwj/setk): Beamformer.beamform (libs/beamformer.py:220-234),
// post-masking (apply_adaptive_beamformer.py:174-175) and inverse_stft
// (libs/utils.py:142-173 -> librosa.istft 0.8.1: irfft, * window, overlap-add,
// / sum(window^2) where > tiny, trim n_fft/2; the inf-norm rescale is scale_kernel).
//
// One wavefront owns a run of consecutive frames, one frame at a time: it transforms the C
// channels one after the other (12 MFMA + ~90 VALU wave-instructions each; the samples two
// transforms ahead in flight), folds conj(w_c) X_c into four complex accumulators per lane (the
// lane's bins are fixed, its weights come from an LDS table as one base address +
// immediates), adds the odd family X[16 + 32 q] of all channels from ONE extra tile per frame,
// scales the frame's spectrum by a power of two into the fp16 operand range,
// inverse-transforms (12 MFMA), windows, and completes one block of hop output samples per
// frame from its own registers (hop = n_fft / 2: see the kernel).  No frame slots, no
// workgroup barrier, no overlap-add loop; ~128 VGPRs: four waves per SIMD where the butterfly
// kernel ran two.  Other hops keep pass2.hip.
#include "common.h"
#include "fft512.h"
#include "mcdft.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace setk {

// raw frame samples in the operand layout of mcdft.h (pass1_mc.hip)
template <class FloatPtr>
SETK_DEV void load_raw_mc2(float (&v)[8], FloatPtr x, int n_samp, int s, int lane, bool valid) {
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

#ifndef SETK_P2MC_THREADS
#define SETK_P2MC_THREADS 512
#endif
#ifndef SETK_P2MC_WAVES_PER_SIMD
#define SETK_P2MC_WAVES_PER_SIMD 4
#endif
#ifndef SETK_P2MC_GROUP
#define SETK_P2MC_GROUP 2
#endif
// forward operand tiles in LDS (8 x ds_read_b128 per transform) instead of 32 registers
#ifndef SETK_P2MC_KLDS
#define SETK_P2MC_KLDS 1
#endif
// ... and the window rows and twiddles too (5 more reads per transform, 20 more registers free)
#ifndef SETK_P2MC_WLDS
#define SETK_P2MC_WLDS 1
#endif
// streaming (nontemporal) hint on the loads whose data is not read again (see load_full)
#ifndef SETK_P2MC_NT
#define SETK_P2MC_NT 1
#endif
// samples requested TWO transforms ahead of their use (groups of two frames only): while frame
// (t0, c) is transformed the whole frame (t0, c + 1) is in flight, and the second half of
// (t0 + 1, c) landed during the previous transform -- 12 registers in flight instead of 8
#ifndef SETK_P2MC_PF2
#define SETK_P2MC_PF2 0
#endif
// 16-bit PCM form: ONE 1024-thread workgroup per CU (the tables once instead of twice) whose waves
// carry the group-boundary half frame of every channel through LDS as packed int16 (8 bytes per
// lane and channel = 4 KB per wave) instead of re-reading it from HBM at the next group -- the
// re-read was 1/3 of the kernel's audio traffic (counter traffic 1.35 x algorithmic).
#ifndef SETK_P2MC_PCM_THREADS
#define SETK_P2MC_PCM_THREADS 1024
#endif
#ifndef SETK_P2MC_PCM_CARRY
#define SETK_P2MC_PCM_CARRY 1
#endif
// float32 form: the same carry for the first SETK_P2MC_F32_CARRY_CH channels only (16 bytes per
// lane and channel: four channels are the 4 KB per wave that fit), 1024-thread workgroups; 0 = the
// two-workgroup form without a carry -- the default: with 4 the 8-channel kernel holds two load
// paths (24 spilled registers) and measured 6.6 % SLOWER in stage 3 at MORE traffic (1.37 x against
// 1.31 x; profiles/rejected/round6_pass2_f32_partial_carry_ab.txt)
#ifndef SETK_P2MC_F32_CARRY_CH
#define SETK_P2MC_F32_CARRY_CH 0
#endif
struct False { static constexpr bool value = false; };
struct True { static constexpr bool value = true; };
constexpr int kP2McThreads = SETK_P2MC_THREADS;
constexpr int kP2McPcmThreads = SETK_P2MC_PCM_THREADS;
constexpr bool kP2McPcmCarry = SETK_P2MC_PCM_CARRY != 0;
constexpr int kP2McF32CarryCh = SETK_P2MC_F32_CARRY_CH;
constexpr int p2mc_threads(bool pcm) { return pcm ? kP2McPcmThreads : (kP2McF32CarryCh > 0 ? 1024 : kP2McThreads); }
// channels whose group-boundary half frame a wave carries through LDS, and the bytes per lane
constexpr int p2mc_carry_ch(int C, bool pcm) {
    return pcm ? (kP2McPcmCarry ? C : 0) : (C < kP2McF32CarryCh ? C : kP2McF32CarryCh);
}
constexpr int p2mc_carry_bytes(bool pcm) { return pcm ? 8 : 16; }

constexpr int kP2McTiles = SETK_P2MC_KLDS ? (SETK_P2MC_WLDS ? 25 : 20) : 12;  // BR_H .. IT_L (10 tiles, contiguous words) + OT_H, OT_L + the forward's 8
// LDS plan (bytes): wtab C * 257 * 8 | operand tiles 25 * 1024 | synthesis rows
// 2048 | a16 scratch NW * 8 * kOddPitch * 4 | yodd NW * 16 * 4 | red 64
// | PCM carry NW * C * 64 * 8
size_t pass2_mc_lds_bytes(int C, bool pcm) {
    const size_t nw = p2mc_threads(pcm) / 64;
    const size_t wt = ((size_t)C * kBins * sizeof(cf) + 15) & ~(size_t)15;
    return wt + kP2McTiles * 1024 + 2048 + nw * SETK_P2MC_GROUP * 8 * mc::kOddPitch * sizeof(float) +
           nw * 16 * sizeof(float) + 64 + nw * p2mc_carry_ch(C, pcm) * 64 * p2mc_carry_bytes(pcm);
}

// sum over the first 8 lanes of every 16-lane row, result in lanes 0..7 of the row
SETK_DEV float row8_sum(float x) {
    int v = __builtin_bit_cast(int, x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));   // j ^ 1
    v = __builtin_bit_cast(int, x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));   // j ^ 2
    v = __builtin_bit_cast(int, x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));  // j ^ 7
    return x;
}
// max over the 64 lanes (values >= 0), wave-uniform result
SETK_DEV float wave_max_nonneg(float x) {
    int v = __builtin_bit_cast(int, x);
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true)));
    v = __builtin_bit_cast(int, x);
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true)));
    v = __builtin_bit_cast(int, x);
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true)));
    v = __builtin_bit_cast(int, x);
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true)));
    const unsigned b = __builtin_bit_cast(unsigned, x);
    unsigned m = __builtin_amdgcn_readlane(b, 0);
    const unsigned m1 = __builtin_amdgcn_readlane(b, 16), m2 = __builtin_amdgcn_readlane(b, 32),
                   m3 = __builtin_amdgcn_readlane(b, 48);
    m = m > m1 ? m : m1;  // non-negative floats order like their bit patterns
    m = m > m2 ? m : m2;
    m = m > m3 ? m : m3;
    return __builtin_bit_cast(float, m);
}

// hop = n_fft / 2 (the CLI's default geometry): every output block of `hop` samples is the sum
// of exactly two frame halves, so a wavefront that walks CONSECUTIVE frames carries the second
// half of its last frame in four registers per lane and finishes a block per frame by itself --
// no frame slots in LDS, no workgroup barrier, no overlap-add loop.  1 / sum(window^2) is folded
// into the synthesis rows (mc_syn); blocks with a single contribution (the first and the last
// one of an utterance, emitted only when center = False) take the per-lane corrections mc_edge.
// A workgroup shares the weight table and the once-per-frame operand tiles; its waves split the
// item's frame range and each recomputes one frame ahead of its sub-range for the carry.
// PCM: UttDesc::audio is planar 16-bit PCM (kAudioPcm16) -- sign-extending 2-byte loads, one
// conversion per sample, and 2^-15 (read_wav's int16 / 32768) folded into the window rows.
template <int C, bool PCM = false>
__global__ __launch_bounds__(p2mc_threads(PCM), SETK_P2MC_WAVES_PER_SIMD) void beamform_istft_mc_kernel(Pass2Args a) {
    constexpr int NT = p2mc_threads(PCM);
    constexpr int NW = NT / 64;
    constexpr int F = kBins;
    constexpr int CC = p2mc_carry_ch(C, PCM);   // channels 0 .. CC - 1 are carried
    constexpr bool CARRY = CC > 0;
    typedef typename std::conditional<PCM, uint2, mc::f4>::type carry_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* p = smem;
    cf* wtab = reinterpret_cast<cf*>(p);  // [C][257]
    p += ((size_t)C * F * sizeof(cf) + 15) & ~(size_t)15;
    mc::u4* tiles = reinterpret_cast<mc::u4*>(p);  // [12][64]: BR_H BR_L BI_H BI_L G0_H G0_L G1_H G1_L IT_H IT_L OT_H OT_L
    p += kP2McTiles * 1024;
    mc::f4* synr = reinterpret_cast<mc::f4*>(p);  // [2][64] float4: synthesis rows 0..3 / 4..7 of a lane
    p += 2048;
    float* a16s = reinterpret_cast<float*>(p);  // [NW][R][8][kOddPitch]
    p += (size_t)NW * SETK_P2MC_GROUP * 8 * mc::kOddPitch * sizeof(float);
    float* yodd_s = reinterpret_cast<float*>(p);  // [NW][16]
    p += NW * 16 * sizeof(float);
    float* red = reinterpret_cast<float*>(p);
    p += 64;
    carry_t* carry_s = reinterpret_cast<carry_t*>(p);  // [NW][CC][64]: packed int16 x 4 / float x 4

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples;
    const int T = ud.num_frames;
    const int hop = kNfft / 2;
    const bool post_mask = (a.flags & 0x4) != 0;
    const bool clamp = (a.flags & 0x2) != 0;

    // The fp16 operand splits of the forward transform hold window x sample x 2^10 and want
    // |sample| <= 1 (65504 is the end of fp16).  The utterance's max |x| is known (pass 1 reduced
    // it into norm_bits before this kernel started): samples above 1 -- float wave files, int16
    // ranges handed over as floats (WaveReader(normalize=False)), C-API callers -- are brought
    // into range by the power of two 2^-e in the window rows and taken out again by 2^e in the
    // synthesis rows.  e = 0 (nothing changes, bit for bit) whenever max |x| <= 1.  The
    // beamformer is linear, every other stage of this kernel is scale free (the inverse
    // transform normalises each frame's spectrum by its own power of two).  16-bit PCM arrives
    // as integers: 2^-15 on the way in, nothing on the way out.
    float in_sc = PCM ? 3.0517578125e-05f : 1.f, out_sc = 1.f;
    if (a.norm_bits) {
        const unsigned nb = a.norm_bits[wi.utt];  // max |x| (natural units) as float bits
        int e = (int)((nb >> 23) & 0xff) - 126;   // max |x| < 2^e
        e = (__builtin_bit_cast(float, nb) <= 1.f) ? 0 : (e > 100 ? 100 : e);
        in_sc *= __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
        out_sc = __builtin_bit_cast(float, (unsigned)(127 + e) << 23);
    }
    {
        const cf* wsrc = reinterpret_cast<const cf*>(a.weight) + (size_t)wi.utt * C * kBinsPad;
        for (int i = tid; i < C * F; i += NT) {
            const int c = i / F, f = i - c * F;
            wtab[i] = wsrc[c * kBinsPad + f];
        }
    }
    mc::stage_tiles(tiles, a.mc_tab, mc::kW_BR_H, 10, tid, NT);
    mc::stage_tiles(tiles + 10 * 64, a.mc_tab, mc::kW_OT_H, 2, tid, NT);
#if SETK_P2MC_KLDS
    mc::stage_tiles(tiles + 12 * 64, a.mc_tab, mc::kW_MC_H, 8, tid, NT);
#if SETK_P2MC_WLDS
    for (int i = tid; i < 128; i += NT) {
        const int l = i & 63, t4 = i >> 6;
        const mc::f4 w = {a.mc_win[(4 * t4 + 0) * 64 + l] * in_sc, a.mc_win[(4 * t4 + 1) * 64 + l] * in_sc,
                          a.mc_win[(4 * t4 + 2) * 64 + l] * in_sc, a.mc_win[(4 * t4 + 3) * 64 + l] * in_sc};
        tiles[20 * 64 + i] = __builtin_bit_cast(mc::u4, w);
    }
    mc::stage_tiles(tiles + 22 * 64, a.mc_tab, mc::kW_TR, 3, tid, NT);
#endif
#endif
    for (int i = tid; i < 128; i += NT) {
        const int l = i & 63, hf = i >> 6;
        synr[i] = (mc::f4){a.mc_syn[(4 * hf + 0) * 64 + l] * out_sc, a.mc_syn[(4 * hf + 1) * 64 + l] * out_sc,
                           a.mc_syn[(4 * hf + 2) * 64 + l] * out_sc, a.mc_syn[(4 * hf + 3) * 64 + l] * out_sc};
    }
#if SETK_P2MC_KLDS && SETK_P2MC_WLDS
    struct { float tr[4], ti[4]; } K;  // (the inverse's conjugate twiddles: re-read there)
#elif SETK_P2MC_KLDS
    struct { float tr[4], ti[4], tri[4]; } K;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        K.tr[r] = mc::tab_f(a.mc_tab, mc::kW_TR + r, lane);
        K.ti[r] = mc::tab_f(a.mc_tab, mc::kW_TI + r, lane);
        K.tri[r] = mc::tab_f(a.mc_tab, mc::kW_TRI + r, lane);
    }
#else
    mc::Fwd K;
    mc::load_fwd(K, a.mc_tab, lane);
#endif
#if !(SETK_P2MC_KLDS && SETK_P2MC_WLDS)
    float win[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) win[e] = gptr(a.mc_win)[e * 64 + lane] * in_sc;
#endif
    float* yoddw = yodd_s + wave * 16;
    const int lane_bin = mc::bin_of(c16, g, 0);
    const cf* wl = wtab + lane_bin;                       // + c * F + 32 r
    const int jo = c16 < C ? c16 : 0;                     // odd-family tile: column = channel
    const cf* wo = wtab + jo * F + 16 + 64 * g;           // w_j[16 + 32 (2 g)], [+ 32] the next
    const bool odd_on = c16 < C;
    const int lane_n = 64 * g + c16;                      // sample 16 (4 g + r) + n2 = lane_n + 16 r
    float omax = 0.f;
    __syncthreads();  // tables ready (the only workgroup barrier before the epilogue)

    // this wave's frames [ta, tb); one frame ahead of ta is recomputed for its second half
    const int per = (wi.t1 - wi.t0 + NW - 1) / NW;
    const int ta = wi.t0 + wave * per;
    const int tb = min(ta + per, wi.t1);
    const int tw = ta > 0 ? ta - 1 : ta;  // first frame computed

    // Channel-major over groups of R consecutive frames: with hop = n_fft / 2 frame t + 1 shares
    // its first half with frame t -- the same lane's registers (mc::sample_of) -- so inside a
    // group a transform loads four new samples per lane, not eight, and the weights of a channel
    // are read once per group.  The R spectra of the group accumulate in registers.
    constexpr int R = SETK_P2MC_GROUP;
    float* a16g = a16s + wave * R * 8 * mc::kOddPitch;  // [R][8 channels][kOddPitch]
    float carry[4] = {0.f, 0.f, 0.f, 0.f};
    // loaders: a whole frame (8 registers) / the second half of a frame (registers 4..7).
    // EDGE: some sample of the group lies outside the signal (numpy "reflect" padding) -- the
    // first and the last group of an utterance; frames past the last one repeat it (computed
    // to keep the group uniform, never emitted).
    // channel c of the utterance: float32 [C][N] or int16 [C][ch_stride] (the conversion is the
    // implicit one of `float = short`: global_load_sshort + v_cvt_f32_i32)
    auto chan = [&](int c) {
        if constexpr (PCM) return (gcshort_p)gptr(ud.audio) + (size_t)c * ud.ch_stride;
        else return gptr(ud.audio) + (size_t)c * n_samp;
    };
    // CARRY: the lane's own four samples of the half frame a group ends with, per channel -- a
    // thread reads back exactly what it wrote (program order suffices, no barrier)
    carry_t* carry_l = carry_s + (size_t)wave * CC * 64 + lane;  // + 64 c
    auto load_full = [&](float (&v)[8], int t, int c, auto edge, bool first) __attribute__((always_inline)) {
        (void)first;
        if constexpr (CARRY) {
            // first half: what this lane parked at the end of the previous group (or the prefill
            // before the first); second half: the only samples of the frame not seen yet
            if (CC == C || c < CC) {
                if constexpr (PCM) {
                    const uint2 pk = carry_l[64 * c];
                    v[0] = (float)(short)(pk.x & 0xffff);
                    v[1] = (float)((int)pk.x >> 16);
                    v[2] = (float)(short)(pk.y & 0xffff);
                    v[3] = (float)((int)pk.y >> 16);
                } else {
                    const mc::f4 pk = carry_l[64 * c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pk[e];
                }
#ifdef SETK_P2MC_ABL_L2  // ablation: every wave reads the same 1 MB (L2 resident) -- wrong results
                const auto x = (gcshort_p)gptr(a.utts[0].audio) + (size_t)c * a.utts[0].ch_stride;
                const int s0 = (min(t, T - 1) & 127) * hop + 4096 + 256, o = 64 * g + c16;
#else
                const auto x = chan(c);
                const int s0 = min(t, T - 1) * hop - a.g.pad + 256, o = 64 * g + c16;
#endif
                if (!decltype(edge)::value) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 + e] = x[s0 + o + 16 * e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 + e] = x[reflect_index(s0 + o + 16 * e, n_samp)];
                }
                return;
            }
        }
#ifdef SETK_P2MC_ABL_L2  // ablation: every wave reads the same 1 MB (L2 resident) -- wrong results
        gcfloat_p x = gptr(a.utts[0].audio) + (size_t)c * n_samp;
        (void)chan;
        const int s0 = (min(t, T - 1) & 127) * hop + 4096, o = 64 * g + c16;
#else
        const auto x = chan(c);
        const int s0 = min(t, T - 1) * hop - a.g.pad, o = 64 * g + c16;
#endif
        if (!decltype(edge)::value) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#if SETK_P2MC_NT
                // (float32 samples only: on the 2-byte loads of the PCM form the hint measured
                //  0.4 - 2.5 % slower, profiles/round5_pass2_nt_pf2_ab.txt)
                // both halves of a group's first frame are read for the last time here (the first
                // one is the re-read of what the previous group fetched as ITS last half): a
                // streaming hint keeps them from pushing the halves that WILL be read again --
                // load_half's -- out of the XCD's L2
                if constexpr (PCM) {
                    v[e] = x[s0 + o + 16 * e];
                    v[4 + e] = x[s0 + o + 256 + 16 * e];
                } else {
                    v[e] = __builtin_nontemporal_load(&x[s0 + o + 16 * e]);
                    v[4 + e] = __builtin_nontemporal_load(&x[s0 + o + 256 + 16 * e]);
                }
#else
                v[e] = x[s0 + o + 16 * e];
                v[4 + e] = x[s0 + o + 256 + 16 * e];
#endif
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = x[reflect_index(s0 + o + 16 * e, n_samp)];
                v[4 + e] = x[reflect_index(s0 + o + 256 + 16 * e, n_samp)];
            }
        }
    };
    // the same half as raw 16-bit samples (CARRY): converted and packed for the carry where consumed
    auto load_half_raw = [&](int (&r)[4], int t, int c, auto edge) __attribute__((always_inline)) {
        if constexpr (PCM) {
#ifdef SETK_P2MC_ABL_L2
            const auto x = (gcshort_p)gptr(a.utts[0].audio) + (size_t)c * a.utts[0].ch_stride;
            const int s0 = (min(t, T - 1) & 127) * hop + 4096 + 512, o = 64 * g + c16;
#else
            const auto x = chan(c);
            const int s0 = min(t, T - 1) * hop - a.g.pad + 256, o = 64 * g + c16;
#endif
            if (!decltype(edge)::value) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = x[s0 + o + 16 * e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = x[reflect_index(s0 + o + 16 * e, n_samp)];
            }
        }
    };
    auto load_half = [&](float (&v)[8], int t, int c, auto edge) __attribute__((always_inline)) {
#ifdef SETK_P2MC_ABL_REHALF  // ablation: re-read the half just loaded (cache hit) -- wrong results
        const auto x = chan(c);
        const int s0 = min(t, T - 1) * hop - a.g.pad, o = 64 * g + c16;
#elif defined(SETK_P2MC_ABL_L2)
        gcfloat_p x = gptr(a.utts[0].audio) + (size_t)c * n_samp;
        const int s0 = (min(t, T - 1) & 127) * hop + 4096 + 256, o = 64 * g + c16;
#else
        const auto x = chan(c);
        const int s0 = min(t, T - 1) * hop - a.g.pad + 256, o = 64 * g + c16;
#endif
        if (!decltype(edge)::value) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 + e] = x[s0 + o + 16 * e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 + e] = x[reflect_index(s0 + o + 16 * e, n_samp)];
        }
    };
#if SETK_P2MC_PF2
    static_assert(SETK_P2MC_GROUP == 2, "the two-ahead prefetch is written for groups of two frames");
    float nxth[4];  // second half of frame t0 + 1 of the NEXT channel
    auto load_half4 = [&](float (&v)[4], int t, int c, auto edge) __attribute__((always_inline)) {
        const auto x = chan(c);
        const int s0 = min(t, T - 1) * hop - a.g.pad + 256, o = 64 * g + c16;
        if (!decltype(edge)::value) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = x[s0 + o + 16 * e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = x[reflect_index(s0 + o + 16 * e, n_samp)];
        }
    };
#endif
    mc::f4 yr[R], yi[R];
    // the R transforms of every channel of one group; `nxt` arrives holding frame (t0, channel 0)
    // and leaves holding frame (t0 + R, channel 0)
    float nxt[8];
    int nraw[4];  // CARRY: the half frame in flight, as loaded
    auto group = [&](int t0, auto edge, bool first_next) __attribute__((always_inline)) {
        (void)first_next;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            yr[k] = (mc::f4){0.f, 0.f, 0.f, 0.f};
            yi[k] = (mc::f4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            asm volatile("" ::: "memory");  // the weights are re-read per group, not kept (64 registers)
            const cf* wc = wl + c * F;
            cf w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = wc[32 * r];
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = nxt[e];
#if SETK_P2MC_PF2
            float xh[4];  // this channel's second half of frame t0 + 1 (requested one channel ago)
#pragma unroll
            for (int e = 0; e < 4; ++e) xh[e] = nxth[e];
#endif
#pragma unroll
            for (int k = 0; k < R; ++k) {
#if SETK_P2MC_PF2
                // two transforms ahead: during (t0, c) the whole frame (t0, c + 1), during
                // (t0 + 1, c) the second half of (t0 + 1, c + 1)
                if (c + 1 < C) {
                    if (k == 0) load_full(nxt, t0, c + 1, edge, true);
                    else load_half4(nxth, t0 + 1, c + 1, edge);
                }
#else
                // what comes next travels while this transform runs: the second half of the
                // next frame of the group, or the first frame of the next channel / group
                if (k + 1 < R) {
                    if constexpr (CARRY && PCM) load_half_raw(nraw, t0 + k + 1, c, edge);
                    else load_half(nxt, t0 + k + 1, c, edge);
                } else if (c + 1 < C) {
                    load_full(nxt, t0, c + 1, edge, first_next);
                }
#endif
                mc::f4 zr, zi, a16;
#if SETK_P2MC_KLDS && SETK_P2MC_WLDS
                {
                    asm volatile("" ::: "memory");
                    const mc::f4 w0 = __builtin_bit_cast(mc::f4, tiles[20 * 64 + lane]);
                    const mc::f4 w1 = __builtin_bit_cast(mc::f4, tiles[21 * 64 + lane]);
                    const mc::f4 q0 = __builtin_bit_cast(mc::f4, tiles[22 * 64 + lane]);
                    const mc::f4 q1 = __builtin_bit_cast(mc::f4, tiles[23 * 64 + lane]);
                    const mc::f4 q2 = __builtin_bit_cast(mc::f4, tiles[24 * 64 + lane]);
                    const float win[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                    const float tr[4] = {q0[0], q0[1], q0[2], q0[3]}, ti[4] = {q1[0], q1[1], q1[2], q1[3]},
                                tri[4] = {q2[0], q2[1], q2[2], q2[3]};
                    mc::forward_t(x, win, [&](int i) { return mc::lds_h8(tiles, 12 + i, lane); }, tr, ti, tri, zr, zi, a16);
                }
#elif SETK_P2MC_KLDS
                mc::forward_t(x, win, [&](int i) { return mc::lds_h8(tiles, 12 + i, lane); }, K.tr, K.ti, K.tri, zr, zi, a16);
#else
                mc::forward(x, win, K, zr, zi, a16);
#endif
                mc::store_a16(a16g + k * 8 * mc::kOddPitch, c, lane, a16);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    yr[k][r] = fmaf(zr[r], w[r].x, fmaf(zi[r], w[r].y, yr[k][r]));
                    yi[k][r] = fmaf(zi[r], w[r].x, fmaf(-zr[r], w[r].y, yi[k][r]));
                }
                if (k + 1 < R) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[e] = x[4 + e];
#if SETK_P2MC_PF2
                        x[4 + e] = xh[e];
#else
                        if constexpr (CARRY && PCM) x[4 + e] = (float)nraw[e];
                        else x[4 + e] = nxt[4 + e];
#endif
                    }
                    if constexpr (CARRY) {
                        // the group's last half frame is the next group's first: R == 2, k == 0
                        static_assert(!CARRY || R == 2, "the LDS carry is written for groups of two frames");
                        if constexpr (PCM) {
                            carry_l[64 * c] = make_uint2(__builtin_amdgcn_perm((unsigned)nraw[1], (unsigned)nraw[0], 0x05040100u),
                                                         __builtin_amdgcn_perm((unsigned)nraw[3], (unsigned)nraw[2], 0x05040100u));
                        } else if (CC == C || c < CC) {
                            carry_l[64 * c] = (mc::f4){nxt[4], nxt[5], nxt[6], nxt[7]};
                        }
                    }
                }
                // one transform at a time: interleaved by the scheduler, the R unrolled
                // transforms keep R working sets alive and spill 150 registers
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto span_is_edge = [&](int t0) {
        const int lo = t0 * hop - a.g.pad, hi = (t0 + R - 1) * hop - a.g.pad + kNfft;
        return lo < 0 || hi > n_samp || t0 + R > T;
    };
    auto request_group = [&](int t0, bool first) __attribute__((always_inline)) {
        if (span_is_edge(t0)) {
            load_full(nxt, t0, 0, True(), first);
#if SETK_P2MC_PF2
            load_half4(nxth, t0 + 1, 0, True());
#endif
        } else {
            load_full(nxt, t0, 0, False(), first);
#if SETK_P2MC_PF2
            load_half4(nxth, t0 + 1, 0, False());
#endif
        }
    };
    if constexpr (CARRY) {
        // the first half of the wave's first frame, every channel: from here on a group reads two
        // half frames per channel from HBM, never three
        if (tw < tb) {
#pragma unroll 1
            for (int c = 0; c < CC; ++c) {
                const auto x = chan(c);
                const int s0 = tw * hop - a.g.pad, o = 64 * g + c16;
                if constexpr (PCM) {
                    int r[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = x[reflect_index(s0 + o + 16 * e, n_samp)];
                    carry_l[64 * c] = make_uint2(__builtin_amdgcn_perm((unsigned)r[1], (unsigned)r[0], 0x05040100u),
                                                 __builtin_amdgcn_perm((unsigned)r[3], (unsigned)r[2], 0x05040100u));
                } else {
                    mc::f4 r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = x[reflect_index(s0 + o + 16 * e, n_samp)];
                    carry_l[64 * c] = r;
                }
            }
        }
    }
    if (tw < tb) request_group(tw, true);
#pragma unroll 1
    for (int t0 = tw; t0 < tb; t0 += R) {
        const int nf = min(R, tb - t0);
        // (the channels 1.. of this group are fetched inside it: from the carry unless it is the
        //  wave's first group)
        if (span_is_edge(t0)) group(t0, True(), t0 == tw);
        else group(t0, False(), t0 == tw);
        if (t0 + R < tb) request_group(t0 + R, false);  // (issued here: its span decides the path)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (k >= nf) break;
            const int t = t0 + k;
            // ---- odd family of all channels: one tile, then the sum over the channel lanes ----
            float yo[4];
            {
                asm volatile("" ::: "memory");  // the once-per-frame tiles are re-read, not kept
                const mc::f4 d = mc::odd_tile(a16g + k * 8 * mc::kOddPitch, mc::lds_h8(tiles, 10, lane),
                                              mc::lds_h8(tiles, 11, lane), lane, C);
                const cf w0 = wo[0], w1 = wo[32];
                yo[0] = odd_on ? fmaf(d[0], w0.x, d[1] * w0.y) : 0.f;
                yo[1] = odd_on ? fmaf(d[1], w0.x, -d[0] * w0.y) : 0.f;
                yo[2] = odd_on ? fmaf(d[2], w1.x, d[3] * w1.y) : 0.f;
                yo[3] = odd_on ? fmaf(d[3], w1.x, -d[2] * w1.y) : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) yo[i] = row8_sum(yo[i]);
            }
            mc::f4 fr = yr[k], fi = yi[k];
            // ---- optional post-mask ----
            if (post_mask) {
                const float* mrow = ud.mask_s + (size_t)t * F;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float m = mrow[lane_bin + 32 * r];
                    if (clamp) m = fminf(m, 1.f);
                    fr[r] *= m;
                    fi[r] *= m;
                }
                float m0 = mrow[16 + 64 * g], m1 = mrow[48 + 64 * g];
                if (clamp) { m0 = fminf(m0, 1.f); m1 = fminf(m1, 1.f); }
                yo[0] *= m0;
                yo[1] *= m0;
                yo[2] *= m1;
                yo[3] *= m1;
            }
            // only Re Y[0], Re Y[256] reach the inverse (numpy irfft drops their imaginary parts)
            fi[0] = (lane == 0) ? 0.f : fi[0];
            fi[3] = (lane == 32) ? 0.f : fi[3];
            // ---- power-of-two scale of the frame into the fp16 operand range: max < 2^11 ----
            float mxl = fmaxf(fmaxf(fabsf(yo[0]), fabsf(yo[1])), fmaxf(fabsf(yo[2]), fabsf(yo[3])));
#pragma unroll
            for (int r = 0; r < 4; ++r) mxl = fmaxf(mxl, fmaxf(fabsf(fr[r]), fabsf(fi[r])));
            const float mxw = wave_max_nonneg(mxl);
            int ex = (int)((__builtin_bit_cast(unsigned, mxw) >> 23) & 0xff);  // mxw < 2^(ex - 126)
            ex = ex < 16 ? 16 : (ex > 250 ? 250 : ex);                         // (zero / tiny / huge frames)
            const float sc = __builtin_bit_cast(float, (unsigned)(264 - ex) << 23);   // 2^(137 - ex)
            const float isc = __builtin_bit_cast(float, (unsigned)(ex - 10) << 23);
            // ---- E16 of the odd family: its tile takes frames as rows; this frame is row 0 ----
            if (c16 == 0) *reinterpret_cast<mc::f4*>(yoddw + 4 * g) = (mc::f4){yo[0] * sc, yo[1] * sc, yo[2] * sc, yo[3] * sc};
            float e16;
            {
                float v[8];
                const mc::f4 v0 = *reinterpret_cast<const mc::f4*>(yoddw + 8 * (g & 1));
                const mc::f4 v1 = *reinterpret_cast<const mc::f4*>(yoddw + 8 * (g & 1) + 4);
                const float rowsel = c16 == 0 ? 1.f : 0.f;  // rows 1..15 of the tile are unused
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v0[e] * rowsel;
                    v[4 + e] = v1[e] * rowsel;
                }
                const mc::f4 d = mc::inv_odd_tile(v, mc::lds_h8(tiles, 8, lane), mc::lds_h8(tiles, 9, lane), lane);
                e16 = d[0];  // lanes g == 0: E16[n2 = l % 16] of row 0
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                fr[r] *= sc;
                fi[r] *= sc;
            }
            float bmid[8];
#if SETK_P2MC_KLDS && SETK_P2MC_WLDS
            {
                const mc::f4 tt0 = __builtin_bit_cast(mc::f4, tiles[22 * 64 + lane]);
                const mc::f4 tt1 = __builtin_bit_cast(mc::f4, tiles[23 * 64 + lane]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    K.tr[r] = tt0[r];
                    K.ti[r] = tt1[r];
                }
            }
#endif
            mc::inverse_a(fr, fi, e16, mc::lds_h8(tiles, 0, lane), mc::lds_h8(tiles, 1, lane),
                          mc::lds_h8(tiles, 2, lane), mc::lds_h8(tiles, 3, lane), K.tr, K.ti, bmid, lane);
            asm volatile("" ::: "memory");
            mc::f4 y0, y1;
            mc::inverse_b(bmid, mc::lds_h8(tiles, 4, lane), mc::lds_h8(tiles, 5, lane),
                          mc::lds_h8(tiles, 6, lane), mc::lds_h8(tiles, 7, lane), y0, y1);
            // ---- block t = first half of frame t + second half of frame t - 1 (the carry); the
            //      rows hold window / 512 / sum(window^2) x the back-scale of the spectra ----
            const mc::f4 s0 = synr[lane], s1 = synr[64 + lane];
            float blk[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) blk[r] = fmaf(y0[r], s0[r] * isc, carry[r]);
            if (t == 0) {  // no frame before the first: a single contribution (center = False only)
#pragma unroll
                for (int r = 0; r < 4; ++r) blk[r] *= gptr(a.mc_edge)[r * 64 + lane];
            }
            if (t >= ta) {
                const int o = t * hop - a.g.pad + lane_n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int oo = o + 16 * r;
                    if (oo >= 0 && oo < ud.out_len) {
                        ud.wave_f32[oo] = blk[r];
                        omax = fmaxf(omax, fabsf(blk[r]));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) carry[r] = y1[r] * (s1[r] * isc);
        }
    }
    // ---- the block after the last frame: its second half alone (center = False only) ----
    if (tb == T && ta < tb) {
        const int o = T * hop - a.g.pad + lane_n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oo = o + 16 * r;
            const float v = carry[r] * gptr(a.mc_edge)[(4 + r) * 64 + lane];
            if (oo >= 0 && oo < ud.out_len) {
                ud.wave_f32[oo] = v;
                omax = fmaxf(omax, fabsf(v));
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
    if ((tid & 63) == 0) red[tid >> 6] = omax;
    __syncthreads();
    if (tid == 0) {
        float m = red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, red[w]);
        atomicMax(a.outmax_bits + wi.utt, __float_as_uint(m));
    }
}

// workgroups of this kernel a CU holds (registers and LDS), for the work-list cut of capi.hip
int pass2_mc_wgs_per_cu(int C, bool pcm) {
    const int by_waves = SETK_P2MC_WAVES_PER_SIMD * 4 / (p2mc_threads(pcm) / 64);
    const int by_lds = (int)((160u << 10) / pass2_mc_lds_bytes(C, pcm));
    return by_waves < by_lds ? (by_waves > 0 ? by_waves : 1) : (by_lds > 0 ? by_lds : 1);
}

template <int C, bool PCM = false>
static hipError_t launch_pass2_mc_t(const Pass2Args& a, int n_items, hipStream_t s) {
    const size_t lds = pass2_mc_lds_bytes(C, PCM);
    auto k = beamform_istft_mc_kernel<C, PCM>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_items), dim3(p2mc_threads(PCM)), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pass2_mc(int C, const Pass2Args& a, int n_items, hipStream_t s, bool pcm16) {
#define SETK_CASE(c) \
    case c: return pcm16 ? launch_pass2_mc_t<c, true>(a, n_items, s) : launch_pass2_mc_t<c>(a, n_items, s);
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

}  // namespace setk
