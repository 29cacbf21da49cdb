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
SETK_DEV void load_frame(cf (&v)[16], gcfloat_p x, int n_samp, int s,
                         int la, const float* win, bool valid, float& mx) {
    const float2* w2 = reinterpret_cast<const float2*>(win);
    if (!valid) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = make_float2(0.f, 0.f);
        return;
    }
    const bool interior = (s >= 0) && (s + kNfft <= n_samp) &&
                          ((((uintptr_t)(x + s)) & 7) == 0);
    if (interior) {
        gcfloat2_p p = (gcfloat2_p)(x + s);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = la + 16 * j;
            const v2f d = p[n];
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

// raw (un-windowed) frame points: v[j] = (x[s+2n], x[s+2n+1]), n = la + 16 j
SETK_DEV void load_raw(cf (&v)[16], gcfloat_p x, int n_samp, int s, int la,
                       bool valid) {
    if (!valid) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = make_float2(0.f, 0.f);
        return;
    }
    const bool interior = (s >= 0) && (s + kNfft <= n_samp) &&
                          ((((uintptr_t)(x + s)) & 7) == 0);
    if (interior) {
        gcfloat2_p p = (gcfloat2_p)(x + s);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const v2f d = p[la + 16 * j];
            v[j] = make_float2(d.x, d.y);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = la + 16 * j;
            v[j] = make_float2(x[reflect_index(s + 2 * n, n_samp)],
                               x[reflect_index(s + 2 * n + 1, n_samp)]);
        }
    }
}

// One quad-row: forward real transform of a frame in three LDS-separated stages
// (no workgroup barrier: the 16 lanes share a wavefront, and the LDS operations
// of a wavefront complete in order).  After stage 3 the slot holds X[0..255]
// and *nyq = X[256].
SETK_DEV void qr_stage1(cf (&v)[16], cf* slot, const cf* tw, int la, int ls) {
    fft256_stage_a<-1>(v, slot, tw, la, ls);
}
// lane la <- value of lane (16 - la) & 15 of the same quad-row (a DPP row):
// row_mirror (la -> 15 - la) followed by row_ror:1
SETK_DEV float qr_partner(float x) {
    int v = __builtin_bit_cast(int, x);
    v = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);  // row_mirror
    v = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, true);  // row_ror:1
    return __builtin_bit_cast(float, v);
}
// Stage 2+3: second radix-16 and the Hermitian split done in registers: lane la
// owns Z[la + 16 kb]; the mirror bin 256 - k of k = la + 16 m lives in lane
// (16 - la) & 15, register 15 - m (lane 0: its own register 16 - m), fetched with
// two DPP moves instead of an LDS round trip.  Writes X[k], X[256-k], m < 8.
template <bool PAD = false>
SETK_DEV void qr_stage23(cf* slot, float* nyq, const cf* tw5, int la, int ls) {
    cf v[16];
    if (PAD)
        fft256_stage_b_pad<-1>(v, slot, la);
    else
        fft256_stage_b<-1>(v, slot, la, ls);
    const bool lane0 = (la == 0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int k = la + 16 * m;
        const cf Zk = v[dft16_pos(m)];
        const cf src = v[dft16_pos(15 - m)];
        cf Zm = make_float2(qr_partner(src.x), qr_partner(src.y));
        const cf own = v[dft16_pos((16 - m) & 15)];
        Zm.x = lane0 ? own.x : Zm.x;
        Zm.y = lane0 ? own.y : Zm.y;
        cf Xk, Xm;
        rfft_split(Zk, Zm, tw5[k], Xk, Xm);
        if (m == 0) {
            // lane 0: k = 0 pairs with itself (Z[256] == Z[0]): X[0], X[256]; and
            // the self-paired bin 128 = conj(Z[128])
            const cf Z128 = v[dft16_pos(8)];
            if (lane0) {
                slot[0] = make_float2(Xk.x, 0.f);
                *nyq = Xm.x;
                slot[128] = make_float2(2.f * Z128.x, -2.f * Z128.y);
            } else {
                slot[k] = Xk;
                slot[256 - k] = Xm;
            }
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

// frames per tile: 32/C transforms fill the 32 quad-rows; capped at 8 (small C
// then uses several producer sets), 4 for C = 4 so that two mask planes fit LDS
__host__ __device__ constexpr int tile_frames(int c) {
    return c == 4 ? 4 : ((32 / c) < 8 ? (32 / c) : 8);
}

// NQ = number of threads sharing one bin's Hermitian pairs (workgroup = 256*NQ
// threads = 16*NQ quad-rows).  NQ = 4: 1024 threads, <= 128 VGPRs, 4 waves/SIMD.
template <int C, bool DUMP, int NQ>
__global__ __launch_bounds__(256 * NQ, NQ) void stft_covar_kernel(Pass1Args a) {
    constexpr int NT = 256 * NQ;
    constexpr int TB = tile_frames(C);  // frames per tile
    constexpr int NF = TB * C;  // transforms per tile (<= 32)
    constexpr int NP = npairs(C);
    constexpr int NPQ = (NP + NQ - 1) / NQ;    // accumulator pairs per thread (max)
    constexpr int NS = (16 * NQ) / NF;         // producer sets (quad-row groups of NF)
    constexpr int F = kBins, FP = kBinsPad;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* xt0 = reinterpret_cast<cf*>(smem);            // [2][NF][256] double-buffered tile
    cf* tw = xt0 + 2 * NF * 256;                      // [16][16]
    cf* tw5 = tw + 256;                               // [128]
    float* win = reinterpret_cast<float*>(tw5 + 128);  // [512]
    float* xn0 = win + kNfft;                         // [2][32] nyquist bins (real)
    float* red = xn0 + 64;                            // [16]
    constexpr int MK = TB * F;                        // mask floats per tile
    float* mks0 = red + 16;                           // [2][MK] speech mask rows of a tile
    float* mkn0 = mks0 + 2 * MK;                      // [2][MK] interferer mask rows

    const int tid = threadIdx.x;
    const int la = tid & 15, grp = tid >> 4;
    const int f = tid & 255, q = tid >> 8;
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples;
    const int T = ud.num_frames;

    if (tid < 256) tw[tid] = a.tw256[tid];
    if (tid < 128) tw5[tid] = a.tw512[tid];
    if (tid < kNfft) win[tid] = a.window[tid];
    if (NT < kNfft && tid + NT < kNfft) win[tid + NT] = a.window[tid + NT];

    cf acc_s[NPQ], acc_n[NPQ];
    float sum_s = 0.f, sum_n = 0.f;
#pragma unroll
    for (int e = 0; e < NPQ; ++e) {
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
    // this quad-row's role: transform `my_i` of the tiles whose index == my_set (mod NS)
    const int my_set = grp / NF;
    const int my_i = grp - my_set * NF;
    const int my_tt = my_i / C, my_c = my_i - my_tt * C;
    const bool producer = my_set < NS;

    // Software pipeline over tiles (double-buffered LDS tile, ONE barrier per
    // tile): while tile k is consumed from buffer k&1, tile k+1 is transformed
    // into the other buffer by producer set (k+1) % NS, its three LDS-separated
    // stages interleaved with the consume frames of the same waves.
    gcfloat_p my_audio = gptr(ud.audio) + (size_t)my_c * n_samp;
    const float2* w2 = reinterpret_cast<const float2*>(win);
    __syncthreads();  // tables ready

    // register prefetch, one production / one tile ahead
    cf raw[16];
    auto fetch_raw = [&](int tb_tile) {
        const int t = tb_tile + my_tt;
        load_raw(raw, my_audio, n_samp, t * a.g.hop - a.g.pad, la, t < wi.t1);
    };
    auto stage1 = [&](cf* slot, int tb_next_own) {
        // window the prefetched frame, start the next fetch, radix-16, transpose
        cf v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float2 d = raw[j];
            const float2 w = w2[la + 16 * j];
            mx = fmaxf(mx, fmaxf(fabsf(d.x), fabsf(d.y)));
            v[j] = make_float2(d.x * w.x, d.y * w.y);
        }
        if (tb_next_own < wi.t1) fetch_raw(tb_next_own);
        qr_stage1(v, slot, tw, la, la ^ (grp & 1));
    };
    // mask rows [tile][F] are contiguous in memory: staged flat, coalesced,
    // one tile ahead, through registers into LDS (clamp applied here)
    constexpr int MKL = (MK + NT - 1) / NT;  // loads per thread
    float mreg_s[MKL], mreg_n[MKL];
    auto fetch_masks = [&](int tb_tile) {
        const int nvalid = min(TB, wi.t1 - tb_tile) * F;
        gcfloat_p src_s = gptr(ud.mask_s) + (size_t)tb_tile * F;
        gcfloat_p src_n = gptr(ud.mask_n) + (size_t)tb_tile * F;
#pragma unroll
        for (int r = 0; r < MKL; ++r) {
            const int i = tid + r * NT;
            float vs = 0.f, vn = 0.f;
#if !(defined(SETK_ABL) && SETK_ABL == 3)
            if (i < nvalid) {
                vs = src_s[i];
                if (has_mn) vn = src_n[i];
            }
#endif
            mreg_s[r] = vs;
            mreg_n[r] = vn;
        }
    };
    auto stash_masks = [&](int b) {
#pragma unroll
        for (int r = 0; r < MKL; ++r) {
            const int i = tid + r * NT;
            if (i < MK) {
                mks0[b * MK + i] = clamp ? fminf(mreg_s[r], 1.f) : mreg_s[r];
                if (has_mn) mkn0[b * MK + i] = mreg_n[r];
            }
        }
    };

    if (!DUMP) {
        fetch_masks(wi.t0);
        stash_masks(0);
    }
    if (producer) fetch_raw(wi.t0 + my_set * TB);
    if (producer && my_set == 0) {
        cf* slot = xt0 + my_i * 256;
        stage1(slot, wi.t0 + NS * TB);
        __builtin_amdgcn_wave_barrier();
        qr_stage23(slot, xn0 + my_i, tw5, la, la ^ (grp & 1));
    }
    __syncthreads();
    int buf = 0;
    int next_set = 1 % NS;  // producer set of tile k+1
    constexpr int FB = (TB + 1) / 2;  // consume frames [0,FB) | [FB,TB) around stage 2+3

    for (int tb = wi.t0; tb < wi.t1; tb += TB, buf ^= 1) {
        const cf* xt = xt0 + buf * NF * 256;
        const float* xn = xn0 + buf * 32;
        cf* slot = xt0 + (buf ^ 1) * NF * 256 + my_i * 256;  // next tile's slot
#if defined(SETK_ABL) && SETK_ABL == 2
        const bool prod = false;
#else
        const bool prod = producer && (my_set == next_set) && (tb + TB < wi.t1);
#endif
        next_set = (next_set + 1 == NS) ? 0 : next_set + 1;
        // ---- mask rows of tile k+1: loads in flight during this tile ----
        const float* mks = mks0 + buf * MK;
        const float* mkn = mkn0 + buf * MK;
        const bool more = tb + TB < wi.t1;
        if (!DUMP && more) fetch_masks(tb + TB);
        auto consume = [&](int tt) {
            cf x[C];
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = xt[(tt * C + c) * 256 + f];
            const bool fvalid = tb + tt < wi.t1;
            const float ws = mks[tt * F + f];
            const float wn = fvalid ? (has_mn ? mkn[tt * F + f] : 1.f - ws) : 0.f;
#if defined(SETK_ABL) && SETK_ABL == 1
            {
                float t = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) t += x[c].x + x[c].y;
                sum_s += t * ws + wn;
                return;
            }
#endif
            if (q == 0) {
                sum_s += ws;
                sum_n += wn;
                accumulate_pairs<C, (0 * NP) / NQ, (1 * NP) / NQ>(x, ws, wn, acc_s, acc_n);
            }
            if (NQ > 1 && q == 1)
                accumulate_pairs<C, (1 * NP) / NQ, (2 * NP) / NQ>(x, ws, wn, acc_s, acc_n);
            if (NQ > 2 && q == 2)
                accumulate_pairs<C, (2 * NP) / NQ, (3 * NP) / NQ>(x, ws, wn, acc_s, acc_n);
            if (NQ > 3 && q == 3)
                accumulate_pairs<C, (3 * NP) / NQ, (4 * NP) / NQ>(x, ws, wn, acc_s, acc_n);
            if (ny_active) {
                const float prod_ny = (ny_item < 2 * NP)
                                          ? xn[tt * C + ny_i] * xn[tt * C + ny_j]
                                          : 1.f;
                const float s256 = mks[tt * F + 256];
                const float n256 = fvalid ? (has_mn ? mkn[tt * F + 256] : 1.f - s256) : 0.f;
                const bool speech = (ny_item < NP) || (ny_item == 2 * NP);
                ny_acc = fmaf(speech ? s256 : n256, prod_ny, ny_acc);
            }
        };

        if (prod) stage1(slot, tb + TB + NS * TB);
        if (DUMP) {
            // spec[c][t][f], f fastest
            for (int i = q; i < NF; i += NQ) {
                const int tt = i / C, c = i - tt * C;
                const int t = tb + tt;
                if (t < wi.t1) {
                    float2* dst = reinterpret_cast<float2*>(a.spec_dump) + ((size_t)c * T + t) * F;
                    dst[f] = xt[i * 256 + f];
                    if (f == 0) dst[256] = make_float2(xn[i], 0.f);
                }
            }
        } else {
#pragma unroll
            for (int tt = 0; tt < FB; ++tt) consume(tt);
        }
        __builtin_amdgcn_wave_barrier();
        if (prod) qr_stage23(slot, xn0 + (buf ^ 1) * 32 + my_i, tw5, la, la ^ (grp & 1));
        if (!DUMP) {
#pragma unroll
            for (int tt = FB; tt < TB; ++tt) consume(tt);
        }
        if (!DUMP && more) stash_masks(buf ^ 1);
#if !(defined(SETK_ABL) && SETK_ABL == 4)
        __syncthreads();
#endif
    }

    if (!DUMP) {
        // ---- max |audio| (the renorm target, WaveReader.maxabs) ----
        if (wi.last) {
            // samples after the last frame's span are never loaded above
            const int covered = (T - 1) * a.g.hop - a.g.pad + kNfft;
            for (int c = 0; c < C; ++c)
                for (int i = covered + tid; i < n_samp; i += NT)
                    mx = fmaxf(mx, fabsf(gptr(ud.audio)[(size_t)c * n_samp + i]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            float bm = red[0];
#pragma unroll
            for (int w = 1; w < NT / 64; ++w) bm = fmaxf(bm, red[w]);
            atomicMax(a.norm_bits + wi.utt, __float_as_uint(bm));
        }

        // ---- partial slab: planes [s.re | s.im | n.re | n.im | sum_s sum_n] ----
        float* P = a.partials + (size_t)wi.part * nplanes_partial(C) * FP;
        if (q == 0) {
            store_pairs<C, (0 * NP) / NQ, (1 * NP) / NQ>(P, f, acc_s, acc_n);
            P[(4 * NP + 0) * FP + f] = sum_s;
            P[(4 * NP + 1) * FP + f] = sum_n;
        }
        if (NQ > 1 && q == 1) store_pairs<C, (1 * NP) / NQ, (2 * NP) / NQ>(P, f, acc_s, acc_n);
        if (NQ > 2 && q == 2) store_pairs<C, (2 * NP) / NQ, (3 * NP) / NQ>(P, f, acc_s, acc_n);
        if (NQ > 3 && q == 3) store_pairs<C, (3 * NP) / NQ, (4 * NP) / NQ>(P, f, acc_s, acc_n);
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

// ---------------------------------------------------------------------------
// Wave-specialised variant: a 1024-thread workgroup = 8 transform waves + 8
// covariance waves (4 waves per SIMD at <= 128 VGPRs: each role alone fits
// the budget that the merged roles overflow).  Tile k+1 is transformed into one
// half of the LDS tile while tile k is folded from the other; one s_barrier
// per tile.  Mask rows go global -> registers (one tile ahead) in the
// covariance waves; the Nyquist-bin weights travel with the transforms.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int ws_tile_frames(int c) { return (32 / c) < 8 ? (32 / c) : 8; }

SETK_DEV void wg_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int C, bool DUMP>
__global__ __launch_bounds__(1024, 4) void stft_covar_ws_kernel(Pass1Args a) {
    constexpr int NT = 1024;
    constexpr int TB = ws_tile_frames(C);
    constexpr int NF = TB * C;            // transforms per tile (<= 32)
    constexpr int NP = npairs(C);
    constexpr int NPQ = (NP + 1) / 2;     // accumulator pairs per covariance thread
    constexpr int NS = 32 / NF;           // producer sets
    constexpr int F = kBins, FP = kBinsPad;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SL = kSlotPad;                      // slot stride (padded transpose)
    cf* xt0 = reinterpret_cast<cf*>(smem);            // [2][NF][SL]
    cf* tw = xt0 + 2 * NF * SL;                       // [16][16]
    cf* tw5 = tw + 256;                               // [128]
    float* win = reinterpret_cast<float*>(tw5 + 128);  // [512]
    float* xn0 = win + kNfft;                         // [2][32] nyquist bins (real)
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

    if (tid < 256) tw[tid] = a.tw256[tid];
    if (tid < 128) tw5[tid] = a.tw512[tid];
    if (tid < kNfft) win[tid] = a.window[tid];

    float mx = 0.f;
    // covariance-role state (declared here: the epilogue stores it)
    const int ct = tid - 512;
    const int f = ct & 255, q = (ct >> 8) & 1;
    cf acc_s[NPQ], acc_n[NPQ];
    float sum_s = 0.f, sum_n = 0.f, ny_acc = 0.f;
    const bool ny_active = !DUMP && ct >= 0 && ct < 2 * NP + 2;

    if (wave < 8) {
#ifndef SETK_ONLY_CONS
        // ================= transform waves =================
        const int la = tid & 15, grp = tid >> 4;
        const int my_set = grp / NF;
        const int my_i = grp - my_set * NF;
        const int my_tt = my_i / C, my_c = my_i - my_tt * C;
        const bool producer = my_set < NS;
        gcfloat_p my_audio = gptr(ud.audio) + (size_t)my_c * n_samp;
        const float2* w2 = reinterpret_cast<const float2*>(win);
        const bool ny_lane = !DUMP && producer && my_c == 0 && la == 0;

        cf raw[16];
        float raw_ms = 0.f, raw_mn = 0.f;
        bool raw_ok = false;
        auto fetch = [&](int tb_tile) {
            const int t = tb_tile + my_tt;
            raw_ok = t < wi.t1;
#ifdef SETK_NO_GLOAD
            load_raw(raw, my_audio, n_samp, t * a.g.hop - a.g.pad, la, raw_ok && n_samp < 0);
#else
            load_raw(raw, my_audio, n_samp, t * a.g.hop - a.g.pad, la, raw_ok);
#endif
            if (ny_lane) {
                raw_ms = 0.f;
                raw_mn = 0.f;
                if (raw_ok) {
                    raw_ms = gptr(ud.mask_s)[(size_t)t * F + 256];
                    if (has_mn) raw_mn = gptr(ud.mask_n)[(size_t)t * F + 256];
                }
            }
        };
        auto produce = [&](int b, int tb_next_own) {
            cf* slot = xt0 + (b * NF + my_i) * SL;
            cf v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 d = raw[j];
                const float2 w = w2[la + 16 * j];
                mx = fmaxf(mx, fmaxf(fabsf(d.x), fabsf(d.y)));
                v[j] = make_float2(d.x * w.x, d.y * w.y);
            }
            if (ny_lane) {
                const float s = clamp ? fminf(raw_ms, 1.f) : raw_ms;
                nym[(b * 2 + 0) * 8 + my_tt] = s;
                nym[(b * 2 + 1) * 8 + my_tt] = raw_ok ? (has_mn ? raw_mn : 1.f - s) : 0.f;
            }
            fft256_stage_a_pad<-1>(v, slot, tw, la);
            __builtin_amdgcn_wave_barrier();
#ifdef SETK_FETCH_MID
            __builtin_amdgcn_sched_barrier(0);
            if (tb_next_own < wi.t1) fetch(tb_next_own);
            __builtin_amdgcn_sched_barrier(0);
#endif
            qr_stage23<true>(slot, xn0 + b * 32 + my_i, tw5, la, 0);
            // the next frame is requested only now: held across the transform it
            // would push the role past 128 VGPRs (the spill reloads then serialise
            // behind the very loads they make room for)
#ifndef SETK_FETCH_MID
            if (tb_next_own < wi.t1) fetch(tb_next_own);
#endif
        };

#ifdef SETK_PROD_PRIO
        __builtin_amdgcn_s_setprio(SETK_PROD_PRIO);
#endif
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
        for (int e = 0; e < NPQ; ++e) {
            acc_s[e] = make_float2(0.f, 0.f);
            acc_n[e] = make_float2(0.f, 0.f);
        }
        // nyquist-bin items (bin 256 is purely real): covariance threads [0, 2*NP+2)
        int ny_i = 0, ny_j = 0;
        {
            const int e = (ct < NP) ? ct : ct - NP;
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
                        float2* dst =
                            reinterpret_cast<float2*>(a.spec_dump) + ((size_t)c * T + t) * F;
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
                        sum_s += ws;
                        sum_n += wn;
                        accumulate_pairs<C, 0, NP / 2>(x, ws, wn, acc_s, acc_n);
                    } else {
                        accumulate_pairs<C, NP / 2, NP>(x, ws, wn, acc_s, acc_n);
                    }
                    if (ny_active) {
                        const float prod_ny =
                            (ct < 2 * NP) ? xn[tt * C + ny_i] * xn[tt * C + ny_j] : 1.f;
                        const bool speech = (ct < NP) || (ct == 2 * NP);
                        const float w256 = nym[(buf * 2 + (speech ? 0 : 1)) * 8 + tt];
                        ny_acc = fmaf(w256, prod_ny, ny_acc);
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
        if (wave >= 8) {
            float* P = a.partials + (size_t)wi.part * nplanes_partial(C) * FP;
            if (q == 0) {
                store_pairs<C, 0, NP / 2>(P, f, acc_s, acc_n);
                P[(4 * NP + 0) * FP + f] = sum_s;
                P[(4 * NP + 1) * FP + f] = sum_n;
            } else {
                store_pairs<C, NP / 2, NP>(P, f, acc_s, acc_n);
            }
            if (ny_active) {
                if (ct < NP) {
                    P[(0 * NP + ct) * FP + 256] = ny_acc;
                    P[(1 * NP + ct) * FP + 256] = 0.f;
                } else if (ct < 2 * NP) {
                    P[(2 * NP + ct - NP) * FP + 256] = ny_acc;
                    P[(3 * NP + ct - NP) * FP + 256] = 0.f;
                } else {
                    P[(4 * NP + ct - 2 * NP) * FP + 256] = ny_acc;
                }
            }
        }
    }
}

template <int C, bool DUMP>
static hipError_t launch_pass1_ws_t(const Pass1Args& a, int n_items, hipStream_t s) {
    constexpr int NF = ws_tile_frames(C) * C;
    const size_t lds = (size_t)2 * NF * kSlotPad * sizeof(cf) + 256 * sizeof(cf) + 128 * sizeof(cf) +
                       kNfft * sizeof(float) + (64 + 32 + 16) * sizeof(float);
    auto k = stft_covar_ws_kernel<C, DUMP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_items), dim3(1024), lds, s, a);
    return hipGetLastError();
}

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
                const cf p = cmulc(x[i], x[j]);
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

// ---------------------------------------------------------------------------
// v3: wave-specialised, with the raw samples DMA'd straight into LDS.
//
// The transform of a frame is split over TWO tile periods and the frames'
// samples are fetched global -> LDS (global_load_lds_dwordx4, no registers)
// into the very slot the spectrum will occupy, one period before they are
// needed.  Four X buffers of TB = 16/C frames rotate:
//   period it:  covariance waves fold tile it           (buffer  it      & 3)
//               set  (it+1)&1 : stage b + split, tile it+1 (buffer (it+1) & 3)
//                               then DMA for tile it+3   (buffer (it+3) & 3,
//                               folded during period it-1, free again)
//               set   it   &1 : window + stage a, tile it+2 (buffer (it+2) & 3,
//                               its DMA was issued in period it-1)
// Each set is 4 waves = 16 quad-rows; one s_barrier per period.  No global
// load sits on a transform's critical path, and the transform waves need no
// prefetch registers.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int v3_tile_frames(int c) { return (16 / c) < 8 ? (16 / c) : 8; }

#define SETK_LDS __attribute__((address_space(3)))

template <int C, bool DUMP>
__global__ __launch_bounds__(1024, 4) void stft_covar_v3_kernel(Pass1Args a) {
    constexpr int NT = 1024;
    constexpr int TB = v3_tile_frames(C);
    constexpr int NF = TB * C;            // transforms per tile (<= 16)
    constexpr int NB = 4;                 // X buffers
    constexpr int NP = npairs(C);
    constexpr int ND = PairSplit<C>::ND, NO = PairSplit<C>::NO;
    constexpr int F = kBins, FP = kBinsPad;
    constexpr int SL = kSlotPad;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* xt0 = reinterpret_cast<cf*>(smem);            // [NB][NF][SL]
    cf* tw = xt0 + NB * NF * SL;                      // [16][16]
    cf* tw5 = tw + 256;                               // [128]
    float* win = reinterpret_cast<float*>(tw5 + 128);  // [512]
    float* xn0 = win + kNfft;                         // [NB][16] nyquist bins (real)
    float* nym = xn0 + NB * 16;                       // [NB][2][8] bin-256 weights
    float* red = nym + NB * 16;                       // [16]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples;
    const int T = ud.num_frames;
    const bool clamp = (a.flags & 0x2) != 0;
    const bool has_mn = ud.mask_n != nullptr;
    const int ntiles = (wi.t1 - wi.t0 + TB - 1) / TB;

    if (tid < 256) tw[tid] = a.tw256[tid];
    if (tid < 128) tw5[tid] = a.tw512[tid];
    if (tid < kNfft) win[tid] = a.window[tid];

#ifdef SETK_TRACE
    unsigned long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto now = []() {
        unsigned long long t;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
#define SETK_T(var) const unsigned long long var = now()
#define SETK_ACC(k, a, b) tr[k] += (b) - (a)
#else
#define SETK_T(var)
#define SETK_ACC(k, a, b)
#endif
    float mx = 0.f;
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
        // ================= transform waves =================
        const int la = tid & 15, lane = tid & 63;
        const int set = wave >> 2;                 // tiles of this parity
        const int qi = (tid >> 4) & 15;            // quad-row within the set
        const bool producer = qi < NF;
        const int my_tt = qi / C, my_c = qi - my_tt * C;
        gcfloat_p my_audio = gptr(ud.audio) + (size_t)my_c * n_samp;
        const float2* w2 = reinterpret_cast<const float2*>(win);
        const bool ny_lane = !DUMP && producer && my_c == 0 && la == 0;
        float raw_ms = 0.f, raw_mn = 0.f, nxt_ms = 0.f, nxt_mn = 0.f;

        // frame of tile j handled by this quad-row
        auto frame_of = [&](int j, int tt) { return wi.t0 + j * TB + tt; };
        // DMA the frames of tile j (a tile of this wave's set) into their slots;
        // the whole wave copies each of its four quad-rows' frames (2 KB each).
        auto issue_dma = [&](int j) {
            if (j >= ntiles) return;
            // (mask load first: waiting for it must not wait for the younger DMAs)
            if (ny_lane) {
                const int t = frame_of(j, my_tt);
                nxt_ms = 0.f;
                nxt_mn = 0.f;
                if (t < wi.t1) {
                    nxt_ms = gptr(ud.mask_s)[(size_t)t * F + 256];
                    if (has_mn) nxt_mn = gptr(ud.mask_n)[(size_t)t * F + 256];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = (wave & 3) * 4 + r;  // wave-uniform
                if (qr < NF) {
                    const int tt = qr / C, c = qr - tt * C;
                    const int t = frame_of(j, tt);
                    const int s = t * a.g.hop - a.g.pad;
#ifdef SETK_NO_DMA
                    if (t < wi.t1 && s >= 0 && s + kNfft <= n_samp && n_samp < 0) {
#else
                    if (t < wi.t1 && s >= 0 && s + kNfft <= n_samp) {
#endif
                        gcfloat_p src = gptr(ud.audio) + (size_t)c * n_samp + s + lane * 4;
                        SETK_LDS char* dst =
                            (SETK_LDS char*)(xt0 + ((j & 3) * NF + qr) * SL);
                        __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
                        __builtin_amdgcn_global_load_lds(src + 256, dst + 1024, 16, 0, 0);
                    }
                }
            }
        };
        // window + first radix-16 + transposed store (in place in the slot)
        auto phase_a = [&](int j) {
            if (j >= ntiles || !producer) return;
            cf* slot = xt0 + ((j & 3) * NF + qi) * SL;
            const int t = frame_of(j, my_tt);
            const int s = t * a.g.hop - a.g.pad;
            const bool valid = t < wi.t1;
            const bool dma = valid && s >= 0 && s + kNfft <= n_samp;
            cf v[16];
            if (dma) {
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) v[jj] = slot[la + 16 * jj];
            } else {
                load_raw(v, my_audio, n_samp, s, la, valid);  // edge (reflect) / padding frame
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const float2 w = w2[la + 16 * jj];
                mx = fmaxf(mx, fmaxf(fabsf(v[jj].x), fabsf(v[jj].y)));
                v[jj] = make_float2(v[jj].x * w.x, v[jj].y * w.y);
            }
            __builtin_amdgcn_wave_barrier();
            fft256_stage_a_pad<-1>(v, slot, tw, la);
        };
        // second radix-16 + Hermitian split; X[0..255] in the slot, X[256] aside
        auto phase_b = [&](int j) {
            if (j >= ntiles || !producer) return;
            const int b = j & 3;
            cf* slot = xt0 + (b * NF + qi) * SL;
            qr_stage23<true>(slot, xn0 + b * 16 + qi, tw5, la, 0);
            if (ny_lane) {
                const bool valid = frame_of(j, my_tt) < wi.t1;
                const float sp = clamp ? fminf(raw_ms, 1.f) : raw_ms;
                nym[(b * 2 + 0) * 8 + my_tt] = sp;
                nym[(b * 2 + 1) * 8 + my_tt] = valid ? (has_mn ? raw_mn : 1.f - sp) : 0.f;
            }
        };

#ifndef SETK_ONLY_CONS
        wg_barrier();  // tables ready
        // periods -3 .. ntiles-1 (see the header); the barrier-free period -3 only
        // starts the first DMA
#pragma unroll 1
        for (int it = -3; it < ntiles; ++it) {
            const bool b_set = ((it + 1) & 1) == set;  // phase b of tile it+1, DMA of tile it+3
            SETK_T(t0);
            if (b_set) {
                // buffer (it+3)&3 was folded in period it-1: start its DMA first, the
                // copy then has this whole period to land
                issue_dma(it + 3);
                SETK_T(t1);
                if (it + 1 >= 0) phase_b(it + 1);
                raw_ms = nxt_ms;
                raw_mn = nxt_mn;
                SETK_T(t2);
                SETK_ACC(0, t0, t1);  // DMA issue
                SETK_ACC(1, t1, t2);  // stage b + split
                if (it >= -2) wg_barrier();
                SETK_T(t3);
                SETK_ACC(2, t2, t3);  // barrier wait (b set)
            } else {
                if (it + 2 >= 0) {
                    // the DMA of tile it+2 was issued one period ago by this wave
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    SETK_T(t1);
                    phase_a(it + 2);
                    SETK_T(t2);
                    SETK_ACC(3, t0, t1);  // DMA wait
                    SETK_ACC(4, t1, t2);  // window + stage a
                }
                SETK_T(t2b);
                if (it >= -2) wg_barrier();
                SETK_T(t3);
                SETK_ACC(5, t2b, t3);  // barrier wait (a set)
            }
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
        // mask rows: global -> registers, two tiles ahead
        float ma_s[TB], ma_n[TB], mb_s[TB], mb_n[TB];
        auto fetch_masks = [&](int j, float (&ms)[TB], float (&mn)[TB]) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                const int t = wi.t0 + j * TB + tt;
                float vs = 0.f, vn = 0.f;
                if (t < wi.t1) {
                    vs = gptr(ud.mask_s)[(size_t)t * F + f];
                    if (has_mn) vn = gptr(ud.mask_n)[(size_t)t * F + f];
                }
                ms[tt] = vs;
                mn[tt] = vn;
            }
        };
        if (!DUMP) {
            fetch_masks(0, ma_s, ma_n);
            fetch_masks(1, mb_s, mb_n);
        }
        wg_barrier();  // tables ready
        wg_barrier();  // period -2
        wg_barrier();  // period -1
        // one period: fold tile `it` with its mask rows m, then refill m for tile it+2
        auto period = [&](int it, float (&m_s)[TB], float (&m_n)[TB]) {
            SETK_T(c0);
            const int b = it & 3;
            const cf* xt = xt0 + b * NF * SL;
            const float* xn = xn0 + b * 16;
            const int tb = wi.t0 + it * TB;
            if (DUMP) {
                for (int i = q; i < NF; i += 2) {
                    const int tt = i / C, c = i - tt * C;
                    const int t = tb + tt;
                    if (t < wi.t1) {
                        float2* dst =
                            reinterpret_cast<float2*>(a.spec_dump) + ((size_t)c * T + t) * F;
                        dst[f] = xt[i * SL + f];
                        if (f == 0) dst[256] = make_float2(xn[i], 0.f);
                    }
                }
            } else {
#pragma unroll
                for (int tt = 0; tt < TB; ++tt) {
                    cf x[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) x[c] = xt[(tt * C + c) * SL + f];
                    const bool fvalid = tb + tt < wi.t1;
                    const float ws = clamp ? fminf(m_s[tt], 1.f) : m_s[tt];
                    const float wn = fvalid ? (has_mn ? m_n[tt] : 1.f - ws) : 0.f;
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
                        const float w256 = nym[(b * 2 + (speech ? 0 : 1)) * 8 + tt];
                        aux0 = fmaf(w256, prod_ny, aux0);
                    }
                }
                fetch_masks(it + 2, m_s, m_n);
            }
            SETK_T(c1);
            wg_barrier();
            SETK_T(c2);
            SETK_ACC(0, c0, c1);  // fold
            SETK_ACC(1, c1, c2);  // barrier wait
        };
#pragma unroll 1
        for (int it = 0; it < ntiles; it += 2) {
            period(it, ma_s, ma_n);
            if (it + 1 < ntiles) period(it + 1, mb_s, mb_n);
        }
#endif
    }

#ifdef SETK_TRACE
    if (a.trace && (tid & 63) == 0)
        for (int k = 0; k < 12; ++k) a.trace[((size_t)blockIdx.x * 16 + wave) * 16 + k] = tr[k];
#endif
    if (!DUMP) {
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

template <int C, bool DUMP>
static hipError_t launch_pass1_v3_t(const Pass1Args& a, int n_items, hipStream_t s) {
    constexpr int NF = v3_tile_frames(C) * C;
    const size_t lds = (size_t)4 * NF * kSlotPad * sizeof(cf) + 256 * sizeof(cf) +
                       128 * sizeof(cf) + kNfft * sizeof(float) + (64 + 64 + 16) * sizeof(float);
    auto k = stft_covar_v3_kernel<C, DUMP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_items), dim3(1024), lds, s, a);
    return hipGetLastError();
}

template <int C, bool DUMP, int NQ>
static hipError_t launch_pass1_t(const Pass1Args& a, int n_items, hipStream_t s) {
    constexpr int TB = tile_frames(C), NF = TB * C;
    const size_t lds = (size_t)2 * NF * 256 * sizeof(cf) + 256 * sizeof(cf) + 128 * sizeof(cf) +
                       kNfft * sizeof(float) + 64 * sizeof(float) + 16 * sizeof(float) +
                       (size_t)4 * TB * kBins * sizeof(float);
    auto k = stft_covar_kernel<C, DUMP, NQ>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (getenv("SETK_DEBUG")) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k),
                                                           256 * NQ, lds);
        fprintf(stderr, "[setk] pass1<%d,%d,%d> lds=%zu items=%d blocks/CU=%d\n", C, (int)DUMP, NQ,
                lds, n_items, nb);
    }
    hipLaunchKernelGGL(k, dim3(n_items), dim3(256 * NQ), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pass1(int C, bool dump, const Pass1Args& a, int n_items, hipStream_t s) {
    // NQ = 2 (512 threads, 2 waves/SIMD).  NQ = 4 (1024 threads, <= 128 VGPRs)
    // was measured 3x slower on MI355X: spills plus 4x redundant tile reads.
    static const int ws = [] {
        const char* e = getenv("SETK_P1_WS");
        return e ? atoi(e) : 2;
    }();
#define SETK_CASE(c)                                                                      \
    case c:                                                                               \
        if (ws == 2) {                                                                    \
            if (dump) return launch_pass1_v3_t<c, true>(a, n_items, s);                    \
            return launch_pass1_v3_t<c, false>(a, n_items, s);                             \
        }                                                                                 \
        if (ws) {                                                                         \
            if (dump) return launch_pass1_ws_t<c, true>(a, n_items, s);                    \
            return launch_pass1_ws_t<c, false>(a, n_items, s);                             \
        }                                                                                 \
        if (dump) return launch_pass1_t<c, true, 2>(a, n_items, s);                        \
        return launch_pass1_t<c, false, 2>(a, n_items, s);
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
