// pass2.hip -- rFFT recompute + w^H x + irFFT + overlap-add, and the renorm.
//
// Replaces (funcwj/setk): Beamformer.beamform (libs/beamformer.py:220-234),
// post-masking (apply_adaptive_beamformer.py:174-175) and inverse_stft
// (libs/utils.py:142-173 -> librosa.istft 0.8.1: irfft, * window, overlap-add,
// / sum(window^2) where > tiny, trim n_fft/2, inf-norm rescale).
//
// The STFT is recomputed from the audio (15 MB/utt, Infinity-Cache sized)
// instead of being stored in pass 1 (31 MB write + 31 MB read).  A workgroup
// walks its frame range in super-tiles of 16 frames; for every channel the 16
// quad-rows transform 16 frames at once and each thread folds conj(w_c) X_c
// into register accumulators (beamforming is a sum over channels, so only one
// channel's spectrum is ever resident in LDS).  The accumulated Y is merged
// into the packed inverse form, transformed back by the same quad-rows, and the
// windowed frames are overlap-added out of LDS with coalesced stores.
#include "common.h"
#include "fft512.h"
#include <cstdio>
#include <cstdlib>

namespace setk {

// LDS row stride of this pass: 18 = 16-byte aligned rows, b128 reads (fft512.h), the
// product at 256 VGPRs; 17 = the 8-byte form pass 1 uses, for tighter register budgets
#ifndef SETK_P2_ROW
#define SETK_P2_ROW 18
#endif
constexpr int kRow2 = SETK_P2_ROW;

// N consecutive entries of a lane's table row (contiguous for even ROW, 16 apart for odd)
template <int N>
SETK_DEV void load_tab(const cf* row, cf (&out)[N]) {
    if (LaneTab<kRow2>::rows) {
        lds_row<N, true>(row, out);
    } else {
#pragma unroll
        for (int n = 0; n < N; ++n) out[n] = row[n * 16];
    }
}

// LDS plan (bytes): slots (16+keep)*2304 (padded 16x16 transpose) | wtab C*257*8 (BF mode) |
// per-lane table rows (window, twiddles; fft512.h) | winsq 2048 | red 16
size_t pass2_lds_bytes(int C, int keep) {
    size_t wt = ((size_t)C * kBins * sizeof(cf) + 15) & ~(size_t)15;
    return (size_t)(kSuperTile + keep) * slot_entries(kRow2) * sizeof(cf) + wt + table_entries(kRow2) * sizeof(cf) + 2048 +
           64;
}

// ISTFT_ONLY: Y comes from the per-item spectrogram (ud.audio reinterpreted as
// spec[t][f]) instead of being beamformed from audio.
//
// Quad-row g of the workgroup owns frame ts + g of the super-tile: it transforms
// the C channels of that frame one after the other, folds conj(w_c) X_c into
// its lanes' accumulators (lane la owns bins k = la + 16 m and 256 - k), merges,
// inverse-transforms and leaves the windowed frame in its LDS slot.  Nothing in
// that chain needs a workgroup barrier; only the overlap-add does.
template <int C, bool ISTFT_ONLY>
__global__ __launch_bounds__(kPass2Threads, SETK_P2_WAVES) void beamform_istft_kernel(Pass2Args a) {
    constexpr int NT = kPass2Threads;
    constexpr int F = kBins;
    constexpr int ST = kSuperTile;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int keep = a.g.keep;
    constexpr int SL = slot_entries(kRow2);              // slot stride in complex entries
    cf* slots = reinterpret_cast<cf*>(smem);  // [(keep + 16)][SL]
    char* p = smem + (size_t)(ST + keep) * SL * sizeof(cf);
    cf* wtab = reinterpret_cast<cf*>(p);  // [C][257]
    p += ((size_t)C * F * sizeof(cf) + 15) & ~(size_t)15;
    cf* win_l = reinterpret_cast<cf*>(p);  // per-lane table rows (synthesis window ==
    cf* tw_l = win_l + LaneTab<kRow2>::size;  // analysis window)
    cf* tw5_l = tw_l + LaneTab<kRow2>::size;
    p += table_entries(kRow2) * sizeof(cf);
    float* winsq = reinterpret_cast<float*>(p);
    p += 2048;
    float* red = reinterpret_cast<float*>(p);

    const int tid = threadIdx.x;
    const int la = tid & 15, grp = tid >> 4;
    const bool lane0 = (la == 0);
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples;
    const int T = ud.num_frames;
    const int hop = a.g.hop;
    const bool post_mask = (a.flags & 0x4) != 0;
    const bool clamp = (a.flags & 0x2) != 0;

    fill_lane_tables<kRow2>(win_l, tw_l, tw5_l, a.window, a.tw256, a.tw512, tid, NT);
    for (int i = tid; i < kNfft; i += NT) winsq[i] = a.winsq[i];
    const cf* win_row = win_l + la * LaneTab<kRow2>::lstride;
    const cf* tw_row = tw_l + la * LaneTab<kRow2>::lstride;
    const cf* tw5_row = tw5_l + la * LaneTab<kRow2>::lstride5;
    if (!ISTFT_ONLY) {
        const cf* wsrc = reinterpret_cast<const cf*>(a.weight) + (size_t)wi.utt * C * kBinsPad;
        for (int i = tid; i < C * F; i += NT) {
            const int c = i / F, f = i - c * F;
            wtab[i] = wsrc[c * kBinsPad + f];
        }
    }
    float omax = 0.f;

    // frames needed to complete the first output position of this range
    const int t_first = max(wi.t0 - keep, 0);
    for (int i = tid; i < keep * SL; i += NT) slots[i] = make_float2(0.f, 0.f);

    for (int ts = t_first; ts < wi.t1; ts += ST) {
        cf* slot = slots + (keep + grp) * SL;  // this quad-row's frame slot
        const int t = ts + grp;
        const bool tvalid = t < T;
        cf Yk[8], Ym[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            Yk[m] = make_float2(0.f, 0.f);
            Ym[m] = make_float2(0.f, 0.f);
        }
        __syncthreads();  // tables ready / slots free (carry copied)
        if (ISTFT_ONLY) {
            if (tvalid) {
                const cf* row = reinterpret_cast<const cf*>(ud.audio) + (size_t)t * F;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int k = la + 16 * m;
                    if (k == 0) {
                        Yk[m] = make_float2(row[0].x, row[256].x);
                        Ym[m] = row[128];
                    } else {
                        Yk[m] = row[k];
                        Ym[m] = row[256 - k];
                    }
                }
            }
        } else {
            cf nxt[16];
            load_raw<const float*>(nxt, ud.audio, n_samp, t * hop - a.g.pad, la, tvalid);
#pragma unroll 1
            for (int c = 0; c < C; ++c) {
                cf v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = nxt[j];
                apply_window<kRow2>(v, win_row);
                if (c + 1 < C)
                    load_raw<const float*>(nxt, ud.audio + (size_t)(c + 1) * n_samp, n_samp, t * hop - a.g.pad,
                             la, tvalid);
                fft256_stage_a_pad<-1, kRow2>(v, slot, tw_row, la);
                __builtin_amdgcn_wave_barrier();
                fft256_stage_b_pad<-1, kRow2>(v, slot, la);  // v[pos(kb)] = Z[la + 16 kb]
                const cf* wc = wtab + c * F;
                // bins k = la + 16 m and 256 - k: one base register each + immediates
                const cf* wlo = wc + la;
                const cf* wmir = wc + (256 - 16 * 7) - la;
                cf t5[8];
                load_tab<8>(tw5_row, t5);
                // Hermitian split in registers: the mirror bin of k = la + 16 m is
                // register 15 - m of lane (16 - la) & 15 (lane 0: own register 16 - m)
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int k = la + 16 * m;
                    const cf Zk = v[dft16_pos(m)];
                    const cf src = v[dft16_pos(15 - m)];
                    const cf own = v[dft16_pos((16 - m) & 15)];
                    cf Zm = make_float2(qr_partner(src.x), qr_partner(src.y));
                    Zm.x = lane0 ? own.x : Zm.x;
                    Zm.y = lane0 ? own.y : Zm.y;
                    cf Xk, Xm;
                    rfft_split(Zk, Zm, t5[m], Xk, Xm);
                    const cf wk = wlo[16 * m], wm = wmir[16 * (7 - m)];
                    // y + x conj(w) as two FMA chains (4 instructions; cadd(cmulc())
                    // compiles to 6 without reassociation)
                    const cf yk1 = make_float2(fmaf(Xk.x, wk.x, fmaf(Xk.y, wk.y, Yk[m].x)),
                                               fmaf(Xk.y, wk.x, fmaf(-Xk.x, wk.y, Yk[m].y)));
                    const cf ym1 = make_float2(fmaf(Xm.x, wm.x, fmaf(Xm.y, wm.y, Ym[m].x)),
                                               fmaf(Xm.y, wm.x, fmaf(-Xm.x, wm.y, Ym[m].y)));
                    if (m == 0) {
                        // lane 0: X[0], X[256] are real and only Re Y[0], Re Y[256]
                        // reach the inverse (numpy irfft drops their imag); the
                        // self-paired bin 128 rides in Ym
                        const cf Z128 = v[dft16_pos(8)];
                        const cf X128 = make_float2(2.f * Z128.x, -2.f * Z128.y);
                        const cf yk0 = make_float2(fmaf(wk.x, Xk.x, Yk[m].x), fmaf(wm.x, Xm.x, Yk[m].y));
                        const cf ym0 = cadd(Ym[m], cmulc(X128, wc[128]));
                        Yk[m] = lane0 ? yk0 : yk1;
                        Ym[m] = lane0 ? ym0 : ym1;
                    } else {
                        Yk[m] = yk1;
                        Ym[m] = ym1;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- optional post-mask, merge into the packed inverse input (registers) ----
        cf Zlo[8], Zhi[8];  // 2 Z'[la + 16 m] and 2 Z'[256 - (la + 16 m)]
        cf t5m[8];
        load_tab<8>(tw5_row, t5m);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int k = la + 16 * m;
            cf yk = Yk[m], ym = Ym[m];
            if (post_mask && tvalid) {
                const float* mrow = ud.mask_s + (size_t)t * F;
                // lane 0, m == 0 carries (Re Y0, Re Y256) in yk and Y128 in ym
                const int ia = k, ib = (k == 0) ? 256 : k;
                float ma = mrow[ia], mb = mrow[ib];
                float mc = mrow[(k == 0) ? 128 : 256 - k];
                if (clamp) { ma = fminf(ma, 1.f); mb = fminf(mb, 1.f); mc = fminf(mc, 1.f); }
                yk.x *= ma;
                yk.y *= mb;
                ym = cscale(ym, mc);
            }
            cf zk, zm;
            irfft_merge(yk, ym, t5m[m], zk, zm);
            if (m == 0) {
                const cf z0 = make_float2(yk.x + yk.y, yk.x - yk.y);
                const cf z128 = make_float2(2.f * ym.x, -2.f * ym.y);
                Zlo[m] = lane0 ? z0 : zk;
                Zhi[m] = lane0 ? z128 : zm;  // lane 0: Zhi[0] holds 2 Z'[128]
            } else {
                Zlo[m] = zk;
                Zhi[m] = zm;
            }
        }
        // ---- inverse transform, windowed frame left in the slot ----
        {
            // v[j] = 2 Z'[la + 16 j]: j < 8 own; j >= 8 is the mirror value the
            // partner lane computed (lane 0: its own Zhi[16 - j], Zhi[0] for j = 8)
            cf v[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = Zlo[j];
#pragma unroll
            for (int j = 8; j < 16; ++j) {
                const cf src = Zhi[15 - j];
                const cf own = Zhi[(16 - j) & 7];
                cf pv = make_float2(qr_partner(src.x), qr_partner(src.y));
                v[j].x = lane0 ? own.x : pv.x;
                v[j].y = lane0 ? own.y : pv.y;
            }
            fft256_stage_a_pad<+1, kRow2>(v, slot, tw_row, la);
            __builtin_amdgcn_wave_barrier();
            fft256_stage_b_pad<+1, kRow2>(v, slot, la);
            const float sc = tvalid ? (1.f / 256.f) : 0.f;
            cf wsyn[16];
            load_tab<16>(win_row, wsyn);
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) {
                const int n = la + 16 * kb;
                const cf z = v[dft16_pos(kb)];
                slot[n] = make_float2(z.x * sc * wsyn[kb].x, z.y * sc * wsyn[kb].y);
            }
        }
        __syncthreads();
        // ---- overlap-add: padded positions [pos0, pos1) are now complete ----
        {
            const float* frames = reinterpret_cast<const float*>(slots);  // [(keep+16)][2 SL]
            int pos0 = max(ts, wi.t0) * hop;
            int pos1 = min(ts + ST, wi.t1) * hop;
            if (wi.last && ts + ST >= wi.t1) pos1 = (T - 1) * hop + kNfft;
            for (int n = pos0 + tid; n < pos1; n += NT) {
                // frames t with t*hop <= n < t*hop + 512
                int t_hi = min(n / hop, T - 1);
                int t_lo = max((n - kNfft) / hop + 1, 0);
                if (n < kNfft) t_lo = 0;
                float v = 0.f, wss = 0.f;
                for (int tt = t_lo; tt <= t_hi; ++tt) {
                    const int off = n - tt * hop;
                    const int sl = tt - ts + keep;  // slot of frame tt
                    v += frames[sl * (2 * SL) + off];
                    wss += winsq[off];
                }
                if (wss > 1.17549435e-38f) v /= wss;
                const int o = n - a.g.pad;
                if (o >= 0 && o < ud.out_len) {
                    ud.wave_f32[o] = v;
                    omax = fmaxf(omax, fabsf(v));
                }
            }
        }
        __syncthreads();
        // ---- carry the last `keep` frames over to the next super-tile ----
        // (source slots [16, 16+keep) and destination slots [0, keep) are disjoint)
        {
            float4* dst = reinterpret_cast<float4*>(slots);
            const float4* src = reinterpret_cast<const float4*>(slots + ST * SL);
            const int n4 = keep * (SL / 2);  // float4 per slot
            for (int i = tid; i < n4; i += NT) dst[i] = src[i];
        }
    }
    // ---- max |out| for the renorm ----
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = omax;
    __syncthreads();
    if (tid == 0) {
        float m = red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, red[w]);
        atomicMax(a.outmax_bits + wi.utt, __float_as_uint(m));
    }
}

template <int C, bool IO>
static hipError_t launch_pass2_t(const Pass2Args& a, int n_items, hipStream_t s) {
    const size_t lds = pass2_lds_bytes(IO ? 1 : C, a.g.keep);
    auto k = beamform_istft_kernel<C, IO>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (getenv("SETK_DEBUG")) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), kPass2Threads, lds);
        fprintf(stderr, "[setk] pass2<%d,%d> lds=%zu items=%d blocks/CU=%d\n", C, (int)IO, lds, n_items, nb);
    }
    hipLaunchKernelGGL(k, dim3(n_items), dim3(kPass2Threads), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pass2(int C, bool istft_only, const Pass2Args& a, int n_items, hipStream_t s) {
    if (istft_only) return launch_pass2_t<1, true>(a, n_items, s);
    switch (C) {
        case 1: return launch_pass2_t<1, false>(a, n_items, s);
        case 2: return launch_pass2_t<2, false>(a, n_items, s);
        case 3: return launch_pass2_t<3, false>(a, n_items, s);
        case 4: return launch_pass2_t<4, false>(a, n_items, s);
        case 5: return launch_pass2_t<5, false>(a, n_items, s);
        case 6: return launch_pass2_t<6, false>(a, n_items, s);
        case 7: return launch_pass2_t<7, false>(a, n_items, s);
        case 8: return launch_pass2_t<8, false>(a, n_items, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// renorm: samps * norm / (max|samps| + eps)   (libs/utils.py:166-168), float32
// or PCM16 (libsndfile float->short: lrint(x * 32767)) output.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_kernel(ScaleArgs a) {
    const int u = blockIdx.y;
    const UttDesc ud = a.utts[u];
    float norm;
    if (a.norm_override)
        norm = a.norm_override[u];
    else
        norm = __uint_as_float(a.norm_bits[u]);
    const float omax = __uint_as_float(a.outmax_bits[u]);
    const float eps = 1.1920928955078125e-07f;
    const float sc = (norm > 0.f) ? norm / (omax + eps) : 1.f;
    const int n = ud.out_len;
    const float* src = ud.wave_f32;
    const int stride = gridDim.x * 256;
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    // 16 bytes per lane where the buffers allow it (they do for torch allocations)
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(ud.wave_out)) & 15) == 0;
    const int n4 = vec ? (n >> 2) : 0;
    if (a.pcm16) {
        int16_t* dst = reinterpret_cast<int16_t*>(ud.wave_out);
        const float4* s4 = reinterpret_cast<const float4*>(src);
        short4* d4 = reinterpret_cast<short4*>(dst);
        for (int i = i0; i < n4; i += stride) {
            const float4 v = s4[i];
            d4[i] = make_short4((short)(int)rintf(v.x * sc * 32767.f), (short)(int)rintf(v.y * sc * 32767.f),
                                (short)(int)rintf(v.z * sc * 32767.f), (short)(int)rintf(v.w * sc * 32767.f));
        }
        for (int i = 4 * n4 + i0; i < n; i += stride) {
            const float v = rintf(src[i] * sc * 32767.f);
            dst[i] = (int16_t)(int)v;
        }
    } else {
        float* dst = reinterpret_cast<float*>(ud.wave_out);
        if (sc == 1.f && dst == src) return;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int i = i0; i < n4; i += stride) {
            float4 v = s4[i];
            v.x *= sc;
            v.y *= sc;
            v.z *= sc;
            v.w *= sc;
            d4[i] = v;
        }
        for (int i = 4 * n4 + i0; i < n; i += stride) dst[i] = src[i] * sc;
    }
}

hipError_t launch_scale(const ScaleArgs& a, int n_utts, int max_len, hipStream_t s) {
    int bx = (max_len + 256 * 16 - 1) / (256 * 16);  // ~4 float4 per thread
    if (bx < 1) bx = 1;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(scale_kernel, dim3(bx, n_utts), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// modular Beamformer.beamform on a stored spectrogram (any F, C <= 16):
// out[t][f] = sum_c conj(w[f][c]) spec[c][t][f]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beamform_spec_kernel(const cf* __restrict__ w,
                                                            const cf* __restrict__ spec, int C,
                                                            int T, int F, cf* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)T * F;
    if (i >= n) return;
    const int f = (int)(i % F);
    cf acc = make_float2(0.f, 0.f);
    for (int c = 0; c < C; ++c) acc = cadd(acc, cmulc(spec[(size_t)c * n + i], w[f * C + c]));
    out[i] = acc;
}

hipError_t launch_beamform_spec(const float* w_fc, const float* spec, int C, int T, int F,
                                float* out, hipStream_t s) {
    const size_t n = (size_t)T * F;
    hipLaunchKernelGGL(beamform_spec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const cf*>(w_fc), reinterpret_cast<const cf*>(spec), C, T,
                       F, reinterpret_cast<cf*>(out));
    return hipGetLastError();
}

}  // namespace setk
