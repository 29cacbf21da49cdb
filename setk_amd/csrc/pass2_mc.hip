// pass2_mc.hip -- pass 2 with the transforms on the matrix cores (mcdft.h):
// rDFT recompute + w^H x + inverse rDFT + window + overlap-add.
//
// Replaces (funcwj/setk): Beamformer.beamform (libs/beamformer.py:220-234),
// post-masking (apply_adaptive_beamformer.py:174-175) and inverse_stft
// (libs/utils.py:142-173 -> librosa.istft 0.8.1: irfft, * window, overlap-add,
// / sum(window^2) where > tiny, trim n_fft/2; the inf-norm rescale is scale_kernel).
//
// One wavefront owns one frame at a time: it transforms the C channels one after the other
// (12 MFMA + ~60 VALU wave-instructions each; the next channel's samples in flight), folds
// conj(w_c) X_c into four complex accumulators per lane (the lane's bins are fixed, its
// weights come from an LDS table as one base address + immediates), adds the odd family
// X[16 + 32 q] of all channels from ONE extra tile per frame, scales the frame's spectrum by a
// power of two into the fp16 operand range, inverse-transforms (12 MFMA), windows and leaves
// the frame in its LDS slot.  No workgroup barrier in that chain; the overlap-add of a
// 16-frame super-tile is pass2.hip's.  ~110 VGPRs: four waves per SIMD (two 512-thread
// workgroups per CU) where the butterfly kernel ran two.
#include "common.h"
#include "fft512.h"
#include "mcdft.h"
#include <cstdio>
#include <cstdlib>

namespace setk {

// raw frame samples in the operand layout of mcdft.h (pass1_mc.hip)
template <class FloatPtr>
SETK_DEV void load_raw_mc2(float (&v)[8], FloatPtr x, int n_samp, int s, int lane, bool valid) {
    if (!valid) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        return;
    }
    const int o = 128 * (lane >> 4) + (lane & 15);
    if (s >= 0 && s + kFrame <= n_samp) {
        FloatPtr p = x + s + o;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[16 * e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = x[reflect_index(s + o + 16 * e, n_samp)];
    }
}

constexpr int kP2McThreads = 512;
constexpr int kP2McWaves = kP2McThreads / 64;

constexpr int kP2McTiles = 12;  // BR_H .. IT_L (10 tiles, contiguous words) + OT_H, OT_L
// LDS plan (bytes): frames (16 + keep) * 2048 | wtab C * 257 * 8 | winsq 2048 | once-per-frame
// operand tiles 12 * 1024 | synthesis window rows 2048 | a16 scratch 8 * 8 * kOddPitch * 4 |
// yodd 8 * 16 * 4 | red 64
size_t pass2_mc_lds_bytes(int C, int keep) {
    const size_t wt = ((size_t)C * kBins * sizeof(cf) + 15) & ~(size_t)15;
    return (size_t)(kSuperTile + keep) * kFrame * sizeof(float) + wt + 2048 + kP2McTiles * 1024 + 2048 +
           (size_t)kP2McWaves * 8 * mc::kOddPitch * sizeof(float) + kP2McWaves * 16 * sizeof(float) + 64;
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

template <int C>
__global__ __launch_bounds__(kP2McThreads, 4) void beamform_istft_mc_kernel(Pass2Args a) {
    constexpr int NT = kP2McThreads;
    constexpr int NW = kP2McWaves;
    constexpr int F = kBins;
    constexpr int ST = kSuperTile;
    constexpr int FPW = ST / NW;  // frames per wave and super-tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int keep = a.g.keep;
    float* frames = reinterpret_cast<float*>(smem);  // [(keep + ST)][512]
    char* p = smem + (size_t)(ST + keep) * kFrame * sizeof(float);
    cf* wtab = reinterpret_cast<cf*>(p);  // [C][257]
    p += ((size_t)C * F * sizeof(cf) + 15) & ~(size_t)15;
    float* winsq = reinterpret_cast<float*>(p);
    p += 2048;
    mc::u4* tiles = reinterpret_cast<mc::u4*>(p);  // [12][64]: BR_H BR_L BI_H BI_L G0_H G0_L G1_H G1_L IT_H IT_L OT_H OT_L
    p += kP2McTiles * 1024;
    mc::f4* synr = reinterpret_cast<mc::f4*>(p);  // [2][64] float4: synthesis window rows 0..3 / 4..7 of a lane
    p += 2048;
    float* a16s = reinterpret_cast<float*>(p);  // [NW][8][kOddPitch]
    p += (size_t)NW * 8 * mc::kOddPitch * sizeof(float);
    float* yodd_s = reinterpret_cast<float*>(p);  // [NW][16]
    p += NW * 16 * sizeof(float);
    float* red = reinterpret_cast<float*>(p);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const WorkItem wi = a.items[blockIdx.x];
    const UttDesc ud = a.utts[wi.utt];
    const int n_samp = ud.num_samples;
    const int T = ud.num_frames;
    const int hop = a.g.hop;
    const bool post_mask = (a.flags & 0x4) != 0;
    const bool clamp = (a.flags & 0x2) != 0;

    for (int i = tid; i < kNfft; i += NT) winsq[i] = a.winsq[i];
    {
        const cf* wsrc = reinterpret_cast<const cf*>(a.weight) + (size_t)wi.utt * C * kBinsPad;
        for (int i = tid; i < C * F; i += NT) {
            const int c = i / F, f = i - c * F;
            wtab[i] = wsrc[c * kBinsPad + f];
        }
    }
    mc::stage_tiles(tiles, a.mc_tab, mc::kW_BR_H, 10, tid, NT);
    mc::stage_tiles(tiles + 10 * 64, a.mc_tab, mc::kW_OT_H, 2, tid, NT);
    for (int i = tid; i < 128; i += NT) {
        const int l = i & 63, hf = i >> 6;
        synr[i] = (mc::f4){a.mc_syn[(4 * hf + 0) * 64 + l], a.mc_syn[(4 * hf + 1) * 64 + l],
                           a.mc_syn[(4 * hf + 2) * 64 + l], a.mc_syn[(4 * hf + 3) * 64 + l]};
    }
    mc::Fwd K;
    mc::load_fwd(K, a.mc_tab, lane);
    float win[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) win[e] = gptr(a.mc_win)[e * 64 + lane];
    float* a16w = a16s + wave * 8 * mc::kOddPitch;  // rows = channels (columns 8..15 of the odd tile re-read row 7)
    float* yoddw = yodd_s + wave * 16;
    const int lane_bin = mc::bin_of(c16, g, 0);
    const cf* wl = wtab + lane_bin;                       // + c * F + 32 r
    const int jo = c16 < C ? c16 : 0;                     // odd-family tile: column = channel
    const cf* wo = wtab + jo * F + 16 + 64 * g;           // w_j[16 + 32 (2 g)], [+ 32] the next
    const bool odd_on = c16 < C;
    float omax = 0.f;

    const int t_first = max(wi.t0 - keep, 0);
    for (int i = tid; i < keep * kFrame; i += NT) frames[i] = 0.f;

    for (int ts = t_first; ts < wi.t1; ts += ST) {
        __syncthreads();  // tables ready / slots free (carry copied)
#pragma unroll 1
        for (int k = 0; k < FPW; ++k) {
            const int fi = wave + NW * k;
            const int t = ts + fi;
            const bool tvalid = t < T;
            float* slot = frames + (size_t)(keep + fi) * kFrame;
            mc::f4 yr = {0.f, 0.f, 0.f, 0.f}, yi = {0.f, 0.f, 0.f, 0.f};
            float raw[8];
            load_raw_mc2(raw, gptr(ud.audio), n_samp, t * hop - a.g.pad, lane, tvalid);
#pragma unroll 1
            for (int c = 0; c < C; ++c) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = raw[e];
                if (c + 1 < C)
                    load_raw_mc2(raw, gptr(ud.audio) + (size_t)(c + 1) * n_samp, n_samp, t * hop - a.g.pad, lane, tvalid);
                mc::f4 zr, zi, a16;
                mc::forward(x, win, K, zr, zi, a16);
                mc::store_a16(a16w, c, lane, a16);
                const cf* wc = wl + c * F;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cf w = wc[32 * r];
                    yr[r] = fmaf(zr[r], w.x, fmaf(zi[r], w.y, yr[r]));
                    yi[r] = fmaf(zi[r], w.x, fmaf(-zr[r], w.y, yi[r]));
                }
            }
            // ---- odd family of all channels: one tile, then the sum over the channel lanes ----
            float yo[4];
            {
                asm volatile("" ::: "memory");  // the once-per-frame tiles are re-read, not kept
                const mc::f4 d = mc::odd_tile(a16w, mc::lds_h8(tiles, 10, lane), mc::lds_h8(tiles, 11, lane), lane, C);
                const cf w0 = wo[0], w1 = wo[32];
                yo[0] = odd_on ? fmaf(d[0], w0.x, d[1] * w0.y) : 0.f;
                yo[1] = odd_on ? fmaf(d[1], w0.x, -d[0] * w0.y) : 0.f;
                yo[2] = odd_on ? fmaf(d[2], w1.x, d[3] * w1.y) : 0.f;
                yo[3] = odd_on ? fmaf(d[3], w1.x, -d[2] * w1.y) : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) yo[i] = row8_sum(yo[i]);
            }
            // ---- optional post-mask ----
            if (post_mask && tvalid) {
                const float* mrow = ud.mask_s + (size_t)t * F;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float m = mrow[lane_bin + 32 * r];
                    if (clamp) m = fminf(m, 1.f);
                    yr[r] *= m;
                    yi[r] *= m;
                }
                float m0 = mrow[16 + 64 * g], m1 = mrow[48 + 64 * g];
                if (clamp) { m0 = fminf(m0, 1.f); m1 = fminf(m1, 1.f); }
                yo[0] *= m0;
                yo[1] *= m0;
                yo[2] *= m1;
                yo[3] *= m1;
            }
            // only Re Y[0], Re Y[256] reach the inverse (numpy irfft drops their imaginary parts)
            yi[0] = (lane == 0) ? 0.f : yi[0];
            yi[3] = (lane == 32) ? 0.f : yi[3];
            // ---- power-of-two scale of the frame into the fp16 operand range: max < 2^11 ----
            float mxl = fmaxf(fmaxf(fabsf(yo[0]), fabsf(yo[1])), fmaxf(fabsf(yo[2]), fabsf(yo[3])));
#pragma unroll
            for (int r = 0; r < 4; ++r) mxl = fmaxf(mxl, fmaxf(fabsf(yr[r]), fabsf(yi[r])));
            const float mxw = wave_max_nonneg(mxl);
            int ex = (int)((__builtin_bit_cast(unsigned, mxw) >> 23) & 0xff);  // mxw < 2^(ex - 126)
            ex = ex < 16 ? 16 : (ex > 250 ? 250 : ex);                         // (zero / tiny / huge frames)
            const float sc = __builtin_bit_cast(float, (unsigned)(264 - ex) << 23);   // 2^(137 - ex)
            const float isc = tvalid ? __builtin_bit_cast(float, (unsigned)(ex - 10) << 23) : 0.f;
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
                yr[r] *= sc;
                yi[r] *= sc;
            }
            float bmid[8];
            mc::inverse_a(yr, yi, e16, mc::lds_h8(tiles, 0, lane), mc::lds_h8(tiles, 1, lane),
                          mc::lds_h8(tiles, 2, lane), mc::lds_h8(tiles, 3, lane), K.tr, K.ti, bmid, lane);
            asm volatile("" ::: "memory");
            mc::f4 y0, y1;
            mc::inverse_b(bmid, mc::lds_h8(tiles, 4, lane), mc::lds_h8(tiles, 5, lane),
                          mc::lds_h8(tiles, 6, lane), mc::lds_h8(tiles, 7, lane), y0, y1);
            // ---- synthesis window (x 1/512), back-scale, frame into its slot ----
            const mc::f4 s0 = synr[lane], s1 = synr[64 + lane];
            float* dst = slot + 64 * g + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dst[16 * r] = y0[r] * (s0[r] * isc);
                dst[256 + 16 * r] = y1[r] * (s1[r] * isc);
            }
        }
        __syncthreads();
        // ---- overlap-add: padded positions [pos0, pos1) are now complete ----
        {
            int pos0 = max(ts, wi.t0) * hop;
            int pos1 = min(ts + ST, wi.t1) * hop;
            if (wi.last && ts + ST >= wi.t1) pos1 = (T - 1) * hop + kNfft;
            for (int n = pos0 + tid; n < pos1; n += NT) {
                int t_hi = min(n / hop, T - 1);
                int t_lo = max((n - kNfft) / hop + 1, 0);
                if (n < kNfft) t_lo = 0;
                float v = 0.f, wss = 0.f;
                for (int tt = t_lo; tt <= t_hi; ++tt) {
                    const int off = n - tt * hop;
                    const int sl = tt - ts + keep;
                    v += frames[sl * kFrame + off];
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
        {
            float4* dst = reinterpret_cast<float4*>(frames);
            const float4* src = reinterpret_cast<const float4*>(frames + ST * kFrame);
            const int n4 = keep * (kFrame / 4);
            for (int i = tid; i < n4; i += NT) dst[i] = src[i];
        }
    }
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

template <int C>
static hipError_t launch_pass2_mc_t(const Pass2Args& a, int n_items, hipStream_t s) {
    const size_t lds = pass2_mc_lds_bytes(C, a.g.keep);
    auto k = beamform_istft_mc_kernel<C>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_items), dim3(kP2McThreads), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pass2_mc(int C, const Pass2Args& a, int n_items, hipStream_t s) {
    switch (C) {
        case 1: return launch_pass2_mc_t<1>(a, n_items, s);
        case 2: return launch_pass2_mc_t<2>(a, n_items, s);
        case 3: return launch_pass2_mc_t<3>(a, n_items, s);
        case 4: return launch_pass2_mc_t<4>(a, n_items, s);
        case 5: return launch_pass2_mc_t<5>(a, n_items, s);
        case 6: return launch_pass2_mc_t<6>(a, n_items, s);
        case 7: return launch_pass2_mc_t<7>(a, n_items, s);
        case 8: return launch_pass2_mc_t<8>(a, n_items, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace setk
