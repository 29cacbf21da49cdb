// fft512.h -- real FFT-512 building blocks for gfx950 (wave64, LDS).
//
// A 512-point real transform is computed as a 256-point complex transform of
// z[n] = x[2n] + i x[2n+1] plus a Hermitian split.  The 256-point transform
// runs on a "quad-row" of 16 lanes: each lane keeps 16 complex points in
// registers, so the whole transform is two in-register radix-16 butterflies
// and ONE 16x16 transpose through LDS (XOR-swizzled, bank-conflict free for
// ds_write_b64 / ds_read_b64).  A wavefront therefore carries 4 independent
// transforms, a 256-thread workgroup 16.
#pragma once
#include <hip/hip_runtime.h>

namespace setk {

typedef float2 cf;

#define SETK_DEV __device__ __forceinline__

// Pointers fetched from descriptor tables in memory are generic to the
// compiler, and generic loads (flat_load) count against BOTH vmcnt and lgkmcnt:
// every LDS wait then also waits for the streaming loads in flight.  gptr()
// states that the pointer is global memory so that global_load is emitted.
#define SETK_GLOBAL __attribute__((address_space(1)))
typedef const SETK_GLOBAL float* gcfloat_p;
typedef float v2f __attribute__((ext_vector_type(2)));  // float2 is a class: no AS-qualified copies
typedef const SETK_GLOBAL v2f* gcfloat2_p;
template <class T>
SETK_DEV const SETK_GLOBAL T* gptr(const T* p) {
    return (const SETK_GLOBAL T*)p;
}

SETK_DEV cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
SETK_DEV cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
SETK_DEV cf cmul(cf a, cf b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
SETK_DEV cf cmulc(cf a, cf b) {
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
SETK_DEV cf cconj(cf a) { return make_float2(a.x, -a.y); }
SETK_DEV cf cscale(cf a, float s) { return make_float2(a.x * s, a.y * s); }

// position of output index q after dft16 (digit swap, an involution)
__host__ __device__ constexpr int dft16_pos(int q) { return (q >> 2) + 4 * (q & 3); }

// 4-point DFT, DIR = -1 forward (e^{-i..}), +1 inverse (unscaled)
template <int DIR>
SETK_DEV void dft4(cf& a, cf& b, cf& c, cf& d) {
    cf s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
    cf r3 = DIR < 0 ? make_float2(s3.y, -s3.x) : make_float2(-s3.y, s3.x);
    a = cadd(s0, s2);
    c = csub(s0, s2);
    b = cadd(s1, r3);
    d = csub(s1, r3);
}

// multiply by exp(DIR * 2*pi*i * M / 16), M compile time
template <int DIR, int M>
SETK_DEV cf twid16(cf v) {
    constexpr float C1 = 0.92387953251128673848f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;
    constexpr float D = (float)DIR;
    if constexpr (M == 1) return cmul(v, make_float2(C1, D * S1));
    if constexpr (M == 2) return make_float2(R2 * (v.x - D * v.y), R2 * (D * v.x + v.y));
    if constexpr (M == 3) return cmul(v, make_float2(S1, D * C1));
    if constexpr (M == 4) return make_float2(-D * v.y, D * v.x);
    if constexpr (M == 6) return make_float2(R2 * (-v.x - D * v.y), R2 * (D * v.x - v.y));
    if constexpr (M == 9) return cmul(v, make_float2(-C1, -D * S1));
    return v;
}

// In-register 16-point DFT.  On return v[dft16_pos(q)] = X[q].
template <int DIR>
SETK_DEV void dft16(cf (&v)[16]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // v[4c+b] = u_b[c]; twiddle W16^{bc}
    v[4 * 1 + 1] = twid16<DIR, 1>(v[4 * 1 + 1]);
    v[4 * 2 + 1] = twid16<DIR, 2>(v[4 * 2 + 1]);
    v[4 * 3 + 1] = twid16<DIR, 3>(v[4 * 3 + 1]);
    v[4 * 1 + 2] = twid16<DIR, 2>(v[4 * 1 + 2]);
    v[4 * 2 + 2] = twid16<DIR, 4>(v[4 * 2 + 2]);
    v[4 * 3 + 2] = twid16<DIR, 6>(v[4 * 3 + 2]);
    v[4 * 1 + 3] = twid16<DIR, 3>(v[4 * 1 + 3]);
    v[4 * 2 + 3] = twid16<DIR, 6>(v[4 * 2 + 3]);
    v[4 * 3 + 3] = twid16<DIR, 9>(v[4 * 3 + 3]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        dft4<DIR>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

// ---- 256-point complex transform on 16 lanes ------------------------------
// Stage A: lane `la` holds v[j] = z[la + 16 j].  Radix-16 over j, twiddle by
// W256^{la q}, store transposed into the LDS slot.  Stage B (no workgroup
// barrier: a quad-row lives in one wavefront, whose LDS operations complete in
// order): lane `la` gathers sub-transform `la`, radix-16, and holds
// Z[la + 16 kb] in v[dft16_pos(kb)].
//
// The 16x16 exchange uses a padded row stride (an XOR swizzle was measured
// first: it costs ~30 loop-invariant address registers), so every LDS address is
// one base register plus an immediate.  The stride is 18 entries = 144 bytes: rows
// stay 16-byte aligned, so a lane fetches its 16 row entries with eight
// ds_read_b128 (256 B/clk/CU; the compiler otherwise pairs 8-byte reads into
// ds_read2_b64 at 128 B/clk), and the sixteen rows of a 16-lane LDS group start
// 36 dwords apart = sixteen distinct multiples of 4 mod 64: conflict free.  The
// per-lane tables (window, twiddles) use the same row form.
// ROW = 18 where the register budget allows the 4-aligned quads of b128 loads
// (pass 2, 256 VGPRs: 0.92 -> 0.84 ms); ROW = 17 (8-byte reads, nothing for the
// compiler to widen) at the 128-VGPR budget of pass 1, where the wide form
// spills (0.90 -> 1.14 ms).
constexpr int kRow5 = 10;             // row stride of the 8-entry split-twiddle rows
__host__ __device__ constexpr int slot_entries(int row) { return 16 * row; }  // per transform slot
// Tables: for even ROW a lane's values form a contiguous 16-byte aligned row
// (entry stride 1, rows ROW / kRow5 apart); for odd ROW they keep the natural
// order (entry stride 16, lane la starts at entry la), 256 + 256 + 128 entries.
__host__ __device__ constexpr int table_entries(int row) {
    return (row % 2 == 0) ? 2 * 16 * row + 16 * kRow5 : 640;
}
template <int ROW> struct LaneTab {
    static constexpr bool rows = (ROW % 2 == 0);
    static constexpr int estride = rows ? 1 : 16;       // between a lane's consecutive entries
    static constexpr int lstride = rows ? ROW : 1;      // between lanes (window, twiddle)
    static constexpr int lstride5 = rows ? kRow5 : 1;   // between lanes (split twiddle)
    static constexpr int size = rows ? 16 * ROW : 256;  // entries of the window / twiddle table
};

// 16 (or N) consecutive complex entries of a 16-byte aligned LDS row.  WIDE:
// ds_read_b128 (needs 4-aligned register quads: fine at 256 VGPRs, measured
// slower at the 128-VGPR budget of pass 1, which reads 8 bytes at a time).
template <int N, bool WIDE = true>
SETK_DEV void lds_row(const cf* row, cf (&out)[N]) {
    if (WIDE) {
        const float4* r4 = reinterpret_cast<const float4*>(__builtin_assume_aligned(row, 16));
#pragma unroll
        for (int n = 0; n < N / 2; ++n) {
            const float4 t = r4[n];
            out[2 * n] = make_float2(t.x, t.y);
            out[2 * n + 1] = make_float2(t.z, t.w);
        }
    } else {
#pragma unroll
        for (int n = 0; n < N; ++n) out[n] = row[n];
    }
}

// Per-lane table rows in LDS (filled once per workgroup):
//   win_l [16][ROW]   row la, entry j: (w[2n], w[2n+1]), n = la + 16 j
//   tw_l  [16][ROW]   row la, entry q: exp(-2 pi i la q / 256)
//   tw5_l [16][kRow5] row la, entry m: exp(-2 pi i (la + 16 m) / 512)
template <int ROW>
SETK_DEV void fill_lane_tables(cf* win_l, cf* tw_l, cf* tw5_l, const float* window,
                               const float2* tw256, const float2* tw512, int tid, int nthreads) {
    for (int i = tid; i < 256; i += nthreads) {
        const int la = i & 15, j = i >> 4, n = la + 16 * j;
        typedef LaneTab<ROW> L;
        win_l[la * L::lstride + j * L::estride] = make_float2(window[2 * n], window[2 * n + 1]);
        tw_l[la * L::lstride + j * L::estride] = tw256[j * 16 + la];
    }
    for (int i = tid; i < 128; i += nthreads) {
        const int la = i & 15, m = i >> 4;
        tw5_l[la * LaneTab<ROW>::lstride5 + m * LaneTab<ROW>::estride] = tw512[la + 16 * m];
    }
}

// v[j] *= window row entry j (tw_row-style row pointer of this lane)
template <int ROW>
SETK_DEV void apply_window(cf (&v)[16], const cf* win_row) {
    if (LaneTab<ROW>::rows) {
        const float4* w4 = reinterpret_cast<const float4*>(__builtin_assume_aligned(win_row, 16));
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const float4 w = w4[j / 2];
            v[j] = make_float2(v[j].x * w.x, v[j].y * w.y);
            v[j + 1] = make_float2(v[j + 1].x * w.z, v[j + 1].y * w.w);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const cf w = win_row[j * 16];
            v[j] = make_float2(v[j].x * w.x, v[j].y * w.y);
        }
    }
}

template <int DIR, int ROW, bool WIDE = (ROW % 2 == 0)>
SETK_DEV void fft256_stage_a_pad(cf (&v)[16], cf* slot, const cf* tw_row, int la) {
    dft16<DIR>(v);
    cf* dst = slot + la;
    if (WIDE) {
        const float4* w4 = reinterpret_cast<const float4*>(__builtin_assume_aligned(tw_row, 16));
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const float4 w = w4[q / 2];  // entries q, q + 1
            cf t0 = v[dft16_pos(q)], t1 = v[dft16_pos(q + 1)];
            if (q) t0 = cmul(t0, make_float2(w.x, DIR > 0 ? -w.y : w.y));
            t1 = cmul(t1, make_float2(w.z, DIR > 0 ? -w.w : w.w));
            dst[q * ROW] = t0;
            dst[(q + 1) * ROW] = t1;
        }
    } else {
        // twiddles fetched two steps ahead (at a tight register budget the compiler
        // otherwise loads each one right before its use and waits out the LDS
        // latency fifteen times)
        cf wq[16];
        wq[1] = tw_row[16];
        wq[2] = tw_row[32];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q + 3 < 16) wq[q + 3] = tw_row[(q + 3) * 16];
            cf t = v[dft16_pos(q)];
            if (q) {
                cf w = wq[q];
                if (DIR > 0) w.y = -w.y;
                t = cmul(t, w);
            }
            dst[q * ROW] = t;
        }
    }
}

template <int DIR, int ROW, bool WIDE = (ROW % 2 == 0)>
SETK_DEV void fft256_stage_b_pad(cf (&v)[16], const cf* slot, int la) {
    lds_row<16, WIDE>(slot + la * ROW, v);
    dft16<DIR>(v);
}

// Hermitian split of the packed transform: from Zk = Z[k], Zm = Z[256-k]
// (Z[256] == Z[0]) produce X[k] and X[256-k] of the 512-point real DFT.
// w = exp(-2 pi i k / 512).  The 1/2 of E = (Zk + conj Zm)/2, O = (Zk - conj Zm)/2i
// is NOT applied here: the analysis window table is pre-scaled by 0.5 (exact in
// binary floating point), which saves four multiplies per bin pair.  The
// self-paired bin 128, X[128] = conj(Z[128]), must therefore be doubled by the
// caller.
SETK_DEV void rfft_split(cf Zk, cf Zm, cf w, cf& Xk, cf& Xm) {
    cf A = make_float2(Zk.x + Zm.x, Zk.y - Zm.y);  // 2 E[k]
    cf O = make_float2(Zk.y + Zm.y, Zm.x - Zk.x);  // 2 O[k]
    cf B = cmul(w, O);
    Xk = cadd(A, B);
    Xm = make_float2(A.x - B.x, -(A.y - B.y));
}

// Inverse of rfft_split: from Y[k], Y[256-k] build 2 Z[k], 2 Z[256-k] of the
// packed inverse transform (the 1/2 is folded into the half-scaled synthesis
// window).  w = exp(-2 pi i k / 512) (conjugated inside).
SETK_DEV void irfft_merge(cf Yk, cf Ym, cf w, cf& Zk, cf& Zm) {
    cf E = make_float2(Yk.x + Ym.x, Yk.y - Ym.y);
    cf D = make_float2(Yk.x - Ym.x, Yk.y + Ym.y);
    cf O = cmulc(D, w);  // D * conj(w)
    Zk = make_float2(E.x - O.y, E.y + O.x);   // E + i O
    Zm = make_float2(E.x + O.y, -E.y + O.x);  // conj(E) + i conj(O)
}

// ---- frame loading and the quad-row real transform (both streaming passes) ----
constexpr int kFrame = 512;

// numpy "reflect" index (no edge repeat); valid for -N < i < 2N-1
SETK_DEV int reflect_index(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// raw (un-windowed) frame points: v[j] = (x[s+2n], x[s+2n+1]), n = la + 16 j.
// One source for both streaming passes; the pointer type selects the address
// space: pass 1 states global memory (gcfloat_p: global_load, its waits do not
// also wait for LDS), pass 2 keeps the generic pointer (its schedule was tuned
// on that form: the global form measured 1.02 ms against 0.91).
template <class FloatPtr> struct RawPair;
template <> struct RawPair<gcfloat_p> {
    typedef gcfloat2_p ptr;
    static SETK_DEV cf get(ptr p, int i) {
        const v2f d = p[i];
        return make_float2(d.x, d.y);
    }
};
template <> struct RawPair<const float*> {
    typedef const float2* ptr;
    static SETK_DEV cf get(ptr p, int i) { return p[i]; }
};

template <class FloatPtr>
SETK_DEV void load_raw(cf (&v)[16], FloatPtr x, int n_samp, int s, int la, bool valid) {
    if (!valid) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = make_float2(0.f, 0.f);
        return;
    }
    const bool interior = (s >= 0) && (s + kFrame <= n_samp) && ((((uintptr_t)(x + s)) & 7) == 0);
    if (interior) {
        typename RawPair<FloatPtr>::ptr p = (typename RawPair<FloatPtr>::ptr)(x + s);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = RawPair<FloatPtr>::get(p, la + 16 * j);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = la + 16 * j;
            v[j] = make_float2(x[reflect_index(s + 2 * n, n_samp)],
                               x[reflect_index(s + 2 * n + 1, n_samp)]);
        }
    }
}

// 16-bit PCM (planar int16, UttDesc::audio_fmt == kAudioPcm16): the pair (x[s+2n], x[s+2n+1]) is
// ONE dword, kept packed while in flight (16 registers per frame where the float form holds
// 32) and unpacked by two SDWA conversions (v_cvt_f32_i32 sext word_0 / word_1) when the
// transform starts.  The 2^-15 of read_wav (libs/utils.py:80-90: int16 / 32768 as float32) is
// folded into the window table -- a power of two, so every product is bit for bit that of the
// float path on the dequantised samples.
typedef const SETK_GLOBAL short* gcshort_p;
typedef const SETK_GLOBAL int* gcint_p;
SETK_DEV void load_raw_pcm(int (&v)[16], gcshort_p x, int n_samp, int s, int la, bool valid) {
    if (!valid) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = 0;
        return;
    }
    // (x is 4-byte aligned and s even by construction: channel strides are even, hop and pad too)
    const bool interior = (s >= 0) && (s + kFrame <= n_samp) && ((((uintptr_t)(x + s)) & 3) == 0);
    if (interior) {
        gcint_p p = (gcint_p)(x + s);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[la + 16 * j];
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = la + 16 * j;
            const int lo = x[reflect_index(s + 2 * n, n_samp)];
            const int hi = x[reflect_index(s + 2 * n + 1, n_samp)];
            v[j] = (lo & 0xffff) | (hi << 16);
        }
    }
}
SETK_DEV cf unpack_pcm(int v) { return make_float2((float)(short)(v & 0xffff), (float)(v >> 16)); }

// lane la <- value of lane (16 - la) & 15 of the same quad-row (a DPP row):
// row_mirror (la -> 15 - la) followed by row_ror:1
SETK_DEV float qr_partner(float x) {
    int v = __builtin_bit_cast(int, x);
    v = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);  // row_mirror
    v = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, true);  // row_ror:1
    return __builtin_bit_cast(float, v);
}

// Second radix-16 and the Hermitian split, in registers: lane la owns
// Z[la + 16 kb]; the mirror bin 256 - k of k = la + 16 m lives in lane
// (16 - la) & 15, register 15 - m (lane 0: its own register 16 - m), fetched with
// two DPP moves instead of an LDS round trip.  Writes X[k], X[256-k] (m < 8) into
// the slot and X[256] to *nyq.
template <int ROW, bool WIDE = (ROW % 2 == 0)>
SETK_DEV void qr_stage23(cf* slot, float* nyq, const cf* tw5_row, int la) {
    cf v[16];
    fft256_stage_b_pad<-1, ROW, WIDE>(v, slot, la);
    const bool lane0 = (la == 0);
    // bins k = la + 16 m and 256 - k as ONE base register each plus an immediate
    // (written as slot[256 - k] the compiler keeps eight address registers)
    cf* lo = slot + la;
    cf* mir = slot + (256 - 16 * 7) - la;
    cf t5[8];
    if (WIDE) {
        lds_row<8, true>(tw5_row, t5);
    } else {
        // narrow path: fetched two steps ahead of their use (see fft256_stage_a_pad)
        t5[0] = tw5_row[0];
        t5[1] = tw5_row[16];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (!WIDE && m + 2 < 8) t5[m + 2] = tw5_row[16 * (m + 2)];
        const cf Zk = v[dft16_pos(m)];
        const cf src = v[dft16_pos(15 - m)];
        cf Zm = make_float2(qr_partner(src.x), qr_partner(src.y));
        const cf own = v[dft16_pos((16 - m) & 15)];
        Zm.x = lane0 ? own.x : Zm.x;
        Zm.y = lane0 ? own.y : Zm.y;
        cf Xk, Xm;
        rfft_split(Zk, Zm, t5[m], Xk, Xm);
        if (m == 0) {
            // lane 0: k = 0 pairs with itself (Z[256] == Z[0]): X[0], X[256]; and
            // the self-paired bin 128 = conj(Z[128])
            const cf Z128 = v[dft16_pos(8)];
            if (lane0) {
                slot[0] = make_float2(Xk.x, 0.f);
                *nyq = Xm.x;
                slot[128] = make_float2(2.f * Z128.x, -2.f * Z128.y);
            } else {
                lo[16 * m] = Xk;
                mir[16 * (7 - m)] = Xm;
            }
        } else {
            lo[16 * m] = Xk;
            mir[16 * (7 - m)] = Xm;
        }
    }
}

}  // namespace setk
