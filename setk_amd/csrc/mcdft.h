// mcdft.h -- the 512-point real DFT (and its inverse) on the matrix cores of gfx950.
//
// Replaces the arithmetic of librosa.stft / librosa.istft behind forward_stft /
// inverse_stft (funcwj/setk libs/utils.py:96-173) inside the fused streaming kernels.
//
// The butterflies of a radix-32 x radix-16 factorisation ARE dense contractions:
//     n = 16 n1 + n2,  k = k1 + 32 q
//     stage 1   A[k1][n2] = sum_n1 xw[16 n1 + n2] W32^(n1 k1)      real input: k1 = 0..16
//     twiddle   B[k1][n2] = A[k1][n2] W512^(n2 k1)
//     stage 2   Z[k1][q]  = sum_n2 B[k1][n2] W16^(n2 q)
//     bins      X[k1 + 32 q] = Z[k1][q] (q < 8),  X[32 (16 - q) - k1] = conj Z[k1][q] (q >= 8; the
//               conjugation is folded into the stage-2 tile, the lanes hold X itself; the rows
//               q >= 8 run backwards inside each group of four so that a lane's four bins
//               ascend by 32 in every lane: one base address + immediates)
//               X[32 q] = Z[0][q] (column 0 carries the real A[0][.]),
//               X[16 + 32 q] = sum_n2 A[16][n2] W32^(n2 (2 q + 1))  ("odd family", one extra
//               tile per 16 transforms)
// Both stages run as v_mfma_f32_16x16x32_f16 with every fp32 operand split into an fp16
// pair (hi, lo) and every product taken as hi*hi + lo*hi + hi*lo: 22 significant bits, fp32
// accumulation -- measured 1.0e-7 relative RMS against a float64 DFT (the fp32 butterfly
// kernels of fft512.h: 0.6e-7).  ONE wavefront owns one transform: 8 samples per lane in,
// 4 complex bins per lane out, no LDS exchange and no cross-lane moves inside a transform
// (the matrix instruction does the data movement), ~60 VALU + 12 MFMA wave-instructions per
// transform against 646 VALU per quad-row (161 per transform) for fft512.h.  The fp16
// matrix pipe runs beside the vector ALUs, so the transform costs the SIMD ~1/3 of the
// issue slots it used to.
//
// Register layouts of v_mfma_f32_16x16x32_f16 (lane l, g = l / 16):
//     A: A[l % 16][8 g + e], e < 8      B: B[8 g + e][l % 16]      D: D[4 g + r][l % 16], r < 4
// so a result tile can feed the next contraction as its A or B operand without moving: the
// contracted index only has to sit in (g, register).  tests/mcdft_model.py is the lane-level
// numpy model this file was written from; tools/ubench/mcdft_probe.hip checks layouts,
// accuracy and rate on the device.
//
// Range: operands must stay inside fp16 (65504).  Forward: the caller scales the window
// table by 2^10 / peak (|x| <= peak) and divides the result (or what it accumulates from
// it) by the same power of two; a stage-1 sum is <= 32 * 2^10.  Inverse: a per-frame power
// of two brings max |Y| below 2^11 (inverse()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace setk {
namespace mc {

#define MC_DEV __device__ __forceinline__

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// ---- constant table (mcdft_tables.h builds it on the host): 32-bit words, [word][lane] ----
enum : int {
    kW_MC_H = 0, kW_MC_L = 4, kW_MS_H = 8, kW_MS_L = 12,     // stage-1 B operands
    kW_AR_H = 16, kW_AR_L = 20, kW_AI_H = 24, kW_AI_L = 28,  // stage-2 A operands
    kW_TR = 32, kW_TI = 36, kW_TRI = 40,                     // W512^((l%16) (4g+r)); TRI: 0 in column 0
    kW_OT_H = 44, kW_OT_L = 48,                              // odd-family tile (A operand)
    kW_BR_H = 52, kW_BR_L = 56, kW_BI_H = 60, kW_BI_L = 64,  // inverse stage over q (B operands)
    kW_G0_H = 68, kW_G0_L = 72, kW_G1_H = 76, kW_G1_L = 80,  // inverse stage over k1 (A operands, rows n1)
    kW_IT_H = 84, kW_IT_L = 88,                              // inverse odd-family tile (B operand)
    kTabWords = 92
};

// Sample of register e of lane l in the stage-1 data operand: K index k = 8 g + e <-> n1 =
// 4 g + e (e < 4: first half of the frame), 16 + 4 g + e - 4 (e >= 4: second half); the
// contraction order is free as long as the tiles follow it (mcdft_tables.h).  With hop = 256
// a lane's registers e >= 4 of frame t ARE its registers e < 4 of frame t + 1: a wave that
// walks consecutive frames of a channel loads four new samples per lane and frame, not eight.
__host__ __device__ constexpr int stage1_n1(int k) {
    return (k % 8 < 4) ? 4 * (k / 8) + k % 8 : 16 + 4 * (k / 8) + k % 8 - 4;
}
__host__ __device__ constexpr int sample_of(int lane, int e) { return 16 * stage1_n1(8 * (lane >> 4) + e) + (lane & 15); }

MC_DEV h8 tab_h8(const unsigned* tab, int word, int lane) {
    u4 w;
    w[0] = tab[(word + 0) * 64 + lane];
    w[1] = tab[(word + 1) * 64 + lane];
    w[2] = tab[(word + 2) * 64 + lane];
    w[3] = tab[(word + 3) * 64 + lane];
    return __builtin_bit_cast(h8, w);
}
MC_DEV float tab_f(const unsigned* tab, int word, int lane) {
    return __builtin_bit_cast(float, tab[word * 64 + lane]);
}

MC_DEV f4 mfma16(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// v - (float)h as ONE v_fma_mix_f32 (v * 1.0 - h, the fp16 operand read from either half of
// its register); the compiler only forms it when v itself is a product (the windowed samples).
// Single-instruction statements with declared operands: the results feed compiler-issued
// v_cvt_pk_f16_f32, so every MFMA hazard stays visible to hipcc.
#ifndef MCDFT_ASM_MIX
#define MCDFT_ASM_MIX 1
#endif
MC_DEV void sub_h2(float a, float b, h2 p, float& la, float& lb) {
#if MCDFT_ASM_MIX
    const unsigned w = __builtin_bit_cast(unsigned, p);
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(la) : "v"(a), "v"(w));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(b), "v"(w));
#else
    la = a - (float)p[0];
    lb = b - (float)p[1];
#endif
}

// fp32 x 8 -> fp16 pair of vectors: hi = rne16(v), lo = rne16(v - hi)
MC_DEV void split8(const float (&v)[8], h8& hi, h8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const h2 p = __builtin_convertvector((f2){v[e], v[e + 1]}, h2);
        hi[e] = p[0];
        hi[e + 1] = p[1];
        float la, lb;
        sub_h2(v[e], v[e + 1], p, la, lb);
        const h2 q = __builtin_convertvector((f2){la, lb}, h2);
        lo[e] = q[0];
        lo[e + 1] = q[1];
    }
}
// the two halves of split8 as separate steps (hi first: see forward()).
// MCDFT_ASM_SPLIT: v_fma_mixlo_f16 / v_fma_mixhi_f16 round an fp32 fused multiply-add straight
// into one half of a packed register -- the hi half of x * w in ONE instruction per element (no
// separate multiply, no v_cvt_pk), the lo half fma(x, w, -hi) in one more: 16 instructions per
// eight windowed samples where the compiler's form takes 24, 12 instead of 16 for the
// mid-transform split.  Each block ends in `s_nop 1`: its results feed an MFMA, and hipcc pads
// no hazards for instructions inside an asm statement (cdna_hip_programming.md section 5.7).
// Measured (8-ch, 125 x 30 s): pass 2 0.745 ms with the asm blocks against 0.70 ms with the
// compiler's own selection (it already forms v_fma_mix_f32 for the windowed samples and
// schedules freely around single instructions; the blocks pin 9 instructions and their nop):
// off by default, kept for the record.
#ifndef MCDFT_ASM_SPLIT
#define MCDFT_ASM_SPLIT 0
#endif
MC_DEV void split8_hi(const float (&v)[8], h8& hi) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const h2 p = __builtin_convertvector((f2){v[e], v[e + 1]}, h2);
        hi[e] = p[0];
        hi[e + 1] = p[1];
    }
}
MC_DEV void split8_lo(const float (&v)[8], h8 hi, h8& lo) {
#if MCDFT_ASM_SPLIT
    const u4 h = __builtin_bit_cast(u4, hi);
    unsigned l0, l1, l2, l3;
    asm("v_fma_mixlo_f16 %0, %4, 1.0, -%12 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %5, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %6, 1.0, -%13 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %7, 1.0, -%13 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %2, %8, 1.0, -%14 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %9, 1.0, -%14 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %10, 1.0, -%15 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %11, 1.0, -%15 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"
        : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
          "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
    lo = __builtin_bit_cast(h8, (u4){l0, l1, l2, l3});
#else
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        float la, lb;
        sub_h2(v[e], v[e + 1], (h2){hi[e], hi[e + 1]}, la, lb);
        const h2 q = __builtin_convertvector((f2){la, lb}, h2);
        lo[e] = q[0];
        lo[e + 1] = q[1];
    }
#endif
}
MC_DEV void split8_mul_hi(const float (&v)[8], const float (&w)[8], h8& hi) {
#if MCDFT_ASM_SPLIT
    unsigned h0, h1, h2_, h3;
    asm("v_fma_mixlo_f16 %0, %4, %12, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %13, 0\n\t"
        "v_fma_mixlo_f16 %1, %6, %14, 0\n\t"
        "v_fma_mixhi_f16 %1, %7, %15, 0\n\t"
        "v_fma_mixlo_f16 %2, %8, %16, 0\n\t"
        "v_fma_mixhi_f16 %2, %9, %17, 0\n\t"
        "v_fma_mixlo_f16 %3, %10, %18, 0\n\t"
        "v_fma_mixhi_f16 %3, %11, %19, 0\n\t"
        "s_nop 1"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2_), "=&v"(h3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
          "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
    hi = __builtin_bit_cast(h8, (u4){h0, h1, h2_, h3});
#else
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const h2 p = __builtin_convertvector((f2){v[e] * w[e], v[e + 1] * w[e + 1]}, h2);
        hi[e] = p[0];
        hi[e + 1] = p[1];
    }
#endif
}
MC_DEV void split8_mul_lo(const float (&v)[8], const float (&w)[8], h8 hi, h8& lo) {
#if MCDFT_ASM_SPLIT
    const u4 h = __builtin_bit_cast(u4, hi);
    unsigned l0, l1, l2, l3;
    asm("v_fma_mixlo_f16 %0, %4, %12, -%20 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %5, %13, -%20 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %6, %14, -%21 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %7, %15, -%21 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %2, %8, %16, -%22 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %9, %17, -%22 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %10, %18, -%23 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %11, %19, -%23 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"
        : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
          "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
          "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
    lo = __builtin_bit_cast(h8, (u4){l0, l1, l2, l3});
#else
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const h2 q = __builtin_convertvector(
            (f2){__builtin_fmaf(v[e], w[e], -(float)hi[e]), __builtin_fmaf(v[e + 1], w[e + 1], -(float)hi[e + 1])}, h2);
        lo[e] = q[0];
        lo[e + 1] = q[1];
    }
#endif
}
// the same for products v[e] * w[e] (window): lo = fma(v, w, -hi) keeps the product's own
// rounding error too
MC_DEV void split8_mul(const float (&v)[8], const float (&w)[8], h8& hi, h8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const h2 p = __builtin_convertvector((f2){v[e] * w[e], v[e + 1] * w[e + 1]}, h2);
        hi[e] = p[0];
        hi[e + 1] = p[1];
        const h2 q = __builtin_convertvector(
            (f2){__builtin_fmaf(v[e], w[e], -(float)p[0]), __builtin_fmaf(v[e + 1], w[e + 1], -(float)p[1])}, h2);
        lo[e] = q[0];
        lo[e + 1] = q[1];
    }
}

// hi*hi + lo*hi + hi*lo with the DATA as the A operand / as the B operand
MC_DEV f4 mm3_data_a(h8 dh, h8 dl, h8 kh, h8 kl) {
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = mfma16(dh, kh, d);
    d = mfma16(dl, kh, d);
    d = mfma16(dh, kl, d);
    return d;
}
MC_DEV f4 mm3_data_b(h8 kh, h8 kl, h8 dh, h8 dl) {
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = mfma16(kh, dh, d);
    d = mfma16(kh, dl, d);
    d = mfma16(kl, dh, d);
    return d;
}

// ---- forward ----
struct Fwd {
    h8 mc_h, mc_l, ms_h, ms_l;  // stage 1
    h8 ar_h, ar_l, ai_h, ai_l;  // stage 2
    float tr[4], ti[4], tri[4];
};
MC_DEV void load_fwd(Fwd& K, const unsigned* tab, int lane) {
    K.mc_h = tab_h8(tab, kW_MC_H, lane);
    K.mc_l = tab_h8(tab, kW_MC_L, lane);
    K.ms_h = tab_h8(tab, kW_MS_H, lane);
    K.ms_l = tab_h8(tab, kW_MS_L, lane);
    K.ar_h = tab_h8(tab, kW_AR_H, lane);
    K.ar_l = tab_h8(tab, kW_AR_L, lane);
    K.ai_h = tab_h8(tab, kW_AI_H, lane);
    K.ai_l = tab_h8(tab, kW_AI_L, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        K.tr[r] = tab_f(tab, kW_TR + r, lane);
        K.ti[r] = tab_f(tab, kW_TI + r, lane);
        K.tri[r] = tab_f(tab, kW_TRI + r, lane);
    }
}

// x[e] = sample sample_of(l, e) of the frame, w[e] its window value (x the range scale).
// Out: lane (c = l % 16, g), register r:  z = X[bin_of(c, g, r)]
//   = X[c + 32 (4 g + r)] for g < 2,  X[32 (13 - 4 g + r) - c] for g >= 2;
//   column 0 is valid for g < 2 and (g, r) = (2, 3) (bin 256); its other registers repeat
//   bins 32 .. 224 from the conjugate side
//   a16[r] (lanes c == 0 only) = A[16][n2 = 4 g + r], the input of the odd-family tile.
// The operand tiles come from `tile(i)`, i = 0..7: mc_h mc_l ms_h ms_l ar_h ar_l ai_h ai_l --
// registers (Fwd) or LDS (stage_tiles + lds_h8: four waves' worth of registers for 8 x
// ds_read_b128 per transform).
template <class TileFn>
MC_DEV void forward_t(const float (&x)[8], const float (&w)[8], TileFn tile, const float (&tr)[4],
                      const float (&ti)[4], const float (&tri)[4], f4& zr, f4& zi, f4& a16) {
    h8 xh, xl;
    // the products of the hi halves go first: the matrix pipe starts while the lo halves
    // are still being formed on the vector ALU
    split8_mul_hi(x, w, xh);
    f4 dc = {0.f, 0.f, 0.f, 0.f}, ds = {0.f, 0.f, 0.f, 0.f};
    {
        const h8 mc_h = tile(0), ms_h = tile(2);
        dc = mfma16(xh, mc_h, dc);
        ds = mfma16(xh, ms_h, ds);
        dc = mfma16(xh, tile(1), dc);
        ds = mfma16(xh, tile(3), ds);
        split8_mul_lo(x, w, xh, xl);
        dc = mfma16(xl, mc_h, dc);
        ds = mfma16(xl, ms_h, ds);
    }
    a16 = ds;
    float b[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        b[r] = fmaf(dc[r], tr[r], -ds[r] * ti[r]);
        b[4 + r] = fmaf(dc[r], ti[r], ds[r] * tri[r]);
    }
    h8 bh, bl;
    split8_hi(b, bh);
    zr = (f4){0.f, 0.f, 0.f, 0.f};
    zi = (f4){0.f, 0.f, 0.f, 0.f};
    {
        const h8 ar_h = tile(4), ai_h = tile(6);
        zr = mfma16(ar_h, bh, zr);
        zi = mfma16(ai_h, bh, zi);
        zr = mfma16(tile(5), bh, zr);
        zi = mfma16(tile(7), bh, zi);
        split8_lo(b, bh, bl);
        zr = mfma16(ar_h, bl, zr);
        zi = mfma16(ai_h, bl, zi);
    }
}
MC_DEV void forward(const float (&x)[8], const float (&w)[8], const Fwd& K, f4& zr, f4& zi, f4& a16) {
    forward_t(x, w, [&](int i) -> h8 {
        switch (i) {
            case 0: return K.mc_h;
            case 1: return K.mc_l;
            case 2: return K.ms_h;
            case 3: return K.ms_l;
            case 4: return K.ar_h;
            case 5: return K.ar_l;
            case 6: return K.ai_h;
            default: return K.ai_l;
        }
    }, K.tr, K.ti, K.tri, zr, zi, a16);
}

// bin of register r of this lane (c = l % 16, g = l / 16)
MC_DEV int bin_of(int c, int g, int r) { return (g < 2) ? c + 32 * (4 * g + r) : 32 * (13 - 4 * g + r) - c; }
MC_DEV bool bin_valid(int c, int g, int r) { return c != 0 || g < 2 || (g == 2 && r == 3); }

// The four bins of a lane into an LDS spectrum slot (complex64, bin fastest): re = the float
// address of bin_of(c, g, 0), im = opaque_next(re) -- the same address + 4 bytes, hidden from
// the compiler so that it emits four ds_write2_b32 (two registers, two offsets each) instead
// of assembling (re, im) register pairs with v_mov for ds_write_b64.
typedef __attribute__((address_space(3))) float* lds_fp;  // a 32-bit LDS address: ds_* for sure
MC_DEV lds_fp to_lds(float* p) { return (lds_fp)p; }
MC_DEV lds_fp opaque_next(lds_fp p) {
    lds_fp q = p + 1;
    asm volatile("" : "+v"(q));
    return q;
}
MC_DEV void store_bins(lds_fp re, lds_fp im, f4 zr, f4 zi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) re[64 * r] = zr[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) im[64 * r] = zi[r];
}

// Odd family X[16 + 32 q], q < 8, of SIXTEEN transforms at once.  a16s: this wave's scratch
// [16][kOddPitch] floats (row j = transform j of the batch, entry n2), written by the lanes
// c == 0 of each transform (store_a16); ot_h / ot_l = tab_h8(tab, kW_OT_H / kW_OT_L, lane).  Out: lane (j = l % 16, g): X_j[16 + 32 (2 g)] =
// (d[0], d[1]),  X_j[16 + 32 (2 g + 1)] = (d[2], d[3]).
constexpr int kOddPitch = 20;  // floats per scratch row: 16-byte aligned, conflict-free b128 reads
MC_DEV void store_a16(float* a16s, int j, int lane, f4 a16) {
    if ((lane & 15) == 0) *reinterpret_cast<f4*>(a16s + j * kOddPitch + (lane >> 4) * 4) = a16;
}
MC_DEV f4 odd_tile(const float* a16s, h8 ot_h, h8 ot_l, int lane, int nrows = 16) {
    const int g = lane >> 4;
    const int j = (lane & 15) < nrows ? (lane & 15) : nrows - 1;  // unused columns repeat a valid row
    float v[8];
    const f4 v0 = *reinterpret_cast<const f4*>(a16s + j * kOddPitch + 8 * (g & 1));
    const f4 v1 = *reinterpret_cast<const f4*>(a16s + j * kOddPitch + 8 * (g & 1) + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = v0[e];
        v[4 + e] = v1[e];
    }
    h8 h, l;
    split8(v, h, l);
    const h8 bop = (g < 2) ? h : l;  // K rows 0..15: hi(A16), 16..31: lo(A16)
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = mfma16(ot_h, bop, d);
    d = mfma16(ot_l, bop, d);  // T_lo x hi only (its K >= 16 half is zero)
    return d;
}

// ---- per-lane tiles staged in LDS ([tile][lane] of 16 bytes: one conflict-free ds_read_b128) ----
// for the constants a kernel needs once per frame and cannot afford to keep in registers
MC_DEV void stage_tiles(u4* lds, const unsigned* tab, int first_word, int ntiles, int tid, int nthreads) {
    for (int i = tid; i < ntiles * 64; i += nthreads) {
        const int t = i >> 6, l = i & 63;
        u4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = tab[(first_word + 4 * t + k) * 64 + l];
        lds[i] = w;
    }
}
MC_DEV h8 lds_h8(const u4* lds, int tile, int lane) { return __builtin_bit_cast(h8, lds[tile * 64 + lane]); }

// ---- inverse ----
struct Inv {
    h8 br_h, br_l, bi_h, bi_l;
    h8 g0_h, g0_l, g1_h, g1_l;
    float tr[4], ti[4];
};
MC_DEV void load_inv(Inv& K, const unsigned* tab, int lane) {
    K.br_h = tab_h8(tab, kW_BR_H, lane);
    K.br_l = tab_h8(tab, kW_BR_L, lane);
    K.bi_h = tab_h8(tab, kW_BI_H, lane);
    K.bi_l = tab_h8(tab, kW_BI_L, lane);
    K.g0_h = tab_h8(tab, kW_G0_H, lane);
    K.g0_l = tab_h8(tab, kW_G0_L, lane);
    K.g1_h = tab_h8(tab, kW_G1_H, lane);
    K.g1_l = tab_h8(tab, kW_G1_L, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        K.tr[r] = tab_f(tab, kW_TR + r, lane);
        K.ti[r] = tab_f(tab, kW_TI + r, lane);
    }
}

// In: the spectrum in the forward's output form (lane (k1 = l % 16, g), register r: Y[bin_of],
// column 0 included: all sixteen entries; Im Y[0] = Im Y[256] = 0), already range-scaled
// (|.| < 2^11);  e16 (lanes g == 0 use it): E16[n2 = l % 16] of the odd family.
// Out: y0[r] = y[16 (4 g + r) + l % 16], y1[r] = y[16 (16 + 4 g + r) + l % 16] (x 512, x scale).
// first half: the stage over q and the conjugate twiddle (tr, ti: the forward's rows)
MC_DEV void inverse_a(f4 yr, f4 yi, float e16, h8 br_h, h8 br_l, h8 bi_h, h8 bi_l, const float (&tr)[4],
                      const float (&ti)[4], float (&b)[8], int lane) {
    float a[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a[r] = yr[r];
        a[4 + r] = yi[r];
    }
    h8 ah, al;
    split8(a, ah, al);
    const f4 cr = mm3_data_a(ah, al, br_h, br_l);  // C[k1 = 4 g + r][n2 = l % 16]
    const f4 ci = mm3_data_a(ah, al, bi_h, bi_l);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // conj twiddle: (cr + i ci)(tr - i ti)
        b[r] = fmaf(cr[r], tr[r], ci[r] * ti[r]);
        b[4 + r] = fmaf(ci[r], tr[r], -cr[r] * ti[r]);
    }
    if (lane < 16) b[4] = e16;  // the (im, k1 = 0) slot carries E16
}
// second half: the stage over k1
MC_DEV void inverse_b(const float (&b)[8], h8 g0_h, h8 g0_l, h8 g1_h, h8 g1_l, f4& y0, f4& y1) {
    h8 bh, bl;
    split8(b, bh, bl);
    y0 = mm3_data_b(g0_h, g0_l, bh, bl);
    y1 = mm3_data_b(g1_h, g1_l, bh, bl);
}
MC_DEV void inverse(f4 yr, f4 yi, float e16, const Inv& K, f4& y0, f4& y1, int lane) {
    float b[8];
    inverse_a(yr, yi, e16, K.br_h, K.br_l, K.bi_h, K.bi_l, K.tr, K.ti, b, lane);
    inverse_b(b, K.g0_h, K.g0_l, K.g1_h, K.g1_l, y0, y1);
}

// E16[n2] of SIXTEEN frames: in lane (j = l % 16, g) v[e] = hi/lo source: the 8 odd-family
// bins of frame j as (re, im) pairs: lanes g < 2 hold (q = 4 (g & 1) + e / 2, part = e % 2) and
// feed hi, lanes g >= 2 the same values and feed lo.  Out: lane (n2 = l % 16, g), register r:
// E16 of frame j = 4 g + r.
MC_DEV f4 inv_odd_tile(const float (&v)[8], h8 it_h, h8 it_l, int lane) {
    h8 h, l;
    split8(v, h, l);
    const h8 aop = ((lane >> 4) < 2) ? h : l;
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = mfma16(aop, it_h, d);
    d = mfma16(aop, it_l, d);
    return d;
}

}  // namespace mc
}  // namespace setk
