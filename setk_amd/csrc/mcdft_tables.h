// mcdft_tables.h -- host side of mcdft.h: the per-lane constant table of the matrix-core
// DFT-512 (fp64 -> fp16 (hi, lo) pairs in the operand layouts of v_mfma_f32_16x16x32_f16)
// and the per-plan window rows.  tests/mcdft_model.py builds the same tiles in numpy.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "mcdft.h"

namespace setk {
namespace mc {

inline uint16_t half_bits(double x) {
    const _Float16 h = (_Float16)x;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
inline double half_value(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (double)h;
}

// tile(i, k): i = the index that lives in l % 16, k = the contracted index 8 g + e.
// Writes the hi words at `word_h` and the lo words at `word_l` (4 words each).
template <class Fn>
inline void put_tile(std::vector<uint32_t>& tab, int word_h, int word_l, Fn tile, bool lo_first16_only = false) {
    for (int l = 0; l < 64; ++l) {
        const int i = l & 15, g = l >> 4;
        for (int w = 0; w < 4; ++w) {
            uint32_t hw = 0, lw = 0;
            for (int s = 0; s < 2; ++s) {
                const int k = 8 * g + 2 * w + s;
                const double v = tile(i, k);
                const uint16_t hb = half_bits(v);
                uint16_t lb = half_bits(v - half_value(hb));
                if (lo_first16_only && k >= 16) lb = 0;
                hw |= (uint32_t)hb << (16 * s);
                lw |= (uint32_t)lb << (16 * s);
            }
            tab[(word_h + w) * 64 + l] = hw;
            tab[(word_l + w) * 64 + l] = lw;
        }
    }
}

inline std::vector<uint32_t> build_table() {
    const double PI = 3.14159265358979323846;
    std::vector<uint32_t> tab((size_t)kTabWords * 64, 0u);
    auto part_of = [](int k) { return (k % 8) / 4; };
    auto sub_of = [](int k) { return 4 * (k / 8) + k % 4; };  // n2 / row / k1 of K index k
    // row 4 g + r of a result tile <-> q: rows 8..15 run backwards inside each group of four, so
    // that the bins of a lane ascend with r in every lane (mcdft.h bin_of)
    auto row_q = [](int row) { return row < 8 ? row : 4 * (row / 4) + 3 - row % 4; };
    // stage 1 (B operands): i = column c, k <-> n1 by stage1_n1 (mcdft.h: consecutive frames
    // share half of a lane's sample registers)
    put_tile(tab, kW_MC_H, kW_MC_L, [&](int c, int k) { return std::cos(2 * PI * stage1_n1(k) * c / 32); });
    put_tile(tab, kW_MS_H, kW_MS_L, [&](int c, int k) {
        const int n1 = stage1_n1(k);
        return c == 0 ? ((n1 & 1) ? -1.0 : 1.0) : -std::sin(2 * PI * n1 * c / 32);
    });
    // stage 2 (A operands): i = row <-> q, k <-> (part, n2)
    put_tile(tab, kW_AR_H, kW_AR_L, [&](int row, int k) {
        const double a = 2 * PI * sub_of(k) * row_q(row) / 16;
        return part_of(k) == 0 ? std::cos(a) : std::sin(a);
    });
    // rows q >= 8 hold the conjugate bins: negated, so that every lane ends up with X itself
    put_tile(tab, kW_AI_H, kW_AI_L, [&](int row, int k) {
        const double a = 2 * PI * sub_of(k) * row_q(row) / 16, sg = row >= 8 ? -1.0 : 1.0;
        return sg * (part_of(k) == 0 ? -std::sin(a) : std::cos(a));
    });
    // twiddles W512^(a b), a = l % 16, b = 4 g + r
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int a = l & 15, b = 4 * (l >> 4) + r;
            const float tr = (float)std::cos(2 * PI * a * b / 512), ti = (float)-std::sin(2 * PI * a * b / 512);
            const float tri = a == 0 ? 0.f : tr;
            std::memcpy(&tab[(kW_TR + r) * 64 + l], &tr, 4);
            std::memcpy(&tab[(kW_TI + r) * 64 + l], &ti, 4);
            std::memcpy(&tab[(kW_TRI + r) * 64 + l], &tri, 4);
        }
    // odd-family tile (A operand): row 4 g' + r' <-> (q = 2 g' + r' / 2, part = r' % 2); K = [hi | lo] of n2
    put_tile(tab, kW_OT_H, kW_OT_L, [&](int row, int k) {
        const int q = 2 * (row / 4) + (row % 4) / 2, part = row % 2, n2 = k % 16;
        const double a = 2 * PI * n2 * (2 * q + 1) / 32;
        return part == 0 ? std::cos(a) : -std::sin(a);
    }, true);
    // inverse stage over q (B operands): i = column n2, k <-> (part, q)
    // (the data rows q >= 8 arrive as X, not conjugated: their imaginary parts enter negated)
    put_tile(tab, kW_BR_H, kW_BR_L, [&](int n2, int k) {
        const double a = 2 * PI * row_q(sub_of(k)) * n2 / 16;
        return part_of(k) == 0 ? std::cos(a) : (sub_of(k) >= 8 ? std::sin(a) : -std::sin(a));
    });
    put_tile(tab, kW_BI_H, kW_BI_L, [&](int n2, int k) {
        const double a = 2 * PI * row_q(sub_of(k)) * n2 / 16;
        return part_of(k) == 0 ? std::sin(a) : (sub_of(k) >= 8 ? -std::cos(a) : std::cos(a));
    });
    // inverse stage over k1 (A operands, two row tiles n1 = 16 tl + i): k <-> (part, k1)
    for (int tl = 0; tl < 2; ++tl)
        put_tile(tab, tl ? kW_G1_H : kW_G0_H, tl ? kW_G1_L : kW_G0_L, [&](int i, int k) {
            const int n1 = 16 * tl + i, k1 = sub_of(k);
            if (k1 == 0) return part_of(k) == 0 ? 1.0 : ((n1 & 1) ? -1.0 : 1.0);
            const double a = 2 * PI * n1 * k1 / 32;
            return part_of(k) == 0 ? 2 * std::cos(a) : -2 * std::sin(a);
        });
    // inverse odd-family tile (B operand): i = column n2, K = [hi | lo] of (q = k / 2, part = k % 2)
    put_tile(tab, kW_IT_H, kW_IT_L, [&](int n2, int k) {
        const int kk = k % 16, q = kk / 2, part = kk % 2;
        const double a = 2 * PI * n2 * (2 * q + 1) / 32;
        return part == 0 ? 2 * std::cos(a) : -2 * std::sin(a);
    }, true);
    return tab;
}

// analysis window rows: [8][64] floats, entry e of lane l = window[sample_of(l, e)] * scale
inline std::vector<float> build_window_rows(const float* window512, double scale) {
    std::vector<float> w(8 * 64);
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) w[e * 64 + l] = (float)(window512[sample_of(l, e)] * scale);
    return w;
}
// synthesis window rows: [8][64], entry 4 tl + r of lane l = window[16 (16 tl + 4 g + r) + l % 16] * scale
inline std::vector<float> build_synth_rows(const float* window512, double scale) {
    std::vector<float> w(8 * 64);
    for (int l = 0; l < 64; ++l)
        for (int tl = 0; tl < 2; ++tl)
            for (int r = 0; r < 4; ++r)
                w[(4 * tl + r) * 64 + l] = (float)(window512[16 * (16 * tl + 4 * (l >> 4) + r) + (l & 15)] * scale);
    return w;
}

}  // namespace mc
}  // namespace setk
