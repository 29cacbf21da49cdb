// mcdft_probe.hip -- device check of csrc/mcdft.h: operand layouts + accuracy of the
// matrix-core DFT-512 (forward and inverse) against a float64 DFT on the host, then its rate
// alone and beside VALU-only waves (does the fp16 matrix pipe run beside the vector ALUs?).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I setk_amd/csrc -o tools/ubench/mcdft_probe tools/ubench/mcdft_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "mcdft_tables.h"

using namespace setk::mc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- forward: one wave = 16 frames ----
__global__ __launch_bounds__(256) void fwd_kernel(const float* frames, const unsigned* tab, const float* wrows,
                                                  float2* out, float inv_scale, int nb) {
    __shared__ __attribute__((aligned(16))) float a16s[4][16 * kOddPitch];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    Fwd K;
    load_fwd(K, tab, lane);
    float win[8];
    for (int e = 0; e < 8; ++e) win[e] = wrows[e * 64 + lane];
    const int b0 = (blockIdx.x * 4 + wave) * 16;
    for (int j = 0; j < 16; ++j) {
        const int b = b0 + j;
        float xw[8];
        for (int e = 0; e < 8; ++e) xw[e] = b < nb ? frames[(size_t)b * 512 + sample_of(lane, e)] : 0.f;
        f4 zr, zi, a16;
        forward(xw, win, K, zr, zi, a16);
        store_a16(a16s[wave], j, lane, a16);
        if (b < nb)
            for (int r = 0; r < 4; ++r) {
                if (!bin_valid(c, g, r)) continue;
                out[(size_t)b * 257 + bin_of(c, g, r)] = make_float2(zr[r] * inv_scale, zi[r] * inv_scale);
            }
    }
    const f4 d = odd_tile(a16s[wave], tab_h8(tab, kW_OT_H, lane), tab_h8(tab, kW_OT_L, lane), lane);
    const int b = b0 + c;
    if (b < nb) {
        out[(size_t)b * 257 + 16 + 32 * (2 * g)] = make_float2(d[0] * inv_scale, d[1] * inv_scale);
        out[(size_t)b * 257 + 16 + 32 * (2 * g + 1)] = make_float2(d[2] * inv_scale, d[3] * inv_scale);
    }
}

// ---- inverse: one wave = 16 frames ----
__global__ __launch_bounds__(256) void inv_kernel(const float2* Y, const unsigned* tab, float* y, int nb) {
    __shared__ float e16s[4][16][17];
    __shared__ float scs[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    Inv K;
    load_inv(K, tab, lane);
    const int b0 = (blockIdx.x * 4 + wave) * 16;
    // per-frame scale
    for (int j = 0; j < 16; ++j) {
        const int b = b0 + j;
        float mx = 0.f;
        if (b < nb)
            for (int i = lane; i < 257; i += 64) mx = fmaxf(mx, fmaxf(fabsf(Y[(size_t)b * 257 + i].x), fabsf(Y[(size_t)b * 257 + i].y)));
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        int ex = 0;
        if (mx > 0.f) frexpf(mx, &ex);
        if (lane == 0) scs[wave][j] = ldexpf(1.f, 11 - ex);
    }
    // odd family of the 16 frames
    {
        const int b = b0 + c;
        const float sc = scs[wave][c];
        float v[8];
        for (int e = 0; e < 8; ++e) {
            const int q = 4 * (g & 1) + e / 2;
            const float2 t = b < nb ? Y[(size_t)b * 257 + 16 + 32 * q] : make_float2(0.f, 0.f);
            v[e] = ((e & 1) ? t.y : t.x) * sc;
        }
        const f4 d = inv_odd_tile(v, tab_h8(tab, kW_IT_H, lane), tab_h8(tab, kW_IT_L, lane), lane);
        for (int r = 0; r < 4; ++r) e16s[wave][4 * g + r][c] = d[r];
    }
    for (int j = 0; j < 16; ++j) {
        const int b = b0 + j;
        if (b >= nb) break;
        const float sc = scs[wave][j];
        f4 yr, yi;
        for (int r = 0; r < 4; ++r) {
            float2 t = Y[(size_t)b * 257 + bin_of(c, g, r)];
            if (c == 0 && ((g == 0 && r == 0) || (g == 2 && r == 3))) t.y = 0.f;
            yr[r] = t.x * sc;
            yi[r] = t.y * sc;
        }
        f4 y0, y1;
        inverse(yr, yi, e16s[wave][j][c], K, y0, y1, lane);
        const float is = 1.f / sc;
        for (int r = 0; r < 4; ++r) {
            y[(size_t)b * 512 + 16 * (4 * g + r) + c] = y0[r] * is;
            y[(size_t)b * 512 + 16 * (16 + 4 * g + r) + c] = y1[r] * is;
        }
    }
}

// ---- rate: transform waves and VALU-only waves in one workgroup ----
// waves [0, ntw): forward transforms of L2-resident frames, spectra written to an LDS tile
// (as pass 1 would); waves [ntw, nw): chains of independent v_fma_f32 (covariance-like).
template <int NW>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void rate_kernel(const float* frames, const unsigned* tab, const float* wrows,
                                                    float* sink, long long* cyc, int ntw, int iters_t, int iters_v, int nfr) {
    extern __shared__ __attribute__((aligned(16))) float2 tile[];  // [NW][272]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const long long t0 = clock64();
    if (wave < ntw) {
        Fwd K;
        load_fwd(K, tab, lane);
        float win[8];
        for (int e = 0; e < 8; ++e) win[e] = wrows[e * 64 + lane];
        float2* slot = tile + wave * 272;
        lds_fp lf = to_lds(reinterpret_cast<float*>(slot + bin_of(c, g, 0)));
        lds_fp li = opaque_next(lf);
        const bool ok = !(c == 0 && g >= 2);
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        int fr = (blockIdx.x * 7 + wave * 3) % nfr;
#pragma unroll 1
        for (int it = 0; it < iters_t; ++it) {
            const float* src = frames + (size_t)fr * 512;
            fr = fr + 1 == nfr ? 0 : fr + 1;
            float xw[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xw[e] = src[sample_of(lane, e)];
            f4 zr, zi, a16;
            forward(xw, win, K, zr, zi, a16);
            acc += a16;
            if (ok) store_bins(lf, li, zr, zi);
            if (c == 0 && g == 2) { lf[64 * 3] = zr[3]; lf[64 * 3 + 1] = zi[3]; }
        }
        if (acc[0] == 12345.f) sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + tile[lane].x;
    } else {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = (float)(lane + i);
        const float m = 1.0000001f, b = 1e-9f;
#pragma unroll 1
        for (int it = 0; it < iters_v; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b));
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += a[i];
        if (s == 12345.f) sink[threadIdx.x] = s;
    }
    const long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

static void host_rdft(const std::vector<double>& x, std::vector<std::complex<double>>& X) {
    const double PI = 3.14159265358979323846;
    X.assign(257, 0.0);
    for (int k = 0; k <= 256; ++k) {
        std::complex<double> s = 0;
        for (int n = 0; n < 512; ++n) s += x[n] * std::exp(std::complex<double>(0, -2 * PI * ((long long)n * k % 512) / 512));
        X[k] = s;
    }
}

template <int NW>
static void run_rate(const float* d_fr, const unsigned* d_tab, const float* d_w, float* d_sink, long long* d_cyc,
                     int ntw, int it_t, int it_v, int nfr, const char* label) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t lds = (size_t)NW * 272 * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate_kernel<NW>, dim3(256), dim3(64 * NW), lds, 0, d_fr, d_tab, d_w, d_sink, d_cyc, ntw, it_t, it_v, nfr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    long long cyc[16];
    CK(hipMemcpy(cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost));
    const double ntr = 256.0 * ntw * it_t, nv = 256.0 * (NW - ntw) * it_v * 128;
    printf("%-34s waves %2d (T %2d | V %2d)  %.3f ms", label, NW, ntw, NW - ntw, best);
    if (ntw && it_t) printf("  | %.2f ns/transform/CU  T-wave %.0f cyc/transform", best * 1e6 / (ntw * (double)it_t), (double)cyc[0] / it_t);
    if (NW > ntw && it_v) printf("  | V-wave %.2f cyc/instr", (double)cyc[NW - 1] / (it_v * 128.0));
    printf("  [%.3g transforms, %.3g v_fma]\n", ntr, nv);
}

int main() {
    const int nb = 200;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 0.2f);
    std::vector<float> fr((size_t)nb * 512), win(512);
    for (auto& v : fr) v = std::fmin(1.f, std::fmax(-1.f, nd(rng)));
    for (int b = 0; b < 8; ++b)
        for (int n = 0; n < 512; ++n) fr[(size_t)b * 512 + n] = (n == 37 * b + 3) ? 1.f : 0.f;  // impulses: layout check
    for (int n = 0; n < 512; ++n) win[n] = 0.5f - 0.5f * std::cos(2 * 3.14159265358979323846 * n / 512);
    const double scale = 1024.0;
    auto tab = build_table();
    auto wrows = build_window_rows(win.data(), scale);
    float *d_fr, *d_w, *d_y, *d_sink;
    unsigned* d_tab;
    float2 *d_X, *d_Y;
    long long* d_cyc;
    CK(hipMalloc(&d_fr, fr.size() * 4));
    CK(hipMalloc(&d_w, wrows.size() * 4));
    CK(hipMalloc(&d_tab, tab.size() * 4));
    CK(hipMalloc(&d_X, (size_t)nb * 257 * 8));
    CK(hipMalloc(&d_Y, (size_t)nb * 257 * 8));
    CK(hipMalloc(&d_y, (size_t)nb * 512 * 4));
    CK(hipMalloc(&d_sink, 4096 * 4));
    CK(hipMalloc(&d_cyc, 16 * 8));
    CK(hipMemcpy(d_fr, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, wrows.data(), wrows.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_X, 0xff, (size_t)nb * 257 * 8));
    hipLaunchKernelGGL(fwd_kernel, dim3((nb + 63) / 64), dim3(256), 0, 0, d_fr, d_tab, d_w, d_X, (float)(1.0 / scale), nb);
    CK(hipDeviceSynchronize());
    std::vector<float2> X((size_t)nb * 257);
    CK(hipMemcpy(X.data(), d_X, X.size() * 8, hipMemcpyDeviceToHost));
    double num = 0, den = 0, mxe = 0;
    int nanc = 0;
    std::vector<std::vector<std::complex<double>>> ref(nb);
    for (int b = 0; b < nb; ++b) {
        std::vector<double> x(512);
        for (int n = 0; n < 512; ++n) x[n] = (double)fr[(size_t)b * 512 + n] * win[n];
        host_rdft(x, ref[b]);
        for (int k = 0; k <= 256; ++k) {
            const std::complex<double> d(X[(size_t)b * 257 + k].x, X[(size_t)b * 257 + k].y);
            if (!(std::abs(d) < 1e30)) { ++nanc; continue; }
            num += std::norm(d - ref[b][k]);
            den += std::norm(ref[b][k]);
            mxe = std::fmax(mxe, std::abs(d - ref[b][k]));
        }
    }
    printf("forward: %d frames, relative RMS error %.3e, max abs %.3e, non-finite %d\n", nb, std::sqrt(num / den), mxe, nanc);
    // inverse of the reference spectra (float32), against the windowed frames
    std::vector<float2> Yh((size_t)nb * 257);
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k <= 256; ++k) Yh[(size_t)b * 257 + k] = make_float2((float)ref[b][k].real(), (float)ref[b][k].imag());
    CK(hipMemcpy(d_Y, Yh.data(), Yh.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(inv_kernel, dim3((nb + 63) / 64), dim3(256), 0, 0, d_Y, d_tab, d_y, nb);
    CK(hipDeviceSynchronize());
    std::vector<float> y((size_t)nb * 512);
    CK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
    num = den = mxe = 0;
    for (int b = 0; b < nb; ++b)
        for (int n = 0; n < 512; ++n) {
            const double r = 512.0 * fr[(size_t)b * 512 + n] * win[n], d = y[(size_t)b * 512 + n];
            num += (d - r) * (d - r);
            den += r * r;
            mxe = std::fmax(mxe, std::fabs(d - r));
        }
    printf("inverse: relative RMS error %.3e, max abs %.3e (of values up to 512)\n", std::sqrt(num / den), mxe);

    // ---- rates ----
    const int nfr = nb;
    run_rate<4>(d_fr, d_tab, d_w, d_sink, d_cyc, 4, 4000, 0, nfr, "T only, 1 wave/SIMD");
    run_rate<8>(d_fr, d_tab, d_w, d_sink, d_cyc, 8, 4000, 0, nfr, "T only, 2 waves/SIMD");
    run_rate<16>(d_fr, d_tab, d_w, d_sink, d_cyc, 16, 4000, 0, nfr, "T only, 4 waves/SIMD");
    run_rate<16>(d_fr, d_tab, d_w, d_sink, d_cyc, 0, 0, 2000, nfr, "V only, 4 waves/SIMD");
    run_rate<16>(d_fr, d_tab, d_w, d_sink, d_cyc, 12, 0, 2000, nfr, "V only, 1 wave/SIMD (12 idle)");
    run_rate<16>(d_fr, d_tab, d_w, d_sink, d_cyc, 4, 0, 2000, nfr, "V only, 3 waves/SIMD (4 idle)");
    run_rate<16>(d_fr, d_tab, d_w, d_sink, d_cyc, 4, 4000, 0, nfr, "T only, 1 wave/SIMD (12 idle)");
    run_rate<16>(d_fr, d_tab, d_w, d_sink, d_cyc, 4, 4000, 2000, nfr, "T x4 beside V x12");
    run_rate<16>(d_fr, d_tab, d_w, d_sink, d_cyc, 4, 8000, 2000, nfr, "T x4 (2x work) beside V x12");
    run_rate<16>(d_fr, d_tab, d_w, d_sink, d_cyc, 8, 4000, 2000, nfr, "T x8 beside V x8");
    return 0;
}
