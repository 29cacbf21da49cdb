// wpe.hip -- one iteration of (G)WPE dereverberation per frequency bin, fp64.
//
// Replaces (funcwj/setk) scripts/sptk/libs/wpe.py: compute_tap_mat (:13-29),
// compute_lambda (:32-55) and wpe_step (:58-81), the building blocks of wpe()
// (:84-110, apply_wpe.py) and facted_wpd() (:113-177, apply_wpd.py):
//
//     yt[k N + n][t] = x[n][t - k - delay]                 (tap-stacked observation)
//     R = sum_t yt yt^H / lambda_t        (NK x NK)        r = sum_t yt x^H / lambda_t
//     G = R^-1 r                          (NK x N)         d = x - G^H yt
//
// The reference evaluates this in complex128 (its lambda is float64 and promotes
// everything); accumulating R in float32 moves the result by 3e-2 on the doc
// example (cond(R) ~ 5e10), so correlation, factorisation and filter run in fp64
// here and only the spectrograms are complex64.
//
// One workgroup owns one bin: a 16 x 16 thread grid accumulates the lower 16 x 16 tiles
// of R and r in registers (thread (ty, tx) holds rows ty + 16 i, columns tx + 16 j <= i;
// the kernel is instantiated per tile count) from a chunk of tap vectors staged in LDS, the
// sums go to LDS, R is Cholesky factored in place with r carried along (it is Hermitian
// positive definite; an all-zero or non-finite R reports SETK_NUM_SINGULAR, the
// reference's LinAlgError), the back substitution runs one right-hand side per wavefront
// and the filter is applied in a second sweep over the frames.
// Layout: spectrograms [F][N][T] (frames contiguous), lambda [F][T] float64.
#include <cstdio>
#include <cstring>
#include "common.h"
#include "../../include/setk_hip.h"

namespace setk {

typedef double2 zd;
#define WD __device__ __forceinline__
WD zd zmk(double a, double b) { return make_double2(a, b); }
WD zd zaddd(zd a, zd b) { return zmk(a.x + b.x, a.y + b.y); }
WD zd zsubd(zd a, zd b) { return zmk(a.x - b.x, a.y - b.y); }
// a * conj(b)
WD zd zmulcd(zd a, zd b) { return zmk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
// acc += a * conj(b)
WD void zfmac(zd& acc, zd a, zd b) {
    acc.x = fma(a.x, b.x, fma(a.y, b.y, acc.x));
    acc.y = fma(a.y, b.x, fma(-a.x, b.y, acc.y));
}

constexpr int kWpeMaxNK = 96;       // R in LDS: NK^2 * 16 B <= 147 KB
constexpr int kWpeMaxNKWide = 256;  // R in global memory (the reference has no bound: libs/wpe.py:58-81)
constexpr int kWpeTC = 16;          // smallest chunk of frames staged at a time
constexpr int kWpeWideNTW = 12;     // tiles per wavefront and correlation pass of the wide form

// [C][T][F] (the library's spectrogram layout) <-> [F][C][T]
__global__ __launch_bounds__(256) void wpe_to_fct_kernel(const float2* __restrict__ spec, int C,
                                                         int T, int F, float2* __restrict__ out) {
    __shared__ float2 tile[32][33];
    const int c = blockIdx.z;
    const int f0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, f = f0 + tx;
        if (t < T && f < F) tile[i][tx] = spec[((size_t)c * T + t) * F + f];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int f = f0 + i, t = t0 + tx;
        if (t < T && f < F) out[((size_t)f * C + c) * T + t] = tile[tx][i];
    }
}

__global__ __launch_bounds__(256) void wpe_from_fct_kernel(const float2* __restrict__ fct, int C,
                                                           int T, int F, float2* __restrict__ spec) {
    __shared__ float2 tile[32][33];
    const int c = blockIdx.z;
    const int f0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int f = f0 + i, t = t0 + tx;
        if (t < T && f < F) tile[i][tx] = fct[((size_t)f * C + c) * T + t];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, f = f0 + tx;
        if (t < T && f < F) spec[((size_t)c * T + t) * F + f] = tile[tx][i];
    }
}

// compute_lambda (libs/wpe.py:32-55): mean over channels of |d|^2 (float32, as the
// reference), summed over the +-ctx frames that exist, divided by their count in
// float64, floored at eps_f32.
__global__ __launch_bounds__(256) void wpe_lambda_kernel(const float2* __restrict__ d, int C, int T,
                                                         int ctx, double* __restrict__ lam) {
    const int f = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float sum = 0.f;
    int count = 0;
    for (int dt = -ctx; dt <= ctx; ++dt) {
        const int u = t + dt;
        if (u < 0 || u >= T) continue;
        float p = 0.f;
        for (int c = 0; c < C; ++c) {
            const float2 v = d[((size_t)f * C + c) * T + u];
            p += v.x * v.x + v.y * v.y;
        }
        sum += p / (float)C;
        ++count;
    }
    lam[(size_t)f * T + t] = fmax((double)sum / (double)count, 1.1920928955078125e-07);
}

// lambda = max(|enh|^2, eps) of a single-channel [T][F] spectrogram (facted_wpd,
// libs/wpe.py:146-149) -> [F][T] float64
__global__ __launch_bounds__(256) void wpe_lambda_from_enh_kernel(const float2* __restrict__ enh,
                                                                  int T, int F,
                                                                  double* __restrict__ lam) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * F) return;
    const int t = (int)(i / F), f = (int)(i % F);
    const float2 v = enh[i];
    lam[(size_t)f * T + t] =
        fmax((double)v.x * v.x + (double)v.y * v.y, 1.1920928955078125e-07);
}

// 1 / lambda as the float32 [T][F] "mask" of the power-weighted covariance
// (facted_wpd: Rd = sum_t d d^H / lambda / T; the scale cancels in the weight)
__global__ __launch_bounds__(256) void wpe_inv_lambda_kernel(const double* __restrict__ lam, int T,
                                                             int F, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * F) return;
    const int t = (int)(i / F), f = (int)(i % F);
    out[i] = (float)(1.0 / lam[(size_t)f * T + t]);
}

struct WpeArgs {
    const float2* x;    // [F][N][T]
    const double* lam;  // [F][T]
    float2* out;        // [F][N][T]
    int* status;        // [F]
    long long* timing;  // SETK_WPE_TIMING: [F][4] cycles (correlation, factor, back-solve, filter)
    double2* rwork;     // wide form only: [F][NK][NK] complex128, this utterance's R / L
    int N, T, taps, delay;
};

typedef double v4d __attribute__((ext_vector_type(4)));

// one workgroup per (bin, utterance): blockIdx.y indexes the argument table.
//
// The correlation runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64; same peak as the
// vector pipe, but one LDS read per 1024 multiply-adds instead of one per 4).  A complex
// outer product is a real one of the stacked parts: with z[(m, p)] = Re / Im of yt[m],
//     S[(m, p)][(n, q)] = sum_t z[(m, p)] z[(n, q)] / lambda,
//     R[m][n] = (S[m0][n0] + S[m1][n1]) + i (S[m1][n0] - S[m0][n1]),
// so a 16 x 16 real tile is an 8 x 8 block of R (or of r, with x in place of yt on the column
// side).  Within a tile row i = 4 (2 h + p) + g is (m = 8 I + 4 h + g, part p) and column j is
// (n = 8 J + j / 2, part j & 1): in the instruction's result layout (lane = j + 16 (i % 4),
// register = i / 4; tools/ubench/mfma_f64_layout.hip) a lane then holds both parts of its two
// rows and its neighbour the other column part, one lane swap away.  Only the tiles J <= I of R
// are computed; the NT = RT (RT + 1) / 2 + RT XT tiles (RT = ceil(NK / 8), XT = ceil(N / 8)) are
// dealt round-robin to the four wavefronts, NTW = ceil(NT / 4) accumulators each.
//
// WIDE (channels x taps beyond what LDS holds, up to 16 x 16): R lives in global memory
// (NK^2 complex128 per workgroup, L2 / MALL resident while it is factored), the correlation
// walks the frames once per group of 4 NTW tiles, everything else is the same code.
template <int NTW, bool WIDE>
__global__ __launch_bounds__(256) void wpe_step_kernel(const WpeArgs* __restrict__ tbl, int TC) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const WpeArgs a = tbl[blockIdx.y];
    const int N = a.N, T = a.T, taps = a.taps, delay = a.delay;
    const int NK = N * taps;
    // per chunk of TC frames: the source frames the tap vectors reach, x[n][t0 - delay -
    // taps + 1 .. t0 + TC - 1 - delay] (W per channel; yt[k N + n][t] is the entry tl + taps - 1
    // - k of channel n), the chunk's own frames x[n][t0 .. t0 + TC) for r, and a strip of zeros
    // that the rows / columns past NK (partial tiles) and the channels past N point at
    const int W = TC + taps - 1;
    // R: [NK][NK] row major; L in place (strictly lower).  G: [NK][N]  r, then y, then G
    zd* R = WIDE ? a.rwork + (size_t)blockIdx.x * NK * NK : reinterpret_cast<zd*>(wsm);
    zd* G = WIDE ? reinterpret_cast<zd*>(wsm) : reinterpret_cast<zd*>(wsm) + (size_t)NK * NK;
    zd* XS = G + (size_t)NK * N;                 // [N][W] | [N][TC] | zeros [TC]
    double* ilam = reinterpret_cast<double*>(XS + (size_t)N * (W + TC) + TC);  // [TC]
    double* idiag = ilam + TC;                   // [NK] 1 / L[k][k]
    int* flag = reinterpret_cast<int*>(idiag + NK);

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int f = blockIdx.x;
    const float2* xf = a.x + (size_t)f * N * T;
    const double* lf = a.lam + (size_t)f * T;
    if (tid == 0) *flag = 0;
    long long clk0 = 0;
    if (a.timing && tid == 0) clk0 = (long long)__builtin_amdgcn_s_memtime();

    // ---- [R | r] = sum_t (yt / lambda) [yt ; x]^H, lower tiles of R only ----
    const int lane = tid & 63, wv = tid >> 6;
    const int RT = (NK + 7) >> 3, XT = (N + 7) >> 3;
    const int NTR = RT * (RT + 1) / 2, NTT = NTR + RT * XT;
    const int zoff = N * (W + TC);
    const double* XSd = reinterpret_cast<const double*>(XS);
    // operand addresses (in doubles, without the frame): A = row (m, p) of the lane's tile
    // row, frame lane / 16 of the step; B = column (n, q)
    const int ai = lane & 15, ag = ai & 3, av = ai >> 2;
    const int kq = lane >> 4;
    for (int i = tid; i < TC; i += 256) XS[zoff + i] = zmk(0.0, 0.0);
    // (one trip unless WIDE: the dispatcher picks NTW >= ceil(NTT / 4) for the LDS form)
    for (int tb = 0; tb < NTT; tb += 4 * NTW) {
    v4d acc[NTW];
    int baseA[NTW], baseB[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
        const int t = tb + wv + 4 * q;
        int I = 0, J = 0;
        bool isx = false;
        if (t < NTR) {
            while ((I + 1) * (I + 2) / 2 <= t) ++I;
            J = t - I * (I + 1) / 2;
        } else if (t < NTT) {
            isx = true;
            I = (t - NTR) / XT;
            J = (t - NTR) - I * XT;
        }
        const int m = 8 * I + 4 * (av >> 1) + ag, n = 8 * J + (ai >> 1);
        int ea = zoff, eb = zoff;
        if (t < NTT && m < NK) ea = (m % N) * W + taps - 1 - m / N;
        if (t < NTT && !isx && n < NK) eb = (n % N) * W + taps - 1 - n / N;
        if (t < NTT && isx && n < N) eb = N * W + n * TC;
        baseA[q] = 2 * (ea + kq) + (av & 1);
        baseB[q] = 2 * (eb + kq) + (ai & 1);
    }
    for (int t0 = 0; t0 < T; t0 += TC) {
        __syncthreads();
        for (int i = tid; i < N * (W + TC); i += 256) {
            int n, src;
            if (i < N * W) {
                n = i / W;
                src = t0 - delay - (taps - 1) + (i - n * W);
            } else {
                const int e = i - N * W;
                n = e / TC;
                src = t0 + (e - n * TC);
            }
            zd v = zmk(0.0, 0.0);
            if (src >= 0 && src < T) {
                const float2 sv = xf[(size_t)n * T + src];
                v = zmk((double)sv.x, (double)sv.y);
            }
            XS[i] = v;
        }
        for (int i = tid; i < TC; i += 256) ilam[i] = (t0 + i < T) ? 1.0 / lf[t0 + i] : 0.0;
        __syncthreads();
#pragma unroll 2
        for (int s4 = 0; s4 < TC; s4 += 4) {
            const double il = ilam[s4 + kq];
#pragma unroll
            for (int q = 0; q < NTW; ++q) {
                const double av_ = XSd[baseA[q] + 2 * s4] * il;
                const double bv_ = XSd[baseB[q] + 2 * s4];
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av_, bv_, acc[q], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    // complex entries out of the real tiles: even lanes hold column part 0, their neighbours
    // part 1
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int t = tb + wv + 4 * q;
        double nb[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) nb[v] = __shfl_xor(acc[q][v], 1);
        if (t < NTT && !(lane & 1)) {
            int I = 0, J = 0;
            bool isx = false;
            if (t < NTR) {
                while ((I + 1) * (I + 2) / 2 <= t) ++I;
                J = t - I * (I + 1) / 2;
            } else {
                isx = true;
                I = (t - NTR) / XT;
                J = (t - NTR) - I * XT;
            }
            const int n = 8 * J + ((lane & 15) >> 1);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int m = 8 * I + 4 * hh + (lane >> 4);
                const zd val = zmk(acc[q][2 * hh] + nb[2 * hh + 1], acc[q][2 * hh + 1] - nb[2 * hh]);
                if (!isx && m < NK && n <= m) R[(size_t)m * NK + n] = val;
                if (isx && m < NK && n < N) G[(size_t)m * N + n] = val;
            }
        }
    }
    }  // tile groups
    __syncthreads();

    // ---- Cholesky R = L L^H (strictly lower triangle in place, 1 / L[k][k] aside), right-
    // looking, with r carried along as extra columns: the forward substitution L y = r comes
    // out of the same sweep ----
    // Pivots at or below eps64 * NK * max diag count as zero: a numerically semi-definite R
    // (fewer frames than N * taps, a duplicated or silent channel) is solved in the subspace
    // it spans -- what is left of such a column after elimination is rounding noise, and
    // dividing by a floored pivot instead lets that noise grow quadratically from column to
    // column until it overflows.  LAPACK's pivoted LU behind numpy.linalg.solve
    // (libs/wpe.py:76) also goes through on such input, with a noise-determined filter; only
    // an all-zero or non-finite R is "singular" (LinAlgError).
    long long clk1 = 0;
    if (a.timing && tid == 0) clk1 = (long long)__builtin_amdgcn_s_memtime();
    double dmax = 0.0;
    for (int k = 0; k < NK; ++k) dmax = fmax(dmax, R[(size_t)k * NK + k].x);
    const double pfloor = dmax * 2.220446049250313e-16 * (double)NK;
    for (int k = 0; k < NK; ++k) {
        double dkk = R[(size_t)k * NK + k].x;
        if (!(dmax > 0.0) || !isfinite(dkk) || !isfinite(dmax)) {
            if (tid == 0) *flag = SETK_NUM_SINGULAR;
            break;  // uniform: every thread reads the same pivot
        }
        // a pivot at the noise level is a direction R does not span: its column is dropped
        // (L[.][k] = 0, y[k] = 0, G[k] = 0) instead of being divided by noise
        const double inv = (dkk > pfloor) ? 1.0 / sqrt(dkk) : 0.0;
        if (!(dkk > pfloor) && tid == 0) *flag = SETK_NUM_RANKDEF;  // reported, not fatal
        // column k below the diagonal (the diagonal entry itself stays: others may still be
        // reading it as their pivot) and row k of r
        for (int i = k + 1 + tid; i < NK; i += 256) {
            const zd v = R[(size_t)i * NK + k];
            R[(size_t)i * NK + k] = zmk(v.x * inv, v.y * inv);
        }
        if (tid >= 256 - N) {
            zd& g = G[(size_t)k * N + (255 - tid)];
            g = zmk(g.x * inv, g.y * inv);
        }
        if (tid == 128) idiag[k] = inv;
        __syncthreads();
        // trailing update: R[i][j] -= L[i][k] conj(L[j][k]) (k < j <= i), r[i] -= L[i][k] y[k]
        for (int i = k + 1 + ty; i < NK; i += 16) {
            const zd li = R[(size_t)i * NK + k];
            for (int j = k + 1 + tx; j <= i; j += 16) {
                const zd p = zmulcd(li, R[(size_t)j * NK + k]);
                R[(size_t)i * NK + j] = zsubd(R[(size_t)i * NK + j], p);
            }
            if (tx < N) {
                const zd y = G[(size_t)k * N + tx];
                zd& g = G[(size_t)i * N + tx];
                g = zmk(g.x - (li.x * y.x - li.y * y.y), g.y - (li.x * y.y + li.y * y.x));
            }
        }
        __syncthreads();
    }
    __syncthreads();
    const int bad = *flag;
    if (tid == 0 && a.status) a.status[f] = bad;
    if (bad && bad != SETK_NUM_RANKDEF) {
        // the reference raises LinAlgError for the whole utterance; leave x in place
        for (int i = tid; i < N * T; i += 256) a.out[(size_t)f * N * T + i] = xf[i];
        return;
    }

    long long clk2 = 0;
    if (a.timing && tid == 0) clk2 = (long long)__builtin_amdgcn_s_memtime();
    // ---- L^H G = y: one wavefront per right-hand side, y in registers (rows lane and
    // lane + 64), no workgroup barrier inside ----
    {
        constexpr int NY = WIDE ? kWpeMaxNKWide / 64 : 2;
        const int lane = tid & 63, w = tid >> 6;
        for (int c = w; c < N; c += 4) {
            zd y[NY];
#pragma unroll
            for (int j = 0; j < NY; ++j)
                y[j] = (lane + 64 * j < NK) ? G[(size_t)(lane + 64 * j) * N + c] : zmk(0.0, 0.0);
            for (int k = NK - 1; k >= 0; --k) {
                zd src = y[0];
#pragma unroll
                for (int j = 1; j < NY; ++j)
                    if ((k >> 6) == j) src = y[j];
                const double inv = idiag[k];
                const zd g = zmk(__shfl(src.x, k & 63) * inv, __shfl(src.y, k & 63) * inv);
                // y[i] -= conj(L[k][i]) g  (i < k);  y[k] = g
#pragma unroll
                for (int j = 0; j < NY; ++j) {
                    const int i = lane + 64 * j;
                    if (64 * j < k && i < k) {
                        const zd l = R[(size_t)k * NK + i];
                        y[j] = zmk(y[j].x - (l.x * g.x + l.y * g.y), y[j].y - (l.x * g.y - l.y * g.x));
                    } else if (i == k) {
                        y[j] = g;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NY; ++j)
                if (lane + 64 * j < NK) G[(size_t)(lane + 64 * j) * N + c] = y[j];
        }
    }
    __syncthreads();
    long long clk3 = 0;
    if (a.timing && tid == 0) clk3 = (long long)__builtin_amdgcn_s_memtime();

    // ---- d[c][t] = x[c][t] - sum_m conj(G[m][c]) yt[m][t]: a thread owns a frame and works
    // through the channels four at a time (every delayed sample is loaded once per group) ----
    for (int t = tid; t < T; t += 256) {
        for (int c0 = 0; c0 < N; c0 += 4) {
            zd s[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = zmk(0.0, 0.0);
            for (int k = 0; k < taps; ++k) {
                const int src = t - k - delay;
                if (src < 0) break;
                for (int n = 0; n < N; ++n) {
                    const float2 yf = xf[(size_t)n * T + src];
                    const double yx = yf.x, yy = yf.y;
                    const zd* g = G + (size_t)(k * N + n) * N + c0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (c0 + q < N) {
                            // conj(g) * y
                            s[q].x += g[q].x * yx + g[q].y * yy;
                            s[q].y += g[q].x * yy - g[q].y * yx;
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (c0 + q < N) {
                    const size_t o = (size_t)(c0 + q) * T + t;
                    const float2 xv = xf[o];
                    a.out[(size_t)f * N * T + o] = make_float2((float)((double)xv.x - s[q].x),
                                                               (float)((double)xv.y - s[q].y));
                }
            }
        }
    }
    if (a.timing) {
        __syncthreads();
        if (tid == 0) {
            const long long clk4 = (long long)__builtin_amdgcn_s_memtime();
            long long* o = a.timing + (size_t)f * 4;
            o[0] = clk1 - clk0;
            o[1] = clk2 - clk1;
            o[2] = clk3 - clk2;
            o[3] = clk4 - clk3;
        }
    }
}

// does R fit LDS next to r and the smallest chunk?
bool wpe_is_wide(int N, int taps) {
    const size_t NK = (size_t)N * taps, W = (size_t)kWpeTC + taps - 1;
    return NK > (size_t)kWpeMaxNK ||
           (NK * NK + NK * N + N * (W + kWpeTC) + kWpeTC) * sizeof(zd) + (kWpeTC + NK) * sizeof(double) + 16 >
               160 * 1024;
}

size_t wpe_lds_bytes_tc(int N, int taps, int TC) {
    const size_t NK = (size_t)N * taps, W = (size_t)TC + taps - 1;
    const size_t r = wpe_is_wide(N, taps) ? 0 : NK * NK;
    return (r + NK * N + N * (W + TC) + TC) * sizeof(zd) + (TC + NK) * sizeof(double) + 16;
}

// bytes of global scratch one utterance needs per bin (0: R is LDS resident)
size_t wpe_wide_bytes_per_bin(int N, int taps) {
    const size_t NK = (size_t)N * taps;
    return wpe_is_wide(N, taps) ? NK * NK * sizeof(zd) : 0;
}

// frames per staged chunk: the largest of 64 / 32 / 16 that leaves R, r and the chunk in
// the 160 KB of a CU
int wpe_chunk_frames(int N, int taps) {
    for (int tc = 64; tc > kWpeTC; tc >>= 1)
        if (wpe_lds_bytes_tc(N, taps, tc) <= 160 * 1024) return tc;
    return kWpeTC;
}

size_t wpe_lds_bytes(int N, int taps) { return wpe_lds_bytes_tc(N, taps, wpe_chunk_frames(N, taps)); }

bool wpe_supported(int N, int taps) {
    const int NK = N * taps;
    return N >= 1 && N <= 16 && taps >= 1 && NK <= kWpeMaxNKWide &&
           wpe_lds_bytes(N, taps) <= 160 * 1024;
}

// what exactly a shape is refused for (R itself moves to global memory when it does not fit
// LDS -- 8 channels x 11 taps, 16 x 5 and up; the bound that remains is NK <= 256)
const char* wpe_limit_message(int N, int taps) {
    static thread_local char buf[256];
    const int NK = N * taps;
    snprintf(buf, sizeof(buf),
             "WPE on the device needs 1 <= channels <= 16 and channels * taps <= %d "
             "(channels = %d, taps = %d: NK = %d)",
             kWpeMaxNKWide, N, taps, NK);
    return buf;
}

hipError_t launch_wpe_transpose(const float* in, int C, int T, int F, float* out, bool to_fct,
                                hipStream_t s) {
    dim3 grid((F + 31) / 32, (T + 31) / 32, C);
    if (to_fct)
        hipLaunchKernelGGL(wpe_to_fct_kernel, grid, dim3(256), 0, s,
                           reinterpret_cast<const float2*>(in), C, T, F,
                           reinterpret_cast<float2*>(out));
    else
        hipLaunchKernelGGL(wpe_from_fct_kernel, grid, dim3(256), 0, s,
                           reinterpret_cast<const float2*>(in), C, T, F,
                           reinterpret_cast<float2*>(out));
    return hipGetLastError();
}

hipError_t launch_wpe_lambda(const float* d_fct, int C, int T, int F, int ctx, double* lam,
                             hipStream_t s) {
    hipLaunchKernelGGL(wpe_lambda_kernel, dim3((T + 255) / 256, F), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(d_fct), C, T, ctx, lam);
    return hipGetLastError();
}

hipError_t launch_wpe_lambda_from_enh(const float* enh_tf, int T, int F, double* lam,
                                      hipStream_t s) {
    const size_t n = (size_t)T * F;
    hipLaunchKernelGGL(wpe_lambda_from_enh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       s, reinterpret_cast<const float2*>(enh_tf), T, F, lam);
    return hipGetLastError();
}

hipError_t launch_wpe_inv_lambda(const double* lam, int T, int F, float* out, hipStream_t s) {
    const size_t n = (size_t)T * F;
    hipLaunchKernelGGL(wpe_inv_lambda_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       lam, T, F, out);
    return hipGetLastError();
}

size_t wpe_args_bytes() { return sizeof(WpeArgs); }

void wpe_fill_args(void* dst, const float* x_fct, const double* lam, float* out_fct, int* status,
                   int N, int T, int taps, int delay, long long* timing, void* rwork) {
    WpeArgs a;
    a.x = reinterpret_cast<const float2*>(x_fct);
    a.lam = lam;
    a.out = reinterpret_cast<float2*>(out_fct);
    a.status = status;
    a.timing = timing;
    a.rwork = static_cast<double2*>(rwork);
    a.N = N;
    a.T = T;
    a.taps = taps;
    a.delay = delay;
    memcpy(dst, &a, sizeof(a));
}

// d_tbl: n_utts argument blocks (device); every utterance has N channels and `taps` taps
hipError_t launch_wpe_step_batch(const void* d_tbl, int n_utts, int N, int F, int taps,
                                 hipStream_t s) {
    const size_t lds = wpe_lds_bytes(N, taps);
    void (*kern)(const WpeArgs*, int) = nullptr;
    const int RT = (N * taps + 7) / 8, XT = (N + 7) / 8;
    const int ntw = (RT * (RT + 1) / 2 + RT * XT + 3) / 4;  // tiles per wavefront, <= 26
    if (wpe_is_wide(N, taps)) kern = wpe_step_kernel<kWpeWideNTW, true>;
#define SETK_WPE_CASE(n) \
    if (!kern && ntw <= n) kern = wpe_step_kernel<n, false>
    SETK_WPE_CASE(1);
    SETK_WPE_CASE(2);
    SETK_WPE_CASE(3);
    SETK_WPE_CASE(4);
    SETK_WPE_CASE(5);
    SETK_WPE_CASE(6);
    SETK_WPE_CASE(8);
    SETK_WPE_CASE(10);
    SETK_WPE_CASE(12);
    SETK_WPE_CASE(14);
    SETK_WPE_CASE(17);
    SETK_WPE_CASE(20);
    SETK_WPE_CASE(23);
    SETK_WPE_CASE(26);
#undef SETK_WPE_CASE
    if (!kern) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(F, n_utts), dim3(256), lds, s, static_cast<const WpeArgs*>(d_tbl),
                       wpe_chunk_frames(N, taps));
    return hipGetLastError();
}

}  // namespace setk
