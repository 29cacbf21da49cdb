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
// One workgroup owns one bin: a 16 x 16 thread grid accumulates [R | r] in
// registers (thread (ty, tx) holds rows ty + 16 i, columns tx + 16 j) from a
// chunk of tap vectors staged in LDS, the sums go to LDS, R is Cholesky
// factored in place (it is Hermitian positive definite; a non-positive pivot
// reports SETK_NUM_SINGULAR, the reference's LinAlgError), G is solved for by
// substitution and the filter is applied in a second sweep over the frames.
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

constexpr int kWpeMaxNK = 96;  // LDS: NK^2 * 16 B <= 147 KB
constexpr int kWpeTC = 16;     // frames per staged chunk
constexpr int kWpeRows = 6;    // ceil(96 / 16) rows / thread
constexpr int kWpeCols = 7;    // ceil((96 + 16) / 16) columns / thread

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
    int N, T, taps, delay;
};

// one workgroup per (bin, utterance): blockIdx.y indexes the argument table
__global__ __launch_bounds__(256) void wpe_step_kernel(const WpeArgs* __restrict__ tbl) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const WpeArgs a = tbl[blockIdx.y];
    const int N = a.N, T = a.T, taps = a.taps, delay = a.delay;
    const int NK = N * taps, D = NK + N;
    zd* R = reinterpret_cast<zd*>(wsm);          // [NK][NK] row major; L in place (lower)
    zd* G = R + (size_t)NK * NK;                 // [NK][N]  r, then G
    zd* V = G + (size_t)NK * N;                  // [kWpeTC][D] staged vectors (yt | x)
    double* ilam = reinterpret_cast<double*>(V + (size_t)kWpeTC * D);  // [kWpeTC]
    int* flag = reinterpret_cast<int*>(ilam + kWpeTC);

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int f = blockIdx.x;
    const float2* xf = a.x + (size_t)f * N * T;
    const double* lf = a.lam + (size_t)f * T;
    if (tid == 0) *flag = 0;

    // ---- [R | r] = sum_t (yt / lambda) [yt ; x]^H ----
    zd acc[kWpeRows][kWpeCols];
#pragma unroll
    for (int i = 0; i < kWpeRows; ++i)
#pragma unroll
        for (int j = 0; j < kWpeCols; ++j) acc[i][j] = zmk(0.0, 0.0);
    for (int t0 = 0; t0 < T; t0 += kWpeTC) {
        __syncthreads();
        for (int i = tid; i < kWpeTC * D; i += 256) {
            const int tl = i / D, n = i - tl * D;
            const int t = t0 + tl;
            zd v = zmk(0.0, 0.0);
            if (t < T) {
                int c, src;
                if (n < NK) {
                    const int k = n / N;
                    c = n - k * N;
                    src = t - k - delay;
                } else {
                    c = n - NK;
                    src = t;
                }
                if (src >= 0) {
                    const float2 s = xf[(size_t)c * T + src];
                    v = zmk((double)s.x, (double)s.y);
                }
            }
            V[i] = v;
        }
        if (tid < kWpeTC) ilam[tid] = (t0 + tid < T) ? 1.0 / lf[t0 + tid] : 0.0;
        __syncthreads();
        for (int tl = 0; tl < kWpeTC; ++tl) {
            const zd* v = V + (size_t)tl * D;
            const double il = ilam[tl];
            zd col[kWpeCols];
#pragma unroll
            for (int j = 0; j < kWpeCols; ++j) {
                const int n = tx + 16 * j;
                col[j] = (n < D) ? v[n] : zmk(0.0, 0.0);
            }
#pragma unroll
            for (int i = 0; i < kWpeRows; ++i) {
                const int m = ty + 16 * i;
                if (m < NK) {
                    const zd row = zmk(v[m].x * il, v[m].y * il);
#pragma unroll
                    for (int j = 0; j < kWpeCols; ++j) zfmac(acc[i][j], row, col[j]);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kWpeRows; ++i)
#pragma unroll
        for (int j = 0; j < kWpeCols; ++j) {
            const int m = ty + 16 * i, n = tx + 16 * j;
            if (m < NK && n < NK) R[(size_t)m * NK + n] = acc[i][j];
            if (m < NK && n >= NK && n < D) G[(size_t)m * N + (n - NK)] = acc[i][j];
        }
    __syncthreads();

    // ---- Cholesky R = L L^H (lower triangle in place), right-looking ----
    // Pivots are floored at eps64 * NK * max diag (as chol_lds in solve.hip does with
    // eps32): a numerically semi-definite R (fewer frames than N * taps, a duplicated
    // or silent channel) is factored with noise-level pivots, the way LAPACK's pivoted LU
    // behind numpy.linalg.solve goes through on such input (libs/wpe.py:76); only an
    // all-zero or non-finite R is "singular" (LinAlgError).
    double dmax = 0.0;
    for (int k = 0; k < NK; ++k) dmax = fmax(dmax, R[(size_t)k * NK + k].x);
    const double pfloor = dmax * 2.220446049250313e-16 * (double)NK;
    for (int k = 0; k < NK; ++k) {
        double dkk = R[(size_t)k * NK + k].x;
        if (!(dmax > 0.0) || !isfinite(dkk) || !isfinite(dmax)) {
            if (tid == 0) *flag = SETK_NUM_SINGULAR;
            break;  // uniform: every thread reads the same pivot
        }
        dkk = fmax(dkk, pfloor);
        const double lkk = sqrt(dkk), inv = 1.0 / lkk;
        __syncthreads();
        for (int i = k + tid; i < NK; i += 256) {
            zd v = R[(size_t)i * NK + k];
            R[(size_t)i * NK + k] = (i == k) ? zmk(lkk, 0.0) : zmk(v.x * inv, v.y * inv);
        }
        __syncthreads();
        const int rem = NK - k - 1;
        // trailing update of the lower triangle: R[i][j] -= L[i][k] conj(L[j][k]), j <= i
        for (int e = tid; e < rem * rem; e += 256) {
            const int i = k + 1 + e / rem, j = k + 1 + e % rem;
            if (j <= i) {
                const zd p = zmulcd(R[(size_t)i * NK + k], R[(size_t)j * NK + k]);
                R[(size_t)i * NK + j] = zsubd(R[(size_t)i * NK + j], p);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    const int bad = *flag;
    if (tid == 0 && a.status) a.status[f] = bad;
    if (bad) {
        // the reference raises LinAlgError for the whole utterance; leave x in place
        for (int i = tid; i < N * T; i += 256) a.out[(size_t)f * N * T + i] = xf[i];
        return;
    }

    // ---- L y = r, L^H G = y (N right-hand sides; thread c owns a column) ----
    for (int k = 0; k < NK; ++k) {
        if (tid < N) {
            const double inv = 1.0 / R[(size_t)k * NK + k].x;
            zd v = G[(size_t)k * N + tid];
            G[(size_t)k * N + tid] = zmk(v.x * inv, v.y * inv);
        }
        __syncthreads();
        for (int e = tid; e < (NK - k - 1) * N; e += 256) {
            const int i = k + 1 + e / N, c = e % N;
            const zd l = R[(size_t)i * NK + k], y = G[(size_t)k * N + c];
            // r[i] -= L[i][k] y[k]
            zd& g = G[(size_t)i * N + c];
            g = zmk(g.x - (l.x * y.x - l.y * y.y), g.y - (l.x * y.y + l.y * y.x));
        }
        __syncthreads();
    }
    for (int k = NK - 1; k >= 0; --k) {
        if (tid < N) {
            const double inv = 1.0 / R[(size_t)k * NK + k].x;
            zd v = G[(size_t)k * N + tid];
            G[(size_t)k * N + tid] = zmk(v.x * inv, v.y * inv);
        }
        __syncthreads();
        for (int e = tid; e < k * N; e += 256) {
            const int i = e / N, c = e % N;
            // y[i] -= conj(L[k][i]) G[k]
            const zd l = R[(size_t)k * NK + i], g = G[(size_t)k * N + c];
            zd& y = G[(size_t)i * N + c];
            y = zmk(y.x - (l.x * g.x + l.y * g.y), y.y - (l.x * g.y - l.y * g.x));
        }
        __syncthreads();
    }

    // ---- d[c][t] = x[c][t] - sum_m conj(G[m][c]) yt[m][t] ----
    for (int i = tid; i < N * T; i += 256) {
        const int c = i / T, t = i - c * T;
        zd s = zmk(0.0, 0.0);
        for (int k = 0; k < taps; ++k) {
            const int src = t - k - delay;
            if (src < 0) break;
            for (int n = 0; n < N; ++n) {
                const float2 y = xf[(size_t)n * T + src];
                const zd g = G[(size_t)(k * N + n) * N + c];
                // conj(g) * y
                s.x += g.x * y.x + g.y * y.y;
                s.y += g.x * y.y - g.y * y.x;
            }
        }
        const float2 xv = xf[i];
        a.out[(size_t)f * N * T + i] = make_float2((float)((double)xv.x - s.x),
                                                   (float)((double)xv.y - s.y));
    }
}

size_t wpe_lds_bytes(int N, int taps) {
    const size_t NK = (size_t)N * taps, D = NK + N;
    return (NK * NK + NK * N + (size_t)kWpeTC * D) * sizeof(zd) + kWpeTC * sizeof(double) + 16;
}

bool wpe_supported(int N, int taps) {
    const int NK = N * taps;
    return N >= 1 && N <= 16 && taps >= 1 && NK <= kWpeMaxNK && NK + N <= 16 * kWpeCols &&
           wpe_lds_bytes(N, taps) <= 160 * 1024;
}

// what exactly a shape is refused for (the LDS bound is the tighter one from 8 channels up:
// C = 8 allows 10 taps, C = 16 allows 4)
const char* wpe_limit_message(int N, int taps) {
    static thread_local char buf[256];
    const int NK = N * taps;
    snprintf(buf, sizeof(buf),
             "WPE on the device needs 1 <= channels <= 16, channels * taps <= %d and "
             "(NK^2 + NK N + %d (NK + N)) complex128 entries <= 160 KB of LDS "
             "(channels = %d, taps = %d: NK = %d, %zu bytes)",
             kWpeMaxNK, kWpeTC, N, taps, NK, wpe_lds_bytes(N, taps));
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
                   int N, int T, int taps, int delay) {
    WpeArgs a;
    a.x = reinterpret_cast<const float2*>(x_fct);
    a.lam = lam;
    a.out = reinterpret_cast<float2*>(out_fct);
    a.status = status;
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
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wpe_step_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wpe_step_kernel, dim3(F, n_utts), dim3(256), lds, s,
                       static_cast<const WpeArgs*>(d_tbl));
    return hipGetLastError();
}

}  // namespace setk
