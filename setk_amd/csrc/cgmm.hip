// cgmm.hip -- CGMM (K = 2) time-frequency mask estimation by EM on the device.
//
// Replaces (funcwj/setk): CgmmTrainer / Cgmm / CgDistribution / Covariance in
// scripts/sptk/libs/cluster.py:94-287, 396-465 as driven by
// scripts/sptk/estimate_cgmm_masks.py:19-71 (K = 2, deterministic init or an
// initial mask; alpha fixed at 1/2 or, with update_alpha, re-estimated per iteration).
//
// Per EM iteration (cluster.py:193-212, 261-287), M = channels:
//   R_k(f)   = sum_t gamma_k M / phi_k  x x^H / max(sum_t gamma_k, eps)      cgmm_accum + finalize
//   (w, V)   = eigh(R_k);  w <- max(w / max(max w, eps), eps)                 cgmm_eig (fp64 Jacobi)
//   phi_k    = max(|x^H R_k^-1 x|, eps) / M,  R^-1 = V diag(1/w) V^H          cgmm_estep
//   gamma_k  = alpha_k N_k / sum_k alpha_k N_k, log N_k = -M log phi_k - sum log w   cgmm_estep
// The quadratic form is evaluated in the eigenbasis, sum_j |v_j^H x|^2 / w_j
// (all terms positive), so float32 does not suffer the cancellation the dense
// x^H R^-1 x would (1/w reaches 8e6); the reference does it in float64.
// Spectrogram layout [C][T][F]; one wavefront handles 32 bins x 2 classes
// (lanes 0-31 class 0, lanes 32-63 class 1), classes meet through __shfl_xor.
#include <cstring>
#include "common.h"
#include "dpp.h"
#include "fft512.h"
#include "../../include/setk_hip.h"

namespace setk {

typedef double2 zd;
#define ZD __device__ __forceinline__
ZD zd zd_add(zd a, zd b) { return make_double2(a.x + b.x, a.y + b.y); }
ZD zd zd_sub(zd a, zd b) { return make_double2(a.x - b.x, a.y - b.y); }
ZD zd zd_mul(zd a, zd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
ZD zd zd_cmul(zd a, zd b) { return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }
ZD zd zd_scale(zd a, double s) { return make_double2(a.x * s, a.y * s); }
ZD double zd_abs2(zd a) { return a.x * a.x + a.y * a.y; }
ZD zd zd_shfl(zd v, int src) { return make_double2(__shfl(v.x, src, 8), __shfl(v.y, src, 8)); }

constexpr float kEps32 = 1.1920928955078125e-07f;
constexpr int kCgInitId = 0, kCgInitMask = 1, kCgEm = 2;
// frames per partial-sum chunk: one wavefront folds a chunk in float32, the chunks are summed in
// float64 (cgmm_finalize).  Shorter chunks = more waves and more partial traffic
// (configs[4], 125 x 30 s: 41.1 ms with 64, 39.5 with 128, 38.6 - 39.1 with 192).
#ifndef SETK_CG_CHUNK
#define SETK_CG_CHUNK 128
#endif
constexpr int kCgChunk = SETK_CG_CHUNK;

struct CgmmArgs {
    const cf* spec;          // [C][T][spitch], F entries used per row
    float* gamma;            // [2][T][F]
    float* phi;              // [2][T][F]
    const float* init_mask;  // [T][F] or null
    float* partials;         // [nchunks][2][2*NP+1][pitch]
    double* R;               // [2][2*NP][pitch]   re | im planes per Hermitian pair
    cf* V;                   // [2][F][C][C]       V[j][i] = component i of eigenvector j
    float* invw;             // [2][F][C]
    float* logdet;           // [2][F]
    float* mask_out;         // [T][F] or null (last E-step)
    float* alpha;            // [2][F] mixture weights (1/2 unless update_alpha, cluster.py:254-257)
    int T, F, pitch, nchunks, tchunk, mode, update_alpha;
    int spitch;              // row pitch of spec in complex entries (F, or padded to 128 bytes)
};

// ---- M-step: weighted outer products, 32 bins x 2 classes per wavefront ----
template <int C>
__global__ __launch_bounds__(64) void cgmm_accum_kernel(const CgmmArgs* __restrict__ tbl, int em) {
    constexpr int NP = npairs(C);
    CgmmArgs a = tbl[blockIdx.z];
    if (em) a.mode = kCgEm;
    if ((int)blockIdx.y >= a.nchunks) return;
    const int lane = threadIdx.x;
    const int f = blockIdx.x * 32 + (lane & 31);
    const int k = lane >> 5;
    const int chunk = blockIdx.y;
    const int t0 = chunk * a.tchunk, t1 = min(a.T, t0 + a.tchunk);
    const int T = a.T, F = a.F;
    // the first launch of a run also sets the mixture weights to 1 / K (cluster.py:446)
    if (!em && chunk == 0 && f < F) a.alpha[(size_t)k * F + f] = 0.5f;
    cf acc[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) acc[e] = make_float2(0.f, 0.f);
    float sumg = 0.f;
    if (f < F && !(a.mode == kCgInitId && k == 1)) {
        for (int t = t0; t < t1; ++t) {
            float w, g;
            if (a.mode == kCgEm) {
                g = a.gamma[((size_t)k * T + t) * F + f];
                w = g * (float)C / a.phi[((size_t)k * T + t) * F + f];
            } else if (a.mode == kCgInitMask) {
                const float m = a.init_mask[(size_t)t * F + f];
                g = (k == 0) ? m : 1.f - m;
                w = g;
            } else {
                g = 1.f;
                w = 1.f;
            }
            sumg += g;
            cf x[C];
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = a.spec[((size_t)c * T + t) * a.spitch + f];
            int e = 0;
#pragma unroll
            for (int i = 0; i < C; ++i)
#pragma unroll
                for (int j = i; j < C; ++j) {
                    const cf p = cmulc(x[i], x[j]);
                    acc[e].x = fmaf(w, p.x, acc[e].x);
                    if (i != j) acc[e].y = fmaf(w, p.y, acc[e].y);
                    ++e;
                }
        }
    }
    if (f < F) {
        float* P = a.partials + ((size_t)chunk * 2 + k) * (2 * NP + 1) * a.pitch;
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            P[(size_t)e * a.pitch + f] = acc[e].x;
            P[(size_t)(NP + e) * a.pitch + f] = acc[e].y;
        }
        P[(size_t)(2 * NP) * a.pitch + f] = sumg;
    }
}

// sum the frame chunks in fp64, normalise (cluster.py:203-205 / 419-440)
__global__ void cgmm_finalize_kernel(const CgmmArgs* __restrict__ tbl, int em, int C) {
    CgmmArgs a = tbl[blockIdx.z];
    if (em) a.mode = kCgEm;
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int NP = npairs(C);
    const int e = blockIdx.y % (2 * NP), k = blockIdx.y / (2 * NP);
    if (f >= a.F) return;
    double* out = a.R + ((size_t)k * 2 * NP + e) * a.pitch;
    if (a.mode == kCgInitId && k == 1) {
        // identity: diagonal pairs are the ones with i == j in the real planes
        int idx = 0, isdiag = 0;
        for (int i = 0; i < C && !isdiag; ++i)
            for (int j = i; j < C; ++j) {
                if (idx == e && i == j) isdiag = 1;
                ++idx;
            }
        out[f] = isdiag ? 1.0 : 0.0;
        return;
    }
    const size_t slab = (size_t)(2 * NP + 1) * a.pitch;
    double acc = 0.0, den = 0.0;
    for (int c = 0; c < a.nchunks; ++c) {
        const float* P = a.partials + ((size_t)c * 2 + k) * slab;
        acc += (double)P[(size_t)e * a.pitch + f];
        den += (double)P[(size_t)(2 * NP) * a.pitch + f];
    }
    // Cgmm.update (cluster.py:246-257): alpha_k = mean_t gamma_k, from the same sums
    if (a.update_alpha && a.mode == kCgEm && e == 0) a.alpha[(size_t)k * a.F + f] = (float)(den / a.T);
    if (a.mode == kCgInitId) den = (double)a.T;
    out[f] = acc / fmax(den, (double)kEps32);
}

// ---- eigendecomposition of R_k(f): one-sided Jacobi, 8 lanes per problem ----

// one Jacobi rotation of column j against column j ^ M (the 7 XOR matchings of a
// sweep, exchanged with DPP moves -- dpp.h), eigenvectors accumulated in v
template <int C, int M>
ZD bool cgmm_jacobi_round(zd (&g)[C], zd (&v)[C], int j) {
    const int p = j ^ M;
    zd gp[C], vp[C];
    double m = 0.0, o = 0.0;
    zd d = make_double2(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        gp[i] = make_double2(dshfl_xor<M>(g[i].x), dshfl_xor<M>(g[i].y));
        vp[i] = make_double2(dshfl_xor<M>(v[i].x), dshfl_xor<M>(v[i].y));
        m += zd_abs2(g[i]);
        o += zd_abs2(gp[i]);
        d = zd_add(d, zd_cmul(g[i], gp[i]));
    }
    const double dd = zd_abs2(d);
    if (dd > 1e-26 * m * o && dd > 0.0) {
        const double absd = sqrt(dd);
        const double sigma = (j < p) ? 1.0 : -1.0;
        const double zeta = sigma * (o - m) / (2.0 * absd);
        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + t * t);
        const double fsc = sigma * cs * t / absd;
        const zd ph = make_double2(d.x * fsc, -d.y * fsc);
#pragma unroll
        for (int i = 0; i < C; ++i) {
            g[i] = zd_sub(zd_scale(g[i], cs), zd_mul(ph, gp[i]));
            v[i] = zd_sub(zd_scale(v[i], cs), zd_mul(ph, vp[i]));
        }
        return true;
    }
    return false;
}

template <int C>
__global__ __launch_bounds__(64) void cgmm_eig_kernel(const CgmmArgs* __restrict__ tbl) {
    constexpr int NP = npairs(C);
    const CgmmArgs a = tbl[blockIdx.z];
    const int tid = threadIdx.x;
    const int j = tid & 7;
    const int F = a.F;
    long prob = (long)blockIdx.x * 8 + (tid >> 3);
    const long nprob = 2L * F;
    const bool live = prob < nprob;
    if (!live) prob = nprob - 1;
    const int k = (int)(prob / F), f = (int)(prob % F);
    // column j of the Hermitian matrix
    zd g[C], v[C];
    const double* base = a.R + (size_t)k * 2 * NP * a.pitch + f;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        g[i] = make_double2(0.0, 0.0);
        v[i] = make_double2((i == j) ? 1.0 : 0.0, 0.0);
        if (j < C) {
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const int e = pair_index(lo, hi, C);
            const double re = base[(size_t)e * a.pitch];
            const double im = (i == j) ? 0.0 : base[(size_t)(NP + e) * a.pitch];
            g[i] = make_double2(re, (i <= j) ? im : -im);
        }
    }
    bool done = false;
    for (int sweep = 0; sweep < 40 && !done; ++sweep) {
        bool rot = false;
        rot |= cgmm_jacobi_round<C, 1>(g, v, j);
        rot |= cgmm_jacobi_round<C, 2>(g, v, j);
        rot |= cgmm_jacobi_round<C, 3>(g, v, j);
        rot |= cgmm_jacobi_round<C, 4>(g, v, j);
        rot |= cgmm_jacobi_round<C, 5>(g, v, j);
        rot |= cgmm_jacobi_round<C, 6>(g, v, j);
        rot |= cgmm_jacobi_round<C, 7>(g, v, j);
        done = !__any(rot);
    }
    double lam2 = 0.0;
#pragma unroll
    for (int i = 0; i < C; ++i) lam2 += zd_abs2(g[i]);
    double lam = sqrt(lam2);
    double lmax = lam;
#pragma unroll
    for (int s = 1; s < 8; s <<= 1) lmax = fmax(lmax, __shfl_xor(lmax, s, 8));
    // cluster.py:107-113
    double w = lam / fmax(lmax, (double)kEps32);
    w = fmax(w, (double)kEps32);
    double lw = (j < C) ? log(w) : 0.0;
#pragma unroll
    for (int s = 1; s < 8; s <<= 1) lw += __shfl_xor(lw, s, 8);
    if (live && j < C) {
        cf* Vd = a.V + (((size_t)k * F + f) * C + j) * C;
#pragma unroll
        for (int i = 0; i < C; ++i) Vd[i] = make_float2((float)v[i].x, (float)v[i].y);
        a.invw[((size_t)k * F + f) * C + j] = (float)(1.0 / w);
        if (j == 0) a.logdet[(size_t)k * F + f] = (float)lw;
    }
}

// ---- E-step: phi, posterior gamma (cluster.py:207-212, 214-235, 261-287) ----
// ACCUM: also fold this E-step's gamma M / phi x x^H into the next M-step's
// partial sums (what cgmm_accum_kernel does in EM mode) -- the spectrogram is
// then streamed once per EM iteration instead of twice and gamma / phi never
// leave the registers; the stand-alone E-step (ACCUM = false) only closes the
// last iteration and writes the outputs.
// 157 VGPRs at C = 6 (V, 1 / w, the accumulators and a prefetched frame): three waves per SIMD;
// forced to 128 registers the kernel spills 47 of them and the EM runs 3 x slower
template <int C, bool ACCUM>
__global__ __launch_bounds__(64) void cgmm_estep_kernel(const CgmmArgs* __restrict__ tbl, int last) {
    constexpr int NP = npairs(C);
    CgmmArgs a = tbl[blockIdx.z];
    if (!last) a.mask_out = nullptr;
    if ((int)blockIdx.y >= a.nchunks) return;
    const int lane = threadIdx.x;
    const int f = blockIdx.x * 32 + (lane & 31);
    const int k = lane >> 5;
    const int chunk = blockIdx.y;
    const int t0 = chunk * a.tchunk, t1 = min(a.T, t0 + a.tchunk);
    const int T = a.T, F = a.F;
    const bool ok = f < F;
    const int fc = ok ? f : F - 1;
    cf Vr[C][C];
    float iw[C];
    const cf* Vd = a.V + ((size_t)k * F + fc) * C * C;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        iw[j] = a.invw[((size_t)k * F + fc) * C + j];
#pragma unroll
        for (int i = 0; i < C; ++i) Vr[j][i] = Vd[j * C + i];
    }
    const float ld = a.logdet[(size_t)k * F + fc];
    const float al_mine = a.alpha[(size_t)k * F + fc], al_other = a.alpha[(size_t)(1 - k) * F + fc];
    cf acc[ACCUM ? NP : 1];
#pragma unroll
    for (int e = 0; e < (ACCUM ? NP : 1); ++e) acc[e] = make_float2(0.f, 0.f);
    float sumg = 0.f;
    // the next frame's bins are requested one iteration ahead
    cf xn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xn[c] = a.spec[((size_t)c * T + min(t0, T - 1)) * a.spitch + fc];
    for (int t = t0; t < t1; ++t) {
        cf x[C];
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = xn[c];
        if (t + 1 < t1) {
#pragma unroll
            for (int c = 0; c < C; ++c) xn[c] = a.spec[((size_t)c * T + t + 1) * a.spitch + fc];
        }
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            cf pr = make_float2(0.f, 0.f);  // v_j^H x
#pragma unroll
            for (int i = 0; i < C; ++i) {
                pr.x += Vr[j][i].x * x[i].x + Vr[j][i].y * x[i].y;
                pr.y += Vr[j][i].x * x[i].y - Vr[j][i].y * x[i].x;
            }
            q = fmaf(iw[j], pr.x * pr.x + pr.y * pr.y, q);
        }
        const float ph = fmaxf(q, kEps32) / (float)C;
        const float lp = -(float)C * logf(ph) - ld;
        const float lo = __shfl_xor(lp, 32);
        const float mx = fmaxf(lp, lo);
        const float mine = al_mine * expf(lp - mx), other = al_other * expf(lo - mx);
        const float g = mine / fmaxf(mine + other, kEps32);
        if (ACCUM) {
            const float w = g * (float)C / ph;
            sumg += g;
            int e = 0;
#pragma unroll
            for (int i = 0; i < C; ++i)
#pragma unroll
                for (int j = i; j < C; ++j) {
                    const cf p = cmulc(x[i], x[j]);
                    acc[e].x = fmaf(w, p.x, acc[e].x);
                    if (i != j) acc[e].y = fmaf(w, p.y, acc[e].y);
                    ++e;
                }
        } else if (ok) {
            a.phi[((size_t)k * T + t) * F + f] = ph;
            a.gamma[((size_t)k * T + t) * F + f] = g;
            if (a.mask_out && k == 0) a.mask_out[(size_t)t * F + f] = g;
        }
    }
    if (ACCUM && ok) {
        float* P = a.partials + ((size_t)chunk * 2 + k) * (2 * NP + 1) * a.pitch;
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            P[(size_t)e * a.pitch + f] = acc[e].x;
            P[(size_t)(NP + e) * a.pitch + f] = acc[e].y;
        }
        P[(size_t)(2 * NP) * a.pitch + f] = sumg;
    }
}

template <int C>
static hipError_t cgmm_run_t(const CgmmArgs* d_tbl, int n_utts, int F, int max_chunks,
                             int num_iters, hipStream_t s) {
    const int NP = npairs(C);
    dim3 g_tf((F + 31) / 32, max_chunks, n_utts);
    dim3 g_fin((F + 255) / 256, 2 * NP * 2, n_utts);
    dim3 g_eig((2 * F + 7) / 8, 1, n_utts);
    // M-step from the initialisation, then per EM iteration ONE pass over the
    // spectrogram (E-step fused with the next M-step's accumulation), and a closing
    // E-step that writes the masks
    for (int it = 0; it <= num_iters; ++it) {
        const int em = it > 0;
        if (!em)
            hipLaunchKernelGGL(cgmm_accum_kernel<C>, g_tf, dim3(64), 0, s, d_tbl, 0);
        else
            hipLaunchKernelGGL((cgmm_estep_kernel<C, true>), g_tf, dim3(64), 0, s, d_tbl, 0);
        hipLaunchKernelGGL(cgmm_finalize_kernel, g_fin, dim3(256), 0, s, d_tbl, em, C);
        hipLaunchKernelGGL(cgmm_eig_kernel<C>, g_eig, dim3(64), 0, s, d_tbl);
    }
    hipLaunchKernelGGL((cgmm_estep_kernel<C, false>), g_tf, dim3(64), 0, s, d_tbl, 1);
    return hipGetLastError();
}

// Fill one utterance's argument block; `scratch` is carved for its work arrays
// (gamma, phi included).  Returns bytes used.
size_t cgmm_fill_args(void* args_out, int C, const float* spec, int T, int F,
                      const float* init_mask, float* gamma_opt, float* mask_out, void* scratch,
                      int update_alpha, int spec_pitch) {
    const int NP = npairs(C);
    CgmmArgs a;
    std::memset(&a, 0, sizeof(a));
    a.spec = reinterpret_cast<const cf*>(spec);
    a.init_mask = init_mask;
    a.mask_out = mask_out;
    a.T = T;
    a.F = F;
    a.pitch = ((F + 7) / 8) * 8;
    a.tchunk = kCgChunk;
    a.nchunks = (T + a.tchunk - 1) / a.tchunk;
    a.mode = init_mask ? kCgInitMask : kCgInitId;
    char* p = static_cast<char*>(scratch);
    auto take = [&](size_t bytes) {
        char* r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    a.gamma = gamma_opt ? gamma_opt : reinterpret_cast<float*>(take((size_t)2 * T * F * 4));
    a.phi = reinterpret_cast<float*>(take((size_t)2 * T * F * 4));
    a.partials = reinterpret_cast<float*>(take((size_t)a.nchunks * 2 * (2 * NP + 1) * a.pitch * 4));
    a.R = reinterpret_cast<double*>(take((size_t)2 * 2 * NP * a.pitch * 8));
    a.V = reinterpret_cast<cf*>(take((size_t)2 * F * C * C * 8));
    a.invw = reinterpret_cast<float*>(take((size_t)2 * F * C * 4));
    a.logdet = reinterpret_cast<float*>(take((size_t)2 * F * 4));
    a.alpha = reinterpret_cast<float*>(take((size_t)2 * F * 4));
    a.update_alpha = update_alpha;
    a.spitch = spec_pitch > 0 ? spec_pitch : F;
    std::memcpy(args_out, &a, sizeof(a));
    return (size_t)(p - static_cast<char*>(scratch));
}

size_t cgmm_args_bytes() { return sizeof(CgmmArgs); }

size_t cgmm_scratch_bytes(int C, int T, int F) {
    const int NP = npairs(C);
    const size_t pitch = ((F + 7) / 8) * 8;
    const size_t nchunks = (T + kCgChunk - 1) / kCgChunk;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return 2 * al((size_t)2 * T * F * 4) + al(nchunks * 2 * (2 * NP + 1) * pitch * 4) +
           al((size_t)2 * 2 * NP * pitch * 8) + al((size_t)2 * F * C * C * 8) +
           al((size_t)2 * F * C * 4) + 2 * al((size_t)2 * F * 4) + 1024;
}

hipError_t launch_cgmm_batch(int C, const void* d_tbl, int n_utts, int F, int max_frames,
                             int num_iters, hipStream_t s) {
    const CgmmArgs* t = static_cast<const CgmmArgs*>(d_tbl);
    const int max_chunks = (max_frames + kCgChunk - 1) / kCgChunk;
    switch (C) {
        case 1: return cgmm_run_t<1>(t, n_utts, F, max_chunks, num_iters, s);
        case 2: return cgmm_run_t<2>(t, n_utts, F, max_chunks, num_iters, s);
        case 3: return cgmm_run_t<3>(t, n_utts, F, max_chunks, num_iters, s);
        case 4: return cgmm_run_t<4>(t, n_utts, F, max_chunks, num_iters, s);
        case 5: return cgmm_run_t<5>(t, n_utts, F, max_chunks, num_iters, s);
        case 6: return cgmm_run_t<6>(t, n_utts, F, max_chunks, num_iters, s);
        case 7: return cgmm_run_t<7>(t, n_utts, F, max_chunks, num_iters, s);
        case 8: return cgmm_run_t<8>(t, n_utts, F, max_chunks, num_iters, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace setk
