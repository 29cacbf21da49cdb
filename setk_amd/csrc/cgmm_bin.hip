// cgmm_bin.hip -- CGMM (K = 2) mask estimation as a BIN-RESIDENT EM: one workgroup
// owns one (utterance, frequency bin), loads that bin's C x T observations ONCE
// (registers + LDS) and runs the initialisation, every EM iteration and the closing
// posterior inside ONE launch.  The bins of the model are independent
// (libs/cluster.py:193-287: alpha, R, phi are all per f), so nothing crosses
// workgroups.
//
// Replaces (funcwj/setk): CgmmTrainer.__init__/train (libs/cluster.py:396-465),
// Cgmm.update/predict (:246-287), CgDistribution.update_parameters/log_pdf
// (:193-235), Covariance (:94-133) for num_classes = 2.
//
// Per pass over the T frames (thread t owns frames t, t + NT, ...):
//   E   q_k = x^H R_k^-1 x by forward substitution with the Cholesky factor of the
//       (eigenvalue-normalised, floored) R_k -- y = L^-1 x, q = |y|^2: every term
//       positive, float32 loses only sqrt(cond) where the dense form loses cond;
//       phi_k = max(q_k, eps) / M; log N_k = -M log phi_k - log det R_k; posterior.
//   M   gamma_k M / phi_k  x x^H folded into 2 x (C(C+1)/2) register accumulators
//       (outer products formed once for both classes), a DPP tree over each 16-lane
//       row in float32 and the 4 x waves row sums in float64.
// Between two passes, wave k (k = 0, 1) solves class k on (C x C) lanes in float64:
//   R_k -> two-sided Jacobi in the parallel (round-robin) order, warm-started from
//   the previous iteration's eigenvectors (A = V^H R V), rotation angles in float32
//   and the rotations themselves exactly unitary in float64, the sweep that starts
//   with every |a_pq|^2 <= 1e-8 a_pp a_qq is the last -> eigenvalues scaled by
//   1 / max(w_max, eps) and floored at eps (cluster.py:107-113) -> R_eff =
//   V diag(w') V^H -> Cholesky.  The factor's 2 x (C^2) floats travel through LDS
//   into SGPRs (they are uniform over the workgroup).
// Layout in: bin-major spectrogram [F][C][Tp] (frames contiguous; written by
// spec_to_binmajor_kernel from the [C][T][Fp] dump); out: bin-major posteriors
// [F][Tp], transposed back to the reference's T x F by binmajor_to_tf_kernel.
// Roofline: VALU issue (about 340 float32 instructions per frame and iteration);
// HBM traffic is one read of the spectrogram + one write of the masks.
#include <cstring>
#include "common.h"
#include "fft512.h"
#include "../../include/setk_hip.h"

namespace setk {

namespace {

typedef double2 zd;
#define ZD __device__ __forceinline__
ZD zd zmk(double a, double b) { return make_double2(a, b); }
ZD zd zadd(zd a, zd b) { return zmk(a.x + b.x, a.y + b.y); }
ZD zd zsub(zd a, zd b) { return zmk(a.x - b.x, a.y - b.y); }
ZD zd zmul(zd a, zd b) { return zmk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
ZD zd zmulc(zd a, zd b) { return zmk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a conj(b)
ZD zd zcmul(zd a, zd b) { return zmk(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }  // conj(a) b
ZD zd zscale(zd a, double s) { return zmk(a.x * s, a.y * s); }
ZD zd zconj(zd a) { return zmk(a.x, -a.y); }

constexpr float kEpsF = 1.1920928955078125e-07f;
constexpr double kEpsD = 1.1920928955078125e-07;
constexpr int kModeInitId = 0, kModeInitMask = 1, kModeEm = 2, kModeFinal = 3;
constexpr int kMaxSweeps = 14;
constexpr double kTol2 = 1e-18;    // rotate while |a_pq|^2 > kTol2 a_pp a_qq
constexpr double kLast2 = 1e-8;    // a sweep whose rotations all start below this is the last

struct CgmmBinArgs {
    const cf* xb;            // [F][C][Tp]
    const float* init_mask;  // [T][F] or null
    float* gamma_bm;         // [nout][F][Tp]
    int T, Tp, F, update_alpha, nout, pad_;
};

// 1 / sqrt(x), float64, two Newton steps on the hardware estimate (x > 0, normal)
ZD double rsq64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    return y;
}
ZD double rcp64(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}

// LDS traffic between the lanes of ONE wave: the LDS executes a wave's operations
// in order, the compiler only has to keep them in program order
ZD void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// compiler-only barrier: LDS loads are not merged or hoisted across it (the frames of a
// chunk are re-read per phase instead of being kept in 12 VGPRs each)
ZD void reload_fence() { asm volatile("" ::: "memory"); }

template <int CTRL, int RM = 0xf>
ZD float dppf(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, true));
}
// sum over each 16-lane row, result in the row's lane 15
ZD float row_sum16(float v) {
    v += dppf<0x111>(v);  // row_shr:1
    v += dppf<0x112>(v);  // row_shr:2
    v += dppf<0x114>(v);  // row_shr:4
    v += dppf<0x118>(v);  // row_shr:8
    return v;
}

ZD float sgpr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// per-class parameters of the E-step, uniform over the workgroup
template <int C>
struct ClassPar {
    static constexpr int NPO = C * (C - 1) / 2;
    float lre[NPO > 0 ? NPO : 1], lim[NPO > 0 ? NPO : 1];  // L_ij, i > j, row-major lower
    float rd[C];                                           // 1 / L_ii
    float ld, alpha;
};
template <int C>
struct ParLayout {
    static constexpr int NPO = C * (C - 1) / 2;
    static constexpr int LRE = 0, LIM = NPO, RD = 2 * NPO, LD = 2 * NPO + C, ALPHA = LD + 1,
                         SIZE = ((ALPHA + 1 + 3) / 4) * 4;
};

template <int C>
ZD void load_par(const float* p, ClassPar<C>& cp) {
    typedef ParLayout<C> PL;
#pragma unroll
    for (int e = 0; e < PL::NPO; ++e) {
        cp.lre[e] = sgpr(p[PL::LRE + e]);
        cp.lim[e] = sgpr(p[PL::LIM + e]);
    }
#pragma unroll
    for (int i = 0; i < C; ++i) cp.rd[i] = sgpr(p[PL::RD + i]);
    cp.ld = sgpr(p[PL::LD]);
    cp.alpha = sgpr(p[PL::ALPHA]);
}

// q = | L^-1 x |^2 by forward substitution
template <int C>
ZD float quad_form(const cf (&x)[C], const ClassPar<C>& cp) {
    cf y[C];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        float re = x[i].x, im = x[i].y;
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const int e = i * (i - 1) / 2 + j;
            re = fmaf(-cp.lre[e], y[j].x, re);
            re = fmaf(cp.lim[e], y[j].y, re);
            im = fmaf(-cp.lre[e], y[j].y, im);
            im = fmaf(-cp.lim[e], y[j].x, im);
        }
        y[i].x = re * cp.rd[i];
        y[i].y = im * cp.rd[i];
        q = fmaf(y[i].x, y[i].x, q);
        q = fmaf(y[i].y, y[i].y, q);
    }
    return q;
}

template <int C>
struct Acc {
    static constexpr int NP = C * (C + 1) / 2, NPO = C * (C - 1) / 2;
    float re[2][NP];
    float im[2][NPO > 0 ? NPO : 1];
    float sg[2];
};

// acc_k += w_k x x^H (upper triangle i <= j, entry x_i conj(x_j))
template <int C, bool BOTH>
ZD void accumulate(const cf (&x)[C], float w0, float w1, Acc<C>& a) {
    int e = 0, eo = 0;
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
        for (int j = i; j < C; ++j) {
            if (i == j) {
                const float p = fmaf(x[i].x, x[i].x, x[i].y * x[i].y);
                a.re[0][e] = fmaf(w0, p, a.re[0][e]);
                if (BOTH) a.re[1][e] = fmaf(w1, p, a.re[1][e]);
            } else {
                const float pr = fmaf(x[i].x, x[j].x, x[i].y * x[j].y);
                const float pi = fmaf(x[i].y, x[j].x, -x[i].x * x[j].y);
                a.re[0][e] = fmaf(w0, pr, a.re[0][e]);
                a.im[0][eo] = fmaf(w0, pi, a.im[0][eo]);
                if (BOTH) {
                    a.re[1][e] = fmaf(w1, pr, a.re[1][e]);
                    a.im[1][eo] = fmaf(w1, pi, a.im[1][eo]);
                }
                ++eo;
            }
            ++e;
        }
}

// circle-method partner of index k in round r of a sweep over m (even) indices
ZD int rr_partner(int r, int k, int m) {
    const int n1 = m - 1;
    if (k == n1) return r;
    int j = 2 * r - k;
    j = j < 0 ? j + n1 : (j >= n1 ? j - n1 : j);
    return j == k ? n1 : j;
}

template <int C, int NT>
struct BinSmem {
    static constexpr int M = (C + 1) & ~1;  // Jacobi dimension (even)
    static constexpr int NP = C * (C + 1) / 2, NPO = C * (C - 1) / 2;
    static constexpr int NV = NP + NPO + 1;  // sums per class
    static constexpr int NW = NT / 64;
    zd A[2][M * M];
    zd V[2][M * M];
    zd W[2][M * M];
    double jp[2][M][4];          // per index: c, sigma.re, sigma.im
    double wv[2][M];             // w' per class
    float red[NW * 4][2 * NV];
    float par[2][ParLayout<C>::SIZE];
    int hasV[2];
};

// ---- the solve of one class by one wave -------------------------------------------------
template <int C, int NT>
__device__ __noinline__ void solve_class(BinSmem<C, NT>& sm, const int k, const int lane, const int mode, const int T,
                    const int update_alpha) {
    typedef BinSmem<C, NT> S;
    typedef ParLayout<C> PL;
    constexpr int M = S::M, NP = S::NP, NPO = S::NPO, NV = S::NV;
    const int i = lane / M, j = lane % M;
    const bool act = lane < M * M;
    const bool in = act && i < C && j < C;
    zd* A = sm.A[k];
    zd* V = sm.V[k];
    zd* W = sm.W[k];

    // --- R_ij from the row sums (float64) ---
    zd a = zmk(0.0, 0.0);
    double sumg = 0.0;
    {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int e = in ? pair_index(lo, hi, C) : 0;
        const int eo = (in && lo != hi) ? (lo * C - lo * (lo + 1) / 2 + (hi - lo - 1)) : 0;
        double re = 0.0, im = 0.0;
        for (int r = 0; r < S::NW * 4; ++r) {
            const float* row = sm.red[r] + k * NV;
            re += (double)row[e];
            im += (double)row[NP + eo];
            sumg += (double)row[NP + NPO];
        }
        if (mode == kModeInitId) {
            if (k == 0) {
                const double rt = 1.0 / (double)T;
                a = zmk(re * rt, (lo != hi) ? (i < j ? im : -im) * rt : 0.0);
            } else {
                a = zmk(i == j ? 1.0 : 0.0, 0.0);
            }
        } else {
            const double rd = 1.0 / fmax(sumg, kEpsD);
            a = zmk(re * rd, (lo != hi) ? (i < j ? im : -im) * rd : 0.0);
        }
        if (!in) a = zmk(0.0, 0.0);
    }
    if (update_alpha && mode == kModeEm && lane == 0) sm.par[k][PL::ALPHA] = (float)(sumg / (double)T);
    if (mode != kModeEm && lane == 0) sm.par[k][PL::ALPHA] = 0.5f;

    // --- power-of-two scale so that trace ~ 1 (float32 angle arithmetic stays in range) ---
    if (act) A[lane] = a;
    wave_lds_fence();
    double tr = 0.0;
#pragma unroll
    for (int d = 0; d < C; ++d) tr += A[d * M + d].x;
    int ex = 0;
    if (tr > 0.0 && tr < 1e300) (void)frexp(tr, &ex);
    const double scl = ldexp(1.0, ex), rscl = ldexp(1.0, -ex);
    a = zscale(a, rscl);
    tr *= rscl;
    const double fl = fmax(kEpsD * tr / (double)C, 1e-290);  // <= eps * w_max
    wave_lds_fence();

    // --- warm start: A <- V^H R V with the previous eigenvectors ---
    zd v = zmk(i == j ? 1.0 : 0.0, 0.0);
    const int warm = sm.hasV[k];
    if (warm) {
        if (act) A[lane] = a;
        wave_lds_fence();
        v = act ? V[lane] : zmk(0.0, 0.0);
        zd w = zmk(0.0, 0.0);
#pragma unroll
        for (int d = 0; d < M; ++d) {
            const zd r = A[(act ? i : 0) * M + d], vv = V[d * M + (act ? j : 0)];
            w = zadd(w, zmul(r, vv));
        }
        if (act) W[lane] = w;
        wave_lds_fence();
        zd t = zmk(0.0, 0.0);
#pragma unroll
        for (int d = 0; d < M; ++d) {
            const zd vv = V[d * M + (act ? i : 0)], ww = W[d * M + (act ? j : 0)];
            t = zadd(t, zcmul(vv, ww));
        }
        a = t;
        if (i == j) a.y = 0.0;
        wave_lds_fence();
    }

    // --- Jacobi sweeps ---
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        bool big = false;
        for (int r = 0; r < M - 1; ++r) {
            const int ip = rr_partner(r, act ? i : 0, M), jp = rr_partner(r, act ? j : 0, M);
            if (act) {
                A[lane] = a;
                V[lane] = v;
            }
            wave_lds_fence();
            // the lane holding a_pq (p < q partners) derives the rotation
            if (act && jp == i && i < j) {
                const double app = A[i * M + i].x, aqq = A[j * M + j].x;
                const double g2 = a.x * a.x + a.y * a.y;
                const double den = fmax(app, fl) * fmax(aqq, fl);
                double c = 1.0;
                zd sg = zmk(0.0, 0.0);  // s e^{i theta}
                if (g2 > kTol2 * den) {
                    if (g2 > kLast2 * den) big = true;
                    const float rg = __frsqrt_rn((float)g2);
                    const float tau = 0.5f * (float)(aqq - app) * rg;
                    if (fabsf(tau) < 1e18f) {
                        const float tf = copysignf(1.0f, tau) /
                                         (fabsf(tau) + __fsqrt_rn(fmaf(tau, tau, 1.0f)));
                        const double t = (double)tf;
                        c = rsq64(__builtin_fma(t, t, 1.0));
                        const double s = t * c;
                        const double rgd = rsq64(g2);
                        sg = zmk(a.x * rgd * s, a.y * rgd * s);
                    }
                }
                // index p (low): sigma = J_qp = -s conj(e); index q (high): J_pq = s e
                sm.jp[k][i][0] = c;
                sm.jp[k][i][1] = -sg.x;
                sm.jp[k][i][2] = sg.y;
                sm.jp[k][j][0] = c;
                sm.jp[k][j][1] = sg.x;
                sm.jp[k][j][2] = sg.y;
            }
            wave_lds_fence();
            if (act) {
                const double ci = sm.jp[k][i][0], cj = sm.jp[k][j][0];
                const zd si = zmk(sm.jp[k][i][1], sm.jp[k][i][2]);
                const zd sj = zmk(sm.jp[k][j][1], sm.jp[k][j][2]);
                const zd a_ijp = A[i * M + jp], a_ipj = A[ip * M + j], a_ipjp = A[ip * M + jp];
                const zd v_ijp = V[i * M + jp];
                // (A J)_{xj} = c_j A_xj + sigma_j A_xj'
                const zd b0 = zadd(zscale(a, cj), zmul(sj, a_ijp));
                const zd b1 = zadd(zscale(a_ipj, cj), zmul(sj, a_ipjp));
                // (J^H B)_{ij} = c_i B_ij + conj(sigma_i) B_i'j
                a = zadd(zscale(b0, ci), zcmul(si, b1));
                if (i == j) a.y = 0.0;
                v = zadd(zscale(v, cj), zmul(sj, v_ijp));
            }
            wave_lds_fence();
        }
        if (!__any(big)) break;
    }

    // --- scaled, floored eigenvalues (cluster.py:107-113) ---
    if (act) {
        A[lane] = a;
        V[lane] = v;
    }
    wave_lds_fence();
    double wmax = -1e300;
#pragma unroll
    for (int d = 0; d < C; ++d) wmax = fmax(wmax, A[d * M + d].x);
    const double rs = 1.0 / fmax(wmax * scl, kEpsD);
    if (act && i == j && i < C) sm.wv[k][i] = fmax(a.x * scl * rs, kEpsD);
    wave_lds_fence();
    float ld = 0.f;
#pragma unroll
    for (int d = 0; d < C; ++d) ld += logf((float)sm.wv[k][d]);

    // --- R_eff = V diag(w') V^H, then its Cholesky factor ---
    zd re = zmk(0.0, 0.0);
#pragma unroll
    for (int d = 0; d < C; ++d) {
        const zd vi = V[(in ? i : 0) * M + d], vj = V[(in ? j : 0) * M + d];
        re = zadd(re, zscale(zmulc(vi, vj), sm.wv[k][d]));
    }
    if (i == j) re.y = 0.0;
    wave_lds_fence();
    // right-looking Cholesky on the lower triangle; lane (i, j), i >= j, owns a_ij
    zd lcol = zmk(0.0, 0.0);  // final L_ij of this lane
#pragma unroll
    for (int kk = 0; kk < C; ++kk) {
        if (in) A[lane] = re;
        wave_lds_fence();
        if (in && i >= j && j >= kk) {
            const double akk = A[kk * M + kk].x;
            if (j == kk) {
                const double rq = rsq64(fmax(akk, 1e-300));
                lcol = (i == kk) ? zmk(akk * rq, rq) : zscale(re, rq);  // diagonal lane: (L_kk, 1 / L_kk)
            } else {
                const zd aik = A[i * M + kk], ajk = A[j * M + kk];
                re = zsub(re, zscale(zmulc(aik, ajk), rcp64(fmax(akk, 1e-300))));
                if (i == j) re.y = 0.0;
            }
        }
        wave_lds_fence();
    }
    if (in) {
        float* p = sm.par[k];
        if (i == j) {
            p[PL::RD + i] = (float)lcol.y;
        } else if (i > j) {
            const int e = i * (i - 1) / 2 + j;
            p[PL::LRE + e] = (float)lcol.x;
            p[PL::LIM + e] = (float)lcol.y;
        }
    }
    if (lane == 0) {
        sm.par[k][PL::LD] = ld;
        sm.hasV[k] = 1;
    }
}

// ---- the passes over the thread's frames -------------------------------------------------
// Frames are walked in chunks of NT x U (thread tid owns frames base + tid + NT u); per
// chunk the quadratic forms of class 0, then of class 1 (only ONE class's Cholesky factor
// sits in SGPRs at a time), then posterior + accumulation.  The first RF frames of
// chunk 0 live in registers (xr), everything else in the LDS tile Xs[C][Tlp].
template <int C, int NT, int RF, int U, bool FIRST, int MODE>
ZD void chunk_pass(const int base, const int tid, const int T, const cf (&xr)[RF > 0 ? RF : 1][C],
                   const cf* Xs, const int Tlp, const float* par0, const float* par1, Acc<C>& acc,
                   const CgmmBinArgs& a, const int f) {
    typedef ParLayout<C> PL;
    auto getx = [&](int u, int t, cf (&x)[C]) {
        if (FIRST && u < RF) {
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = xr[u < RF ? u : 0][c];
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = Xs[c * Tlp + (t - NT * RF)];
        }
    };
    if (MODE == kModeInitId || MODE == kModeInitMask) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = base + tid + NT * u;
            if (t < T) {
                cf x[C];
                getx(u, t, x);
                if (MODE == kModeInitId) {
                    accumulate<C, false>(x, 1.0f, 0.f, acc);
                } else {
                    const float g0 = a.init_mask[(size_t)t * a.F + f], g1 = 1.0f - g0;
                    acc.sg[0] += g0;
                    acc.sg[1] += g1;
                    accumulate<C, true>(x, g0, g1, acc);
                }
            }
            reload_fence();
        }
        return;
    }
    float q0[U], q1[U];
    {
        ClassPar<C> p;
        load_par<C>(par0, p);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = base + tid + NT * u;
            q0[u] = 1.f;
            if (t < T) {
                cf x[C];
                getx(u, t, x);
                q0[u] = quad_form<C>(x, p);
            }
            reload_fence();
        }
    }
    {
        ClassPar<C> p;
        load_par<C>(par1, p);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = base + tid + NT * u;
            q1[u] = 1.f;
            if (t < T) {
                cf x[C];
                getx(u, t, x);
                q1[u] = quad_form<C>(x, p);
            }
            reload_fence();
        }
    }
    const float ld0 = sgpr(par0[PL::LD]), ld1 = sgpr(par1[PL::LD]);
    const float al0 = sgpr(par0[PL::ALPHA]), al1 = sgpr(par1[PL::ALPHA]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int t = base + tid + NT * u;
        if (t < T) {
            // cluster.py:207-235, 261-287
            const float ph0 = fmaxf(q0[u], kEpsF) * (1.0f / C), ph1 = fmaxf(q1[u], kEpsF) * (1.0f / C);
            const float l0 = fmaf(-(float)C, __logf(ph0), -ld0);
            const float l1 = fmaf(-(float)C, __logf(ph1), -ld1);
            const float mx = fmaxf(l0, l1);
            const float n0 = al0 * __expf(l0 - mx), n1 = al1 * __expf(l1 - mx);
            const float rden = __frcp_rn(fmaxf(n0 + n1, kEpsF));
            const float g0 = n0 * rden, g1 = n1 * rden;
            if (MODE == kModeEm) {
                cf x[C];
                getx(u, t, x);
                acc.sg[0] += g0;
                acc.sg[1] += g1;
                accumulate<C, true>(x, g0 * __frcp_rn(ph0) * (float)C, g1 * __frcp_rn(ph1) * (float)C,
                                    acc);
            } else {
                a.gamma_bm[(size_t)f * a.Tp + t] = g0;
                if (a.nout > 1) a.gamma_bm[((size_t)a.F + f) * a.Tp + t] = g1;
            }
        }
        reload_fence();
    }
}

template <int C, int NT, int RF, int U, int MODE>
ZD void frames_pass(const int tid, const int T, const cf (&xr)[RF > 0 ? RF : 1][C], const cf* Xs,
                    const int Tlp, const float* par0, const float* par1, Acc<C>& acc,
                    const CgmmBinArgs& a, const int f) {
    chunk_pass<C, NT, RF, U, true, MODE>(0, tid, T, xr, Xs, Tlp, par0, par1, acc, a, f);
    for (int base = NT * U; base < T; base += NT * U)
        chunk_pass<C, NT, RF, U, false, MODE>(base, tid, T, xr, Xs, Tlp, par0, par1, acc, a, f);
}

template <int C, int NT, int RF, int WPS>
__global__ __launch_bounds__(NT, WPS) void cgmm_bin_em_kernel(const CgmmBinArgs* __restrict__ tbl,
                                                              int num_iters) {
    typedef BinSmem<C, NT> S;
    constexpr int NP = S::NP, NPO = S::NPO, NV = S::NV;
    constexpr int U = 4;
    static_assert(RF <= U, "register frames belong to chunk 0");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    S& sm = *reinterpret_cast<S*>(smem_raw);
    cf* Xs = reinterpret_cast<cf*>(smem_raw + ((sizeof(S) + 15) & ~(size_t)15));  // [C][Tlp]

    const CgmmBinArgs a = tbl[blockIdx.y];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int T = a.T, Tp = a.Tp;
    const int Tl = max(T - NT * RF, 0);          // frames kept in LDS
    const int Tlp = (Tl + 1) & ~1;               // row pitch of Xs

    // ---- load the bin: frames tid + NT i; i < RF in registers, the rest in LDS ----
    const cf* xg = a.xb + (size_t)f * C * Tp;
    cf xr[RF > 0 ? RF : 1][C];
#pragma unroll
    for (int i = 0; i < RF; ++i) {
        const int t = tid + NT * i;
#pragma unroll
        for (int c = 0; c < C; ++c) xr[i][c] = (t < T) ? xg[(size_t)c * Tp + t] : make_float2(0.f, 0.f);
    }
    for (int t = NT * RF + tid; t < T; t += NT) {
#pragma unroll
        for (int c = 0; c < C; ++c) Xs[c * Tlp + (t - NT * RF)] = xg[(size_t)c * Tp + t];
    }
    if (tid < 2) sm.hasV[tid] = 0;
    __syncthreads();

    const int npass = num_iters + 2;
    for (int pass = 0; pass < npass; ++pass) {
        const int mode = pass == 0 ? (a.init_mask ? kModeInitMask : kModeInitId)
                                   : (pass == npass - 1 ? kModeFinal : kModeEm);
        Acc<C> acc;
#pragma unroll
        for (int e = 0; e < NP; ++e) acc.re[0][e] = acc.re[1][e] = 0.f;
#pragma unroll
        for (int e = 0; e < (NPO > 0 ? NPO : 1); ++e) acc.im[0][e] = acc.im[1][e] = 0.f;
        acc.sg[0] = acc.sg[1] = 0.f;

        if (mode == kModeEm)
            frames_pass<C, NT, RF, U, kModeEm>(tid, T, xr, Xs, Tlp, sm.par[0], sm.par[1], acc, a, f);
        else if (mode == kModeFinal)
            frames_pass<C, NT, RF, U, kModeFinal>(tid, T, xr, Xs, Tlp, sm.par[0], sm.par[1], acc, a, f);
        else if (mode == kModeInitId)
            frames_pass<C, NT, RF, U, kModeInitId>(tid, T, xr, Xs, Tlp, sm.par[0], sm.par[1], acc, a, f);
        else
            frames_pass<C, NT, RF, U, kModeInitMask>(tid, T, xr, Xs, Tlp, sm.par[0], sm.par[1], acc, a, f);
        if (mode == kModeFinal) break;

        // ---- row sums -> LDS ----
        {
            float* row = sm.red[wave * 4 + (lane >> 4)];
            const bool wr = (lane & 15) == 15;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int e = 0; e < NP; ++e) {
                    const float s = row_sum16(acc.re[k][e]);
                    if (wr) row[k * NV + e] = s;
                }
#pragma unroll
                for (int e = 0; e < NPO; ++e) {
                    const float s = row_sum16(acc.im[k][e]);
                    if (wr) row[k * NV + NP + e] = s;
                }
                const float s = row_sum16(acc.sg[k]);
                if (wr) row[k * NV + NP + NPO] = s;
            }
        }
        __syncthreads();
        if (wave < 2) solve_class<C, NT>(sm, wave, lane, mode, T, a.update_alpha);
        __syncthreads();
    }
}

// ---- layout changes -----------------------------------------------------------------------
// [C][T][Fp] (bin fastest) -> [F][C][Tp] (frame fastest), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void spec_to_binmajor_kernel(const cf* const* __restrict__ src,
                                                               const CgmmBinArgs* __restrict__ tbl,
                                                               int C, int spitch, int ntt) {
    __shared__ cf tile[32][33];
    const CgmmBinArgs a = tbl[blockIdx.z];
    const int T = a.T, F = a.F, Tp = a.Tp;
    const int tt = blockIdx.x % ntt, c = blockIdx.x / ntt;
    const int f0 = blockIdx.y * 32, t0 = tt * 32;
    if (c >= C || t0 >= T) return;
    const cf* s = src[blockIdx.z] + (size_t)c * T * spitch;
    cf* d = const_cast<cf*>(a.xb);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, ff = f0 + tx;
        tile[ty + 8 * r][tx] = (t < T && ff < F) ? s[(size_t)t * spitch + ff] : make_float2(0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ff = f0 + ty + 8 * r, t = t0 + tx;
        if (ff < F && t < T) d[((size_t)ff * C + c) * Tp + t] = tile[tx][ty + 8 * r];
    }
}

// posteriors [nout][F][Tp] -> mask [T][F] (class 0) and optionally gamma [2][T][F]
__global__ __launch_bounds__(256) void binmajor_to_tf_kernel(const CgmmBinArgs* __restrict__ tbl,
                                                             float* const* __restrict__ mask_out,
                                                             float* const* __restrict__ gamma_out) {
    __shared__ float tile[32][33];
    const CgmmBinArgs a = tbl[blockIdx.z];
    const int T = a.T, F = a.F, Tp = a.Tp;
    const int nft = (F + 31) / 32;
    const int ft = blockIdx.y % nft, k = blockIdx.y / nft;
    if (k >= a.nout) return;
    const int t0 = blockIdx.x * 32, f0 = ft * 32;
    if (t0 >= T) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* s = a.gamma_bm + (size_t)k * F * Tp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ff = f0 + ty + 8 * r, t = t0 + tx;
        tile[ty + 8 * r][tx] = (ff < F && t < T) ? s[(size_t)ff * Tp + t] : 0.f;
    }
    __syncthreads();
    float* m = mask_out ? mask_out[blockIdx.z] : nullptr;
    float* g = gamma_out ? gamma_out[blockIdx.z] : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, ff = f0 + tx;
        if (t < T && ff < F) {
            const float v = tile[tx][ty + 8 * r];
            if (k == 0 && m) m[(size_t)t * F + ff] = v;
            if (g) g[((size_t)k * T + t) * F + ff] = v;
        }
    }
}

template <int C, int NT, int RF>
size_t bin_lds_bytes(int max_frames) {
    const int Tl = std::max(max_frames - NT * RF, 0);
    const int Tlp = (Tl + 1) & ~1;
    return ((sizeof(BinSmem<C, NT>) + 15) & ~(size_t)15) + (size_t)C * Tlp * sizeof(cf);
}

constexpr size_t kLdsLimit = 160 * 1024;

template <int C, int NT, int RF, int WPS>
hipError_t launch_bin_t(const CgmmBinArgs* d_tbl, int n_utts, int F, int max_frames, int num_iters,
                        hipStream_t s) {
    const size_t lds = bin_lds_bytes<C, NT, RF>(max_frames);
    auto kern = cgmm_bin_em_kernel<C, NT, RF, WPS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(F, n_utts), dim3(NT), lds, s, d_tbl, num_iters);
    return hipGetLastError();
}

template <int C>
hipError_t launch_bin_c(const CgmmBinArgs* d_tbl, int n_utts, int F, int max_frames, int num_iters,
                        int nt, hipStream_t s) {
    if (nt == 256) return launch_bin_t<C, 256, 1, 4>(d_tbl, n_utts, F, max_frames, num_iters, s);
    return launch_bin_t<C, 512, 1, 4>(d_tbl, n_utts, F, max_frames, num_iters, s);
}

template <int C>
size_t lds_need_c(int max_frames, int nt) {
    return nt == 256 ? bin_lds_bytes<C, 256, 1>(max_frames) : bin_lds_bytes<C, 512, 1>(max_frames);
}

size_t lds_need(int C, int max_frames, int nt) {
    switch (C) {
        case 1: return lds_need_c<1>(max_frames, nt);
        case 2: return lds_need_c<2>(max_frames, nt);
        case 3: return lds_need_c<3>(max_frames, nt);
        case 4: return lds_need_c<4>(max_frames, nt);
        case 5: return lds_need_c<5>(max_frames, nt);
        case 6: return lds_need_c<6>(max_frames, nt);
        case 7: return lds_need_c<7>(max_frames, nt);
        case 8: return lds_need_c<8>(max_frames, nt);
    }
    return (size_t)-1;
}

}  // namespace

size_t cgmm_bin_args_bytes() { return sizeof(CgmmBinArgs); }

// frames pitch of the bin-major arrays
int cgmm_bin_pitch(int T) { return (T + 3) & ~3; }

// threads per workgroup for a batch whose longest utterance has max_frames frames;
// 0 when the bin does not fit the CU (caller then uses the streaming kernels of cgmm.hip)
int cgmm_bin_threads(int C, int max_frames) {
    if (C < 1 || C > kMaxChannels) return 0;
    const int nt = max_frames > 1024 ? 512 : 256;
    if (lds_need(C, max_frames, nt) <= kLdsLimit) return nt;
    if (nt == 256 && lds_need(C, max_frames, 512) <= kLdsLimit) return 512;
    return 0;
}

void cgmm_bin_fill_args(void* out, const float* xb, const float* init_mask, float* gamma_bm, int T,
                        int F, int update_alpha, int nout) {
    CgmmBinArgs a;
    std::memset(&a, 0, sizeof(a));
    a.xb = reinterpret_cast<const cf*>(xb);
    a.init_mask = init_mask;
    a.gamma_bm = gamma_bm;
    a.T = T;
    a.Tp = cgmm_bin_pitch(T);
    a.F = F;
    a.update_alpha = update_alpha;
    a.nout = nout;
    std::memcpy(out, &a, sizeof(a));
}

hipError_t launch_cgmm_bin(int C, const void* d_tbl, const float* const* d_spec_ptrs, int spec_pitch,
                           float* const* d_mask_ptrs, float* const* d_gamma_ptrs, int n_utts, int F,
                           int max_frames, int num_iters, int nout, hipStream_t s) {
    const CgmmBinArgs* t = static_cast<const CgmmBinArgs*>(d_tbl);
    const int nt = cgmm_bin_threads(C, max_frames);
    if (!nt) return hipErrorInvalidValue;
    {
        const int ntt = (max_frames + 31) / 32;
        dim3 g(ntt * C, (F + 31) / 32, n_utts);
        hipLaunchKernelGGL(spec_to_binmajor_kernel, g, dim3(256), 0, s,
                           reinterpret_cast<const cf* const*>(d_spec_ptrs), t, C, spec_pitch, ntt);
    }
    hipError_t e = hipErrorInvalidValue;
    switch (C) {
        case 1: e = launch_bin_c<1>(t, n_utts, F, max_frames, num_iters, nt, s); break;
        case 2: e = launch_bin_c<2>(t, n_utts, F, max_frames, num_iters, nt, s); break;
        case 3: e = launch_bin_c<3>(t, n_utts, F, max_frames, num_iters, nt, s); break;
        case 4: e = launch_bin_c<4>(t, n_utts, F, max_frames, num_iters, nt, s); break;
        case 5: e = launch_bin_c<5>(t, n_utts, F, max_frames, num_iters, nt, s); break;
        case 6: e = launch_bin_c<6>(t, n_utts, F, max_frames, num_iters, nt, s); break;
        case 7: e = launch_bin_c<7>(t, n_utts, F, max_frames, num_iters, nt, s); break;
        case 8: e = launch_bin_c<8>(t, n_utts, F, max_frames, num_iters, nt, s); break;
    }
    if (e != hipSuccess) return e;
    {
        dim3 g((max_frames + 31) / 32, ((F + 31) / 32) * nout, n_utts);
        hipLaunchKernelGGL(binmajor_to_tf_kernel, g, dim3(256), 0, s, t, d_mask_ptrs, d_gamma_ptrs);
    }
    return hipGetLastError();
}

}  // namespace setk
