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
// Per pass over the T frames (thread t owns frames t + NT u, u < U; the first RF of them
// in registers, the others in the LDS tile), in phases that each keep few registers live:
//   E0, E1  q_k = x^H R_k^-1 x by forward substitution with the Cholesky factor of the
//       (eigenvalue-normalised, floored) R_k -- y = L^-1 x, q = |y|^2: every term
//       positive, float32 loses only sqrt(cond) where the dense form loses cond.  One
//       class's factor (C^2 floats, uniform over the workgroup) sits in SGPRs at a time.
//   P   phi_k = max(q_k, eps) / M; log N_k = -M log phi_k - log det R_k; posterior;
//       weights gamma_k M / phi_k.
//   R, I  real, then imaginary parts of  sum_t w_k x x^H  in 2 x C(C+1)/2 (2 x C(C-1)/2)
//       register accumulators, a halving DPP butterfly over the wavefront in float32, wave
//       sums to LDS; the waves' sums are added in float64 by the solving wave.
// Between two passes, wave k (k = 0, 1) solves class k on (C x C) lanes in float64:
//   fast path (EM iterations of bins without near-silent frames): Cholesky of R_k and a
//   float32 bound  lambda_min >= 1 / trace(R^-1) >= 1.1 eps trace(R)  certifying that the
//   reference's eigenvalue floor is inactive -- then its normalisation by lambda_max is a
//   pure scale that cancels between phi and log det, and the factor of R itself serves;
//   otherwise: two-sided Jacobi in the parallel (round-robin) order, warm-started from
//   the previous eigenvectors (A = V^H R V), rotation angles in float32 and the rotations
//   themselves exactly unitary in float64, the sweep that starts with every
//   |a_pq|^2 <= 1e-8 a_pp a_qq is the last -> eigenvalues scaled by 1 / max(w_max, eps)
//   and floored at eps (cluster.py:107-113) -> R_eff = V diag(w') V^H -> Cholesky.
// Layout in: bin-major spectrogram [F][C][Tp] (frames contiguous; written by
// spec_to_binmajor_kernel from the [C][T][Fp] dump); out: bin-major posteriors
// [F][Tp], transposed back to the reference's T x F by binmajor_to_tf_kernel.
// Roofline: VALU issue (about 340 float32 instructions per frame and iteration);
// HBM traffic is one read of the spectrogram + one write of the masks.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
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

// Diagnostic build (-DSETK_CGMM_PHASES, tools/cgmm_phases.sh): wave 0's clock at the phase
// boundaries of a pass, summed per bin into slots 8.. of the SETK_CGMM_TIMING array.  The
// product build reads no clock inside a pass.
constexpr int kTimingSlots = 24;
#ifdef SETK_CGMM_PHASES
#define PH_BEGIN(tm, on) long long ph_t_ = ((tm) && (on)) ? (long long)__builtin_readcyclecounter() : 0
#define PH(tm, on, slot)                                                     \
    do {                                                                     \
        if ((tm) && (on)) {                                                  \
            const long long n_ = (long long)__builtin_readcyclecounter();    \
            (tm)[slot] += n_ - ph_t_;                                        \
            ph_t_ = n_;                                                      \
        }                                                                    \
    } while (0)
#else
#define PH_BEGIN(tm, on)
#define PH(tm, on, slot)
#endif

struct CgmmBinArgs {
    const cf* xb;            // [F][C][Tp]
    const float* init_mask;  // [T][F] or null
    float* gamma_bm;         // [nout][F][Tp]
    long long* timing;       // diagnostic (SETK_CGMM_TIMING): [F][kTimingSlots] cycle counts, or null
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
ZD void reload_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int CTRL, int RM = 0xf>
ZD float dppf(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, true));
}
// Sums of N per-lane values over the wavefront (pairwise tree in float32) as a HALVING
// butterfly: the two steps inside a quad each halve the number of values a lane carries
// (a lane keeps one half of its values and hands the other half to its partner: two
// selects + one DPP add per surviving value), the steps across the quads of a row
// (row_shr:4, row_shr:8) and across the rows (two ds_bpermute hops) then run on a quarter
// of the values: 3.25 N + 4 N / 4 instructions against 6 N for the plain tree, and the
// totals end up spread over lanes 60..63, which store them with one instruction each.
// Lane 60 + p holds in out[r] the total of value (p & 1) N1 + (p >> 1) N2 + r.
template <int N>
struct Bfly {
    static constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2;
};
template <int N>
ZD void butterfly_sum(const float (&v)[N], float (&out)[Bfly<N>::N2], const int lane) {
    constexpr int N1 = Bfly<N>::N1, N2 = Bfly<N>::N2;
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    float w[N1];
#pragma unroll
    for (int r = 0; r < N1; ++r) {
        const float lo = v[r], hi = (N1 + r < N) ? v[N1 + r] : 0.f;
        const float keep = b0 ? hi : lo, send = b0 ? lo : hi;
        w[r] = keep + dppf<0xB1>(send);  // quad_perm:[1,0,3,2]
    }
#pragma unroll
    for (int r = 0; r < N2; ++r) {
        const float lo = w[r], hi = (N2 + r < N1) ? w[N2 + r] : 0.f;
        const float keep = b1 ? hi : lo, send = b1 ? lo : hi;
        out[r] = keep + dppf<0x4E>(send);  // quad_perm:[2,3,0,1]
    }
#pragma unroll
    for (int r = 0; r < N2; ++r) out[r] += dppf<0x114>(out[r]);  // row_shr:4
#pragma unroll
    for (int r = 0; r < N2; ++r) out[r] += dppf<0x118>(out[r]);  // row_shr:8 -> lanes 12..15
    const int up16 = ((lane - 16) & 63) * 4, up32 = ((lane - 32) & 63) * 4;
#pragma unroll
    for (int r = 0; r < N2; ++r)
        out[r] += __builtin_bit_cast(
            float, __builtin_amdgcn_ds_bpermute(up16, __builtin_bit_cast(int, out[r])));
#pragma unroll
    for (int r = 0; r < N2; ++r)
        out[r] += __builtin_bit_cast(
            float, __builtin_amdgcn_ds_bpermute(up32, __builtin_bit_cast(int, out[r])));
}
// flat index of out[r] in lane 60 + p, or -1 (padding)
template <int N>
ZD int butterfly_index(const int p, const int r) {
    constexpr int N1 = Bfly<N>::N1, N2 = Bfly<N>::N2;
    const int i1 = (p >> 1) * N2 + r;
    const int i = (p & 1) * N1 + i1;
    return (i1 < N1 && i < N) ? i : -1;
}

template <int N, typename Fn, int I = 0>
ZD void static_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        static_for<N, Fn, I + 1>(static_cast<Fn&&>(fn));
    }
}

ZD float sgprl(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
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

// ---- packed fp32 (v_pk_*_f32: one instruction on an aligned VGPR pair) -------------------------
// SETK_CGMM_PK=1: the frames phases hold a complex sample as a pair (re, im) and the two classes'
// accumulators of one outer-product entry as a pair (class 0, class 1).  A complex multiply-
// accumulate of the forward substitution is two packed instructions where the plain form needs
// four, x_i conj(x_j) two instead of four, the two classes' weighted sums of one entry ONE
// instead of two; broadcasts, swaps and signs ride in op_sel / neg (inline assembly: the compiler
// folds broadcasts but not the mixed negations; not volatile, so it schedules freely).  The waves
// of this kernel are issue-limited one by one (section 5 of DESIGN.md), which is where a packed
// instruction pays: round 2 measured none at the occupancy of the streaming kernels.
#ifndef SETK_CGMM_PK
#define SETK_CGMM_PK 1
#endif
typedef float pk2 __attribute__((ext_vector_type(2)));
// t - (l.x + i l.y) * y with a uniform l (an SGPR pair): the step of y = L^-1 x
ZD pk2 pk_cmsub_s(pk2 t, pk2 l, pk2 y) {
    pk2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
        : "=v"(r) : "s"(l), "v"(y), "v"(t));                      // r = t - l.x (y.x, y.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]"
        : "=v"(t) : "s"(l), "v"(y), "v"(r));                      // t = r + l.y (y.y, -y.x)
    return t;
}
// a * conj(w) = (a.x w.x + a.y w.y, a.y w.x - a.x w.y)
ZD pk2 pk_cmulc(pk2 a, pk2 w) {
    pk2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));      // (a.x w.x, a.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));                                        // + (a.y w.y, -a.x w.y)
    return r;
}
// c + (p.x, p.x) * w   /   c + (p.y, p.y) * w
ZD pk2 pk_fma_bc_lo(pk2 p, pk2 w, pk2 c) {
    pk2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(p), "v"(w), "v"(c));
    return r;
}
ZD pk2 pk_fma_bc_hi(pk2 p, pk2 w, pk2 c) {
    pk2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(p), "v"(w), "v"(c));
    return r;
}
// t * (s.x, s.x) / t * (s.y, s.y) with a uniform s
ZD pk2 pk_scale_s_lo(pk2 t, pk2 s) {
    pk2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "s"(s), "v"(t));
    return r;
}
ZD pk2 pk_scale_s_hi(pk2 t, pk2 s) {
    pk2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "s"(s), "v"(t));
    return r;
}

// the packed form of ClassPar: (L_ij.re, L_ij.im) and (1 / L_ii, 1 / L_i+1,i+1) as uniform pairs
template <int C>
struct ClassParPk {
    static constexpr int NPO = C * (C - 1) / 2;
    pk2 l[NPO > 0 ? NPO : 1];
    pk2 rd[(C + 1) / 2];
};
template <int C>
ZD void load_par_pk(const float* p, ClassParPk<C>& cp) {
    typedef ParLayout<C> PL;
#pragma unroll
    for (int e = 0; e < PL::NPO; ++e) cp.l[e] = (pk2){sgpr(p[PL::LRE + e]), sgpr(p[PL::LIM + e])};
#pragma unroll
    for (int i = 0; i < (C + 1) / 2; ++i)
        cp.rd[i] = (pk2){sgpr(p[PL::RD + 2 * i]), (2 * i + 1 < C) ? sgpr(p[PL::RD + 2 * i + 1]) : 0.f};
}
// q = | L^-1 x |^2, the operations of quad_form pair by pair
template <int C>
ZD float quad_form_pk(const pk2 (&x)[C], const ClassParPk<C>& cp) {
    pk2 y[C];
    pk2 q2 = (pk2){0.f, 0.f};
#pragma unroll
    for (int i = 0; i < C; ++i) {
        pk2 t = x[i];
#pragma unroll
        for (int j = 0; j < i; ++j) t = pk_cmsub_s(t, cp.l[i * (i - 1) / 2 + j], y[j]);
        y[i] = (i & 1) ? pk_scale_s_hi(t, cp.rd[i / 2]) : pk_scale_s_lo(t, cp.rd[i / 2]);
        q2 = __builtin_elementwise_fma(y[i], y[i], q2);
    }
    return q2.x + q2.y;
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
    static constexpr int NV = NP + NPO + 1;  // sums per class: re | im | sum gamma
    static constexpr int NW = NT / 64;
    zd A[2][M * M];
    zd V[2][M * M];
    zd W[2][M * M];
    double jp[2][M][4];          // per index: c, sigma.re, sigma.im
    double wv[2][M];             // w' per class
    double tot[2][NV];           // float64 sums over the rows of `red`
    float red[NW][2 * NV];
    float par[2][ParLayout<C>::SIZE];
    float cert[2][M];
    int hasV[2];
    int tiny;                    // some frame has |x|^2 < 1.5 eps: phi's floor may act
    int nfast[2];
};

// Cholesky factor of the Hermitian positive definite matrix whose entry (i, j) the lane
// holds in `re` (lower triangle used), cooperatively through sm.A[k].  Returns the lane's
// L_ij (i > j) or, on a diagonal lane, (L_ii, 1 / L_ii).  ok = every pivot positive.
template <int C, int M>
ZD zd chol_lanes(zd* A, zd re, const int i, const int j, const int lane, const bool in, bool& ok) {
    zd lcol = zmk(0.0, 0.0);
    ok = true;
#pragma unroll
    for (int kk = 0; kk < C; ++kk) {
        if (in) A[lane] = re;
        wave_lds_fence();
        if (in && i >= j && j >= kk) {
            const double akk = A[kk * M + kk].x;
            if (!(akk > 0.0)) ok = false;
            const double rq = rsq64(fmax(akk, 1e-300));
            if (j == kk) {
                lcol = (i == kk) ? zmk(akk * rq, rq) : zscale(re, rq);
            } else {
                const zd aik = A[i * M + kk], ajk = A[j * M + kk];
                re = zsub(re, zscale(zmulc(aik, ajk), rq * rq));
                if (i == j) re.y = 0.0;
            }
        }
        wave_lds_fence();
    }
    return lcol;
}

template <int C>
ZD void write_factor(float* p, const zd lcol, const int i, const int j, const bool in) {
    typedef ParLayout<C> PL;
    if (in) {
        if (i == j) {
            p[PL::RD + i] = (float)lcol.y;
        } else if (i > j) {
            const int e = i * (i - 1) / 2 + j;
            p[PL::LRE + e] = (float)lcol.x;
            p[PL::LIM + e] = (float)lcol.y;
        }
    }
}

// ---- the solve of one class by one wave -------------------------------------------------
// exact: reproduce the reference's absolute max(w_max, eps) (initial covariance; bins with
// near-silent frames).  Otherwise the matrix may carry an arbitrary scale (fast passes drop
// it) and only relative quantities are used -- equal to the reference's because there
// lambda_max(R) >= M eps holds by construction (trace(R_eff^-1 R) = M^2, R_eff^-1 <= I / eps).
template <int C, int NT>
__device__ __noinline__ void solve_class(BinSmem<C, NT>& sm, const int k, const int lane,
                                         const int mode, const int T, const int update_alpha,
                                         long long* tm) {
    const bool ph_on = (k == 0 && lane == 0);
    (void)tm;
    (void)ph_on;
    PH_BEGIN(tm, ph_on);
    typedef BinSmem<C, NT> S;
    typedef ParLayout<C> PL;
    constexpr int M = S::M, NP = S::NP, NPO = S::NPO, NV = S::NV;
    const int i = lane / M, j = lane % M;
    const bool act = lane < M * M;
    const bool in = act && i < C && j < C;
    zd* A = sm.A[k];
    zd* V = sm.V[k];
    zd* W = sm.W[k];
    float* par = sm.par[k];
    const bool exact = (mode != kModeEm) || sm.tiny;

    if (mode == kModeInitId && k == 1) {
        // R = I (cluster.py:423): L = I, log det = 0, eigenvectors = I
        if (in) {
            if (i == j) par[PL::RD + i] = 1.0f;
            if (i > j) {
                par[PL::LRE + i * (i - 1) / 2 + j] = 0.f;
                par[PL::LIM + i * (i - 1) / 2 + j] = 0.f;
            }
        }
        if (act) V[lane] = zmk(i == j ? 1.0 : 0.0, 0.0);
        if (lane == 0) {
            par[PL::LD] = 0.f;
            par[PL::ALPHA] = 0.5f;
            sm.hasV[k] = 1;
        }
        return;
    }

    // --- float64 sums of the row partials ---
    for (int v = lane; v < NV; v += 64) {
        double t = 0.0;
#pragma unroll 4
        for (int r = 0; r < S::NW; ++r) t += (double)sm.red[r][k * NV + v];
        sm.tot[k][v] = t;
    }
    wave_lds_fence();
    PH(tm, ph_on, 16);
    zd a = zmk(0.0, 0.0);
    const double sumg = sm.tot[k][NV - 1];
    if (in) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int e = pair_index(lo, hi, C);
        const double re = sm.tot[k][e];
        double im = 0.0;
        if (lo != hi) {
            im = sm.tot[k][NP + lo * C - lo * (lo + 1) / 2 + (hi - lo - 1)];
            if (i > j) im = -im;
        }
        const double rd = (mode == kModeInitId) ? 1.0 / (double)T : 1.0 / fmax(sumg, kEpsD);
        a = zmk(re * rd, im * rd);
    }
    if (lane == 0) {
        if (mode != kModeEm) par[PL::ALPHA] = 0.5f;
        else if (update_alpha) par[PL::ALPHA] = (float)(sumg / (double)T);
    }

    // --- power-of-two scale: trace in [0.5, 1) (the diagonal sums are in `tot`: no round trip) ---
    double tr = 0.0;
    {
        const double rd = (mode == kModeInitId) ? 1.0 / (double)T : 1.0 / fmax(sumg, kEpsD);
#pragma unroll
        for (int d = 0; d < C; ++d) tr += sm.tot[k][pair_index(d, d, C)] * rd;
    }
    int ex = 0;
    if (tr > 0.0 && tr < 1e300) ex = __builtin_amdgcn_frexp_exp(tr);
    const double scl = __builtin_amdgcn_ldexp(1.0, ex), rscl = __builtin_amdgcn_ldexp(1.0, -ex);
    a = zscale(a, rscl);
    tr *= rscl;
    PH(tm, ph_on, 17);

    // --- fast path: Cholesky of R itself when the eigenvalue floor provably rests ---
    if (!exact) {
        bool ok;
        const zd lcol = chol_lanes<C, M>(A, a, i, j, lane, in, ok);
        PH(tm, ph_on, 18);
        // P = prod 1 / L_ii (uniform: the diagonal lanes are known at compile time), det R = 1 / P^2
        double pd = 1.0;
#pragma unroll
        for (int d = 0; d < C; ++d) {
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double2loint(lcol.y), d * M + d);
            const unsigned hi = __builtin_amdgcn_readlane((unsigned)__double2hiint(lcol.y), d * M + d);
            pd *= __hiloint2double((int)hi, (int)lo);
        }
        const bool all_ok = __all(ok);
        // First certificate, from what the factorisation already holds (eigenvalues l_1 >= .. >= l_C):
        //   l_C = det / (l_1 .. l_{C-1}),   l_1 .. l_{C-1} <= f(l_1) = l_1 ((trace - l_1) / (C - 2))^(C - 2)   (AM-GM
        //   on the middle ones), f rises up to trace / (C - 1) and falls after it, and
        //   l_1 >= |R|_F^2 / trace  (sum l_i^2 <= l_1 sum l_i)  -- so  l_C >= det / f(max(|R|_F^2 / trace, trace / (C - 1))).
        // With  1 >= 1.1 eps trace f P^2  (det = 1 / P^2) that is  l_C >= 1.1 eps trace >= 1.1 eps l_1: the
        // reference's eigenvalue floor rests.  One wave sum and a dozen uniform operations; the L^-1
        // bound below decides what this one leaves open.
        float fr = in ? (float)(a.x * a.x + a.y * a.y) : 0.f;
        fr += dppf<0xB1>(fr);    // quad_perm:[1,0,3,2]
        fr += dppf<0x4E>(fr);    // quad_perm:[2,3,0,1]
        fr += dppf<0x141>(fr);   // row_half_mirror
        fr += dppf<0x140>(fr);   // row_mirror: every lane of a row holds the row's sum
        const float fro = sgprl(fr, 0) + sgprl(fr, 16) + sgprl(fr, 32) + sgprl(fr, 48);
        double fP;
        if constexpr (C >= 3) {
            const double m = (double)fro * (1.0 - 1e-5) / tr;   // (float32 sum of <= 64 terms: 4e-6)
            const double x = fmax(m, tr / (double)(C - 1));
            const double y = fmax(tr - x, 0.0) * (1.0 / (double)(C - 2));
            fP = x;
#pragma unroll
            for (int d = 0; d < C - 2; ++d) fP *= y;
        } else {
            fP = (C == 2) ? tr : 1.0;
        }
        bool certified = all_ok && pd > 0.0 && pd < 1e140 && fP > 0.0 && (1.1 * kEpsD * tr * fP * pd * pd <= 1.0);
        if (!certified) {
            // Second certificate: lambda_min >= 1 / trace(R^-1), trace(R^-1) = |L^-1|_F^2.
            // A now holds garbage of the elimination; publish L for the bound
            if (in && i >= j) W[lane] = (i == j) ? zmk(lcol.x, lcol.y) : lcol;
            wave_lds_fence();
            // column `lane` of L^-1 by forward substitution (float32 is plenty for a bound)
            float n2 = 0.f;
            if (lane < C) {
                float yr[C], yi[C];
#pragma unroll
                for (int r = 0; r < C; ++r) {
                    float sr = (r == lane) ? 1.f : 0.f, si = 0.f;
#pragma unroll
                    for (int c2 = 0; c2 < r; ++c2) {
                        const zd l = W[r * M + c2];
                        const float lr = (float)l.x, li = (float)l.y;
                        sr -= lr * yr[c2] - li * yi[c2];
                        si -= lr * yi[c2] + li * yr[c2];
                    }
                    const float rdg = (float)W[r * M + r].y;
                    yr[r] = (r < lane) ? 0.f : sr * rdg;
                    yi[r] = (r < lane) ? 0.f : si * rdg;
                    n2 += yr[r] * yr[r] + yi[r] * yi[r];
                }
                sm.cert[k][lane] = n2;
            }
            wave_lds_fence();
            float tinv = 0.f;
#pragma unroll
            for (int d = 0; d < C; ++d) tinv += sm.cert[k][d];
            // lambda_min >= 1 / trace(R^-1); lambda_max <= trace(R)
            certified = all_ok && (tinv * (float)tr * (1.1f * kEpsF) < 1.0f) && tinv > 0.f;
        }
        PH(tm, ph_on, 19);
        if (certified) {
            write_factor<C>(par, lcol, i, j, in);
            if (lane == 0) {
                // log det R = -2 log P, P = m 2^e
                const double m = __builtin_amdgcn_frexp_mant(pd);
                const int e = __builtin_amdgcn_frexp_exp(pd);
                par[PL::LD] = -2.0f * (__logf((float)m) + (float)e * 0.6931471805599453f);
                sm.hasV[k] = 0;  // the eigenvectors are stale now
                sm.nfast[k] += 1;
            }
            PH(tm, ph_on, 20);
            return;
        }
        wave_lds_fence();
    }

    const double fl = fmax(kEpsD * tr / (double)C, 1e-290);  // <= eps * w_max
    // --- warm start: A <- V^H R V with the previous eigenvectors ---
    zd v = zmk(i == j ? 1.0 : 0.0, 0.0);
    if (sm.hasV[k]) {
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
    const double sabs = exact ? scl : 1.0;
    const double rs = exact ? 1.0 / fmax(wmax * scl, kEpsD) : rcp64(fmax(wmax, 1e-300));
    if (act && i == j && i < C) {
        const double wp = fmax(a.x * sabs * rs, kEpsD);
        sm.wv[k][i] = wp;
        sm.cert[k][i] = __logf((float)wp);
    }
    wave_lds_fence();
    float ld = 0.f;
#pragma unroll
    for (int d = 0; d < C; ++d) ld += sm.cert[k][d];

    // --- R_eff = V diag(w') V^H, then its Cholesky factor ---
    zd re = zmk(0.0, 0.0);
#pragma unroll
    for (int d = 0; d < C; ++d) {
        const zd vi = V[(in ? i : 0) * M + d], vj = V[(in ? j : 0) * M + d];
        re = zadd(re, zscale(zmulc(vi, vj), sm.wv[k][d]));
    }
    if (i == j) re.y = 0.0;
    wave_lds_fence();
    bool ok;
    const zd lcol = chol_lanes<C, M>(A, re, i, j, lane, in, ok);
    write_factor<C>(par, lcol, i, j, in);
    if (lane == 0) {
        par[PL::LD] = ld;
        sm.hasV[k] = 1;
    }
    PH(tm, ph_on, 21);   // the exact path: everything after the fast path's attempt
}

// ---- one pass over the thread's frames -----------------------------------------------------
// Thread tid owns frames t_u = tid + NT u, u < U (one chunk: T <= NT U); frames u < RF live
// in registers, the others in Xs[c][tid + NT (u - RF)].
template <int C, int NT, int U, int RF, int MODE>
ZD void frames_pass(BinSmem<C, NT>& sm, const int tid, const int T, const cf (&xr)[RF > 0 ? RF : 1][C],
                    const cf* Xs, const int Tlp, const CgmmBinArgs& a, const int f) {
    typedef BinSmem<C, NT> S;
    typedef ParLayout<C> PL;
    constexpr int NP = S::NP, NPO = S::NPO, NV = S::NV;
    auto getx = [&](auto uc, cf (&x)[C]) {
        constexpr int u = decltype(uc)::value;
        if constexpr (u < RF) {
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = xr[u][c];
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = Xs[c * Tlp + tid + NT * (u - RF)];
        }
    };
    constexpr bool PK = SETK_CGMM_PK != 0;
    auto getxp = [&](auto uc, pk2 (&x)[C]) {
        constexpr int u = decltype(uc)::value;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const cf v = (u < RF) ? xr[u < RF ? u : 0][c] : Xs[c * Tlp + tid + NT * (u < RF ? 0 : u - RF)];
            x[c] = (pk2){v.x, v.y};
        }
    };
    float w0[U], w1[U];
    float sg0 = 0.f, sg1 = 0.f;
    long long* tm = a.timing ? a.timing + (size_t)f * kTimingSlots : nullptr;
    (void)tm;
    PH_BEGIN(tm, tid == 0);
    if constexpr (MODE == kModeInitId) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            w0[u] = 1.0f;
            w1[u] = 0.f;
        }
    } else if constexpr (MODE == kModeInitMask) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tid + NT * u;
            const float g0 = (t < T) ? a.init_mask[(size_t)t * a.F + f] : 0.f;
            w0[u] = g0;
            w1[u] = 1.0f - g0;
            if (t < T) {
                sg0 += g0;
                sg1 += 1.0f - g0;
            }
        }
    } else {
        float q0[U], q1[U];
        auto e_phase = [&](const float* par, float (&q)[U]) __attribute__((always_inline)) {
            if constexpr (PK) {
                ClassParPk<C> p;
                load_par_pk<C>(par, p);
                static_for<U>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    q[u] = 1.f;
                    if (tid + NT * u < T) {
                        pk2 x[C];
                        getxp(uc, x);
                        q[u] = quad_form_pk<C>(x, p);
                    }
                    reload_fence();
                });
            } else {
                ClassPar<C> p;
                load_par<C>(par, p);
                static_for<U>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    q[u] = 1.f;
                    if (tid + NT * u < T) {
                        cf x[C];
                        getx(uc, x);
                        q[u] = quad_form<C>(x, p);
                    }
                    reload_fence();
                });
            }
        };
        e_phase(sm.par[0], q0);
        reload_fence();
        PH(tm, tid == 0, 8);
        e_phase(sm.par[1], q1);
        reload_fence();
        PH(tm, tid == 0, 9);
        constexpr float kLog2e = 1.4426950408889634f;
        const float ld0 = sgpr(sm.par[0][PL::LD]) * kLog2e, ld1 = sgpr(sm.par[1][PL::LD]) * kLog2e;
        const float al0 = sgpr(sm.par[0][PL::ALPHA]), al1 = sgpr(sm.par[1][PL::ALPHA]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tid + NT * u;
            // cluster.py:207-235, 261-287, in base 2 on the hardware log2 / exp2 / rcp
            // (1 ulp each; phi >= eps / M is far from the subnormal range)
            const float ph0 = fmaxf(q0[u], kEpsF) * (1.0f / C), ph1 = fmaxf(q1[u], kEpsF) * (1.0f / C);
            const float l0 = fmaf(-(float)C, __builtin_amdgcn_logf(ph0), -ld0);
            const float l1 = fmaf(-(float)C, __builtin_amdgcn_logf(ph1), -ld1);
            const float mx = fmaxf(l0, l1);
            const float n0 = al0 * __builtin_amdgcn_exp2f(l0 - mx), n1 = al1 * __builtin_amdgcn_exp2f(l1 - mx);
            const float rden = __builtin_amdgcn_rcpf(fmaxf(n0 + n1, kEpsF));
            const float g0 = n0 * rden, g1 = n1 * rden;
            if constexpr (MODE == kModeEm) {
                w0[u] = g0 * __builtin_amdgcn_rcpf(ph0) * (float)C;
                w1[u] = g1 * __builtin_amdgcn_rcpf(ph1) * (float)C;
                if (t < T) {
                    sg0 += g0;
                    sg1 += g1;
                }
            } else {
                if (t < T) {
                    a.gamma_bm[(size_t)f * a.Tp + t] = g0;
                    if (a.nout > 1) a.gamma_bm[((size_t)a.F + f) * a.Tp + t] = g1;
                }
            }
        }
        PH(tm, tid == 0, 10);
        if constexpr (MODE == kModeFinal) return;
    }

    const int wave = tid >> 6, lane = tid & 63;
    float* row = sm.red[wave];
    const bool wr = lane >= 60;
    // ---- packed form: the entries (i <= j) in two row groups (as many registers as the plain
    //      phases hold); per group ONE pass over the frames for both parts and both classes: an
    //      entry's two class sums are a pair, x_i conj(x_j) gives real and imaginary part at once ----
    if constexpr (PK) {
        // rows [0, RS) hold at least half of the NP entries
        constexpr int RS = [] {
            int n = 0, r = 0;
            while (r < C && 2 * n < NP) n += C - r++;
            return r;
        }();
        auto group = [&](auto lo_c, auto hi_c, auto last_c) __attribute__((always_inline)) {
            constexpr int ILO = decltype(lo_c)::value, IHI = decltype(hi_c)::value;
            constexpr bool LAST = decltype(last_c)::value;
            // entries of rows [ILO, IHI): e in [ELO, ELO + NR), off-diagonal eo in [OLO, OLO + NI)
            constexpr int ELO = ILO * C - ILO * (ILO - 1) / 2, EHI = IHI * C - IHI * (IHI - 1) / 2;
            constexpr int NR = EHI - ELO, NIo = NR - (IHI - ILO);
            constexpr int OLO = ELO - ILO;
            constexpr int NVAL = 2 * NR + 2 * NIo + (LAST ? 2 : 0);
            if constexpr (NR > 0) {
                pk2 accR[NR], accI[NIo > 0 ? NIo : 1];
#pragma unroll
                for (int e = 0; e < NR; ++e) accR[e] = (pk2){0.f, 0.f};
#pragma unroll
                for (int e = 0; e < (NIo > 0 ? NIo : 1); ++e) accI[e] = (pk2){0.f, 0.f};
                static_for<U>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    if (tid + NT * u < T) {
                        pk2 x[C];
                        getxp(uc, x);
                        const pk2 w = (pk2){w0[u], w1[u]};
                        int e = 0, eo = 0;
#pragma unroll
                        for (int i = ILO; i < IHI; ++i)
#pragma unroll
                            for (int j = i; j < C; ++j) {
                                if (i == j) {
                                    const pk2 t2 = x[i] * x[i];
                                    pk2 pp;
                                    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(pp) : "v"(t2));
                                    accR[e] = __builtin_elementwise_fma(pp, w, accR[e]);
                                } else {
                                    const pk2 pq = pk_cmulc(x[i], x[j]);
                                    accR[e] = pk_fma_bc_lo(pq, w, accR[e]);
                                    accI[eo] = pk_fma_bc_hi(pq, w, accI[eo]);
                                    ++eo;
                                }
                                ++e;
                            }
                    }
                    reload_fence();
                });
                PH(tm, tid == 0, LAST ? 13 : 11);
                // [re class 0 | re class 1 | im class 0 | im class 1 | the two posterior sums]
                float v[NVAL];
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    v[e] = accR[e].x;
                    v[NR + e] = accR[e].y;
                }
#pragma unroll
                for (int e = 0; e < NIo; ++e) {
                    v[2 * NR + e] = accI[e].x;
                    v[2 * NR + NIo + e] = accI[e].y;
                }
                if constexpr (LAST) {
                    v[NVAL - 2] = sg0;
                    v[NVAL - 1] = sg1;
                }
                float tot[Bfly<NVAL>::N2];
                butterfly_sum<NVAL>(v, tot, lane);
                if (wr) {
#pragma unroll
                    for (int r = 0; r < Bfly<NVAL>::N2; ++r) {
                        const int i = butterfly_index<NVAL>(lane & 3, r);
                        if (i >= 0) {
                            int dst;
                            if (i < NR) dst = ELO + i;
                            else if (i < 2 * NR) dst = NV + ELO + (i - NR);
                            else if (i < 2 * NR + NIo) dst = NP + OLO + (i - 2 * NR);
                            else if (i < 2 * NR + 2 * NIo) dst = NV + NP + OLO + (i - 2 * NR - NIo);
                            else dst = (i == NVAL - 2) ? NV - 1 : 2 * NV - 1;
                            row[dst] = tot[r];
                        }
                    }
                }
                PH(tm, tid == 0, LAST ? 14 : 12);
            }
        };
        group(std::integral_constant<int, 0>{}, std::integral_constant<int, RS>{}, std::integral_constant<bool, RS == C>{});
        reload_fence();
        if constexpr (RS < C)
            group(std::integral_constant<int, RS>{}, std::integral_constant<int, C>{}, std::integral_constant<bool, true>{});
        return;
    }
    // ---- real parts: sum_t w_k Re(x_i conj x_j), i <= j (class k at k NP + e) ----
    {
        float acc[2 * NP];
#pragma unroll
        for (int e = 0; e < 2 * NP; ++e) acc[e] = 0.f;
        static_for<U>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (tid + NT * u < T) {
                cf x[C];
                getx(uc, x);
                int e = 0;
#pragma unroll
                for (int i = 0; i < C; ++i)
#pragma unroll
                    for (int j = i; j < C; ++j) {
                        const float pr = fmaf(x[i].x, x[j].x, x[i].y * x[j].y);
                        acc[e] = fmaf(w0[u], pr, acc[e]);
                        acc[NP + e] = fmaf(w1[u], pr, acc[NP + e]);
                        ++e;
                    }
            }
            reload_fence();
        });
        PH(tm, tid == 0, 11);
        float tot[Bfly<2 * NP>::N2];
        butterfly_sum<2 * NP>(acc, tot, lane);
        if (wr) {
#pragma unroll
            for (int r = 0; r < Bfly<2 * NP>::N2; ++r) {
                const int i = butterfly_index<2 * NP>(lane & 3, r);
                if (i >= 0) row[(i >= NP ? NV - NP : 0) + i] = tot[r];
            }
        }
    }
    reload_fence();
    PH(tm, tid == 0, 12);
    // ---- imaginary parts, i < j (class k at k NPO + e), and the posterior sums ----
    {
        constexpr int NI = 2 * NPO + 2;
        float acc[NI];
#pragma unroll
        for (int e = 0; e < NI; ++e) acc[e] = 0.f;
        static_for<U>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (tid + NT * u < T) {
                cf x[C];
                getx(uc, x);
                int e = 0;
#pragma unroll
                for (int i = 0; i < C; ++i)
#pragma unroll
                    for (int j = i + 1; j < C; ++j) {
                        const float pi = fmaf(x[i].y, x[j].x, -x[i].x * x[j].y);
                        acc[e] = fmaf(w0[u], pi, acc[e]);
                        acc[NPO + e] = fmaf(w1[u], pi, acc[NPO + e]);
                        ++e;
                    }
            }
            reload_fence();
        });
        PH(tm, tid == 0, 13);
        acc[2 * NPO] = sg0;
        acc[2 * NPO + 1] = sg1;
        float tot[Bfly<NI>::N2];
        butterfly_sum<NI>(acc, tot, lane);
        if (wr) {
#pragma unroll
            for (int r = 0; r < Bfly<NI>::N2; ++r) {
                const int i = butterfly_index<NI>(lane & 3, r);
                if (i >= 0) {
                    // i < NPO: class 0; i < 2 NPO: class 1; then the two posterior sums
                    const int dst = i < NPO ? NP + i
                                            : (i < 2 * NPO ? NV + NP + (i - NPO)
                                                           : (i == 2 * NPO ? NV - 1 : 2 * NV - 1));
                    row[dst] = tot[r];
                }
            }
        }
    }
    PH(tm, tid == 0, 14);
}

template <int C, int NT, int U, int RF, int WPS>
__global__ __launch_bounds__(NT, WPS) void cgmm_bin_em_kernel(const CgmmBinArgs* __restrict__ tbl,
                                                              int num_iters, int Tlp) {
    typedef BinSmem<C, NT> S;
    static_assert(RF <= U, "register frames are frames of the one chunk");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    S& sm = *reinterpret_cast<S*>(smem_raw);
    cf* Xs = reinterpret_cast<cf*>(smem_raw + ((sizeof(S) + 15) & ~(size_t)15));  // [C][Tlp]

    const CgmmBinArgs a = tbl[blockIdx.y];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int T = a.T, Tp = a.Tp;

    // ---- load the bin; flag bins that hold a near-silent frame ----
    const cf* xg = a.xb + (size_t)f * C * Tp;
    cf xr[RF > 0 ? RF : 1][C];
    float nmin = 1e30f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int t = tid + NT * u;
        float n2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const cf x = (t < T) ? xg[(size_t)c * Tp + t] : make_float2(0.f, 0.f);
            n2 = fmaf(x.x, x.x, fmaf(x.y, x.y, n2));
            if (u < RF) xr[u < RF ? u : 0][c] = x;
            else if (t < T) Xs[c * Tlp + tid + NT * (u - RF)] = x;
        }
        if (t < T) nmin = fminf(nmin, n2);
    }
    if (tid < 2) {
        sm.hasV[tid] = 0;
        sm.nfast[tid] = 0;
    }
    if (tid == 0) sm.tiny = 0;
    __syncthreads();
    if (nmin < 1.5f * kEpsF) sm.tiny = 1;
    __syncthreads();

    const int npass = num_iters + 2;
    for (int pass = 0; pass < npass; ++pass) {
        const int mode = pass == 0 ? (a.init_mask ? kModeInitMask : kModeInitId)
                                   : (pass == npass - 1 ? kModeFinal : kModeEm);
        // the register-resident frames are constant over the passes: without this the compiler
        // hoists their 21 + 15 outer products per frame out of the pass loop and keeps them in
        // registers for the whole kernel (RF = 4: 165 spilled registers, RF = 0: none)
#pragma unroll
        for (int u = 0; u < RF; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) asm volatile("" : "+v"(xr[u][c].x), "+v"(xr[u][c].y));
        const long long tc0 = a.timing ? (long long)__builtin_readcyclecounter() : 0;
        if (mode == kModeEm)
            frames_pass<C, NT, U, RF, kModeEm>(sm, tid, T, xr, Xs, Tlp, a, f);
        else if (mode == kModeFinal)
            frames_pass<C, NT, U, RF, kModeFinal>(sm, tid, T, xr, Xs, Tlp, a, f);
        else if (mode == kModeInitId)
            frames_pass<C, NT, U, RF, kModeInitId>(sm, tid, T, xr, Xs, Tlp, a, f);
        else
            frames_pass<C, NT, U, RF, kModeInitMask>(sm, tid, T, xr, Xs, Tlp, a, f);
        if (mode == kModeFinal) break;
        const long long tc2 = a.timing ? (long long)__builtin_readcyclecounter() : 0;
        __syncthreads();
        const long long tc3 = a.timing ? (long long)__builtin_readcyclecounter() : 0;
        if (wave < 2) {
            // the two solving waves are this workgroup's critical path while the CU's
            // other workgroup streams frames: let them issue first
            __builtin_amdgcn_s_setprio(3);
            solve_class<C, NT>(sm, wave, lane, mode, T, a.update_alpha,
                               a.timing ? a.timing + (size_t)f * kTimingSlots : nullptr);
            __builtin_amdgcn_s_setprio(0);
        }
        const long long tc4 = a.timing ? (long long)__builtin_readcyclecounter() : 0;
        __syncthreads();
        if (a.timing && tid == 0) {
            long long* tm = a.timing + (size_t)f * kTimingSlots;
            tm[0] += tc2 - tc0;   // frames + row sums
            tm[2] += tc3 - tc2;   // barrier wait (wave 0)
            tm[3] += tc4 - tc3;   // solve (wave 0 = class 0)
            tm[4] += (long long)__builtin_readcyclecounter() - tc4;
            tm[5] += 1;
            tm[6] = sm.nfast[0];
            tm[7] = sm.nfast[1];
        }
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

struct BinCfg {
    int nt, u, rf, wps;
};
// frames per thread U, of which RF in registers; WPS = waves per SIMD the register budget
// is cut for (4: 128 VGPRs, 3: 168)
constexpr BinCfg kCfgs[] = {
    {128, 5, 2, 4},   // T <=  640 (10 s): two waves per bin, eight workgroups per CU (500 x 6-ch x
                      // 10 s: 29.5 ms against 33.1 for 256 x 4; {128, 8, 3, 3} for <= 1024 frames
                      // loses to 256 x 4: 27.4 against 24.1 ms at 300 x 15 s)
    {256, 2, 1, 4},   // T <=  512
    {256, 4, 2, 4},   // T <= 1024
    {256, 8, 4, 3},   // T <= 2048, THREE workgroups per CU (6 ch x 30 s: 16.9 ms per 125
                      // utterances against 21.0 for the next line: half the wave sums per bin
                      // and two other workgroups to hide a solve behind)
    {512, 4, 1, 4},   // T <= 2048, two workgroups per CU
    {512, 8, 2, 4},   // T <= 4096
};
// (six-wave workgroups {384, 5, 2, 3} only get one workgroup per CU placed; four
// workgroups of 256 threads need RF = 5 at 128 VGPRs: 120 spilled registers)
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
constexpr size_t kLdsLimit = 160 * 1024;

// row pitch of the LDS tile: the frames of the longest utterance that are not in registers
inline int lds_frames(int max_frames, int nt, int rf) { return (std::max(max_frames - nt * rf, 0) + 1) & ~1; }

template <int C, int NT>
size_t bin_lds_bytes(int tlp) {
    return ((sizeof(BinSmem<C, NT>) + 15) & ~(size_t)15) + (size_t)C * tlp * sizeof(cf);
}

template <int C, int I>
hipError_t launch_cfg(const CgmmBinArgs* d_tbl, int n_utts, int F, int max_frames, int num_iters,
                      hipStream_t s) {
    constexpr BinCfg c = kCfgs[I];
    const int tlp = lds_frames(max_frames, c.nt, c.rf);
    const size_t lds = bin_lds_bytes<C, c.nt>(tlp);
    if (lds > kLdsLimit) return hipErrorInvalidValue;
    // 7 and 8 channels carry 2 x 49 / 2 x 64 + 2 accumulators per half pass and 98 / 128
    // SGPR-bound factor entries: they need 140 - 240 VGPRs (8 ch at the 128-VGPR budget:
    // 87 ms per 125 x 30 s against 61 ms for the streaming kernels, 45 ms without spills)
    constexpr int wps = C < 7 ? c.wps : ((c.nt <= 256 && c.u <= 5) ? 3 : 2);
    auto kern = cgmm_bin_em_kernel<C, c.nt, c.u, c.rf, wps>;
    // per launch, like wpe.hip and solve.hip: the attribute belongs to the CURRENT device's copy of
    // the kernel, so a process-wide "already set" flag would skip the second device of a process
    // (and two threads could race on it); the call itself is a table lookup
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(F, n_utts), dim3(c.nt), lds, s, d_tbl, num_iters, tlp);
    return hipGetLastError();
}

template <int C>
hipError_t launch_bin_c(const CgmmBinArgs* d_tbl, int n_utts, int F, int max_frames, int num_iters,
                        int cfg, hipStream_t s) {
    switch (cfg) {
        case 0: return launch_cfg<C, 0>(d_tbl, n_utts, F, max_frames, num_iters, s);
        case 1: return launch_cfg<C, 1>(d_tbl, n_utts, F, max_frames, num_iters, s);
        case 2: return launch_cfg<C, 2>(d_tbl, n_utts, F, max_frames, num_iters, s);
        case 3: return launch_cfg<C, 3>(d_tbl, n_utts, F, max_frames, num_iters, s);
        case 4: return launch_cfg<C, 4>(d_tbl, n_utts, F, max_frames, num_iters, s);
        case 5: return launch_cfg<C, 5>(d_tbl, n_utts, F, max_frames, num_iters, s);
    }
    return hipErrorInvalidValue;
}

template <int C>
bool cfg_fits(int cfg, int max_frames) {
    const BinCfg c = kCfgs[cfg];
    const int tlp = lds_frames(max_frames, c.nt, c.rf);
    return (c.nt == 128 ? bin_lds_bytes<C, 128>(tlp)
                        : (c.nt == 256 ? bin_lds_bytes<C, 256>(tlp) : bin_lds_bytes<C, 512>(tlp))) <=
           kLdsLimit;
}

bool cfg_fits_c(int C, int cfg, int max_frames) {
    switch (C) {
        case 1: return cfg_fits<1>(cfg, max_frames);
        case 2: return cfg_fits<2>(cfg, max_frames);
        case 3: return cfg_fits<3>(cfg, max_frames);
        case 4: return cfg_fits<4>(cfg, max_frames);
        case 5: return cfg_fits<5>(cfg, max_frames);
        case 6: return cfg_fits<6>(cfg, max_frames);
        case 7: return cfg_fits<7>(cfg, max_frames);
        case 8: return cfg_fits<8>(cfg, max_frames);
    }
    return false;
}

}  // namespace

size_t cgmm_bin_args_bytes() { return sizeof(CgmmBinArgs); }
int cgmm_bin_timing_slots() { return kTimingSlots; }

// frames pitch of the bin-major arrays
int cgmm_bin_pitch(int T) { return (T + 3) & ~3; }

// configuration (index into kCfgs) for a batch whose longest utterance has max_frames
// frames; -1 when the bin does not fit a CU (the caller then uses the streaming kernels of
// cgmm.hip).  SETK_CGMM_CFG=<index> forces one (A/B measurements).
int cgmm_bin_config(int C, int max_frames) {
    if (C < 1 || C > kMaxChannels) return -1;
    static const int forced = [] {
        const char* e = getenv("SETK_CGMM_CFG");
        return (e && *e) ? atoi(e) : -1;
    }();
    if (forced >= 0 && forced < kNumCfgs && kCfgs[forced].nt * kCfgs[forced].u >= max_frames &&
        cfg_fits_c(C, forced, max_frames))
        return forced;
    for (int i = 0; i < kNumCfgs; ++i)
        if (kCfgs[i].nt * kCfgs[i].u >= max_frames && cfg_fits_c(C, i, max_frames)) return i;
    return -1;
}

int cgmm_bin_threads(int C, int max_frames) {
    const int c = cgmm_bin_config(C, max_frames);
    return c < 0 ? 0 : kCfgs[c].nt;
}

void cgmm_bin_fill_args(void* out, const float* xb, const float* init_mask, float* gamma_bm, int T,
                        int F, int update_alpha, int nout, void* timing) {
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
    a.timing = static_cast<long long*>(timing);
    std::memcpy(out, &a, sizeof(a));
}

hipError_t launch_cgmm_bin(int C, const void* d_tbl, const float* const* d_spec_ptrs, int spec_pitch,
                           float* const* d_mask_ptrs, float* const* d_gamma_ptrs, int n_utts, int F,
                           int max_frames, int num_iters, int nout, hipStream_t s) {
    const CgmmBinArgs* t = static_cast<const CgmmBinArgs*>(d_tbl);
    const int cfg = cgmm_bin_config(C, max_frames);
    if (cfg < 0) return hipErrorInvalidValue;
    if (d_spec_ptrs) {  // null: the bin-major arrays were filled by stft_binmajor_kernel
        const int ntt = (max_frames + 31) / 32;
        dim3 g(ntt * C, (F + 31) / 32, n_utts);
        hipLaunchKernelGGL(spec_to_binmajor_kernel, g, dim3(256), 0, s,
                           reinterpret_cast<const cf* const*>(d_spec_ptrs), t, C, spec_pitch, ntt);
    }
    hipError_t e = hipErrorInvalidValue;
    switch (C) {
        case 1: e = launch_bin_c<1>(t, n_utts, F, max_frames, num_iters, cfg, s); break;
        case 2: e = launch_bin_c<2>(t, n_utts, F, max_frames, num_iters, cfg, s); break;
        case 3: e = launch_bin_c<3>(t, n_utts, F, max_frames, num_iters, cfg, s); break;
        case 4: e = launch_bin_c<4>(t, n_utts, F, max_frames, num_iters, cfg, s); break;
        case 5: e = launch_bin_c<5>(t, n_utts, F, max_frames, num_iters, cfg, s); break;
        case 6: e = launch_bin_c<6>(t, n_utts, F, max_frames, num_iters, cfg, s); break;
        case 7: e = launch_bin_c<7>(t, n_utts, F, max_frames, num_iters, cfg, s); break;
        case 8: e = launch_bin_c<8>(t, n_utts, F, max_frames, num_iters, cfg, s); break;
    }
    if (e != hipSuccess) return e;
    {
        dim3 g((max_frames + 31) / 32, ((F + 31) / 32) * nout, n_utts);
        hipLaunchKernelGGL(binmajor_to_tf_kernel, g, dim3(256), 0, s, t, d_mask_ptrs, d_gamma_ptrs);
    }
    return hipGetLastError();
}

}  // namespace setk
