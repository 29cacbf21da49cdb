// solve.hip -- batched per-bin weight solve (C x C complex, C <= 8), fp64.
//
// Replaces (funcwj/setk): solve_pevd (libs/beamformer.py:31-63: numpy eigh /
// scipy eigh(A, B) in a python loop over bins), MvdrBeamformer.weight
// (:527-539), GevdBeamformer.weight (:674-682), PmwfBeamformer.weight
// (:632-659) + rank1_constraint (:66-84), MpdrBeamformer.weight (:555-571) and
// do_ban (:14-28).
//
// One problem = one (utterance, bin).  8 lanes cooperate on a problem (lane j
// owns column j), 8 problems per wavefront, one wavefront per workgroup:
//   * principal eigenvectors: one-sided (Hestenes) Jacobi on the columns, the
//     7 rounds of a sweep pair column j with j ^ m and exchange them with DPP moves;
//   * Cholesky of the (noise) covariance cooperatively in LDS (row per lane),
//     triangular solves per lane (each lane its own right-hand side);
//   * the generalised problem is reduced with the Cholesky factor
//     (C~ = L^-1 Rs L^-H), solved by the same Jacobi and back-substituted.
// The kernel is latency bound and negligible in bytes/flops next to the two
// streaming passes; fp64 keeps it at least as accurate as the reference's
// LAPACK c64/c128 calls.
#include "common.h"
#include "dpp.h"
#include <hip/hip_runtime.h>
#include "../../include/setk_hip.h"

namespace setk {

typedef double2 cd;

#define SD __device__ __forceinline__

SD cd zadd(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
SD cd zsub(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }
SD cd zmul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
SD cd zmulc(cd a, cd b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
// conj(a) * b
SD cd zcmul(cd a, cd b) { return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }
SD cd zscale(cd a, double s) { return make_double2(a.x * s, a.y * s); }
SD cd zdiv(cd a, cd b) {
    const double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}
SD double zabs2(cd a) { return a.x * a.x + a.y * a.y; }
// lanes per problem: 4 for C <= 4 (sixteen problems per wavefront: with 8 lanes half of them
// idled through every step of a 4-channel solve -- configs[1]: 500 utterances x 257 bins),
// 8 for C <= 8, 16 for 8 < C <= 16
template <int C> struct Grp { static constexpr int W = (C > 8) ? 16 : (C > 4) ? 8 : 4; };
template <int W>
SD cd zshfl(cd v, int src) { return make_double2(__shfl(v.x, src, W), __shfl(v.y, src, W)); }

constexpr int kKindPevd = 100;  // internal: plain solve_pevd(Rs[, Rn])
constexpr double kEpsF32 = 1.1920928955078125e-07;

// one rotation of column j against column j ^ M; returns true if it rotated
// floor2: de Rijk's threshold as in jacobi_round_f32 -- inner products at the rounding level of
// the LARGEST column (eps64 |g_max|^2) are noise; on a rank-deficient matrix (a real recording
// of a few coherent sources on many microphones) the columns of the null space never meet the
// relative bound and the sweeps would run to their limit (SETK_NUM_NOCONV on 100+ bins of a
// 16-channel recording before this bound existed).
template <int C, int M>
SD bool jacobi_round(cd (&g)[C], int j, double floor2) {
    const double tol2 = 1e-16;  // see jacobi_pevd
    const int p = j ^ M;
    cd gp[C];
    double m = 0.0, o = 0.0;
    cd d = make_double2(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        gp[i] = make_double2(dshfl_xor<M>(g[i].x), dshfl_xor<M>(g[i].y));
        m += zabs2(g[i]);
        o += zabs2(gp[i]);
        d = zadd(d, zcmul(g[i], gp[i]));
    }
    const double dd = zabs2(d);
    if (dd > tol2 * m * o && dd > floor2) {
        const double absd = sqrt(dd);
        const double sigma = (j < p) ? 1.0 : -1.0;
        const double zeta = sigma * (o - m) / (2.0 * absd);
        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + t * t);
        const double sn = cs * t;
        const double f = sigma * sn / absd;
        const cd ph = make_double2(d.x * f, -d.y * f);  // sigma*sn*conj(d)/|d|
#pragma unroll
        for (int i = 0; i < C; ++i) g[i] = zsub(zscale(g[i], cs), zmul(ph, gp[i]));
        return true;
    }
    return false;
}

// One-sided Jacobi on the columns of a Hermitian PSD matrix: lane j passes
// column j in g.  Rotating column pairs until all are mutually orthogonal turns
// G = A into A V = V Lambda, so the column of largest norm IS lambda_max v_max:
// the principal eigenvector is that column normalised -- V itself is never
// accumulated (half the shuffles and flops of the textbook form).  Returns the
// vector replicated in `out` (unit 2-norm), its eigenvalue, and raises `noconv`
// on sweep exhaustion.  Stop: |g_p^H g_q| <= 1e-8 |g_p||g_q| -- the outputs are
// float32 and the sweeps converge quadratically, so a tighter bound only adds a
// last, idle sweep (1e-20 measured 0.181 ms for the reduce+solve stage, 1e-16
// 0.174, 1e-14 0.170).
template <int C>
SD void jacobi_pevd(cd (&g)[C], int j, cd (&out)[C], double& lam, int& noconv) {
    constexpr int W = Grp<C>::W;
    bool done = false;
    for (int sweep = 0; sweep < 40 && !done; ++sweep) {
        double mm = 0.0;
#pragma unroll
        for (int i = 0; i < C; ++i) mm += zabs2(g[i]);
#pragma unroll
        for (int sft = 1; sft < W; sft <<= 1) mm = fmax(mm, __shfl_xor(mm, sft, W));
        const double floor2 = 1e-28 * mm * mm;  // (~(64 eps64 |g_max|^2)^2)
        bool rot = false;
        rot |= jacobi_round<C, 1>(g, j, floor2);
        rot |= jacobi_round<C, 2>(g, j, floor2);
        rot |= jacobi_round<C, 3>(g, j, floor2);
        if constexpr (W >= 8) {
            rot |= jacobi_round<C, 4>(g, j, floor2);
            rot |= jacobi_round<C, 5>(g, j, floor2);
            rot |= jacobi_round<C, 6>(g, j, floor2);
            rot |= jacobi_round<C, 7>(g, j, floor2);
        }
        if constexpr (W == 16) {
            rot |= jacobi_round<C, 8>(g, j, floor2);
            rot |= jacobi_round<C, 9>(g, j, floor2);
            rot |= jacobi_round<C, 10>(g, j, floor2);
            rot |= jacobi_round<C, 11>(g, j, floor2);
            rot |= jacobi_round<C, 12>(g, j, floor2);
            rot |= jacobi_round<C, 13>(g, j, floor2);
            rot |= jacobi_round<C, 14>(g, j, floor2);
            rot |= jacobi_round<C, 15>(g, j, floor2);
        }
        done = !__any(rot);
    }
    if (!done) noconv = 1;
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < C; ++i) m += zabs2(g[i]);
    // argmax over the lanes of the group (lowest lane wins ties)
    double best = m;
    int bj = j;
#pragma unroll
    for (int s = 1; s < W; s <<= 1) {
        const double ob = __shfl_xor(best, s, W);
        const int oj = __shfl_xor(bj, s, W);
        if (ob > best || (ob == best && oj < bj)) {
            best = ob;
            bj = oj;
        }
    }
    lam = sqrt(best);
    const double inv = (lam > 0.0) ? 1.0 / lam : 0.0;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        out[i] = zscale(zshfl<W>(g[i], bj), inv);
        if (!(lam > 0.0)) out[i] = make_double2((i == 0) ? 1.0 : 0.0, 0.0);
    }
}

// y = M x for a Hermitian M held one column per lane (col[i] = M[i][j]):
// lane j forms (M x)_j = sum_m conj(col[m]) x[m], the results are all-gathered.
template <int C>
SD void herm_matvec(const cd (&col)[C], const cd (&x)[C], cd (&y)[C]) {
    cd mine = make_double2(0.0, 0.0);
#pragma unroll
    for (int m = 0; m < C; ++m) mine = zadd(mine, zcmul(col[m], x[m]));
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = zshfl<Grp<C>::W>(mine, i);
}

// ---- the same sweeps in float32, polished in float64 --------------------------------------
// What the reference computes with LAPACK's SINGLE precision cheevd (solve_pevd on complex64,
// libs/beamformer.py:40-46) does not need 13 000 fp64 instructions per wavefront: the sweeps
// run on a float32 copy of the matrix (scaled to max diag = 1), whose rotations are half the
// instructions (the partner column arrives through the DPP operand of the fused multiply-adds
// themselves, no separate moves of 64-bit halves) at twice the issue rate; the principal
// column they find is then polished against the float64 matrix by two power steps (each
// multiplies the error components by lambda_i / lambda_1 <= 1: it can only help) and its
// eigenvalue is the float64 Rayleigh norm.  One-sided Jacobi is backward stable in its working
// precision, so before the polish the vector is what LAPACK's float32 solver would give; after
// it, bins with a clear gap are float64-accurate.
template <int M>
SD float fxor(float x) {
    return __builtin_bit_cast(float, dpp_xor<M>(__builtin_bit_cast(int, x)));
}
// floor2: columns far below the principal one carry its rounding noise (eps32 |g_max| per
// entry), so their inner products with it never fall under the RELATIVE bound; below
// ~eps32 |g_max|^2 an inner product is noise and rotating on it would go on forever (de Rijk's
// threshold).  What is left un-annihilated there moves the principal vector by ~1e-6 at most,
// and the float64 power steps shrink exactly those components by lambda_small / lambda_max.
template <int C, int M>
SD bool jacobi_round_f32(float2 (&g)[C], int j, float floor2) {
    const float tol2 = 1e-11f;  // |g_p^H g_q|^2 <= tol2 |g_p|^2 |g_q|^2: the float32 noise floor is ~1e-13
    const int p = j ^ M;
    float2 gp[C];
    float m = 0.f;
    float2 d = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        gp[i] = make_float2(fxor<M>(g[i].x), fxor<M>(g[i].y));
        m = fmaf(g[i].x, g[i].x, fmaf(g[i].y, g[i].y, m));
        d.x = fmaf(g[i].x, gp[i].x, fmaf(g[i].y, gp[i].y, d.x));   // conj(g) * gp
        d.y = fmaf(g[i].x, gp[i].y, fmaf(-g[i].y, gp[i].x, d.y));
    }
    const float o = fxor<M>(m);
    const float dd = fmaf(d.x, d.x, d.y * d.y);
    if (dd > tol2 * m * o && dd > floor2) {
        const float rabs = __builtin_amdgcn_rsqf(dd);
        const float sigma = (j < p) ? 1.f : -1.f;
        const float zeta = sigma * (o - m) * 0.5f * rabs;
        const float t = copysignf(1.f, zeta) / (fabsf(zeta) + sqrtf(fmaf(zeta, zeta, 1.f)));
        const float cs = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f));
        const float f = sigma * cs * t * rabs;
        const float2 ph = make_float2(d.x * f, -d.y * f);  // sigma * sn * conj(d) / |d|
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const float2 t0 = g[i];
            g[i].x = fmaf(t0.x, cs, fmaf(-ph.x, gp[i].x, ph.y * gp[i].y));
            g[i].y = fmaf(t0.y, cs, fmaf(-ph.x, gp[i].y, -ph.y * gp[i].x));
        }
        return true;
    }
    return false;
}

#ifndef SETK_SOLVE_FP64_ONLY
#define SETK_SOLVE_FP64_ONLY 0
#endif
// a[i] = A[i][j] (lane j owns column j of the Hermitian PSD A, 8 lanes per problem)
template <int C>
SD void pevd_mixed(const cd (&a)[C], int j, cd (&out)[C], double& lam, int& noconv) {
    constexpr int W = Grp<C>::W;
    if constexpr (W > 8 || SETK_SOLVE_FP64_ONLY) {
        cd g[C];
#pragma unroll
        for (int i = 0; i < C; ++i) g[i] = a[i];
        jacobi_pevd<C>(g, j, out, lam, noconv);
        return;
    } else {
        double dg = 0.0;
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (i == j) dg = a[i].x;
#pragma unroll
        for (int s = 1; s < W; s <<= 1) dg = fmax(dg, __shfl_xor(dg, s, W));
        const double rs = (dg > 0.0) ? 1.0 / dg : 0.0;
        float2 g[C];
#pragma unroll
        for (int i = 0; i < C; ++i) g[i] = make_float2((float)(a[i].x * rs), (float)(a[i].y * rs));
        bool done = false;
        for (int sweep = 0; sweep < 40 && !done; ++sweep) {
            float mm = 0.f;
#pragma unroll
            for (int i = 0; i < C; ++i) mm = fmaf(g[i].x, g[i].x, fmaf(g[i].y, g[i].y, mm));
            mm = fmaxf(mm, fxor<1>(mm));
            mm = fmaxf(mm, fxor<2>(mm));
            if constexpr (W == 8) mm = fmaxf(mm, fxor<4>(mm));
            const float floor2 = 1e-12f * mm * mm;  // (~(8 eps32 |g_max|^2)^2)
            bool rot = false;
            rot |= jacobi_round_f32<C, 1>(g, j, floor2);
            rot |= jacobi_round_f32<C, 2>(g, j, floor2);
            rot |= jacobi_round_f32<C, 3>(g, j, floor2);
            if constexpr (W == 8) {
                rot |= jacobi_round_f32<C, 4>(g, j, floor2);
                rot |= jacobi_round_f32<C, 5>(g, j, floor2);
                rot |= jacobi_round_f32<C, 6>(g, j, floor2);
                rot |= jacobi_round_f32<C, 7>(g, j, floor2);
            }
            done = !__any(rot);
        }
        // (no float64 fallback in this kernel: its working set alone costs a wave per SIMD; sweep
        //  exhaustion is reported as SETK_NUM_NOCONV like jacobi_pevd's, and never seen in tests)
        if (!done) noconv = 1;
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < C; ++i) m = fmaf(g[i].x, g[i].x, fmaf(g[i].y, g[i].y, m));
        float best = m;
        int bj = j;
#pragma unroll
        for (int s = 1; s < W; s <<= 1) {
            const float ob = __shfl_xor(best, s, W);
            const int oj = __shfl_xor(bj, s, W);
            if (ob > best || (ob == best && oj < bj)) {
                best = ob;
                bj = oj;
            }
        }
        cd v[C];
        double nn = 0.0;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            v[i] = make_double2((double)__shfl(g[i].x, bj, W), (double)__shfl(g[i].y, bj, W));
            nn += zabs2(v[i]);
        }
        double inv = (nn > 0.0) ? 1.0 / sqrt(nn) : 0.0;
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = zscale(v[i], inv);
        // two power steps against the float64 matrix; the last norm is the eigenvalue
        lam = 0.0;
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            cd y[C];
            herm_matvec<C>(a, v, y);
            nn = 0.0;
#pragma unroll
            for (int i = 0; i < C; ++i) nn += zabs2(y[i]);
            lam = sqrt(nn);
            inv = (lam > 0.0) ? 1.0 / lam : 0.0;
#pragma unroll
            for (int i = 0; i < C; ++i) v[i] = zscale(y[i], inv);
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            out[i] = v[i];
            if (!(lam > 0.0)) out[i] = make_double2((i == 0) ? 1.0 : 0.0, 0.0);
        }
    }
}

template <int C>
SD void fix_gauge(cd (&v)[C]) {
    const double a = sqrt(zabs2(v[0]));
    if (a > 0.0) {
        const cd ph = make_double2(v[0].x / a, -v[0].y / a);
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = zmul(v[i], ph);
        v[0].y = 0.0;
    }
}

// Cooperative Cholesky M = L L^H in LDS (column-major C x C, lane j = row j).
// The reference solves these systems with LAPACK's pivoted LU, which fails only
// on an exactly zero pivot: a numerically semi-definite covariance (noise mask
// non-zero in fewer than C frames) is factored with rounding noise as pivots
// and the utterance is NOT skipped.  To keep that behaviour, pivots are floored
// at eps_f32 * max diag(M) (the noise level of the float32-accumulated input);
// only an all-zero / non-finite / negative-diagonal matrix reports 1.
template <int C>
SD int chol_lds(const cd (&col)[C], cd* L, double* piv, int j, bool zero_is_identity = false) {
    // col[i] = M[i][j] (lane j owns column j of the Hermitian M), so the row
    // element M[j][k] is conj(col[k])
    double diag = 0.0;
#pragma unroll
    for (int i = 0; i < C; ++i)
        if (i == j) diag = col[i].x;
    double scale = diag;
#pragma unroll
    for (int s = 1; s < Grp<C>::W; s <<= 1) scale = fmax(scale, __shfl_xor(scale, s, Grp<C>::W));
    if (zero_is_identity && scale == 0.0) {
        // SETK_FLAG_STRICT_REFERENCE, GEV: the reference goes through on an all-zero Rn
        // (hegvd refuses, scipy.linalg.eig takes over: libs/beamformer.py:54-59); the pencil
        // (Rs, I) stands in for what QZ makes of (Rs, 0)
#pragma unroll
        for (int k = 0; k < C; ++k)
            if (j >= k && j < C) L[k * C + j] = make_double2(j == k ? 1.0 : 0.0, 0.0);
        __syncthreads();
        return 0;
    }
    const int bad0 = !(scale > 0.0);
    int bad = bad0;
    const double floor_piv = kEpsF32 * scale;
    // Flooring a pivot in the middle of an unpivoted factorisation is only safe when little
    // follows it: on a strongly rank-deficient matrix (a REAL 16-channel recording of two or
    // three coherent sources: 10+ noise-level pivots in a row) what is left of the later
    // columns is divided by sqrt(floor) again and again, and |L| grows until it overflows
    // (doc/ssl/asset/egs.wav: NaN in 33 bins, growth 1e10 already on its first 8 channels).
    // So a problem that meets the floor is factored again with the floor ADDED to its diagonal
    // up front (8 eps_f32 max diag: above the negative rounding eigenvalues of a float32
    // covariance) -- then the matrix IS positive definite, Cholesky needs no pivoting and |L|
    // stays below sqrt(max diag).  Problems that never meet the floor are untouched.
    double load = 0.0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool hit = false;
        bad = bad0;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            cd s = make_double2(0.0, 0.0);
            if (j >= k && j < C) {
                s = make_double2(col[k].x, -col[k].y);  // M[j][k]
                if (j == k) s.x += load;
                for (int m = 0; m < k; ++m) s = zsub(s, zmulc(L[m * C + j], L[m * C + k]));
            }
            if (j == k) *piv = s.x;
            __syncthreads();
            double d = *piv;
            if (!(d == d)) bad = 1;  // NaN
            if (!(d >= floor_piv)) hit = true;
            d = fmax(d, floor_piv);
            const double rd = (d > 0.0) ? 1.0 / sqrt(d) : 0.0;
            if (j >= k && j < C) L[k * C + j] = (j == k) ? make_double2(d * rd, 0.0) : zscale(s, rd);
            __syncthreads();
        }
        if (attempt == 1 || !__any(hit && !bad0)) break;  // (wave-uniform: the barriers above need everyone)
        load = (hit && !bad0) ? 8.0 * floor_piv : 0.0;
    }
    return bad;
}

// L y = b (in place on b)
template <int C>
SD void fwd_solve(const cd* L, cd (&b)[C]) {
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const double r = 1.0 / L[k * C + k].x;
        b[k] = zscale(b[k], r);
#pragma unroll
        for (int i = k + 1; i < C; ++i) b[i] = zsub(b[i], zmul(L[k * C + i], b[k]));
        // keep the LDS loads of later columns from being hoisted up here: all 36
        // entries of L in flight at once cost 144 VGPRs
        __builtin_amdgcn_sched_barrier(0);
    }
}
// L^H x = y (in place on y)
template <int C>
SD void bwd_solve(const cd* L, cd (&y)[C]) {
#pragma unroll
    for (int k = C - 1; k >= 0; --k) {
        cd s = y[k];
#pragma unroll
        for (int i = k + 1; i < C; ++i) s = zsub(s, zcmul(L[k * C + i], y[i]));
        y[k] = zscale(s, 1.0 / L[k * C + k].x);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int W>
SD double group_sum(double v) {
#pragma unroll
    for (int s = 1; s < W; s <<= 1) v += __shfl_xor(v, s, W);
    return v;
}

// generalised principal eigenvector of (Rs, Rn) with L = chol(Rn) already in
// LDS: returns v = L^-H y, ||y|| = 1 (so v^H Rn v = 1), gauge on y.
template <int C>
SD void gev_vector(const cd (&rs)[C], const cd* L, cd* Wk, int j, bool gauge, cd (&vout)[C],
                   int& noconv) {
    cd x[C];
#pragma unroll
    for (int i = 0; i < C; ++i) x[i] = rs[i];
    fwd_solve<C>(L, x);  // column j of X = L^-1 Rs
    __syncthreads();
    if (j < C) {
#pragma unroll
        for (int i = 0; i < C; ++i) Wk[j * C + i] = x[i];
    }
    __syncthreads();
    cd c[C];
#pragma unroll
    for (int m = 0; m < C; ++m) {
        const cd t = (j < C) ? Wk[m * C + j] : make_double2(0.0, 0.0);  // X[j][m]
        c[m] = make_double2(t.x, -t.y);
    }
    fwd_solve<C>(L, c);  // column j of C~ = L^-1 X^H
    if (j >= C) {
#pragma unroll
        for (int i = 0; i < C; ++i) c[i] = make_double2(0.0, 0.0);
    }
    cd y[C];
    double lam;
    pevd_mixed<C>(c, j, y, lam, noconv);  // (C~ is Hermitian PSD like Rs: same mixed-precision solver)
    if (gauge) fix_gauge<C>(y);
    bwd_solve<C>(L, y);
#pragma unroll
    for (int i = 0; i < C; ++i) vout[i] = y[i];
}

// KIND is a template parameter: every beamformer gets its own register
// allocation (a single runtime-switched kernel needed 354 VGPRs because the
// allocator sees the union of all branches), and the matrices are fetched when
// a step needs them instead of up front: 0.225 -> 0.178 ms for the reduce+solve
// stage of the 125 x 257-bin MVDR batch.  Forcing 3 or 4 waves per SIMD
// (168 / 128 VGPRs, spilling the eigenvector across the Cholesky) measured the
// same time, so the budget stays at 2.
#ifndef SETK_SOLVE_WAVES
#define SETK_SOLVE_WAVES 2
#endif
template <int C, int KIND>
__global__ __launch_bounds__(64, (C > 8) ? 1 : SETK_SOLVE_WAVES) void solve_kernel(SolveArgs a, int pitch, int lds_mats) {
    constexpr int NP = npairs(C);
    constexpr int W = Grp<C>::W;   // lanes per problem
    constexpr int PW = 64 / W;     // problems per wavefront (= workgroup)
    // dynamic LDS, PW problems per workgroup: [L | Wk | Rs | Rn] (lds_mats of them).
    // L: Cholesky factor (all kinds but plain pevd); Wk: transposes of the reduced
    // pencil; Rs, Rn: full matrices, only for the PMWF reference-channel search.
    extern __shared__ __attribute__((aligned(16))) char solve_smem[];
    cd* mats = reinterpret_cast<cd*>(solve_smem);
    double* sPiv = reinterpret_cast<double*>(mats + (size_t)lds_mats * PW * C * C);

    const int tid = threadIdx.x;
    const int j = tid & (W - 1), q = tid / W;
    const int F = a.num_bins;
    const long n_prob = (long)a.n_utts * F;
    long prob = (long)blockIdx.x * PW + q;
    const bool live = prob < n_prob;
    if (!live) prob = n_prob - 1;  // keep the lanes in step; no stores
    const int u = (int)(prob / F), f = (int)(prob % F);

    cd* L = mats + ((size_t)0 * PW + q) * C * C;
    cd* Wk = mats + ((size_t)(lds_mats > 1 ? 1 : 0) * PW + q) * C * C;
    cd* Rsf = mats + ((size_t)(lds_mats > 2 ? 2 : 0) * PW + q) * C * C;
    cd* Rnf = mats + ((size_t)(lds_mats > 3 ? 3 : 0) * PW + q) * C * C;
    double* piv = &sPiv[q];

    constexpr int kind = KIND;
    const bool gauge = (a.flags & SETK_FLAG_NO_GAUGE) == 0;
    const bool have_rn = a.planes >= 4 * NP && kind != SETK_BF_MPDR;

    // ---- column j of a Hermitian matrix from the packed planes (pair `which`:
    // 0 speech, 1 noise, 2 observation) ----
    const float* base = a.covar + (size_t)u * a.planes * pitch + f;
    // fused reduction of pass 1's partial slabs (covar_finalize_kernel's sums and scales,
    // evaluated here: same order over the slabs, same float32 expressions)
    const bool fused = a.partials != nullptr;
    const int planes_in = 4 * NP + 2;
    const size_t slab = (size_t)planes_in * pitch;
    const float* P = nullptr;
    int nparts = 0;
    float scale_of[3] = {0.f, 0.f, 0.f};
    if (fused) {
        const UttDesc ud = a.utts[u];
        P = a.partials + (size_t)ud.part0 * slab + f;
        nparts = ud.nparts;
        float den0 = 0.f, den1 = 0.f;
        for (int p = 0; p < nparts; ++p) {
            den0 += P[p * slab + (size_t)(4 * NP + 0) * pitch];
            den1 += P[p * slab + (size_t)(4 * NP + 1) * pitch];
        }
        scale_of[0] = a.num_scale / fmaxf(den0, 1e-6f);
        scale_of[1] = a.num_scale / fmaxf(den1, 1e-6f);
        scale_of[2] = a.num_scale / fmaxf((float)ud.num_frames, 1e-6f);
    }
    bool finite = true;
    auto load_col = [&](int which, cd (&m)[C]) {
#pragma unroll
        for (int i = 0; i < C; ++i) {
            m[i] = make_double2(0.0, 0.0);
            if (j < C) {
                const int lo = i < j ? i : j, hi = i < j ? j : i;
                const int e = pair_index(lo, hi, C);
                const double sgn = (i <= j) ? 1.0 : -1.0;  // (i,j) stored for i<=j
                float re, im;
                if (!fused) {
                    re = base[(size_t)((2 * which + 0) * NP + e) * pitch];
                    im = base[(size_t)((2 * which + 1) * NP + e) * pitch];
                } else if (which < 2) {
                    float ar = 0.f, ai = 0.f;
                    for (int p = 0; p < nparts; ++p) {
                        ar += P[p * slab + (size_t)((2 * which + 0) * NP + e) * pitch];
                        ai += P[p * slab + (size_t)((2 * which + 1) * NP + e) * pitch];
                    }
                    re = ar * scale_of[which];
                    im = ai * scale_of[which];
                } else {  // Ry: the speech and noise numerators of every slab, over the frame count
                    float ar = 0.f, ai = 0.f;
                    for (int p = 0; p < nparts; ++p) {
                        ar += P[p * slab + (size_t)(0 * NP + e) * pitch] + P[p * slab + (size_t)(2 * NP + e) * pitch];
                        ai += P[p * slab + (size_t)(1 * NP + e) * pitch] + P[p * slab + (size_t)(3 * NP + e) * pitch];
                    }
                    re = ar * scale_of[2];
                    im = ai * scale_of[2];
                }
                m[i] = make_double2(re, (i == j) ? 0.0 : sgn * im);
                finite = finite && isfinite(re) && isfinite(im);
            }
        }
    };

    int st_sing = 0, st_noconv = 0;
    cd w[C];
#pragma unroll
    for (int i = 0; i < C; ++i) w[i] = make_double2(0.0, 0.0);

    if (kind == kKindPevd && !have_rn) {
        cd g[C];
        load_col(0, g);
        double lam;
        pevd_mixed<C>(g, j, w, lam, st_noconv);
        if (gauge) fix_gauge<C>(w);
    } else if (kind == kKindPevd || kind == SETK_BF_GEVD) {
        {
            cd rn[C];
            load_col(1, rn);
            st_sing |= chol_lds<C>(rn, L, piv, j,
                                   kind == SETK_BF_GEVD && (a.flags & SETK_FLAG_STRICT_REFERENCE) != 0);
        }
        cd rs[C];
        load_col(0, rs);
        gev_vector<C>(rs, L, Wk, j, gauge, w, st_noconv);
    } else if (kind == SETK_BF_MVDR) {
        cd d[C];
        {
            cd g[C];
            load_col(0, g);
            double lam;
            pevd_mixed<C>(g, j, d, lam, st_noconv);
        }
        if (gauge) fix_gauge<C>(d);
        {
            cd rn[C];
            load_col(1, rn);
            st_sing |= chol_lds<C>(rn, L, piv, j);
        }
        cd num[C];
#pragma unroll
        for (int i = 0; i < C; ++i) num[i] = d[i];
        fwd_solve<C>(L, num);
        bwd_solve<C>(L, num);
        cd den = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < C; ++i) den = zadd(den, zcmul(d[i], num[i]));
#pragma unroll
        for (int i = 0; i < C; ++i) w[i] = zdiv(num[i], den);
    } else if (kind == SETK_BF_MPDR || kind == SETK_BF_MPDR_WHITEN) {
        cd sv[C];
        if (kind == SETK_BF_MPDR) {
            cd g[C];
            load_col(0, g);
            double lam;
            pevd_mixed<C>(g, j, sv, lam, st_noconv);
            if (gauge) fix_gauge<C>(sv);
        } else {
            cd rn[C];
            load_col(1, rn);
            st_sing |= chol_lds<C>(rn, L, piv, j);
            cd v[C];
            {
                cd rs[C];
                load_col(0, rs);
                gev_vector<C>(rs, L, Wk, j, gauge, v, st_noconv);
            }
            herm_matvec<C>(rn, v, sv);
            __syncthreads();
        }
        {
            cd ry[C];
            load_col(2, ry);
            st_sing |= chol_lds<C>(ry, L, piv, j);
        }
        cd num[C];
#pragma unroll
        for (int i = 0; i < C; ++i) num[i] = sv[i];
        fwd_solve<C>(L, num);
        bwd_solve<C>(L, num);
        cd den = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < C; ++i) den = zadd(den, zcmul(sv[i], num[i]));
#pragma unroll
        for (int i = 0; i < C; ++i) w[i] = zdiv(num[i], den);
    } else if (kind == SETK_BF_PMWF) {
        cd rs[C], rn[C];
        load_col(0, rs);
        load_col(1, rn);
        st_sing |= chol_lds<C>(rn, L, piv, j);
        if (a.rank1 != SETK_RANK1_NONE) {
            cd pv[C];
            if (a.rank1 == SETK_RANK1_EIG) {
                cd g[C];
#pragma unroll
                for (int i = 0; i < C; ++i) g[i] = rs[i];
                double lam;
                pevd_mixed<C>(g, j, pv, lam, st_noconv);
            } else {
                cd v[C];
                gev_vector<C>(rs, L, Wk, j, false, v, st_noconv);
                herm_matvec<C>(rn, v, pv);
            }
            double dj = 0.0, pn = 0.0;
#pragma unroll
            for (int i = 0; i < C; ++i) {
                if (i == j) dj = rs[i].x;
                pn += zabs2(pv[i]);
            }
            const double tr = group_sum<Grp<C>::W>(dj);
            const double sc = tr / fmax(pn, kEpsF32);
            cd pj = make_double2(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < C; ++i)
                if (i == j) pj = pv[i];
#pragma unroll
            for (int i = 0; i < C; ++i)
                rs[i] = (j < C) ? zscale(zmulc(pv[i], pj), sc) : make_double2(0.0, 0.0);
        }
        cd x[C];
#pragma unroll
        for (int i = 0; i < C; ++i) x[i] = rs[i];
        fwd_solve<C>(L, x);
        bwd_solve<C>(L, x);  // column j of Rn^-1 Rs
        cd diag = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (i == j) diag = x[i];
        cd den = make_double2(group_sum<Grp<C>::W>(diag.x) + (double)a.pmwf_beta, group_sum<Grp<C>::W>(diag.y));
#pragma unroll
        for (int i = 0; i < C; ++i) x[i] = zdiv(x[i], den);
        if (a.pmwf_ref >= 0) {
#pragma unroll
            for (int i = 0; i < C; ++i) w[i] = zshfl<Grp<C>::W>(x[i], a.pmwf_ref);
        } else {
            // estimated SNR of every candidate column (libs/beamformer.py:620-630):
            // needs the full (possibly rank-1 replaced) Rs and Rn in every lane
            __syncthreads();
            if (j < C) {
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    Rsf[j * C + i] = rs[i];
                    Rnf[j * C + i] = rn[i];
                }
            }
            __syncthreads();
            double ps = 0.0, pn = 0.0;
#pragma unroll
            for (int i = 0; i < C; ++i) {
                cd s1 = make_double2(0.0, 0.0), s2 = s1;
#pragma unroll
                for (int m = 0; m < C; ++m) {
                    s1 = zadd(s1, zmul(Rsf[m * C + i], x[m]));
                    s2 = zadd(s2, zmul(Rnf[m * C + i], x[m]));
                }
                ps += zcmul(x[i], s1).x;
                pn += zcmul(x[i], s2).x;
            }
            if (live && j < C) {
                // per-bin terms; pmwf_select_kernel sums them in bin order (an atomic
                // accumulation would make the argmax depend on the arrival order)
                a.snr_acc[((size_t)prob * C + j) * 2 + 0] = ps;
                a.snr_acc[((size_t)prob * C + j) * 2 + 1] = pn;
                float2* wm = reinterpret_cast<float2*>(a.wmat) + ((size_t)prob * C + j) * C;
#pragma unroll
                for (int i = 0; i < C; ++i) wm[i] = make_float2((float)x[i].x, (float)x[i].y);
            }
        }
    }

    // ---- blind analytic normalisation (do_ban) ----
    if ((a.flags & SETK_FLAG_BAN) && !(kind == SETK_BF_PMWF && a.pmwf_ref < 0)) {
        cd rn[C];  // zero when the call carries no noise covariance (as before)
#pragma unroll
        for (int i = 0; i < C; ++i) rn[i] = make_double2(0.0, 0.0);
        if (have_rn) load_col(1, rn);
        cd uj = make_double2(0.0, 0.0);  // (Rn w)_j = sum_m conj(Rn[m][j]) w[m]
        cd wj = make_double2(0.0, 0.0);
#pragma unroll
        for (int m = 0; m < C; ++m) uj = zadd(uj, zcmul(rn[m], w[m]));
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (i == j) wj = w[i];
        const double nom = group_sum<Grp<C>::W>(zabs2(uj));
        const double den = group_sum<Grp<C>::W>(zcmul(wj, uj).x);
        const double filt = sqrt(nom) / fmax(den, kEpsF32);
#pragma unroll
        for (int i = 0; i < C; ++i) w[i] = zscale(w[i], filt);
    }

    // ---- store ----
    int st = SETK_NUM_OK;
    if (st_noconv) st = SETK_NUM_NOCONV;
    if (st_sing) st = SETK_NUM_SINGULAR;
    bool wfin = true;
#pragma unroll
    for (int i = 0; i < C; ++i) wfin = wfin && isfinite(w[i].x) && isfinite(w[i].y);
    if (!__all(finite) || !wfin) st = (st == SETK_NUM_OK) ? SETK_NUM_NONFINITE : st;
    if (live) {
        if (j < C && !(kind == SETK_BF_PMWF && a.pmwf_ref < 0)) {
            cd wj = make_double2(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < C; ++i)
                if (i == j) wj = w[i];
            float2* dst = reinterpret_cast<float2*>(a.weight) + ((size_t)u * C + j) * pitch + f;
            *dst = make_float2((float)wj.x, (float)wj.y);
        }
        if (j == 0) {
            if (a.bin_status) a.bin_status[prob] = st;
            if (a.status && st) atomicMax(a.status + u, st);
        }
    }
}


// ---------------------------------------------------------------------------
// SETK_FLAG_STRICT_REFERENCE: where would numpy.linalg.solve have raised?
//
// The reference hands the noise (MVDR libs/beamformer.py:536, PMWF :646) or observation
// (MPDR :568) covariance to numpy.linalg.solve: LAPACK ?gesv, i.e. an LU with partial
// pivoting in the matrix's own precision (complex64: the covariances are complex64 einsums),
// which reports "singular" only when the pivot search finds an EXACTLY zero column
// (?getf2: `if A(jp, j) != 0` else info = j); numpy turns that into LinAlgError for the whole
// stack and the CLI skips the utterance (apply_adaptive_beamformer.py:170-172).  On rounded
// data that happens for structurally singular input -- a duplicated, silent or power-of-two
// scaled channel, an all-zero covariance -- where the elimination cancels exactly; a
// covariance that is merely rank deficient to rounding (few mask frames, a channel that is
// 0.3 x another, a real 16-channel recording of three sources) goes through on noise-level
// pivots, in the reference and here.
//
// One thread per (utterance, bin): the same elimination in float32 without fused
// multiply-adds (right-looking, rows swapped, pivot = first maximum of |re| + |im| as
// icamax), SETK_NUM_SINGULAR when a pivot column is exactly zero.  The weights are not
// touched: they stay the float64 Cholesky's.  Which bin cancels exactly is rounding luck
// (LAPACK builds differ in their operation order as well); what is reproduced, and tested
// against the unmodified reference (tests/golden/ref_skipset.json), is the decision per
// utterance.
// ---------------------------------------------------------------------------
// CM: the matrix the thread's (scratch) arrays are sized for -- 4, 8 or 16 >= num_channels
#pragma clang fp contract(off)
template <int CM>
__global__ __launch_bounds__(64) void lu_refusal_kernel(SolveArgs a, int pitch, int which) {
    const int C = a.num_channels, F = a.num_bins;
    const int NP = npairs(C);
    const long n_prob = (long)a.n_utts * F;
    const long prob = (long)blockIdx.x * 64 + threadIdx.x;
    if (prob >= n_prob) return;
    const int u = (int)(prob / F), f = (int)(prob % F);
    float ar[CM][CM], ai[CM][CM];
    const bool fused = a.partials != nullptr;
    const size_t slab = (size_t)(4 * NP + 2) * pitch;
    const float* base = a.covar + (size_t)u * a.planes * pitch + f;
    const float* P = nullptr;
    int nparts = 0;
    float scale = 0.f;
    if (fused) {
        const UttDesc ud = a.utts[u];
        P = a.partials + (size_t)ud.part0 * slab + f;
        nparts = ud.nparts;
        float den = 0.f;
        if (which < 2) {
            for (int p = 0; p < nparts; ++p) den += P[p * slab + (size_t)(4 * NP + which) * pitch];
        } else {
            den = (float)ud.num_frames;
        }
        scale = a.num_scale / fmaxf(den, 1e-6f);
    }
    for (int i = 0; i < C; ++i)
        for (int k = i; k < C; ++k) {
            const int e = pair_index(i, k, C);
            float re, im;
            if (!fused) {
                re = base[(size_t)((2 * which + 0) * NP + e) * pitch];
                im = base[(size_t)((2 * which + 1) * NP + e) * pitch];
            } else {
                // the sums of load_col (solve_kernel), same order, same float32 expressions
                float sr = 0.f, si = 0.f;
                for (int p = 0; p < nparts; ++p) {
                    if (which < 2) {
                        sr += P[p * slab + (size_t)((2 * which + 0) * NP + e) * pitch];
                        si += P[p * slab + (size_t)((2 * which + 1) * NP + e) * pitch];
                    } else {
                        sr += P[p * slab + (size_t)(0 * NP + e) * pitch] + P[p * slab + (size_t)(2 * NP + e) * pitch];
                        si += P[p * slab + (size_t)(1 * NP + e) * pitch] + P[p * slab + (size_t)(3 * NP + e) * pitch];
                    }
                }
                re = sr * scale;
                im = si * scale;
            }
            // An imaginary part below the rounding level of its real part is the residue of the
            // multiply-adds that built the sum (a.y b.x - a.x b.y of EQUAL operands is the rounding
            // error of one product, not zero).  The reference's einsum carries the same residue
            // in both triangles of its (not exactly Hermitian) matrix, where it cancels when a
            // row is eliminated with an equal one; here only the upper triangle exists and the
            // mirrored residue would have the opposite sign.  Dropping it keeps the rows of a
            // duplicated (or 2^k-scaled) channel equal, as the reference's are.
            if (i == k || fabsf(im) <= 8.f * 1.1920929e-07f * fabsf(re)) im = 0.f;
            ar[i][k] = re;
            ai[i][k] = im;
            ar[k][i] = re;
            ai[k][i] = -im;
        }
    bool singular = false;
    for (int k = 0; k < C && !singular; ++k) {
        int p = k;
        float best = fabsf(ar[k][k]) + fabsf(ai[k][k]);
        for (int i = k + 1; i < C; ++i) {
            const float v = fabsf(ar[i][k]) + fabsf(ai[i][k]);
            if (v > best) {
                best = v;
                p = i;
            }
        }
        if (best == 0.f) {
            singular = true;
            break;
        }
        if (p != k)
            for (int m = 0; m < C; ++m) {
                const float tr = ar[k][m], ti = ai[k][m];
                ar[k][m] = ar[p][m];
                ai[k][m] = ai[p][m];
                ar[p][m] = tr;
                ai[p][m] = ti;
            }
        const float pr = ar[k][k], pi = ai[k][k];
        const float d = pr * pr + pi * pi;
        for (int i = k + 1; i < C; ++i) {
            // l = a_ik / pivot as a conj(p) / |p|^2: a row that EQUALS the pivot row (or is a power
            // of two times it) gets l = 1 (2^k) exactly and cancels exactly, in every bin -- the
            // structural cases do not hang on rounding luck as they do with LAPACK's
            // multiply-by-reciprocal, which finds them in some bins of the 257 only
            const float lr = (ar[i][k] * pr + ai[i][k] * pi) / d;
            const float li = (ai[i][k] * pr - ar[i][k] * pi) / d;
            for (int m = k + 1; m < C; ++m) {
                ar[i][m] = ar[i][m] - (lr * ar[k][m] - li * ai[k][m]);
                ai[i][m] = ai[i][m] - (lr * ai[k][m] + li * ar[k][m]);
            }
        }
    }
    if (singular) {
        if (a.bin_status) atomicMax(a.bin_status + prob, SETK_NUM_SINGULAR);
        if (a.status) atomicMax(a.status + u, SETK_NUM_SINGULAR);
    }
}
#pragma clang fp contract(fast)

hipError_t launch_solve(const SolveArgs& a, hipStream_t s) {
    const long n_prob = (long)a.n_utts * a.num_bins;
    const int pw = a.num_channels > 8 ? 4 : (a.num_channels > 4 ? 8 : 16);  // problems per wavefront (64 / Grp<C>::W)
    const int blocks = (int)((n_prob + pw - 1) / pw);
    const int pitch = (a.num_bins == kBins) ? kBinsPad : ((a.num_bins + 7) / 8) * 8;
    // L only: MVDR, MPDR; + Wk: the reduced-pencil kinds; + Rs, Rn: PMWF SNR search
    int lds_mats = 2;
    if (a.kind == SETK_BF_MVDR || a.kind == SETK_BF_MPDR) lds_mats = 1;
    if (a.kind == SETK_BF_PMWF) lds_mats = 4;
    // 16 channels with all four matrices resident (PMWF) pass the 64 KB a kernel gets
    // without asking
#define SETK_LAUNCH(c, k)                                                              \
    do {                                                                               \
        const size_t lds = (size_t)lds_mats * pw * c * c * sizeof(cd) + pw * sizeof(double); \
        if (lds > (64u << 10)) {                                                       \
            hipError_t e = hipFuncSetAttribute(                                        \
                reinterpret_cast<const void*>(solve_kernel<c, k>),                     \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
            if (e != hipSuccess) return e;                                             \
        }                                                                              \
        hipLaunchKernelGGL((solve_kernel<c, k>), dim3(blocks), dim3(64), lds, s, a,    \
                           pitch, lds_mats);                                           \
    } while (0)
#define SETK_CASE(c)                                                                   \
    case c:                                                                            \
        switch (a.kind) {                                                              \
            case SETK_BF_MVDR: SETK_LAUNCH(c, SETK_BF_MVDR); break;                    \
            case SETK_BF_GEVD: SETK_LAUNCH(c, SETK_BF_GEVD); break;                    \
            case SETK_BF_PMWF: SETK_LAUNCH(c, SETK_BF_PMWF); break;                    \
            case SETK_BF_MPDR: SETK_LAUNCH(c, SETK_BF_MPDR); break;                    \
            case SETK_BF_MPDR_WHITEN: SETK_LAUNCH(c, SETK_BF_MPDR_WHITEN); break;      \
            case kKindPevd: SETK_LAUNCH(c, kKindPevd); break;                          \
            default: return hipErrorInvalidValue;                                      \
        }                                                                              \
        break;
    switch (a.num_channels) {
        SETK_CASE(1)
        SETK_CASE(2)
        SETK_CASE(3)
        SETK_CASE(4)
        SETK_CASE(5)
        SETK_CASE(6)
        SETK_CASE(7)
        SETK_CASE(8)
        SETK_CASE(16)  // 8 < C <= 16 arrive padded to 16 (capi.hip run_weights)
        default:
            return hipErrorInvalidValue;
    }
#undef SETK_CASE
#undef SETK_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if ((a.flags & SETK_FLAG_STRICT_REFERENCE) &&
        (a.kind == SETK_BF_MVDR || a.kind == SETK_BF_PMWF || a.kind == SETK_BF_MPDR ||
         a.kind == SETK_BF_MPDR_WHITEN)) {
        // the matrix the reference's numpy.linalg.solve factors: Rn (MVDR, PMWF), Ry (MPDR)
        const int which = (a.kind == SETK_BF_MVDR || a.kind == SETK_BF_PMWF) ? 1 : 2;
        const dim3 grid((unsigned)((n_prob + 63) / 64));
        if (a.num_channels <= 4) hipLaunchKernelGGL(lu_refusal_kernel<4>, grid, dim3(64), 0, s, a, pitch, which);
        else if (a.num_channels <= 8) hipLaunchKernelGGL(lu_refusal_kernel<8>, grid, dim3(64), 0, s, a, pitch, which);
        else hipLaunchKernelGGL(lu_refusal_kernel<kMaxChannels16>, grid, dim3(64), 0, s, a, pitch, which);
        e = hipGetLastError();
    }
    return e;
}

// ---------------------------------------------------------------------------
// PMWF with SNR-selected reference channel: pick argmax_c ps/max(eps, pn) per
// utterance (libs/beamformer.py:650-653), copy that column out, optional BAN.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pmwf_select_kernel(SolveArgs a, int pitch, int* ref_out) {
    const int u = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int C = a.num_channels, F = a.num_bins;
    const int NP = npairs(C);
    // sum_f of the per-bin (ps, pn) terms in a fixed order: thread c sums channel c
    __shared__ double s_snr[kMaxChannels16][2];
    if (threadIdx.x < C) {
        double ps = 0.0, pn = 0.0;
        for (int b = 0; b < F; ++b) {
            ps += a.snr_acc[(((size_t)u * F + b) * C + threadIdx.x) * 2 + 0];
            pn += a.snr_acc[(((size_t)u * F + b) * C + threadIdx.x) * 2 + 1];
        }
        s_snr[threadIdx.x][0] = ps;
        s_snr[threadIdx.x][1] = pn;
    }
    __syncthreads();
    int ref = 0;
    double best = -1e300;
    for (int c = 0; c < C; ++c) {
        const double ps = s_snr[c][0];
        const double pn = s_snr[c][1];
        const double r = ps / fmax(kEpsF32, pn);
        if (r > best) {
            best = r;
            ref = c;
        }
    }
    if (f == 0 && ref_out) ref_out[u] = ref;
    if (f >= F) return;
    const size_t prob = (size_t)u * F + f;
    const float2* wm = reinterpret_cast<const float2*>(a.wmat) + (prob * C + ref) * C;
    double2 w[kMaxChannels16];
    for (int i = 0; i < C; ++i) w[i] = make_double2(wm[i].x, wm[i].y);
    if (a.flags & SETK_FLAG_BAN) {
        const float* base = a.covar + (size_t)u * a.planes * pitch + f;
        double nom = 0.0, den = 0.0;
        for (int i = 0; i < C; ++i) {
            double2 ui = make_double2(0.0, 0.0);
            for (int m = 0; m < C; ++m) {
                const int lo = i < m ? i : m, hi = i < m ? m : i;
                const int e = pair_index(lo, hi, C);
                const double re = base[(size_t)(2 * NP + e) * pitch];
                double im = (i == m) ? 0.0 : base[(size_t)(3 * NP + e) * pitch];
                if (i > m) im = -im;  // Rn[i][m] = conj(Rn[m][i])
                ui.x += re * w[m].x - im * w[m].y;
                ui.y += re * w[m].y + im * w[m].x;
            }
            nom += ui.x * ui.x + ui.y * ui.y;
            den += w[i].x * ui.x + w[i].y * ui.y;
        }
        const double filt = sqrt(nom) / fmax(den, kEpsF32);
        for (int i = 0; i < C; ++i) {
            w[i].x *= filt;
            w[i].y *= filt;
        }
    }
    for (int i = 0; i < C; ++i) {
        float2* dst = reinterpret_cast<float2*>(a.weight) + ((size_t)u * C + i) * pitch + f;
        *dst = make_float2((float)w[i].x, (float)w[i].y);
    }
}

hipError_t launch_pmwf_select(const SolveArgs& a, int* ref_out, hipStream_t s) {
    const int pitch = (a.num_bins == kBins) ? kBinsPad : ((a.num_bins + 7) / 8) * 8;
    dim3 grid((a.num_bins + 255) / 256, a.n_utts);
    hipLaunchKernelGGL(pmwf_select_kernel, grid, dim3(256), 0, s, a, pitch, ref_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// layout helpers for the modular API
// ---------------------------------------------------------------------------
// covar[F][C][C] complex64 -> packed planes [2*NP][pitch] starting at plane0
// Cp >= C: the matrix is embedded in a Cp x Cp one, blkdiag(M, pad_diag * I): with
// pad_diag = 1 for the matrices that get factored (Rn, Ry) and 0 for Rs the padded
// problem has the original's solution in its first C components and zeros after.
__global__ void pack_covar_kernel(const float2* fcc, int F, int C, int Cp, float pad_diag,
                                  int pitch, float* planes, int plane0) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= pitch) return;
    const int NP = npairs(Cp);
    for (int i = 0; i < Cp; ++i)
        for (int j = i; j < Cp; ++j) {
            const int e = pair_index(i, j, Cp);
            float2 v = make_float2(0.f, 0.f);
            if (f < F) {
                // the element of the LOWER triangle, as LAPACK's UPLO = 'L' (numpy eigh,
                // scipy eigh(lower=True)) reads it: a float32 covariance is Hermitian
                // only to rounding and the two triangles give eigenvectors 1e-4 apart
                // on a pencil with a 1 % eigenvalue gap
                if (j < C) {
                    const float2 l = fcc[((size_t)f * C + j) * C + i];
                    v = make_float2(l.x, i == j ? 0.f : -l.y);
                } else if (i == j) {
                    v = make_float2(pad_diag, 0.f);
                }
            }
            planes[(size_t)(plane0 + e) * pitch + f] = v.x;
            planes[(size_t)(plane0 + NP + e) * pitch + f] = v.y;
        }
}

hipError_t launch_pack_covar(const float* fcc, int F, int C, int Cp, float pad_diag,
                             float* planes, int plane0, hipStream_t s) {
    const int pitch = (F == kBins) ? kBinsPad : ((F + 7) / 8) * 8;
    hipLaunchKernelGGL(pack_covar_kernel, dim3((pitch + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(fcc), F, C, Cp, pad_diag, pitch, planes,
                       plane0);
    return hipGetLastError();
}

// packed planes of a batch -> covar[u][F][C][C] complex64 (Hermitian completion),
// pair set starting at plane0 of each utterance's `planes` planes
__global__ void unpack_covar_kernel(const float* planes, int n_planes, int plane0, int F, int C,
                                    int pitch, float2* out) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int u = blockIdx.y;
    if (f >= F) return;
    const int NP = npairs(C);
    const float* base = planes + ((size_t)u * n_planes + plane0) * pitch + f;
    float2* o = out + ((size_t)u * F + f) * C * C;
    for (int i = 0; i < C; ++i)
        for (int j = i; j < C; ++j) {
            const int e = pair_index(i, j, C);
            const float re = base[(size_t)e * pitch], im = base[(size_t)(NP + e) * pitch];
            o[i * C + j] = make_float2(re, i == j ? 0.f : im);
            if (i != j) o[j * C + i] = make_float2(re, -im);
        }
}

hipError_t launch_unpack_covar(const float* planes, int n_utts, int n_planes, int plane0, int F,
                               int C, float* out, hipStream_t s) {
    const int pitch = (F == kBins) ? kBinsPad : ((F + 7) / 8) * 8;
    hipLaunchKernelGGL(unpack_covar_kernel, dim3((F + 255) / 256, n_utts), dim3(256), 0, s, planes,
                       n_planes, plane0, F, C, pitch, reinterpret_cast<float2*>(out));
    return hipGetLastError();
}

// weight planes of a batch [u][C][pitch] float2 -> w[u][F][C] complex64
__global__ void unpack_weight_batch_kernel(const float2* wp, int F, int C, int pitch, float2* w) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int u = blockIdx.y;
    if (f >= F) return;
    for (int c = 0; c < C; ++c)
        w[((size_t)u * F + f) * C + c] = wp[((size_t)u * C + c) * pitch + f];
}

hipError_t launch_unpack_weight_batch(const float* wplanes, int n_utts, int F, int C, float* w,
                                      hipStream_t s) {
    const int pitch = (F == kBins) ? kBinsPad : ((F + 7) / 8) * 8;
    hipLaunchKernelGGL(unpack_weight_batch_kernel, dim3((F + 255) / 256, n_utts), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(wplanes), F, C, pitch,
                       reinterpret_cast<float2*>(w));
    return hipGetLastError();
}

// weight planes [C][pitch] float2 -> w[F][C] complex64
__global__ void unpack_weight_kernel(const float2* wp, int F, int C, int pitch, float2* w_fc) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    for (int c = 0; c < C; ++c) w_fc[(size_t)f * C + c] = wp[(size_t)c * pitch + f];
}

hipError_t launch_unpack_weight(const float* wplanes, int F, int C, float* w_fc, hipStream_t s) {
    const int pitch = (F == kBins) ? kBinsPad : ((F + 7) / 8) * 8;
    hipLaunchKernelGGL(unpack_weight_kernel, dim3((F + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(wplanes), F, C, pitch,
                       reinterpret_cast<float2*>(w_fc));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// stand-alone do_ban (libs/beamformer.py:14-28) and the rank-1 rebuild of
// rank1_constraint (libs/beamformer.py:75-84) on [F][C][C] / [F][C] arrays;
// one thread per bin, fp64.
// ---------------------------------------------------------------------------
__global__ void ban_kernel(const float2* __restrict__ w, const float2* __restrict__ Rn, int F,
                           int C, float2* __restrict__ out) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    // nominator = w^H Rn Rn w = sum_i (w^H Rn)_i (Rn w)_i ; denominator = Re w^H Rn w
    double nom_re = 0.0, nom_im = 0.0, den = 0.0;
    for (int i = 0; i < C; ++i) {
        double ux = 0.0, uy = 0.0;  // (Rn w)_i
        double vx = 0.0, vy = 0.0;  // (w^H Rn)_i = sum_m conj(w_m) Rn[m][i]
        for (int m = 0; m < C; ++m) {
            const float2 r = Rn[((size_t)f * C + i) * C + m];
            const float2 rt = Rn[((size_t)f * C + m) * C + i];
            const float2 x = w[(size_t)f * C + m];
            ux += (double)r.x * x.x - (double)r.y * x.y;
            uy += (double)r.x * x.y + (double)r.y * x.x;
            vx += (double)x.x * rt.x + (double)x.y * rt.y;
            vy += (double)x.x * rt.y - (double)x.y * rt.x;
        }
        const float2 wi = w[(size_t)f * C + i];
        nom_re += vx * ux - vy * uy;
        nom_im += vx * uy + vy * ux;
        den += (double)wi.x * ux + (double)wi.y * uy;
    }
    const double filt = sqrt(sqrt(nom_re * nom_re + nom_im * nom_im)) / fmax(den, kEpsF32);
    for (int i = 0; i < C; ++i) {
        const float2 x = w[(size_t)f * C + i];
        out[(size_t)f * C + i] = make_float2((float)(x.x * filt), (float)(x.y * filt));
    }
}

hipError_t launch_ban(const float* w, const float* Rn, int F, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(ban_kernel, dim3((F + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(w), reinterpret_cast<const float2*>(Rn), F,
                       C, reinterpret_cast<float2*>(out));
    return hipGetLastError();
}

// pv: principal (generalised) eigenvectors [F][C]; if Rn != null pv <- Rn pv.
// out = tr(Rs) / max(tr(p p^H), eps) * p p^H
__global__ void rank1_kernel(const float2* __restrict__ pv, const float2* __restrict__ Rs,
                             const float2* __restrict__ Rn, int F, int C,
                             float2* __restrict__ out) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    double px[kMaxChannels16], py[kMaxChannels16];
    for (int i = 0; i < C; ++i) {
        if (Rn) {
            double ux = 0.0, uy = 0.0;
            for (int m = 0; m < C; ++m) {
                const float2 r = Rn[((size_t)f * C + i) * C + m];
                const float2 x = pv[(size_t)f * C + m];
                ux += (double)r.x * x.x - (double)r.y * x.y;
                uy += (double)r.x * x.y + (double)r.y * x.x;
            }
            px[i] = ux;
            py[i] = uy;
        } else {
            px[i] = pv[(size_t)f * C + i].x;
            py[i] = pv[(size_t)f * C + i].y;
        }
    }
    double trx = 0.0, try_ = 0.0, pn = 0.0;
    for (int i = 0; i < C; ++i) {
        trx += Rs[((size_t)f * C + i) * C + i].x;
        try_ += Rs[((size_t)f * C + i) * C + i].y;
        pn += px[i] * px[i] + py[i] * py[i];
    }
    const double d = fmax(pn, kEpsF32);
    const double sx = trx / d, sy = try_ / d;
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
            // p_i conj(p_j)
            const double rx = px[i] * px[j] + py[i] * py[j];
            const double ry = py[i] * px[j] - px[i] * py[j];
            out[((size_t)f * C + i) * C + j] =
                make_float2((float)(sx * rx - sy * ry), (float)(sx * ry + sy * rx));
        }
}

hipError_t launch_rank1(const float* pv, const float* Rs, const float* Rn, int F, int C,
                        float* out, hipStream_t s) {
    hipLaunchKernelGGL(rank1_kernel, dim3((F + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(pv), reinterpret_cast<const float2*>(Rs),
                       reinterpret_cast<const float2*>(Rn), F, C, reinterpret_cast<float2*>(out));
    return hipGetLastError();
}

}  // namespace setk
