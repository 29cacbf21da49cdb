// cgmm_k.hip -- CGMM mask estimation for ANY number of classes and up to 16 channels:
// the general (and unhurried) twin of the K = 2 kernels in cgmm_bin.hip / cgmm.hip.
//
// Replaces (funcwj/setk): CgmmTrainer.__init__/train (scripts/sptk/libs/cluster.py:396-465)
// with num_classes != 2 -- the start is the caller's gamma0, the reference's
// np.random.uniform(size=[K, F, T]) normalised over K, drawn from the legacy global generator
// that estimate_cgmm_masks.py:28 seeds with --seed -- and, for 9 - 16 channels, the K = 2
// starts as well (deterministic Rs = x x^H / T, Rn = I, or an initial mask); Cgmm.update /
// predict (:246-287), CgDistribution.update_parameters / log_pdf (:193-235), Covariance
// (:94-133: eigh, w / max(w_max, eps), floor eps, R^-1 = V diag(1 / w) V^H, log det = sum log w).
//
// One 256-thread workgroup owns one frequency bin of one utterance for the whole run (the bins
// of the model are independent) and does what the reference does, in float64 like the
// reference: per class the weighted outer products (thread = one Hermitian pair x one chunk
// of the frames), a serial cyclic Jacobi of the C x C covariance by one thread per class,
// the quadratic forms and posteriors (thread = frame).  gamma and phi of the previous
// iteration live in a global work area.  No attempt at speed: K > 2 and wide arrays are rare
// options (30 s x 8 ch x K = 3: ~25 ms); the K = 2, C <= 8 default keeps its tuned kernels.
#include <hip/hip_runtime.h>
#include "common.h"
#include "fft512.h"
#include "../../include/setk_hip.h"

namespace setk {

namespace {

constexpr int kKMax = 4;
constexpr int kCMax = 16;
constexpr double kEpsK = 1.1920928955078125e-07;  // np.finfo(np.float32).eps (libs/utils.py:16)

struct zc {
    double x, y;
};
__device__ __forceinline__ zc zmk(double a, double b) { return zc{a, b}; }

// cyclic Jacobi of the Hermitian A (n x n, row major, in place: diagonal = eigenvalues),
// V = eigenvectors in columns.  The rotation of (p, q) with a_pq = g e, |e| = 1:
//   U = [[c, s e], [-s conj(e), c]],  A <- U^H A U,  V <- V U
// (checked against numpy.linalg.eigh in the form of /tmp's numpy twin: eigenvalues 5e-15,
// residual 3e-15 at n = 16).
// Returns false when the sweep limit was reached with rotations still pending.
__device__ bool jacobi_herm(zc* A, zc* V, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = zmk(i == j ? 1.0 : 0.0, 0.0);
    // an off-diagonal entry below the rounding level of the matrix is left alone: a rotation
    // cannot make it smaller (its own arithmetic errs by ~eps trace), and a rank-deficient
    // covariance -- more channels than sources: a_pp a_qq ~ 0 -- would otherwise rotate through all
    // 60 sweeps at that level, in 220 of 257 bins of a 9-channel scene.  This is LAPACK's accuracy
    // class too (absolute in the norm of the matrix); eigenvalues down there are floored at
    // eps_float32 lambda_max afterwards (cluster.py:107-113)
    double tr = 0.0;
    for (int i = 0; i < n; ++i) tr += fabs(A[i * n + i].x);
    const double abs2 = 1.6e-31 * tr * tr;  // (4e-16 trace)^2
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rot = false;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const zc apq = A[p * n + q];
                const double g2 = apq.x * apq.x + apq.y * apq.y;
                const double app = A[p * n + p].x, aqq = A[q * n + q].x;
                if (!(g2 > 1e-34 * fabs(app * aqq)) || !(g2 > abs2) || g2 == 0.0) continue;
                rot = true;
                const double g = sqrt(g2);
                const double ex = apq.x / g, ey = apq.y / g;
                const double tau = (aqq - app) / (2.0 * g);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                const double sx = s * ex, sy = s * ey;  // s e
                for (int i = 0; i < n; ++i) {  // columns p, q of A and of V
                    for (int m = 0; m < 2; ++m) {
                        zc* M = m ? V : A;
                        const zc cp = M[i * n + p], cq = M[i * n + q];
                        // new p = c cp - s conj(e) cq ; new q = s e cp + c cq
                        M[i * n + p] = zmk(c * cp.x - (sx * cq.x + sy * cq.y), c * cp.y - (sx * cq.y - sy * cq.x));
                        M[i * n + q] = zmk((sx * cp.x - sy * cp.y) + c * cq.x, (sx * cp.y + sy * cp.x) + c * cq.y);
                    }
                }
                for (int j = 0; j < n; ++j) {  // rows p, q of A
                    const zc rp = A[p * n + j], rq = A[q * n + j];
                    // new p = c rp - s e rq ; new q = s conj(e) rp + c rq
                    A[p * n + j] = zmk(c * rp.x - (sx * rq.x - sy * rq.y), c * rp.y - (sx * rq.y + sy * rq.x));
                    A[q * n + j] = zmk((sx * rp.x + sy * rp.y) + c * rq.x, (sx * rp.y - sy * rp.x) + c * rq.y);
                }
            }
        if (!rot) return true;
    }
    return false;
}

struct CgmmKArgs {
    const cf* spec;           // [C][T][F]
    const double* gamma0;     // [K][F][T] (K > 2, or any K with a full start) or null
    const float* init_mask;   // [T][F] (K = 2) or null
    float* gamma_out;         // [K][T][F]
    double* work;             // [F][2][K][T]: gamma | phi of the previous E-step
    int* status;              // [F] SETK_NUM_* (worst over classes and iterations) or null
    int C, T, F, K, num_iters, update_alpha;
};

__global__ __launch_bounds__(256) void cgmm_k_kernel(CgmmKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char kmem[];
    const int C = a.C, T = a.T, F = a.F, K = a.K;
    const int NP = C * (C + 1) / 2;
    const int NCH = 256 / NP > 0 ? 256 / NP : 1;  // frame chunks per pair
    zc* Ak = reinterpret_cast<zc*>(kmem);               // [K][C][C]
    zc* Vk = Ak + kKMax * kCMax * kCMax;                 // [K][C][C]
    zc* red = Vk + kKMax * kCMax * kCMax;                // [NCH][NP]
    double* wk = reinterpret_cast<double*>(red + 256);   // [K][C] floored, scaled eigenvalues
    double* logdet = wk + kKMax * kCMax;                 // [K]
    double* alpha = logdet + kKMax;                      // [K]
    double* sumg = alpha + kKMax;                        // [K]
    double* scr = sumg + kKMax;                          // [256] block reductions

    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    double* gam = a.work + (size_t)f * 2 * K * T;  // [K][T]
    double* phi = gam + (size_t)K * T;             // [K][T]
    const cf* xf = a.spec + f;
    auto X = [&](int c, int t) { return xf[((size_t)c * T + t) * F]; };

    // pair of this thread (i <= j), chunk
    const int e = tid % NP, ch = tid / NP;
    int pi = 0, pj = 0;
    {
        int cnt = 0;
        for (int i = 0; i < C; ++i)
            for (int j = i; j < C; ++j) {
                if (cnt == e) { pi = i; pj = j; }
                ++cnt;
            }
    }
    auto block_sum = [&](double v) {
        scr[tid] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) scr[tid] += scr[tid + o];
            __syncthreads();
        }
        const double r = scr[0];
        __syncthreads();
        return r;
    };

    // ---- start: gamma (and the weights of the first M-step) ----
    const bool det_init = a.gamma0 == nullptr && a.init_mask == nullptr;  // K = 2: Rs = x x^H / T, Rn = I
    for (int k = 0; k < K; ++k)
        for (int t = tid; t < T; t += 256) {
            double g;
            if (a.gamma0) g = a.gamma0[((size_t)k * F + f) * T + t];
            else if (a.init_mask) g = k == 0 ? (double)a.init_mask[(size_t)t * F + f] : 1.0 - (double)a.init_mask[(size_t)t * F + f];
            else g = 1.0;
            gam[(size_t)k * T + t] = g;
        }
    if (tid < K) alpha[tid] = 1.0 / K;
    __syncthreads();

    for (int it = 0; it <= a.num_iters; ++it) {
        // Cgmm.update: alpha from the posteriors the M-step is about to use
        if (it > 0 && a.update_alpha) {
            for (int k = 0; k < K; ++k) {
                double s = 0.0;
                for (int t = tid; t < T; t += 256) s += gam[(size_t)k * T + t];
                s = block_sum(s);
                if (tid == 0) alpha[k] = s / T;
            }
            __syncthreads();
        }
        // ---- M-step: R_k = sum_t w x x^H / max(sum_t gamma_k, eps) ----
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int t = tid; t < T; t += 256) s += gam[(size_t)k * T + t];
            s = block_sum(s);
            if (tid == 0) sumg[k] = s;
            zc acc = zmk(0.0, 0.0);
            if (ch < NCH && !(it == 0 && det_init && k == 1)) {
                for (int t = ch; t < T; t += NCH) {
                    const double g = gam[(size_t)k * T + t];
                    const double w = it == 0 ? g : g * C / phi[(size_t)k * T + t];
                    const cf xi = X(pi, t), xj = X(pj, t);
                    acc.x += w * ((double)xi.x * xj.x + (double)xi.y * xj.y);
                    acc.y += w * ((double)xi.y * xj.x - (double)xi.x * xj.y);
                }
                red[ch * NP + e] = acc;
            }
            __syncthreads();
            if (tid < NP) {
                zc r = zmk(0.0, 0.0);
                for (int q = 0; q < NCH; ++q) {
                    r.x += red[q * NP + tid].x;
                    r.y += red[q * NP + tid].y;
                }
                double den = fmax(sumg[k], kEpsK);
                if (it == 0 && det_init) den = (double)T;  // Rs = x x^H / T (cluster.py:421)
                r.x /= den;
                r.y /= den;
                if (it == 0 && det_init && k == 1) r = zmk(pi == pj ? 1.0 : 0.0, 0.0);  // Rn = I
                zc* A = Ak + k * C * C;
                A[pi * C + pj] = pi == pj ? zmk(r.x, 0.0) : r;
                if (pi != pj) A[pj * C + pi] = zmk(r.x, -r.y);
            }
            __syncthreads();
        }
        // ---- Covariance: eigh, scale by max(w_max, eps), floor eps ----
        if (tid < K) {
            zc* A = Ak + tid * C * C;
            zc* V = Vk + tid * C * C;
            const bool conv = jacobi_herm(A, V, C);
            double wmax = -1e300;
            bool finite = true;
            for (int i = 0; i < C; ++i) {
                wmax = fmax(wmax, A[i * C + i].x);
                finite = finite && isfinite(A[i * C + i].x);
            }
            // numpy.linalg.eigh raises on these (cluster.py:104-113: "Eigenvalues did not converge");
            // the posteriors go on being computed, the status says what they are worth
            if (a.status && !(conv && finite))
                atomicMax(a.status + f, finite ? SETK_NUM_NOCONV : SETK_NUM_NONFINITE);
            const double sc = fmax(wmax, kEpsK);
            double ld = 0.0;
            for (int i = 0; i < C; ++i) {
                const double w = fmax(A[i * C + i].x / sc, kEpsK);
                wk[tid * C + i] = w;
                ld += log(w);
            }
            logdet[tid] = ld;
        }
        __syncthreads();
        // ---- phi = max(|x^H R^-1 x|, eps) / M in the eigenbasis; posteriors (Cgmm.predict) ----
        for (int t = tid; t < T; t += 256) {
            double lp[kKMax], ph[kKMax], lmax = -1e300;
            for (int k = 0; k < K; ++k) {
                const zc* V = Vk + k * C * C;
                double q = 0.0;
                for (int j = 0; j < C; ++j) {
                    double pr = 0.0, pm = 0.0;  // v_j^H x
                    for (int i = 0; i < C; ++i) {
                        const cf x = X(i, t);
                        const zc v = V[i * C + j];
                        pr += v.x * x.x + v.y * x.y;
                        pm += v.x * x.y - v.y * x.x;
                    }
                    q += (pr * pr + pm * pm) / wk[k * C + j];
                }
                ph[k] = fmax(fabs(q), kEpsK) / C;
                lp[k] = -(double)C * log(ph[k]) - logdet[k];
                lmax = fmax(lmax, lp[k]);
            }
            double den = 0.0, nom[kKMax];
            for (int k = 0; k < K; ++k) {
                nom[k] = exp(lp[k] - lmax) * alpha[k];
                den += nom[k];
            }
            den = fmax(den, kEpsK);
            for (int k = 0; k < K; ++k) {
                gam[(size_t)k * T + t] = nom[k] / den;
                phi[(size_t)k * T + t] = ph[k];
            }
        }
        __syncthreads();
    }
    for (int k = 0; k < K; ++k)
        for (int t = tid; t < T; t += 256)
            a.gamma_out[((size_t)k * T + t) * F + f] = (float)gam[(size_t)k * T + t];
}

}  // namespace

size_t cgmm_k_work_bytes(int K, int T, int F) { return (size_t)F * 2 * K * T * sizeof(double); }
bool cgmm_k_supported(int C, int K) { return C >= 1 && C <= kCMax && K >= 2 && K <= kKMax; }

hipError_t launch_cgmm_k(const float* spec, const double* gamma0, const float* init_mask, float* gamma_out,
                         double* work, int* status, int C, int T, int F, int K, int num_iters,
                         int update_alpha, hipStream_t s) {
    CgmmKArgs a;
    a.spec = reinterpret_cast<const cf*>(spec);
    a.gamma0 = gamma0;
    a.init_mask = init_mask;
    a.gamma_out = gamma_out;
    a.work = work;
    a.status = status;
    a.C = C;
    a.T = T;
    a.F = F;
    a.K = K;
    a.num_iters = num_iters;
    a.update_alpha = update_alpha;
    const size_t lds = (size_t)(2 * kKMax * kCMax * kCMax + 256) * sizeof(zc) +
                       (size_t)(kKMax * kCMax + 3 * kKMax + 256) * sizeof(double);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cgmm_k_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cgmm_k_kernel, dim3(F), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace setk
