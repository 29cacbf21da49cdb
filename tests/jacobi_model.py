"""
numpy model of the solve inside cgmm_bin_em_kernel (setk_amd/csrc/cgmm_bin.hip): the
two-sided Jacobi in the parallel (round-robin) order with float32 rotation angles, the
stopping rule and the eigenvalue floor, written lane by lane the way the kernel does it.
The kernel was written from this model; tests/test_host_cpu.py checks the model against
LAPACK (no GPU needed), the GPU tests check the kernel against the oracle.
"""
import numpy as np

EPS = np.finfo(np.float32).eps
TOL2 = 1e-18    # rotate while |a_pq|^2 > TOL2 a_pp a_qq          (kTol2)
LAST2 = 1e-8    # a sweep whose rotations all start below is last (kLast2)


def rr_partner(r, k, m):
    """circle method, m even: partner of index k in round r (rr_partner in the kernel)"""
    n1 = m - 1
    if k == n1:
        return r
    j = (2 * r - k) % n1
    return n1 if j == k else j


def jacobi(R, V0=None, max_sweeps=14):
    n = R.shape[0]
    m = n + (n & 1)
    A = np.zeros((m, m), complex)
    A[:n, :n] = R
    V = np.eye(m, dtype=complex)
    if V0 is not None:                      # warm start: A = V^H R V
        V[:n, :n] = V0
        A[:n, :n] = V0.conj().T @ R @ V0
    tr = np.trace(A).real
    fl = max(EPS * tr / n, 1e-290)
    sweeps = 0
    for sweeps in range(1, max_sweeps + 1):
        big = False
        for r in range(m - 1):
            J = np.eye(m, dtype=complex)
            for p in range(m):
                q = rr_partner(r, p, m)
                if q < p:
                    continue
                apq, app, aqq = A[p, q], A[p, p].real, A[q, q].real
                g2 = apq.real**2 + apq.imag**2
                den = max(app, fl) * max(aqq, fl)
                if not g2 > TOL2 * den:
                    continue
                big |= g2 > LAST2 * den
                with np.errstate(all="ignore"):
                    rg = np.float32(1) / np.sqrt(np.float32(g2))
                    tau = np.float32(0.5) * np.float32(aqq - app) * rg
                    if not abs(tau) < 1e18:
                        continue
                    t = float(np.copysign(np.float32(1), tau) /
                              (abs(tau) + np.sqrt(np.float32(1) + tau * tau)))
                c = 1 / np.sqrt(1 + t * t)
                sg = apq / np.sqrt(g2) * (t * c)
                J[p, p] = J[q, q] = c
                J[p, q] = sg
                J[q, p] = -np.conj(sg)
            A = J.conj().T @ A @ J
            V = V @ J
        if not big:
            break
    return np.diag(A).real[:n].copy(), V[:n, :n], sweeps


def effective_inverse(R, V0=None):
    """(R_eff^-1, log det R_eff, V, sweeps) with the reference's scaling and floor
    (libs/cluster.py:107-113)."""
    w, V, sweeps = jacobi(R, V0)
    w = np.maximum(w / max(w.max(), EPS), EPS)
    return (V / w) @ V.conj().T, float(np.sum(np.log(w))), V, sweeps


def lapack_reference(R):
    w, v = np.linalg.eigh((R + R.conj().T) / 2)
    w = np.maximum(w / max(w.max(), EPS), EPS)
    return (v / w) @ v.conj().T, float(np.sum(np.log(w))), v * np.sqrt(w)


def certificate(R):
    """The fast path's bound: True when the eigenvalue floor is provably inactive."""
    tr = np.trace(R).real
    try:
        L = np.linalg.cholesky(R)
    except np.linalg.LinAlgError:
        return False
    tinv = np.sum(np.abs(np.linalg.inv(L).astype(np.complex64))**2)
    return bool(tinv * tr * 1.1 * EPS < 1.0)
