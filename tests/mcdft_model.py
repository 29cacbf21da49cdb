"""Lane-level numpy model of the matrix-core real DFT-512 (csrc/mcdft.h).

Test infrastructure: the kernels of csrc/mcdft.h were written from this model
and tests/test_mcdft_model.py pins it against numpy.fft (forward and inverse,
float16-split operands, the MFMA register layouts of gfx950).

A 512-point real transform is factored 512 = 32 x 16 (n = 16 n1 + n2,
k = k1 + 32 q) into two dense contractions that run on the matrix cores as
v_mfma_f32_16x16x32_f16, every fp32 operand split into an fp16 pair (hi, lo)
and every product taken as hi*hi + lo*hi + hi*lo:

    stage 1   A[k1][n2] = sum_n1 xw[16 n1 + n2] W32^(n1 k1)     (real input: k1 = 0..16)
    twiddle   B[k1][n2] = A[k1][n2] W512^(n2 k1)
    stage 2   Z[k1][q]  = sum_n2 B[k1][n2] W16^(n2 q)
    bins      X[k1 + 32 q] = Z[k1][q]            q < 8
              X[32 (16 - q) - k1] = conj Z[k1][q] q >= 8          (k1 = 1..15)
              X[32 q] = Z[0][q], q = 0..8 (column 0 carries the real A[0][.])
              X[16 + 32 q] = sum_n2 A[16][n2] W32^(n2 (2 q + 1))  (the "odd" family, batched over
                                                                   16 frames in one extra tile)

MFMA 16x16x32 layouts (cdna_hip_programming.md section 3; checked on hardware by
tools/ubench/mcdft_probe.hip):  A: lane l holds A[l % 16][8 (l / 16) + e], e < 8;
B: lane l holds B[8 (l / 16) + e][l % 16];  D: lane l, register r holds
D[4 (l / 16) + r][l % 16].
"""
import numpy as np

N = 512
F = 257


# ---------------------------------------------------------------- fp16 pairs
def split16(x):
    """fp32 -> (hi, lo) float16 pair, hi = rtz-free round-to-nearest, lo = fp16(x - hi)."""
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def mfma(a, b, c=None):
    """D = A B + C with fp16 operands and fp32 accumulation (products exact in fp32)."""
    d = a.astype(np.float64) @ b.astype(np.float64)
    if c is not None:
        d = d + c
    return d.astype(np.float32)


def mm3(ah, al, bh, bl):
    """hi*hi + lo*hi + hi*lo, three MFMAs chained through the accumulator."""
    d = mfma(ah, bh)
    d = mfma(al, bh, d)
    d = mfma(ah, bl, d)
    return d


# ---------------------------------------------------------------- constant tiles (fp64 -> fp16 pairs)
def k1_n1():
    """K index of the stage-1 data operand: k = 8 g + e <-> n1 = 4 g + e (e < 4: first half of
    the frame), 16 + 4 g + e - 4 (e >= 4: second half).  A lane's registers e >= 4 of frame t are
    its registers e < 4 of frame t + 1 when hop = 256: consecutive frames share them."""
    k = np.arange(32)
    g, e = k // 8, k % 8
    return np.where(e < 4, 4 * g + e, 16 + 4 * g + e - 4)


def stage1_tiles():
    """B operands of stage 1 (rows in the K order of k1_n1): Mc[n1][c] = cos(2 pi n1 c / 32);
    Ms[n1][c] = -sin(2 pi n1 c / 32), column 0 of Ms carries the k1 = 16 row (-1)^n1."""
    n1 = k1_n1()[:, None]
    c = np.arange(16)[None, :]
    mc = np.cos(2 * np.pi * n1 * c / 32)
    ms = -np.sin(2 * np.pi * n1 * c / 32)
    ms[:, 0] = (-1.0) ** k1_n1()
    return mc, ms


def k2_part_n2():
    """K index of the stage-2 data operand: k = 8 g + e <-> (part = e / 4, n2 = 4 g + e % 4)."""
    k = np.arange(32)
    g, e = k // 8, k % 8
    return e // 4, 4 * g + e % 4


def row_q():
    """Row 4 g + r of a result tile <-> q: rows 8..15 run backwards inside each group of four
    (q = 4 g + 3 - r), so that the bins of a lane ascend with r in every lane:
    bin = c + 32 (4 g + r) for g < 2,  32 (13 - 4 g + r) - c for g >= 2."""
    row = np.arange(16)
    g, r = row // 4, row % 4
    return np.where(g < 2, row, 4 * g + 3 - r)


def bin_of(c, g, r):
    return c + 32 * (4 * g + r) if g < 2 else 32 * (13 - 4 * g + r) - c


def stage2_tiles():
    """A operands of stage 2 (rows <-> q by row_q): out_re = sum cos Br + sin Bi, out_im = sum -sin Br + cos Bi."""
    part, n2 = k2_part_n2()
    q = row_q()[:, None]
    ang = 2 * np.pi * n2[None, :] * q / 16
    ar = np.where(part[None, :] == 0, np.cos(ang), np.sin(ang))
    ai = np.where(part[None, :] == 0, -np.sin(ang), np.cos(ang))
    ai[8:, :] *= -1  # rows 8..15 (q >= 8) hold the conjugate bins: the lanes end up with X itself
    return ar, ai


def odd_tile():
    """A operand of the odd-family tile: rows 4 g + r <-> (q = 2 g + r / 2, part = r % 2),
    K = 32: k < 16 multiplies hi(A16[n2 = k]), k >= 16 multiplies lo(A16[n2 = k - 16])."""
    rows = np.arange(16)
    q, part = 2 * (rows // 4) + (rows % 4) // 2, rows % 2
    n2 = np.arange(16)[None, :]
    ang = 2 * np.pi * n2 * (2 * q[:, None] + 1) / 32
    t = np.where(part[:, None] == 0, np.cos(ang), -np.sin(ang))
    return np.concatenate([t, t], axis=1)  # [16][32]


def twiddle(a, b):
    """W512^(a b)."""
    return np.exp(-2j * np.pi * a * b / 512)


# ---------------------------------------------------------------- forward
def forward(xw, scale=1024.0):
    """xw: [B][512] windowed frames (fp32).  Returns X [B][257] complex64 through the
    fp16-split pipeline.  `scale` (a power of two) lifts the operands into the fp16 range;
    the result is divided by it again."""
    xw = np.asarray(xw, np.float32) * np.float32(scale)
    nb = xw.shape[0]
    mc, ms = stage1_tiles()
    mch, mcl = split16(mc)
    msh, msl = split16(ms)
    ar, ai = stage2_tiles()
    arh, arl = split16(ar)
    aih, ail = split16(ai)
    X = np.zeros((nb, F), np.complex64)
    a16 = np.zeros((nb, 16), np.float32)
    n2 = np.arange(16)
    c = np.arange(16)
    tw = twiddle(n2[:, None], c[None, :])  # [n2][c]
    tr, ti = tw.real.astype(np.float32), tw.imag.astype(np.float32)
    tr_im = tr.copy()
    tr_im[:, 0] = 0.0  # column 0: Bi = 0 (its Ds carries A16, routed to the odd tile)
    part, kn2 = k2_part_n2()
    for b in range(nb):
        a1 = xw[b].reshape(32, 16).T[:, k1_n1()]  # [n2][k <-> n1]
        a1h, a1l = split16(a1)
        dc = mm3(a1h, a1l, mch, mcl)  # [n2][c]
        ds = mm3(a1h, a1l, msh, msl)
        a16[b] = ds[:, 0]
        br = (dc * tr - ds * ti).astype(np.float32)
        bi = (dc * ti + ds * tr_im).astype(np.float32)
        b2 = np.where(part[:, None] == 0, br[kn2, :], bi[kn2, :])  # [k][c]
        b2h, b2l = split16(b2)
        zr = mm3(arh, arl, b2h, b2l)  # [q][c]
        zi = mm3(aih, ail, b2h, b2l)
        z = zr + 1j * zi
        for cc in range(1, 16):
            for row in range(16):
                X[b, bin_of(cc, row // 4, row % 4)] = z[row, cc]
        for row in range(16):
            if row_q()[row] <= 8:
                X[b, bin_of(0, row // 4, row % 4)] = z[row, 0]
    # odd family, 16 frames per tile
    ot = odd_tile()
    oth, otl = split16(ot)
    otl2 = otl.copy()
    otl2[:, 16:] = 0  # second MFMA: T_lo x hi only
    for b0 in range(0, nb, 16):
        blk = a16[b0:b0 + 16]
        nblk = blk.shape[0]
        h, l = split16(blk)
        bop = np.zeros((32, 16), np.float16)
        bop[:16, :nblk] = h.T
        bop[16:, :nblk] = l.T
        d = mfma(oth, bop)
        d = mfma(otl2, bop, d)  # [row][frame]
        for j in range(nblk):
            for q in range(8):
                g, rr = q // 2, 2 * (q % 2)
                X[b0 + j, 16 + 32 * q] = d[4 * g + rr, j] + 1j * d[4 * g + rr + 1, j]
    return (X / np.float32(scale)).astype(np.complex64)


# ---------------------------------------------------------------- inverse
def inv_stage2_tiles():
    """B operands of the inverse stage over q (data = A operand, rows k1):
    C[k1][n2] = sum_q Yz[k1][q] W16^(-n2 q); K index k = 8 g + e <-> (part = e / 4, q = 4 g + e % 4).
    out_re = sum cos Yr - sin Yi, out_im = sum sin Yr + cos Yi."""
    part, row = k2_part_n2()
    q = row_q()[row]
    n2 = np.arange(16)[None, :]
    ang = 2 * np.pi * q[:, None] * n2 / 16
    br = np.where(part[:, None] == 0, np.cos(ang), -np.sin(ang))
    bi = np.where(part[:, None] == 0, np.sin(ang), np.cos(ang))
    conj = (part == 1) & (q >= 8)  # the data rows q >= 8 arrive as X (not conjugated)
    br[conj, :] *= -1
    bi[conj, :] *= -1
    return br, bi  # [k][n2]


def inv_stage1_tiles():
    """A operands of the inverse stage over k1 (rows n1, two tiles of 16):
    y[16 n1 + n2] = E0 + (-1)^n1 E16 + 2 sum_{k1=1..15} Er cos(th) - Ei sin(th), th = 2 pi n1 k1 / 32.
    K index k = 8 g + e <-> (part = e / 4, k1 = 4 g + e % 4); the (im, k1 = 0) slot carries E16."""
    part, k1 = k2_part_n2()
    n1 = np.arange(32)[:, None]
    ang = 2 * np.pi * n1 * k1[None, :] / 32
    g = np.where(part[None, :] == 0, 2 * np.cos(ang), -2 * np.sin(ang))
    g[:, (part == 0) & (k1 == 0)] = 1.0
    g[:, (part == 1) & (k1 == 0)] = ((-1.0) ** np.arange(32))[:, None]
    return g  # [n1][k]


def inv_odd_tile():
    """E16[n2] = 2 Re sum_{q<8} Y[16 + 32 q] W512^(-16 n2) W16^(-n2 q)
               = 2 sum_q Yr cos(ph) - Yi sin(ph), ph = 2 pi n2 (2 q + 1) / 32.
    B operand [K][n2]: k < 16 <-> hi of (q = k / 2, part = k % 2), k >= 16 the lo parts."""
    k = np.arange(16)
    q, part = k // 2, k % 2
    n2 = np.arange(16)[None, :]
    ph = 2 * np.pi * n2 * (2 * q[:, None] + 1) / 32
    t = np.where(part[:, None] == 0, 2 * np.cos(ph), -2 * np.sin(ph))
    return np.concatenate([t, t], axis=0)  # [32][16]


def inverse(Y):
    """Y: [B][257] complex (Hermitian half spectra).  Returns y [B][512] fp32 = irfft(Y) * 512
    (unscaled; the caller's window table carries 1/512) through the fp16-split pipeline with
    a per-frame power-of-two scale."""
    Y = np.asarray(Y, np.complex64)
    nb = Y.shape[0]
    br, bi = inv_stage2_tiles()
    brh, brl = split16(br)
    bih, bil = split16(bi)
    g1 = inv_stage1_tiles()
    g1h, g1l = split16(g1)
    it = inv_odd_tile()
    ith, itl = split16(it)
    itl2 = itl.copy()
    itl2[16:, :] = 0
    part, kq = k2_part_n2()
    k1 = np.arange(16)
    n2 = np.arange(16)
    out = np.zeros((nb, N), np.float32)
    # per-frame scale: max |component| -> below 2^11
    mx = np.maximum(np.abs(Y.real).max(axis=1), np.abs(Y.imag).max(axis=1))
    ex = np.where(mx > 0, np.ceil(np.log2(np.maximum(mx, 1e-38))), 0)
    sc = np.exp2(11 - ex).astype(np.float32)
    Ys = (Y * sc[:, None]).astype(np.complex64)
    # odd family: E16 for 16 frames per tile (A = data rows = frames)
    e16 = np.zeros((nb, 16), np.float32)
    for b0 in range(0, nb, 16):
        blk = Ys[b0:b0 + 16][:, 16 + 32 * np.arange(8)]  # [j][q]
        nblk = blk.shape[0]
        flat = np.zeros((nblk, 16), np.float32)
        flat[:, 0::2] = blk.real
        flat[:, 1::2] = blk.imag
        h, l = split16(flat)
        aop = np.zeros((16, 32), np.float16)
        aop[:nblk, :16] = h
        aop[:nblk, 16:] = l
        d = mfma(aop, ith)
        d = mfma(aop, itl2, d)  # [j][n2]
        e16[b0:b0 + nblk] = d[:nblk]
    for b in range(nb):
        # the forward's output form: lane (k1, q) holds X[bin_of(k1, q)]
        yz = np.zeros((16, 16), np.complex64)  # [k1][row]
        for c in range(16):
            for row in range(16):
                yz[c, row] = Ys[b, bin_of(c, row // 4, row % 4)]
        yz[0, 0] = yz[0, 0].real
        yz[0, 11] = yz[0, 11].real  # row 11 <-> q = 8: bin 256
        a = np.where(part[None, :] == 0, yz.real[:, kq], yz.imag[:, kq]).astype(np.float32)  # [k1][k]
        ah, al = split16(a)
        cr = mm3(ah, al, brh, brl)  # [k1][n2]
        ci = mm3(ah, al, bih, bil)
        tw = np.conj(twiddle(k1[:, None], n2[None, :]))
        er = (cr * tw.real - ci * tw.imag).astype(np.float32)
        ei = (cr * tw.imag + ci * tw.real).astype(np.float32)
        ei[0, :] = e16[b]  # the (im, k1 = 0) slot carries E16
        bop = np.where(part[:, None] == 0, er[kq, :], ei[kq, :])  # [k][n2]
        bh, bl = split16(bop)
        y = mm3(g1h, g1l, bh, bl)  # [n1][n2]
        out[b] = (y / sc[b]).reshape(N)
    return out
