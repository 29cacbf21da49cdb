"""The lane-level model of the matrix-core DFT-512 (tests/mcdft_model.py, the blueprint of
csrc/mcdft.h) against numpy.fft: forward and inverse, fp16-split operands, the tile layouts
and the odd-family tiles.  CPU only; the device twin is tools/ubench/mcdft_probe.hip and the
-m gpu parity suites (the fused kernels run on these transforms)."""
import numpy as np
import pytest

import mcdft_model as m


def _hann():
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(512) / 512)).astype(np.float32)


@pytest.mark.parametrize("amp", [1.0, 0.2, 1e-3, 3e-5])
def test_forward_matches_rfft(amp):
    rng = np.random.default_rng(11)
    x = np.clip(rng.standard_normal((24, 512)) * amp, -1, 1).astype(np.float32)
    xw = x * _hann()
    ref = np.fft.rfft(xw.astype(np.float64), axis=1)
    got = m.forward(xw)
    rel = np.sqrt((np.abs(got - ref) ** 2).mean() / (np.abs(ref) ** 2).mean())
    # 22 significant bits per operand: fp32 class (the 1e-4 STFT bar of the north star is 1000x away);
    # below ~1e-4 of full scale the lo halves go subnormal (absolute floor 2^-24 / 2^10 per
    # sample) and the error degrades gracefully: 1e-6 at one PCM16 step of amplitude
    assert rel < (3e-7 if amp >= 1e-3 else 3e-6), rel


def test_forward_impulses_pin_every_index_map():
    # an impulse at n has the spectrum exp(-2 pi i n k / 512): any slip in (n1, n2), (k1, q), the
    # conjugate rows or the odd family shows up as an O(1) error in some bin
    xw = np.zeros((16, 512), np.float32)
    pos = [0, 1, 15, 16, 17, 31, 32, 100, 255, 256, 257, 300, 383, 496, 510, 511]
    for b, n in enumerate(pos):
        xw[b, n] = 0.75
    got = m.forward(xw)
    k = np.arange(257)
    for b, n in enumerate(pos):
        ref = 0.75 * np.exp(-2j * np.pi * n * k / 512)
        assert np.abs(got[b] - ref).max() < 2e-6, (n, np.abs(got[b] - ref).argmax())


def test_inverse_matches_irfft_and_scales_per_frame():
    rng = np.random.default_rng(5)
    Y = (rng.standard_normal((20, 257)) + 1j * rng.standard_normal((20, 257))).astype(np.complex64)
    Y *= (10.0 ** rng.uniform(-6, 3, size=(20, 1))).astype(np.float32)  # frames of very different level
    Y[:, 0] = Y[:, 0].real
    Y[:, 256] = Y[:, 256].real
    ref = np.fft.irfft(Y.astype(np.complex128), axis=1) * 512
    got = m.inverse(Y)
    for b in range(20):
        rel = np.sqrt(((got[b] - ref[b]) ** 2).mean() / (ref[b] ** 2).mean())
        assert rel < 3e-7, (b, rel)


def test_round_trip():
    rng = np.random.default_rng(2)
    xw = (rng.standard_normal((8, 512)) * 0.1).astype(np.float32) * _hann()
    y = m.inverse(m.forward(xw)) / 512
    assert np.abs(y - xw).max() < 2e-7


def test_bins_of_a_lane_ascend_and_cover_the_half_spectrum():
    seen = {}
    for c in range(16):
        for g in range(4):
            bins = [m.bin_of(c, g, r) for r in range(4)]
            assert bins == [bins[0] + 32 * r for r in range(4)]
            for r, b in enumerate(bins):
                if c == 0 and not (g < 2 or (g == 2 and r == 3)):
                    continue  # column 0: conjugate-side duplicates
                assert b not in seen
                seen[b] = (c, g, r)
    odd = {16 + 32 * q for q in range(8)}
    assert set(seen) | odd == set(range(257)) and not (set(seen) & odd)
