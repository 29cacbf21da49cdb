"""
Synthetic multi-channel workload of SURVEY 8(d) / BASELINE.md section 3 (used by
bench.py, smoke() and the examples).  Deterministic per utterance index.

    rng = numpy.random.default_rng(1234 + index)
    one point source s ~ N(0,1) delayed by an integer d_c in [0, 8) samples per
    channel (gain 0.3) + spatially white N(0, 0.05^2) noise, whole mix x 0.2
"""
import numpy as np
import scipy.signal

EPSILON = np.finfo(np.float32).eps


def synth_utterance(index, num_channels, num_samples, return_parts=False):
    rng = np.random.default_rng(1234 + index)
    src = rng.standard_normal(num_samples + 8).astype(np.float32)
    delays = rng.integers(0, 8, size=num_channels)
    speech = np.stack([src[8 - d:8 - d + num_samples] for d in delays]) * 0.3
    noise = rng.standard_normal((num_channels, num_samples)).astype(np.float32) * 0.05
    mix = ((speech + noise) * 0.2).astype(np.float32)
    if return_parts:
        return mix, (speech * 0.2).astype(np.float32), (noise * 0.2).astype(np.float32)
    return mix



def synth_scene(index, num_channels, num_samples, return_parts=False):
    """A scene with time-frequency structure for the blind (CGMM) path: a gated,
    spectrally coloured point source through a short random FIR per channel, in
    spatially correlated (full-rank) diffuse noise.  default_rng(4321 + index)."""
    rng = np.random.default_rng(4321 + index)
    C, N = num_channels, num_samples
    # source: coloured noise (two resonances), gated on/off in 0.1 - 0.6 s segments
    src = rng.standard_normal(N + 64)
    for fc, r in ((rng.uniform(300, 900), 0.97), (rng.uniform(1500, 3000), 0.9)):
        a1, a2 = -2 * r * np.cos(2 * np.pi * fc / 16000.0), r * r
        src = src + 0.7 * scipy.signal.lfilter([1.0], [1.0, a1, a2], src) * (1 - r)
    env = np.zeros(N + 64)
    pos, on = 0, bool(rng.integers(0, 2))
    while pos < N + 64:
        seg = int(rng.uniform(0.1, 0.6) * 16000)
        if on:
            env[pos:pos + seg] = rng.uniform(0.5, 1.0)
        on = not on
        pos += seg
    src = src * env
    src = src / max(np.sqrt(np.mean(src**2)), 1e-9)
    taps = rng.standard_normal((C, 16)) * np.exp(-np.arange(16) / 3.0)
    taps[:, 0] += 1.0
    speech = np.stack([np.convolve(src, taps[c])[32:32 + N] for c in range(C)]) * 0.25
    mixm = np.eye(C) + 0.4 * rng.standard_normal((C, C))
    noise = (mixm @ rng.standard_normal((C, N))) * 0.08
    mix = ((speech + noise) * 0.2).astype(np.float32)
    if return_parts:
        return mix, (speech * 0.2).astype(np.float32), (noise * 0.2).astype(np.float32)
    return mix


def irm_from_spectra(S, V):
    """compute_mask.py:77-107 "irm": |S| / sqrt(|S|^2 + |V|^2 + eps) on
    channel-0 spectrograms (any matching shape)."""
    s = np.abs(S)
    v = np.abs(V)
    return (s / np.sqrt(s**2 + v**2 + EPSILON)).astype(np.float32)
