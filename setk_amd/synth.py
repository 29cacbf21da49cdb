"""
Synthetic multi-channel workload of SURVEY 8(d) / BASELINE.md section 3 (used by
bench.py, smoke() and the examples).  Deterministic per utterance index.

    rng = numpy.random.default_rng(1234 + index)
    one point source s ~ N(0,1) delayed by an integer d_c in [0, 8) samples per
    channel (gain 0.3) + spatially white N(0, 0.05^2) noise, whole mix x 0.2
"""
import numpy as np

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


def irm_from_spectra(S, V):
    """compute_mask.py:77-107 "irm": |S| / sqrt(|S|^2 + |V|^2 + eps) on
    channel-0 spectrograms (any matching shape)."""
    s = np.abs(S)
    v = np.abs(V)
    return (s / np.sqrt(s**2 + v**2 + EPSILON)).astype(np.float32)
