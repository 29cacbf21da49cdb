#!/usr/bin/env python
"""Kernel-level view of the secondary paths (for rocprofv3 --kernel-trace --stats): the resident
WPE engine, the unfused engine at n_fft = 400 and at 12 channels.  python tools/prof_secondary.py <which>"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(which):
    import torch
    from setk_amd import synth
    from setk_amd.engine import BatchDereverb, BatchEnhancer
    from setk_amd.libs.utils import device_stft
    N = 160000
    if which == "wpe":
        mixes = [synth.synth_utterance(3000 + i, 4, N).astype(np.float32) for i in range(4)] * 4
        eng = BatchDereverb(taps=10, delay=3, context=1, num_iters=3, frame_len=512, frame_hop=128,
                            window="hann", center=True)
        fn = lambda: eng.run(mixes)  # noqa: E731
        n = len(mixes)
    else:
        C, kw = (4, dict(frame_len=400, frame_hop=160, round_power_of_two=False)) if which == "nfft400" else \
            (12, dict(frame_len=512, frame_hop=256))
        eng = BatchEnhancer(beamformer="mvdr", **kw)
        items = []
        for i in range(16):
            mix, sp, nz = synth.synth_utterance(3200 + (i % 4), C, N, return_parts=True)
            st = device_stft(np.stack([sp[0], nz[0]]), kw["frame_len"], kw["frame_hop"],
                             kw.get("round_power_of_two", True), True, "hann")
            items.append((mix, synth.irm_from_spectra(st[0], st[1]), None))
        fn = lambda: eng.enhance(items)  # noqa: E731
        n = 16
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print(f"{which}: {1e3 * (time.perf_counter() - t0) / 3 / n:.3f} ms per utterance ({n} per call)")


if __name__ == "__main__":
    main(sys.argv[1])
