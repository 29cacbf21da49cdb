#!/usr/bin/env python
"""Bit-level fingerprint of the fused path's outputs for the loaded library (SETK_LIB):
    SETK_LIB=$PWD/_abl/libsetk_x.so python tools/ab_bits.py
prints one sha256 per case (float32 and PCM16 waveforms of ragged batches at several hops and
channel counts).  Two builds that print the same lines are bit-identical on these cases."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from setk_amd import _ffi, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for C, hop, center, lens, kind in ((8, 256, True, [480000, 123457, 9000], 0), (4, 128, True, [64000, 30001], 0),
                                       (6, 160, False, [50000, 20000, 777 + 2048], 2), (1, 256, True, [16000], 0),
                                       (3, 64, True, [12000, 5000], 1), (8, 200, True, [40000], 0),
                                       (2, 512, True, [30000, 8192], 0)):
        ctx = _ffi.Context(0)
        ctx.stft_plan(512, hop, 512, center)
        h = hashlib.sha256()
        for pcm in (False, True):
            audio, masks, waves = [], [], []
            for u, N in enumerate(lens):
                mix = synth.synth_utterance(7000 + 13 * u + C, C, N).astype(np.float32)
                T = ctx.num_frames(N)
                rng = np.random.default_rng(u + 100 * C)
                audio.append(torch.from_numpy(mix).to(dev))
                masks.append(torch.from_numpy((0.1 + 0.8 * rng.random((T, 257))).astype(np.float32)).to(dev))
                L = ctx.istft_num_samples(T)
                waves.append(torch.zeros(L, dtype=torch.int16 if pcm else torch.float32, device=dev))
            opts = _ffi.BfOpts()
            opts.kind = kind
            opts.pmwf_ref = 0
            opts.flags = _ffi.FLAG_OUT_PCM16 if pcm else 0
            ctx.enhance_batch(opts, C, [a.data_ptr() for a in audio], lens, [m.data_ptr() for m in masks], None,
                              [w.data_ptr() for w in waves])
            torch.cuda.synchronize()
            for w in waves:
                h.update(w.cpu().numpy().tobytes())
        print(f"C={C} hop={hop} center={center} kind={kind}: {h.hexdigest()[:24]}")
        ctx.close()


if __name__ == "__main__":
    main()
