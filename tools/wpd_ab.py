#!/usr/bin/env python
"""engine.BatchWpd: per-call time of a batch (4-ch 10 s, 16-bit frames in, PCM_16 + mask out) with
the CGMM of an outer iteration as ONE launch for the batch (default) or one per utterance
(SETK_WPD_CGMM_PER_UTT=1), and the per-utterance numpy mirror beside it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(utts=8, seconds=10.0, mirror=True):
    import torch
    from setk_amd import synth
    from setk_amd.engine import BatchWpd, Pcm16Frames
    from setk_amd.libs import wavio
    N = int(seconds * 16000)
    mix = [Pcm16Frames(np.ascontiguousarray(wavio.float_to_pcm16(synth.synth_utterance(3150 + i, 4, N).T)))
           for i in range(4)]
    mix = (mix * ((utts + 3) // 4))[:utts]
    eng = BatchWpd(taps=10, delay=3, context=1, wpd_iters=3, cgmm_iters=20, pcm16=True)
    eng.run(mix)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        res = eng.run(mix)
    dt = (time.perf_counter() - t0) / reps
    print(f"resident ({'per-utterance' if eng._cgmm_per_utt else 'batched'} CGMM): "
          f"{1e3 * dt / utts:.2f} ms per utterance, {utts} per call, failed {sum(r is None for r in res)}")
    if mirror:
        eng._one_by_mirror(mix[0])  # (first call: plans, module imports)
        t0 = time.perf_counter()
        for u in mix[:2]:
            eng._one_by_mirror(u)
        dm = (time.perf_counter() - t0) / 2
        print(f"numpy mirror: {1e3 * dm:.2f} ms per utterance -> {dm / (dt / utts):.1f} x")
    eng.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8, mirror=os.environ.get("WPD_AB_MIRROR", "1") == "1")
