#!/usr/bin/env python
"""Throughput of BASELINE configs[4]: 6-ch 16 kHz CGMM (K=2, 20 EM iterations) mask
estimation feeding MVDR, one GPU, inputs resident in HBM.  Prints one JSON line.
(Side measurement for DESIGN.md; bench.py is the contract benchmark.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=64)
    ap.add_argument("--channels", type=int, default=6)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import torch
    from setk_amd import _ffi, synth
    from setk_amd.engine import CgmmEstimator
    dev = torch.device("cuda", 0)
    N = int(a.seconds * 16000)
    ctx = _ffi.Context(0)
    est = CgmmEstimator(num_iters=a.iters, ctx=ctx)
    distinct = [torch.from_numpy(synth.synth_utterance(i, a.channels, N)).to(dev) for i in range(min(8, a.utts))]
    audio = [distinct[i] if i < 8 else distinct[i % 8].clone() for i in range(a.utts)]
    T = ctx.num_frames(N) if ctx.plan else None
    est._plan()
    T = ctx.num_frames(N)
    L = ctx.istft_num_samples(T)
    waves = [torch.empty(L, dtype=torch.float32, device=dev) for _ in range(a.utts)]
    opts = _ffi.BfOpts(kind=_ffi.BF_MVDR, flags=_ffi.FLAG_CLAMP_MASK, pmwf_ref=-1)

    def step():
        masks = est.estimate_device(audio)
        # soften: the synthetic scene yields near-binary masks (rank-deficient noise covariance)
        ctx.enhance_batch(opts, a.channels, [t.data_ptr() for t in audio], [N] * a.utts,
                          [m.data_ptr() for m in masks], None, [w.data_ptr() for w in waves],
                          want_status=False)
        torch.cuda.synchronize()

    step()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"workload": f"{a.channels}-ch {a.seconds:g} s x {a.utts} utterances, CGMM "
                      f"{a.iters} it -> MVDR", "ms_per_batch": round(dt * 1e3, 2),
                      "rtf": round(a.utts * a.seconds / dt, 1)}))


if __name__ == "__main__":
    main()
