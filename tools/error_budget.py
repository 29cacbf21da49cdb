#!/usr/bin/env python
"""
Where does the deviation from the oracle on REAL recordings come from?  (round-5 review, 1b)

For `2spk.wav` (7 ch) and `noisy.wav` (5 ch) of the reference's spatial-clustering doc, with
the masks the unmodified reference estimated (tests/golden/doc_spatial_clustering.npz), per
beamformer: the deviation of the device's covariances, weights and waveform from the oracle's
(complex64 covariances, as the reference's einsum) AND of both from a float64 evaluation of
the same formulas on the same float32 STFT ("truth64").  If the device is as close to truth64
as the oracle is, the device-vs-oracle distance is two independent float32 roundings of the
covariances amplified by the conditioning of the solve -- nothing a higher-precision
reduction on the device could remove, because the oracle's (= the reference's) own rounding
is half of it.     python tools/error_budget.py  > profiles/round5_error_budget.txt
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")

import torch  # noqa: E402

from oracle import np_oracle as o  # noqa: E402  (checker, not product)
from setk_amd import _ffi  # noqa: E402

KINDS = {"mvdr": dict(kind=0), "gevd": dict(kind=1), "pmwf-0": dict(kind=2, pmwf_beta=0.0, pmwf_ref=-1)}


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))


def device(ctx, kind, samps, mask):
    dev = torch.device("cuda:0")
    C, N = samps.shape
    a = torch.from_numpy(np.ascontiguousarray(samps)).to(dev)
    m = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.float32)).to(dev)
    out = torch.empty(ctx.istft_num_samples(ctx.num_frames(N)), dtype=torch.float32, device=dev)
    taps = dict(Rs=torch.empty((1, 257, C, C), dtype=torch.complex64, device=dev),
                Rn=torch.empty((1, 257, C, C), dtype=torch.complex64, device=dev),
                weight=torch.empty((1, 257, C), dtype=torch.complex64, device=dev))
    st = ctx.enhance_batch(_ffi.BfOpts(flags=_ffi.FLAG_CLAMP_MASK, **KINDS[kind]), C, [a.data_ptr()], [N],
                           [m.data_ptr()], None, [out.data_ptr()], taps=taps)
    torch.cuda.synchronize()
    return out.cpu().numpy(), st, {k: v[0].cpu().numpy() for k, v in taps.items()}


def truth64(kind, X, mask):
    """The oracle's formulas in float64 / complex128 on the same (float32) spectrogram."""
    m = np.minimum(mask, 1).astype(np.float64)
    X = X.astype(np.complex128)
    Rs, Rn = o.compute_covar(X, m), o.compute_covar(X, 1 - m)
    if kind == "mvdr":
        w = o.mvdr_weight(Rs, Rn, gauge=True)
    elif kind == "gevd":
        w = o.gevd_weight(Rs, Rn, gauge=True)
    else:
        w = o.pmwf_weight(Rs, Rn, beta=0, gauge=True)
    return Rs, Rn, w


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "doc_spatial_clustering.npz"))
    ctx = _ffi.Context(0)
    ctx.stft_plan(512, 256, 512, True)
    cases = {"2spk (7 ch)": ((g["pcm_2spk"].astype(np.float32) / 32768.0).T, g["saved_2spk"][0]),
             "noisy (5 ch)": ((g["pcm_noisy"].astype(np.float32) / 32768.0).T, g["saved_noisy"])}
    print("relative deviations (2-norm over all bins / channels / samples); cond = cond(Rn), float64\n")
    for name, (samps, mask) in cases.items():
        samps = np.ascontiguousarray(samps)
        for kind in KINDS:
            ref, parts = o.enhance_utterance(samps, mask, kind=kind, gauge=True, return_parts=True)
            wav, st, t = device(ctx, kind, samps, mask)
            try:
                Rs64, Rn64, w64 = truth64(kind, parts["stft"], mask)
            except np.linalg.LinAlgError as e:
                print(f"{name:13s} {kind:7s} status {st}  the float64 evaluation raises ({e}): in float64 some Rn "
                      f"is EXACTLY singular (the reference's complex64 rounding is what lets it through); "
                      f"waveform device-oracle {rel(wav[:len(ref)], ref[:len(wav)]):.2e}")
                continue
            cond = np.linalg.cond(Rn64)
            wp = cond < 1e4  # well-posed bins
            e_w = np.linalg.norm(t["weight"] - parts["weight"], axis=1) / np.linalg.norm(parts["weight"], axis=1)
            e_w_dev64 = np.linalg.norm(t["weight"] - w64, axis=1) / np.linalg.norm(w64, axis=1)
            e_w_ora64 = np.linalg.norm(parts["weight"] - w64, axis=1) / np.linalg.norm(w64, axis=1)
            print(f"{name:13s} {kind:7s} status {st}  cond(Rn): median {np.median(cond):.3g}, max {cond.max():.3g}, "
                  f"{int(wp.sum())} bins < 1e4")
            print(f"    Phi_nn   device-oracle {rel(t['Rn'], parts['Rn']):.2e} | device-truth64 {rel(t['Rn'], Rn64):.2e} | "
                  f"oracle-truth64 {rel(parts['Rn'], Rn64):.2e}")
            print(f"    Phi_ss   device-oracle {rel(t['Rs'], parts['Rs']):.2e} | device-truth64 {rel(t['Rs'], Rs64):.2e} | "
                  f"oracle-truth64 {rel(parts['Rs'], Rs64):.2e}")
            print(f"    weights  device-oracle all bins {rel(t['weight'], parts['weight']):.2e}; per bin (cond < 1e4): "
                  f"median {np.median(e_w[wp]):.2e}, max {e_w[wp].max():.2e} | device-truth64 median "
                  f"{np.median(e_w_dev64[wp]):.2e}, max {e_w_dev64[wp].max():.2e} | oracle-truth64 median "
                  f"{np.median(e_w_ora64[wp]):.2e}, max {e_w_ora64[wp].max():.2e}")
            # the waveform the oracle's own pipeline gives with the float64 weights
            enh64 = o.beamform(w64.astype(np.complex128), parts["stft"].astype(np.complex128))
            ref64 = o.inverse_stft(enh64, norm=parts["norm"], frame_len=512, frame_hop=256, window="hann",
                                   center=True, transpose=False)
            n = min(len(wav), len(ref), len(ref64))
            print(f"    waveform device-oracle {rel(wav[:n], ref[:n]):.2e} | device-truth64 {rel(wav[:n], ref64[:n]):.2e} | "
                  f"oracle-truth64 {rel(ref[:n], ref64[:n]):.2e}")
    ctx.close()


if __name__ == "__main__":
    main()
