#!/usr/bin/env python
"""Throughput of the paths AROUND the fused hot path (SURVEY 8f-4 consumers and the
unfused engine): one JSON line, also embedded in bench.py's `other_configs`.

  wpe_4ch_10taps      GWPE, 4 ch x 10 s, 10 taps, delay 3, 3 iterations (libs/wpe.py:84-110)
  df_on_mask_4ch      compute_df_on_mask.py per utterance: STFT -> masked covariance ->
                      principal eigenvector -> directional features
  mvdr_nfft400_4ch    --frame-len 400 --round-power-of-two false (Bluestein STFT, unfused engine)
  mvdr_12ch           12 channels (wide covariance / 16-lane solve, unfused engine)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SR = 16000


def timed(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def run(utts=16, seconds=10.0):
    import torch
    from setk_amd import _ffi, synth
    from setk_amd.engine import BatchEnhancer
    from setk_amd.libs import wpe as W
    from setk_amd.libs import spatial as S
    from setk_amd.libs import beamformer as B
    from setk_amd.libs.utils import device_stft
    N = int(seconds * SR)
    res = {}
    # ---- WPE ----
    mixes = [synth.synth_utterance(3000 + i, 4, N) for i in range(min(utts, 4))]
    specs = [np.ascontiguousarray(np.transpose(device_stft(m, 512, 128, True, True, "hann"),
                                               (2, 0, 1))) for m in mixes]   # F x N x T
    dt = timed(lambda: [W.wpe(s, taps=10, delay=3, context=1, num_iters=3) for s in specs], 2)
    res["wpe_4ch_10taps"] = {
        "workload": f"4-ch {seconds:g} s, hop 128, 10 taps x 3 iterations, numpy in / numpy out, "
                    f"{len(specs)} utterances one at a time",
        "ms_per_utt": round(1e3 * dt / len(specs), 2),
        "value": round(len(specs) * seconds / dt, 1)}
    if hasattr(W, "wpe_batch"):
        dtb = timed(lambda: W.wpe_batch(specs * 4, taps=10, delay=3, context=1, num_iters=3), 2)
        res["wpe_4ch_10taps_batched"] = {
            "workload": f"the same, {4 * len(specs)} utterances per setk_wpe_batch call",
            "ms_per_utt": round(1e3 * dtb / (4 * len(specs)), 2),
            "value": round(4 * len(specs) * seconds / dtb, 1)}
    # the CLI's engine: samples in, waveforms out, spectra never leave the device
    from setk_amd.engine import BatchDereverb
    eng = BatchDereverb(taps=10, delay=3, context=1, num_iters=3, frame_len=512, frame_hop=128,
                        window="hann", center=True)
    auds = [m.astype(np.float32) for m in mixes] * 4
    dtr = timed(lambda: eng.run(auds), 3)
    res["wpe_4ch_10taps_resident"] = {
        "workload": f"the same from samples to samples (engine.BatchDereverb: STFT -> WPE -> inverse "
                    f"STFT on the device), {len(auds)} utterances per call",
        "ms_per_utt": round(1e3 * dtr / len(auds), 2),
        "value": round(len(auds) * seconds / dtr, 1)}
    # ---- directional features on a mask ----
    mix, sp, nz = synth.synth_utterance(3100, 4, N, return_parts=True)
    spec = device_stft(mix, 512, 256, True, True, "hann")          # C x T x F
    s0 = device_stft(np.stack([sp[0], nz[0]]), 512, 256, True, True, "hann")
    mask = synth.irm_from_spectra(s0[0], s0[1])

    def df_one():
        obs = np.transpose(spec, (0, 2, 1))
        Rs = B.compute_covar(obs, mask)
        sv = B.solve_pevd(Rs)
        return S.directional_feats(obs, sv.T)
    dt = timed(df_one, 5)
    res["df_on_mask_4ch"] = {"workload": f"4-ch {seconds:g} s, compute_covar -> solve_pevd -> "
                                         "directional_feats, numpy in / numpy out, one utterance",
                             "ms_per_utt": round(1e3 * dt, 2), "value": round(seconds / dt, 1)}
    # the CLI's engine since round 5: samples + masks in, feature maps out, spectrograms /
    # covariances / steer vectors stay on the device (engine.BatchDirectionalFeatures)
    from setk_amd.engine import BatchDirectionalFeatures, Pcm16Frames
    from setk_amd.libs import wavio
    dfe = BatchDirectionalFeatures([(0, 1), (0, 2), (1, 3)], frame_len=512, frame_hop=256, center=True)
    # (16-bit frames, as the command line hands them over: the wave files are PCM16)
    pairs = [(Pcm16Frames(np.ascontiguousarray(wavio.float_to_pcm16(mix.T))), mask)] * 32
    dfe.run(pairs)
    dtr = timed(lambda: dfe.run(pairs), 3)
    res["df_on_mask_4ch_resident"] = {
        "workload": f"the same from SAMPLES (16-bit frames; STFT included) for {len(pairs)} utterances per call, "
                    "chunks of 8 on two lanes: upload, setk_stft_batch -> setk_covar -> setk_pevd -> "
                    "setk_directional_feats on device pointers, download -- one chunk's transfers under "
                    "the other's kernels",
        "ms_per_utt": round(1e3 * dtr / len(pairs), 2), "value": round(len(pairs) * seconds / dtr, 1),
        "per_utterance_numpy_path_incl_stft_ms": None}
    # what compute_df_on_mask.py cost per utterance before: the SpectrogramReader's STFT
    # (numpy out) + the three numpy-in / numpy-out operators
    def df_old():
        sp_ = device_stft(mix, 512, 256, True, True, "hann")
        obs = np.transpose(sp_, (0, 2, 1))
        sv = B.solve_pevd(B.compute_covar(obs, mask))
        return S.directional_feats(obs, sv.T, df_pair=[(0, 1), (0, 2), (1, 3)])
    res["df_on_mask_4ch_resident"]["per_utterance_numpy_path_incl_stft_ms"] = round(1e3 * timed(df_old, 5), 2)
    # ---- factorised WPD (apply_wpd.py): resident engine against the per-utterance mirror ----
    from setk_amd.engine import BatchWpd
    wpd = BatchWpd(taps=10, delay=3, context=1, wpd_iters=3, cgmm_iters=20, frame_len=512,
                   frame_hop=256, center=True, pcm16=True)
    wmix = [Pcm16Frames(np.ascontiguousarray(wavio.float_to_pcm16(
        synth.synth_utterance(3150 + i, 4, N).T))) for i in range(4)] * 2
    wpd.run(wmix)
    dtw = timed(lambda: wpd.run(wmix), 2)
    dtm = timed(lambda: [wpd._one_by_mirror(u) for u in wmix[:2]], 1) / 2
    wpd.close()
    res["wpd_4ch_resident"] = {
        "workload": f"4-ch {seconds:g} s, 10 taps, 3 outer iterations x 20 CGMM iterations, from 16-bit "
                    f"frames to PCM_16 samples + speech mask, {len(wmix)} utterances per call "
                    "(engine.BatchWpd: one scratch block, every stage on device pointers)",
        "ms_per_utt": round(1e3 * dtw / len(wmix), 2), "value": round(len(wmix) * seconds / dtw, 1),
        "per_utterance_numpy_mirror_ms": round(1e3 * dtm, 2),
        "speedup": round(dtm / (dtw / len(wmix)), 2)}
    # ---- unfused engine: n_fft = 400, 12 channels ----
    for label, C, kw in (("mvdr_nfft400_4ch", 4, dict(frame_len=400, frame_hop=160,
                                                      round_power_of_two=False)),
                         ("mvdr_12ch", 12, dict(frame_len=512, frame_hop=256))):
        eng = BatchEnhancer(beamformer="mvdr", **kw)
        items = []
        for i in range(utts):
            mix, sp, nz = synth.synth_utterance(3200 + (i % 4), C, N, return_parts=True)
            n_fft = kw["frame_len"] if kw.get("round_power_of_two") is False else 512
            st = device_stft(np.stack([sp[0], nz[0]]), kw["frame_len"], kw["frame_hop"],
                             kw.get("round_power_of_two", True), True, "hann")
            items.append((mix, synth.irm_from_spectra(st[0], st[1]), None))
        dt = timed(lambda: eng.enhance(items), 2)
        res[label] = {"workload": f"{C}-ch {seconds:g} s x {utts} utterances, "
                                  f"STFT {kw['frame_len']}/{kw['frame_hop']}, unfused engine "
                                  "(stand-alone operators, host arrays in / out)",
                      "ms_per_utt": round(1e3 * dt / utts, 2),
                      "value": round(utts * seconds / dt, 1)}
    # ---- the general EM (K = 3, and K = 2 on a 12-channel array) and the wide WPE form ----
    from setk_amd.libs.cluster import CgmmTrainer
    for label, C, K in (("cgmm_general_k3_4ch", 4, 3), ("cgmm_general_k2_12ch", 12, 2)):
        obs = np.transpose(device_stft(synth.synth_utterance(3300 + C, C, N), 512, 256, True, True, "hann"),
                           (0, 2, 1))                                        # C x F x T
        np.random.seed(777)
        dt = timed(lambda: CgmmTrainer(obs, K).train(20), 1)
        res[label] = {"workload": f"{C}-ch {seconds:g} s, CGMM K = {K}, 20 EM iterations, general float64 "
                                  "device EM (cgmm_k.hip), numpy in / numpy out, one utterance",
                      "ms_per_utt": round(1e3 * dt, 1), "value": round(seconds / dt, 1)}
    wide = np.ascontiguousarray(np.transpose(
        device_stft(synth.synth_utterance(3400, 16, N), 512, 128, True, True, "hann"), (2, 0, 1)))
    dt = timed(lambda: W.wpe(wide, taps=10, delay=3, context=1, num_iters=3), 1)
    res["wpe_16ch_10taps_wide"] = {
        "workload": f"16-ch {seconds:g} s, hop 128, 10 taps x 3 iterations (NK = 160: R in global scratch), "
                    "numpy in / numpy out, one utterance",
        "ms_per_utt": round(1e3 * dt, 1), "value": round(seconds / dt, 1)}
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=10.0)
    a = ap.parse_args()
    print(json.dumps(run(a.utts, a.seconds)))
