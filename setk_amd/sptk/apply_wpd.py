#!/usr/bin/env python
"""
Joint dereverberation & denoising (factorised WPD) on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/apply_wpd.py`` (same positional arguments,
options and defaults, :64-113): per utterance the WPE step, the CGMM mask
estimation and the power-weighted MVDR run on the GPU (setk_amd.libs.wpe), the
enhanced channel is written as PCM_16 wav, optionally with the speech mask.
"""
import argparse

import numpy as np

from .. import _ffi
from .._ffi import SetkUnsupported
from setk_amd.dist import Shard
from setk_amd.libs.data_handler import SpectrogramReader, WaveWriter
from setk_amd.libs.opts import StftParser, strtobool
from setk_amd.libs.utils import get_logger, inverse_stft
from setk_amd.libs.wpe import facted_wpd

logger = get_logger(__name__)


def run(args):
    stft_kwargs = {
        "frame_len": args.frame_len,
        "frame_hop": args.frame_hop,
        "window": args.window,
        "center": args.center,
        "transpose": True  # T x F
    }
    shard = Shard()
    if shard.world == 1:
        _ffi.set_torch_free()  # numpy arrays in and out of the library: nothing here needs torch
    reader = SpectrogramReader(args.wav_scp, round_power_of_two=args.round_power_of_two,
                               **stft_kwargs)
    num_done = 0
    with WaveWriter(args.dst_dir, sr=args.sr) as writer:
        for key in shard.assign_by_duration(reader):
            obs = reader[key]
            logger.info(f"Processing utt {key}...")
            if obs.ndim != 3:
                raise RuntimeError(f"Expected 3D array, but got {obs.ndim}")
            try:
                tf_mask, wpd_enh = facted_wpd(obs, wpd_iters=args.wpd_iters,
                                              cgmm_iters=args.cgmm_iters,
                                              update_alpha=args.update_alpha,
                                              context=args.context, taps=args.taps,
                                              delay=args.delay)
            except np.linalg.LinAlgError:
                logger.warning(f"{key}: Failed cause LinAlgError in wpd")
                continue
            except SetkUnsupported as e:
                # a shape beyond the device kernels' limits (channels x taps): skip the
                # utterance like a numerical failure instead of ending the run
                logger.warning(f"{key}: skipped, {e}")
                continue
            norm = reader.maxabs(key)
            samps = inverse_stft(wpd_enh, norm=norm, **stft_kwargs)
            writer.write(key, samps)
            if args.dump_mask:
                np.save(f"{args.dst_dir}/{key}", tf_mask[..., 0])
            num_done += 1
            if not num_done % 100:
                logger.info(f"Processed {num_done:d} utterances...")
    shard.barrier()
    if shard.world > 1:
        num_done = int(round(shard.sum_counts([num_done])[0]))
    if shard.rank == 0:
        logger.info(f"Processed {num_done:d} utterances over {len(reader):d}")
    shard.close()


def build_parser():
    parser = argparse.ArgumentParser(
        description="Command to do joint dereverbration & denoising algorithm "
        "(facted form of WPD)",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter, parents=[StftParser.parser])
    parser.add_argument("wav_scp", type=str, help="Multi-channel rspecifier in kaldi format")
    parser.add_argument("dst_dir", type=str, help="Location to dump enhanced audio")
    parser.add_argument("--taps", default=10, type=int, help="Value of taps used in WPE")
    parser.add_argument("--delay", default=3, type=int, help="Value of delay used in WPE")
    parser.add_argument("--context", default=1, type=int,
                        help="Context value to compute PSD matrix in WPE algorithm")
    parser.add_argument("--wpd-iters", default=3, type=int, help="Number of iterations for WPD")
    parser.add_argument("--cgmm-iters", default=20, type=int, help="Number of iterations for WPD")
    parser.add_argument("--update-alpha", type=strtobool, default=False,
                        help="If true, update alpha in M-step")
    parser.add_argument("--sr", type=int, default=16000, help="Sample rate of the input audio")
    parser.add_argument("--dump-mask", default=False, type=strtobool,
                        help="Dump cgmm mask or not")
    return parser


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
