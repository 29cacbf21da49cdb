#!/usr/bin/env python
"""
GWPE dereverberation on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/apply_wpe.py`` (same positional arguments,
options and defaults, :86-135; multi-channel PCM_16 wav out): STFT, the WPE
iterations (setk_wpe, fp64) and the inverse STFT of every channel run on the
GPU.  ``--nara-wpe true`` is refused: that third-party package is not part of
this path.
"""
import argparse

import numpy as np

from .._ffi import SetkUnsupported
from setk_amd.dist import Shard
from setk_amd.libs.data_handler import SpectrogramReader, WaveWriter
from setk_amd.libs.opts import StftParser, strtobool
from setk_amd.libs.utils import get_logger, inverse_stft
from setk_amd.libs.wpe import wpe

logger = get_logger(__name__)


def run(args):
    if args.nara_wpe:
        raise RuntimeError("--nara-wpe: the nara_wpe package is not available here")
    stft_kwargs = {
        "frame_len": args.frame_len,
        "frame_hop": args.frame_hop,
        "window": args.window,
        "center": args.center,
        "transpose": True  # T x F
    }
    shard = Shard()
    reader = SpectrogramReader(args.wav_scp, round_power_of_two=args.round_power_of_two,
                               **stft_kwargs)
    num_done = 0
    with WaveWriter(args.dst_dir, sr=args.sr) as writer:
        for key in shard.assign_by_duration(reader):
            reverbed = reader[key]
            logger.info(f"Processing utt {key}...")
            if reverbed.ndim == 2:
                reverbed = reverbed[None, ...]
            reverbed = np.transpose(reverbed, (2, 0, 1))  # N x T x F => F x N x T
            try:
                dereverb = wpe(reverbed, num_iters=args.num_iters, context=args.context,
                               taps=args.taps, delay=args.delay)
            except np.linalg.LinAlgError:
                logger.warning(f"{key}: Failed cause LinAlgError in wpe")
                continue
            except SetkUnsupported as e:
                # a shape beyond the device kernels' limits (channels x taps): skip the
                # utterance like a numerical failure instead of ending the run
                logger.warning(f"{key}: skipped, {e}")
                continue
            dereverb = np.transpose(dereverb, (1, 2, 0))  # F x N x T => N x T x F
            samps = np.stack([inverse_stft(spectra, **stft_kwargs) for spectra in dereverb])
            writer.write(key, samps)
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
        description="Command to do GWPE dereverbration algorithm (recommended "
        "configuration: 512/128/blackman)",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter, parents=[StftParser.parser])
    parser.add_argument("wav_scp", type=str, help="Multi-channel rspecifier in kaldi format")
    parser.add_argument("dst_dir", type=str, help="Location to dump dereverbrated files")
    parser.add_argument("--taps", default=10, type=int, help="Value of taps used in GWPE algorithm")
    parser.add_argument("--delay", default=3, type=int,
                        help="Value of delay used in GWPE algorithm")
    parser.add_argument("--context", default=1, dest="context", type=int,
                        help="Context value to compute PSD matrix in GWPE algorithm")
    parser.add_argument("--num-iters", default=3, type=int,
                        help="Number of iterations to step in GWPE")
    parser.add_argument("--sample-rate", type=int, default=16000, dest="sr",
                        help="Waveform data sample rate")
    parser.add_argument("--nara-wpe", type=strtobool, default=False, help="Use nara-wpe package")
    return parser


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
