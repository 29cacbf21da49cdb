#!/usr/bin/env python
"""
GWPE dereverberation on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/apply_wpe.py`` (same positional arguments,
options and defaults, :86-135; multi-channel PCM_16 wav out): STFT, the WPE
iterations (setk_wpe_batch, fp64, --batch-utts utterances per launch) and the
inverse STFT of every channel run on the GPU.  ``--nara-wpe true`` is refused: that third-party package is not part of
this path.
"""
import argparse

import numpy as np

from .._ffi import SetkUnsupported
from setk_amd.dist import Shard
from setk_amd.libs.data_handler import SpectrogramReader, WaveWriter
from setk_amd.libs.opts import StftParser, strtobool
from setk_amd.libs.utils import get_logger, inverse_stft
from setk_amd.libs.wpe import wpe_batch

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

        def flush(pending):
            """--batch-utts utterances (grouped by channel count) per setk_wpe_batch call:
            one launch per iteration over every (bin, utterance)."""
            done = 0
            groups = {}
            for key, rev in pending:
                groups.setdefault(rev.shape[1], []).append((key, rev))
            for _, items in groups.items():
                try:
                    outs = wpe_batch([r for _, r in items], num_iters=args.num_iters,
                                     context=args.context, taps=args.taps, delay=args.delay)
                except SetkUnsupported as e:
                    # a shape beyond the device kernels' limits (channels x taps): skip the
                    # utterances like a numerical failure instead of ending the run
                    for key, _ in items:
                        logger.warning(f"{key}: skipped, {e}")
                    continue
                for (key, _), dereverb in zip(items, outs):
                    if dereverb is None:
                        logger.warning(f"{key}: Failed cause LinAlgError in wpe")
                        continue
                    dereverb = np.transpose(dereverb, (1, 2, 0))  # F x N x T => N x T x F
                    samps = np.stack([inverse_stft(spectra, **stft_kwargs) for spectra in dereverb])
                    writer.write(key, samps)
                    done += 1
            return done

        pending = []
        for key in shard.assign_by_duration(reader):
            reverbed = reader[key]
            logger.info(f"Processing utt {key}...")
            if reverbed.ndim == 2:
                reverbed = reverbed[None, ...]
            pending.append((key, np.transpose(reverbed, (2, 0, 1))))  # N x T x F => F x N x T
            if len(pending) >= max(1, args.batch_utts):
                num_done += flush(pending)
                pending = []
        num_done += flush(pending)
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
    parser.add_argument("--batch-utts", type=int, default=16,
                        help="[setk_amd] utterances per batched WPE call")
    return parser


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
