#!/usr/bin/env python
"""
Directional features from TF-masks on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/compute_df_on_mask.py`` (same positional arguments,
options, defaults and Kaldi archive output).  Where the reference walks a SpectrogramReader
one utterance at a time through numpy, this front end feeds batches of WAVE SAMPLES to
``engine.BatchDirectionalFeatures``: one upload per batch, the spectrograms, speech
covariances and steer vectors stay on the device (setk_stft_batch -> setk_covar -> setk_pevd
-> setk_directional_feats on device pointers), one download of the T x F feature maps.
The archive keeps the table's order; utterances without a mask are reported and left out.
Under ``torchrun`` every rank would write the same archive, so the tool runs on one rank only.
"""
import argparse

import numpy as np

from setk_amd import _ffi
from setk_amd.engine import BatchDirectionalFeatures, Pcm16Frames
from setk_amd.libs.data_handler import ArchiveWriter, NumpyReader, ScriptReader, WaveReader
from setk_amd.libs.opts import StftParser
from setk_amd.libs.utils import get_logger

logger = get_logger(__name__)


def parse_pairs(text):
    """"0,1;2,3" -> [(0, 1), (2, 3)]."""
    pairs = []
    for item in text.split(";"):
        item = item.strip()
        if item:
            i, j = item.split(",")
            pairs.append((int(i), int(j)))
    if not pairs:
        raise RuntimeError(f"Bad configurations with --pair {text}")
    return pairs


def run(args):
    pairs = parse_pairs(args.df_pair)
    waves = WaveReader(args.wav_scp)
    masks = (NumpyReader if args.fmt == "numpy" else ScriptReader)(args.mask_scp)
    # device buffers, streams and copies come from the library unless a table is wider than the
    # fused STFT (more than 8 channels go through torch tensors)
    _ffi.set_torch_free(waves.first_channels_at_most(8))
    engine = BatchDirectionalFeatures(pairs, frame_len=args.frame_len, frame_hop=args.frame_hop,
                                      center=bool(args.center), window=args.window,
                                      round_power_of_two=bool(args.round_power_of_two))
    logger.info(f"Compute directional feature with {pairs}")
    total, written, failed = len(waves), 0, 0
    with ArchiveWriter(args.dup_ark, args.scp) as ark:
        queue = []

        def drain():
            nonlocal written, failed
            if not queue:
                return
            results = engine.run([(samps, mask) for _, samps, mask in queue])
            for (key, _, _), (feats, code) in zip(queue, results):
                if code:
                    failed += 1
                    logger.error(f"Raise linalg error: {key}")
                    continue
                ark.write(key, feats)
                written += 1
                if written % 1000 == 0:
                    logger.info(f"Processed {written:d} utterance...")
            del queue[:]

        for key in waves.index_keys:
            if key not in masks:
                logger.warning(f"Missing TF-mask for utterance {key}")
                continue
            frames = waves.read_pcm16(key)
            samps = Pcm16Frames(frames) if frames is not None else waves.read(key)
            queue.append((key, samps, np.asarray(masks[key])))
            if len(queue) >= args.batch_utts:
                drain()
        drain()
    engine.close()
    logger.info(f"Processed {written:d} utterances over {total:d}")
    if failed:
        logger.warning(f"{failed:d} utterances skipped on numerical errors")


def build_parser():
    parser = argparse.ArgumentParser(
        description="Command to compute directional features for arbitrary arrays, "
        "based on estimated TF-masks",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter, parents=[StftParser.parser])
    parser.add_argument("wav_scp", type=str, help="Multi-Channel wave scripts in kaldi format")
    parser.add_argument("mask_scp", type=str,
                        help="Scripts of masks in kaldi's archive or numpy's ndarray")
    parser.add_argument("dup_ark", type=str, help="Location to dump features in kaldi's archives")
    parser.add_argument("--scp", type=str, default="",
                        help="If assigned, generate corresponding feature scripts")
    parser.add_argument("--mask-format", dest="fmt", choices=["kaldi", "numpy"], default="kaldi",
                        help="Define format of masks, in kaldi's archives or numpy's ndarray")
    parser.add_argument("--df-pair", type=str, default="0,1",
                        help="Microphone pairs for directional feature computation")
    parser.add_argument("--batch-utts", type=int, default=32,
                        help="[setk_amd] utterances per GPU batch")
    return parser


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
