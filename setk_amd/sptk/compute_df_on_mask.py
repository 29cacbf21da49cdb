#!/usr/bin/env python
"""
Directional features from TF-masks on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/compute_df_on_mask.py`` (same positional
arguments, options, defaults and Kaldi archive output): per utterance
STFT -> speech covariance -> principal eigenvector -> directional features,
every stage a kernel of libsetk_hip.so through the mirrored ``libs`` API.
Under ``torchrun`` every rank would write the same archive, so the tool runs
on one rank only.
"""
import argparse

import numpy as np

from setk_amd import _ffi
from setk_amd.libs.beamformer import compute_covar, solve_pevd
from setk_amd.libs.data_handler import ArchiveWriter, NumpyReader, ScriptReader, SpectrogramReader
from setk_amd.libs.opts import StftParser
from setk_amd.libs.spatial import directional_feats
from setk_amd.libs.utils import get_logger

logger = get_logger(__name__)


def run(args):
    stft_kwargs = {
        "frame_len": args.frame_len,
        "frame_hop": args.frame_hop,
        "round_power_of_two": args.round_power_of_two,
        "window": args.window,
        "center": args.center,
        "transpose": False  # F x T
    }
    _ffi.set_torch_free()  # numpy arrays in and out of the library: nothing here needs torch
    feat_reader = SpectrogramReader(args.wav_scp, **stft_kwargs)
    mask_reader = {"numpy": NumpyReader, "kaldi": ScriptReader}[args.fmt](args.mask_scp)
    df_pair = [tuple(map(int, p.split(","))) for p in args.df_pair.split(";")]
    if not len(df_pair):
        raise RuntimeError(f"Bad configurations with --pair {args.df_pair}")
    logger.info(f"Compute directional feature with {df_pair}")

    num_done = 0
    with ArchiveWriter(args.dup_ark, args.scp) as writer:
        for key, obs in feat_reader:
            if key in mask_reader:
                speech_masks = mask_reader[key]
                _, F, _ = obs.shape
                if speech_masks.shape[0] == F:
                    speech_masks = np.transpose(speech_masks)
                speech_masks = np.minimum(speech_masks, 1)
                speech_covar = compute_covar(obs, speech_masks)  # obs: N x F x T
                sv = solve_pevd(speech_covar)
                df = directional_feats(obs, sv.T, df_pair=df_pair)
                writer.write(key, df)
                num_done += 1
                if not num_done % 1000:
                    logger.info(f"Processed {num_done:d} utterance...")
            else:
                logger.warning(f"Missing TF-mask for utterance {key}")
    logger.info(f"Processed {num_done:d} utterances over {len(feat_reader):d}")


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
    return parser


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
