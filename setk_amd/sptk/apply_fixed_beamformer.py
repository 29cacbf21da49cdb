#!/usr/bin/env python
"""
Fixed (pre-designed) beamformer on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/apply_fixed_beamformer.py`` (same
positional arguments, options, defaults, outputs {dst_dir}/{key}.wav PCM16):
weights F x M, or B x F x M with ``--beam`` giving every utterance its beam.
With the n_fft = 512 plan the utterances go through the batched
beamform + inverse-STFT kernel (setk_apply_weights_batch); other transform
sizes use the stand-alone operators.  The reference applies a single F x M
weight through the beam table and fails on it (``if beamformer:`` is true for
an object, apply_fixed_beamformer.py:41-43); here that case simply works.
"""
import argparse

import numpy as np

from setk_amd import _ffi
from setk_amd.dist import Shard
from setk_amd.engine import FixedBatchBeamformer, Pcm16Frames
from setk_amd.libs.data_handler import ScpReader, WaveReader, WaveWriter
from setk_amd.libs.opts import StftParser
from setk_amd.libs.utils import get_logger

logger = get_logger(__name__)


def run(args):
    weights = np.load(args.weights)  # F x N or B x F x N
    if weights.ndim == 2:
        weights = weights[None]
        beam_index = None
    elif weights.ndim == 3:
        if not args.beam:
            raise RuntimeError("--beam must be assigned, as there are multiple beams")
        beam_index = ScpReader(args.beam, value_processor=int)
    else:
        raise RuntimeError(f"Expect weights in shape F x M or B x F x M, got {weights.shape}")
    shard = Shard()
    device = shard.device if shard.world > 1 else None
    n_fft = 2**int(np.ceil(np.log2(args.frame_len))) if args.round_power_of_two else args.frame_len
    if n_fft == 512 and shard.torch_free_ok:
        # the batch engine brings its own buffers and stream (more than 8 channels: torch)
        _ffi.set_torch_free(weights.shape[-1] <= 8)
    engine = FixedBatchBeamformer(weights, frame_len=args.frame_len, frame_hop=args.frame_hop,
                                  center=bool(args.center),
                                  round_power_of_two=bool(args.round_power_of_two),
                                  window=args.window, pcm16=True, device=device)
    wav_reader = WaveReader(args.wav_scp)
    keys = shard.assign_by_duration(wav_reader)
    num_done = 0
    with WaveWriter(args.dst_dir) as writer:

        def flush(pending):
            if not pending:
                return 0
            outs = engine.run([(s, b) for (_, s, b) in pending])
            for (key, _, _), pcm in zip(pending, outs):
                writer.write_pcm16(key, pcm)
            return len(pending)

        pending = []
        for key in keys:
            logger.info(f"Processing utterance {key}...")
            pcm = wav_reader.read_pcm16(key)
            samps = Pcm16Frames(pcm) if pcm is not None else wav_reader.read(key)
            beam = 0 if beam_index is None else beam_index[key]
            pending.append((key, samps, beam))
            if len(pending) >= args.batch_utts:
                num_done += flush(pending)
                pending = []
        num_done += flush(pending)
    total = int(shard.sum_counts([num_done])[0])
    if shard.rank == 0:
        logger.info(f"Processed {total:d} utterances")
    shard.barrier()
    shard.close()


def build_parser():
    parser = argparse.ArgumentParser(
        description="Command to run fixed beamformer. Runing this command needs "
        "to design fixed beamformer first.",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter, parents=[StftParser.parser])
    parser.add_argument("wav_scp", type=str, help="Multi-channel wave scripts in Kaldi format")
    parser.add_argument("weights", type=str,
                        help="Fixed beamformer weights in numpy format " +
                        "(in shape F x M or B x F x M)")
    parser.add_argument("dst_dir", type=str, help="Location to dump the enhanced audio")
    parser.add_argument("--beam", type=str, default="",
                        help="Beam index to use in beamformer weights (in shape B x F x M)")
    parser.add_argument("--batch-utts", type=int, default=64,
                        help="[setk_amd] utterances per GPU batch")
    return parser


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
