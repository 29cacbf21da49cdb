#!/usr/bin/env python
"""
Joint dereverberation & denoising (factorised WPD) on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/apply_wpd.py`` (same positional arguments, options and
defaults, :64-113; outputs {dst_dir}/{key}.wav PCM_16 and, with --dump-mask, {key}.npy).  The
reference walks a SpectrogramReader and carries every stage through numpy; this front end
hands batches of WAVE SAMPLES to ``engine.BatchWpd``: one upload per batch, the spectrogram and
every intermediate of the outer iterations (WPE step, CGMM posteriors, the two covariances,
MVDR weights, enhanced spectrum) live in one device scratch block, the inverse STFT, the
renorm to max |samples| and the PCM_16 conversion run on the device, one slab comes down.
Utterances are dealt over the ranks of a ``torchrun`` launch by duration.
"""
import argparse

import numpy as np

from .. import _ffi
from setk_amd.dist import Shard
from setk_amd.engine import BatchWpd, Pcm16Frames
from setk_amd.libs.data_handler import WaveReader, WaveWriter
from setk_amd.libs.opts import StftParser, strtobool
from setk_amd.libs.utils import get_logger

logger = get_logger(__name__)


def run(args):
    shard = Shard()
    waves = WaveReader(args.wav_scp)
    if shard.torch_free_ok:
        # buffers, streams and copies come from the library (wider tables: torch tensors)
        _ffi.set_torch_free(waves.first_channels_at_most(8))
    engine = BatchWpd(taps=args.taps, delay=args.delay, context=args.context,
                      wpd_iters=args.wpd_iters, cgmm_iters=args.cgmm_iters,
                      update_alpha=bool(args.update_alpha), frame_len=args.frame_len,
                      frame_hop=args.frame_hop, center=bool(args.center), window=args.window,
                      round_power_of_two=bool(args.round_power_of_two), pcm16=True,
                      device=shard.device if shard.world > 1 else None)
    mine = shard.assign_by_duration(waves)
    done = 0
    with WaveWriter(args.dst_dir, sr=args.sr) as writer:

        def emit(group):
            """One device batch: utterances of ONE channel count."""
            nonlocal done
            try:
                results = engine.run([samps for _, samps in group])
            except _ffi.SetkUnsupported as e:
                # a shape beyond the device kernels (channels x taps): skipped like a numerical
                # failure, the run goes on
                for key, _ in group:
                    logger.warning(f"{key}: skipped, {e}")
                return
            for (key, _), res in zip(group, results):
                if res is None:
                    logger.warning(f"{key}: Failed cause LinAlgError in wpd")
                    continue
                wave, mask = res
                writer.write_pcm16(key, wave)
                if args.dump_mask:
                    np.save(f"{args.dst_dir}/{key}", mask.astype(np.float64))
                done += 1
                if done % 100 == 0:
                    logger.info(f"Processed {done:d} utterances...")

        pending = {}  # channel count -> [(key, samples)]
        for key in mine:
            logger.info(f"Processing utt {key}...")
            frames = waves.read_pcm16(key)
            samps = Pcm16Frames(frames) if frames is not None else np.atleast_2d(waves.read(key))
            group = pending.setdefault(samps.num_channels if frames is not None else samps.shape[0], [])
            group.append((key, samps))
            if len(group) >= args.batch_utts:
                emit(group)
                del group[:]
        for group in pending.values():
            if group:
                emit(group)
    if engine.rank_deficient_bins:
        logger.warning(f"{engine.rank_deficient_bins:d} frequency bins had a rank-deficient tap correlation "
                       "(columns at the noise level dropped; numpy.linalg.solve goes through on such input too)")
    engine.close()
    shard.barrier()
    total = int(round(shard.sum_counts([done])[0])) if shard.world > 1 else done
    if shard.rank == 0:
        logger.info(f"Processed {total:d} utterances over {len(waves):d}")
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
    parser.add_argument("--batch-utts", type=int, default=8,
                        help="[setk_amd] utterances per GPU batch")
    return parser


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
