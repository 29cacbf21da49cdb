#!/usr/bin/env python
"""
Speech / noise mask estimation with the CGMM model on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/estimate_cgmm_masks.py`` (same
positional arguments, options, defaults, outputs {dst_dir}/{key}.npy float32
T x F -- K x T x F for more than two classes, :62-64 --, skip-if-exists behaviour :38).
--num-classes 2 runs the tuned batch kernels;
3 and 4 start from the reference's seeded random posteriors (--seed, drawn on the host from
numpy's legacy generator exactly as the reference does) and run the general device EM
(csrc/cgmm_k.hip); --solve-permu aligns the classes over frequency on the host
(libs/cluster.permu_aligner); both take the one-utterance-at-a-time path.
"""
import argparse
from pathlib import Path

import numpy as np

from setk_amd import _ffi
from setk_amd.dist import Shard
from setk_amd.engine import CgmmEstimator, Pcm16Frames
from setk_amd.libs.cluster import CgmmTrainer, permu_aligner
from setk_amd.libs.data_handler import NumpyReader, NumpyWriter, ScriptReader, SpectrogramReader
from setk_amd.libs.opts import StftParser, strtobool
from setk_amd.libs.utils import get_logger

logger = get_logger(__name__)


def build_parser():
    parser = argparse.ArgumentParser(
        description="Speech & Noise mask estimation using CGMM model",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter, parents=[StftParser.parser])
    parser.add_argument("wav_scp", type=str, help="Multi-channel wave scripts in kaldi format")
    parser.add_argument("dst_dir", type=str, help="Location to dump estimated speech masks")
    parser.add_argument("--num-iters", type=int, default=20,
                        help="Number of iterations to train CGMM parameters")
    parser.add_argument("--num-classes", type=int, default=2, help="Number of the cluster")
    parser.add_argument("--seed", type=int, default=777, help="Random seed for initialization")
    parser.add_argument("--init-mask", type=str, default="", dest="init_mask",
                        help="Initial TF-mask for cgmm initialization")
    parser.add_argument("--solve-permu", type=strtobool, default=False,
                        help="If true, solving permutation problems")
    parser.add_argument("--update-alpha", type=strtobool, default=False,
                        help="If true, update alpha in M-step")
    parser.add_argument("--mask-format", type=str, dest="fmt", default="numpy",
                        choices=["kaldi", "numpy"], help="Mask storage format")
    parser.add_argument("--batch-utts", type=int, default=32,
                        help="[setk_amd] utterances per EM launch (n_fft = 512, no --init-mask)")
    return parser


def run(args):
    # estimate_cgmm_masks.py:28 of the reference: the K > 2 start is drawn from numpy's legacy
    # global generator, seeded once per run
    np.random.seed(args.seed)
    if not 2 <= args.num_classes <= 4:
        raise _ffi.SetkUnsupported(f"--num-classes {args.num_classes}: the device EM implements 2 .. 4")
    stft_kwargs = dict(frame_len=args.frame_len, frame_hop=args.frame_hop,
                       round_power_of_two=args.round_power_of_two, window=args.window,
                       center=args.center, transpose=False)
    shard = Shard()
    n_fft = 2**int(np.ceil(np.log2(args.frame_len))) if args.round_power_of_two else args.frame_len
    if n_fft == 512 and not args.init_mask and not args.solve_permu and args.num_classes == 2:
        return run_batched(args, shard)
    reader = SpectrogramReader(args.wav_scp, **stft_kwargs)
    MaskReader = {"numpy": NumpyReader, "kaldi": ScriptReader}
    init_reader = MaskReader[args.fmt](args.init_mask) if args.init_mask else None
    num_done = 0
    with NumpyWriter(args.dst_dir) as writer:
        dst_dir = Path(args.dst_dir)
        mine = shard.assign_by_duration(reader)
        # The K > 2 start comes out of ONE generator that the reference advances utterance by
        # utterance in table order (estimate_cgmm_masks.py:28, cluster.py:429-434).  With the
        # table dealt over several ranks every rank walks the whole table and DISCARDS the draws
        # of the utterances that are not its own (K x F x T doubles each, T from the wave
        # header), so an 8-GPU run starts every utterance exactly as the one-process run does.
        # (per utterance: one whose key is missing from --init-mask still draws)
        random_start = args.num_classes != 2
        walk = list(reader.index_keys) if (random_start and shard.world > 1) else mine
        own = set(mine)
        num_bins = n_fft // 2 + 1
        # what exists is decided once, before any rank writes: the walk of every rank must skip
        # the same utterances (the reference does not draw for an utterance it skips, :38)
        existing = {key for key in walk if (dst_dir / f"{key}.npy").exists()}
        shard.barrier()
        for key in walk:
            if key in existing:
                if key in own:
                    logger.info(f"Training utterance {key} ... Skip")
                continue
            if key not in own:
                if not (init_reader and key in init_reader):
                    np.random.uniform(size=args.num_classes * num_bins * _num_frames(reader, key, args, n_fft))
                continue
            stft = reader[key]
            if stft.ndim == 2:
                stft = stft[None]
            init_mask = None
            if init_reader and key in init_reader:
                init_mask = init_reader[key]
                # T x F -> F x T; all K masks of a K > 2 model: K x T x F -> K x F x T
                # (estimate_cgmm_masks.py:50-52 of the reference)
                init_mask = np.transpose(init_mask) if init_mask.ndim == 2 else np.transpose(init_mask, (0, 2, 1))
                logger.info("Using external TF-mask to initialize cgmm")
            trainer = CgmmTrainer(stft, args.num_classes, gamma=init_mask,
                                  update_alpha=bool(args.update_alpha))
            masks = np.transpose(trainer.train(args.num_iters), (0, 2, 1))  # K x T x F
            num_done += 1
            if args.solve_permu:
                masks = permu_aligner(masks)
                logger.info("Permutation alignment done on each frequency")
            # the speech mask for two classes, every class's mask (K x T x F) otherwise (:62-64)
            writer.write(key, (masks[0] if args.num_classes == 2 else masks).astype(np.float32))
            logger.info(f"Training utterance {key} ... Done")
    shard.barrier()
    if shard.world > 1:
        num_done = int(round(shard.sum_counts([num_done])[0]))
    if shard.rank == 0:
        logger.info(f"Train {num_done:d} utterances over {len(reader):d}")
    shard.close()


def _num_frames(reader, key, args, n_fft):
    """STFT frames of an utterance from its wave header (librosa's count: 1 + N // hop with
    centring, 1 + (N - n_fft) // hop without); pipes and globs are decoded for their length."""
    n = reader.peek_nsamps(key)
    if n is None:
        samps = reader.read(key)
        n = samps.shape[-1]
    if args.center:
        return 1 + n // args.frame_hop
    return 1 + (n - n_fft) // args.frame_hop


def run_batched(args, shard):
    """Fast path: waves in, masks out, STFT + EM for a batch of utterances on the GPU."""
    from setk_amd.libs.data_handler import WaveReader
    reader = WaveReader(args.wav_scp)
    if shard.torch_free_ok:
        # CgmmEstimator.estimate brings its own buffers and stream: no torch in this process
        # (bins that fit no resident configuration -- more than 8 channels -- go through torch)
        _ffi.set_torch_free(reader.first_channels_at_most(8))
    est = CgmmEstimator(frame_len=args.frame_len, frame_hop=args.frame_hop,
                        center=bool(args.center), round_power_of_two=True, window=args.window,
                        num_iters=args.num_iters, update_alpha=bool(args.update_alpha),
                        device=shard.device if shard.world > 1 else None)
    num_done = 0
    with NumpyWriter(args.dst_dir) as writer:
        dst_dir = Path(args.dst_dir)
        pending = []

        def flush():
            nonlocal num_done
            if not pending:
                return
            masks = est.estimate([s for _, s in pending])
            for (key, _), m in zip(pending, masks):
                writer.write(key, m.astype(np.float32))
                logger.info(f"Training utterance {key} ... Done")
                num_done += 1
            pending.clear()

        for key in shard.assign_by_duration(reader):
            if (dst_dir / f"{key}.npy").exists():
                logger.info(f"Training utterance {key} ... Skip")
                continue
            # one 16-bit PCM file: its frames go up as stored and are converted on the device
            pcm = reader.read_pcm16(key)
            if pcm is not None:
                pending.append((key, Pcm16Frames(pcm)))
            else:
                samps = reader.read(key)
                pending.append((key, samps[None] if samps.ndim == 1 else samps))
            if len(pending) >= args.batch_utts:
                flush()
        flush()
    shard.barrier()
    if shard.world > 1:
        num_done = int(round(shard.sum_counts([num_done])[0]))
    if shard.rank == 0:
        logger.info(f"Train {num_done:d} utterances over {len(reader):d}")
    shard.close()


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
