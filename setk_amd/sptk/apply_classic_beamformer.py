#!/usr/bin/env python
"""
Delay-and-sum / superdirective beamformers for linear and circular arrays on
the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/apply_classic_beamformer.py`` and its two
front ends ``apply_ds_beamformer.py`` / ``apply_sd_beamformer.py`` (same positional
arguments, options, defaults, log lines, outputs {dst_dir}/{key}.wav PCM16).  The
weights are the reference's closed forms of the geometry (libs/beamformer.py:
133-212, 343-512; host float64, a few kilobytes per direction); applying them is
the batched beamform + inverse-STFT kernel (setk_apply_weights_batch, one beam per
distinct DoA of the batch, SETK_FLAG_NO_RENORM unless --normalize).  The online
mode (--chunk-len > 0, one DoA per chunk) beamforms chunk by chunk through the
stand-alone operators, as the reference does (:20-31).
"""
import argparse
import math

import numpy as np

from setk_amd import _ffi
from setk_amd.dist import Shard
from setk_amd.engine import FixedBatchBeamformer, Pcm16Frames
from setk_amd.libs.beamformer import (CircularDSBeamformer, CircularSDBeamformer,
                                      LinearDSBeamformer, LinearSDBeamformer)
from setk_amd.libs.data_handler import ScpReader, SpectrogramReader, WaveReader, WaveWriter
from setk_amd.libs.opts import StftParser, str2tuple, strtobool
from setk_amd.libs.utils import check_doa, get_logger, inverse_stft, nextpow2

logger = get_logger(__name__)
beamformers = ["ds", "sd"]


def do_online_beamform(beamformer, doa, stft_mat, args):
    chunk_size = args.chunk_len
    enh_chunks = []
    for c in range(len(doa)):
        base = chunk_size * c
        enh_chunks.append(beamformer.run(doa[c], stft_mat[:, :, base:base + chunk_size],
                                         c=args.speed, sr=args.sr))
    return np.hstack(enh_chunks)


def process_doa(doa, online):
    return list(map(float, doa)) if online else float(doa)


def parse_doa(args, online):
    if args.utt2doa:
        reader = ScpReader(args.utt2doa, value_processor=lambda doa: process_doa(doa, online),
                           num_tokens=-1 if online else 2)
        utt2doa = reader.get
        logger.info(f"Use --utt2doa={args.utt2doa} for each utterance")
    else:
        doa = process_doa(args.doa.split(",") if online else args.doa, online)
        utt2doa = lambda _: doa  # noqa: E731
        logger.info(f"Use --doa={args.doa} for all utterances")
    return utt2doa


def make_beamformer(args):
    table = {
        "ds": {"linear": lambda: LinearDSBeamformer(linear_topo=args.linear_topo),
               "circular": lambda: CircularDSBeamformer(radius=args.circular_radius,
                                                        num_arounded=args.circular_around,
                                                        center=args.circular_center)},
        "sd": {"linear": lambda: LinearSDBeamformer(linear_topo=args.linear_topo),
               "circular": lambda: CircularSDBeamformer(radius=args.circular_radius,
                                                        num_arounded=args.circular_around,
                                                        center=args.circular_center)},
    }
    return table[args.beamformer][args.geometry]()


def run_online(args, beamformer, utt2doa, shard):
    stft_kwargs = dict(frame_len=args.frame_len, frame_hop=args.frame_hop, window=args.window,
                       center=args.center, transpose=False)
    reader = SpectrogramReader(args.wav_scp, round_power_of_two=args.round_power_of_two,
                               **stft_kwargs)
    done = 0
    with WaveWriter(args.dst_dir, sr=args.sr) as writer:
        for key in shard.assign_by_duration(reader):
            stft_src = reader[key]
            doa = utt2doa(key)
            if doa is None:
                logger.info(f"Missing doa for utterance {key}")
                continue
            if not check_doa(args.geometry, doa, True):
                logger.info(f"Invalid doa {doa} for utterance {key}")
                continue
            num_chunks = math.ceil(stft_src.shape[-1] / args.chunk_len)
            if len(doa) != num_chunks:
                mn = math.ceil(stft_src.shape[-1] / len(doa))
                mx = math.floor(stft_src.shape[-1] / max(len(doa) - 1, 1))
                logger.info(f"Invalid chunk length {args.chunk_len} for utterance {key},"
                            f" expected --chunk-len from {mn} to {mx}")
                continue
            stft_enh = do_online_beamform(beamformer, doa, stft_src, args)
            norm = reader.maxabs(key) if args.normalize else None
            writer.write(key, inverse_stft(stft_enh, **stft_kwargs, norm=norm))
            done += 1
    return done, len(reader)


def run_offline(args, beamformer, utt2doa, shard):
    n_fft = nextpow2(args.frame_len) if args.round_power_of_two else args.frame_len
    num_bins = n_fft // 2 + 1
    wav_reader = WaveReader(args.wav_scp, sr=args.sr)
    device = shard.device if shard.world > 1 else None
    if n_fft == 512 and shard.torch_free_ok:
        # the batch engine brings its own buffers and stream (more than 8 channels: torch)
        _ffi.set_torch_free(wav_reader.first_channels_at_most(8))
    done = 0
    # ONE engine for the run (pinned slabs, device twin and stream are allocated once, not per
    # batch), the DS / SD weights computed once per DoA (SD: an N x N solve per bin) and the
    # engine's table replaced only when a batch brings a DoA not seen before
    wcache, order = {}, []
    state = {"engine": None, "rows": 0}
    with WaveWriter(args.dst_dir, sr=args.sr) as writer:

        def flush(pending):
            if not pending:
                return 0
            for (_, _, d) in pending:
                if d not in wcache:
                    wcache[d] = beamformer.weight(d, num_bins, c=args.speed, sr=args.sr)
                    order.append(d)
            if state["engine"] is None:
                state["engine"] = FixedBatchBeamformer(
                    np.stack([wcache[d] for d in order]), frame_len=args.frame_len,
                    frame_hop=args.frame_hop, center=bool(args.center),
                    round_power_of_two=bool(args.round_power_of_two), window=args.window, pcm16=True,
                    device=device, renorm=bool(args.normalize))
            elif state["rows"] != len(order):
                state["engine"].set_weights(np.stack([wcache[d] for d in order]))
            state["rows"] = len(order)
            outs = state["engine"].run([(s, order.index(d)) for (_, s, d) in pending])
            for (key, _, _), pcm in zip(pending, outs):
                writer.write_pcm16(key, pcm)
            return len(pending)

        pending = []
        for key in shard.assign_by_duration(wav_reader):
            doa = utt2doa(key)
            if doa is None:
                logger.info(f"Missing doa for utterance {key}")
                continue
            if not check_doa(args.geometry, doa, False):
                logger.info(f"Invalid doa {doa:.2f} for utterance {key}")
                continue
            pcm = wav_reader.read_pcm16(key)
            samps = Pcm16Frames(pcm) if pcm is not None else wav_reader.read(key)
            nch = samps.num_channels if isinstance(samps, Pcm16Frames) else \
                (1 if samps.ndim == 1 else samps.shape[0])
            if nch != beamformer.num_mics:
                raise ValueError("Shape of obs do not match with number" +
                                 f"of microphones, {beamformer.num_mics} vs {nch}")
            pending.append((key, samps, doa))
            if len(pending) >= args.batch_utts:
                done += flush(pending)
                pending = []
        try:
            done += flush(pending)
        finally:
            if state["engine"] is not None:
                state["engine"].close()
    return done, len(wav_reader)


def run(args):
    beamformer = make_beamformer(args)
    online = args.chunk_len > 0
    utt2doa = parse_doa(args, online)
    shard = Shard()
    if online:
        done, total = run_online(args, beamformer, utt2doa, shard)
    else:
        done, total = run_offline(args, beamformer, utt2doa, shard)
    done = int(shard.sum_counts([done])[0])
    if shard.rank == 0:
        logger.info(f"Processed {done} utterances over {total}")
    shard.barrier()
    shard.close()


def build_parser(description="Command to apply classic beamformer (linear & circular array).",
                 with_beamformer=True):
    parser = argparse.ArgumentParser(description=description,
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                     parents=[StftParser.parser])
    parser.add_argument("wav_scp", type=str, help="Rspecifier for multi-channel wave file")
    parser.add_argument("dst_dir", type=str, help="Directory to dump enhanced results")
    if with_beamformer:
        parser.add_argument("--beamformer", type=str, default="ds", choices=beamformers,
                            help="Type of classic beamformer to apply")
    parser.add_argument("--sr", type=int, default=16000, help="Sample rate of the input wave")
    parser.add_argument("--speed", type=float, default=343, help="Speed of sound")
    parser.add_argument("--geometry", type=str, choices=["linear", "circular"], default="linear",
                        help="Geometry of the microphone array")
    parser.add_argument("--linear-topo", type=str2tuple, default=(),
                        help="Topology of linear microphone arrays")
    parser.add_argument("--circular-around", type=int, default=6,
                        help="Number of the micriphones in circular arrays")
    parser.add_argument("--circular-radius", type=float, default=0.05,
                        help="Radius of circular array")
    parser.add_argument("--circular-center", type=strtobool, default=False,
                        help="Is there a microphone put in the center of the circular array?")
    parser.add_argument("--utt2doa", type=str, default="",
                        help="Given DoA for each utterances, in degrees")
    parser.add_argument("--doa", type=str, default="0",
                        help="DoA for all utterances if --utt2doa is not assigned")
    parser.add_argument("--normalize", type=strtobool, default=False,
                        help="Normalize stft after enhancement?")
    parser.add_argument("--chunk-len", type=int, default=-1,
                        help="Number frames per chunk (for online setups)")
    parser.add_argument("--batch-utts", type=int, default=64,
                        help="[setk_amd] utterances per GPU batch")
    return parser


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
