#!/usr/bin/env python
"""
GWPE dereverberation on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/apply_wpe.py`` (same positional arguments,
options and defaults, :86-135; multi-channel PCM_16 wav out): the samples go up
once per batch, the STFT, the WPE iterations (setk_wpe_batch, fp64, --batch-utts
utterances per launch) and the inverse STFT of every channel run on the GPU
(setk_amd.engine.BatchDereverb), the waveforms come down once.  ``--nara-wpe true`` is refused: that third-party package is not part of
this path.
"""
import argparse

import numpy as np

from .. import _ffi
from .._ffi import SetkUnsupported
from setk_amd.dist import Shard
from setk_amd.engine import BatchDereverb, Pcm16Frames, _channels_and_size
from setk_amd.libs.data_handler import WaveReader, WaveWriter
from setk_amd.libs.opts import StftParser, strtobool
from setk_amd.libs.utils import get_logger

logger = get_logger(__name__)


def run(args):
    if args.nara_wpe:
        raise RuntimeError("--nara-wpe: the nara_wpe package is not available here")
    shard = Shard()
    device = shard.device if shard.world > 1 else None
    n_fft = 2**int(np.ceil(np.log2(args.frame_len))) if args.round_power_of_two else args.frame_len
    reader = WaveReader(args.wav_scp)  # 16 kHz tables like the reference (SpectrogramReader)
    if n_fft == 512 and shard.torch_free_ok:
        # the engine brings its own buffers and stream; more than 8 channels (the unfused STFT
        # path) run through torch and need it imported first
        nch = next((reader.peek_channels(k) for k in reader.index_keys), None)
        _ffi.set_torch_free(nch is not None and nch <= 8 and nch * args.taps <= 256)
    engine = BatchDereverb(taps=args.taps, delay=args.delay, context=args.context,
                           num_iters=args.num_iters, frame_len=args.frame_len,
                           frame_hop=args.frame_hop, center=bool(args.center),
                           round_power_of_two=bool(args.round_power_of_two), window=args.window,
                           device=device, pcm16=True)
    num_done = 0
    with WaveWriter(args.dst_dir, sr=args.sr) as writer:

        def flush(pending):
            """--batch-utts utterances (grouped by channel count) per engine call: samples
            up once, STFT -> WPE iterations -> inverse STFT on the device, waveforms down
            once."""
            done = 0
            groups = {}
            for key, samps in pending:
                groups.setdefault(_channels_and_size(samps)[0], []).append((key, samps))
            for _, items in groups.items():
                try:
                    seen = engine.rank_deficient_bins
                    outs = engine.run([s for _, s in items])
                    if engine.rank_deficient_bins > seen:
                        # (per batch: the utterances of one launch share the status fetch)
                        logger.warning(f"{items[0][0]} .. {items[-1][0]}: rank-deficient tap correlation in "
                                       f"{engine.rank_deficient_bins - seen} bins, columns at the noise level "
                                       "dropped (the reference's solve returns a noise-determined filter there)")
                except SetkUnsupported as e:
                    # a shape beyond the device kernels' limits (channels x taps): skip the
                    # utterances like a numerical failure instead of ending the run
                    for key, _ in items:
                        logger.warning(f"{key}: skipped, {e}")
                    continue
                for (key, _), samps in zip(items, outs):
                    if samps is None:
                        logger.warning(f"{key}: Failed cause LinAlgError in wpe")
                        continue
                    writer.write_pcm16(key, samps)  # multi-channel frames, quantised on the device
                    done += 1
            return done

        pending = []
        for key in shard.assign_by_duration(reader):
            logger.info(f"Processing utt {key}...")
            pcm = reader.read_pcm16(key)
            pending.append((key, Pcm16Frames(pcm) if pcm is not None else reader.read(key)))
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
