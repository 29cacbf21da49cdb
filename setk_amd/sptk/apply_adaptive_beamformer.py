#!/usr/bin/env python
# coding=utf-8
"""
Do mvdr/gevd/pmwf/mpdr adaptive beamforming on the MI355X.

Drop-in for funcwj/setk ``scripts/sptk/apply_adaptive_beamformer.py``: same
positional arguments, same options and defaults (:183-258), same log lines and
the same outputs ({dst_dir}/{key}.wav, mono PCM_16).  The compute loop
(:130-178) is replaced by the batched HIP engine (setk_amd.engine): utterances
are decoded once, grouped into batches, enhanced on the GPU and written back.

Differences, all deliberate (DESIGN.md "Reference bugs"):
  * ``--itf-mask`` really opens the interferer table (the reference opens
    tgt_mask again, :86-87)
  * the block-online mode works (the reference raises TypeError, :42-45)
  * under ``torchrun`` (WORLD_SIZE > 1) every rank takes its share of the
    utterances -- the replacement for ``run.pl JOB=1:nj`` sharding
  * extra options ``--batch-utts`` / ``--device`` / ``--device-ingest`` (defaults
    keep the reference behaviour)
"""
import argparse
import os
import math

import numpy as np

from setk_amd import _ffi
from setk_amd.dist import Shard
from setk_amd.engine import BatchEnhancer, Pcm16Frames, compute_vad_masks
from setk_amd.libs import wavio
from setk_amd.libs.beamformer import OnlineGevdBeamformer, OnlineMvdrBeamformer
from setk_amd.libs.data_handler import (NumpyReader, ScriptReader, SpectrogramReader,
                                        WaveReader, WaveWriter)
from setk_amd.libs.opts import StftParser, strtobool
from setk_amd.libs.utils import get_logger, inverse_stft, nextpow2

logger = get_logger(__name__)
beamformers = ["mvdr", "mpdr", "mpdr-whiten", "gevd", "pmwf-0", "pmwf-1"]

_OPTIONS = (
    # flags, kwargs  (names, defaults and help follow the reference CLI)
    (("--itf-mask",), dict(type=str, default="",
                           help="Scripts of interfering masks in kaldi's archive or numpy's ndarray")),
    (("--mask-format",), dict(dest="fmt", choices=["kaldi", "numpy"], default="kaldi",
                              help="Define format of masks, kaldi's archives or numpy's ndarray")),
    (("--beamformer",), dict(type=str, default="mvdr", choices=beamformers,
                             help="Type of adaptive beamformer to apply")),
    (("--pmwf-ref",), dict(type=int, default=-1, help="Reference channel for PMWF beamformer")),
    (("--sr",), dict(type=int, default=16000, help="Sample rate of the waveform")),
    (("--ban",), dict(type=strtobool, default=False,
                      help="Do Blind Analytical Normalization (BAN) or not")),
    (("--rank1-appro",), dict(type=str, default="", choices=["", "none", "eig", "gev"],
                              help="Weather to use rank1 approximation in PMWF")),
    (("--post-masking",), dict(dest="mask", type=strtobool, default=False,
                               help="Masking enhanced spectrogram after beamforming or not")),
    (("--vad-proportion",), dict(type=float, default=1,
                                 help="Energy proportion to filter silence masks [0.5, 1]")),
    (("--online.alpha",), dict(default=0.8, dest="alpha", type=float,
                               help="Remember coefficient when updating covariance matrix")),
    (("--online.chunk-size",), dict(default=-1, type=int, dest="chunk_size",
                                    help="If >= 64, using online beamformer instead")),
    (("--online.channels",), dict(default=4, type=int, dest="channels",
                                  help="Number of channels available")),
    (("--strict-reference",), dict(default=False, type=strtobool,
                                   help="[setk_amd] skip the utterances whose noise covariance is "
                                        "structurally singular, as the reference's numpy.linalg.solve does "
                                        "(an exactly zero complex64 LU pivot: duplicated / silent channel, "
                                        "all-zero covariance; the decision per utterance is reproduced on the "
                                        "tested table tests/golden/ref_skipset.json, individual bins can differ "
                                        "from LAPACK's by rounding); default: such covariances are regularised "
                                        "and the utterance is enhanced (INTEGRATION.md)")),
    (("--batch-utts",), dict(default=32, type=int,
                             help="[setk_amd] utterances enhanced per GPU batch (at most)")),
    (("--batch-mb",), dict(default=80, type=int,
                           help="[setk_amd] streaming pipeline: a batch also closes at this many MB of "
                                "input (page-locked slab size; 0 = by --batch-utts alone)")),
    (("--device",), dict(default=-1, type=int,
                         help="[setk_amd] GPU ordinal (default: LOCAL_RANK or 0)")),
    (("--device-ingest",), dict(default=True, type=lambda v: str(v).lower() in ("true", "1", "yes"),
                                help="[setk_amd] upload 16-bit PCM files as stored and "
                                     "convert on the GPU (same samples as the host decode)")),
    (("--pipeline",), dict(default=True, type=lambda v: str(v).lower() in ("true", "1", "yes"),
                           help="[setk_amd] streaming host pipeline (pinned staging, read / "
                                "H2D / compute / D2H / write overlapped); false: one batch "
                                "at a time")),
    (("--zero-copy",), dict(default=False, type=lambda v: str(v).lower() in ("true", "1", "yes"),
                            help="[setk_amd] DMA wave / mask payloads straight out of the page "
                                 "cache (mmap + hipHostRegister) instead of staging them through "
                                 "page-locked slabs (measured slower inside the pipeline: "
                                 "profiles/r02l_e2e_zero_copy_sweep.txt)")),
    (("--h2d",), dict(default="batch", type=str, choices=["payload", "batch"],
                      help="[setk_amd] host-to-device copies of the pipeline: ONE per batch from "
                           "the page-locked slab (default: ~15 %% faster on first-read inputs, "
                           "profiles/r04_e2e_h2d_ab.txt), or one per payload as soon as it is read")),
    (("--pipeline-depth",), dict(default=3, type=int,
                                 help="[setk_amd] batches in flight in the host pipeline")),
    (("--read-threads",), dict(default=0, type=int,
                               help="[setk_amd] file reader threads (0: half the cores, <= 12)")),
    (("--numa",), dict(default="auto", type=str,
                       help="[setk_amd] host placement: auto = reader / writer threads and the pinned "
                            "staging on the NUMA node of this rank's GPU; off; or a node number")),
    (("--skip-existing",), dict(default=False, type=lambda v: str(v).lower() in ("true", "1", "yes"),
                                help="[setk_amd] resume: utterances whose {dst_dir}/{key}.wav already "
                                     "exists (non-empty) are not enhanced again")),
    (("--requeue",), dict(default=False, type=lambda v: str(v).lower() in ("true", "1", "yes"),
                          help="[setk_amd] with --skip-existing: deal only the MISSING utterances over the "
                               "ranks (a re-launch after a rank died shares that rank's leftovers among "
                               "all of them; what `python -m setk_amd.launch` passes on its retries)")),
    (("--profile",), dict(default="", type=str,
                          help="[setk_amd] write a JSON run summary (wall clock from the first "
                               "scp read to the last wav close, stage times, bytes) here")),
)


def build_parser():
    parser = argparse.ArgumentParser(
        description="Command to run adaptive(mvdr/gevd/pmwf) beamformer",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter, parents=[StftParser.parser])
    parser.add_argument("wav_scp", type=str, help="Multi-channel wave scripts in kaldi format")
    parser.add_argument("tgt_mask", type=str,
                        help="Scripts of target masks in kaldi's archive or numpy's ndarray")
    parser.add_argument("dst_dir", type=str, help="Location to dump enhanced wave files")
    for flags, kw in _OPTIONS:
        parser.add_argument(*flags, **kw)
    return parser


def do_online_beamform(beamformer, speech_mask, interf_mask, stft_mat, args):
    """Chunked beamforming with recursive covariance updates (:25-47).
    speech_mask T x F, stft_mat N x F x T -> F x T."""
    chunk = args.chunk_size
    beamformer.reset_stats(args.alpha)
    out = []
    for c in range(math.ceil(stft_mat.shape[-1] / chunk)):
        sl = slice(chunk * c, chunk * (c + 1))
        mask_n = None if interf_mask is None else interf_mask[sl]
        out.append(beamformer.run(speech_mask[sl], np.ascontiguousarray(stft_mat[:, :, sl]),
                                  mask_n=mask_n, ban=args.ban))
    return np.hstack(out)


def _complete_wav(path):
    """True if `path` is a WAV file whose size is what its RIFF / data chunk headers say (the
    writer renames finished files into place; this also rejects files truncated by other
    means, e.g. a full disk under an older version)."""
    try:
        size = os.path.getsize(path)
        if size <= 44:
            return False
        with open(path, "rb") as f:
            head = f.read(44)
    except OSError:
        return False
    if len(head) < 44 or head[:4] != b"RIFF" or head[8:12] != b"WAVE" or head[36:40] != b"data":
        return False
    riff = int.from_bytes(head[4:8], "little")
    data = int.from_bytes(head[40:44], "little")
    return riff + 8 == size and data + 44 == size


def _arm_fault_injection(shard, writer):
    """Test hook of the re-queue path (tests/test_host_asan.py): SETK_FAULT_INJECT=
    "<rank>:<n>[:<attempt>]" makes that rank die WITHOUT any clean-up -- as a killed process
    or a lost GPU would -- after its n-th wave file (in attempt <attempt> of
    `python -m setk_amd.launch`, default 0).  Only with SETK_TESTING=1; unset: nothing happens."""
    spec = os.environ.get("SETK_FAULT_INJECT", "")
    if not spec or os.environ.get("SETK_TESTING") != "1":
        return  # (a test hook: inert in a production environment even if the variable leaks in)
    parts = spec.split(":")
    rank, after = int(parts[0]), int(parts[1])
    attempt = int(parts[2]) if len(parts) > 2 else 0
    if rank != shard.rank or attempt != int(os.environ.get("SETK_LAUNCH_ATTEMPT", "0")):
        return
    count = [0]

    def wrap(inner):
        def write(*a, **k):
            r = inner(*a, **k)
            count[0] += 1
            if count[0] >= after:
                logger.error(f"SETK_FAULT_INJECT: rank {shard.rank} dies after {count[0]} files")
                os._exit(17)
            return r
        return write
    writer.record = wrap(writer.record)   # every finished file passes through record()


def _deal(args, shard, reader):
    """This rank's utterances.  Default: the deal is computed on the FULL table (every rank keeps
    the utterances it had) and --skip-existing then drops what is already there.  --requeue
    (a re-launch after a failure): the missing utterances are found first -- by every rank, from
    the same directory listing: nobody writes before the barrier below -- and only they are dealt,
    so the leftovers of a rank that died are shared by all ranks instead of waiting for it."""
    if getattr(args, "skip_existing", False) and getattr(args, "requeue", False) and shard.world > 1:
        left = _drop_existing(args, list(reader.index_keys))
        shard.barrier()
        return shard.assign_by_duration(reader, keys=left)
    return _drop_existing(args, shard.assign_by_duration(reader))


def _drop_existing(args, keys):
    """--skip-existing: a re-run after an interruption (or a re-queued shard of a failed
    rank) only does what is missing.  The sharding above is computed on the full table, so
    every rank keeps the utterances it had."""
    if not getattr(args, "skip_existing", False):
        return keys
    left = []
    for k in keys:
        path = os.path.join(args.dst_dir, f"{k}.wav")
        done = _complete_wav(path)
        if not done:
            left.append(k)
    if len(left) != len(keys):
        logger.info(f"--skip-existing: {len(keys) - len(left)} of {len(keys)} utterances already "
                    f"in {args.dst_dir}")
    return left


def run_online(args, shard):
    stft_kwargs = dict(frame_len=args.frame_len, frame_hop=args.frame_hop, window=args.window,
                       center=args.center, transpose=False)
    reader = SpectrogramReader(args.wav_scp, round_power_of_two=args.round_power_of_two,
                               **stft_kwargs)
    MaskReader = {"numpy": NumpyReader, "kaldi": ScriptReader}[args.fmt]
    tgt = MaskReader(args.tgt_mask)
    itf = MaskReader(args.itf_mask) if args.itf_mask else None
    num_bins = nextpow2(args.frame_len) // 2 + 1
    cls = {"mvdr": OnlineMvdrBeamformer, "gevd": OnlineGevdBeamformer}.get(args.beamformer)
    if cls is None:
        raise KeyError(args.beamformer)
    beamformer = cls(num_bins, args.channels, args.alpha)
    logger.info(f"Using online {args.beamformer} beamformer, chunk size = {args.chunk_size:d}")
    num_done = 0
    keys = _deal(args, shard, reader)
    with WaveWriter(args.dst_dir, sr=args.sr) as writer:
        for key in keys:
            if key not in tgt:
                continue
            stft_mat = reader[key]
            power, norm = reader.power(key), reader.maxabs(key)
            logger.info(f"Processing utterance {key}, " +
                        f"signal power {10 * np.log10(power + 1e-5):.2f}...")
            speech_mask = tgt[key]
            interf_mask = None
            if itf is None:
                speech_mask = np.minimum(speech_mask, 1)
            else:
                interf_mask = itf[key]
            F = stft_mat.shape[1]
            if speech_mask.shape[0] == F and speech_mask.shape[1] != F:
                speech_mask = np.transpose(speech_mask)
                if interf_mask is not None:
                    interf_mask = np.transpose(interf_mask)
            if 0.5 < args.vad_proportion < 1:
                vad_mask, N = compute_vad_masks(stft_mat[0], args.vad_proportion)
                logger.info(f"Filtering {N} TF-masks...")
                speech_mask = np.where(vad_mask, 1.0e-4, speech_mask)
                if interf_mask is not None:
                    interf_mask = np.where(vad_mask, 1.0e-4, interf_mask)
            try:
                stft_enh = do_online_beamform(beamformer, speech_mask, interf_mask, stft_mat, args)
            except np.linalg.LinAlgError:
                logger.error(f"Raise linalg error: {key}")
                continue
            if args.mask:
                stft_enh = stft_enh * np.transpose(speech_mask)
            samps = inverse_stft(stft_enh, norm=norm, **stft_kwargs)
            writer.write(key, samps)
            num_done += 1
    return num_done, len(reader)


def _fast_path_ok(args):
    """The streaming pipeline drives the fused n_fft = 512 kernels; the other
    configurations go through BatchEnhancer.enhance one batch at a time."""
    n_fft = nextpow2(args.frame_len) if args.round_power_of_two else args.frame_len
    if n_fft != 512 or not args.pipeline:
        return False  # (more than 8 channels: the pipeline hands those batches to the engine)
    if 0.5 < args.vad_proportion < 1:
        return False  # the VAD threshold is a host-side sort over |X_0|
    if args.itf_mask and args.beamformer == "mpdr-whiten":
        return False
    return True


def _since_process_start():
    """Seconds since this process was created (diagnostic: the `marks` of --profile)."""
    import time
    try:
        with open("/proc/self/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        return time.clock_gettime(time.CLOCK_BOOTTIME) - ticks / os.sysconf("SC_CLK_TCK")
    except Exception:  # noqa: BLE001
        return None


def run_offline(args, shard):
    import time
    t_start = time.perf_counter()
    marks = {"process_start_to_first_scp_read": _since_process_start()}
    args._marks, args._t_start = marks, t_start

    def mark(name):
        marks[name] = round(time.perf_counter() - t_start, 4)
    wav_reader = WaveReader(args.wav_scp, sr=args.sr)
    MaskReader = {"numpy": NumpyReader, "kaldi": ScriptReader}[args.fmt]
    tgt = MaskReader(args.tgt_mask)
    itf = MaskReader(args.itf_mask) if args.itf_mask else None
    if itf is not None:
        logger.info(f"Using interfering masks from {args.itf_mask}")
    logger.info(f"Using offline {args.beamformer} beamformer")
    device = None if args.device < 0 else args.device
    if device is None and shard.world > 1:
        device = shard.device
    if _fast_path_ok(args) and shard.torch_free_ok:
        # the streaming pipeline gets its buffers, streams and events from the library: a
        # single-process run never imports torch (0.9 s of a 2 s run on 192 utterances)
        # (more than 8 channels run the unfused engine through torch tensors)
        _ffi.set_torch_free(wav_reader.first_channels_at_most(8))
    engine = BatchEnhancer(beamformer=args.beamformer, frame_len=args.frame_len,
                           frame_hop=args.frame_hop, center=bool(args.center),
                           round_power_of_two=bool(args.round_power_of_two), window=args.window,
                           ban=bool(args.ban), pmwf_ref=args.pmwf_ref,
                           rank1_appro=args.rank1_appro, post_mask=bool(args.mask),
                           vad_proportion=args.vad_proportion, pcm16=True, device=device,
                           strict_reference=bool(getattr(args, "strict_reference", False)))
    mark("tables_read_engine_built")   # (the engine's first use of the library initialises HIP)
    # before any thread pool or pinned slab exists: they inherit the placement
    from setk_amd import numa
    placement = numa.bind(engine.ctx, getattr(args, "numa", "auto"))
    if placement.get("bound"):
        logger.info(f"rank {shard.rank}: bound to NUMA node {placement['node']} "
                    f"({placement['cpus']} CPUs, GPU {placement.get('pci_bus_id')})")
    keys = _deal(args, shard, wav_reader)
    mark("numa_bound_keys_dealt")
    summary = dict(mode="batch", utts=0, rank=shard.rank, world=shard.world,
                   assigned_samples=shard.assigned_weight, numa=placement, marks=marks)
    num_done = 0
    with WaveWriter(args.dst_dir, sr=args.sr) as writer:
        _arm_fault_injection(shard, writer)
        if _fast_path_ok(args):
            num_done, stats = _run_pipeline(args, engine, writer, wav_reader, tgt, itf, keys)
            summary.update(mode="pipeline", stages=stats)
        else:
            num_done = _run_batches(args, engine, writer, wav_reader, tgt, itf, keys)
    mark("last_wav_closed")
    summary["wall_s"] = time.perf_counter() - t_start
    summary["utts"] = num_done
    logger.info(f"rank {shard.rank}: {num_done} utterances in {summary['wall_s']:.2f} s "
                "(first scp read to last wav close)")
    if args.profile:
        import json
        import sys
        if shard.world > 1:
            shard.barrier()  # (connects the ranks if nothing has yet: the record names the backend)
        summary["dist_backend"] = shard.backend
        summary["torch_loaded"] = "torch" in sys.modules
        path = args.profile if shard.world == 1 else f"{args.profile}.rank{shard.rank}"
        with open(path, "w") as f:
            json.dump(summary, f, indent=1)
    return num_done, len(wav_reader)


def _run_pipeline(args, engine, writer, wav_reader, tgt, itf, keys):
    """Streaming path: setk_amd.pipeline.StreamPipeline."""
    import sys
    import threading
    from setk_amd.pipeline import OpenFiles, StreamPipeline, mask_source, wav_source
    # ~18 threads share the interpreter lock here, and this (planning) thread never blocks: with
    # the default 5 ms switch interval a reader that comes back from its copy waits up to 5 ms
    # to get the lock for the few microseconds of bookkeeping between two payloads
    sys.setswitchinterval(float(os.environ.get("SETK_SWITCH_INTERVAL", "0.0005")))
    files = OpenFiles()
    lock = threading.Lock()

    def announce(key, power):
        logger.info(f"Processing utterance {key}, " +
                    f"signal power {10 * np.log10(power + 1e-5):.2f}...")

    def sink(key, pcm, status, error):
        if error is not None:
            raise error
        if status != 0:
            # the reference's np.linalg.LinAlgError branch (:170-172)
            logger.error(f"Raise linalg error: {key}")
            return False
        dst = writer.file_for(key, ".wav")
        wavio.write_pcm16(str(dst), pcm, writer.sr)
        with lock:
            writer.record(key, dst)
        return True

    pipe = StreamPipeline(engine, sink, announce=announce, batch_utts=args.batch_utts, batch_mb=args.batch_mb,
                          depth=args.pipeline_depth, read_threads=args.read_threads or None,
                          zero_copy=args.zero_copy, h2d=args.h2d)
    import time
    marks, t_start = getattr(args, "_marks", {}), getattr(args, "_t_start", time.perf_counter())
    marks["pipeline_built"] = round(time.perf_counter() - t_start, 4)
    wide = []  # more than 8 channels: stand-alone operators, after the pipeline drained
    try:
        for key in keys:
            if key not in tgt:
                continue
            audio = wav_source(wav_reader, key, files) if args.device_ingest else None
            if audio is None:
                samps = wav_reader.read(key)
                audio = samps[None] if samps.ndim == 1 else samps
            if (audio[1] if isinstance(audio, tuple) else audio.shape[0]) > 8:
                wide.append(key)
                continue
            pipe.submit(key, audio, mask_source(tgt, key, files, engine.num_bins),
                        None if itf is None else mask_source(itf, key, files, engine.num_bins))
        marks["all_submitted"] = round(time.perf_counter() - t_start, 4)
    finally:
        num_done, stats = pipe.close()
        marks["pipeline_closed_slabs_released"] = round(time.perf_counter() - t_start, 4)
        files.close()
    if wide:
        num_done += _run_batches(args, engine, writer, wav_reader, tgt, itf, wide)
    return num_done, stats


def _run_batches(args, engine, writer, wav_reader, tgt, itf, keys):
    """One batch at a time (n_fft != 512, VAD filtering, --pipeline false)."""
    num_done = 0

    def flush(pending):
        done = 0
        if not pending:
            return done
        results = engine.enhance([(s, m, i) for (_, s, m, i) in pending])
        for (key, _, _, _), (pcm, status) in zip(pending, results):
            if status != 0:
                # the reference's np.linalg.LinAlgError branch (:170-172)
                logger.error(f"Raise linalg error: {key}")
                continue
            writer.write_pcm16(key, pcm)
            done += 1
        return done

    pending = []
    for key in keys:
        if key not in tgt:
            continue
        pcm = wav_reader.read_pcm16(key) if args.device_ingest else None
        if pcm is not None:
            # 16-bit PCM file: upload the frames as stored, convert on the device
            ch0 = pcm[:, 0].astype(np.float32) / np.float32(32768.0)
            samps = Pcm16Frames(pcm)
        else:
            samps = wav_reader.read(key)
            if samps.ndim == 1:
                samps = samps[None]
            ch0 = samps[0]
        power = np.linalg.norm(ch0, 2)**2 / ch0.size
        logger.info(f"Processing utterance {key}, " +
                    f"signal power {10 * np.log10(power + 1e-5):.2f}...")
        pending.append((key, samps, tgt[key], None if itf is None else itf[key]))
        if len(pending) >= args.batch_utts:
            num_done += flush(pending)
            pending = []
    num_done += flush(pending)
    return num_done


def run(args):
    shard = Shard()
    try:
        if args.chunk_size <= 0:
            num_done, total = run_offline(args, shard)
        else:
            if args.chunk_size < 32:
                raise RuntimeError(f"Seems chunk size({args.chunk_size:.2f}) " +
                                   "too small for online beamformer")
            num_done, total = run_online(args, shard)
        shard.barrier()
        if shard.world > 1:
            num_done = int(round(shard.sum_counts([num_done])[0]))
        if shard.rank == 0:
            logger.info(f"Processed {num_done:d} utterances out of {total:d}")
    finally:
        shard.close()


def main(argv=None):
    args = build_parser().parse_args(argv)
    run(args)
