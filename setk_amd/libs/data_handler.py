"""
Readers / writers of the beamformer CLI boundary (the subset of
scripts/sptk/libs/data_handler.py on the hot path): Kaldi style scp tables of
wave files, Kaldi archives and numpy masks in, wave files out.

Behaviour kept from the reference (data_handler.py line numbers):
  * scp values may be a path, a glob of per-channel files (sorted), an
    ``ark_path:offset`` pair, a shell pipe ending in ``|``; the table itself may
    be ``-`` (stdin) or a pipe (:76-106, 139-170, 326-393)
  * duplicate keys -> ValueError, malformed lines -> RuntimeError (:157-168)
  * Reader protocol: __len__, __contains__, __iter__ -> (key, object),
    __getitem__ by key or integer position (:173-236)
  * WaveWriter writes {dir}/{key}.wav as PCM_16 and creates the directory
    (:283-285, 590-605)

One deliberate difference: WaveReader decodes a file once and keeps the last
utterance, where the reference decodes the same file three times per utterance
(_load, power, maxabs; apply_adaptive_beamformer.py:130-133).
"""
import codecs
import glob
import os
import subprocess
import sys
import threading
import warnings
import _thread
from io import BytesIO, TextIOWrapper
from pathlib import Path

from . import wavio

import numpy as np

from . import kaldi_io
from .utils import filekey, forward_stft, read_wav, write_wav, device_stft

__all__ = [
    "ArchiveReader", "ArchiveWriter", "WaveWriter", "NumpyWriter", "SpectrogramReader",
    "ScriptReader", "WaveReader", "NumpyReader", "ScpReader", "Reader", "Writer", "parse_scps"
]


def run_command(command, wait=True):
    proc = subprocess.Popen(command, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if not wait:
        return proc
    out, err = proc.communicate()
    if proc.returncode != 0:
        raise Exception(f'There was an error while running the command "{command}":\n'
                        f"{err.decode()}\n")
    return out, err


def _pipe_open(command, mode, background=True):
    if mode not in ("rb", "r"):
        raise RuntimeError("Now only support input from pipe")
    proc = subprocess.Popen(command, shell=True, stdout=subprocess.PIPE)

    def waiter():
        proc.wait()
        if proc.returncode != 0:
            warnings.warn(f'Command "{command}" exited with status {proc.returncode}')
            _thread.interrupt_main()

    if background:
        threading.Thread(target=waiter, daemon=True).start()
    else:
        waiter()
    return proc.stdout


_MODES = ("r", "w", "rb", "wb")


class ext_open(object):
    """``with ext_open(spec, mode) as fd``: `spec` is a path, ``-`` (the process's
    stdin/stdout) or ``command |`` (read side of a shell pipe).  Ordinary files
    are closed on exit, the standard streams and pipes are left alone
    (reference behaviour: data_handler.py:76-138)."""

    def __init__(self, fname, mode):
        if mode not in _MODES:
            raise ValueError(f"Unknown open mode: {mode}")
        self.spec = fname.strip() if fname else fname
        self.mode = mode
        self.fd = None
        self._owned = False

    def _open(self):
        spec, mode = self.spec, self.mode
        binary, writing = mode.endswith("b"), mode.startswith("w")
        if not spec:
            return None
        if spec == "-":
            std = sys.stdout if writing else sys.stdin
            return std.buffer if binary else std
        if spec.endswith("|"):
            pipe = _pipe_open(spec[:-1], mode, background=binary)
            return pipe if binary else TextIOWrapper(pipe)
        if not writing and not os.path.exists(spec):
            raise FileNotFoundError(f'Could not find common file: "{spec}"')
        self._owned = True
        return open(spec, mode) if binary else codecs.open(spec, mode, encoding="utf-8")

    def open(self):
        self.fd = self._open()
        return self.fd

    def close(self):
        if self._owned and self.fd is not None:
            self.fd.close()
        self.fd, self._owned = None, False

    def __enter__(self):
        return self.open()

    def __exit__(self, *exc):
        self.close()


def parse_scps(scp_path, value_processor=lambda x: x, num_tokens=2, restrict=True):
    """Kaldi script file -> ordered dict key -> value."""
    table = {}
    with ext_open(scp_path, "r") as f:
        for lineno, raw in enumerate(f, start=1):
            toks = raw.strip().split()
            if not toks:
                raise RuntimeError(f"For {scp_path}, format error in line[{lineno:d}]: {raw}")
            if toks[-1] == "|":
                key, value = toks[0], " ".join(toks[1:])
            else:
                if (num_tokens >= 2 and len(toks) != num_tokens) or (restrict and len(toks) < 2):
                    raise RuntimeError(f"For {scp_path}, format error in " +
                                       f"line[{lineno:d}]: {raw}")
                key, value = (toks[0], toks[1]) if num_tokens == 2 else (toks[0], toks[1:])
            if key in table:
                raise ValueError(f"Duplicated key '{key}' exists in {scp_path}")
            table[key] = value_processor(value)
    return table


class Reader(object):
    """key -> object table with sequential and random access."""

    def __init__(self, index_dict):
        self.index_dict = index_dict
        self.index_keys = list(index_dict.keys())

    def _load(self, key):
        return self.index_dict[key]

    def __len__(self):
        return len(self.index_dict)

    def __contains__(self, key):
        return key in self.index_dict

    def __iter__(self):
        for key in self.index_keys:
            yield key, self._load(key)

    def __getitem__(self, index):
        if type(index) not in [int, str]:
            raise IndexError(f"Unsupported index type: {type(index)}")
        if type(index) == int:
            n = len(self.index_keys)
            if index >= n or index < 0:
                raise KeyError(f"position {index:d} outside the table of {n:d} entries")
            index = self.index_keys[index]
        if index not in self.index_dict:
            raise KeyError(f"Missing utterance {index}!")
        return self._load(index)

    def get(self, index, default=None):
        return self[index] if index in self else default


class ScpReader(Reader):
    def __init__(self, scp_rspecifier, value_processor=lambda x: x, num_tokens=2,
                 restrict=True):
        super().__init__(
            parse_scps(scp_rspecifier, value_processor=value_processor, num_tokens=num_tokens,
                       restrict=restrict))


class Writer(object):
    """Sink of (key, ndarray) pairs: either one archive stream (path, ``-`` or
    nothing) or a directory that receives one file per key; optionally a Kaldi
    script file with one ``key<TAB>location`` line per object
    (reference: data_handler.py:275-323)."""

    def __init__(self, obj_path_or_dir, scp_path=None, is_dir=False):
        self.is_dir = bool(is_dir)
        if obj_path_or_dir == "-" and scp_path:
            warnings.warn("archives go to stdout: the script file would hold no usable offsets, "
                          "not writing it")
            scp_path = None
        if self.is_dir:
            self.target = Path(obj_path_or_dir).absolute()
            self.target.mkdir(parents=True, exist_ok=True)
        else:
            self.target = obj_path_or_dir if obj_path_or_dir == "-" else os.path.abspath(
                obj_path_or_dir)
        self._ark = None if self.is_dir else ext_open(self.target, "wb")
        self._scp = ext_open(scp_path, "w")
        self.ark_fd = None
        self.scp_fd = None

    def __enter__(self):
        if self._ark is not None:
            self.ark_fd = self._ark.open()
        self.scp_fd = self._scp.open()
        return self

    def __exit__(self, *exc):
        if self._ark is not None:
            self._ark.close()
        self._scp.close()
        self.ark_fd = self.scp_fd = None

    def record(self, key, location):
        """One line of the output script file (if any)."""
        if self.scp_fd is not None:
            self.scp_fd.write(f"{key}\t{location}\n")

    def file_for(self, key, suffix):
        return self.target / f"{key}{suffix}"

    @staticmethod
    def check_args(data):
        if not isinstance(data, np.ndarray):
            raise RuntimeError("Instance of Writer accepts np.ndarray object, " +
                               f"but got {type(data)}")

    def write(self, key, data):
        raise NotImplementedError


class ArchiveReader(object):
    """Sequential reader of a Kaldi archive (path, '-' or pipe)."""

    def __init__(self, ark_or_pipe):
        self.ark_or_pipe = ark_or_pipe

    def __iter__(self):
        with ext_open(self.ark_or_pipe, "rb") as fd:
            yield from kaldi_io.read_float_ark(fd)


class WaveReader(ScpReader):
    """Single/multi-channel wave table (reference data_handler.py:326-413)."""

    def __init__(self, wav_scp, sr=16000, normalize=True):
        super().__init__(wav_scp)
        self.sr = sr
        self.normalize = normalize
        self.wav_ark_mgr = {}
        self._last = (None, None)  # (key, samples) of the most recent full read

    def read_internal(self, addr, beg=None, end=None):
        if isinstance(addr, str) and ":" in addr:
            tokens = addr.split(":")
            if len(tokens) != 2:
                raise RuntimeError(f"Value format error: {addr}")
            fname, offset = tokens[0], int(tokens[1])
            if fname not in self.wav_ark_mgr:
                self.wav_ark_mgr[fname] = open(fname, "rb")
            ark = self.wav_ark_mgr[fname]
            ark.seek(offset)
            return read_wav(ark, beg=beg, end=end, normalize=self.normalize, sr=self.sr)
        return read_wav(addr, beg=beg, end=end, normalize=self.normalize, sr=self.sr)

    def read(self, key, beg=None, end=None):
        """C x N matrix or N vector."""
        if beg is None and end is None and self._last[0] == key:
            return self._last[1]
        fname = self.index_dict[key].rstrip()
        if fname[-1] == "|":
            stdout, _ = run_command(fname[:-1], wait=True)
            samps = self.read_internal(BytesIO(stdout))
        else:
            wav_list = glob.glob(fname)
            if ":" in fname and not wav_list:
                wav_list = [fname]
            if len(wav_list) == 0:
                raise RuntimeError(f"Could not find file matches template '{fname}'")
            if len(wav_list) == 1:
                samps = self.read_internal(wav_list[0], beg=beg, end=end)
            else:
                samps = np.vstack(
                    [self.read_internal(addr, beg=beg, end=end) for addr in sorted(wav_list)])
        if beg is None and end is None:
            self._last = (key, samps)
        return samps

    def read_pcm16(self, key):
        """Interleaved int16 frames [N, C] of an entry that is ONE 16-bit PCM file
        (plain path or path.ark:offset), else None: the device then converts and
        transposes (setk_pcm16_to_float) what read() does on the host.  Only with
        normalize=True (the x / 32768 scaling)."""
        if not self.normalize:
            return None
        fname = self.index_dict[key].rstrip()
        if fname[-1] == "|":
            return None
        wav_list = glob.glob(fname)
        if ":" in fname and not wav_list:
            wav_list = [fname]
        if len(wav_list) != 1:
            return None
        addr = wav_list[0]
        if ":" in addr:
            tokens = addr.split(":")
            if len(tokens) != 2:
                raise RuntimeError(f"Value format error: {addr}")
            if tokens[0] not in self.wav_ark_mgr:
                self.wav_ark_mgr[tokens[0]] = open(tokens[0], "rb")
            ark = self.wav_ark_mgr[tokens[0]]
            ark.seek(int(tokens[1]))
            pcm, sr = wavio.read_pcm16_frames(ark)
        else:
            pcm, sr = wavio.read_pcm16_frames(addr)
        if sr != self.sr:
            raise RuntimeError(f"Expect sr={self.sr} of {addr}, get {sr} instead")
        return pcm

    def _load(self, key):
        return self.read(key)

    def peek_nsamps(self, key):
        """Samples per channel from the wave header alone (no decode), or None when
        the entry is a pipe / a glob of several files.  Feeds the duration-balanced
        sharding (the reference balances with split_scp.pl on utterance counts)."""
        fname = self.index_dict[key].rstrip()
        if fname[-1] == "|":
            return None
        wav_list = glob.glob(fname)
        if ":" in fname and not wav_list:
            wav_list = [fname]
        if len(wav_list) != 1:
            return None
        addr = wav_list[0]
        try:
            if ":" in addr and not os.path.exists(addr):
                path, offset = addr.rsplit(":", 1)
                with open(path, "rb") as fd:
                    fd.seek(int(offset))
                    info = wavio.read_header(fd)
            else:
                with open(addr, "rb") as fd:
                    info = wavio.read_header(fd)
        except (OSError, ValueError, wavio.WaveFormatError):
            return None
        frame = info["channels"] * (info["bits"] // 8)
        return info["data_bytes"] // frame if frame else None

    def peek_channels(self, key):
        """Channel count from the wave header(s) alone, or None for a pipe / unreadable header
        (a glob of single-channel files counts its files)."""
        fname = self.index_dict[key].rstrip()
        if fname[-1] == "|":
            return None
        wav_list = glob.glob(fname)
        if ":" in fname and not wav_list:
            wav_list = [fname]
        if not wav_list:
            return None
        if len(wav_list) > 1:
            return len(wav_list)
        addr = wav_list[0]
        try:
            if ":" in addr and not os.path.exists(addr):
                path, offset = addr.rsplit(":", 1)
                with open(path, "rb") as fd:
                    fd.seek(int(offset))
                    return wavio.read_header(fd)["channels"]
            with open(addr, "rb") as fd:
                return wavio.read_header(fd)["channels"]
        except (OSError, ValueError, wavio.WaveFormatError):
            return None

    def first_channels_at_most(self, limit):
        """True if the first utterance's header shows at most `limit` channels (corpora are
        uniform; what the CLIs use to decide on the torch-free mode before any decode)."""
        for key in self.index_keys:
            n = self.peek_channels(key)
            return n is not None and n <= limit
        return True

    def maxabs(self, key):
        return np.max(np.abs(self.read(key)))

    def duration(self, key):
        return self.read(key).shape[-1] / self.sr

    def nsamps(self, key):
        return self.read(key).shape[-1]

    def power(self, key):
        samps = self.read(key)
        s = samps if samps.ndim == 1 else samps[0]
        return np.linalg.norm(s, 2)**2 / s.size


class NumpyReader(ScpReader):
    def _load(self, key):
        return np.load(self.index_dict[key])


class SpectrogramReader(WaveReader):
    """Single/multi-channel STFT table: N x F x T (or F x T) complex64, computed
    on the GPU for all channels in one call (reference data_handler.py:483-503
    loops forward_stft over channels and np.stacks)."""

    def __init__(self, wav_scp, normalize=True, **kwargs):
        super().__init__(wav_scp, normalize=normalize)
        self.stft_kwargs = kwargs

    def _load(self, key):
        samps = super().read(key)
        if samps.ndim == 1:
            return forward_stft(samps, **self.stft_kwargs)
        kw = dict(frame_len=1024, frame_hop=256, round_power_of_two=True, center=False,
                  window="hann", transpose=True)
        kw.update(self.stft_kwargs)
        plain = not any(kw.get(k, False) for k in ("apply_abs", "apply_log", "apply_pow"))
        if not plain:
            return np.stack([forward_stft(np.ascontiguousarray(s), **self.stft_kwargs)
                             for s in samps])
        spec = device_stft(samps, kw["frame_len"], kw["frame_hop"], kw["round_power_of_two"],
                           kw["center"], kw["window"])  # C x T x F
        return spec if kw["transpose"] else np.ascontiguousarray(np.transpose(spec, (0, 2, 1)))


def _split_ark_address(addr):
    """'path/to.ark:1234' -> ('path/to.ark', 1234) (the path may contain ':')."""
    path, sep, offset = addr.rpartition(":")
    if not sep:
        raise ValueError("Unsupported scripts address format")
    return path, int(offset)


class ScriptReader(ScpReader):
    """Kaldi scp of 'ark_path:offset' values -> float matrix/vector.  Archives
    stay open for the life of the reader (reference: data_handler.py:506-535)."""

    def __init__(self, ark_scp):
        super().__init__(ark_scp, value_processor=_split_ark_address)
        self._arks = {}

    def _seek(self, path, offset):
        fd = self._arks.get(path)
        if fd is None:
            fd = self._arks[path] = open(path, "rb")
        fd.seek(offset)
        return fd

    def _load(self, key):
        path, offset = self.index_dict[key]
        return kaldi_io.read_float_mat_vec(self._seek(path, offset), direct_access=True)

    def locate(self, key):
        """(archive path, byte offset of the binary marker) of an entry."""
        return self.index_dict[key]


class ArchiveWriter(Writer):
    """Kaldi binary archive (+ scp with byte offsets of the '\\0B' markers)."""

    def __init__(self, ark_path, scp_path=None, dtype=np.float32):
        if not ark_path:
            raise RuntimeError("Seem configure path of archives as None")
        super().__init__(ark_path, scp_path)
        self.dtype = dtype

    def write(self, key, obj):
        self.check_args(obj)
        fd = self.ark_fd
        kaldi_io.write_token(fd, key)
        offset = fd.tell() if self.target != "-" else None
        kaldi_io.write_binary_symbol(fd)
        kaldi_io.write_float_mat_vec(fd, obj.astype(self.dtype))
        if offset is not None:
            self.record(key, f"{self.target}:{offset:d}")


class WaveWriter(Writer):
    """{dir}/{key}.wav, PCM_16 (reference: data_handler.py:590-605)."""

    def __init__(self, dump_dir, scp_path=None, sr=16000, normalize=True):
        super().__init__(dump_dir, scp_path, is_dir=True)
        self.sr = sr
        self.normalize = normalize

    def write(self, key, obj):
        self.check_args(obj)
        dst = self.file_for(key, ".wav")
        write_wav(str(dst), obj, sr=self.sr, normalize=self.normalize)
        self.record(key, dst)

    def write_pcm16(self, key, pcm):
        """Samples that already are 16-bit PCM (the device quantises with
        libsndfile's rule): no host conversion."""
        dst = self.file_for(key, ".wav")
        wavio.write_pcm16(str(dst), pcm, self.sr)
        self.record(key, dst)


class NumpyWriter(Writer):
    def __init__(self, dump_dir, scp_path=None):
        super().__init__(dump_dir, scp_path, is_dir=True)

    def write(self, key, obj):
        self.check_args(obj)
        dst = self.file_for(key, ".npy")
        np.save(dst, obj)
        self.record(key, dst)
