"""
RIFF/WAVE codec used by read_wav / write_wav (the reference delegates to
SoundFile/libsndfile, scripts/sptk/libs/utils.py:45-92, which is not part of
this image).  Supports PCM 8/16/24/32, IEEE float 32/64 and
WAVE_FORMAT_EXTENSIBLE (the doc assets are EXTENSIBLE PCM16).

Conversions follow libsndfile's defaults: integer PCM -> float32 divides by
2^(bits-1); float -> PCM_16 multiplies by 32767 and rounds to nearest
(lrintf) without clipping.
"""
import io
import os
import struct

import numpy as np

WAVE_FORMAT_PCM = 1
WAVE_FORMAT_IEEE_FLOAT = 3
WAVE_FORMAT_EXTENSIBLE = 0xFFFE


class WaveFormatError(RuntimeError):
    pass


def _read_exact(fd, n):
    data = fd.read(n)
    if len(data) != n:
        raise WaveFormatError("truncated wave stream")
    return data


def read_header(fd):
    """Parse chunks up to 'data'.  Returns dict(fmt, channels, sr, bits,
    data_bytes) with the stream positioned at the first sample."""
    riff = _read_exact(fd, 12)
    if riff[:4] != b"RIFF" or riff[8:12] != b"WAVE":
        raise WaveFormatError("not a RIFF/WAVE stream")
    info = None
    while True:
        head = fd.read(8)
        if len(head) < 8:
            raise WaveFormatError("no data chunk")
        cid, size = head[:4], struct.unpack("<I", head[4:])[0]
        if cid == b"fmt ":
            body = _read_exact(fd, size + (size & 1))
            fmt, ch, sr, _, align, bits = struct.unpack("<HHIIHH", body[:16])
            if fmt == WAVE_FORMAT_EXTENSIBLE and size >= 26:
                fmt = struct.unpack("<H", body[24:26])[0]
            info = dict(fmt=fmt, channels=ch, sr=sr, bits=bits, align=align)
        elif cid == b"data":
            if info is None:
                raise WaveFormatError("data chunk before fmt chunk")
            info["data_bytes"] = size
            return info
        else:
            fd.seek(size + (size & 1), io.SEEK_CUR) if fd.seekable() else _read_exact(
                fd, size + (size & 1))


def _decode(raw, info):
    fmt, bits, ch = info["fmt"], info["bits"], info["channels"]
    if fmt == WAVE_FORMAT_PCM:
        if bits == 16:
            data = np.frombuffer(raw, dtype="<i2")
        elif bits == 32:
            data = np.frombuffer(raw, dtype="<i4")
        elif bits == 8:
            data = np.frombuffer(raw, dtype=np.uint8).astype(np.int16) - 128
        elif bits == 24:
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            data = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
            data = np.where(data & 0x800000, data - (1 << 24), data).astype(np.int32)
        else:
            raise WaveFormatError(f"unsupported PCM width {bits}")
    elif fmt == WAVE_FORMAT_IEEE_FLOAT:
        data = np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8")
    else:
        raise WaveFormatError(f"unsupported wave format tag {fmt}")
    return data.reshape(-1, ch) if ch > 1 else data


def read(file, start=0, stop=None, dtype="float32"):
    """soundfile.read work-alike: returns (data[N] or data[N, C], sr)."""
    own = isinstance(file, (str, bytes)) or hasattr(file, "__fspath__")
    fd = open(file, "rb") if own else file
    try:
        info = read_header(fd)
        frame = info["channels"] * (info["bits"] // 8)
        n_total = info["data_bytes"] // frame if frame else 0
        start = 0 if start is None else int(start)
        stop = n_total if stop is None else min(int(stop), n_total)
        if start:
            if fd.seekable():
                fd.seek(start * frame, io.SEEK_CUR)
            else:
                _read_exact(fd, start * frame)
        n = max(0, stop - start)
        raw = fd.read(n * frame)
        raw = raw[:(len(raw) // frame) * frame] if frame else raw
        data = _decode(raw, info)
    finally:
        if own:
            fd.close()
    bits, is_pcm = info["bits"], info["fmt"] == WAVE_FORMAT_PCM
    if dtype == "float32":
        if is_pcm:
            data = data.astype(np.float32) / np.float32(1 << (bits - 1))
        else:
            data = data.astype(np.float32)
    elif dtype == "int16":
        if is_pcm and bits == 16:
            data = data.astype(np.int16)
        elif is_pcm:
            data = (data.astype(np.int64) >> max(0, bits - 16)).astype(np.int16)
        else:
            data = np.rint(data.astype(np.float64) * 32767.0).astype(np.int16)
    else:
        data = data.astype(dtype)
    return data, info["sr"]


def read_pcm16_frames(file):
    """Interleaved frames of a 16-bit PCM wave as they lie in the file:
    (int16 array [N, C], sr), or (None, sr) when the stream is not PCM16 (the
    caller then takes read()).  Feeds the device-side ingest
    (setk_pcm16_to_float): no host conversion, half the upload."""
    own = isinstance(file, (str, bytes)) or hasattr(file, "__fspath__")
    fd = open(file, "rb") if own else file
    try:
        info = read_header(fd)
        if info["fmt"] != WAVE_FORMAT_PCM or info["bits"] != 16:
            return None, info["sr"]
        ch = info["channels"]
        raw = fd.read(info["data_bytes"])
        raw = raw[:(len(raw) // (2 * ch)) * 2 * ch]
        return np.frombuffer(raw, dtype="<i2").reshape(-1, ch), info["sr"]
    finally:
        if own:
            fd.close()


def float_to_pcm16(samps):
    """libsndfile float -> short: lrintf(x * 0x7FFF), no clipping (wraps)."""
    pcm = np.rint(np.asarray(samps, dtype=np.float64) * 32767.0)
    return pcm.astype(np.int64).astype(np.int16)


def write_pcm16(file, pcm, sr):
    """pcm: int16 [N] or [N, C]."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    ch = 1 if pcm.ndim == 1 else pcm.shape[1]
    # the samples are written out of the array itself: a bytes copy of every waveform is a
    # memcpy under the GIL in each writer thread of the streaming pipeline
    data = memoryview(pcm.reshape(-1)).cast("B") if pcm.size else b""
    nbytes = pcm.nbytes
    hdr = b"RIFF" + struct.pack("<I", 36 + nbytes) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, WAVE_FORMAT_PCM, ch, int(sr), int(sr) * ch * 2, ch * 2, 16)
    hdr += b"data" + struct.pack("<I", nbytes)
    if isinstance(file, (str, bytes)) or hasattr(file, "__fspath__"):
        # written under a temporary name and renamed into place: a run killed in the middle of a
        # file never leaves a truncated {key}.wav behind (--skip-existing trusts what it finds)
        final = os.fspath(file)
        tmp = final + (b".part" if isinstance(final, bytes) else ".part")
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
        try:
            bufs, want = [hdr, data], len(hdr) + nbytes
            n = os.writev(fd, bufs)
            while n < want:  # short write: continue from where it stopped
                whole = hdr + bytes(data)
                n += os.write(fd, memoryview(whole)[n:])
        except BaseException:
            os.close(fd)
            try:
                os.unlink(tmp)
            except OSError:
                pass
            raise
        os.close(fd)
        os.replace(tmp, final)
    else:
        file.write(hdr)
        file.write(data)


def write(file, data, samplerate):
    """soundfile.write work-alike for WAV: float input -> PCM_16."""
    data = np.asarray(data)
    if data.dtype.kind == "f":
        pcm = float_to_pcm16(data)
    else:
        pcm = data.astype(np.int16)
    write_pcm16(file, pcm, samplerate)
