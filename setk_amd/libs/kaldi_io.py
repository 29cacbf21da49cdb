"""
Kaldi binary table I/O for the mask inputs of the beamformer CLI
(the format handled by scripts/sptk/libs/kaldi_io.py in the reference):

    <key> ' ' '\\0' 'B' <object>
    object  := 'FM ' | 'DM ' <i32 rows> <i32 cols> <row-major payload>
             | 'FV ' | 'DV ' <i32 dim> <payload>
             | 'CM ' | 'CM2 ' | 'CM3 ' <GlobalHeader f32 min, f32 range, i32 rows, i32 cols> ...
    <i32>   := '\\x04' <little-endian int32>

Readers return read-only numpy views where the reference does
(np.frombuffer).  Compressed matrices (Kaldi CompressedMatrix, one-byte with
per-column headers / two-byte / one-byte) are expanded with vectorised numpy.
"""
import struct

import numpy as np

_FLOAT_TYPES = {"FM": np.float32, "DM": np.float64, "FV": np.float32, "DV": np.float64}


class KaldiFormatError(RuntimeError):
    pass


def _need(cond, msg):
    if not cond:
        raise KaldiFormatError(msg)


def read_token(fd):
    """Bytes up to (and consuming) the next space; None at end of stream."""
    chars = []
    while True:
        c = fd.read(1)
        if c in (b" ", b""):
            break
        chars.append(c)
    tok = b"".join(chars).decode().strip()
    return tok or None


def write_token(fd, token):
    fd.write((token + " ").encode())


def expect_binary(fd):
    flag = fd.read(2)
    _need(flag == b"\0B", f"Expect binary flags '\\0B', but gets {flag!r}")


def write_binary_symbol(fd):
    fd.write(b"\0B")


def read_key(fd):
    key = read_token(fd)
    if key:
        expect_binary(fd)
    return key


def read_int32(fd):
    size = fd.read(1)
    _need(size == b"\x04", f"Expect '\\04', but gets {size!r}")
    return struct.unpack("<i", fd.read(4))[0]


def write_int32(fd, value):
    fd.write(b"\x04" + struct.pack("<i", int(value)))


def read_common_mat(fd, kind=None):
    kind = kind or read_token(fd)
    _need(kind in ("FM", "DM"), f"Unknown matrix type: {kind}")
    dt = np.dtype(_FLOAT_TYPES[kind])
    rows, cols = read_int32(fd), read_int32(fd)
    payload = fd.read(dt.itemsize * rows * cols)
    _need(len(payload) == dt.itemsize * rows * cols, "truncated matrix payload")
    return np.frombuffer(payload, dtype=dt).reshape(rows, cols)


def read_float_vec(fd, direct_access=False, kind=None):
    if direct_access:
        expect_binary(fd)
    kind = kind or read_token(fd)
    _need(kind in ("FV", "DV"), f"Unknown vector type: {kind}")
    dt = np.dtype(_FLOAT_TYPES[kind])
    dim = read_int32(fd)
    return np.frombuffer(fd.read(dt.itemsize * dim), dtype=dt)


def uncompress(payload, kind, head):
    """Expand a Kaldi CompressedMatrix body.  head = (min, range, rows, cols)."""
    vmin, vrange, rows, cols = head
    if kind == "CM":
        _need(len(payload) == cols * (8 + rows), "bad CM payload size")
        pch = np.frombuffer(payload[:8 * cols], dtype="<u2").astype(np.float32)
        pch = pch.reshape(cols, 4).T * np.float32(vrange) / np.float32(65535.0) + np.float32(vmin)
        q = np.frombuffer(payload[8 * cols:], dtype=np.uint8).astype(np.float32)
        q = q.reshape(cols, rows).T
        p0, p25, p75, p100 = pch
        lo = q * (p25 - p0) / 64.0 + p0
        mid = (q - 64) * (p75 - p25) / 128.0 + p25
        hi = (q - 192) * (p100 - p75) / 63.0 + p75
        return np.where(q <= 64, lo, np.where(q >= 193, hi, mid))
    if kind == "CM2":
        step = float(vrange / 65535.0)
        q = np.frombuffer(payload, dtype="<u2").astype(np.float32)
    elif kind == "CM3":
        step = float(vrange / 255.0)
        q = np.frombuffer(payload, dtype=np.uint8).astype(np.float32)
    else:
        raise KaldiFormatError(f"Unknown matrix compressing type: {kind}")
    return vmin + q.reshape(rows, cols) * step


def read_compress_mat(fd, kind=None):
    kind = kind or read_token(fd)
    head = struct.unpack("<ffii", fd.read(16))
    rows, cols = head[2], head[3]
    nbytes = {"CM": cols * (8 + rows), "CM2": 2 * rows * cols, "CM3": rows * cols}.get(kind)
    _need(nbytes is not None, f"Unknown matrix compressing type: {kind}")
    return uncompress(fd.read(nbytes), kind, head)


def _read_typed(fd, kind, allow_vec):
    """Dispatch on the type token that precedes every Kaldi object (the token is
    read, never peeked: BufferedReader.peek may return a single byte at a buffer
    boundary)."""
    _need(kind is not None, "unexpected end of archive")
    if kind[0] == "C":
        return read_compress_mat(fd, kind)
    _need(kind[0] != "S", "sparse matrices are not supported on the mask path")
    if kind in ("FV", "DV"):
        _need(allow_vec, f"Unknown matrix type: {kind}")
        return read_float_vec(fd, kind=kind)
    return read_common_mat(fd, kind)


def read_general_mat(fd, direct_access=False):
    if direct_access:
        expect_binary(fd)
    return _read_typed(fd, read_token(fd), allow_vec=False)


def read_float_mat_vec(fd, direct_access=False):
    """Matrix or vector at the current position (scp offsets point at '\\0B')."""
    if direct_access:
        expect_binary(fd)
    return _read_typed(fd, read_token(fd), allow_vec=True)


def write_common_mat(fd, mat):
    _need(mat.dtype in (np.float32, np.float64), "matrix must be float32/float64")
    _need(mat.ndim == 2, f"Only support 2D matrix, but got {mat.ndim:d}")
    write_token(fd, "FM" if mat.dtype == np.float32 else "DM")
    write_int32(fd, mat.shape[0])
    write_int32(fd, mat.shape[1])
    fd.write(np.ascontiguousarray(mat).tobytes())


def write_float_vec(fd, vec):
    _need(vec.dtype in (np.float32, np.float64), "vector must be float32/float64")
    _need(vec.ndim == 1, f"Only support vector, but got {vec.ndim:d}D matrix")
    write_token(fd, "FV" if vec.dtype == np.float32 else "DV")
    write_int32(fd, vec.size)
    fd.write(np.ascontiguousarray(vec).tobytes())


def write_float_mat_vec(fd, obj):
    if not isinstance(obj, np.ndarray):
        raise TypeError(f"Unsupport type: {type(obj)}")
    if obj.ndim == 2:
        write_common_mat(fd, obj)
    else:
        write_float_vec(fd, obj)


def read_float_ark(fd):
    """Sequential (key, matrix|vector) pairs of a binary archive."""
    while True:
        key = read_key(fd)
        if not key:
            return
        yield key, read_float_mat_vec(fd)
