"""
Streaming host pipeline of the beamformer CLI: disk -> pinned staging slab ->
H2D -> fused kernels -> D2H -> wav files, several batches in flight.

This replaces the reference's per-utterance loop
(scripts/sptk/apply_adaptive_beamformer.py:130-178 on top of
libs/data_handler.py:345-413: decode, compute, encode, one utterance at a time on
one core).  The kernels need ~16 us per 8-ch / 30-s utterance; reading its
7.7 MB of PCM and 1.9 MB of mask and crossing PCIe costs 20x that, so the host
side is organised as a pipeline with every stage overlapped:

  plan      (caller's thread)  wave / mask headers only: channels, samples, frames,
                               payload offsets -> byte offsets inside a slab
  read+H2D  (library's reader  every file payload of the batch into the slot's page-locked slab --
             pool, copy-in     the wav's 16-bit frames and the mask's float32 rows exactly as
             stream)           stored, no host conversion -- by ONE call of setk_host_read_payloads
                               (native threads: a MADV_SEQUENTIAL mapping + memcpy per payload, no
                               interpreter lock in between; SETK_READ_MODE=mmap | preadv: the
                               interpreter's thread pool, one task per utterance), then one async
                               copy of the slab (or of each payload's slice).
                               Option zero_copy: mmap the file, pin its page-cache pages
                               (hipHostRegister) and DMA from where they lie; alone that is
                               0.23 ms + 52 GB/s per 7.7 MB file, inside the pipeline it
                               measured slower than staging (memory-map lock), so it is off
                               by default.
  compute   (compute stream)   setk_pcm16_to_float_batch + setk_enhance_batch, status
                               and PCM16 output written into the device out-slab
  D2H       (copy-out stream)  ONE hipMemcpyAsync per batch
  write     (thread pool)      RIFF header + the slab's int16 samples -> {key}.wav

A slab slot (pinned host in/out + device in/out + device float audio) is owned
by one batch from `read` to `write`; `depth` slots bound the memory and give the
overlap.  Inputs the fast path cannot take as stored (non-PCM16 wavs, pipes,
per-channel globs, compressed / transposed / float64 masks) are decoded on the
host by the ordinary readers and copied into the slab as float32.

PyTorch is plumbing here: pinned / device buffers, streams and events.
"""
import contextlib
import mmap
import os
import queue
import struct
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _ffi
from .libs import wavio

# how a payload gets from the page cache into the page-locked slab: "native" (default: the
# library's reader pool, one call per batch, setk_host_read_payloads), "mmap" or "preadv" (the
# interpreter's reader threads, one task per utterance)
READ_MODE = os.environ.get("SETK_READ_MODE", "native")
# payloads below this size are read with preadv: an mmap + munmap pair costs ~250 us under load
# (the unmap's TLB shootdown reaches every CPU the process's threads ran on), as much as copying
# 2.5 MB (tools/ubench/read_small.py)
MMAP_MIN_BYTES = int(os.environ.get("SETK_MMAP_MIN_KB", "256")) << 10

ALIGN = 256


# SETK_CM_DEVICE=0: compressed Kaldi masks are decoded on the host (libs/kaldi_io.uncompress) as in
# rounds 1 - 5 instead of on the device
DEVICE_CM_DECODE = os.environ.get("SETK_CM_DEVICE", "1") != "0"


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


# ----------------------------------------------------------------------------
# sources: what can be read straight into a slab
# ----------------------------------------------------------------------------
class Mapping(object):
    """Read-only mmap of a whole file, pinned for DMA while a batch is in flight."""

    __slots__ = ("mm", "view", "addr", "size", "pinned")

    def __init__(self, fd, size):
        import mmap
        self.mm = mmap.mmap(fd, size, prot=mmap.PROT_READ)
        self.view = np.frombuffer(self.mm, dtype=np.uint8)
        self.addr = self.view.ctypes.data
        self.size = size
        self.pinned = False

    def close(self, ctx):
        if self.pinned:
            ctx.host_unregister(self.addr)
            self.pinned = False
        self.view = None
        try:
            self.mm.close()
        except BufferError:  # a stray export keeps the map alive until collected
            pass


class Payload(object):
    """A contiguous byte range of a file that IS the data wanted (int16 wav frames [N][C]
    or float32 mask rows [T][F]), or a host array to copy.  The file is named, not held open:
    the thread that reads it opens and closes it.  (Descriptors held until their batch was read
    made the process's descriptor table grow to thousands of entries, and every doubling of that
    table in a multi-threaded process waits for an RCU grace period: 0.1 - 0.15 s each on the
    256-core host, tools/ubench/open_probe.py.)"""

    __slots__ = ("path", "offset", "nbytes", "array", "fsize", "cm")

    def __init__(self, path=None, offset=0, nbytes=0, array=None, fsize=0, cm=None):
        self.path, self.offset, self.nbytes, self.array = path, offset, nbytes, array
        self.fsize = fsize  # size of the file (0: unknown -> staged path)
        # a Kaldi CompressedMatrix body (kind, vmin, vrange, rows, cols): the bytes go up as stored
        # and are expanded to float32 [T][F] on the device (setk_kaldi_cm_decode_batch)
        self.cm = cm

    def zero_copy_ok(self):
        """The payload is (nearly) the whole file: worth pinning the file's pages."""
        return (self.array is None and self.fsize > 0 and self.nbytes >= (32 << 10) and
                self.fsize - self.nbytes <= (64 << 10))

    def load_into(self, dst):
        """dst: writable uint8 numpy view of exactly nbytes."""
        if self.array is not None:
            dst[:] = np.frombuffer(self.array, dtype=np.uint8)
            return
        fd = os.open(self.path, os.O_RDONLY)
        try:
            if READ_MODE != "preadv" and self.nbytes >= MMAP_MIN_BYTES and \
                    os.fstat(fd).st_size >= self.offset + self.nbytes:  # (a short file: IOError below)
                # A read() marks every page accessed, and the FIRST access of a page moves it
                # between the kernel's LRU lists under a shared lock: first reads of fresh
                # page-cache pages reach 15 - 19 GB/s with 6 - 12 threads however they are
                # issued.  Faulting the pages in through a MADV_SEQUENTIAL mapping skips that:
                # 43 - 54 GB/s on the same files (tools/ubench/first_read_mmap.py).  The copy
                # out of the mapping releases the GIL like the syscall does.
                lo = self.offset & ~(mmap.ALLOCATIONGRANULARITY - 1)
                m = mmap.mmap(fd, self.offset - lo + self.nbytes, flags=mmap.MAP_SHARED,
                              prot=mmap.PROT_READ, offset=lo)
                try:
                    m.madvise(mmap.MADV_SEQUENTIAL)
                    src = np.frombuffer(m, dtype=np.uint8, count=self.nbytes, offset=self.offset - lo)
                    np.copyto(dst, src)
                    del src
                finally:
                    try:
                        m.close()
                    except BufferError:  # pragma: no cover
                        pass
                return
            got = 0
            mv = memoryview(dst)
            while got < self.nbytes:  # os.preadv releases the GIL, no seek state
                n = os.preadv(fd, [mv[got:]], self.offset + got)
                if n <= 0:
                    raise IOError("truncated payload")
                got += n
        finally:
            os.close(fd)


class OpenFiles(object):
    """Small cache of read-only descriptors for the header probes of the planning thread (the
    reference's readers also keep archives open, data_handler.py:343, 522-529): an archive
    that holds many entries is opened once, plain files pass through."""

    def __init__(self, limit=32):
        self.fds = {}
        self.limit = limit
        self.lock = threading.Lock()

    def get(self, path):
        with self.lock:
            fd = self.fds.get(path)
            if fd is None:
                if len(self.fds) >= self.limit:
                    # oldest quarter (nothing else refers to these descriptors)
                    for p in list(self.fds)[:self.limit // 4]:
                        os.close(self.fds.pop(p))
                fd = self.fds[path] = os.open(path, os.O_RDONLY)
            return fd

    def close(self):
        with self.lock:
            for fd in self.fds.values():
                os.close(fd)
            self.fds.clear()


def probe_wav(files, path, offset=0):
    """Header of a wave file (or of a wave inside an archive at `offset`) ->
    (info dict, absolute payload offset)."""
    fd = files.get(path)

    class _Buf(object):  # minimal file interface over the header bytes for wavio
        def __init__(self, b):
            self.b, self.p = b, 0

        def read(self, n):
            out = self.b[self.p:self.p + n]
            self.p += len(out)
            return out

        def seekable(self):
            return True

        def seek(self, n, whence=0):
            self.p = self.p + n if whence == 1 else n

    for window in (4096, 1 << 20):  # headers with long LIST chunks need the second try
        buf = _Buf(os.pread(fd, window, offset))
        try:
            info = wavio.read_header(buf)
            return info, offset + buf.p
        except wavio.WaveFormatError:
            if window != 4096:
                raise


def probe_npy(files, path):
    """-> (shape, dtype, fortran, payload offset) of a .npy file."""
    fd = files.get(path)
    head = os.pread(fd, 4096, 0)
    if head[:6] != b"\x93NUMPY":
        raise ValueError(f"{path}: not a .npy file")
    major = head[6]
    if major == 1:
        hlen = struct.unpack("<H", head[8:10])[0]
        start = 10
    else:
        hlen = struct.unpack("<I", head[8:12])[0]
        start = 12
    if start + hlen > len(head):
        head = os.pread(fd, start + hlen, 0)
    import ast
    d = ast.literal_eval(head[start:start + hlen].decode("latin1"))
    return tuple(d["shape"]), np.dtype(d["descr"]), bool(d["fortran_order"]), start + hlen


def probe_kaldi_matrix(files, path, offset):
    """Binary Kaldi float matrix at `offset` (the scp offset points at '\\0B') ->
    (rows, cols, dtype, payload offset), or None for compressed / vector entries."""
    fd = files.get(path)
    head = os.pread(fd, 32, offset)
    p = 0
    if head[:2] == b"\x00B":
        p = 2
    tok = head[p:p + 3]
    if tok not in (b"FM ", b"DM "):
        return None
    p += 3
    if head[p] != 4 or head[p + 5] != 4:
        return None
    rows = struct.unpack("<i", head[p + 1:p + 5])[0]
    cols = struct.unpack("<i", head[p + 6:p + 10])[0]
    return rows, cols, np.dtype("<f4" if tok == b"FM " else "<f8"), offset + p + 10


def probe_kaldi_compressed(files, path, offset):
    """Binary Kaldi CompressedMatrix at `offset` -> (kind, vmin, vrange, rows, cols, body offset,
    body bytes), or None.  Layout (libs/kaldi_io.py:295-318 of the reference): '\\0B', the token
    'CM ' | 'CM2 ' | 'CM3 ', the global header <f32 min, f32 range, i32 rows, i32 cols>, the body."""
    fd = files.get(path)
    head = os.pread(fd, 40, offset)
    p = 2 if head[:2] == b"\x00B" else 0
    for tok, kind in ((b"CM2 ", "CM2"), (b"CM3 ", "CM3"), (b"CM ", "CM")):
        if head[p:p + len(tok)] == tok:
            p += len(tok)
            break
    else:
        return None
    if len(head) < p + 16:
        return None
    vmin, vrange, rows, cols = struct.unpack("<ffii", head[p:p + 16])
    if rows <= 0 or cols <= 0:
        return None
    nbytes = {"CM": cols * (8 + rows), "CM2": 2 * rows * cols, "CM3": rows * cols}[kind]
    return kind, vmin, vrange, rows, cols, offset + p + 16, nbytes


# ----------------------------------------------------------------------------
# one utterance of a batch
# ----------------------------------------------------------------------------
class Job(object):
    __slots__ = ("key", "C", "N", "T", "L", "pcm16", "audio", "mask", "itf", "off_audio",
                 "off_mask", "off_itf", "off_f32", "off_out", "error", "power", "pw_idx",
                 "off_mask_f32", "off_itf_f32")

    def __init__(self, key):
        self.key = key
        self.error = None


class _Slot(object):
    """Buffers of one in-flight batch: device slabs and their page-locked twins, from the
    library's own allocator (setk_device_alloc / setk_host_alloc)."""

    def __init__(self, ctx, in_cap, f32_cap, out_cap):
        self.ctx = ctx
        self.lock = threading.Lock()
        self.maps = []  # Mapping objects pinned for the batch in flight
        self.in_cap = self.f32_cap = self.out_cap = 0
        self.d_in = self.d_f32 = self.d_out = 0      # device addresses
        self.h_in = self.h_out = 0                   # page-locked host addresses
        self.np_in = self.np_out = None              # ... and their uint8 views
        self.ensure(in_cap, f32_cap, out_cap)
        self.e_in = ctx.event_create()
        self.e_compute = ctx.event_create()
        self.e_out = ctx.event_create()

    def ensure(self, in_cap, f32_cap, out_cap):
        ctx = self.ctx
        if in_cap > self.in_cap:
            if self.d_in:
                ctx.device_free(self.d_in)
            if self.h_in:
                self.np_in = None
                ctx.host_free(self.h_in)
                self.h_in = 0
            self.d_in = ctx.device_alloc(in_cap)
            self.in_cap = in_cap  # page-locked twin: made when a payload needs staging
        if f32_cap > self.f32_cap:
            if self.d_f32:
                ctx.device_free(self.d_f32)
            self.d_f32 = ctx.device_alloc(f32_cap)
            self.f32_cap = f32_cap
        if out_cap > self.out_cap:
            if self.d_out:
                ctx.device_free(self.d_out)
            if self.h_out:
                self.np_out = None
                ctx.host_free(self.h_out)
            self.h_out, self.np_out = ctx.host_alloc(out_cap)
            self.d_out = ctx.device_alloc(out_cap)
            self.out_cap = out_cap

    def staging(self):
        """The page-locked twin of d_in (allocated on first use)."""
        with self.lock:
            if self.np_in is None:
                self.h_in, self.np_in = self.ctx.host_alloc(self.in_cap)
            return self.np_in

    def release(self):
        ctx = self.ctx
        self.np_in = self.np_out = None
        for p in (self.d_in, self.d_f32, self.d_out):
            if p:
                ctx.device_free(p)
        for p in (self.h_in, self.h_out):
            if p:
                ctx.host_free(p)
        for e in (self.e_in, self.e_compute, self.e_out):
            ctx.event_destroy(e)
        self.d_in = self.d_f32 = self.d_out = self.h_in = self.h_out = 0


class StreamPipeline(object):
    """
    pipe = StreamPipeline(engine, sink, ...)      engine: a BatchEnhancer (n_fft 512)
    pipe.submit(key, wav_payload_or_array, mask_payload_or_array, itf_or_None)
    ...
    pipe.close()  -> (num_done, stats)

    announce(key, power) is called in submission order when a batch has come back
    (the CLI's "Processing utterance ..." line); sink(key, pcm_int16_view, status,
    error) -> bool from writer threads, in no particular order (status != 0: the
    reference's LinAlgError case; error: an exception raised while reading).
    """

    def __init__(self, engine, sink, announce=None, batch_utts=32, depth=3, read_threads=None,
                 write_threads=4, slab_mb=0, zero_copy=False, h2d="batch", batch_mb=80):
        self.engine = engine
        self.ctx = engine.ctx
        self.sink = sink
        self.announce = announce
        self.zero_copy = bool(zero_copy)
        # "payload": every payload is DMA'd as soon as it has been read (copies overlap the
        # reads of the same batch); "batch": ONE copy of the whole slab once the batch is in
        # (55 GB/s for one 300 MB copy against 36 GB/s for 64 pieces, profiles/r02z_*)
        if h2d not in ("payload", "batch"):
            raise ValueError(f"h2d must be 'payload' or 'batch', got {h2d!r}")
        self.h2d = h2d
        self.batch_utts = max(1, int(batch_utts))
        # A batch closes at `batch_utts` utterances OR `batch_mb` MB of input, whichever comes
        # first: what a run pays per slab -- page-locking it, the ramp until three are in flight,
        # unpinning it at the end -- grows with its BYTES, while ~80 MB already keeps the copy
        # engine and the kernels busy (round 6, profiles/round6/e2e_batch_size_sweep.txt: 192 files
        # of 8-ch 30 s in 0.54 s with 77 MB slabs against 0.82 s with 307 MB ones; 1536 files 0.87 - 0.91
        # against 0.95 s).  0 = count only.
        self.batch_bytes = max(0, int(batch_mb)) << 20
        self.pending_bytes = 0
        self.F = engine.num_bins
        ncpu = os.cpu_count() or 4
        # more readers are slower: the page-cache copies of many threads contend in the kernel
        # (profiles/r02z_e2e_host_path.txt: 1024 utterances in 0.60 - 0.71 s with 6 - 16, 0.82 - 0.93 with 32)
        self.read_threads = read_threads or max(4, min(12, ncpu // 2))
        self.readers = ThreadPoolExecutor(self.read_threads, thread_name_prefix="setk-read")
        self.writers = ThreadPoolExecutor(write_threads, thread_name_prefix="setk-write")
        self.depth = depth
        self.free_slots = queue.Queue()
        self.slots_made = 0
        self.min_in = slab_mb << 20
        self.launch_q = queue.Queue(maxsize=depth)
        self.done_q = queue.Queue(maxsize=depth)
        self.pending = []
        self.group = None
        self.num_done = 0
        self.num_failed = 0
        self.exc = None
        self.stats = dict(batches=0, utts=0, bytes_in=0, bytes_out=0, t_read=0.0, t_h2d_wait=0.0,
                          t_launch=0.0, t_d2h_wait=0.0, t_write=0.0, t_plan=0.0, t_slot_wait=0.0,
                          t_alloc=0.0, zero_copy_payloads=0, staged_payloads=0)
        self.lock = threading.Lock()
        engine._plan()
        # buffers, streams and events come from the library: no torch in this process
        self.s_in = self.ctx.stream_create()
        self.s_compute = self.ctx.stream_create()
        self.s_out = self.ctx.stream_create()
        self.slots = []
        self.t_first = None
        self.launcher = threading.Thread(target=self._launch_loop, name="setk-launch", daemon=True)
        self.completer = threading.Thread(target=self._complete_loop, name="setk-done", daemon=True)
        self.launcher.start()
        self.completer.start()

    # ---- planning (caller's thread) ---------------------------------------------
    def submit(self, key, audio, mask, itf=None):
        """audio: Payload of int16 frames + (C, N) via .meta, or a C x N float32 array;
        mask / itf: Payload of float32 [T][F] rows or a T x F array.  See make_*()."""
        if self.exc:
            raise self.exc
        if self.t_first is None:
            self.t_first = time.perf_counter()
        t0 = time.perf_counter()
        job = Job(key)
        if isinstance(audio, tuple):          # (Payload, C, N): PCM16 as stored
            job.audio, job.C, job.N = audio
            job.pcm16 = True
        else:
            a = np.ascontiguousarray(audio, dtype=np.float32)
            if a.ndim == 1:
                a = a[None]
            job.C, job.N = a.shape
            job.audio = Payload(array=a, nbytes=a.nbytes)
            job.pcm16 = False
            job.power = float(np.dot(a[0].astype(np.float64), a[0])) / max(a.shape[1], 1)
        job.T, job.L = self.engine.frames_and_length(job.N)
        job.mask = self._mask_payload(mask, job.T)
        job.itf = None if itf is None else self._mask_payload(itf, job.T)
        g = (job.C, job.itf is not None)
        if self.pending and (g != self.group or len(self.pending) >= self.batch_utts):
            self._dispatch()
        self.group = g
        self.pending.append(job)
        self.pending_bytes += job.audio.nbytes + job.mask.nbytes + (job.itf.nbytes if job.itf is not None else 0)
        with self.lock:
            self.stats["t_plan"] += time.perf_counter() - t0
        if len(self.pending) >= self.batch_utts or (self.batch_bytes and self.pending_bytes >= self.batch_bytes):
            self._dispatch()

    def _mask_payload(self, m, T):
        if isinstance(m, Payload) and m.cm is not None:
            # (the shape rules of engine.condition_mask / apply_adaptive_beamformer.py:146-151)
            rows, cols = m.cm[3], m.cm[4]
            if rows == self.F and cols != self.F:
                rows, cols = cols, rows
            if cols != self.F:
                raise ValueError("Input mask matrix should be shape as " +
                                 f"[num_frames x num_bins], now is {(m.cm[3], m.cm[4])}")
            if rows != T:
                raise ValueError("Shape of input obs do not match with mask matrix, " +
                                 f"{T} frames vs {(m.cm[3], m.cm[4])}")
            return m
        if isinstance(m, Payload):
            if m.nbytes != T * self.F * 4:
                raise ValueError("Shape of input obs do not match with mask matrix, " +
                                 f"{T} frames vs {m.nbytes // (4 * self.F)} mask rows")
            return m
        m = self.engine.condition_mask(m, T)
        m = np.ascontiguousarray(m, dtype=np.float32)
        return Payload(array=m, nbytes=m.nbytes)

    def _get_slot(self, in_cap, f32_cap, out_cap):
        t0 = time.perf_counter()
        try:
            return self._get_slot_inner(in_cap, f32_cap, out_cap)
        finally:
            with self.lock:
                self.stats["t_slot_wait"] += time.perf_counter() - t0

    def _get_slot_inner(self, in_cap, f32_cap, out_cap):
        if self.slots_made < self.depth and self.free_slots.empty():
            self.slots_made += 1
            grow = 1.25  # head room: later batches of ragged lengths reuse the buffers
            t0 = time.perf_counter()
            slot = _Slot(self.ctx, max(int(in_cap * grow), self.min_in),
                         int(f32_cap * grow), int(out_cap * grow))
            self.slots.append(slot)
            with self.lock:
                self.stats["t_alloc"] += time.perf_counter() - t0
        else:
            while True:
                try:
                    slot = self.free_slots.get(timeout=0.5)
                    break
                except queue.Empty:
                    if self.exc:
                        raise self.exc
            slot.ensure(in_cap, f32_cap, out_cap)
        return slot

    def _dispatch(self):
        batch, self.pending = self.pending, []
        self.pending_bytes = 0
        if not batch:
            return
        # slab layout: [audio payloads | masks | itf masks]; out: [pcm16 waves | status | power]
        off = 0
        f32 = 0
        out = 0
        for j in batch:
            j.off_audio = off
            off = _align(off + j.audio.nbytes)
        for j in batch:
            j.off_mask = off
            off = _align(off + j.mask.nbytes)
            if j.itf is not None:
                j.off_itf = off
                off = _align(off + j.itf.nbytes)
        # the device twin of a 16-bit payload: planar int16 [C][stride] (the fused kernels read
        # it directly, SETK_FLAG_IN_PCM16) -- or float32 [C][N] when the batch mixes sample
        # formats or the geometry is not the matrix-core pass 2's
        direct = self.engine.pcm_direct_ok and all(j.pcm16 for j in batch)
        for j in batch:
            j.off_f32 = f32
            if j.pcm16:
                f32 = _align(f32 + (2 * j.C * ((j.N + 7) & ~7) if direct else 4 * j.C * j.N))
            for p, name in ((j.mask, "off_mask_f32"), (j.itf, "off_itf_f32")):
                setattr(j, name, None)
                if p is not None and p.cm is not None:
                    setattr(j, name, f32)          # the expanded mask lives in the device-only slab
                    f32 = _align(f32 + 4 * j.T * self.F)
            j.off_out = out
            out = _align(out + 2 * j.L)
        n = len(batch)
        off_status = out
        off_power = _align(off_status + 4 * n)
        out_total = _align(off_power + 8 * n)
        slot = self._get_slot(max(off, ALIGN), max(f32, ALIGN), out_total)
        t0 = time.perf_counter()
        if READ_MODE == "native" and not self.zero_copy:
            futs = [self.readers.submit(self._read_batch, batch, slot)]
        else:
            futs = [self.readers.submit(self._read_job, j, slot) for j in batch]
        self.launch_q.put((batch, slot, futs, off, off_status, off_power, out_total, t0))

    # ---- read stage (pool) --------------------------------------------------------
    def _to_device(self, payload, slot, off):
        """One payload -> device slab at `off`, enqueued on the copy-in stream."""
        ctx = self.ctx
        dst = slot.d_in + off
        stream = self.s_in
        if self.zero_copy and payload.zero_copy_ok():
            fd = os.open(payload.path, os.O_RDONLY)
            try:
                m = Mapping(fd, payload.fsize)  # the mapping keeps the file, not the descriptor
            finally:
                os.close(fd)
            if ctx.host_register(m.addr, m.size):
                m.pinned = True
                ctx.memcpy_h2d_async(dst, m.addr + payload.offset, payload.nbytes, stream)
                with slot.lock:
                    slot.maps.append(m)
                return True
            m.close(ctx)  # e.g. the same file twice in one batch: stage it instead
        buf = slot.staging()
        view = buf[off:off + payload.nbytes]
        payload.load_into(view)
        if self.h2d == "payload" or self.zero_copy:
            # (with --zero-copy the slab holds only the payloads that could not be mapped:
            # they are copied one by one, there is no whole-slab copy)
            ctx.memcpy_h2d_async(dst, view.ctypes.data, payload.nbytes, stream)
        return False

    def _read_job(self, job, slot):
        try:
            zc = self._to_device(job.audio, slot, job.off_audio)
            zc += self._to_device(job.mask, slot, job.off_mask)
            n = 2
            if job.itf is not None:
                zc += self._to_device(job.itf, slot, job.off_itf)
                n += 1
            with self.lock:
                self.stats["zero_copy_payloads"] += int(zc)
                self.stats["staged_payloads"] += n - int(zc)
        except Exception as e:  # reported per utterance by the completer
            job.error = e

    def _read_batch(self, batch, slot):
        """The read stage of a whole batch: its file payloads in ONE call of the library's reader
        pool (no interpreter lock between payloads), host arrays copied here."""
        buf = slot.staging()
        base = buf.ctypes.data
        paths, offs, sizes, dsts, owner = [], [], [], [], []
        arrays = []
        for j in batch:
            for p, off in ((j.audio, j.off_audio), (j.mask, j.off_mask),
                           (j.itf, j.off_itf if j.itf is not None else 0)):
                if p is None:
                    continue
                if p.array is not None:
                    arrays.append((j, p, off))
                else:
                    paths.append(p.path)
                    offs.append(p.offset)
                    sizes.append(p.nbytes)
                    dsts.append(base + off)
                    owner.append(j)
        try:
            status = _ffi.host_read_payloads(paths, offs, sizes, dsts, self.read_threads, MMAP_MIN_BYTES)
        except Exception as e:
            for j in batch:
                j.error = e
            return
        for j, path, st in zip(owner, paths, status):
            if st and j.error is None:
                j.error = IOError("truncated payload") if st == 5 else OSError(st, os.strerror(st), path)
        for j, p, off in arrays:
            try:
                p.load_into(buf[off:off + p.nbytes])
            except Exception as e:
                j.error = e
        if self.h2d == "payload":
            for j in batch:
                if j.error is None:
                    for p, off in ((j.audio, j.off_audio), (j.mask, j.off_mask),
                                   (j.itf, j.off_itf if j.itf is not None else 0)):
                        if p is not None:
                            self.ctx.memcpy_h2d_async(slot.d_in + off, base + off, p.nbytes, self.s_in)
        with self.lock:
            self.stats["staged_payloads"] += len(paths) + len(arrays)

    # ---- H2D + kernels + D2H (one thread owns the handle) ---------------------------
    def _launch_loop(self):
        try:
            while True:
                item = self.launch_q.get()
                if item is None:
                    self.done_q.put(None)
                    return
                batch, slot, futs, used_in, off_status, off_power, out_total, t0 = item
                for f in futs:
                    f.result()
                t1 = time.perf_counter()
                good = [j for j in batch if j.error is None]
                if good:
                    self._launch(good, len(batch), slot, used_in, off_status, off_power, out_total)
                t2 = time.perf_counter()
                with self.lock:
                    self.stats["t_read"] += t1 - t0
                    self.stats["t_launch"] += t2 - t1
                    self.stats["bytes_in"] += used_in
                    self.stats["bytes_out"] += out_total
                self.done_q.put((batch, slot, off_status, off_power, bool(good)))
        except BaseException as e:  # surfaces in submit()/close()
            self.exc = e
            self.done_q.put(None)

    def _launch(self, jobs, n_all, slot, used_in, off_status, off_power, out_total):
        ctx, eng = self.ctx, self.engine
        dbg = os.environ.get("SETK_PIPE_DEBUG")
        tt = [time.perf_counter()]
        # the payload copies were enqueued on the copy-in stream by the reader threads --
        # or the slab goes over in one piece now
        if self.h2d == "batch" and not self.zero_copy:
            ctx.memcpy_h2d_async(slot.d_in, slot.staging().ctypes.data, used_in, self.s_in)
        ctx.event_record(slot.e_in, self.s_in)
        tt.append(time.perf_counter())
        C = jobs[0].C
        has_itf = jobs[0].itf is not None
        base_in, base_f32, base_out = slot.d_in, slot.d_f32, slot.d_out
        ctx.stream_wait_event(self.s_compute, slot.e_in)
        stream = self.s_compute
        pcm = [j for j in jobs if j.pcm16]
        direct = eng.pcm_direct_ok and len(pcm) == len(jobs)
        if pcm:
            # int16 frames -> planar int16 C x stride (or float32 C x N); sum(x0^2) of the k-th
            # converted utterance lands at off_power + 8 k of the out-slab (log line only)
            for k, j in enumerate(pcm):
                j.pw_idx = k
            ingest = ctx.pcm16_deinterleave_batch if direct else ctx.pcm16_to_float_batch
            ingest(C, [base_in + j.off_audio for j in pcm],
                   [j.N for j in pcm], [base_f32 + j.off_f32 for j in pcm],
                   power0=base_out + off_power, stream=stream)
        tt.append(time.perf_counter())
        # Kaldi CompressedMatrix masks: the archive's bytes came up as stored, expanded here
        cms = []
        for j in jobs:
            for p, off_raw, off_f32 in ((j.mask, j.off_mask, j.off_mask_f32),
                                        (j.itf, j.off_itf if j.itf is not None else 0, j.off_itf_f32)):
                if p is not None and p.cm is not None:
                    kind, vmin, vrange, rows, cols = p.cm
                    cms.append((kind, vmin, vrange, rows, cols, rows == self.F and cols != self.F,
                                base_in + off_raw, base_f32 + off_f32))
        if cms:
            ctx.kaldi_cm_decode_batch(cms, stream=stream)
        aptr = [(base_f32 + j.off_f32) if j.pcm16 else (base_in + j.off_audio) for j in jobs]
        mptr = [(base_in + j.off_mask) if j.off_mask_f32 is None else (base_f32 + j.off_mask_f32) for j in jobs]
        iptr = [(base_in + j.off_itf) if j.off_itf_f32 is None else (base_f32 + j.off_itf_f32)
                for j in jobs] if has_itf else None
        wptr = [base_out + j.off_out for j in jobs]
        kind = eng.opts_kw["kind"]
        if has_itf and kind == _ffi.BF_MPDR:
            iptr = None  # plain MPDR never reads the interferer mask
        flags = eng.base_flags | _ffi.FLAG_OUT_PCM16 | (0 if has_itf else _ffi.FLAG_CLAMP_MASK) | \
            (_ffi.FLAG_IN_PCM16 if direct else 0)
        opts = _ffi.BfOpts(flags=flags, **eng.opts_kw)
        # status of job i at off_status + 4 * i (positions within `jobs`)
        ctx.enhance_batch(opts, C, aptr, [j.N for j in jobs], mptr, iptr, wptr,
                          stream=stream, status_ptr=base_out + off_status)
        tt.append(time.perf_counter())
        ctx.event_record(slot.e_compute, self.s_compute)
        ctx.stream_wait_event(self.s_out, slot.e_compute)
        ctx.memcpy_d2h_async(slot.h_out, slot.d_out, out_total, self.s_out)
        ctx.event_record(slot.e_out, self.s_out)
        tt.append(time.perf_counter())
        if dbg:
            print("launch ms: h2d %.2f pcm %.2f enhance %.2f d2h %.2f" % tuple(
                1e3 * (b - a) for a, b in zip(tt, tt[1:])), file=sys.stderr)

    # ---- completion: wait for D2H, hand the samples to the writers ------------------
    def _complete_loop(self):
        """Drains done_q until the launcher's sentinel WHATEVER happens to a batch: a
        writer / sink failure is remembered (self.exc, raised by submit() / close()),
        the batch's slot still returns to the pool and the following batches are still
        consumed -- the launcher can always finish and close() can always join,
        independently of how depth and the queue sizes are chosen."""
        while True:
            item = self.done_q.get()
            if item is None:
                return
            batch, slot, off_status, off_power, launched = item
            try:
                self._complete_batch(batch, slot, off_status, off_power, launched)
            except BaseException as e:
                if self.exc is None:
                    self.exc = e
            finally:
                try:
                    if not launched:
                        self.ctx.stream_synchronize(self.s_in)  # copies of a batch that never launched
                    for m in slot.maps:
                        m.close(self.ctx)
                except BaseException as e:  # pragma: no cover
                    if self.exc is None:
                        self.exc = e
                slot.maps = []
                self.free_slots.put(slot)

    def _complete_batch(self, batch, slot, off_status, off_power, launched):
        t0 = time.perf_counter()
        if launched:
            self.ctx.event_synchronize(slot.e_out)
        t1 = time.perf_counter()
        good = [j for j in batch if j.error is None]
        status = np.frombuffer(slot.np_out[off_status:off_status + 4 * len(good)], dtype=np.int32)
        power = np.frombuffer(slot.np_out[off_power:off_power + 8 * len(good)], dtype=np.float64)
        futs = []
        gi = 0
        for j in batch:
            if j.error is not None:
                futs.append(self.writers.submit(self.sink, j.key, None, -1, j.error))
                continue
            if self.announce is not None:
                self.announce(j.key, float(power[j.pw_idx]) / max(j.N, 1) if j.pcm16 else j.power)
            pcm = np.frombuffer(slot.np_out[j.off_out:j.off_out + 2 * j.L], dtype=np.int16)
            futs.append(self.writers.submit(self.sink, j.key, pcm, int(status[gi]), None))
            gi += 1
        ok, first = 0, None
        for f in futs:  # wait for EVERY writer before the slot's memory is reused
            try:
                ok += 1 if f.result() else 0
            except BaseException as e:
                first = first or e
        t2 = time.perf_counter()
        with self.lock:
            self.num_done += ok
            self.stats["batches"] += 1
            self.stats["utts"] += len(batch)
            self.stats["t_d2h_wait"] += t1 - t0
            self.stats["t_write"] += t2 - t1
        if first is not None:
            raise first

    def close(self):
        """Flush, wait for everything in flight, return (num_done, stats)."""
        try:
            if not self.exc:
                self._dispatch()
        finally:
            self.launch_q.put(None)
            self.launcher.join()
            self.completer.join()
            self.readers.shutdown()
            self.writers.shutdown()
            # every wav is closed: the clock stops here, then the slabs go back
            wall = (time.perf_counter() - self.t_first) if self.t_first else 0.0
            t0 = time.perf_counter()
            self._release()
            self.stats["t_release"] = time.perf_counter() - t0
        if self.exc:
            raise self.exc
        st = dict(self.stats)
        st["wall_s"] = wall
        st["read_threads"] = self.read_threads
        st["read_mode"] = "zero_copy" if self.zero_copy else READ_MODE
        st["depth"] = self.depth
        st["batch_utts"] = self.batch_utts
        st["batch_mb"] = self.batch_bytes >> 20
        return self.num_done, st


    def _release(self):
        """Everything has drained: give the slabs, events and streams back."""
        try:
            for st in (self.s_in, self.s_compute, self.s_out):
                self.ctx.stream_synchronize(st)
            for slot in self.slots:
                slot.release()
            for st in (self.s_in, self.s_compute, self.s_out):
                self.ctx.stream_destroy(st)
        except Exception as e:  # the original failure, if any, is the one to report
            if self.exc is None:
                self.exc = e
        self.slots = []


# ----------------------------------------------------------------------------
# scp entries -> sources
# ----------------------------------------------------------------------------
_NO_LOCK = contextlib.nullcontext()


def wav_source(wav_reader, key, files, lock=None):
    """What submit() takes as `audio` for an entry of a WaveReader table: the PCM16
    payload as it lies in the file when the entry is ONE 16-bit PCM file (plain
    path or path.ark:offset), else the decoded C x N float32 array.  lock: held around
    the table reader's own decode (its archive handles are not for concurrent use) when
    several threads prepare sources."""
    import glob
    fname = wav_reader.index_dict[key].rstrip()
    if wav_reader.normalize and fname and fname[-1] != "|":
        hits = glob.glob(fname)
        if ":" in fname and not hits:
            hits = [fname]
        if len(hits) == 1:
            path, offset = hits[0], 0
            if not os.path.exists(path) and ":" in path:
                path, off = path.rsplit(":", 1)
                offset = int(off)
            info, data_off = probe_wav(files, path, offset)
            if info["sr"] != wav_reader.sr:
                raise RuntimeError(f"Expect sr={wav_reader.sr} of {path}, get {info['sr']} instead")
            if info["fmt"] == wavio.WAVE_FORMAT_PCM and info["bits"] == 16:
                ch = info["channels"]
                n = info["data_bytes"] // (2 * ch)
                size = os.fstat(files.get(path)).st_size
                n = min(n, max(0, (size - data_off) // (2 * ch)))
                return (Payload(path=path, offset=data_off, nbytes=2 * ch * n, fsize=size),
                        ch, n)
    with lock or _NO_LOCK:
        samps = wav_reader.read(key)
    return samps[None] if samps.ndim == 1 else samps


def mask_source(reader, key, files, num_bins, lock=None):
    """Payload of the float32 [T][F] rows when the mask is stored that way (a
    C-ordered float32 .npy, or a Kaldi FM matrix), else the loaded array."""
    from .libs.data_handler import NumpyReader, ScriptReader
    try:
        if isinstance(reader, NumpyReader):
            path = reader.index_dict[key]
            shape, dt, fortran, off = probe_npy(files, path)
            if dt == np.dtype("<f4") and not fortran and len(shape) == 2 and shape[1] == num_bins:
                return Payload(path=path, offset=off, nbytes=4 * shape[0] * shape[1],
                               fsize=os.fstat(files.get(path)).st_size)
        elif isinstance(reader, ScriptReader):
            path, offset = reader.locate(key)
            hit = probe_kaldi_matrix(files, path, offset)
            if hit and hit[2] == np.dtype("<f4") and hit[1] == num_bins:
                return Payload(path=path, offset=hit[3], nbytes=4 * hit[0] * hit[1],
                               fsize=os.fstat(files.get(path)).st_size)
            if hit is None and DEVICE_CM_DECODE:
                cm = probe_kaldi_compressed(files, path, offset)
                if cm and num_bins in (cm[3], cm[4]):
                    return Payload(path=path, offset=cm[5], nbytes=cm[6], cm=cm[:5])
    except (OSError, ValueError, KeyError, SyntaxError):
        pass
    with lock or _NO_LOCK:
        return reader[key]
