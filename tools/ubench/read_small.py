#!/usr/bin/env python
"""Per-payload cost of the pipeline's read stage for SHORT utterances (8-ch 10 s: 2.56 MB of
frames + 0.64 MB of mask rows per utterance): preadv into a slab against mmap + copy (+ the
munmap and its TLB shootdown), fresh files and re-read files, 6 / 12 / 24 threads.
    python tools/ubench/read_small.py [bytes=2560044] [files=1024]"""
import mmap
import os
import shutil
import sys
import threading
import time

import numpy as np

D = "/dev/shm/setk_read_small"
SZ = int(sys.argv[1]) if len(sys.argv) > 1 else 2_560_044
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024


def make():
    shutil.rmtree(D, ignore_errors=True)
    os.makedirs(D)
    blob = np.random.randint(0, 255, SZ, dtype=np.uint8).tobytes()
    for i in range(N):
        with open(f"{D}/{i}.bin", "wb") as f:
            f.write(blob)


PINNED = os.environ.get("READ_SMALL_PINNED") == "1"
_ctx = None


def _slab(nbytes):
    """plain numpy memory, or (READ_SMALL_PINNED=1) the page-locked memory the pipeline copies into"""
    global _ctx
    if not PINNED:
        return np.ones(nbytes, dtype=np.uint8)
    if _ctx is None:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        from setk_amd import _ffi
        _ffi.set_torch_free()
        _ctx = _ffi.Context(0)
    addr, view = _ctx.host_alloc(nbytes)
    view[:] = 1
    return view


def run_native(nt, per_call=64):
    """the library's reader pool (setk_host_read_payloads): calls of `per_call` payloads, as the
    pipeline issues them per batch"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from setk_amd import _ffi
    _ffi.set_torch_free()
    slab = _slab(per_call * SZ)
    base = slab.ctypes.data
    t0 = time.perf_counter()
    for i0 in range(0, N, per_call):
        idx = range(i0, min(N, i0 + per_call))
        st = _ffi.host_read_payloads([f"{D}/{i}.bin" for i in idx], [0] * len(idx), [SZ] * len(idx),
                                     [base + k * SZ for k in range(len(idx))], nt,
                                     int(os.environ.get("READ_SMALL_MMAP_MIN_KB", "256")) << 10)
        assert not any(st), st
    dt = time.perf_counter() - t0
    return N * SZ / dt / 1e9, 1e6 * dt * nt / N


def run(nt, mode):
    if mode == "native":
        return run_native(nt)
    slab = _slab(nt * SZ)

    def work(k):
        dst = slab[k * SZ:(k + 1) * SZ]
        mv = memoryview(dst)
        for i in range(k, N, nt):
            fd = os.open(f"{D}/{i}.bin", os.O_RDONLY)
            if mode == "preadv":
                got = 0
                while got < SZ:
                    got += os.preadv(fd, [mv[got:]], got)
            else:
                m = mmap.mmap(fd, SZ, flags=mmap.MAP_SHARED | (mmap.MAP_POPULATE if mode == "mmap_populate" else 0),
                              prot=mmap.PROT_READ)
                if mode == "mmap_seq":
                    m.madvise(mmap.MADV_SEQUENTIAL)
                np.copyto(dst, np.frombuffer(m, dtype=np.uint8))
                m.close()
            os.close(fd)
    th = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return N * SZ / dt / 1e9, 1e6 * dt * nt / N


def main():
    print(f"# {N} files of {SZ} bytes in /dev/shm; GB/s aggregate (us per file per thread); destination: "
          f"{'page-locked (setk_host_alloc)' if PINNED else 'numpy'}")
    try:
        for mode in (("mmap_seq", "native") if PINNED else ("preadv", "mmap_seq", "mmap_populate", "native")):
            for nt in (6, 12, 24):
                make()
                a, ua = run(nt, mode)
                b, ub = run(nt, mode)
                print(f"{mode:14s} threads={nt:2d}: first read {a:5.1f} GB/s ({ua:6.0f} us), re-read {b:5.1f} GB/s ({ub:6.0f} us)",
                      flush=True)
    finally:
        shutil.rmtree(D, ignore_errors=True)


if __name__ == "__main__":
    main()
