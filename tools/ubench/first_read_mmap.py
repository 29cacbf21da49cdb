#!/usr/bin/env python
"""First read of freshly written /dev/shm files: preadv into a buffer (what the pipeline does)
against mmap + copy out of the mapping (plain, MAP_POPULATE, madvise WILLNEED / HUGEPAGE), and
against a writer that is the SAME process (pages first touched by the reader's own threads).
    python tools/ubench/first_read_mmap.py"""
import mmap
import os
import shutil
import threading
import time

import numpy as np

D = "/dev/shm/setk_first_read_mmap"
N, SZ = 128, 9_600_000


def make():
    shutil.rmtree(D, ignore_errors=True)
    os.makedirs(D)
    blob = np.random.randint(0, 255, SZ, dtype=np.uint8).tobytes()
    for i in range(N):
        with open(f"{D}/{i}.bin", "wb") as f:
            f.write(blob)


def run(nt, mode):
    dst = [np.ones(SZ, dtype=np.uint8) for _ in range(nt)]

    def work(k):
        for i in range(k, N, nt):
            fd = os.open(f"{D}/{i}.bin", os.O_RDONLY)
            if mode == "preadv":
                os.preadv(fd, [memoryview(dst[k])], 0)
            else:
                flags = mmap.MAP_SHARED | (mmap.MAP_POPULATE if mode == "mmap_populate" else 0)
                m = mmap.mmap(fd, SZ, flags=flags, prot=mmap.PROT_READ)
                if mode == "mmap_willneed":
                    m.madvise(mmap.MADV_WILLNEED)
                if mode == "mmap_seq":
                    m.madvise(mmap.MADV_SEQUENTIAL)
                np.copyto(dst[k], np.frombuffer(m, dtype=np.uint8))
                m.close()
            os.close(fd)
    th = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return N * SZ / (time.perf_counter() - t0) / 1e9


def main():
    try:
        for mode in ("preadv", "mmap", "mmap_populate", "mmap_willneed", "mmap_seq"):
            for nt in (1, 6, 12):
                make()
                first = run(nt, mode)
                second = run(nt, mode)
                print(f"{mode:14s} threads={nt:2d}: first read {first:5.1f} GB/s, second {second:5.1f}", flush=True)
    finally:
        shutil.rmtree(D, ignore_errors=True)


if __name__ == "__main__":
    main()
