#!/usr/bin/env python
"""First read of freshly written /dev/shm files (what the end-to-end bench does) against a second
read, by reader threads, with and without POSIX_FADV_NOREUSE on the descriptor.
    python tools/ubench/first_read.py"""
import os
import shutil
import threading
import time

import numpy as np

D = "/dev/shm/setk_first_read"
N, SZ = 128, 9_600_000


def make():
    shutil.rmtree(D, ignore_errors=True)
    os.makedirs(D)
    blob = np.random.randint(0, 255, SZ, dtype=np.uint8).tobytes()
    for i in range(N):
        with open(f"{D}/{i}.bin", "wb") as f:
            f.write(blob)


def run(nt, noreuse):
    dst = [np.ones(SZ, dtype=np.uint8) for _ in range(nt)]

    def work(k):
        for i in range(k, N, nt):
            fd = os.open(f"{D}/{i}.bin", os.O_RDONLY)
            if noreuse:
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_NOREUSE)
            os.preadv(fd, [memoryview(dst[k])], 0)
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
        for noreuse in (False, True):
            for nt in (1, 6, 12, 32):
                make()
                first = run(nt, noreuse)
                second = run(nt, noreuse)
                third = run(nt, noreuse)
                print(f"noreuse={noreuse} threads={nt}: first read {first:.1f} GB/s, second {second:.1f}, "
                      f"third {third:.1f}", flush=True)
    finally:
        shutil.rmtree(D, ignore_errors=True)


if __name__ == "__main__":
    main()
