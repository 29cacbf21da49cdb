#!/usr/bin/env python
"""How fast can this host move file bytes out of /dev/shm?  preadv into a private buffer
and mmap + memcpy, by thread count (the reader threads of setk_amd/pipeline.py do the first).
    python tools/ubench/shm_read.py"""
import mmap
import os
import shutil
import threading
import time

import numpy as np

D = "/dev/shm/setk_rdtest"
N, SZ = 96, 9_600_000


def run(kind, nt):
    dst = [np.ones(SZ, dtype=np.uint8) for _ in range(nt)]

    def work(k):
        for i in range(k, N, nt):
            fd = os.open(f"{D}/{i}.bin", os.O_RDONLY)
            if kind == "preadv":
                os.preadv(fd, [memoryview(dst[k])], 0)
            else:
                m = mmap.mmap(fd, SZ, flags=mmap.MAP_SHARED | (mmap.MAP_POPULATE if kind == "mmap+populate" else 0),
                              prot=mmap.PROT_READ)
                src = np.frombuffer(m, dtype=np.uint8)
                np.copyto(dst[k], src)
                del src
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
    os.makedirs(D, exist_ok=True)
    try:
        buf = np.random.randint(0, 255, SZ, dtype=np.uint8).tobytes()
        for i in range(N):
            with open(f"{D}/{i}.bin", "wb") as f:
                f.write(buf)
        for kind in ("preadv", "mmap", "mmap+populate"):
            print(kind, {nt: round(run(kind, nt), 1) for nt in (1, 2, 4, 8, 16, 32)}, "GB/s by threads", flush=True)
    finally:
        shutil.rmtree(D, ignore_errors=True)


if __name__ == "__main__":
    main()
