#!/usr/bin/env python
"""Per-file system-call costs on /dev/shm for freshly written files (what the CLI's header probing
does once per input file): open, pread of the first 4 KB, fstat, a bulk pread, close; first and
second pass."""
import os
import shutil
import time

import numpy as np

D = "/dev/shm/setk_syscalls"
N, SZ = 1024, 7_680_044


def main():
    os.makedirs(D, exist_ok=True)
    try:
        blob = np.random.randint(0, 255, SZ, dtype=np.uint8).tobytes()
        for i in range(N):
            with open(f"{D}/f{i}", "wb") as f:
                f.write(blob)
        buf = bytearray(SZ)
        for rnd in (1, 2):
            t = dict(open=0.0, pread4k=0.0, fstat=0.0, bulk=0.0, close=0.0)
            for i in range(N):
                t0 = time.perf_counter()
                fd = os.open(f"{D}/f{i}", os.O_RDONLY)
                t1 = time.perf_counter()
                os.pread(fd, 4096, 0)
                t2 = time.perf_counter()
                os.fstat(fd)
                t3 = time.perf_counter()
                os.preadv(fd, [memoryview(buf)], 0)
                t4 = time.perf_counter()
                os.close(fd)
                t5 = time.perf_counter()
                t["open"] += t1 - t0
                t["pread4k"] += t2 - t1
                t["fstat"] += t3 - t2
                t["bulk"] += t4 - t3
                t["close"] += t5 - t4
            print(f"pass {rnd}: per file (us):", {k: round(1e6 * v / N, 1) for k, v in t.items()},
                  f"bulk = {N * SZ / t['bulk'] / 1e9:.1f} GB/s (one thread)")
    finally:
        shutil.rmtree(D, ignore_errors=True)


if __name__ == "__main__":
    main()
