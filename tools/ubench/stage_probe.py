#!/usr/bin/env python
"""Where does the staged input path of setk_amd/pipeline.py lose time?  By thread count:
  copy->pinned   numpy copy of RAM into a torch page-locked slab (is pinned memory slow to write?)
  read->plain    preadv of /dev/shm files into ordinary memory
  read->pinned   preadv into the page-locked slab (what the reader threads do)
  +h2d           the same followed by one hipMemcpyAsync per payload on one stream, then a sync
    python tools/ubench/stage_probe.py"""
import os
import shutil
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from setk_amd import _ffi  # noqa: E402

D = "/dev/shm/setk_stage_probe"
N, SZ = 96, 9_600_000


def threads(nt, fn):
    th = [threading.Thread(target=fn, args=(k,)) for k in range(nt)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


def main():
    os.makedirs(D, exist_ok=True)
    try:
        blob = np.random.randint(0, 255, SZ, dtype=np.uint8).tobytes()
        for i in range(N):
            with open(f"{D}/{i}.bin", "wb") as f:
                f.write(blob)
        fds = [os.open(f"{D}/{i}.bin", os.O_RDONLY) for i in range(N)]
        dev = torch.device("cuda", 0)
        ctx = _ffi.Context(0)
        pin = torch.empty(N * SZ, dtype=torch.uint8, pin_memory=True)
        pin.fill_(1)
        pbuf = pin.numpy()
        plain = np.ones(N * SZ, dtype=np.uint8)
        src = np.ones(SZ, dtype=np.uint8)
        d_in = torch.empty(N * SZ, dtype=torch.uint8, device=dev)
        s_in = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        for nt in (1, 4, 8, 16, 32):
            def copy_pinned(k):
                for i in range(k, N, nt):
                    np.copyto(pbuf[i * SZ:(i + 1) * SZ], src)

            def read_into(buf):
                def fn(k):
                    for i in range(k, N, nt):
                        os.preadv(fds[i], [memoryview(buf[i * SZ:(i + 1) * SZ])], 0)
                return fn

            def read_h2d(k):
                for i in range(k, N, nt):
                    v = pbuf[i * SZ:(i + 1) * SZ]
                    os.preadv(fds[i], [memoryview(v)], 0)
                    ctx.memcpy_h2d_async(d_in.data_ptr() + i * SZ, v.ctypes.data, SZ, s_in.cuda_stream)
            r = {}
            r["copy->pinned"] = N * SZ / threads(nt, copy_pinned) / 1e9
            r["read->plain"] = N * SZ / threads(nt, read_into(plain)) / 1e9
            r["read->pinned"] = N * SZ / threads(nt, read_into(pbuf)) / 1e9
            t0 = time.perf_counter()
            dt_issue = threads(nt, read_h2d)
            s_in.synchronize()
            dt_all = time.perf_counter() - t0
            r["read->pinned+h2d (threads done)"] = N * SZ / dt_issue / 1e9
            r["read->pinned+h2d (copies landed)"] = N * SZ / dt_all / 1e9
            print(nt, "threads:", {k: round(v, 1) for k, v in r.items()}, "GB/s", flush=True)
        # one big copy of the whole slab, for reference
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.memcpy_h2d_async(d_in.data_ptr(), pbuf.ctypes.data, N * SZ, s_in.cuda_stream)
        s_in.synchronize()
        print("one hipMemcpyAsync of the slab: %.1f GB/s" % (N * SZ / (time.perf_counter() - t0) / 1e9))
    finally:
        shutil.rmtree(D, ignore_errors=True)


if __name__ == "__main__":
    main()
