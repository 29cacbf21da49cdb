#!/usr/bin/env python
"""Can the H2D copy DMA straight out of the page cache?  mmap a wav in /dev/shm,
hipHostRegister the mapping, hipMemcpyAsync from it -- against the product's route
(preadv into a pinned slab, then one copy).  Prints per-file costs."""
import ctypes
import mmap
import os
import sys
import time

import numpy as np
import torch

N, SZ = 64, 7680044
d = "/dev/shm/setk_hostreg"
os.makedirs(d, exist_ok=True)
blob = np.random.randint(0, 255, SZ, dtype=np.uint8).tobytes()
for i in range(N):
    with open(f"{d}/f{i}", "wb") as f:
        f.write(blob)
rt = torch.cuda.cudart()
dev = torch.device("cuda", 0)
dst = torch.empty(N * SZ, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()

# (a) product route: preadv into pinned memory (one thread here), one H2D
pin = torch.empty(N * SZ, dtype=torch.uint8, pin_memory=True)
buf = pin.numpy()
fds = [os.open(f"{d}/f{i}", os.O_RDONLY) for i in range(N)]
t0 = time.perf_counter()
for i in range(N):
    mv = memoryview(buf[i * SZ:(i + 1) * SZ])
    got = 0
    while got < SZ:
        got += os.preadv(fds[i], [mv[got:]], got)
t1 = time.perf_counter()
dst.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"preadv -> pinned: {1e3 * (t1 - t0) / N:.3f} ms/file ({N * SZ / (t1 - t0) / 1e9:.1f} GB/s, 1 thread); "
      f"H2D of the slab: {1e3 * (t2 - t1) / N:.3f} ms/file ({N * SZ / (t2 - t1) / 1e9:.1f} GB/s)")

# (b) register the page-cache pages
for flags, name in ((0, "default"), (8, "read-only")):
    try:
        reg = cp = unreg = 0.0
        ok = True
        for i in range(N):
            mm = mmap.mmap(fds[i], 0, prot=mmap.PROT_READ)
            a = np.frombuffer(mm, dtype=np.uint8)
            ptr = a.ctypes.data
            ta = time.perf_counter()
            rc = rt.cudaHostRegister(ptr, SZ, flags)
            tb = time.perf_counter()
            if int(rc) != 0:
                print(f"hipHostRegister({name}) failed with {rc}")
                ok = False
                del a
                mm.close()
                break
            src = torch.from_numpy(np.frombuffer(mm, dtype=np.uint8)) if False else None
            # raw async copy through the runtime
            lib = ctypes.CDLL("libamdhip64.so")
            tc0 = time.perf_counter()
            e = lib.hipMemcpyAsync(ctypes.c_void_p(dst.data_ptr() + i * SZ), ctypes.c_void_p(ptr),
                                   ctypes.c_size_t(SZ), 1, ctypes.c_void_p(0))
            lib.hipDeviceSynchronize()
            tc1 = time.perf_counter()
            rt.cudaHostUnregister(ptr)
            tc2 = time.perf_counter()
            reg += tb - ta
            cp += tc1 - tc0
            unreg += tc2 - tc1
            del a
            mm.close()
        if ok:
            print(f"mmap + hipHostRegister({name}): register {1e3 * reg / N:.3f} ms/file, copy "
                  f"{1e3 * cp / N:.3f} ms/file ({SZ * N / cp / 1e9:.1f} GB/s), unregister "
                  f"{1e3 * unreg / N:.3f} ms/file")
    except Exception as ex:  # noqa: BLE001
        print(f"hipHostRegister({name}): {type(ex).__name__}: {ex}")
import shutil
shutil.rmtree(d, ignore_errors=True)
