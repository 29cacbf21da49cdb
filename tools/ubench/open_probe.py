import os, time, shutil, numpy as np
d='/dev/shm/openprobe'; shutil.rmtree(d, ignore_errors=True); os.makedirs(d+'/wav'); os.makedirs(d+'/mask')
a=np.random.randint(0,255,7_680_044,dtype=np.uint8).tobytes(); b=np.random.randint(0,255,1_928_656,dtype=np.uint8).tobytes()
n=1024
open(f"{d}/wav/u0.wav","wb").write(a); open(f"{d}/mask/u0.npy","wb").write(b)
for i in range(1,n):
    shutil.copyfile(f"{d}/wav/u0.wav", f"{d}/wav/u{i}.wav"); shutil.copyfile(f"{d}/mask/u0.npy", f"{d}/mask/u{i}.npy")
def timed(paths, keep):
    ts=[]; fds=[]
    for p in paths:
        t0=time.perf_counter(); fd=os.open(p,os.O_RDONLY); ts.append(time.perf_counter()-t0)
        if keep: fds.append(fd)
        else: os.close(fd)
    for fd in fds: os.close(fd)
    ts=np.array(ts)*1e6
    return f"mean {ts.mean():.1f} us, median {np.median(ts):.1f}, p99 {np.percentile(ts,99):.1f}, max {ts.max():.0f}"
w=[f"{d}/wav/u{i}.wav" for i in range(n)]; m=[f"{d}/mask/u{i}.npy" for i in range(n)]
print("wav only, close each :", timed(w, False))
print("npy only, close each :", timed(m, False))
print("wav only, keep open  :", timed(w, True))
print("npy only, keep open  :", timed(m, True))
alt=[x for pair in zip(w,m) for x in pair]
print("alternating, keep open:", timed(alt, True))
import resource; print("RLIMIT_NOFILE", resource.getrlimit(resource.RLIMIT_NOFILE))
shutil.rmtree(d)
