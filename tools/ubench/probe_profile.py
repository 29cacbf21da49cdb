import os, time, shutil, sys, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.getcwd())
from setk_amd import synth
from setk_amd.libs import wavio
from setk_amd.libs.data_handler import WaveReader, NumpyReader
from setk_amd.pipeline import OpenFiles, wav_source, mask_source
d='/dev/shm/pyov'; shutil.rmtree(d, ignore_errors=True); os.makedirs(d+'/wav'); os.makedirs(d+'/mask')
C,N,T=8,480000,1876
wavio.write_pcm16(f"{d}/wav/u0.wav", wavio.float_to_pcm16(synth.synth_utterance(0,C,N).T),16000)
np.save(f"{d}/mask/u0.npy", np.random.rand(T,257).astype(np.float32))
n=1024
with open(f"{d}/wav.scp","w") as ws, open(f"{d}/mask.scp","w") as ms:
    for i in range(n):
        if i:
            shutil.copyfile(f"{d}/wav/u0.wav", f"{d}/wav/u{i}.wav"); shutil.copyfile(f"{d}/mask/u0.npy", f"{d}/mask/u{i}.npy")
        ws.write(f"u{i} {d}/wav/u{i}.wav\n"); ms.write(f"u{i} {d}/mask/u{i}.npy\n")
wr=WaveReader(f"{d}/wav.scp", sr=16000); mr=NumpyReader(f"{d}/mask.scp")
files=OpenFiles()
pr=cProfile.Profile()
t0=time.perf_counter()
pr.enable()
for k in wr.index_keys:
    a=wav_source(wr,k,files); m=mask_source(mr,k,files,257)
pr.disable()
t1=time.perf_counter()
print("probe wav+mask per utt: %.1f us (with profiler)"%(1e6*(t1-t0)/n))
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
shutil.rmtree(d)
