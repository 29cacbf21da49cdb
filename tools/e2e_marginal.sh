#!/usr/bin/env bash
# Marginal end-to-end cost per utterance of the streaming CLI: pipeline clock at 256 and 1024
# utterances (8-ch 30 s PCM16 wav + npy mask in /dev/shm) for each flag set given as an argument.
#   bash tools/e2e_marginal.sh "--beamformer mvdr" "--pipeline-depth 5 --beamformer mvdr" ...
for flags in "$@"; do
  a=$(python tools/e2e_sweep.py 256 "$flags" 2>/dev/null | grep -o "pipeline [0-9.]* s" | grep -o "[0-9.]*")
  b=$(python tools/e2e_sweep.py 1024 "$flags" 2>/dev/null | tee /tmp/last_sweep.txt | grep -o "pipeline [0-9.]* s" | grep -o "[0-9.]*")
  python -c "print('[$flags] pipeline', $a, $b, 's -> marginal %.3f ms/utt' % (1e3*($b-$a)/768))"
  grep -o "{.*}" /tmp/last_sweep.txt
done
