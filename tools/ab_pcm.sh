#!/usr/bin/env bash
# A/B of library builds on the 16-bit PCM form of the timed step (bench.py's int16_ingest leg, with
# its counters): bash tools/ab_pcm.sh <rounds> lib1.so lib2.so ...   ("default" = the in-tree build)
# Interleaved rounds; prints per build: step from frames, planar-resident step, stage times, counter
# traffic over algorithmic bytes of both streaming kernels, and the exact-parity flag.
R=$1; shift
for r in $(seq $R); do
  for L in "$@"; do
    if [ "$L" = default ]; then unset SETK_LIB; else export SETK_LIB=$PWD/$L; fi
    python bench.py --steps 100 --warmup 30 --cpu-sample 0 --full-batch 0 --e2e-utts 0 2>/dev/null | tail -1 > /tmp/ab.json
    python - "$L" $r <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
i = d["int16_ingest"]; rf = i["roofline"]
print(f"AB round {sys.argv[2]} {sys.argv[1]}: f32 step {d['ms_per_step']} | pcm from frames {i['ms_per_step']} planar {i['enhance_only_ms']} "
      f"stage1 {i['stage_ms']['stft_covar']} stage3 {i['stage_ms']['beamform_istft']} | traffic/alg pass1 {rf.get('traffic_over_algorithmic')} "
      f"pass2 {rf['pass2'].get('traffic_over_algorithmic')} | bit-identical to f32 path {i['bit_identical_to_float32_path_on_pcm_over_32768']} "
      f"| f32 stage3 {d['stage_ms']['beamform_istft']} f32 pass2 traffic/alg {d.get('pass2_traffic_over_algorithmic')} "
      f"| pcm pass2 read/write GB {((rf.get('pmc') or {}).get('pass2_pcm') or {}).get('hbm_read_bytes', 0) / 1e9:.3f} / "
      f"{((rf.get('pmc') or {}).get('pass2_pcm') or {}).get('hbm_write_bytes', 0) / 1e9:.3f} (alg {rf['pass2']['alg_bytes_per_launch'] / 1e9:.3f})")
PY
  done
done
