export TMPDIR=/tmp
python tools/_cm_probe.py 2>&1 | grep -v amdgpu.ids | tail -8
bash tools/gpu_full.sh round6_g tests
