cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; mkdir -p gpurun_out/round5
timeout 900 python tools/stress.py 150 5 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/round5/stress_150_seed5.txt
timeout 900 python tools/stress.py 150 6 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/round5/stress_150_seed6.txt
