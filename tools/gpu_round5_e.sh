cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
python tools/ubench/read_small.py 2560044 1024 2>&1 | tee gpurun_out/round5/read_small_2p5MB.txt
python tools/ubench/read_small.py 643200 2048 2>&1 | tee gpurun_out/round5/read_small_0p6MB.txt
python tools/error_budget.py 2>&1 | grep -v Warn | tee gpurun_out/round5/error_budget.txt
