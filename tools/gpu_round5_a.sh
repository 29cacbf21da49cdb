set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > gpurun_out/r5a_pytest.log 2>&1
tail -15 gpurun_out/r5a_pytest.log
(time timeout 600 python bench.py) > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
tail -5 gpurun_out/r5a_bench.err
head -c 6000 gpurun_out/r5a_bench.json
