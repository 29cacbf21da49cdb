cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_cgmm.py tests/test_gpu_baseline_sizes.py -q -p no:cacheprovider -k "cgmm or cfg4" 2>&1 | tail -6 | cut -c1-250
python tools/bench_cgmm.py --utts 125 --seconds 30 --steps 3 2>/dev/null | tail -1 | cut -c1-300
