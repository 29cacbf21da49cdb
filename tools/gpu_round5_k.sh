cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_wpe.py -q -p no:cacheprovider -k "consumers or batch_wpd" 2>&1 | tail -25 | cut -c1-300
