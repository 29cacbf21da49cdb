cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api.py -q -p no:cacheprovider -k "streaming or pipeline or consumers" --maxfail=5 2>&1 | tail -5
