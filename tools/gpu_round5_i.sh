cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round5
export TMPDIR=/tmp
O=gpurun_out/round5
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_cgmm.py -q -p no:cacheprovider -k "strict_reference or non_finite or rccl" 2>&1 | tail -8 | cut -c1-250
PLIST="1 2 4" bash tools/e2e_steady.sh 2048 10 > /dev/null 2>&1; cp gpurun_out/e2e_steady.txt $O/e2e_steady.txt; cat $O/e2e_steady.txt
SETK_PCM16_DIRECT=0 PLIST="1 2" bash tools/e2e_steady.sh 2048 10 > /dev/null 2>&1; cp gpurun_out/e2e_steady.txt $O/e2e_steady_float_twin.txt; cat $O/e2e_steady_float_twin.txt
