cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api.py -q -p no:cacheprovider -k "consumers" 2>&1 | tail -15 | cut -c1-300
python tools/bench_consumers.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k in ('df_on_mask_4ch','df_on_mask_4ch_resident'): print(k, {a:b for a,b in d[k].items() if a!='workload'})
"
