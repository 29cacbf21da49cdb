cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5t
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "consumers or wpd or wpe" --maxfail=10 2>&1 | tail -3
for i in 1 2; do
python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|INFO"
import json, sys
sys.path.insert(0, "tools")
import bench_consumers as b
r = b.run()
for k in ("df_on_mask_4ch_resident", "wpd_4ch_resident"):
    print(k, {kk: v for kk, v in r[k].items() if kk != "workload"})
PY
done
