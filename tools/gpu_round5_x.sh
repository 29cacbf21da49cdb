cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "cgmm or clustering or sixteen or doc_pipeline" --maxfail=8 2>&1 | grep -v INFO | tail -12
python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|INFO"
import sys, json
sys.path.insert(0, "tools")
import bench_consumers as b
r = b.run()
for k in ("cgmm_general_k3_4ch", "cgmm_general_k2_12ch"):
    print(k, {kk: v for kk, v in r[k].items() if kk != "workload"})
PY
