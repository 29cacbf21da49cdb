#!/usr/bin/env bash
# Build an experimental variant of one translation unit into _abl/libsetk_<name>.so
#   bash tools/mk_abl.sh <name> <unit, e.g. pass1> [extra hipcc flags...]
# NOILP=1 drops the max-ilp scheduler flag, SCHED=<name> selects another strategy.  The other objects come from the
# regular build (run python -m setk_amd.build first).
set -e
NAME=$1; UNIT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/_abl"
ILP="-mllvm -amdgpu-sched-strategy=max-ilp"; [ -n "$NOILP" ] && ILP=""; [ -n "$SCHED" ] && ILP="-mllvm -amdgpu-sched-strategy=$SCHED"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize $ILP \
  -Wno-unused-result "$@" -c "$ROOT/setk_amd/csrc/$UNIT.hip" -o "$ROOT/_abl/${UNIT}_$NAME.o"
OBJS=""
for u in pass1 pass1_mc pass2 pass2_mc solve modular cgmm cgmm_bin cgmm_k wpe comm hostio capi; do
  if [ "$u" = "$UNIT" ]; then OBJS="$OBJS $ROOT/_abl/${UNIT}_$NAME.o"; else OBJS="$OBJS $ROOT/setk_amd/csrc/_obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/_abl/libsetk_$NAME.so" $OBJS
echo "_abl/libsetk_$NAME.so"
