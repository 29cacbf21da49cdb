#!/usr/bin/env bash
# Sanitizer build of the host side of libsetk_hip.so (the C ABI, the launch wrappers, the
# descriptor-table builders): AddressSanitizer + UBSan on the host code, device code
# compiled as in the product.   bash tools/asan_build.sh   ->  _abl/libsetk_asan.so
# Run a test selection under it on a GPU box with tools/asan_run.sh.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT="$ROOT/_abl/asan"; mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -shared-libsan"
pids=""
for u in pass1 pass2 solve modular cgmm cgmm_bin wpe capi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize \
    -Wno-unused-result $SAN -c "$ROOT/setk_amd/csrc/$u.hip" -o "$OUT/$u.o" &
  pids="$pids $!"
done
for p in $pids; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -o "$ROOT/_abl/libsetk_asan.so" "$OUT"/*.o
echo "_abl/libsetk_asan.so"
