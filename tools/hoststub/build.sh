#!/usr/bin/env bash
# Host-only sanitizer build of libsetk_hip.so against the HIP stand-in of hip_stub.cpp:
# AddressSanitizer + UBSan over the C ABI, the launch wrappers and the descriptor-table
# builders, no device code, no GPU needed.   bash tools/hoststub/build.sh -> _abl/libsetk_hostasan.so
# (A sanitizer build against the REAL runtime cannot run in this image: ROCm's ASAN runtime
# intercepts hsa_amd_memory_pool_allocate and needs the ASAN flavour of the ROCm libraries,
# /opt/rocm/lib/asan, which is not installed -- tools/asan_build.sh keeps that recipe.)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
# PLAIN=1: the same without the sanitizers (-O2) -> _abl/libsetk_hoststub.so, for profiling the
# host pipeline on a machine without a GPU (kernels do nothing, copies are memcpy)
if [ -n "$PLAIN" ]; then
  OUT="$ROOT/_abl/hoststub"; LIBOUT="$ROOT/_abl/libsetk_hoststub.so"; SAN="-O2"
else
  OUT="$ROOT/_abl/hostasan"; LIBOUT="$ROOT/_abl/libsetk_hostasan.so"
  SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -shared-libsan"
fi
mkdir -p "$OUT"
CLANG=/opt/rocm/lib/llvm/bin/clang++
pids=""
for u in pass1 pass1_mc pass2 pass2_mc solve modular cgmm cgmm_bin cgmm_k wpe comm hostio capi; do
  src="$ROOT/setk_amd/csrc/$u.hip"
  if [ ! -f "$OUT/$u.o" ] || [ "$src" -nt "$OUT/$u.o" ] || [ -n "$(find "$ROOT/setk_amd/csrc" "$ROOT/include" -name '*.h' -newer "$OUT/$u.o")" ]; then
    /opt/rocm/bin/hipcc --cuda-host-only --offload-arch=gfx950 -O1 -std=c++17 -fPIC -Wno-unused-result $SAN \
      -c "$src" -o "$OUT/$u.o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait "$p"; done
$CLANG -x c++ -std=c++17 -O1 -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $SAN \
  -c "$ROOT/tools/hoststub/hip_stub.cpp" -o "$OUT/hip_stub.o"
# a host-only object still names its (absent) device image: give each name a dummy
rm -f "$OUT/fatbin_dummies.o"
nm --undefined-only "$OUT"/*.o | awk '/__hip_fatbin_/ {print "char " $2 "[16];"}' | sort -u > "$OUT/fatbin_dummies.c"
/opt/rocm/lib/llvm/bin/clang -fPIC -c "$OUT/fatbin_dummies.c" -o "$OUT/fatbin_dummies.o"
$CLANG -shared -fPIC $SAN -o "$LIBOUT" "$OUT"/*.o
echo "$LIBOUT"
