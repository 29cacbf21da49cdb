#!/usr/bin/env bash
# Run pytest (default: the C-ABI, pipeline and CLI tests) against the sanitizer build of
# tools/asan_build.sh.   bash tools/asan_run.sh [pytest args...]
# NEEDS the ASAN flavour of the ROCm libraries (/opt/rocm/lib/asan): with the stock ones ROCm's
# ASAN runtime fails inside its hsa_amd_memory_pool_allocate interceptor at the first device
# allocation (seen on the MI355X box of this image: "allocator is trying to allocate 0x400000
# bytes").  What runs everywhere is the host-only variant, tests/test_host_asan.py.
# The interpreter is not instrumented, so the ASAN runtime is preloaded; leak checking is off
# (CPython and the HIP runtime hold allocations to exit), the shadow-gap protection is off for
# the HIP runtime's fixed mappings.  Any report fails the run (halt_on_error, exit code 99).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$ROOT/_abl/libsetk_asan.so" ] || bash "$ROOT/tools/asan_build.sh"
export SETK_LIB="$ROOT/_abl/libsetk_asan.so"
export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:exitcode=99:abort_on_error=0"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export SETK_BUILD_FORCE=0
cd "$ROOT"
if [ $# -eq 0 ]; then set -- tests/test_gpu_api.py tests/test_gpu_enhance.py tests/test_gpu_cgmm.py tests/test_gpu_wpe.py tests/test_gpu_classic.py -m gpu -x -q; fi
LD_PRELOAD="$RT" python -m pytest "$@"
