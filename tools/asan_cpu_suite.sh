#!/bin/bash
# Host side of libphasm_overlap.so under AddressSanitizer + UBSan (CPU build only: the device code is compiled as
# usual, hipcc ignores -fsanitize for gfx950), then the CPU test suite (or the tests named on the command line)
# against that build.  Covers the packers, the two host stores, the FASTA / GFA2 threads, the writers and the
# result lifetimes.   usage: tools/asan_cpu_suite.sh [pytest args...]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${ASAN_OUT:-/tmp/phasm_asan}"
mkdir -p "$OUT"
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fsanitize=address,undefined -fno-omit-frame-pointer \
    -fPIC -shared -Wall -Wno-unused-result -Wno-option-ignored -o "$OUT/libphasm_overlap_asan.so" "$ROOT/phasm_amd/csrc/c_api.hip"
cd "$ROOT"
if [ $# -eq 0 ]; then set -- tests -q -m "not gpu"; fi
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD="$RT" \
    PHASM_LIB="$OUT/libphasm_overlap_asan.so" PHASM_SKIP_ASAN_TEST=1 python -m pytest "$@"
