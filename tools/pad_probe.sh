#!/bin/bash
# VALU-issue sensitivity of k_verify_a and k_scan_probe (measurement infrastructure; run on the GPU box):
#
#     tools/pad_probe.sh gpurun_out/pad && python3 tools/pad_report.py gpurun_out/pad
#
# Builds the library with n extra full-rate VALU instructions per 16-byte compare (PO_VER_PAD) / per filter position
# (PO_SCAN_PAD), n = 0, 4, 8, and collects SQ_INSTS_VALU + SQ_BUSY_CU_CYCLES of config 2 for each build (one
# `rocprofv3 --pmc` pass each, counters only, the program itself behind `--`).  tools/pad_report.py turns the three
# points per kernel into  slope = d(SQ_BUSY_CU_CYCLES) / d(SQ_INSTS_VALU x 2.15 / 4):  1 = every added VALU cycle
# lengthens the kernel (VALU-issue bound), 0 = the pipe had room (latency bound).
set -uo pipefail
OUT="${1:?output directory}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$OUT"
OUT="$(cd "$OUT" && pwd)"
export TMPDIR=/tmp
for n in 0 4 8; do
    lib="$OUT/libphasm_pad$n.so"
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DPO_VER_PAD=$n -DPO_SCAN_PAD=$n -o "$lib" "$ROOT/phasm_amd/csrc/c_api.hip" || exit 1
    export PHASM_LIB="$lib"
    (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/pad$n" -- python3 "$ROOT/tools/perf_probe.py" --config cfg2 --iters 3 > "$OUT/pad$n.log" 2>&1) || echo "pass pad$n failed" >&2
    rm -f "$lib"
done
unset PHASM_LIB
echo done >&2
