#!/bin/bash
# FETCH_SIZE calibrated against KNOWN byte counts per access pattern (MI355X_MICROARCH.md "HBM": only the
# wide streaming read is calibrated there).  tools/ubench.hip built with -DUBENCH_MAIN runs every pattern once as a kernel
# of its own name and prints the bytes it requests; the counter passes below give FETCH_SIZE / the TCC request counters per
# kernel.  Counter-only runs, the program itself behind `--`.
#
#     tools/fetch_calib.sh gpurun_out/fetch_calib && python3 tools/fetch_calib.py gpurun_out/fetch_calib profiles/r04_fetch_calib.json
set -uo pipefail
OUT="${1:?output directory}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$OUT"
export TMPDIR=/tmp
BIN="$ROOT/tools/ubench_cal"
if [ ! -x "$BIN" ] || [ "$BIN" -ot "$ROOT/tools/ubench.hip" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DUBENCH_MAIN -o "$BIN" "$ROOT/tools/ubench.hip" || exit 1
fi
"$BIN" > "$OUT/known.json" || exit 1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "req TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "hit TCC_HIT_sum TCC_MISS_sum" "tcc TCC_REQ_sum TCC_READ_sum"; do
    set -- $pass
    name="$1"; shift
    echo "== pass $name: $*" >&2
    rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- "$BIN" > "$OUT/$name.log" 2>&1 || echo "pass $name failed (see $OUT/$name.log)" >&2
done
echo done >&2
