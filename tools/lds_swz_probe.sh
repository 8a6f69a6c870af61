#!/bin/bash
# LDS bank conflicts of k_verify_a, with and without a padded layout of a in LDS (measurement
# infrastructure, run on the GPU box):
#
#     tools/lds_swz_probe.sh gpurun_out/lds_swz
#
# Two builds of the library (-DPO_VER_LDS_SWZ=0 / 1: one pad dword per 32 dwords of a), each run once plain (kernel time
# from the library's own HIP events, config 2, 5 calls) and once under `rocprofv3 --pmc` (counters only, the program
# itself behind `--`).  The report lines: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of k_verify_a, SQ_INSTS_VALU,
# SQ_BUSY_CU_CYCLES, ms_verify_kernel.
set -uo pipefail
OUT="${1:?output directory}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$OUT"
OUT="$(cd "$OUT" && pwd)"
export TMPDIR=/tmp
for n in 0 1; do
    lib="$OUT/libphasm_swz$n.so"
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DPO_VER_LDS_SWZ=$n -o "$lib" "$ROOT/phasm_amd/csrc/c_api.hip" || exit 1
    export PHASM_LIB="$lib"
    python3 "$ROOT/tools/perf_probe.py" --config cfg2 --iters 5 > "$OUT/time$n.log" 2>&1 || echo "timing run $n failed" >&2
    (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/pmc$n" -- python3 "$ROOT/tools/perf_probe.py" --config cfg2 --iters 3 > "$OUT/pmc$n.log" 2>&1) || echo "pass pmc$n failed" >&2
    rm -f "$lib"
done
unset PHASM_LIB
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
for n in (0, 1):
    acc = {}
    for f in glob.glob(os.path.join(out, "pmc%d" % n, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rd = csv.reader(fh)
            h = next(rd)
            ki, ci, vi = h.index("Kernel_Name"), h.index("Counter_Name"), h.index("Counter_Value")
            for row in rd:
                if "k_verify_a" in row[ki]:
                    acc.setdefault(row[ci], []).append(float(row[vi]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    ms = []
    for line in open(os.path.join(out, "time%d.log" % n)):
        if line.startswith("{"):
            ms.append(json.loads(line)["ms_verify_kernel"])
    ratio = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else float("nan")
    print("PO_VER_LDS_SWZ=%d: k_verify_a %.4f ms (min of %d calls); SQ_LDS_BANK_CONFLICT %.4g / SQ_LDS_IDX_ACTIVE %.4g = %.3f; SQ_INSTS_VALU %.4g; SQ_BUSY_CU_CYCLES %.4g"
          % (n, min(ms) if ms else float("nan"), len(ms), m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_LDS_IDX_ACTIVE", 0), ratio,
             m.get("SQ_INSTS_VALU", 0), m.get("SQ_BUSY_CU_CYCLES", 0)))
PY
echo done >&2
