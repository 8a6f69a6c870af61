#!/bin/bash
set -uo pipefail
ROOT="$GRAFT_REPO_ROOT"; OUT="$ROOT/gpurun_out/dp_pmc"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
P="$ROOT/tools/perf_probe.py"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/sq1" -- python3 "$P" --config cfg4 --iters 1 --max-diff 400 --band 8 > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --output-format csv -d "$OUT/sq2" -- python3 "$P" --config cfg4 --iters 1 --max-diff 400 --band 8 > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAVES --output-format csv -d "$OUT/sq3" -- python3 "$P" --config cfg4 --iters 1 --max-diff 400 --band 8 > "$OUT/sq3.log" 2>&1
cd "$ROOT"
python3 - <<'PY'
import csv, glob, collections
for d in ("sq1","sq2","sq3"):
    for f in glob.glob("gpurun_out/dp_pmc/%s/**/*counter_collection.csv" % d, recursive=True):
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "k_extend_bits" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"])
        print(d, dict(acc))
PY
