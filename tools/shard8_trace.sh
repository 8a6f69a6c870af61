#!/bin/bash
set -uo pipefail
ROOT="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cat > /tmp/shard8.py <<'PY'
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper
ov = ExactOverlapper()
for name, seq in synth.oriented(synth.generate_reads(synth.CONFIGS["cfg2"])):
    ov.add_sequence(name, seq)
ov.upload()
os.environ["PHASM_NO_INDEX_REUSE"] = "1"
slot = torch.zeros((600000, 4), dtype=torch.int32, device="cuda")     # (the exchange slot of phasm_amd/dist.py)
torch.cuda.synchronize()
for it in range(6):
    t0 = time.perf_counter()
    r, written = ov.candidates_result_into(1000, 3, 8, slot.data_ptr() + 16, slot.shape[0] - 1)
    t1 = time.perf_counter()
    n = len(r); r.free()
    st = ov.stats()
    print("shard 3/8: wall %.3f ms  device %.3f ms  index %.3f scan %.3f fill %.3f verify %.3f select %.3f  cands %d written %s" % ((t1-t0)*1e3, st["ms_total"], st["ms_index"], st["ms_scan_count"], st["ms_scan_fill"], st["ms_verify"], st["ms_select"], n, written), flush=True)
PY
cd /tmp
python3 /tmp/shard8.py
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/s8 -- python3 /tmp/shard8.py > /dev/null 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/s8/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last call: find last k_call_init / k_call_reset start
idx = [i for i, r in enumerate(rows) if "k_call_init" in r["Kernel_Name"] or "k_call_reset" in r["Kernel_Name"]]
seg = rows[idx[-1]:]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = t0
tot = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:60]))
    tot += e - s; prev_end = e
print("kernels %d, kernel time %.1f us, span %.1f us" % (len(seg), tot / 1e3, (prev_end - t0) / 1e3))
PY
rm -rf gpurun_out/s8
