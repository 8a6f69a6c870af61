#!/usr/bin/env python3
"""tools/pad_probe.sh output -> VALU-issue sensitivity per kernel (see there)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FULL_RATE_CYCLES = 2.15   # measured: tools/valu_calib.py (SQ_INSTS_VALU x 2.15 / (4 x SQ_BUSY_CU_CYCLES) = 1 for a saturating v_xor loop)


def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"].replace("void ", "").replace("po::", "").split("(")[0]
                if k.startswith("k_verify_a") or k.startswith("k_scan_probe"):
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    root = sys.argv[1]
    pts = {n: load(os.path.join(root, "pad%d" % n)) for n in (0, 4, 8)}
    out = {}
    for k in sorted(pts[0]):
        rows = []
        for n in (0, 4, 8):
            c = pts[n].get(k)
            if c:
                rows.append((n, c["SQ_INSTS_VALU"], c["SQ_BUSY_CU_CYCLES"], c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
        if len(rows) < 2:
            continue
        (n0, i0, b0, _), (n1, i1, b1, _) = rows[0], rows[-1]
        slope = (b1 - b0) / ((i1 - i0) * FULL_RATE_CYCLES / 4.0)
        out[k] = {"points": [{"pad": n, "SQ_INSTS_VALU": i, "SQ_BUSY_CU_CYCLES": b, "wait_any_frac": w} for n, i, b, w in rows],
                  "slope_busy_cycles_per_added_valu_cycle": slope,
                  "added_valu_cycles_frac": (i1 - i0) * FULL_RATE_CYCLES / 4.0 / b0, "time_growth_frac": (b1 - b0) / b0}
        print("%-40s" % k, " ".join("pad %d: INSTS_VALU %.4g BUSY_CU %.4g (wait %.2f)" % r for r in rows))
        print("%-40s slope %.2f  (+%.1f %% of the launch's cycles as VALU work -> +%.1f %% time)" % ("", slope, 100 * out[k]["added_valu_cycles_frac"], 100 * out[k]["time_growth_frac"]))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
