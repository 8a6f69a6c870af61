#!/usr/bin/env python3
"""Counter CSVs of tools/fetch_calib.sh + the known byte counts -> bytes per FETCH_SIZE KiB per access pattern.

    python3 tools/fetch_calib.py <dir> profiles/r04_fetch_calib.json

factor = bytes the pattern requested / (FETCH_SIZE * 1024): what FETCH_SIZE has to be multiplied by for that pattern
(the guide's rule for wide streaming reads is 2)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            rd = csv.reader(fh)
            header = next(rd)
            ki, ci, vi = header.index("Kernel_Name"), header.index("Counter_Name"), header.index("Counter_Value")
            for row in rd:
                name = row[ki].replace("void ", "").split("(")[0]
                acc[name][row[ci]].append(float(row[vi]))
    return acc


def main():
    src, dst = sys.argv[1], sys.argv[2]
    known = json.load(open(os.path.join(src, "known.json")))
    merged = defaultdict(dict)
    for p in sorted(d for d in os.listdir(src) if os.path.isdir(os.path.join(src, d))):
        for k, ctrs in counters(os.path.join(src, p)).items():
            for c, v in ctrs.items():
                merged[k][c] = sum(v) / len(v)
    out = {"_comment": "tools/fetch_calib.sh: FETCH_SIZE (KiB) per kernel of tools/ubench.hip -DUBENCH_MAIN against the bytes "
                       "each access pattern requests; factor = bytes_requested / (FETCH_SIZE * 1024)", "patterns": {}}
    for k, info in known.items():
        c = merged.get(k, {})
        e = dict(info)
        e["counters"] = c
        if c.get("FETCH_SIZE"):
            e["fetch_bytes_counted"] = c["FETCH_SIZE"] * 1024
            e["factor"] = info["bytes_requested"] / (c["FETCH_SIZE"] * 1024)
        out["patterns"][k] = e
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, e in out["patterns"].items():
        print("%-22s %-72s factor %s" % (k, e["pattern"], "%.3f" % e["factor"] if "factor" in e else "-"))


if __name__ == "__main__":
    main()
