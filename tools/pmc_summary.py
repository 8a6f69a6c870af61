#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
want = sys.argv[2:] or None
for k, ctrs in acc.items():
    if want and not any(w in k for w in want):
        continue
    print(k)
    for c, v in sorted(ctrs.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
