#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name (sum over dispatches)."""
import csv
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0][:100]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
names = sorted({c for v in agg.values() for c in v})
print("kernel,dispatches," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -max(kv[1].values())):
    print(k + "," + str(len(calls[k])) + "," + ",".join(f"{v.get(n, 0):.4g}" for n in names))
