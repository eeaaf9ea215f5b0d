#!/usr/bin/env python3
"""SQ_INSTS_VALU per kernel from a rocprofv3 --pmc counter_collection.csv, per unit (argv[2], default 1e6)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
unit = float(sys.argv[2]) if len(sys.argv) > 2 else 1e6
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if m:
            agg[m.group(1)].append(float(r["Counter_Value"]))
tot = 0.0
for k, v in agg.items():
    print(f"{k:28s} launches {len(v):3d}  wave-instructions per unit {sum(v) / len(v) / unit:10.4f}")
    tot += sum(v) / len(v)
print(f"{'all':28s}                wave-instructions per unit {tot / unit:10.4f}")
