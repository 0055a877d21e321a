#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean of each counter per kernel."""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")
            if "anonymous" in row["Kernel_Name"]:
                k = row["Kernel_Name"].split("::")[1].split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if not any(n in k for n in ("tick", "reset", "random_actions")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
