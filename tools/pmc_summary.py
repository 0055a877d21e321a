#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean of each counter per kernel.  --last N: only the last N dispatches of a
kernel in each file (the RUNNING batch of a bench.py run that starts with a burn-in), default: all.  usage: pmc_summary.py [--last N] [--only tick,rollout] FILE..."""
import csv, sys
from collections import defaultdict
args = sys.argv[1:]
last = 0
only = ("tick", "reset", "random_actions", "rollout")
while args and args[0] in ("--last", "--only"):
    if args[0] == "--last":
        last = int(args[1])
    else:
        only = tuple(args[1].split(","))
    args = args[2:]
acc = defaultdict(lambda: defaultdict(list))
for path in args:
    per = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")
            if "anonymous" in row["Kernel_Name"]:
                k = row["Kernel_Name"].split("::")[1].split("(")[0]
            per[k][row["Counter_Name"]].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
    for k, cs in per.items():
        for c, v in cs.items():
            v.sort()
            acc[k][c] += [x for _, x in (v[-last:] if last else v)]
for k, cs in acc.items():
    if not any(n in k for n in only):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
