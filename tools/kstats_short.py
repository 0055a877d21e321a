#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv with shortened kernel names: python tools/kstats_short.py stats.csv [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms, {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    name = r["Name"]
    m = re.search(r"(MT\d+x\d+x\d+)", name)
    if name.startswith("Cijk"):
        short = "GEMM " + name[5:14] + " " + (m.group(1) if m else "")
    else:
        short = re.sub(r"at::native::|\(anonymous namespace\)::|void |c10::", "", name)[:110]
    print(f"{int(r['TotalDurationNs']) / 1e6:9.2f} ms {100 * int(r['TotalDurationNs']) / tot:5.1f}%  {int(r['Calls']):7d} x {float(r['AverageNs']) / 1e3:8.1f} us  {short}")
