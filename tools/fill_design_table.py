#!/usr/bin/env python3
"""Regenerate the roofline table of DESIGN.md §4.5 from the committed profiles (profiles/<tag>_<shape>_kernel_stats.csv: rocprofv3 kernel trace;
profiles/<tag>_traffic_<shape>[_rollout].json: PMC passes).  usage: tools/fill_design_table.py [tag]   (rewrites the block between the ROOFLINE_TABLE markers)"""
import csv, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
SH = [("lab", "labyrinth 2v1 ×4096 (headline)", 3, 64, 4096), ("agh", "agh-map 2v1 ×4096 (configs[2] shard)", 3, 64, 4096),
      ("3v2", "grandbyrinth 3v2 ×8192 (configs[3])", 5, 64, 8192), ("mixed", "five maps ×16384 (configs[4])", 3, 64, 16384),
      ("r90", "labyrinth ×4096, **90 rays** (the reference's sensor; 1682 B)", 3, 90, 4096)]
rows = ["| Shape | kernel | µs per tick (trace) | M env-steps/s | HBM frac | HBM bytes ÷ algorithmic | VALU issue busy | lane utilisation | compute frac (busy × utilisation) | VALU / SALU per env-step | LDS bank conflicts per active cycle |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for sh, label, A, R, N in SH:
    ks = ROOT / "profiles" / f"{tag}_{sh}_kernel_stats.csv"
    if not ks.exists():
        continue
    alg = (108 * A + 3 * A * R + 6 * R + 8) * N
    stats = {}
    for r in csv.DictReader(ks.open()):
        nm = r["Name"]
        for k in ("step_kernel_pooled", "rollout_kernel_pooled", "step_kernel", "rollout_kernel"):
            if "::" + k + "<" in nm:
                stats[k] = float(r["AverageNs"])
                break
    first = True
    for kind, suffix in (("step", ""), ("rollout", "_rollout")):
        tf = ROOT / "profiles" / f"{tag}_traffic_{sh}{suffix}.json"
        if not tf.exists():
            continue
        t = json.loads(tf.read_text())
        kern, tpl = t["kernel"], t.get("ticks_per_launch", 1)
        if kern not in stats:
            continue
        us = stats[kern] / 1e3 / tpl
        v = t.get("valu", {})
        busy, util = v.get('valu_issue_busy_frac', float('nan')), v.get('lane_utilisation', float('nan'))
        rows.append(f"| {label if first else ''} | `{kern}`{' T = 64' if tpl > 1 else ''} | {us:.2f} | {N / us:.1f} | {alg / (us * 1e-6) / 8e12:.4f} | "
                    f"{t['hbm_bytes_per_tick'] / alg:.2f} | {busy:.2f} | {util:.2f} | {busy * util:.2f} | {v.get('valu_insts_per_wave', 0):.0f} / {v.get('salu_insts_per_wave', 0):.0f} | "
                    f"{v.get('lds_bank_conflict_per_active_cycle', float('nan')):.2f} |")
        first = False
block = "<!-- ROOFLINE_TABLE (tools/fill_design_table.py) -->\n" + "\n".join(rows) + "\n<!-- /ROOFLINE_TABLE -->"
p = ROOT / "DESIGN.md"
s = p.read_text()
if "<!-- ROOFLINE_TABLE" in s:
    a, b = s.index("<!-- ROOFLINE_TABLE"), s.index("<!-- /ROOFLINE_TABLE -->") + len("<!-- /ROOFLINE_TABLE -->")
elif "<!-- R4_TABLE" in s:
    a, b = s.index("<!-- R4_TABLE"), s.index("<!-- /R4_TABLE -->") + len("<!-- /R4_TABLE -->")
else:   # the placeholder table of the first draft
    a = s.index("| Shape | kernel | µs per tick (trace)")
    b = s.index("\n\n", a)
s = s[:a] + block + s[b:]
p.write_text(s)
print(block)
