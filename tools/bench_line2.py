"""One bench.py JSON line -> one short line: one-tick rate + kernel time, and the resident rollout's when the line carries it (--shape-only)."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
out = [sys.argv[1] if len(sys.argv) > 1 else "", f"{d['value'] / 1e6:.2f} M", f"{d['roofline']['kernel_ms'] * 1e3:.2f} us"]
for k, v in (d.get("extra") or {}).items():
    if "resident rollout" in k:
        out += ["| resident", f"{v['value'] / 1e6:.2f} M", f"{v['kernel_ms_per_tick'] * 1e3:.2f} us/tick"]
print(*out)
