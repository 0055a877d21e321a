import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"] / 1e6, 2), "M steps/s", round(d["ms_per_step"] * 1e3, 1), "us/step",
      round(d["roofline"]["kernel_ms"] * 1e3, 1), "us kernel")
