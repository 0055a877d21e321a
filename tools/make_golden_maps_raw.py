#!/usr/bin/env python3
"""The reference's map DATA (maps_templates/*.json: window, canvas, blocks, agents) as ONE fixture, tests/golden/reference_maps.json
= {map name: the parsed JSON of the reference's file, nothing added or removed}, so that the CPU oracle can parse the maps from the
reference's own schema with its own code (oracle/cat_oracle_host.py) on a box that has no /root/reference.  Data only: no
reference source or script is copied.  "sources" records the size and SHA-256 of each file the entry was parsed from.

Run in the build container:  python3 -B tools/make_golden_maps_raw.py [/root/reference/maps_templates]
"""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def main() -> None:
    src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/maps_templates")
    out = {"maps": {}, "sources": {}}
    for f in sorted(src.glob("*.json")):
        raw = f.read_bytes()
        out["maps"][f.stem] = json.loads(raw)
        out["sources"][f.name] = {"bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest()}
        print(f"{f.name}: {len(raw)} bytes")
    (ROOT / "tests" / "golden" / "reference_maps.json").write_text(json.dumps(out, separators=(",", ":")) + "\n")


if __name__ == "__main__":
    main()
