#!/usr/bin/env python3
"""Copies the reference's map DATA files (maps_templates/*.json: window, canvas, blocks, agents) into tests/golden/maps_raw/
unchanged, so that the CPU oracle can parse the maps from the reference's own schema with its own code
(oracle/cat_oracle_host.py) on a box that has no /root/reference.  Data only: no reference source or script is copied.

Run in the build container:  python3 -B tools/make_golden_maps_raw.py [/root/reference/maps_templates]
"""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def main() -> None:
    src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/maps_templates")
    dst = ROOT / "tests" / "golden" / "maps_raw"
    dst.mkdir(parents=True, exist_ok=True)
    index = {}
    for f in sorted(src.glob("*.json")):
        raw = f.read_bytes()
        json.loads(raw)                                   # must be valid JSON
        (dst / f.name).write_bytes(raw)
        index[f.name] = {"bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest()}
        print(f"{f.name}: {len(raw)} bytes")
    (dst / "INDEX.json").write_text(json.dumps({"source": "maps_templates/*.json of the reference, byte for byte", "files": index},
                                               indent=1) + "\n")


if __name__ == "__main__":
    main()
