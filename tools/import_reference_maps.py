#!/usr/bin/env python3
"""Convert the reference's map JSONs (maps_templates/*.json) into the compact bundled schema.

Run in the build container only (needs /root/reference):

    python3 -B tools/import_reference_maps.py [/root/reference/maps_templates]

Output: as_cops_and_thieves_amd/maps_data/<name>.cmap.json with
  window / canvas : as in the source file
  rings           : one flat [x0,y0,x1,y1,...] list per block, AFTER the reference's block
                    rules (rect expansion with w/h default 1, poly rings closed) —
                    src/maps/map.py:35-61
  agents          : the source's agent list verbatim, or absent when the source has none
This is map DATA (inputs), not reference code.
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from as_cops_and_thieves_amd.maps import parse_block  # noqa: E402


def main() -> None:
    src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/maps_templates")
    dst = ROOT / "as_cops_and_thieves_amd" / "maps_data"
    dst.mkdir(parents=True, exist_ok=True)
    for f in sorted(src.glob("*.json")):
        d = json.loads(f.read_text())
        out = {"window": d["window"], "canvas": d["canvas"], "rings": []}
        for blk in d["objects"]["blocks"]:
            ring = parse_block(blk)
            flat = []
            for x, y in ring:
                flat += [int(x) if float(x).is_integer() else x, int(y) if float(y).is_integer() else y]
            out["rings"].append(flat)
        if "agents" in d:
            out["agents"] = d["agents"]
        (dst / f"{f.stem}.cmap.json").write_text(json.dumps(out, separators=(",", ":")) + "\n")
        print(f"{f.name}: {len(out['rings'])} rings -> {f.stem}.cmap.json")


if __name__ == "__main__":
    main()
