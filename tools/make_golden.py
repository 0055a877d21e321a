#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference checkout (build container only).

Only DATA is extracted: the [tool.physical-params] table of the reference's pyproject.toml, the
sensor literals of src/agents/entity.py (by regex on the assignment lines), SimpleEnv/BaseEnv
default arguments, the ObjectType enum values, and per-map structural facts recomputed here with
scipy's Qhull (an implementation independent of as_cops_and_thieves_amd.maps.convex_hull).

    python3 -B tools/make_golden.py [/root/reference]
"""
import json
import re
import sys
from pathlib import Path

import numpy as np
import tomli
from scipy.spatial import ConvexHull

ROOT = Path(__file__).resolve().parents[1]
REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


def grab(path: Path, pattern: str):
    for no, line in enumerate(path.read_text().splitlines(), 1):
        m = re.search(pattern, line)
        if m:
            return m.group(1), f"{path.relative_to(REF)}:{no}"
    raise KeyError(pattern)


def main() -> None:
    phys = tomli.loads((REF / "pyproject.toml").read_text())["tool"]["physical-params"]
    ent = REF / "src/agents/entity.py"
    lits = {}
    for key, pat in (("ray_length", r"self\._ray_length = ([0-9.]+)"), ("num_rays", r"self\._num_rays = ([0-9]+)"),
                     ("ray_radius", r"segment_query_first\(origin, pymunk\.Vec2d\(\*end\), ([0-9.]+), ray_filter\)")):
        val, where = grab(ent, pat)
        lits[key] = {"value": float(val), "source": where}
    val, where = grab(REF / "src/environments/simple_env.py", r"time_step: float = 1 / ([0-9.]+)")
    lits["simple_env_time_step_denominator"] = {"value": float(val), "source": where}
    val, where = grab(REF / "src/environments/base_env.py", r"time_step: float = 1 / ([0-9.]+)")
    lits["base_env_time_step_denominator"] = {"value": float(val), "source": where}
    val, where = grab(REF / "src/environments/base_env.py", r"max_step_count: int = ([0-9]+)")
    lits["max_step_count"] = {"value": float(val), "source": where}
    val, where = grab(REF / "src/maps/map.py", r"pymunk\.Poly\(space\.static_body, vs, radius=([0-9.]+)\)")
    lits["wall_radius"] = {"value": float(val), "source": where}
    obj = {}
    for no, line in enumerate((REF / "src/utils/object_types.py").read_text().splitlines(), 1):
        m = re.match(r"\s+([A-Z]+) = ([0-9]+)", line)
        if m:
            obj[m.group(1)] = int(m.group(2))
    (OUT / "reference_constants.json").write_text(json.dumps(
        {"physical_params": phys, "literals": lits, "object_types": obj,
         "generated_by": "tools/make_golden.py"}, indent=1) + "\n")

    facts = {}
    for f in sorted((REF / "maps_templates").glob("*.json")):
        d = json.loads(f.read_text())
        per_shape, dropped = [], 0
        for blk in d["objects"]["blocks"]:
            if blk.get("type", "rect") == "rect":
                x, y = blk["x"], blk["y"]
                w = blk.get("w") if blk.get("w") is not None else 1
                h = blk.get("h") if blk.get("h") is not None else 1
                pts = [(x, y), (x + w, y), (x + w, y + h), (x, y + h)]
            else:
                pts = [(v["x"], v["y"]) for v in blk["vs"]]
            uniq = sorted(set(map(tuple, pts)))
            hull = ConvexHull(np.asarray(uniq, dtype=float))
            # Qhull keeps collinear points out of .vertices already
            per_shape.append(len(hull.vertices))
            if len(hull.vertices) < len(uniq):
                dropped += 1
        facts[f.stem] = {"shapes": len(per_shape), "hull_edges_total": int(sum(per_shape)),
                         "hull_edges_max": int(max(per_shape)), "shapes_dropping_vertices": dropped,
                         "window": list(d["window"].values()), "has_agents": "agents" in d,
                         "roster": [a["type"] for a in d.get("agents", [])]}
    (OUT / "map_facts.json").write_text(json.dumps(
        {"maps": facts, "generated_by": "tools/make_golden.py (scipy.spatial.ConvexHull)"}, indent=1) + "\n")
    print(json.dumps(facts, indent=1))


if __name__ == "__main__":
    main()
