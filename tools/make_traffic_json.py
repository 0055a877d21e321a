#!/usr/bin/env python3
"""profiles/<tag>_final_pmc_summary.txt (+ <tag>_pmc_from_reset.txt) -> profiles/<tag>_traffic.json: HBM bytes per tick_kernel launch
from the FETCH_SIZE / WRITE_SIZE passes (corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE reports half the
bytes of a coalesced stream, so it is doubled; the values are in KB) and the per-wave instruction figures from the SQ
passes.  bench.py replays this file into its JSON line, labelled with its source and regime.  usage: make_traffic_json.py r03"""
import json
import re
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = Path(__file__).resolve().parents[1]


def counters(path):
    txt = path.read_text()
    sec = txt[txt.index("tick_kernel"):]
    body = sec[sec.index("\n") + 1:]
    nxt = re.search(r"^\S", body, re.M)                      # the next kernel's header, if any
    sec = body if nxt is None else body[:nxt.start()]
    return {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+mean\s+([0-9.eE+-]+)", sec, re.M)}


val = counters(root / "profiles" / f"{tag}_final_pmc_summary.txt")
reset_file = root / "profiles" / f"{tag}_pmc_from_reset.txt"
from_reset = counters(reset_file) if reset_file.exists() else None
waves = val["SQ_WAVES"]
hbm = lambda v: int(round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024))
out = {
    "workload": "labyrinth 2v1, 4096 envs, 64 rays (bench.py defaults)",
    "workload_key": {"map": "labyrinth", "envs": 4096, "rays": 64, "cops": 2, "thieves": 1},
    "kernel": "tick_kernel",
    "regime": "running batch (the last 25 launches of `bench.py --steps 20 --warmup 5`, after its 600 burn-in ticks (mid-episode): the launches the bench line times)",
    "burn_in": 600,
    "FETCH_SIZE_KB_per_launch": val["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": val["WRITE_SIZE"],
    "hbm_bytes_per_launch": hbm(val),
    "hbm_bytes_per_launch_from_reset": hbm(from_reset) if from_reset else None,
    "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of a "
               "coalesced stream); this kernel mixes 16-byte record loads with 8-byte table gathers, so the read side is an upper estimate",
    "collected_with": f"tools/collect_profiles.sh {tag}: rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py --steps 20 --warmup 5 "
                      "--no-cpu-baseline --no-extras, counters of the last 25 tick_kernel launches; from_reset: the same with --burn-in 0 (all 25 launches)",
    "source": f"profiles/{tag}_final_pmc_summary.txt",
    "valu": {
        "valu_insts_per_wave": val["SQ_INSTS_VALU"] / waves, "salu_insts_per_wave": val["SQ_INSTS_SALU"] / waves,
        "lds_insts_per_wave": val["SQ_INSTS_LDS"] / waves,
        "lane_utilisation": val["SQ_THREAD_CYCLES_VALU"] / (64 * val["SQ_ACTIVE_INST_VALU"]),
        "valu_issue_busy_frac": 4 * val["SQ_ACTIVE_INST_VALU"] / 1024 / (val["SQ_BUSY_CYCLES"] / 32),
        "lds_bank_conflict_per_active_cycle": val["SQ_LDS_BANK_CONFLICT"] / val["SQ_ACTIVE_INST_LDS"],
        "note": "valu_issue_busy_frac = 4 x SQ_ACTIVE_INST_VALU (issue slots of 4 cycles, summed over 1024 SIMDs) / 1024 / (SQ_BUSY_CYCLES / 32: "
                "per shader engine, 32 of them); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)"},
}
(root / "profiles" / f"{tag}_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
