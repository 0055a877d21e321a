#!/usr/bin/env python3
"""A PMC summary (tools/pmc_summary.py) -> profiles/<tag>_traffic[_<shape>[_rollout]].json: HBM bytes per launch of one kernel from the
FETCH_SIZE / WRITE_SIZE passes (corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE reports half the bytes of a
coalesced stream, so it is doubled; the values are in KB) and the per-wave instruction figures from the SQ passes.  bench.py replays
these files into its JSON line (matched by `workload_key` and `kernel`), labelled with their source and regime.
usage: make_traffic_json.py TAG [--shape NAME --key JSON --kernel tick_kernel|rollout_kernel --ticks-per-launch T --burn-in B
                                 --summary FILE --from-reset FILE --out FILE]        (defaults: the headline workload, tick_kernel)"""
import argparse
import hashlib
import json
import re
import subprocess
from pathlib import Path

root = Path(__file__).resolve().parents[1]
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--shape", default="")
ap.add_argument("--key", default='{"map": "labyrinth", "envs": 4096, "rays": 64, "cops": 2, "thieves": 1}')
ap.add_argument("--kernel", default="tick_kernel")
ap.add_argument("--ticks-per-launch", type=int, default=1)
ap.add_argument("--burn-in", type=int, default=600)
ap.add_argument("--summary", default=None)
ap.add_argument("--from-reset", default=None)
ap.add_argument("--out", default=None)
a = ap.parse_args()
tag = a.tag
key = json.loads(a.key)
suffix = (f"_{a.shape}" if a.shape else "") + ("_rollout" if a.kernel.startswith("rollout_kernel") else "")
summary = Path(a.summary) if a.summary else root / "profiles" / (f"{tag}_{a.shape}_pmc_summary.txt" if a.shape else f"{tag}_final_pmc_summary.txt")
reset_file = Path(a.from_reset) if a.from_reset else root / "profiles" / f"{tag}_pmc_from_reset.txt"
out_file = Path(a.out) if a.out else root / "profiles" / f"{tag}_traffic{suffix}.json"


def source_sha16():
    """sha256 over the env core's translation unit (cat_sim.hip + the cat_sim_*.h it includes), as bench.source_sha16 computes it."""
    h = hashlib.sha256()
    csrc = root / "as_cops_and_thieves_amd" / "csrc"
    for f in [csrc / "cat_sim.hip", *sorted(csrc.glob("cat_sim_*.h"))]:
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def git_head():
    """The commit the profile is collected on: build/GIT_HEAD (tools/stamp_head.sh, written before the gpurun call -- the GPU box gets no .git), else git itself."""
    f = root / "build" / "GIT_HEAD"
    if f.exists():
        return f.read_text().strip()
    try:
        return subprocess.run(["git", "rev-parse", "HEAD"], cwd=root, capture_output=True, text=True, check=True).stdout.strip()
    except Exception:   # noqa: BLE001
        return None


def counters_one(path, kernel):
    txt = path.read_text()
    m = re.search(r"^" + re.escape(kernel) + r"<", txt, re.M) or re.search(r"^" + re.escape(kernel), txt, re.M)   # "step_kernel<", not "step_kernel_pooled<"
    sec = txt[m.start():]
    body = sec[sec.index("\n") + 1:]
    nxt = re.search(r"^\S", body, re.M)                      # the next kernel's header, if any
    sec = body if nxt is None else body[:nxt.start()]
    return {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+mean\s+([0-9.eE+-]+)", sec, re.M)}


def counters(path, kernel):
    """Counter means of one kernel; "a+b" (a sim of two parts: one dispatch each per entry, side by side): the SUM over both -- every figure
    derived below is a ratio of such sums, or a count per tick."""
    out = {}
    for k in kernel.split("+"):
        for c, v in counters_one(path, k).items():
            out[c] = out.get(c, 0.0) + v
    return out


val = counters(summary, a.kernel)
from_reset = counters(reset_file, a.kernel) if (not a.kernel.startswith("rollout_kernel") and not a.shape and reset_file.exists()) else None
waves, T = val["SQ_WAVES"], a.ticks_per_launch
hbm = lambda v: int(round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024))
per_step = waves * T          # one wave per env slot: wave-instructions per env-step = counter / (waves x ticks per launch)
regime = (f"running batch (the last 25 {a.kernel} launches of `bench.py --steps 20 --warmup 5`, after its burn-in ticks (mid-episode): the "
          "launches the bench line times)" if not a.kernel.startswith("rollout_kernel") else
          f"running batch (the last 4 rollout_kernel launches of the same command: {T} ticks per launch, after the one-launch-per-tick region)")
out = {
    "workload": f"{key['map']} {key['cops']}v{key['thieves']}, {key['envs']} envs, {key['rays']} rays",
    "workload_key": key,
    "kernel": a.kernel,
    "ticks_per_launch": T,
    "source_sha16": source_sha16(),   # bench.py: profile_stale
    "git_head": git_head(),
    "regime": regime,
    "burn_in": a.burn_in,
    "FETCH_SIZE_KB_per_launch": val["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": val["WRITE_SIZE"],
    "hbm_bytes_per_launch": hbm(val),
    "hbm_bytes_per_tick": hbm(val) / T,
    "hbm_bytes_per_launch_from_reset": hbm(from_reset) if from_reset else None,
    "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of a "
               "coalesced stream); this kernel mixes 16-byte record loads with 4/8-byte table gathers, so the read side is an upper estimate",
    "collected_with": f"tools/collect_profiles.sh {tag}: rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py <shape> --steps 20 --warmup 5 "
                      "--shape-only, counters of the kernel's last launches (25 tick_kernel / 4 rollout_kernel)",
    "source": str(summary.relative_to(root)) if summary.is_relative_to(root) else str(summary),
    "valu": {
        "valu_insts_per_wave": val["SQ_INSTS_VALU"] / per_step, "salu_insts_per_wave": val["SQ_INSTS_SALU"] / per_step,
        "lds_insts_per_wave": val["SQ_INSTS_LDS"] / per_step,
        "per": "env-step (one wave per env slot; rollout_kernel: counter / (waves x ticks per launch))",
        "lane_utilisation": val["SQ_THREAD_CYCLES_VALU"] / (64 * val["SQ_ACTIVE_INST_VALU"]),
        "valu_issue_busy_frac": 4 * val["SQ_ACTIVE_INST_VALU"] / 1024 / (val["SQ_BUSY_CYCLES"] / 32),
        "note": "valu_issue_busy_frac = 4 x SQ_ACTIVE_INST_VALU (issue slots of 4 cycles, summed over 1024 SIMDs) / 1024 / (SQ_BUSY_CYCLES / 32: "
                "per shader engine, 32 of them); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)"},
}
if "SQ_LDS_BANK_CONFLICT" in val and "SQ_ACTIVE_INST_LDS" in val:
    out["valu"]["lds_bank_conflict_per_active_cycle"] = val["SQ_LDS_BANK_CONFLICT"] / val["SQ_ACTIVE_INST_LDS"]
out_file.write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
