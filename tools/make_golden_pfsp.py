#!/usr/bin/env python3
"""Golden vectors for the PFSP archive from the reference's own module (stdlib only, loaded by
file path; run with python3 -B so no bytecode lands in /root/reference):

    python3 -B tools/make_golden_pfsp.py [/root/reference]

Records, for scripted outcome sequences, the win_rates.json the reference writes, the PFSP weight
of every archived policy, "latest", and which policy random.choices picks for fixed seeds with the
candidates in iteration order."""
import contextlib
import importlib.util
import io
import json
import random
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
spec = importlib.util.spec_from_file_location("ref_pau", REF / "src/utils/policy_archive_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

scenarios = []
rng = random.Random(7)
with contextlib.redirect_stdout(io.StringIO()):
    for sc in range(6):
        with tempfile.TemporaryDirectory() as td:
            arch = Path(td) / "cops"
            arch.mkdir()
            n_pol = rng.randint(1, 6)
            buf = rng.choice([3, 5, 20])
            ck = Path(td) / "ckpt.pt"
            ck.write_bytes(b"x")
            iters = sorted(rng.sample(range(0, 40), n_pol))
            for it in iters:
                ref.add_policy_to_archive(str(ck), arch, it, "cop")
            events = []
            for _ in range(rng.randint(0, 60)):
                name = f"cop_iter_{rng.choice(iters)}.pt"
                won = rng.random() < 0.6
                events.append([name, won])
                ref.update_policy_win_rate(arch, name, won, buf)
            rates = json.loads((arch / "win_rates.json").read_text()) if (arch / "win_rates.json").exists() else {}
            loaded = ref.load_win_rates(arch)
            weights = {}
            for it in iters:
                name = f"cop_iter_{it}.pt"
                st = loaded.get(name)
                wr = 0.5
                if st and st["games"] > 0:
                    ro = st.get("recent_outcomes")
                    wr = sum(ro) / len(ro) if ro is not None and len(ro) > 0 else st["wins"] / st["games"]
                weights[name] = max(1e-3, 1.0 - abs(wr - 0.5) * 2.0)
            # PFSP picks with the candidate list in iteration order (the port's documented order)
            cands = [f"cop_iter_{it}.pt" for it in iters]
            picks = {}
            for seed in (0, 1, 2, 3, 4):
                picks[str(seed)] = random.Random(seed).choices(cands, weights=[weights[c] for c in cands], k=1)[0]
            # sanity: the reference's own sampler returns a member and agrees on "latest"
            latest = Path(ref.get_latest_policy_from_archive(arch, "cop")).name
            random.seed(0)
            assert Path(ref.sample_policy_from_archive(arch, "cop", "pfsp")).name in cands
            scenarios.append({"iterations": iters, "buffer_size": buf, "events": events, "win_rates_json": rates,
                              "pfsp_weights": weights, "latest": latest, "picks_by_seed": picks})
out = ROOT / "tests" / "golden" / "pfsp_golden.json"
out.write_text(json.dumps({"scenarios": scenarios, "generated_by": "tools/make_golden_pfsp.py (reference policy_archive_utils.py)"}, indent=1) + "\n")
print(f"{len(scenarios)} scenarios -> {out}")
