#!/usr/bin/env python3
"""Golden vectors for the PFSP archive from the reference's own module (stdlib only, loaded by
file path; run with python3 -B so no bytecode lands in /root/reference):

    python3 -B tools/make_golden_pfsp.py [/root/reference]

Every recorded value is an output of the reference's own functions: the win_rates.json it writes for a
scripted outcome sequence, "latest", the PFSP weights exactly as ``sample_policy_from_archive`` hands them to
``random.choices`` (captured by a spy on that call), and the policy it returns after ``random.seed(s)``.
The archive path handed to the reference is a ``Path`` whose ``glob`` yields iteration order (plain
``Path.glob`` order is whatever the filesystem returns, so the reference's pick is not reproducible
otherwise); this build's sampler documents the same order."""
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

_CHOICES = random.choices


class SortedGlobPath(type(Path())):
    """glob() in iteration order (see the module docstring)."""

    def glob(self, pattern):
        return sorted(super().glob(pattern), key=lambda q: int(q.stem.split("_")[-1]))


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
            latest = Path(ref.get_latest_policy_from_archive(arch, "cop")).name
            seen = {}

            def spy(population, weights=None, k=1, _orig=_CHOICES):
                seen["cands"], seen["weights"] = [Path(c).name for c in population], list(weights)
                return _orig(population, weights=weights, k=k)
            picks = {}
            for seed in (0, 1, 2, 3, 4):
                random.seed(seed)
                ref.random.choices = spy
                try:
                    picks[str(seed)] = Path(ref.sample_policy_from_archive(SortedGlobPath(arch), "cop", "pfsp")).name
                finally:
                    ref.random.choices = _CHOICES
            cands = [f"cop_iter_{it}.pt" for it in iters]
            assert seen["cands"] == cands
            weights = dict(zip(seen["cands"], seen["weights"]))
            scenarios.append({"iterations": iters, "buffer_size": buf, "events": events, "win_rates_json": rates,
                              "pfsp_weights": weights, "latest": latest, "picks_by_seed": picks})
out = ROOT / "tests" / "golden" / "pfsp_golden.json"
out.write_text(json.dumps({"scenarios": scenarios, "generated_by": "tools/make_golden_pfsp.py: every value is an output of the reference's policy_archive_utils.py (weights captured from its random.choices call, picks after random.seed)"}, indent=1) + "\n")
print(f"{len(scenarios)} scenarios -> {out}")
