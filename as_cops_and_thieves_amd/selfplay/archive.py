"""Policy archive + prioritised fictitious self-play (PFSP) opponent sampling.

Same file formats, function names and behaviour as reference ``src/utils/policy_archive_utils.py``:
``win_rates.json`` per role archive (``{policy_file: {wins, games, recent_outcomes[<=buffer],
buffer_size}}``, :11-55), archive naming ``{role}_iter_{n}.pt`` (:97-110), "latest" by the
iteration number parsed from the file stem (:113-126), and the PFSP weight
``max(1e-3, 1 - 2*|wr - 0.5|)`` over the last-``buffer_size`` outcomes, falling back to
wins/games, then to 0.5 (:128-199).  Golden vectors produced by the reference's own module:
``tests/golden/pfsp_golden.json`` (``tools/make_golden_pfsp.py``).

Differences: an explicit ``random.Random`` may be passed (the reference draws from the global
``random`` module); candidate order is sorted by iteration number for reproducibility (the
reference uses ``Path.glob`` order); nothing is printed unless ``verbose``.
"""
from __future__ import annotations

import json
import random
import shutil
from collections import deque
from pathlib import Path
from typing import Dict, List, Optional, Tuple

WIN_RATES_FILENAME = "win_rates.json"
DEFAULT_WIN_RATE = 0.5


def load_win_rates(role_archive_path: Path) -> dict:
    f = Path(role_archive_path) / WIN_RATES_FILENAME
    if not f.exists():
        return {}
    try:
        raw = json.loads(f.read_text())
    except json.JSONDecodeError:
        print(f"Warning: Could not decode JSON from {f}. Returning empty win rates.")
        return {}
    for name, data in raw.items():
        if isinstance(data.get("recent_outcomes"), list):
            raw[name]["recent_outcomes"] = deque(data["recent_outcomes"], maxlen=data.get("buffer_size", 20))
        if "buffer_size" not in data:
            raw[name]["buffer_size"] = 20
    return raw


def save_win_rates(role_archive_path: Path, win_rates_data: dict) -> None:
    out = {}
    for name, data in win_rates_data.items():
        out[name] = dict(data)
        if isinstance(data.get("recent_outcomes"), deque):
            out[name]["recent_outcomes"] = list(data["recent_outcomes"])
    (Path(role_archive_path) / WIN_RATES_FILENAME).write_text(json.dumps(out, indent=4))


def update_policy_win_rate(role_archive_path: Path, policy_filename: str, won_episode: bool,
                           buffer_size: int, verbose: bool = False) -> dict:
    data = load_win_rates(role_archive_path)
    if policy_filename not in data:
        data[policy_filename] = {"wins": 0, "games": 0, "recent_outcomes": deque(maxlen=buffer_size),
                                 "buffer_size": buffer_size}
    st = data[policy_filename]
    if st.get("buffer_size") != buffer_size or not isinstance(st["recent_outcomes"], deque):
        st["recent_outcomes"] = deque(list(st.get("recent_outcomes", [])), maxlen=buffer_size)
        st["buffer_size"] = buffer_size
    st["games"] += 1
    if won_episode:
        st["wins"] += 1
    st["recent_outcomes"].append(1 if won_episode else 0)
    save_win_rates(role_archive_path, data)
    if verbose:
        print(f"Updated win rate for {policy_filename}: {st['wins']}/{st['games']}")
    return st


def add_policy_to_archive(checkpoint_path: str, role_archive_path: Path, iteration_number: int,
                          role_prefix: str) -> Path:
    role_archive_path = Path(role_archive_path)
    role_archive_path.mkdir(parents=True, exist_ok=True)
    dst = role_archive_path / f"{role_prefix}_iter_{iteration_number}.pt"
    shutil.copy(checkpoint_path, dst)
    return dst


def _policy_files(role_archive_path: Path, role_prefix: str) -> List[Path]:
    files = list(Path(role_archive_path).glob(f"{role_prefix}_iter_*.pt"))
    return sorted(files, key=lambda p: int(p.stem.split("_")[-1]))


def get_latest_policy_from_archive(role_archive_path: Path, role_prefix: str) -> Optional[str]:
    if not Path(role_archive_path).exists():
        return None
    files = _policy_files(role_archive_path, role_prefix)
    return str(files[-1]) if files else None


def current_win_rate(stats: Optional[dict]) -> float:
    """Window win rate -> overall win rate -> 0.5 (policy_archive_utils.py:152-170)."""
    if stats and stats.get("games", 0) > 0:
        recent = stats.get("recent_outcomes")
        if recent is not None and len(recent) > 0:
            return sum(recent) / len(recent)
        return stats["wins"] / stats["games"]
    return DEFAULT_WIN_RATE


def pfsp_weight(win_rate: float) -> float:
    return max(1e-3, 1.0 - abs(win_rate - 0.5) * 2.0)          # policy_archive_utils.py:173-175


def pfsp_distribution(role_archive_path: Path, role_prefix: str) -> Tuple[List[str], List[float]]:
    data = load_win_rates(role_archive_path)
    files = _policy_files(role_archive_path, role_prefix)
    return [str(p) for p in files], [pfsp_weight(current_win_rate(data.get(p.name))) for p in files]


def sample_policy_from_archive(role_archive_path: Path, role_prefix: str, strategy: str = "latest",
                               rng: Optional[random.Random] = None) -> Optional[str]:
    rng = rng or random
    if not Path(role_archive_path).exists():
        return None
    files = [str(p) for p in _policy_files(role_archive_path, role_prefix)]
    if not files:
        return None
    if strategy == "latest":
        return get_latest_policy_from_archive(role_archive_path, role_prefix)
    if strategy == "random":
        return str(rng.choice(files))
    if strategy == "pfsp":
        cands, weights = pfsp_distribution(role_archive_path, role_prefix)
        return rng.choices(cands, weights=weights, k=1)[0]
    print(f"Unknown sampling strategy: {strategy}. Defaulting to latest.")
    return get_latest_policy_from_archive(role_archive_path, role_prefix)
