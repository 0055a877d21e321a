"""CPU: the host-side pieces of bench.py that need no GPU (the measurement itself is covered by the -m gpu tests)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_algorithmic_bytes_follow_survey_8d():
    assert bench.algorithmic_bytes_per_env_step(3, 64) == 1292          # 108 A + 3 A R + 6 R + 8
    assert bench.algorithmic_bytes_per_env_step(5, 64) == 108 * 5 + 3 * 5 * 64 + 6 * 64 + 8


def test_short_regions_carry_no_per_dispatch_events():
    assert bench.event_stride(20) == 0      # the driver's --steps 20: one event pair around all 20 launches (timed_steps), none attached to a dispatch
    assert bench.event_stride(2000) == 8


def test_cpu_share_is_bounded_by_the_box_share():
    n = bench.host_cpu_share()
    assert 1 <= n <= 16


def test_committed_traffic_file_matches_the_headline_workload_and_says_its_regime():
    newest = sorted((ROOT / "profiles").glob("r*_traffic.json"))[-1]
    prof = json.loads(newest.read_text())
    assert prof["workload_key"] == {"map": "labyrinth", "envs": 4096, "rays": 64, "cops": 2, "thieves": 1}
    assert "running batch" in prof["regime"] and prof["burn_in"] == 600
    assert prof["hbm_bytes_per_launch"] > bench.algorithmic_bytes_per_env_step(3, 64) * 4096       # counters >= algorithmic bytes
    assert prof["hbm_bytes_per_launch_from_reset"] and prof["hbm_bytes_per_launch_from_reset"] <= prof["hbm_bytes_per_launch"]
    assert 0.3 < prof["valu"]["lane_utilisation"] < 1.0


def test_plain_multi_gpu_command_is_a_launcher_invocation(monkeypatch):
    """`python bench.py --gpus 4` without WORLD_SIZE: the parent builds a torch.distributed.run command line on 127.0.0.1 with one
    process per GPU and hands its own arguments through (the GPU test runs it; here the command is only inspected)."""
    seen = {}

    class _Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _Done()
    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    monkeypatch.setenv("RANK", "3")                      # stale launcher variables must not leak into the children
    assert bench.launch_ranks(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert "RANK" not in seen["env"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
