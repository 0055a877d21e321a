"""GPU: bench.py's N > 1 branch without an 8-GPU node.  Two fresh rank processes are started by
torch.distributed.run (the launcher runs before anything touches the GPU) with CAT_BENCH_REHEARSE=1: both ranks use
the one GPU of the box and the timing reductions go over gloo.  What is checked is the flow the driver's N = 2, 4, 8
runs take: env shards keyed by disjoint env_id_offsets, barrier + MAX-over-ranks timing, ONE JSON line from rank 0
whose value is the whole-job aggregate."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_two_rank_bench_prints_one_aggregate_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CAT_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    envs, steps = 1024, 60
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "10",
           "--envs", str(envs)]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["scaling"] == "weak"
    assert d["config"]["env_id_offsets"] == [0, envs]               # disjoint contiguous shards of the global env ids
    assert "gloo" in d["config"]["timing_reductions"]
    assert "cpu_baseline" not in d                                  # N = 1 only
    (label, agh), = [(k, v) for k, v in d["extra"].items() if k.startswith("agh-map 2v1 x4096 per GPU")]   # configs[2]'s per-GPU shard, on every rank
    assert agh["value"] > 1e6 and agh["envs_per_gpu"] == 4096 and agh["resident_rollout"]["value"] > agh["value"]
    # configs[2]'s other half: the MAPPO learner on the sharded envs, its gradient | KL buffer all-reduced under the job's group
    (_, lt), = [(k, v) for k, v in d["extra"].items() if k.startswith("learner_collect_plus_update, 2 GPUs")]
    assert lt.get("value") and lt["ranks"] == 2 and lt["allreduce_backend"] == "gloo" and lt["rccl_ranks"] is None, lt
    # the resident rollout launch on the headline workload, both ranks
    (_, rr), = [(k, v) for k, v in d["extra"].items() if "resident rollout" in k]
    assert rr["T"] == 64 and rr["value"] > d["value"] and rr["kernel_ms_per_tick"] > 0
    # whole-job aggregate: both ranks' env-steps over the slowest rank's time
    assert d["value"] == pytest.approx(2 * envs * steps / (d["ms_per_step"] * 1e-3 * steps), rel=1e-9)
    assert d["roofline"]["kernel_ms"] > 0 and d["value"] > 1e6


def test_plain_command_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` typed as a plain command (no launcher, no WORLD_SIZE): the parent starts two fresh rank
    processes itself before anything touches the GPU and relays rank 0's single line -- the form the driver's SCALE runs may use."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CAT_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    envs, steps = 1024, 60
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "10",
                          "--envs", str(envs), "--no-extras"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps
    assert d["config"]["env_id_offsets"] == [0, envs]
    assert d["config"]["rccl_ranks"] is None and "gloo" in d["config"]["timing_reductions"]   # rehearsal: no RCCL claim
    assert d["roofline"]["kernel_launches_timed"] == steps // 8 + (1 if steps % 8 else 0)


def test_a_rank_count_that_contradicts_the_launcher_is_refused():
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                         env=dict(os.environ, WORLD_SIZE="4", RANK="0"), capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "rank count must equal --gpus" in (res.stderr + res.stdout)


def test_four_rank_bench_rehearsal():
    """More ranks than the two of the tests above, as far as one GPU box goes: a box admits six processes on its card, so with the test runner itself on the
    GPU four rank processes is the most this suite starts (the eight-rank flow itself -- shard arithmetic, rank-0-only IO, identical replicas, clean exits --
    runs on CPU: tests/test_self_play_two_ranks_cpu.py::test_eight_ranks_share_one_self_play_job, tests/test_sharding_gloo.py).  The plain command starts
    its own four ranks; each takes its own env shard; ONE line comes back whose value is the whole-job aggregate over the slowest rank's time.  No scaling
    curve is measured here or anywhere in this repo until the driver's SCALE run executes on a multi-GPU node."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CAT_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    envs, steps = 256, 40
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--steps", str(steps), "--warmup", "5", "--envs", str(envs), "--no-extras"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["steps"] == steps and d["scaling"] == "weak" and d["device_errors"] == 0
    assert d["config"]["env_id_offsets"] == [0, envs, 2 * envs, 3 * envs]
    assert d["config"]["rccl_ranks"] is None and "gloo" in d["config"]["timing_reductions"]
    assert d["value"] == pytest.approx(4 * envs * steps / (d["ms_per_step"] * 1e-3 * steps), rel=1e-9)
    # four ranks time-share ONE GPU here: the aggregate stays in the range of what the card does alone on this many envs (N = 1-consistent), no more
    assert 1e6 < d["value"] < 4e8
