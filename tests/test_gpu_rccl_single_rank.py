"""GPU: the learner's collective goes through RCCL.  The box has one GPU and RCCL refuses two ranks on one device
("Duplicate GPU detected", tools/rccl_probe.py), so this is a ONE-rank "nccl" process group with CAT_FORCE_ALLREDUCE=1: the
trainer then sends its [G, P + 1] gradient | KL buffer through ncclAllReduce between the two captured step graphs, as
every rank of an 8-GPU job does.  Checked: RCCL initialises on this stack, the collective runs on the trainer's stream
between graph replays, and the update equals the one without the collective (a one-rank all-reduce is the identity)."""
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

CHILD = textwrap.dedent("""
    import os, sys, datetime, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["CAT_ROOT"])
    from as_cops_and_thieves_amd import VecCopsEnv, load_preset
    from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, RoleConfig, TrainerConfig

    def run(force):
        os.environ["CAT_FORCE_ALLREDUCE"] = "1" if force else "0"
        env = VecCopsEnv(load_preset("squarinth"), 256, num_rays=64, max_step_count=60, seed=2)
        rc = RoleConfig(learning_epochs=1, mini_batches=2, random_timesteps=0, learning_starts=0)
        tr = MAPPOTrainer(env, {"cop": rc, "thief": rc}, TrainerConfig(horizon=16, policy_freeze_duration=0, opponent_freeze_duration=0), seed=3)
        for _ in range(3):
            tr.collect(); tr.update()
        torch.cuda.synchronize()
        out = {r: rl.fp.master.clone() for r, rl in tr.roles.items()}
        graphs = all(bool(rl._graphs) for rl in tr.roles.values())
        env.close()
        return out, graphs

    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=60), device_id=torch.device("cuda:0"))
        probe = torch.ones(8, device="cuda:0"); dist.all_reduce(probe); torch.cuda.synchronize()
    except Exception as exc:
        print("RCCL_UNAVAILABLE", repr(exc)[:300]); sys.exit(0)
    assert dist.get_backend() == "nccl"
    with_rccl, g1 = run(True)
    without, g2 = run(False)
    dist.destroy_process_group()
    assert g1 and g2
    for r in with_rccl:
        moved = float((with_rccl[r] - without[r]).abs().max())
        assert torch.isfinite(with_rccl[r]).all() and moved <= 1e-6, (r, moved)
    print("RCCL_SINGLE_RANK_OK", torch.cuda.nccl.version())
""")


def test_trainer_all_reduce_runs_over_rccl_on_one_rank(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CAT_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if "RCCL_UNAVAILABLE" in res.stdout:
        pytest.skip("RCCL could not start on this box: " + res.stdout.strip()[-300:])
    assert res.returncode == 0 and "RCCL_SINGLE_RANK_OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
