"""GPU: data-parallel training of the native learner with TWO ranks on the one GPU of the box.  RCCL refuses two ranks on
one device (tests/test_gpu_rccl_single_rank.py covers RCCL with one rank), so the collective here is gloo on GPU tensors;
everything else is the production path: per-rank env shards (disjoint ``env_id_offset``s), the libcat_learn.so kernels,
the replayed HIP graphs with the all-reduce of the [G, P + 1] gradient | KL buffer between them.  Both ranks must end with
bit-identical parameters and optimiser step counts although their rollouts differ -- also when a tiny KL threshold makes
them stop early (the KL statistics travel in the same buffer, so they take the same decision)."""
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
    import os, sys, datetime, hashlib, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["CAT_ROOT"])
    from as_cops_and_thieves_amd import VecCopsEnv, load_preset
    from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, RoleConfig, TrainerConfig
    rank, kl = int(os.environ["RANK"]), float(os.environ["CAT_TEST_KL"])
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
    env = VecCopsEnv(load_preset("squarinth"), 256, num_rays=64, max_step_count=60, seed=2, env_id_offset=256 * rank)
    rc = RoleConfig(learning_epochs=2, mini_batches=2, random_timesteps=0, learning_starts=0, kl_threshold=kl, learning_rate=3e-3)
    tr = MAPPOTrainer(env, {"cop": rc, "thief": rc}, TrainerConfig(horizon=16, timesteps=64, policy_freeze_duration=0, opponent_freeze_duration=0), seed=0)
    tr.train()
    torch.cuda.synchronize()
    rl = next(iter(tr.roles.values()))
    assert rl.native and rl._graphs and tr._graph is not None
    digest = hashlib.sha256(rl.fp.master.cpu().numpy().tobytes() + rl.steps.cpu().numpy().tobytes()).hexdigest()
    data = hashlib.sha256(rl.buf["pin"].float().cpu().numpy().tobytes()).hexdigest()
    print(f"RESULT rank={rank} params={digest} data={data} max_steps={int(rl.steps.max())} finite={bool(torch.isfinite(rl.fp.master).all())}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    env.close()
""")


@pytest.mark.parametrize("kl", ["0.0", "1e-7"])
def test_two_gpu_ranks_end_with_identical_parameters(tmp_path, kl):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, CAT_ROOT=str(ROOT), CAT_TEST_KL=kl, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    rows = [dict(kv.split("=") for kv in l.split()[1:]) for l in res.stdout.splitlines() if l.startswith("RESULT")]
    assert len(rows) == 2 and {r["rank"] for r in rows} == {"0", "1"}
    assert rows[0]["params"] == rows[1]["params"] and rows[0]["finite"] == "True"
    assert rows[0]["data"] != rows[1]["data"]                                # the shards really differ
    assert rows[0]["max_steps"] == rows[1]["max_steps"]
    if float(kl) > 0:
        assert int(rows[0]["max_steps"]) < 4 * 2 * 2                         # minibatches were skipped, identically on both ranks
