#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched Cops-and-Thieves env core on MI355X.

One "step" = one tick of the whole env batch on every GPU: cat_step_fused = synthetic Philox actions
generated in the step kernel + the full BaseEnv.step pipeline + the auto-reset of the episodes that end
with the tick, ONE step_kernel launch.  Workload =
BASELINE.json configs[1]: 2 cops vs 1 thief, labyrinth map, 4096 envs per GPU, 64 rays/agent.
Env slots shard across GPUs with no data-path collective (weak scaling); only the timing uses a
barrier + MAX all-reduce.  Prints ONE JSON line on rank 0; at N = 1 it also carries `extra`: the other
BASELINE shapes measured in the same run (kernel_ms each) and the learner's collect + update rate on the headline workload
(SURVEY 8f rank 2); the headline stays configs[1].
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_env_step(A: int, R: int) -> int:
    """actions 4A + body state r/w 2*48A + counters 8 + obs 3AR +
    team-shared 6R + team positions 4A + rewards 4A."""
    return 108 * A + 3 * A * R + 6 * R + 8


def host_cpu_share() -> int:
    """Threads for the all-cores leg: the smallest of the scheduler affinity, the cgroup CPU quota (cpu.max) and 16 -- a GPU
    box exposes all 256 host threads to every tenant but gives one GPU's job a 16-core share (256 threads on 256 envs
    measured 1 k env-steps/s against 0.47 M with 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(cfg, cmap, budget_s: float = 12.0):
    """Times the CPU oracle (a port, not Pymunk: pymunk is not installable here) on this host:
    single thread (the like-for-like of the reference's single process), then all cores."""
    import numpy as np
    from dataclasses import replace
    from oracle import cat_oracle
    n = 256
    c = replace(cfg, n_envs=n)
    res = {}
    ncores = host_cpu_share()
    for label, threads in (("1core", 1), ("allcores", ncores)):
        cat_oracle.lib().cato_set_threads(threads)
        sim = cat_oracle.OracleSim(c, [cmap])
        sim.reset()
        acts = [sim.random_actions(t) for t in range(64)]
        t0 = time.perf_counter()
        ticks = 0
        while time.perf_counter() - t0 < budget_s / 2:
            out = sim.step(acts[ticks % 64])
            sim.reset(mask=out["terminated"].copy())
            ticks += 1
        dt = time.perf_counter() - t0
        res[label] = (n * ticks / dt, threads, ticks)
    cat_oracle.lib().cato_set_threads(1)
    v1, _, ticks1 = res["1core"]
    va, ca, _ = res["allcores"]
    out = {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{n} envs x {ticks1} ticks of the same workload, CPU restatement (not Pymunk), 1 thread",
           "allcores": {"value": va, "cores": ca}}
    # the single-process Pymunk figure, when this host has the package (this image does not: the
    # attempt is recorded either way).  oracle/pymunk_double.py drives Pymunk with the reference's own call sequence.
    from oracle import pymunk_double
    if pymunk_double.available():
        spec = cmap.spec
        try:
            rate, ticks = pymunk_double.time_random_rollout(spec["map_data"], cfg.n_rays, 6.0, roster=spec.get("roster"),
                                                            start_positions=spec.get("start_positions"), scale=spec.get("scale"))
            out["pymunk_1core"] = {"value": rate, "unit": "env-steps/s", "cores": 1, "kind": "reference dependency (Pymunk) through "
                                   "oracle/pymunk_double.py", "sample": f"1 env x {ticks} ticks, random actions"}
        except Exception as exc:   # noqa: BLE001
            out["pymunk_1core"] = {"value": None, "error": repr(exc)[:200]}
    else:
        out["pymunk_1core"] = {"value": None, "error": "import pymunk failed: not installed on this host"}
    return out


def event_stride(steps: int) -> int:
    """Which step_kernel launches of the timed region carry their own pair of HIP events: every 8th on a long run; NONE (0) when
    K <= 40 -- a dispatch with events attached costs ~4 us of host / command-processor time (K = 20: 37.4 us per step with every
    launch bracketed, 34.5 with every 2nd, 32.4 with none), which a 20-step region does not absorb.  A short region is
    timed by ONE pair of events recorded on the kernel's stream around all K launches instead (timed_steps)."""
    if os.environ.get("CAT_BENCH_EVENT_STRIDE"):      # measurement of the instrumentation's own cost
        return max(0, int(os.environ["CAT_BENCH_EVENT_STRIDE"]))
    return 0 if steps <= 40 else 8


class HipEvents:
    """Raw hipEvent_t handles from the HIP runtime torch has already loaded (torch.cuda.Event creates its
    handle lazily and cannot be attached to a dispatch)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        path = "libamdhip64.so"
        with open("/proc/self/maps") as f:      # the very copy of the runtime this process already uses
            for line in f:
                if "libamdhip64.so" in line:
                    path = line.split()[-1]
                    break
        self.lib = C.CDLL(path)
        self.lib.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

    def create(self) -> int:
        h = self.C.c_void_p()
        if self.lib.hipEventCreate(self.C.byref(h)) != 0:
            raise RuntimeError("hipEventCreate failed")
        return h.value

    def record(self, event: int, stream: int) -> None:
        if self.lib.hipEventRecord(event, stream) != 0:
            raise RuntimeError("hipEventRecord failed")

    def elapsed_ms(self, a: int, b: int) -> float:
        ms = self.C.c_float()
        rc = self.lib.hipEventElapsedTime(self.C.byref(ms), a, b)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed ({rc})")
        return float(ms.value)


EXTRA_WORKLOADS = (   # the other BASELINE.json shapes, measured in the same run beside the headline (N = 1 only)
    ("agh-map 2v1 x4096 (configs[2] per-GPU shard)", dict(map="agh-map", cops=2, thieves=1, envs=4096)),
    ("grandbyrinth 3v2 x8192 (configs[3])", dict(map="grandbyrinth", cops=3, thieves=2, envs=8192)),
    ("five maps mixed 2v1 x16384 (configs[4])", dict(map="mixed", cops=2, thieves=1, envs=16384)),
    ("labyrinth 2v1 x4096, every agent spawned inside the maze", dict(map="labyrinth-inside", cops=2, thieves=1, envs=4096)),
    ("labyrinth 2v1 x4096 with the reference's own 90-ray sensor (entity.py:86)", dict(map="labyrinth", cops=2, thieves=1, envs=4096, rays=90)),
    ("squarinth 1v1 x4096, 90 rays: the reference's own default roster, map and sensor (configs[0], batched)", dict(map="squarinth", cops=1, thieves=1, envs=4096, rays=90)),
)


def build_sim(map_name: str, cops: int, thieves: int, envs: int, rays: int, rank: int, dev):
    """CatSim for one bench workload: a preset map, or "mixed" = all five maps interleaved across env slots."""
    import numpy as np
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import load_preset
    from as_cops_and_thieves_amd.sim import CatSim
    if map_name == "mixed":   # BASELINE configs[4]
        names = ["agh-map", "grandbyrinth", "labyrinth", "lbirinth", "squarinth"]
        cmaps = [load_preset(n, cops, thieves).compile() for n in names]
        slot = (np.arange(envs) % len(cmaps)).astype(np.int32)
        cmap = cmaps[2]
    else:
        cmaps, slot = [load_preset(map_name, cops, thieves).compile()], None
        cmap = cmaps[0]
    cfg = SimConfig(n_envs=envs, n_cops=cops, n_thieves=thieves, n_rays=rays, max_step_count=400, seed=0,
                    env_id_offset=rank * envs)
    return CatSim(cfg, cmaps, slot, device=dev), cfg, cmap


def timed_steps(sim, steps: int, warmup: int, fence, hip: "HipEvents", first_tick: int = 0):
    """W untimed + K timed rollout steps (cat_step_fused: in-kernel Philox actions + tick + auto-reset, ONE launch per
    step).  Long regions: every event_stride(K)-th step_kernel launch carries a pair of HIP events attached to the
    dispatch itself (hipExtLaunchKernelGGL start/stop events, on the stream the kernel is launched on), so kernel_ms is
    the kernel's own duration, as in a rocprofv3 kernel trace.  Short regions (K <= 40, the driver's --steps 20): one pair of
    HIP events recorded on that stream around ALL K launches, kernel_ms = their span / K -- every launch of the region is timed,
    the figure includes the gaps between consecutive launches (it is an upper bound of the trace's average), and no launch pays for
    instrumentation.  Returns (seconds of the timed region on this rank, mean kernel ms, launches timed, method)."""
    for t in range(warmup):
        sim.step_fused(None, tick=first_tick + t, auto_reset=True)
    stride = event_stride(steps)
    ev = {k: (hip.create(), hip.create()) for k in range(0, steps, stride)} if stride else {}
    span = (hip.create(), hip.create())
    stream = sim._stream()
    fence()
    t0 = time.perf_counter()
    hip.record(span[0], stream)
    for k in range(steps):
        if k in ev:
            sim.arm_kernel_timing(*ev[k])
        sim.step_fused(None, tick=first_tick + warmup + k, auto_reset=True)
    hip.record(span[1], stream)
    fence()
    elapsed = time.perf_counter() - t0
    if ev:
        return elapsed, sum(hip.elapsed_ms(a, b) for a, b in ev.values()) / len(ev), len(ev), "HIP events attached to the dispatch"
    return elapsed, hip.elapsed_ms(*span) / steps, steps, "one HIP event pair on the kernel's stream around the K launches, / K"


def timed_rollout(sim, T: int, fence, hip: "HipEvents", first_tick: int, reps: int = 4):
    """The resident rollout launch (cat_rollout_fused: T ticks per launch, map staged once, state records kept in LDS, every tick's
    outputs written to [T, N, ...] buffers) on the batch as it stands: one untimed launch, then `reps` timed ones, each with a pair
    of HIP events attached to its dispatch.  Returns (seconds of the timed region on this rank, kernel ms per TICK, launches timed)."""
    sim.rollout_fused(T, None, tick=first_tick, auto_reset=True)
    ev = [(hip.create(), hip.create()) for _ in range(reps)]
    fence()
    t0 = time.perf_counter()
    for r, (a, b) in enumerate(ev):
        sim.arm_kernel_timing(a, b)
        sim.rollout_fused(T, None, tick=first_tick + (r + 1) * T, auto_reset=True)
    fence()
    elapsed = time.perf_counter() - t0
    return elapsed, sum(hip.elapsed_ms(a, b) for a, b in ev) / (reps * T), reps


def source_sha16() -> str:
    """sha256 of the shipped env-core source, first 16 hex digits: what a committed profile must have been collected on to describe this build."""
    import hashlib
    h = hashlib.sha256()
    csrc = ROOT / "as_cops_and_thieves_amd" / "csrc"
    for f in [csrc / "cat_sim.hip", *sorted(csrc.glob("cat_sim_*.h"))]:    # the env core's one translation unit
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def compute_block(valu):
    """The compute-side roofline of a replayed profile (the path is bound by VALU issue, not by HBM): the share of the SIMDs' vector issue
    slots in use x the share of lanes active in an issued instruction = the fraction of lane-issue slots doing work."""
    if not valu:
        return None
    busy, util = valu.get("valu_issue_busy_frac"), valu.get("lane_utilisation")
    return {"bound": "valu_issue", "valu_issue_busy": busy, "lane_utilisation": util,
            "frac": (busy * util) if busy is not None and util is not None else None,
            "valu_per_env_step": valu.get("valu_insts_per_wave"), "salu_per_env_step": valu.get("salu_insts_per_wave"),
            "lds_per_env_step": valu.get("lds_insts_per_wave"), "lds_bank_conflict_per_active_cycle": valu.get("lds_bank_conflict_per_active_cycle"),
            "what": "frac = valu_issue_busy x lane_utilisation: 1.0 = every SIMD issues a 64-lane vector instruction in every issue slot (FP64 geometry and "
                    "integer bookkeeping alike); instruction counts are wave-instructions per env-step"}


def find_profile(wl: dict, kernel: str):
    """The newest committed PMC profile (profiles/r*_traffic*.json) of this workload and kernel, or None.  PMC counters need
    rocprofv3 passes of their own, so the bench line REPLAYS them and says so."""
    default_key = {"map": "labyrinth", "envs": 4096, "rays": 64, "cops": 2, "thieves": 1}
    for tfile in sorted((ROOT / "profiles").glob("r*_traffic*.json"), reverse=True):
        prof = json.loads(tfile.read_text())
        if prof.get("workload_key", default_key) == wl and prof.get("kernel", "tick_kernel") == kernel:
            prof["_file"] = tfile.name
            prof["_stale"] = prof.get("source_sha16") != source_sha16()   # collected on another cat_sim.hip than the one shipped (or before r05: unrecorded)
            return prof
    return None


def replay_profile(entry: dict, wl: dict, kernel: str, ticks_per_launch: int = 1) -> None:
    """Adds the replayed HBM traffic (bytes per env batch TICK) and VALU figures of a committed profile to an `extra` entry."""
    prof = find_profile(wl, kernel)
    if prof is None:
        entry.update({"traffic": None, "traffic_source": None})
        return
    entry["traffic"] = prof["hbm_bytes_per_launch"] / prof.get("ticks_per_launch", 1)
    entry["traffic_source"] = f"profiles/{prof['_file']} (committed rocprofv3 --pmc passes; replayed, not measured in this run)"
    entry["traffic_over_algorithmic"] = entry["traffic"] / (algorithmic_bytes_per_env_step(wl["cops"] + wl["thieves"], wl["rays"]) * wl["envs"])
    entry["profile_stale"] = prof["_stale"]
    entry["profile_git_head"] = prof.get("git_head")
    if prof.get("valu"):
        entry["lane_utilisation"] = prof["valu"].get("lane_utilisation")
        entry["valu"] = prof["valu"]
        entry["compute"] = compute_block(prof["valu"])


MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (the 5 PF headline figure includes 2:1 sparsity)


def learner_flops(rays: int, agents: int, epochs: int) -> dict:
    """FLOPs (2 per multiply-add) of the reference's LSTM pair per agent and sample, from its layer sizes (lstm_policy_net.py:28-53,
    lstm_value_net.py:46-75; models.py): Conv1d(C->64,k5,s2) Conv1d(64->32,k5,s3) Linear(32*L2->256) LSTM(256->128, 1 or 2 layers) and the heads.
    Collection = one forward of both networks per agent and env-step; the update = `epochs` passes of forward + backward (2x a forward: data and
    weight gradients) over every stored sample."""
    l1 = (rays - 5) // 2 + 1
    l2 = (l1 - 5) // 3 + 1
    trunk = lambda c: l1 * 64 * c * 5 + l2 * 32 * 64 * 5 + 32 * l2 * 256
    policy = trunk(2) + 4 * 128 * (256 + 128) + (128 * 128 + 128 * 64 + 64 * 4)
    value = trunk(4) + 4 * 128 * (256 + 128) + 4 * 128 * (128 + 128) + (128 * 256 + 256 * 128 + 128 * 64 + 64)
    fwd = 2.0 * (policy + value) * agents
    return {"forward_per_env_step": fwd, "collect_per_env_step": fwd, "update_per_env_step": 3.0 * fwd * epochs,
            "policy_macs_per_sample": policy, "value_macs_per_sample": value}


def learner_throughput(map_name: str, n_envs: int, rays: int, horizon: int = 16, rank: int = 0, world: int = 1, reduce_device=None) -> dict:
    """SURVEY 8(f) rank 2 beside the headline: env-steps/s of the MAPPO trainer on the same env workload -- rollout
    collection (env tick + the six stacked LSTM networks per tick) plus the PPO update of ``CFG_AGENT`` (4 epochs x 4
    minibatches), both as replayed HIP graphs over libcat_learn.so.  3 untimed rollout+update rounds (the graphs are
    captured there), then 5 timed ones.  A failure is reported, not raised: the headline does not depend on it."""
    import time
    import torch
    try:
        from as_cops_and_thieves_amd import VecCopsEnv, load_preset
        from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, TrainerConfig
        # world > 1: every rank trains on its own env shard inside the job's process group -- the trainer all-reduces its
        # [G, P + 1] gradient | KL buffer every optimiser step (RCCL over xGMI when the group is nccl)
        env = tr = None
        build_error = None
        try:    # what can fail on ONE rank only (memory, a build problem, an unsupported shape) fails here, before the trainer's first collective
            env = VecCopsEnv(load_preset(map_name), num_envs=n_envs, num_rays=rays, max_step_count=400, env_id_offset=rank * n_envs)
            tr = MAPPOTrainer(env, None, TrainerConfig(horizon=horizon), seed=0)
        except Exception as exc:   # noqa: BLE001
            build_error = exc
        if world > 1:   # every rank learns whether every rank is ready: all run the leg or all skip it (a rank missing from the all-reduces would hang the rest)
            import torch.distributed as dist
            ok = torch.tensor([0 if build_error else 1], dtype=torch.int32, device=reduce_device if reduce_device is not None else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if env is not None:
                    env.close()
                return {"value": None, "error": repr(build_error)[:200] if build_error else "another rank could not build its trainer"}
        elif build_error is not None:
            raise build_error
        for _ in range(3):
            tr.collect(); tr.update()
        torch.cuda.synchronize()
        rounds, tc, tu = 5, 0.0, 0.0
        for _ in range(rounds):
            t0 = time.perf_counter(); tr.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
            tr.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
            tc += t1 - t0; tu += t2 - t1
        steps = world * rounds * tr.tcfg.horizon * n_envs
        if world > 1:
            from as_cops_and_thieves_amd.sharding import max_over_ranks
            tc, tu = max_over_ranks(tc, device=reduce_device), max_over_ranks(tu, device=reduce_device)
        out = {"value": steps / (tc + tu), "unit": "env-steps/s", "rounds": rounds, "horizon": tr.tcfg.horizon,
               "collect_ms": 1e3 * tc / rounds, "update_ms": 1e3 * tu / rounds, "dtype": "bf16",
               "graphs": bool(tr._graph is not None and all(rl._graphs for rl in tr.roles.values()))}
        # MFMA roofline of the learner: the networks' FLOPs per env-step from their layer sizes against the time the two phases take
        fl = learner_flops(rays, len(env.possible_agents), max(rl.cfg.learning_epochs for rl in tr.roles.values()))
        per_round = tr.tcfg.horizon * n_envs           # env-steps of one rank's rollout
        out["mfma_roofline"] = {
            "bound": "mfma", "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "flops_per_env_step": fl["collect_per_env_step"] + fl["update_per_env_step"],
            "collect_flops_per_env_step": fl["collect_per_env_step"], "update_flops_per_env_step": fl["update_per_env_step"],
            "achieved": (fl["collect_per_env_step"] + fl["update_per_env_step"]) * per_round / ((tc + tu) / rounds) / 1e12,
            "achieved_update": fl["update_per_env_step"] * per_round / (tu / rounds) / 1e12,
            "achieved_collect": fl["collect_per_env_step"] * per_round / (tc / rounds) / 1e12,
            "what": "per GPU; network FLOPs only (2 per multiply-add; update = epochs x (forward + 2x backward)); the collect phase also holds the env ticks"}
        out["mfma_roofline"]["frac"] = out["mfma_roofline"]["achieved"] / MFMA_BF16_PEAK_TFLOPS
        out["mfma_roofline"]["frac_update"] = out["mfma_roofline"]["achieved_update"] / MFMA_BF16_PEAK_TFLOPS
        out["collect_us_per_tick"] = 1e6 * tc / rounds / tr.tcfg.horizon
        if world > 1:
            import torch.distributed as dist
            out.update({"ranks": world, "envs_per_gpu": n_envs, "allreduce_backend": dist.get_backend(),
                        "allreduces_per_update": 16, "param_digest": tr.param_digest()[:16]})
        env.close()
        return out
    except Exception as exc:   # noqa: BLE001
        if world > 1:   # the other ranks are inside the trainer's collectives: a swallowed error here would leave them waiting until the RCCL
            raise       # timeout -- let torchrun tear the group down instead
        return {"value": None, "error": repr(exc)[:200]}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` typed as a plain command: this process has not touched the GPU (torch is not even imported
    yet) and starts N fresh rank processes, one per GPU, through torch.distributed.run as a CHILD process (never an exec);
    rank 0's JSON line goes straight to the inherited stdout.  Returns the launcher's exit status."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096, help="env slots per GPU")
    ap.add_argument("--rays", type=int, default=64)
    ap.add_argument("--map", default="labyrinth")
    ap.add_argument("--cops", type=int, default=2)
    ap.add_argument("--thieves", type=int, default=1)
    ap.add_argument("--burn-in", type=int, default=600, help="untimed ticks after the reset and BEFORE the warm-up: one and a half episode "
                    "lengths, so that a short timed region (the driver's --steps 20 --warmup 5) sees the batch in the middle of an episode, "
                    "agents spread over the map, and not the ticks after a reset (every env of a random-action labyrinth batch times out at "
                    "tick 400 together: 400 would land on the next reset); kernel time there = the whole-episode average (DESIGN 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE shapes (the `extra` object)")
    ap.add_argument("--rollout-ticks", type=int, default=64, help="T of the resident-rollout legs in `extra` (cat_rollout_fused)")
    ap.add_argument("--shape-only", action="store_true", help="profiling runs of one shape (tools/collect_profiles.sh): this workload's one-launch-per-tick "
                    "region and its resident-rollout leg only -- no other shapes, no learner legs, no CPU baseline")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and (world_env is None or (world_env == "1" and "RANK" not in os.environ)):
        sys.exit(launch_ranks(args.gpus))          # plain `python bench.py --gpus N`: start the N ranks ourselves

    import torch
    from as_cops_and_thieves_amd.sharding import max_over_ranks

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit(f"bench.py --gpus {args.gpus} was started inside a {world}-rank group (WORLD_SIZE={world}): the rank count "
                 f"must equal --gpus; run plain `python bench.py --gpus {args.gpus}` (it starts its own ranks) or "
                 f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus}")
    rehearse = os.environ.get("CAT_BENCH_REHEARSE") == "1"   # flow check on a 1-GPU box: gloo, ranks share the GPU
    timing_backend = None
    rccl_ranks = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            local_rank %= torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
            timing_backend = "gloo (CAT_BENCH_REHEARSE=1: ranks share one GPU)"
        else:
            torch.cuda.set_device(local_rank)
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL
                probe = torch.ones(1, device=torch.device("cuda", local_rank))
                dist.all_reduce(probe)                                                        # fails here if RCCL cannot start
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"RCCL all-reduce of ones over {world} ranks returned {probe.item()}")
                timing_backend = "nccl (RCCL)"
                rccl_ranks = dist.get_world_size()    # what the RCCL group itself reports (and the sum above confirmed)
            except Exception as exc:   # the simulator needs no collective: only the timing barrier / MAX does
                print(f"[bench] RCCL unavailable ({exc!r}); timing reductions over gloo", file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("gloo")
                rehearse = True   # CPU tensors for the reductions
                timing_backend = f"gloo (RCCL could not start: {type(exc).__name__})"
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    def fence() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    hip = HipEvents()
    sim, cfg, cmap = build_sim(args.map, args.cops, args.thieves, args.envs, args.rays, rank, dev)
    sim.reset()
    for t in range(args.burn_in):
        sim.step_fused(None, tick=t, auto_reset=True)
    elapsed, tick_ms, n_timed, tick_method = timed_steps(sim, args.steps, args.warmup, fence, hip, first_tick=args.burn_in)
    elapsed = max_over_ranks(elapsed, device=None if rehearse else dev)   # the slowest rank bounds the whole-job rate
    TR = args.rollout_ticks
    roll = None
    if not args.no_extras or args.shape_only:   # the resident rollout launch on the same running batch (value stays one launch per tick)
        r_el, r_kms, r_n = timed_rollout(sim, TR, fence, hip, first_tick=args.burn_in + args.warmup + args.steps)
        r_el = max_over_ranks(r_el, device=None if rehearse else dev)
        roll = (r_el, r_kms, r_n)
    episodes = int(sim.get_state()["reset_count"].sum().item())
    dev_errors = {"headline": sim.device_errors()}   # CAT_DEVERR_* raised by any launch of the burn-in, the timed region or the resident leg (0 = none)
    one_tick_kernel = sim.one_tick_kernel      # "step_kernel" / "step_kernel_pooled": one tick through the resident rollout scheduler (rounds 1 - 3: "tick_kernel")
    rollout_kernel = sim.rollout_kernel        # the resident launch of the same sim
    sim.close()
    offsets = [cfg.env_id_offset]
    if world > 1:   # which global env ids each rank simulated (disjoint contiguous shards)
        offsets = [None] * world
        dist.all_gather_object(offsets, int(cfg.env_id_offset))

    copy_gbs = None
    if rank == 0 and world == 1:   # the HBM peak as a stream copy measures it on this box (read + write bytes)
        a = torch.empty(1 << 28, dtype=torch.float32, device=dev); b = torch.empty_like(a)   # 1 GiB each
        b.copy_(a); torch.cuda.synchronize(dev)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            b.copy_(a)
        c1.record(); torch.cuda.synchronize(dev)
        copy_gbs = 10 * 2 * a.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del a, b
    def measure_extra(w: dict) -> dict:
        """One of the other BASELINE shapes, on every rank of the job (its own env shard), same fence + MAX-over-ranks rule."""
        s2, c2, m2 = build_sim(w["map"], w["cops"], w["thieves"], w["envs"], w.get("rays", args.rays), rank, dev)
        s2.reset()
        k_steps = 300                                   # own step counts: the driver's --steps 20 --warmup 5 would only
        e2, k2, _, _ = timed_steps(s2, k_steps, 100, fence, hip)   # see the first ticks after the reset
        e2 = max_over_ranks(e2, device=None if rehearse else dev)
        bytes2 = algorithmic_bytes_per_env_step(c2.n_agents, c2.n_rays) * c2.n_envs
        r_el, r_kms, r_n = timed_rollout(s2, TR, fence, hip, first_tick=100 + k_steps)
        r_el = max_over_ranks(r_el, device=None if rehearse else dev)
        kern2, roll2 = s2.one_tick_kernel, s2.rollout_kernel
        dev_errors[w["map"] + (f" {w['rays']} rays" if "rays" in w else "")] = s2.device_errors()
        s2.close()
        key = {"map": w["map"], "envs": w["envs"], "rays": w.get("rays", args.rays), "cops": w["cops"], "thieves": w["thieves"]}
        ent = {"value": world * c2.n_envs * k_steps / e2, "unit": "env-steps/s", "steps": k_steps,
               "ms_per_step": 1e3 * e2 / k_steps, "kernel": kern2, "kernel_ms": k2, "envs_per_gpu": c2.n_envs,
               "roofline_frac": bytes2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        replay_profile(ent, key, kern2)
        res = {"T": TR, "launches_timed": r_n, "value": world * c2.n_envs * TR * r_n / r_el, "unit": "env-steps/s",
               "kernel_ms_per_tick": r_kms, "roofline_frac": bytes2 / (r_kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "what": "cat_rollout_fused: T ticks per launch, map staged once, state records resident in LDS, every tick's outputs written"}
        res["kernel"] = roll2
        replay_profile(res, key, roll2)
        ent["resident_rollout"] = res
        return ent

    extra = None
    if world > 1 and not args.no_extras and not args.shape_only:    # BASELINE configs[2]: agh-map, 32768 envs over 8 GPUs = 4096 per GPU, all ranks
        label, w = EXTRA_WORKLOADS[0]
        extra = {f"agh-map 2v1 x4096 per GPU (configs[2]), {world} GPUs": measure_extra(w)}
        # configs[2]'s other half: MAPPO on the sharded envs, gradients all-reduced under the SAME process group the timing uses
        lt = learner_throughput("agh-map", w["envs"], args.rays, rank=rank, world=world, reduce_device=None if rehearse else dev)
        lt["rccl_ranks"] = rccl_ranks
        extra[f"learner_collect_plus_update, {world} GPUs (agh-map 2v1 x4096 per GPU, CFG_AGENT)"] = lt
    if rank == 0 and world == 1 and not args.no_extras and not args.shape_only:   # the other BASELINE shapes, same process, same box
        extra = {label: measure_extra(w) for label, w in EXTRA_WORKLOADS}
        extra["learner_collect_plus_update"] = learner_throughput(args.map, cfg.n_envs, args.rays)
        extra["learner_collect_plus_update, 90 rays"] = learner_throughput(args.map, cfg.n_envs, 90)
        # 128-tick rollouts (8 BPTT windows per env and update): the setting with which the cops learn to catch random
        # thieves (tests/test_gpu_mappo.py::test_cops_learn_to_catch_random_thieves_on_squarinth; TrainerConfig's default)
        extra["learner_collect_plus_update, 128-tick rollouts"] = learner_throughput(args.map, cfg.n_envs, args.rays, horizon=128)
    if rank == 0:
        A, R = cfg.n_agents, cfg.n_rays
        bytes_launch = algorithmic_bytes_per_env_step(A, R) * cfg.n_envs
        achieved = bytes_launch / (tick_ms * 1e-3) / 1e9
        # HBM bytes per launch / VALU figures: NOT measured in this run (PMC counters need rocprofv3 passes of their own);
        # replayed from the committed PMC summary of this exact workload and labelled as such
        traffic = valu = traffic_source = traffic_regime = traffic_from_reset = None
        wl = {"map": args.map, "envs": cfg.n_envs, "rays": R, "cops": args.cops, "thieves": args.thieves}
        prof = find_profile(wl, one_tick_kernel)   # newest round whose workload and kernel match
        if prof is not None:
            traffic, valu = prof["hbm_bytes_per_launch"], prof.get("valu")
            traffic_regime = prof.get("regime", "from reset (--burn-in 0: 25 launches straight after the reset)")
            traffic_from_reset = prof.get("hbm_bytes_per_launch_from_reset")
            traffic_source = (f"profiles/{prof['_file']} (committed rocprofv3 --pmc passes of `bench.py --steps 20 --warmup 5 "
                              f"--burn-in {prof.get('burn_in', 0)}`; replayed, not measured in this run)")
        if roll is not None:
            r_el, r_kms, r_n = roll
            res = {"T": TR, "launches_timed": r_n, "value": world * cfg.n_envs * TR * r_n / r_el, "unit": "env-steps/s",
                   "kernel": rollout_kernel, "kernel_ms_per_tick": r_kms, "kernel_ms_method": "HIP events attached to each dispatch, / T",
                   "roofline_frac": bytes_launch / (r_kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "envs_per_gpu": cfg.n_envs,
                   "what": "cat_rollout_fused: T ticks per launch on the same running batch -- map staged once, state records resident in "
                           "LDS for the T ticks, EVERY tick's outputs written to [T, N, ...] buffers; the headline `value` stays one launch per tick"}
            replay_profile(res, wl, rollout_kernel)
            extra = dict(extra or {})
            extra[f"{args.map} {args.cops}v{args.thieves} x{cfg.n_envs}, T={TR} resident rollout"] = res
        stale = bool(prof["_stale"]) if prof is not None else None
        line = {
            "metric": "env-steps/sec (whole node) at 4096 parallel envs, 2v1 agents, 64-ray sensors",
            "value": world * cfg.n_envs * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.cops} cops vs {args.thieves} thieves, {args.map} map "
                                   f"({cmap.n_shapes} hull walls / {cmap.n_planes} planes), {cfg.n_envs} envs per GPU, "
                                   f"{R} rays/agent, dt=1/60, max_step_count=400, Philox random actions, auto-reset",
                       "envs_per_gpu": cfg.n_envs, "rays": R, "agents": A, "map": args.map,
                       "parallelism": f"env-sharded x{world}, no data-path collective",
                       "episodes_reset_per_gpu": episodes, "burn_in_steps": args.burn_in, "env_id_offsets": offsets},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_regime": traffic_regime,
                         "traffic_from_reset": traffic_from_reset, "traffic_source": traffic_source,
                         "kernel": one_tick_kernel, "kernel_ms": tick_ms, "kernel_launches_timed": n_timed, "kernel_ms_method": tick_method,
                         "algorithmic_bytes_per_launch": bytes_launch, "valu": valu, "valu_source": traffic_source,
                         "compute": compute_block(valu), "profile_stale": stale, "profile_source_sha16": prof.get("source_sha16") if prof else None,
                         "profile_git_head": prof.get("git_head") if prof else None, "shipped_source_sha16": source_sha16(),
                         "hbm_stream_copy_GBs": copy_gbs,
                         "frac_of_stream_copy": (achieved / copy_gbs) if copy_gbs else None,
                         "note": "path is FP64-VALU/LDS bound, not HBM bound (SURVEY 8d); fraction reported as contracted"},
        }
        if timing_backend:
            line["config"]["timing_reductions"] = timing_backend
            line["config"]["rccl_ranks"] = rccl_ranks     # what the RCCL group reported; null = the timing went over gloo
        if extra is not None:
            line["extra"] = extra
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only
            line["cpu_baseline"] = cpu_baseline(cfg, cmap)
    # device-side error words of every sim of every rank (the reference raises at once on a bad call, entity.py:126-134; the launches here are
    # asynchronous, so the kernels flag and the host reads the word after each leg): the line carries the OR over ranks and legs, and a run that
    # raised any flag exits non-zero after printing it
    err_or = 0
    for v in dev_errors.values():
        err_or |= int(v)
    if world > 1:
        flags = [None] * world
        dist.all_gather_object(flags, err_or)
        err_or = 0
        for v in flags:
            err_or |= int(v)
    if rank == 0:
        line["device_errors"] = err_or
        if err_or:
            line["device_errors_by_leg"] = dev_errors
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if err_or:
        sys.exit(f"bench.py: device-side error flags 0x{err_or:x} were raised (CAT_DEVERR_*: 1 bad action, 2 contact dropped, 4 scheduler): results invalid")


if __name__ == "__main__":
    main()
