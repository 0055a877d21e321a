import sys, os
sys.path.insert(0, '/root/repo')
import torch
from as_cops_and_thieves_amd import VecCopsEnv, load_preset
from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, RoleConfig, TrainerConfig, compute_gae
rc = RoleConfig(random_timesteps=0, learning_starts=0, learning_rate=3e-4)
env = VecCopsEnv(load_preset("squarinth"), num_envs=1024, num_rays=64, max_step_count=400, seed=1)
tc = TrainerConfig(policy_freeze_duration=0, opponent_freeze_duration=0, graph_rollout=False, graph_update=True)
tr = MAPPOTrainer(env, {"cop": rc, "thief": rc}, tc, seed=0)
tr.collect(); tr.update()          # captures
for it in range(2):
    tr.collect()
    torch.cuda.synchronize()
    for r, rl in tr.roles.items():
        rl.start = tr._start_buf
        rl.idx.copy_(torch.randperm(rl.N, device=rl.device)[:rl.B])
        rl._step_forward_backward(); torch.cuda.synchronize()
        ar_e, st_e = rl.ar.clone(), rl.stat.clone()
        rl.ar.zero_()
        rl._graphs[0].replay(); torch.cuda.synchronize()
        ar_g, st_g = rl.ar.clone(), rl.stat.clone()
        rl._graphs[0].replay(); torch.cuda.synchronize()
        ar_g2 = rl.ar.clone()
        d = (ar_g - ar_e)[:, :-1]
        print(it, r, "eager |g|", ar_e[:, :-1].norm(dim=1).tolist(), "graph |g|", ar_g[:, :-1].norm(dim=1).tolist(), "replay2 |g|", ar_g2[:, :-1].norm(dim=1).tolist(),
              "diff policy", (d * rl.col_policy).norm(dim=1).tolist(), "diff value", (d * rl.col_value).norm(dim=1).tolist(),
              "loss e", st_e[:2].tolist(), "loss g", st_g[:2].tolist(), "kl e/g", ar_e[:, -1].tolist(), ar_g[:, -1].tolist(), "finite", bool(torch.isfinite(ar_g).all()), flush=True)
        # which parameters differ most
        worst = []
        for n, (o, k, shp) in rl.fp.offsets.items():
            worst.append((float(d[:, o:o + k].norm()), float(ar_e[:, o:o + k].norm()), n))
        worst.sort(reverse=True)
        print("   ", [(f"{a:.2e}", f"{b:.2e}", n) for a, b, n in worst[:6]], flush=True)
    tr.update()
