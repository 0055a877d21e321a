import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from as_cops_and_thieves_amd import _native as nat
nat.LIB_PATH = nat.PKG / sys.argv[1]
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import load_preset
from as_cops_and_thieves_amd.sim import CatSim
name = sys.argv[2]; N = 4096
sim = CatSim(SimConfig(n_envs=N, n_rays=64, seed=0), [load_preset(name).compile()])
sim.reset()
for t in range(30): sim.step_fused(None, t, auto_reset=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(200): sim.step_fused(None, 30 + t, auto_reset=False)
e1.record(); torch.cuda.synchronize()
print(sys.argv[1], name, f"{1e3 * e0.elapsed_time(e1) / 200:.1f} us/tick")
