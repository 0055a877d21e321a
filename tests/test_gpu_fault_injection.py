"""GPU: the pooled scheduler's failure path.  A diagnostic build of the env core (-DCAT_FAULT_INJECT: slot 0 of every workgroup reserves its first ring
entry and never writes it; the waits are bounded at 2^14 looks instead of 2^22) must END every launch -- the round that waits for the entry gives up, the
idle waves' watchdog sees that nothing moves -- with CAT_DEVERR_SCHEDULER in the device error word, `check_errors()` raising like the reference does on
a bad call (entity.py:126-134 raises at once), and the process alive for the next call.  The shipped library is built without the switch."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "build" / "var" / "fault_inject.so"
SRC = ROOT / "as_cops_and_thieves_amd" / "csrc" / "cat_sim.hip"

CHILD = r"""
import sys, torch
from as_cops_and_thieves_amd import _native as nat
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import load_preset
from as_cops_and_thieves_amd.sim import CatSim, CatSimError
sim = CatSim(SimConfig(n_envs=40, n_rays=64, max_step_count=50, seed=3), [load_preset("labyrinth").compile()], device="cuda:0")
assert sim.one_tick_kernel == "step_kernel_pooled" and sim.rollout_kernel == "rollout_kernel_pooled"
sim.reset()
assert sim.device_errors() == 0            # the reset kernel does not use the ring
sim.step_fused(None, tick=0, auto_reset=True)
torch.cuda.synchronize()                   # the launch ended
flags = sim.device_errors(clear=False)
assert flags & nat.DEVERR_SCHEDULER, flags
try:
    sim.check_errors()
    raise SystemExit("check_errors() did not raise")
except CatSimError as exc:
    assert "invalid" in str(exc)
assert sim.device_errors() == 0            # cleared by check_errors
sim.rollout_fused(8, None, tick=1, auto_reset=True)   # the resident launch: slot 0 never finishes its first tick, the others finish all 8
torch.cuda.synchronize()
assert sim.device_errors() & nat.DEVERR_SCHEDULER
sim.close()
print("fault injection ok")
"""


def _build():
    from as_cops_and_thieves_amd import _native
    if LIB.exists() and all(LIB.stat().st_mtime >= f.stat().st_mtime for f in _native.sources()):
        return
    LIB.parent.mkdir(parents=True, exist_ok=True)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
           f"-I{ROOT / 'include'}", "-DCAT_QUICK_BUILD", "-DCAT_FAULT_INJECT=1", "-o", str(LIB), str(SRC)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]


def test_missing_ring_entry_raises_the_scheduler_flag_and_the_process_survives():
    _build()
    env = dict(os.environ, CAT_SIM_LIB=str(LIB), CAT_POOL="1", PYTHONPATH=str(ROOT))
    res = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert res.returncode == 0, (res.stdout[-1000:], res.stderr[-3000:])
    assert "fault injection ok" in res.stdout
