"""ctypes binding of libcat_sim.so (include/cat_sim.h).  No CPU fallback: if the shared library
is missing, or no HIP device is usable, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

from .config import C_FIELDS_F64, C_FIELDS_I32

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
LIB_PATH = PKG / "libcat_sim.so"
if os.environ.get("CAT_SIM_LIB"):          # diagnostic builds (tools/ab_kernel.sh): another build of the same source
    LIB_PATH = Path(os.environ["CAT_SIM_LIB"]).resolve()
SRC = PKG / "csrc" / "cat_sim.hip"          # one translation unit: it includes csrc/cat_sim_*.h


def sources():
    """Every file of the env core's translation unit (what a rebuild depends on, what bench.py hashes into `profile_stale`)."""
    return [SRC, *sorted((PKG / "csrc").glob("cat_sim_*.h")), ROOT / "include" / "cat_sim.h"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared"]

OUT_FIELDS = ("obs_distance", "obs_type", "hit_shape", "shared_distance", "shared_type",
              "team_positions", "reward", "terminated", "truncated", "winner")
STATE_FIELDS = ("pos", "vel", "vbias", "tc", "leaf_bb", "wall_shape", "wall_age", "wall_jn",
                "pair_age", "pair_jn", "step_count", "reset_count")
WALL_CACHE = 8
DEVERR_BAD_ACTION, DEVERR_CONTACT_DROPPED, DEVERR_SCHEDULER = 1, 2, 4

ERRORS = {-1: "CAT_ERR_BAD_CONFIG", -2: "CAT_ERR_BAD_MAP", -3: "CAT_ERR_BAD_SLOT_MAP",
          -4: "CAT_ERR_NO_DEVICE", -5: "CAT_ERR_HIP", -6: "CAT_ERR_BAD_ARG"}


class CatConfig(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in C_FIELDS_I32] + [("env_id_offset", C.c_int64), ("seed", C.c_uint64)]
                + [(n, C.c_double) for n in C_FIELDS_F64])


class CatTables(C.Structure):
    _fields_ = [("ray_dx", C.c_void_p), ("ray_dy", C.c_void_p), ("cop_reward_lut", C.c_void_p),
                ("thief_reward_lut", C.c_void_p)]


class CatOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in OUT_FIELDS]


class CatState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in STATE_FIELDS]


class NativeLibraryMissing(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP extension in-tree for gfx950 (works without a GPU: hipcc cross-compiles)."""
    stale = not LIB_PATH.exists() or any(LIB_PATH.stat().st_mtime < f.stat().st_mtime for f in sources())
    if force or stale:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, *HIPCC_FLAGS, f"-I{ROOT / 'include'}", "-o", str(LIB_PATH), str(SRC)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or res.returncode != 0:
            print(" ".join(cmd))
            print(res.stdout, res.stderr)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed building libcat_sim.so")
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the env core.")
    # torch first: it ships its own HIP/HSA runtime, and the runtime loaded FIRST serves every library of the process
    # (same soname).  Loading this library before torch would bring the system runtime in, torch's remaining bundled
    # pieces would then start a second HSA instance, and one of the two sees no device.
    import torch  # noqa: F401
    L = C.CDLL(str(LIB_PATH))
    vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
    L.cat_abi_version.restype = i32
    L.cat_last_error.restype = C.c_char_p
    L.cat_last_error.argtypes = [vp]
    if hasattr(L, "cat_one_tick_kernel"):      # absent from diagnostic builds of earlier sources (CAT_SIM_LIB, tools/ab_*.sh)
        L.cat_one_tick_kernel.restype = C.c_char_p
        L.cat_one_tick_kernel.argtypes = [vp]
    if hasattr(L, "cat_rollout_kernel"):
        L.cat_rollout_kernel.restype = C.c_char_p
        L.cat_rollout_kernel.argtypes = [vp]
    L.cat_create.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp]
    L.cat_destroy.argtypes = [vp]
    L.cat_reset.argtypes = [vp, vp, vp, vp, vp]
    L.cat_reset_done.argtypes = [vp, vp, vp]
    L.cat_step.argtypes = [vp, vp, vp, vp]
    L.cat_step_fused.argtypes = [vp, vp, u64, i32, vp, vp]
    L.cat_rollout_fused.argtypes = [vp, i32, vp, u64, i32, vp, vp]
    L.cat_get_state.argtypes = [vp, vp, vp]
    L.cat_set_state.argtypes = [vp, vp, vp]
    L.cat_random_actions.argtypes = [vp, u64, vp, vp]
    L.cat_set_seed.argtypes = [vp, u64, vp]
    L.cat_device_errors.argtypes = [vp, vp, i32, vp]
    L.cat_arm_kernel_timing.argtypes = [vp, vp, vp]
    L.cat_num_agents.argtypes = [vp]
    if hasattr(L, "cat_chunks_per_unit"):
        L.cat_chunks_per_unit.argtypes = [vp, i32]
        L.cat_chunks_per_unit.restype = i32
    L.cat_num_shapes.argtypes = [vp, i32]
    L.cat_selftest_arith.argtypes = [i32, vp, vp, vp, i32, i32, vp]
    L.cat_debug_grid_lookup.argtypes = [vp, i32, C.c_double, C.c_double, i32, vp, i32]
    L.cat_grid_build_host.argtypes = [vp, vp, vp, C.c_size_t, C.c_double, vp]
    L.cat_grid_build_host.restype = i32
    L.cat_grid_lookup_host.argtypes = [vp, C.c_double, C.c_double, i32, vp, i32]
    L.cat_grid_lookup_host.restype = i32
    L.cat_grid_bytes_host.argtypes = [vp]
    L.cat_grid_bytes_host.restype = C.c_longlong
    L.cat_grid_free_host.argtypes = [vp]
    L.cat_grid_free_host.restype = None
    L.cat_map_wall_bb_depth_host.argtypes = [vp, C.c_size_t, C.c_double]
    L.cat_map_wall_bb_depth_host.restype = i32
    for name in ("cat_create", "cat_destroy", "cat_reset", "cat_reset_done", "cat_step", "cat_step_fused", "cat_rollout_fused", "cat_get_state",
                 "cat_set_state", "cat_random_actions", "cat_set_seed", "cat_device_errors", "cat_arm_kernel_timing", "cat_num_agents", "cat_num_shapes", "cat_selftest_arith", "cat_debug_grid_lookup"):
        getattr(L, name).restype = i32
    _lib = L
    return L


EXPORTED_SYMBOLS = ("cat_abi_version", "cat_one_tick_kernel", "cat_rollout_kernel", "cat_chunks_per_unit", "cat_last_error", "cat_create", "cat_destroy", "cat_reset",
                    "cat_reset_done", "cat_step", "cat_step_fused", "cat_rollout_fused", "cat_get_state", "cat_set_state", "cat_random_actions",
                    "cat_set_seed", "cat_device_errors", "cat_arm_kernel_timing", "cat_num_agents", "cat_num_shapes", "cat_selftest_arith", "cat_debug_grid_lookup",
                    "cat_grid_build_host", "cat_grid_lookup_host", "cat_grid_bytes_host", "cat_grid_free_host", "cat_map_wall_bb_depth_host")
