"""ctypes wrapper around oracle/_build/libcat_oracle.so — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see cat_oracle.h): this checks the HIP library against a CPU restatement of
the reference + Chipmunk2D's published algorithm, not against a running Pymunk.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from . import cat_oracle_host as host

C_FIELDS_I32, C_FIELDS_F64 = host.CONFIG_I32, host.CONFIG_F64

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "_build" / "libcat_oracle.so"
K_WALL = 8


def build(force: bool = False) -> Path:
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < (HERE / "cat_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE)] + (["-B"] if force else []), check=True,
                       capture_output=True)
    return LIB_PATH


class _Config(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in C_FIELDS_I32] + [("env_id_offset", C.c_int64), ("seed", C.c_uint64)]
                + [(n, C.c_double) for n in C_FIELDS_F64])


class _Tables(C.Structure):
    _fields_ = [("ray_dx", C.c_void_p), ("ray_dy", C.c_void_p), ("cop_lut", C.c_void_p), ("thief_lut", C.c_void_p)]


_OUT_FIELDS = ("obs_distance", "obs_type", "hit_shape", "shared_distance", "shared_type",
               "team_positions", "reward", "terminated", "truncated", "winner")
_STATE_FIELDS = ("pos", "vel", "vbias", "tc", "leaf_bb", "wall_shape", "wall_age", "wall_jn",
                 "pair_age", "pair_jn", "step_count", "reset_count")


class _Outputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _OUT_FIELDS]


class _State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _STATE_FIELDS]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB_PATH))
        _lib.cato_last_error.restype = C.c_char_p
        _lib.cato_f64_to_f16.restype = C.c_uint16
        _lib.cato_f64_to_f16.argtypes = [C.c_double]
        _lib.cato_f16_to_f64.restype = C.c_double
        _lib.cato_f16_to_f64.argtypes = [C.c_uint16]
        _lib.cato_obs_distance_f16.restype = C.c_uint16
        _lib.cato_obs_distance_f16.argtypes = [C.c_double] * 4
        _lib.cato_segment_query.restype = C.c_int
        _lib.cato_set_wall_subset.restype = None
        _lib.cato_set_wall_subset.argtypes = [C.c_void_p]
        _lib.cato_segment_query.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_int, C.c_void_p, C.c_void_p]
        _lib.cato_point_query_any.restype = C.c_int
        _lib.cato_point_query_any.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        _lib.cato_random_actions.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        _lib.cato_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cato_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cato_get_state.argtypes = [C.c_void_p, C.c_void_p]
        _lib.cato_set_state.argtypes = [C.c_void_p, C.c_void_p]
        _lib.cato_destroy.argtypes = [C.c_void_p]
        _lib.cato_set_threads.argtypes = [C.c_int]
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def alloc_outputs(N: int, A: int, R: int) -> dict:
    return dict(
        obs_distance=np.zeros((N, A, R), np.uint16), obs_type=np.zeros((N, A, R), np.uint8),
        hit_shape=np.zeros((N, A, R), np.int32), shared_distance=np.zeros((N, 2, R), np.uint16),
        shared_type=np.zeros((N, 2, R), np.uint8), team_positions=np.zeros((N, A, 2), np.uint16),
        reward=np.zeros((N, A), np.float32), terminated=np.zeros(N, np.uint8),
        truncated=np.zeros(N, np.uint8), winner=np.zeros(N, np.int8))


def alloc_state(N: int, A: int) -> dict:
    NP = A * (A - 1) // 2
    return dict(
        pos=np.zeros((N, A, 2)), vel=np.zeros((N, A, 2)), vbias=np.zeros((N, A, 2)), tc=np.zeros((N, A, 2)),
        leaf_bb=np.zeros((N, A, 4)), wall_shape=np.zeros((N, A, K_WALL), np.int32),
        wall_age=np.zeros((N, A, K_WALL), np.int32), wall_jn=np.zeros((N, A, K_WALL)),
        pair_age=np.zeros((N, max(NP, 1)), np.int32)[:, :NP], pair_jn=np.zeros((N, max(NP, 1)))[:, :NP],
        step_count=np.zeros(N, np.int32), reset_count=np.zeros(N, np.int32))


class OracleSim:
    """N independent envs advanced by the scalar CPU restatement."""

    def __init__(self, cfg, maps: Sequence, slot_map_ids: Optional[np.ndarray] = None):
        """``cfg``: any record with the workload fields (a product SimConfig is read field by field; ``bias_coef`` is recomputed
        here).  ``maps``: product CompiledMaps are used only for their ``spec`` (the inputs): the geometry blob, the ray table and
        the reward tables are built by cat_oracle_host.py, not taken from the product."""
        L = lib()
        v = host.config_values(cfg)
        self.cfg, self.N, self.A, self.R = cfg, v["n_envs"], v["n_cops"] + v["n_thieves"], v["n_rays"]
        c = _Config()
        for n in C_FIELDS_I32 + C_FIELDS_F64:
            setattr(c, n, v[n])
        c.env_id_offset, c.seed = v["env_id_offset"], v["seed"]
        dx, dy = host.ray_table(v["n_rays"], v["ray_length"])
        self._keep = [dx, dy, *host.reward_tables()]
        t = _Tables(*[_ptr(a) for a in self._keep])
        blobs = [m if isinstance(m, (bytes, bytearray)) else host.blob_for(m) for m in maps]
        arr = (C.c_char_p * len(blobs))(*blobs)
        sizes = (C.c_size_t * len(blobs))(*[len(b) for b in blobs])
        ids = None if slot_map_ids is None else np.ascontiguousarray(slot_map_ids, np.int32)
        h = C.c_void_p()
        rc = L.cato_create(C.byref(c), C.byref(t), arr, sizes, len(blobs), _ptr(ids), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"cato_create failed ({rc}): {L.cato_last_error().decode()}")
        self._h = h
        self.out = alloc_outputs(self.N, self.A, self.R)
        self._out_struct = _Outputs(*[_ptr(self.out[n]) for n in _OUT_FIELDS])

    def __del__(self):
        if getattr(self, "_h", None):
            lib().cato_destroy(self._h)
            self._h = None

    def reset(self, mask: Optional[np.ndarray] = None, positions: Optional[np.ndarray] = None) -> dict:
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p = None if positions is None else np.ascontiguousarray(positions, np.float64)
        lib().cato_reset(self._h, _ptr(m), _ptr(p), C.byref(self._out_struct))
        return self.out

    def step(self, actions: np.ndarray) -> dict:
        a = np.ascontiguousarray(actions, np.int32)
        assert a.shape == (self.N, self.A)
        lib().cato_step(self._h, _ptr(a), C.byref(self._out_struct))
        return self.out

    def random_actions(self, tick: int) -> np.ndarray:
        a = np.zeros((self.N, self.A), np.int32)
        lib().cato_random_actions(self._h, tick, _ptr(a))
        return a

    def get_state(self) -> dict:
        st = alloc_state(self.N, self.A)
        st = {k: np.ascontiguousarray(v) for k, v in st.items()}
        lib().cato_get_state(self._h, C.byref(_State(*[_ptr(st[n]) for n in _STATE_FIELDS])))
        return st

    def set_state(self, **arrays) -> None:
        keep = {k: np.ascontiguousarray(v, dtype=alloc_state(1, self.A)[k].dtype) for k, v in arrays.items()}
        lib().cato_set_state(self._h, C.byref(_State(*[_ptr(keep.get(n)) for n in _STATE_FIELDS])))

    def segment_query(self, env, self_agent, a, b, r2, los=False, walls=None):
        """``walls``: visit only these wall ids (diagnostic: the list a candidate table holds for the ray); None = every wall."""
        alpha = C.c_double()
        pt = (C.c_double * 2)()
        mask = None
        if walls is not None:
            mask = np.zeros(1024, np.uint8)        # one byte per wall id (maps hold at most 256 shapes)
            mask[list(walls)] = 1
            lib().cato_set_wall_subset(mask.ctypes.data_as(C.c_void_p))
        try:
            sh = lib().cato_segment_query(self._h, env, self_agent, a[0], a[1], b[0], b[1], r2, int(los),
                                          C.byref(alpha), pt)
        finally:
            if mask is not None:
                lib().cato_set_wall_subset(None)
        return sh, alpha.value, (pt[0], pt[1])

    def point_query_any(self, env, self_agent, p, maxd):
        return bool(lib().cato_point_query_any(self._h, env, self_agent, p[0], p[1], maxd))
