"""`CatSim`: thin owner of one libcat_sim handle on one GPU, with torch tensors as the caller-owned
device buffers.  torch is plumbing here (device memory, streams); every computation of the env
tick happens inside the HIP kernels behind the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _native as nat
from . import tables
from .config import C_FIELDS_F64, C_FIELDS_I32, SimConfig
from .maps import CompiledMap

_OUT_SPEC = {  # name -> (shape fn, torch dtype)
    "obs_distance": (lambda N, A, R: (N, A, R), torch.float16),
    "obs_type": (lambda N, A, R: (N, A, R), torch.uint8),
    "hit_shape": (lambda N, A, R: (N, A, R), torch.int32),
    "shared_distance": (lambda N, A, R: (N, 2, R), torch.float16),
    "shared_type": (lambda N, A, R: (N, 2, R), torch.uint8),
    "team_positions": (lambda N, A, R: (N, A, 2), torch.float16),
    "reward": (lambda N, A, R: (N, A), torch.float32),
    "terminated": (lambda N, A, R: (N,), torch.uint8),
    "truncated": (lambda N, A, R: (N,), torch.uint8),
    "winner": (lambda N, A, R: (N,), torch.int8),
}


def _state_spec(N: int, A: int):
    NP = A * (A - 1) // 2
    K = nat.WALL_CACHE
    f, i = torch.float64, torch.int32
    return {"pos": ((N, A, 2), f), "vel": ((N, A, 2), f), "vbias": ((N, A, 2), f), "tc": ((N, A, 2), f),
            "leaf_bb": ((N, A, 4), f), "wall_shape": ((N, A, K), i), "wall_age": ((N, A, K), i),
            "wall_jn": ((N, A, K), f), "pair_age": ((N, NP), i), "pair_jn": ((N, NP), f),
            "step_count": ((N,), i), "reset_count": ((N,), i)}


class CatSimError(RuntimeError):
    pass


class CatSim:
    def __init__(self, cfg: SimConfig, maps: Sequence[CompiledMap], slot_map_ids=None,
                 device: Optional[torch.device | int | str] = None, debug_hit_shape: bool = False):
        self._h = None
        self._L = nat.lib()  # raises NativeLibraryMissing when the extension is not built
        if not torch.cuda.is_available():
            raise CatSimError("no HIP device visible to torch: libcat_sim has no CPU path")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise CatSimError(f"device must be a GPU, got {dev}")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.cfg, self.N, self.A, self.R = cfg, cfg.n_envs, cfg.n_agents, cfg.n_rays
        self.maps = list(maps)
        c = nat.CatConfig()
        for n in C_FIELDS_I32 + C_FIELDS_F64:
            setattr(c, n, getattr(cfg, n))
        c.env_id_offset, c.seed = cfg.env_id_offset, cfg.seed
        dx, dy = tables.ray_table(cfg.sensor)
        host = [dx, dy, tables.cop_reward_lut(), tables.thief_reward_lut()]
        t = nat.CatTables(*[a.ctypes.data_as(C.c_void_p) for a in host])
        blobs = [m.to_blob() for m in self.maps]
        arr = (C.c_char_p * len(blobs))(*blobs)
        sizes = (C.c_size_t * len(blobs))(*[len(b) for b in blobs])
        ids = None if slot_map_ids is None else np.ascontiguousarray(slot_map_ids, np.int32)
        h = C.c_void_p()
        rc = self._L.cat_create(C.byref(c), C.byref(t), arr, sizes, len(blobs),
                                None if ids is None else ids.ctypes.data_as(C.c_void_p), dev.index, C.byref(h))
        if rc != 0:
            raise CatSimError(f"cat_create: {nat.ERRORS.get(rc, rc)}: {self._L.cat_last_error(None).decode()}")
        self._h = h
        with torch.cuda.device(dev):
            self.out: Dict[str, torch.Tensor] = {
                k: torch.zeros(fn(self.N, self.A, self.R), dtype=dt, device=dev)
                for k, (fn, dt) in _OUT_SPEC.items() if (k != "hit_shape" or debug_hit_shape)}
        self._out_struct = nat.CatOutputs(*[self.out[k].data_ptr() if k in self.out else None
                                            for k in nat.OUT_FIELDS])

    # ------------------------------------------------------------------------------------
    def close(self) -> None:
        if self._h is not None:
            self._L.cat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise CatSimError(f"{what}: {nat.ERRORS.get(rc, rc)}: {self._L.cat_last_error(self._h).decode()}")

    @property
    def one_tick_kernel(self) -> str:
        """The kernel ``step`` / ``step_fused`` launch ("step_kernel"; diagnostic builds of earlier sources: "tick_kernel")."""
        return self._L.cat_one_tick_kernel(self._h).decode() if hasattr(self._L, "cat_one_tick_kernel") else "tick_kernel"

    @property
    def rollout_kernel(self) -> str:
        """The kernel ``rollout_fused`` launches: "rollout_kernel", or "rollout_kernel_pooled" where cat_create chose the pooled ray fan."""
        if hasattr(self._L, "cat_rollout_kernel"):
            return self._L.cat_rollout_kernel(self._h).decode()
        return "rollout_kernel"   # diagnostic builds of earlier sources

    def chunks_per_unit(self, resident: bool) -> int:
        """Chunk form of the ray fan: 64-ray chunks of a slot that one work unit traces with one shared item list (1 = chunk by chunk)."""
        return int(self._L.cat_chunks_per_unit(self._h, int(resident))) if hasattr(self._L, "cat_chunks_per_unit") else 1

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset(self, mask: Optional[torch.Tensor] = None, positions: Optional[torch.Tensor] = None):
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            assert mask.shape == (self.N,)
        if positions is not None:
            positions = positions.to(device=self.device, dtype=torch.float64).contiguous()
            assert positions.shape == (self.N, self.A, 2)
        self._check(self._L.cat_reset(self._h, None if mask is None else mask.data_ptr(),
                                      None if positions is None else positions.data_ptr(),
                                      C.byref(self._out_struct), self._stream()), "cat_reset")
        return self.out

    def reset_done(self):
        self._check(self._L.cat_reset_done(self._h, C.byref(self._out_struct), self._stream()), "cat_reset_done")
        return self.out

    def step(self, actions: torch.Tensor):
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        if actions.shape != (self.N, self.A):
            raise ValueError(f"actions must have shape {(self.N, self.A)}, got {tuple(actions.shape)}")
        self._check(self._L.cat_step(self._h, actions.data_ptr(), C.byref(self._out_struct), self._stream()), "cat_step")
        return self.out

    def step_fused(self, actions: Optional[torch.Tensor] = None, tick: int = 0, auto_reset: bool = True):
        """One launch: (synthetic Philox actions when ``actions`` is None) + step + auto-reset."""
        ptr = None
        if actions is not None:
            if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
                actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
            if actions.shape != (self.N, self.A):
                raise ValueError(f"actions must have shape {(self.N, self.A)}, got {tuple(actions.shape)}")
            ptr = actions.data_ptr()
        self._check(self._L.cat_step_fused(self._h, ptr, int(tick), int(auto_reset), C.byref(self._out_struct),
                                           self._stream()), "cat_step_fused")
        return self.out

    def rollout_buffers(self, T: int) -> Dict[str, torch.Tensor]:
        """Caller-owned output buffers with a leading T for ``rollout_fused`` (cached per T)."""
        cache = self.__dict__.setdefault("_rollout_bufs", {})
        if T not in cache:
            with torch.cuda.device(self.device):
                cache[T] = {k: torch.zeros((T,) + tuple(fn(self.N, self.A, self.R)), dtype=dt, device=self.device)
                            for k, (fn, dt) in _OUT_SPEC.items() if k in self.out}
        return cache[T]

    def rollout_fused(self, T: int, actions: Optional[torch.Tensor] = None, tick: int = 0, auto_reset: bool = True,
                      out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """ONE launch for T consecutive ticks (cat_rollout_fused): the map stays in LDS, every env's state record stays in LDS
        for the T ticks, and row t of every ``[T, N, ...]`` buffer holds what ``step_fused(actions[t], tick + t, auto_reset)``
        leaves in ``self.out``.  ``actions``: ``[T, N, A]`` int32 or None (synthetic Philox actions of ticks tick .. tick+T-1)."""
        T = int(T)
        bufs = self.rollout_buffers(T) if out is None else out
        for k, v in bufs.items():
            fn, dt = _OUT_SPEC[k]
            if tuple(v.shape) != (T,) + tuple(fn(self.N, self.A, self.R)) or v.dtype != dt or not v.is_contiguous() or v.device != self.device:
                raise ValueError(f"rollout buffer {k!r} must be a contiguous {dt} tensor of shape {(T,) + tuple(fn(self.N, self.A, self.R))} on {self.device}")
        ptr = None
        if actions is not None:
            if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
                actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
            if actions.shape != (T, self.N, self.A):
                raise ValueError(f"actions must have shape {(T, self.N, self.A)}, got {tuple(actions.shape)}")
            ptr = actions.data_ptr()
        view = nat.CatOutputs(*[bufs[k].data_ptr() if k in bufs else None for k in nat.OUT_FIELDS])
        self._check(self._L.cat_rollout_fused(self._h, T, ptr, int(tick), int(auto_reset), C.byref(view), self._stream()),
                    "cat_rollout_fused")
        return bufs

    def random_actions(self, tick: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((self.N, self.A), dtype=torch.int32, device=self.device)
        self._check(self._L.cat_random_actions(self._h, int(tick), out.data_ptr(), self._stream()), "cat_random_actions")
        return out

    def arm_kernel_timing(self, start_event: int, stop_event: int) -> None:
        """Attach two raw ``hipEvent_t`` handles to the NEXT step's tick-kernel dispatch (its own begin/end)."""
        self._check(self._L.cat_arm_kernel_timing(self._h, start_event, stop_event), "cat_arm_kernel_timing")

    def device_errors(self, clear: bool = True) -> int:
        """Device-side error flags raised by the kernels since the last clear (``_native.DEVERR_*``); synchronises."""
        flags = C.c_uint32(0)
        self._check(self._L.cat_device_errors(self._h, C.byref(flags), int(clear), self._stream()), "cat_device_errors")
        return int(flags.value)

    def check_errors(self) -> None:
        """Raise for what the asynchronous launches could not report: an action outside 0..3 (the reference raises on
        it) or a dropped wall contact."""
        flags = self.device_errors()
        if flags & nat.DEVERR_BAD_ACTION:
            raise ValueError("an action outside 0..3 was passed to the step kernel (it was applied as 'no impulse')")
        if flags & nat.DEVERR_CONTACT_DROPPED:
            raise CatSimError("a wall contact was dropped: an agent touched more than WALL_CACHE walls in one step")
        if flags & nat.DEVERR_SCHEDULER:
            raise CatSimError("internal: a work item of the pooled ray fan never arrived; the launch left instead of hanging and its results are invalid")

    def set_seed(self, seed: int) -> None:
        self.cfg.seed = int(seed) & (2**64 - 1)
        self._check(self._L.cat_set_seed(self._h, self.cfg.seed, self._stream()), "cat_set_seed")

    def get_state(self) -> Dict[str, torch.Tensor]:
        st = {k: torch.zeros(shape, dtype=dt, device=self.device) for k, (shape, dt) in _state_spec(self.N, self.A).items()}
        view = nat.CatState(*[st[k].data_ptr() if st[k].numel() else None for k in nat.STATE_FIELDS])
        self._check(self._L.cat_get_state(self._h, C.byref(view), self._stream()), "cat_get_state")
        return st

    def set_state(self, **arrays) -> None:
        spec = _state_spec(self.N, self.A)
        keep = {}
        for k, v in arrays.items():
            shape, dt = spec[k]
            t = torch.as_tensor(v).to(device=self.device, dtype=dt).contiguous()
            assert tuple(t.shape) == tuple(shape), (k, t.shape, shape)
            keep[k] = t
        view = nat.CatState(*[keep[k].data_ptr() if k in keep and keep[k].numel() else None for k in nat.STATE_FIELDS])
        self._check(self._L.cat_set_state(self._h, C.byref(view), self._stream()), "cat_set_state")
        torch.cuda.current_stream(self.device).synchronize()  # sources may be temporaries
