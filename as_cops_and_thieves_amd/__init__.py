"""MI355X-native batched Cops-and-Thieves env core (drop-in for the env hot path of
Hevagog/as-cops-and-thieves).  The compute lives in ``libcat_sim.so`` (HIP, gfx950); see DESIGN.md."""
from .constants import ObjectType, PhysicalParams, SensorParams, SpaceParams, load_physical_params  # noqa: F401
from .config import SimConfig  # noqa: F401
from .maps import CompiledMap, Map, load_preset, bundled_map_path  # noqa: F401

__all__ = ["ObjectType", "PhysicalParams", "SensorParams", "SpaceParams", "load_physical_params", "SimConfig",
           "CompiledMap", "Map", "load_preset", "bundled_map_path", "BaseEnv", "SimpleEnv", "VecCopsEnv", "raw_env", "CatSim"]


def __getattr__(name):  # torch-dependent parts are imported lazily
    if name in ("BaseEnv", "SimpleEnv", "VecCopsEnv", "raw_env"):
        from . import environments
        return getattr(environments, name)
    if name == "CatSim":
        from .sim import CatSim
        return CatSim
    raise AttributeError(name)
