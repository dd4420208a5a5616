"""fleetrl_amd -- MI355X-native batched FleetRL `FleetEnv.step()` hot path.

    from fleetrl_amd import FleetEnv, FleetVecEnv, FleetVectorEnv, FleetMixedVecEnv
"""
__version__ = "0.1.0"


def __getattr__(name):  # lazy: importing the package must not require the HIP library (build() imports it first)
    if name in ("FleetEnv", "FleetVecEnv", "FleetVectorEnv", "FleetCore"):
        from . import vec_env

        return getattr(vec_env, name)
    if name == "FleetMixedVecEnv":
        from . import mixed

        return mixed.FleetMixedVecEnv
    if name in ("FleetBatch", "FleetHipError"):
        from . import batch

        return getattr(batch, name)
    raise AttributeError(name)
