"""fleetrl_amd -- MI355X-native batched FleetRL `FleetEnv.step()` hot path."""
__version__ = "0.1.0"
