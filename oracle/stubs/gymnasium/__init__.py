"""Stand-in for the two `gymnasium` names the reference imports (TEST INFRASTRUCTURE ONLY).

`gymnasium` is not installed in the build container and there is no network.  The
reference touches only `gym.Env` (base class, `super().__init__()`) and
`gym.spaces.Box(low, high, shape=None, dtype=...)`
(/root/reference/fleetrl/fleet_env/fleet_environment.py:3,50,118,316-325).
This module exists solely so `oracle/gen_golden.py` can import the unmodified
reference; nothing in the product imports it.
"""
from . import spaces  # noqa: F401


class Env:
    metadata = {}
    render_mode = None

    def __init__(self, *a, **k):
        pass
