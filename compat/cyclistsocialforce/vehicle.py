"""cyclistsocialforce.vehicle -> cyclistsocialforce_amd.vehicle (see the package docstring)"""
from cyclistsocialforce_amd.vehicle import *  # noqa: F401,F403
from cyclistsocialforce_amd import vehicle as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
