"""cyclistsocialforce.scenario -> cyclistsocialforce_amd.scenario (see the package docstring)"""
from cyclistsocialforce_amd.scenario import *  # noqa: F401,F403
from cyclistsocialforce_amd import scenario as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
