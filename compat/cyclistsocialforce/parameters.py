"""cyclistsocialforce.parameters -> cyclistsocialforce_amd.parameters (see the package docstring)"""
from cyclistsocialforce_amd.parameters import *  # noqa: F401,F403
from cyclistsocialforce_amd import parameters as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
