"""cyclistsocialforce.intersection -> cyclistsocialforce_amd.intersection (see the package docstring)"""
from cyclistsocialforce_amd.intersection import *  # noqa: F401,F403
from cyclistsocialforce_amd import intersection as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
