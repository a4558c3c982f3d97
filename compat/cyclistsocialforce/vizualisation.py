"""cyclistsocialforce.vizualisation -> cyclistsocialforce_amd.vizualisation (see the package docstring)"""
from cyclistsocialforce_amd.vizualisation import *  # noqa: F401,F403
from cyclistsocialforce_amd import vizualisation as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
