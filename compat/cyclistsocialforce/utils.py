"""cyclistsocialforce.utils -> cyclistsocialforce_amd.utils (see the package docstring)"""
from cyclistsocialforce_amd.utils import *  # noqa: F401,F403
from cyclistsocialforce_amd import utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
