"""The reference's package name on the MI355X engine.

Put this directory in front of (or instead of) the reference's `src/` on PYTHONPATH and scripts written against
`cyclistsocialforce` - `from cyclistsocialforce.vehicle import InvPendulumBicycle`, `from cyclistsocialforce.intersection
import SocialForceIntersection`, `from cyclistsocialforce.scenario import Scenario` (demoCSFstandalone.py:23-25) - run on the
HIP engine unmodified:

    PYTHONPATH=/path/to/this/repo/compat:/path/to/this/repo python demo/demoCSFstandalone.py -m invpendulum

Every module here only re-exports its namesake of `cyclistsocialforce_amd`; there is no code of its own and no CPU
fallback (no HIP device: the engine raises)."""
