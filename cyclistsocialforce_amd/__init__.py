"""MI355X-native stepping engine for the cyclist social-force model.

Drop-in for the per-tick hot path of chris-konrad/cyclistsocialforce: the same `vehicle.*` constructors,
`.step()` and state arrays, `intersection.SocialForceIntersection` and `scenario.Scenario`, executed by
hand-written HIP kernels (gfx950) behind the C ABI of include/csf.h.  No CPU fallback.
"""
__version__ = "0.1.0"
