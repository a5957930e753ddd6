"""graspqp_amd -- MI355X-native grasp-optimisation inner loop behind GraspQP's plugin surface.

The compute path is hand-written HIP for gfx950 behind a C-ABI shared library
(``include/graspqp_hip.h`` -> ``graspqp_amd/lib/libgraspqp_hip.so``); there is no CPU fallback:
every op raises if the library is missing.  Pure-data helpers (``hands``, ``utils``) import without it.
"""

__version__ = "0.1.0"
