"""Top-level alias so U-NO model code that does ``from integral_operators import *``
(reference darcy_flow_uno2d.py:10, navier_stokes_uno2d.py:9, navier_stokes_uno3d.py:6) picks up
the MI355X-native operator blocks unchanged.  The implementation lives in uno_amd/."""
from uno_amd.integral_operators import *  # noqa: F401,F403
from uno_amd.integral_operators import __all__  # noqa: F401
