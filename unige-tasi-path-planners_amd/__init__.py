"""MI355X-native cost-propagation engine for the grid replanners
(Field D*, Shifted-Grid FM, Multi-Stencil DFM) -- Python host mirror.

The product is the C-ABI library ``libufm.so`` (include/ufm.h); this package
only binds it with ctypes (plain pointers, no torch types) and mirrors the
reference planner surface (ReplannerBase.h:39-123) for tests and bench.py.
There is no CPU fallback: importing works anywhere, but creating a planner
without the HIP library / a GPU raises.
"""
from .capi import (  # noqa: F401
    ALGO_FD, ALGO_SG, ALGO_DFM, LOOP_OK, LOOP_FAILURE_NO_GRAPH, LOOP_FAILURE_NO_GOAL,
    UfmError, Planner, BatchPlanner, Stats, PathInfo, load_library, library_path, build_library, use_library,
)
from . import capi  # noqa: F401
from . import synth  # noqa: F401
from . import episode  # noqa: F401
from . import harness  # noqa: F401
from . import tolerances  # noqa: F401
