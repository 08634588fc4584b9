"""Navigation-only `MatterSim` (pybind11, C++17).  `import speaker_follower_amd.sim` puts the built
extension on sys.path so that the reference's `import MatterSim` (tasks/R2R/env.py:5) resolves."""
import os
import sys

_DIR = os.path.dirname(os.path.abspath(__file__))
if _DIR not in sys.path:
    sys.path.insert(0, _DIR)


def load():
    import MatterSim
    return MatterSim


def load_frontier():
    """The state-factored search's bookkeeping (frontier_core.cpp): sf_frontier.StateFactored."""
    import sf_frontier
    return sf_frontier


def load_sweep():
    """The batched panorama sweep (sweep_py.cpp): sf_sweep.sweep_scan."""
    import sf_sweep
    return sf_sweep
