"""cspn_monodepth_amd — MI355X-native CSPN affinity-propagation engine.

Host side (Python, PyTorch-ROCm for memory/streams/autograd) over a C-ABI HIP library
(include/cspn_hip.h -> cspn_monodepth_amd/libcspn_hip.so).  Scope: the hot path of
dontLoveBugs/CSPN_monodepth only; see DESIGN.md.
"""
from . import _lib, base, evaluation, functional, graphs, network, post_process
from .functional import (cspn3_affinity_propagate, pac_affinity_propagate, set_default_plan)
from .post_process import CSPN_new, CSPN_ours

__all__ = ["_lib", "base", "evaluation", "functional", "graphs", "network", "post_process", "CSPN_new", "CSPN_ours",
           "cspn3_affinity_propagate", "pac_affinity_propagate", "set_default_plan"]
__version__ = "0.1.0"
