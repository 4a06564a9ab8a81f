"""Mirror of the reference package network/libs/post_process (CSPN_new, CSPN_ours)."""
from . import CSPN_new, CSPN_ours
from .CSPN_new import AffinityPropagate

__all__ = ["CSPN_new", "CSPN_ours", "AffinityPropagate"]
