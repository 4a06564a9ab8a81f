"""Drop-in for the reference's 3x3 CSPN module (network/libs/post_process/CSPN_new.py:17-128).

Same constructor and forward signature as the reference class, constructed as
``post_process.AffinityPropagate(24, 3)`` and called positionally
``self.post_process_layer(guidance, x, sparse_depth)`` (network/unet_cspn_nyu.py:357-358, :386).
No parameters, no buffers (empty state_dict) — checkpoints are unaffected by the swap.
The recurrence itself runs in libcspn_hip.so (hand-written gfx950 kernels); see functional.py.
"""
import torch.nn as nn

from ..functional import cspn3_affinity_propagate, cspn3_refine_and_score


class AffinityPropagate(nn.Module):

    def __init__(self, prop_time, prop_kernel, plan=None):
        super(AffinityPropagate, self).__init__()
        self.prop_time = prop_time
        self.prop_kernel = prop_kernel
        self.in_feature = 1      # attributes kept for parity with CSPN_new.py:23-24
        self.out_feature = 1
        self.plan = plan         # optional launch plan (functional.set_default_plan / cspn_plan fields)
        if prop_kernel != 3:
            # The reference module silently returns an (H-1)x(W-1) map for prop_kernel=5
            # (ones-kernel of shape 1x2x2, CSPN_new.py:122); the K x K configurations are served by
            # post_process.CSPN_ours.AffinityPropagate instead.
            raise ValueError("CSPN_new.AffinityPropagate implements the 3x3 propagation only; "
                             "use CSPN_ours.AffinityPropagate for K x K affinity kernels")

    def forward(self, guidance, blur_depth, sparse_depth=None):
        """guidance [B,C>=8,H,W] (channels 0..7 used), blur_depth [B,1,H,W], sparse_depth [B,1,H,W] | None
        -> refined depth [B,1,H,W]."""
        return cspn3_affinity_propagate(guidance, blur_depth, sparse_depth, self.prop_time, self.plan)

    def forward_scored(self, guidance, blur_depth, sparse_depth, target, acc):
        """Extension (not in the reference): inference forward whose last propagation launch also accumulates the
        depth metrics of the result vs `target` into `acc` (cspn_monodepth_amd.evaluation.new_accumulator)."""
        return cspn3_refine_and_score(guidance, blur_depth, sparse_depth, target, acc, self.prop_time, self.plan)
