"""Drop-in for the reference's K x K (pixel-adaptive) CSPN module
(network/libs/post_process/CSPN_ours.py:18-54, which loops network/libs/base/pac.py:124-144).

Argument order is (x, guided) — the opposite of CSPN_new — exactly as in the reference
(CSPN_ours.py:24; call site network/unet_ours.py:333 uses the keyword ``sparse_depth=``).
K is inferred from the channel count (CSPN_ours.py:31-32).
"""
import torch
import torch.nn as nn

from ..functional import pac_affinity_propagate, pac_refine_and_score


class AffinityPropagate(nn.Module):

    def __init__(self, prop_time, plan=None, state_dtype="reference"):
        super(AffinityPropagate, self).__init__()
        self.times = prop_time
        self.plan = plan
        # "reference": half inputs keep an fp32 depth state and return fp32, which is what the reference
        # computes (its kernel tensor is created in fp32, CSPN_ours.py:37, so everything after the fp16
        # softmax is promoted).  None: keep the input dtype for the state (fp16 storage, 52 B/px/step).
        self.state_dtype = state_dtype

    def forward(self, x, guided, sparse_depth=None):
        """x [B,1,H,W], guided [B,K*K-1,H,W], sparse_depth [B,1,H,W] | None -> [B,1,H,W]."""
        sdt = self.state_dtype
        if sdt == "reference":
            sdt = torch.float32 if x.dtype == torch.float16 else None
        return pac_affinity_propagate(x, guided, sparse_depth, self.times, self.plan, sdt)

    def forward_scored(self, x, guided, sparse_depth, target, acc):
        """Extension (not in the reference): see CSPN_new.AffinityPropagate.forward_scored."""
        sdt = self.state_dtype
        if sdt == "reference":
            sdt = torch.float32 if x.dtype == torch.float16 else None
        return pac_refine_and_score(x, guided, sparse_depth, target, acc, self.times, self.plan, sdt)
