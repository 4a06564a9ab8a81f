"""HIP-graph capture of an inference forward (torch.cuda.CUDAGraph is hipGraph on ROCm).

The engine only enqueues kernels on the caller's stream — no allocation, no synchronisation inside the `.so` —
so a whole `AffinityPropagate` forward (and its scored variant) can be captured once and replayed.  That removes
the per-call host work (Python, allocator, three launches), which is what bounds small per-GPU batches
(BASELINE config 4: one 1216x352 image per GPU).  Inputs are copied into static buffers before each replay.
"""
import torch

from . import functional


class GraphedForward(object):
    """graphed = GraphedForward(fn, *example_tensors);  out = graphed(*tensors)   (same shapes / dtypes).

    `fn` is any no-grad callable built on this package (e.g. a module's forward or forward_scored bound to an
    accumulator).  The returned tensor is a static buffer overwritten by the next call."""

    def __init__(self, fn, *example, warmup=3):
        self.static_in = [None if t is None else t.clone() for t in example]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *tensors, copy_inputs=True):
        # a replayed weight-resident launch that timed out must not go unnoticed: the previous replay's error word is
        # looked at before the next one (free: a pinned host word); `synchronize()` below checks the last one
        functional.check_resident_errors()
        if copy_inputs:
            for dst, src in zip(self.static_in, tensors):
                if dst is not None and src is not None and dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        self.graph.replay()
        functional.mark_resident_pending(self.static_out)
        return self.static_out

    def synchronize(self):
        """Wait for the last replay and raise if a weight-resident launch inside it timed out; returns the output."""
        torch.cuda.current_stream().synchronize()
        functional.check_resident_errors()
        return self.static_out
