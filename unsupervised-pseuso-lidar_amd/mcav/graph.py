"""hipGraph capture of the forward + backward part of the training step.

One step issues ~450 kernel launches; replayed as one hipGraph they cost one host call.  Captured: zero_grad, both depth
passes, the pose net, the fused loss (forward and its gradients) and the whole backward, including the weight re-packing
kernels (the packed copies are invalidated right before capture so their launches are part of the graph and therefore
re-run on every replay, after each optimiser update).  NOT captured: the gradient all-reduce and the Adam launch (host
scalars: step count, learning rate), which run eagerly after the replay.
Inputs live in static buffers; a replay is only valid for the batch shape it was captured with (one graph per shape).
"""
import torch


class GraphedForwardBackward:
    def __init__(self, fwd_bwd, arena, example_inputs, warmup=3):
        """fwd_bwd(*inputs) -> tuple of tensors (e.g. the two losses); must zero the gradients itself."""
        from . import streams
        streams.SERIAL = True                      # single-stream issue from here on (see streams.py)
        self.arena = arena
        self.static_in = [x.clone() for x in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                arena.bump()
                out = fwd_bwd(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        arena.bump()                              # every packed weight copy is stale -> its pack kernel is captured
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out = fwd_bwd(*self.static_in)
        self.static_out = tuple(o.detach() for o in out)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_out
