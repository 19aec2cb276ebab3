"""Fork / join of an independent sub-graph onto a second HIP stream.

The pose network does not depend on the depth network until the loss (reference trainer.py:296-311): its seven small,
latency-bound conv layers run beside the depth network's large kernels instead of between them.  autograd replays each
backward node on the stream its forward ran on, so the backward pass inherits the same overlap without further code.
"""
import torch


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)


# Set by mcav/graph.py: a captured hipGraph is replayed on ONE queue by this ROCm (its branches do not overlap), so under
# capture everything is issued on the capturing stream -- same launches, no cross-stream edges in the graph.
SERIAL = False

# Set by mcav/graph.py while a step is captured as main-chain and side-chain graphs (class _DualCapture): the side stream captures graphs of
# its own.  A Branch then runs the pose network's launches on that side stream (models/pose/pose_net.py switches streams INSIDE its autograd
# node, so that autograd sees one stream and adds no cross-stream event of its own, which would tie the two captures together), and
# Branch.join() is where the pair of graphs is cut: what follows on the main stream needs the branch's result, and a graph can only wait for
# another graph's node through the host (see _DualCapture.join).
DUAL = None


class Branch:
    def __init__(self):
        self.stream = None
        self.inline = False

    def fork(self, fn, *args):
        """Run fn(*args) on the branch stream, ordered after everything already queued on the current stream."""
        self.inline = SERIAL or DUAL is not None or not any(t.is_cuda for t in _tensors(args))
        self.cut = DUAL is not None
        if self.inline:
            return fn(*args)
        cur = torch.cuda.current_stream()
        if self.stream is None or self.stream.device != cur.device:
            self.stream = torch.cuda.Stream(device=cur.device)
        self.stream.wait_stream(cur)
        for t in _tensors(args):
            if t.is_cuda:
                t.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            return fn(*args)

    def join(self, out):
        """Make the current stream wait for the branch; `out` (tensors made on the branch) becomes safe to use on it."""
        if self.inline:
            if getattr(self, "cut", False) and DUAL is not None:
                DUAL.split()                       # two-graph capture: the main chain continues in the next graph, behind the side chain's
            return out
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.stream)
        for t in _tensors(out):
            if t.is_cuda:
                t.record_stream(cur)
        return out
