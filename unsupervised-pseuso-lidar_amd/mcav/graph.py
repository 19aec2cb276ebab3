"""hipGraph capture of the training step.

One step issues a few hundred kernel launches; replayed as one hipGraph they cost one host call, which is what the small, host-issue-bound
configurations need (BASELINE.json configs[0]-sized batches; configs[4] asks for one captured graph per resolution).  Captured: zero_grad,
the weight re-packing kernels (the packed copies are invalidated right before capture so that their launches are part of the graph and
re-run on every replay), both depth passes, the pose net, the fused loss (forward and its gradients), the whole backward and -- on one
rank -- the fused Adam update (reference trainer.py:261-266 as ONE graph: its step count, learning rate and gradient scale are read from a
device record, mcav_adam_step_dev).  With more than one rank the collectives and Adam stay outside the graph, and replay and the bucketed
overlap of mcav/dist.py work TOGETHER (round 4; BASELINE.json configs[4]: "hipGraph-captured step + overlapped all-reduce"): where the eager
backward announces a gradient bucket as final (mcav.nn.grads_ready: the decoder, then encoder layer4), the capture leaves an EXTERNAL
event-record node (mcav_event_record_external: hipEventRecordExternal on the capturing stream).  A replay signals those events as it gets
there; right after graph.replay() the host queues, per bucket, "communication stream waits for the bucket's event, then all-reduce"
(GradSync.ready_range), so each bucket's collective runs while the rest of the captured backward is still executing; GradSync.finish()
reduces what no bucket covered, waits for all of them, Adam follows eagerly.  No collective is issued from inside warm-up or capture.
Inputs live in static buffers; a replay is only valid for the batch shape it was captured with: StepGraphs keeps one graph per shape.
"""
import ctypes
import torch


class GraphedStep:
    def __init__(self, fwd_bwd, opt, example_inputs, capture_adam=True, warmup=2, buffers=()):
        """fwd_bwd(*inputs) -> tuple of tensors (e.g. the two losses); must zero the gradients itself.  opt: FusedAdam.
        Warm-up and capture really run the step: everything they change (parameters, moments, step count, `buffers` such as the BatchNorm
        running statistics) is put back afterwards, so constructing the graph leaves the training state untouched."""
        from . import nn as N
        from . import streams
        self.opt, self.arena = opt, opt.arena()
        self.capture_adam = bool(capture_adam)
        self.static_in = [x.clone() for x in example_inputs]
        arena = self.arena
        keep = [arena.flat.clone(), opt._m.clone(), opt._v.clone()] + [b.clone() for b in buffers]
        step0 = opt._step
        hook, N.GRADS_READY = N.GRADS_READY, None          # no collectives from inside warm-up / capture (see the module docstring)
        # buckets under replay: the GradSync this arena registered (mcav.dist.enable_overlap), if any; events are created BEFORE the capture
        from . import dist as mdist
        from . import lib as L
        self.sync = mdist._SYNC.get(id(arena)) if (mdist.parallel() and not self.capture_adam) else None
        self.marks = []                                    # (arena range, external event) in the order the captured backward reaches them
        self._events = []
        if self.sync is not None:
            for _ in range(8):
                ev = L.c_p()
                L.check(L.lib().mcav_event_create(ctypes.byref(ev)), "mcav_event_create")
                self._events.append(ev)

        def mark(params):
            # called from inside the captured backward (mcav.nn.grads_ready, after the bucket's slab reductions have been issued)
            r = self.sync.span(list(params))
            if r is None or len(self.marks) >= len(self._events) or any(not (r[1] <= lo or hi <= r[0]) for (lo, hi), _ in self.marks):
                return
            ev = self._events[len(self.marks)]
            L.check(L.lib().mcav_event_record_external(ev, L.stream()), "mcav_event_record_external")
            self.marks.append((r, ev))
        # This ROCm replays the captured branches on one queue (rounds 1-2), so the step is captured on ONE stream: same launches, no
        # cross-stream edges.  (Round 3 tried to keep the three streams in the capture and force the runtime's parallel graph queues,
        # DEBUG_HIP_FORCE_GRAPH_QUEUES=4: the process died inside the capture without a Python error; not pursued.)
        serial0, streams.SERIAL = streams.SERIAL, True
        try:
            def whole():
                out = fwd_bwd(*self.static_in)
                if self.capture_adam:
                    opt.step_capturable()
                return out
            # warm-up announces the buckets to a no-op: the batched slab reductions are then cut at the same points as under capture, so their
            # device tables exist before it (building one is a host -> device copy, which a capturing stream refuses)
            if self.sync is not None:
                N.GRADS_READY = lambda params: None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    arena.bump()
                    whole()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            arena.bump()                                  # every packed weight copy is stale -> its pack kernel is captured
            opt.device_state()                            # host -> device scalars are up to date BEFORE capture (no copy inside it)
            self.graph = torch.cuda.CUDAGraph()
            if self.sync is not None:
                N.GRADS_READY = mark
            with torch.cuda.graph(self.graph):
                out = whole()
            N.GRADS_READY = None
            self.static_out = tuple(o.detach() for o in out)
        finally:
            N.GRADS_READY = hook
            streams.SERIAL = serial0
        with torch.no_grad():                              # undo what warm-up and capture did to the training state
            arena.flat.copy_(keep[0]); opt._m.copy_(keep[1]); opt._v.copy_(keep[2])
            for b, k in zip(buffers, keep[3:]):
                b.copy_(k)
        opt._step = step0
        opt._dev_mirror = None
        arena.bump()

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        if self.capture_adam:
            self.opt.device_state()                       # lr / grad_scale / step count as the host has them now
        self.graph.replay()
        for r, ev in self.marks:                          # each bucket's collective waits for ITS point of the replaying graph
            self.sync.ready_range(r, ev)
        if self.capture_adam:
            self.opt.note_replayed()
        else:
            self.arena.bump()                             # (the eager Adam that follows bumps again; harmless)
        return self.static_out


class StepGraphs:
    """One captured step per input shape (BASELINE.json configs[4]: batches of two resolutions alternate through one process)."""

    def __init__(self, fwd_bwd, opt, capture_adam=True, buffers=()):
        self.fwd_bwd, self.opt, self.capture_adam, self.buffers = fwd_bwd, opt, capture_adam, tuple(buffers)
        self.graphs = {}

    def __call__(self, *inputs):
        key = tuple(tuple(x.shape) for x in inputs)
        g = self.graphs.get(key)
        if g is None:
            g = self.graphs[key] = GraphedStep(self.fwd_bwd, self.opt, inputs, self.capture_adam, buffers=self.buffers)
        return g(*inputs)


GraphedForwardBackward = GraphedStep          # round-1 name
