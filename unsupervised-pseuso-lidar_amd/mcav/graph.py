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


DUAL_DEFAULT = True        # capture the weight-gradient chain as a second graph replayed on a second stream (see _DualCapture)


class _DualCapture:
    """Two chains of graphs captured at once.  This ROCm replays a graph's branches on ONE queue, which costs a replayed step the overlap the
    eager step gets from its side streams (batch 12: 12.8 ms replayed against 12.3 eager in round 3).  So the side stream captures graphs of
    its OWN -- the pose network's forward, then the weight-gradient chain and the pose network's backward -- replayed on that stream beside
    the main chain's, and the two order their work through external event nodes: at a fork the main graph records an event
    (mcav_event_record_external) that the side graph waits for (mcav_event_wait_external).  The other direction cannot be a node: a wait node
    refers to the record that is pending when ITS graph is launched, and of two graphs only one can be launched first.  So the main graph is
    launched first (the side graph's waits -- and a bucket's all-reduce outside both -- then see this replay's records), and where the main
    chain needs the side chain's result -- the poses before the loss, every weight gradient before Adam -- the chains are CUT (split(): the
    side graph's last node records an event, both captures end, the next pair begins in the same memory pools) and the host orders the main
    stream behind that event between the launches (GraphedStep.__call__): main[0], side[0], wait, main[1], side[1], wait, Adam."""

    def __init__(self, device, nevents=1024):
        from . import lib as L
        self.L = L
        self.stream = torch.cuda.Stream(device=device)            # side chain
        self.main_stream = torch.cuda.Stream(device=device)       # main chain
        self.events = []
        for _ in range(nevents):
            ev = L.c_p()
            L.check(L.lib().mcav_event_create(ctypes.byref(ev)), "mcav_event_create")
            self.events.append(ev)
        self.used = 0
        self.main_graphs, self.side_graphs, self.cut_events = [], [], []
        self.join_event = None

    def __del__(self):
        try:
            for ev in self.events:
                self.L.lib().mcav_event_destroy(ev)
        except Exception:
            pass

    def _next(self):
        if self.used >= len(self.events):      # (one event per weight-gradient launch of the step: ~50 for ResNet-18, ~120 for ResNet-50)
            raise self.L.MCAVError("two-graph capture: more than %d fork points in one step" % len(self.events))
        ev = self.events[self.used]
        self.used += 1
        return ev

    def _begin_pair(self):
        gm, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.stream):
            gs.capture_begin(pool=self.side_graphs[0].pool(), capture_error_mode="relaxed") if self.side_graphs else gs.capture_begin(capture_error_mode="relaxed")
        with torch.cuda.stream(self.main_stream):
            gm.capture_begin(pool=self.main_graphs[0].pool(), capture_error_mode="relaxed") if self.main_graphs else gm.capture_begin(capture_error_mode="relaxed")
        self.main_graphs.append(gm)
        self.side_graphs.append(gs)

    def _end_pair(self):
        with torch.cuda.stream(self.main_stream):
            self.main_graphs[-1].capture_end()
        with torch.cuda.stream(self.stream):
            self.side_graphs[-1].capture_end()

    def begin(self):
        self._begin_pair()

    def fork(self, main):
        L, ev = self.L, self._next()
        L.check(L.lib().mcav_event_record_external(ev, L.c_p(main.cuda_stream)), "mcav_event_record_external(fork)")
        L.check(L.lib().mcav_event_wait_external(ev, L.c_p(self.stream.cuda_stream)), "mcav_event_wait_external(fork)")

    def split(self):
        """The main chain needs what the side chain has produced so far: cut both (called on the main chain's thread, its stream current)."""
        L, ev = self.L, self._next()
        L.check(L.lib().mcav_event_record_external(ev, L.c_p(self.stream.cuda_stream)), "mcav_event_record_external(cut)")
        self.cut_events.append(ev)
        self._end_pair()
        self._begin_pair()

    def join(self, main):
        """End of the side chain: its last node records the join event; the host orders the launching stream behind it after both launches,
        and what follows -- the Adam launch -- stays outside the graphs.  (A wait NODE in the main graph was the first version: it saw the
        previous replay's record and Adam ran without the weight gradients.)"""
        L, ev = self.L, self._next()
        L.check(L.lib().mcav_event_record_external(ev, L.c_p(self.stream.cuda_stream)), "mcav_event_record_external(join)")
        self.join_event = ev

    def end(self, failed=False):
        self._end_pair()
        if failed:                                 # the step raised: the captures are closed, the caller re-raises
            return
        if self.join_event is None:                # a step without a backward pass has no join to order Adam behind
            raise self.L.MCAVError("two-graph capture: the step never joined its side chain (no backward pass?)")
        self.cut_events.append(self.join_event)

    def replay(self):
        L = self.L
        cur = L.stream()
        for gm, gs, ev in zip(self.main_graphs, self.side_graphs, self.cut_events):
            gm.replay()                            # main chain first: its records are pending when the side chain's waits are enqueued
            with torch.cuda.stream(self.stream):
                gs.replay()
            L.check(L.lib().mcav_stream_wait_event(cur, ev), "mcav_stream_wait_event(cut)")


class GraphedStep:
    def __init__(self, fwd_bwd, opt, example_inputs, capture_adam=True, warmup=2, buffers=(), dual=None):
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
            if N.WGRAD_SIDE.dual is not None and N.WGRAD_SIDE.forked:
                # two-graph capture: the bucket's weight gradients and slab reductions are nodes of the SIDE graph -- bring it level with the
                # main graph (BatchNorm gradients), then the bucket's event is a node of the side graph
                N.WGRAD_SIDE.dual.fork(torch.cuda.current_stream())
                L.check(L.lib().mcav_event_record_external(ev, L.c_p(N.WGRAD_SIDE.dual.stream.cuda_stream)), "mcav_event_record_external")
            else:
                L.check(L.lib().mcav_event_record_external(ev, L.stream()), "mcav_event_record_external")
            self.marks.append((r, ev))
        # This ROCm replays the captured branches of ONE graph on one queue (rounds 1-2), so nothing is gained by cross-stream edges inside a
        # capture: the eager path's side streams are switched off (SERIAL) and the step is captured either on one stream (dual=False) or as
        # two chains of graphs on two streams that the host and external event nodes keep in order (_DualCapture, the default).
        serial0, streams.SERIAL = streams.SERIAL, True
        try:
            want_dual = DUAL_DEFAULT if dual is None else bool(dual)
            adam_in_graph = [self.capture_adam]

            def whole():
                out = fwd_bwd(*self.static_in)
                if adam_in_graph[0]:
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
            self.dual = None
            if self.sync is not None:
                N.GRADS_READY = mark
            if want_dual:
                adam_in_graph[0] = False                  # two graphs: the update follows the join, outside both (see _DualCapture.join)
                out = self._capture_dual(whole, arena.flat.device)
            else:
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

    def __del__(self):
        try:
            from . import lib as L
            for ev in getattr(self, "_events", ()):
                L.lib().mcav_event_destroy(ev)
        except Exception:
            pass

    def _capture_dual(self, whole, device):
        """Main chain and side chain as pairs of graphs captured at the same time (class _DualCapture)."""
        from . import nn as N
        from . import streams
        self.dual = _DualCapture(device)
        torch.cuda.synchronize()
        N.WGRAD_SIDE.dual = self.dual
        streams.DUAL = self.dual
        try:
            self.dual.begin()
            try:
                with torch.cuda.stream(self.dual.main_stream):
                    out = whole()
            except BaseException:
                self.dual.end(failed=True)
                raise
            self.dual.end()
        finally:
            N.WGRAD_SIDE.dual = None
            N.WGRAD_SIDE.stream = None
            streams.DUAL = None
        self.graph = None
        return out

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        if self.capture_adam:
            self.opt.device_state()                       # lr / grad_scale / step count as the host has them now
        if self.dual is not None:
            self.dual.replay()                            # main[0], side[0], wait, main[1], side[1], wait (class _DualCapture)
        else:
            self.graph.replay()
        for r, ev in self.marks:                          # each bucket's collective waits for ITS point of the replaying graph
            self.sync.ready_range(r, ev)
        if self.capture_adam and self.dual is not None:
            self.opt.step_capturable()                    # the update, one launch behind the join (advances the host's counters itself)
        elif self.capture_adam:
            self.opt.note_replayed()
        else:
            self.arena.bump()                             # (the eager Adam that follows bumps again; harmless)
        return self.static_out


class StepGraphs:
    """One captured step per input shape (BASELINE.json configs[4]: batches of two resolutions alternate through one process)."""

    def __init__(self, fwd_bwd, opt, capture_adam=True, buffers=(), dual=None):
        self.fwd_bwd, self.opt, self.capture_adam, self.buffers, self.dual = fwd_bwd, opt, capture_adam, tuple(buffers), dual
        self.graphs = {}

    def __call__(self, *inputs):
        key = tuple(tuple(x.shape) for x in inputs)
        g = self.graphs.get(key)
        if g is None:
            g = self.graphs[key] = GraphedStep(self.fwd_bwd, self.opt, inputs, self.capture_adam, buffers=self.buffers, dual=self.dual)
        return g(*inputs)


GraphedForwardBackward = GraphedStep          # round-1 name
