"""Data parallelism: one process per GPU, one RCCL all-reduce of the flat gradient arena per step.

The reference has no distributed code; this is the single exchange step the sharded path needs.  Each rank takes a
disjoint slice of the batch stream, gradients are SUMMED over ranks in one collective on the contiguous arena
(63.7 MB fp32 for ResNet-18 + PoseNet) and scaled by 1/world_size inside the fused Adam kernel, every rank applies
the same update.  BatchNorm uses per-rank batch statistics (as stock DDP without SyncBN).
"""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def parallel():
    """True when the collective path runs: more than one rank -- or MCAV_DP_FORCE=1 with an initialised 1-rank group (the RCCL dry run on a
    one-GPU box: communicator, streams and Work.wait() ordering exercised, the sums are the identity)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MCAV_DP_FORCE", "0") == "1"


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend=None):
    """torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR/PORT).  backend 'nccl' is RCCL on ROCm."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if (ws <= 1 and os.environ.get("MCAV_DP_FORCE", "0") != "1") or dist.is_initialized():
        return rank(), world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, device_id=torch.device("cuda", local))      # bind the communicator to this rank's GPU
    else:
        dist.init_process_group(backend=backend)
    return rank(), world()


def broadcast_parameters(arena, src=0):
    if parallel():
        dist.broadcast(arena.flat, src)
        arena.bump()


def announced_stages():
    """MCAV_DP_BUCKETS: which parameter groups a backward pass announces as final (each becomes one asynchronous bucket of the all-reduce,
    in this order of completion): any of decoder, layer4, layer3, layer2, layer1; "none" = one collective after backward.  Default
    "decoder,layer4" -- 70 % of the arena's bytes with half of the backward still to run (xGMI sizing in DESIGN.md section 7)."""
    v = os.environ.get("MCAV_DP_BUCKETS", "decoder,layer4").strip().lower()
    return frozenset() if v in ("", "none", "0") else frozenset(t.strip() for t in v.split(","))


def remainder_bucket_elems():
    """MCAV_DP_BUCKET_MB: the part of the arena no announcement covered goes out in collectives of at most this many MiB (0 = one)."""
    mb = float(os.environ.get("MCAV_DP_BUCKET_MB", "0") or 0)
    return int(mb * (1 << 20) / 4) if mb > 0 else 0


class GradSync:
    """The gradient all-reduce, overlapped with backward in a few contiguous buckets of the flat arena.

    A network's backward announces parameter groups whose gradients are final (`mcav.nn.grads_ready`: the decoder, then
    encoder layer4 -- 70 % of the bytes are known with half of the backward still to run).  Each announcement all-reduces the
    arena range those parameters span, asynchronously, ordered after the work issued so far on the announcing stream and on
    the weight-gradient stream; `finish()` reduces whatever range was not announced and waits for everything.  Every rank
    runs the same code, so the collectives are issued in the same order everywhere.  Summation is per element, so the
    result is identical to one all-reduce of the whole arena."""

    def __init__(self, arena):
        self.arena = arena
        self.works = []
        self.done = []          # [lo, hi) ranges already handed to a collective this step
        self.sizes = []         # bytes of each collective issued this step, in issue order
        self.last_buckets = []  # ... of the last finished step (bench.py reports it)
        self.comm = None        # communication stream of ready_range()

    def span(self, params):
        a = self.arena
        index = {id(p): i for i, p in enumerate(a.params)}
        idx = sorted(index[id(p)] for p in params if id(p) in index)
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return None                                           # not one contiguous run of arena slots: leave it to finish()
        last = idx[-1]
        hi = a.offsets[last + 1] if last + 1 < len(a.offsets) else a.numel
        return a.offsets[idx[0]], hi

    def ready(self, params):
        if not parallel():
            return
        r = self.span(params)
        if r is None or any(not (r[1] <= lo or hi <= r[0]) for lo, hi in self.done):
            return
        from . import nn as N
        buf = self.arena.gflat[r[0]:r[1]]
        side = N.WGRAD_SIDE.stream if (buf.is_cuda and N.WGRAD_SIDE.forked) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())          # BatchNorm gradients are written on the calling stream
            with torch.cuda.stream(side):                          # ... and the weight gradients on the wgrad stream, in order
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
        self.done.append(r)
        self.sizes.append(4 * (r[1] - r[0]))

    def ready_range(self, r, event=None):
        """All-reduce the arena range r = [lo, hi) asynchronously on the communication stream.  event: an mcav external event (mcav/graph.py)
        that a REPLAYING hipGraph signals when the gradients of this range are final -- the communication stream waits for it, so the
        collective starts while the rest of the graph's backward pass is still running (replay and overlap together: BASELINE.json configs[4])."""
        if not parallel() or r is None or any(not (r[1] <= lo or hi <= r[0]) for lo, hi in self.done):
            return
        from . import lib as L
        buf = self.arena.gflat[r[0]:r[1]]
        if self.comm is None or self.comm.device != buf.device:
            self.comm = torch.cuda.Stream(device=buf.device)
        if event is not None:
            L.check(L.lib().mcav_stream_wait_event(L.c_p(self.comm.cuda_stream), event), "mcav_stream_wait_event")
        else:
            self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
        self.done.append(r)
        self.sizes.append(4 * (r[1] - r[0]))

    def finish(self):
        """Call after backward() has returned (all streams joined).  Reduces the remaining ranges, then waits for all."""
        if parallel():
            pos = 0
            step = remainder_bucket_elems()
            for lo, hi in sorted(self.done) + [(self.arena.numel, self.arena.numel)]:
                while lo > pos:
                    end = min(lo, pos + step) if step else lo
                    self.works.append(dist.all_reduce(self.arena.gflat[pos:end], op=dist.ReduceOp.SUM, async_op=True))
                    self.sizes.append(4 * (end - pos))
                    pos = end
                pos = max(pos, hi)
            for w in self.works:
                w.wait()
            self.last_buckets = self.sizes
        self.works, self.done, self.sizes = [], [], []


_SYNC = {}


def enable_overlap(arena):
    """Overlap the gradient all-reduce of this arena with backward (no-op on one rank).  Returns the GradSync."""
    from . import nn as N
    gs = GradSync(arena)
    _SYNC[id(arena)] = gs
    N.GRADS_READY = gs.ready if parallel() else None
    return gs


def allreduce_gradients(arena):
    """Sum the gradient arena over ranks.  Returns the scale Adam must apply (1/world).
    One collective, or -- after enable_overlap(arena) -- the remainder of the bucketed, backward-overlapped reduction."""
    w = world()
    if parallel():
        gs = _SYNC.get(id(arena))
        if gs is not None and gs.arena is arena:
            gs.finish()
        else:
            dist.all_reduce(arena.gflat, op=dist.ReduceOp.SUM)
    return 1.0 / w


def shard_indices(indices, r=None, w=None):
    """Disjoint, equal-sized slices of a (seeded-shuffled) index list; the tail that does not divide is dropped."""
    r = rank() if r is None else r
    w = world() if w is None else w
    per = len(indices) // w
    return indices[r * per:(r + 1) * per]


def check_ranks_agree(arena, loss=None):
    """After a run: every rank must hold the SAME parameters (same seed, same all-reduced gradients, same Adam update).  All-gathers two
    float64 checksums of the parameter arena (sum, sum of |.|) and the last losses; raises when a rank differs in a single bit of either
    checksum.  -> dict for the bench line.  One rank: trivially true."""
    flat = arena.flat.detach()
    mine = torch.stack([flat.double().sum(), flat.double().abs().sum()] + [l.detach().double().reshape(()) for l in (loss or [])]).reshape(1, -1)
    w = world()
    if w == 1:
        return {"ranks": 1, "parameters_equal_across_ranks": True, "parameter_checksum": [float(mine[0, 0]), float(mine[0, 1])]}
    allv = [torch.zeros_like(mine) for _ in range(w)]
    dist.all_gather(allv, mine)
    allv = torch.cat(allv, 0).cpu()
    equal = bool((allv[:, :2] == allv[0:1, :2]).all())
    out = {"ranks": w, "parameters_equal_across_ranks": equal, "parameter_checksum": [float(allv[0, 0]), float(allv[0, 1])],
           "loss_per_rank": [[round(float(x), 6) for x in row[2:]] for row in allv]}
    if not equal:
        raise RuntimeError("data-parallel ranks hold different parameters after the run: checksums %s" % allv[:, :2].tolist())
    return out
