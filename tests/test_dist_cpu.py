"""CPU, 2 gloo ranks: the data-parallel pieces (index sharding, one all-reduce on the flat gradient arena)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mcav import dist as mdist
    from mcav.arena import Arena
    r, w = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # different initial weights per rank on purpose
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    a = Arena(params)
    mdist.broadcast_parameters(a, 0)
    params[0].grad.fill_(float(rank + 1))
    params[1].grad.fill_(10.0 * (rank + 1))
    scale = mdist.allreduce_gradients(a)
    idx = mdist.shard_indices(list(range(11)))
    out.put((rank, a.flat.tolist(), a.gflat.tolist(), scale, idx))      # plain lists: a tensor in the queue is a handle the exiting worker takes with it
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, f0, g0, s0, i0), (_, f1, g1, s1, i1) = res
    assert f0 == f1                                         # parameters broadcast from rank 0
    assert g0 == g1 and s0 == s1 == 0.5                     # summed gradients, 1/world for Adam
    assert float(g0[0]) == 3.0 and float(g0[16]) == 30.0    # 1+2 and 10+20 (second tensor starts at the 16-float slot)
    assert i0 == [0, 1, 2, 3, 4] and i1 == [5, 6, 7, 8, 9]  # disjoint equal slices, remainder dropped


def _worker_overlap(rank, world, port, out):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mcav import dist as mdist
    from mcav import nn as N
    from mcav.arena import Arena
    mdist.init_from_env("gloo")
    torch.manual_seed(7)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (5, 9, 16, 3, 8, 2)]
    a = Arena(params)
    mdist.enable_overlap(a)
    assert N.GRADS_READY is not None
    results = []
    for step in range(2):                                  # two steps: the bucket bookkeeping resets
        g = torch.Generator().manual_seed(1000 * step + rank)
        a.gflat.copy_(torch.randn(a.numel, generator=g))
        N.grads_ready(params[3:5])                         # "decoder" announced first
        N.grads_ready(params[1:3])                         # then another contiguous group
        N.grads_ready([params[0], params[5]])              # not contiguous: ignored, left to finish()
        N.grads_ready(params[2:4])                         # overlaps ranges already in flight: ignored
        scale = mdist.allreduce_gradients(a)
        results.append(a.gflat.tolist())
    gs = mdist._SYNC[id(a)]
    buckets = list(gs.last_buckets)
    # the remainder in collectives of at most 32 bytes (MCAV_DP_BUCKET_MB): same sums, more collectives
    os.environ["MCAV_DP_BUCKET_MB"] = str(32.0 / (1 << 20))
    g = torch.Generator().manual_seed(5000 + rank)
    a.gflat.copy_(torch.randn(a.numel, generator=g))
    N.grads_ready(params[3:5])
    mdist.allreduce_gradients(a)
    results.append(a.gflat.tolist())
    small = list(gs.last_buckets)
    # bench.py's self-validation: equal parameters pass, a rank that drifted by one ulp in one element is caught
    agree = mdist.check_ranks_agree(a, [torch.tensor(float(rank))])
    caught = False
    if rank == 1:
        with torch.no_grad():
            a.flat[3] = torch.nextafter(a.flat[3], torch.tensor(float("inf")))
    try:
        mdist.check_ranks_agree(a)
    except RuntimeError:
        caught = True
    out.put((rank, results, scale, buckets, small, agree, caught))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_overlap_equals_one_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, r0, s0, b0, sm0, ag0, c0), (_, r1, s1, b1, sm1, ag1, c1) = res
    assert s0 == s1 == 0.5
    numel = len(r0[0])
    for step, seed in ((0, 0), (1, 1000), (2, 5000)):
        want = sum(torch.randn(numel, generator=torch.Generator().manual_seed(seed + r)) for r in range(2))
        assert r0[step] == r1[step]
        assert torch.equal(torch.tensor(r0[step]), want)           # every element summed exactly once
    assert b0 == b1 and len(b0) == 4 and sum(b0) == 4 * numel     # two announced buckets + the two remainder ranges around them
    assert sm0 == sm1 and max(sm0[1:]) <= 32 and sum(sm0) == 4 * numel and len(sm0) > len(b0)
    assert ag0["parameters_equal_across_ranks"] and ag0["ranks"] == 2 and ag0["loss_per_rank"] == [[0.0], [1.0]] and ag0 == ag1
    assert c0 and c1                                               # the one-ulp drift of rank 1 raised on BOTH ranks


def test_announced_stage_knob(monkeypatch):
    from mcav import dist as mdist
    monkeypatch.delenv("MCAV_DP_BUCKETS", raising=False)
    assert mdist.announced_stages() == {"decoder", "layer4"}
    monkeypatch.setenv("MCAV_DP_BUCKETS", "none")
    assert mdist.announced_stages() == frozenset()
    monkeypatch.setenv("MCAV_DP_BUCKETS", "decoder, layer4,layer3")
    assert mdist.announced_stages() == {"decoder", "layer4", "layer3"}
    monkeypatch.setenv("MCAV_DP_BUCKET_MB", "16")
    assert mdist.remainder_bucket_elems() == 4 << 20


def test_gradsync_buckets_on_the_real_networks():
    """The ranges the depth net's backward announces (decoder first, then encoder layer4: models/depth/resnet_dispnet.py _DispResNetPairFn,
    mcav/depthnet.py encoder_backward) must be contiguous runs of the REAL DispResNet + PoseNet arena, disjoint, and cover most of its
    bytes -- otherwise GradSync.ready() silently leaves everything to finish() and nothing overlaps with backward."""
    from mcav import dist as mdist
    from mcav.arena import Arena
    from models.depth.resnet_dispnet import DispResNet
    from models.pose.pose_net import PoseNet
    for layers, min_cover in ((18, 0.70), (50, 0.55)):
        depth, pose = DispResNet(layers), PoseNet()
        params = list(depth.parameters()) + list(pose.parameters())
        a = Arena(params)
        gs = mdist.GradSync(a)
        r_dec = gs.span(list(depth.decoder.parameters()))
        r_l4 = gs.span(list(depth.encoder.encoder.layer4.parameters()))
        assert r_dec is not None and r_l4 is not None, "announced groups are not contiguous arena ranges"
        assert r_dec[1] <= r_l4[0] or r_l4[1] <= r_dec[0], "announced ranges overlap"
        for lo, hi in (r_dec, r_l4):
            assert 0 <= lo < hi <= a.numel
        # each range holds exactly its group's parameters (16-byte slot padding included, nothing else)
        n_dec = sum((p.numel() + 3) // 4 * 4 for p in depth.decoder.parameters())
        n_l4 = sum((p.numel() + 3) // 4 * 4 for p in depth.encoder.encoder.layer4.parameters())
        assert r_dec[1] - r_dec[0] == n_dec and r_l4[1] - r_l4[0] == n_l4
        cover = (n_dec + n_l4) / a.numel
        assert cover >= min_cover, (layers, cover)
        # a group that is NOT one run (layer4 + pose net: the decoder lies between them) is refused rather than mis-reduced
        assert gs.span(list(depth.encoder.encoder.layer4.parameters()) + list(pose.parameters())) is None
