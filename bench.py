#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: triplets ("images")/sec through one full training step.

One step = trainer.py:261-266 + :290-313 of the reference: zero_grad, 2 x DispResNet forward (tgt, ref0), PoseNet
forward, fused warp/L1/smoothness loss, backward through everything, (gradient all-reduce when N > 1), Adam.
Workload at N=1: BASELINE.json configs[1] -- batch 12, 192x640 KITTI-shaped triplets, ResNet-18 depth encoder +
6-DoF PoseNet, fp32.  Synthetic seeded inputs resident in HBM before the timed region; random-init weights.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     conv stage (igemm + wgrad MFMA kernels): algorithmic FLOPs / summed kernel time, measured with HIP events
               on the launch stream in a separate instrumented step after the timed region (never part of `value`)
  cpu_baseline the CPU oracle (oracle/, stock PyTorch ops) timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "unsupervised-pseuso-lidar_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

# conv FLOPs fwd per triplet at 192x640 R18 = 2 x 16.03 + 0.89 GF; fwd + dgrad + wgrad = 3x
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA (MI355X_MICROARCH.md); the bf16 conv stage also holds the fp32 launches the bf16 kernels do not cover
PEAK_HBM_GBS = 8000.0
MODE_FP32 = ("fp32 (the ResNet trunk's 3x3 stride-1 convolutions and their data gradients: fp32 contractions on split operands -- three bf16 planes per "
             "operand, six bf16-MFMA plane products, fp32 accumulation; every other launch: fp32 MFMA)")
def _latest(*names):
    for n in names:
        if os.path.exists(os.path.join(REPO, "profiles", n)):
            return os.path.join(REPO, "profiles", n)
    return os.path.join(REPO, "profiles", names[-1])


TRAFFIC_JSON = _latest("r04_traffic.json", "r03_traffic.json", "r02_traffic.json")      # rocprofv3 PMC passes (tools/pmc_summary.py)
ROCPROF_CSV = _latest("r04_kernel_stats_one_stream.csv", "r03_kernel_stats_one_stream.csv", "r02_kernel_stats_one_stream.csv")      # rocprofv3 --kernel-trace --stats of `bench.py --serial`


def PROFILE_MODE(path):
    """Which --dtype a committed profile was taken in: rounds 1-3 ran every contraction on the fp32 MFMA by default (today's fp32-mfma)."""
    return "fp32" if os.path.basename(path) >= "r04" else "fp32-mfma"
CONV_KERNELS = ("igemm_kernel", "igemm_tab_kernel", "igemm_bf16", "wgrad_kernel", "wgrad_tab_kernel", "wgrad_bf16", "conv3x3_halo", "conv3x3r_c1",
                "stem7x7s2", "splitk_finish", "wgrad_reduce", "wgrad_presum", "patch3x3", "conv3x3_patch", "wgrad3x3_patch")


def rocprof_conv_ms_per_step():
    """Conv-stage kernel time per step from the committed rocprofv3 summary of the one-stream run (the same kernels the event timer brackets:
    MFMA GEMMs, halo / stencil / stem kernels, the K-split finish and the weight-gradient slab reduction); steps = launches of the Adam kernel."""
    import csv
    try:
        rows = list(csv.reader(l for l in open(ROCPROF_CSV) if not l.startswith("#")))
    except Exception:
        return None
    steps = sum(int(r[1]) for r in rows[1:] if "adam_kernel" in r[0] or "adam_dev_kernel" in r[0])
    conv = sum(float(r[2]) for r in rows[1:] if any(c in r[0] for c in CONV_KERNELS))
    return conv / steps / 1e6 if steps else None


def measured_traffic(steps_in_profile=3):
    """HBM bytes from the committed PMC passes of this same workload: (conv stage per step, fused loss kernel per launch)."""
    try:
        k = json.load(open(TRAFFIC_JSON))["kernels"]
    except Exception:
        return None, None
    conv = sum(v["hbm_bytes_per_launch"] * v["launches"] for n, v in k.items() if any(c in n for c in CONV_KERNELS))
    # The fused loss kernel is also launched as a device-side no-op (backward with unit upstream): take its largest launch.  Its loads are
    # 4 B per lane, an access width the guide leaves uncalibrated: undoubled, FETCH_SIZE equals the compulsory read bytes (11 planes) within
    # 1 %, so it is NOT doubled here (doubling would claim that every input plane is fetched twice).
    warp = [(v.get("fetch_size_kb_max_launch_raw", 0.0) + v.get("write_size_kb_max_launch_raw", 0.0)) * 1024.0
            for n, v in k.items() if "warp_loss_l1_kernel" in n or "warp_loss_kernel" in n]
    return conv / steps_in_profile, (warp[0] if warp and warp[0] > 0 else None)


def workload_label(B, H, W, layers, ssim, dtype="fp32"):
    if dtype == "fp32-mfma":
        return "fp32-MFMA-only variant of " + workload_label(B, H, W, layers, ssim)
    if dtype == "fp32-split":
        dtype = "fp32"
    if dtype == "bf16":
        if (B, H, W, layers, ssim) == (12, 192, 640, 18, False):
            return "BASELINE.json configs[2] per-GPU shape"
        if (B, layers, ssim) == (12, 18, False) and (H, W) == (256, 832):
            return "BASELINE.json configs[4] per-GPU shape (256x832 steps)"
        return "bf16 variant"
    if (B, H, W, layers, ssim) == (12, 192, 640, 18, False):
        return "BASELINE.json configs[1]"
    if (H, W, layers, ssim) == (320, 1024, 50, True):
        return "BASELINE.json configs[3]" + ("" if B == 12 else " at batch %d" % B)
    if (B, H, W, layers, ssim) == (2, 64, 128, 18, False):
        return "BASELINE.json configs[0] shape on the GPU"
    if (B, H, W, layers, ssim) == (4, 192, 640, 18, False):
        return "north_star's 4x3x192x640 batches (configs[1] at batch 4)"
    return "variant of BASELINE.json configs[1]"


def synthetic_samples(B, H, W, rank, step=0):
    from dataloaders import synthetic_batch
    return synthetic_batch(B, H, W, seed=1234 + 1000 * rank + step)


def build(device, lr=1e-4, seed=0, depth_layers=18, ssim=False, dtype="fp32"):
    from models.depth.resnet_dispnet import DispResNet
    from models.pose.pose_net import PoseNet
    from mcav.optim import FusedAdam
    from losses import Losses
    torch.manual_seed(seed)
    # fp32 (default): fp32 results, kernel per launch (mcav.nn.DEFAULT_MMA: the trunk's 3x3 stride-1 convolutions on split operands); fp32-mfma:
    # every launch on the fp32 MFMA; bf16: the conv tiles of the depth net (98 % of the FLOPs) on bf16-rounded operands
    net_dtype = {"fp32": None, "fp32-split": "fp32-split", "fp32-mfma": "fp32-mfma", "bf16": torch.bfloat16}[dtype]
    depth = DispResNet(depth_layers, dtype=net_dtype)
    pose = PoseNet(dtype=net_dtype) if dtype in ("fp32-split", "fp32-mfma") else PoseNet()
    pose.init_weights()
    depth.to(device).train()
    pose.to(device).train()
    opt = FusedAdam(list(depth.parameters()) + list(pose.parameters()), lr)
    return depth, pose, opt, Losses(ssim=ssim)


def make_step(depth, pose, opt, crit, samples, pair=True, graph=False):
    from mcav import dist as mdist
    tgt, refs, K = samples["tgt"], samples["ref_imgs"], samples["intrinsics"]

    from mcav import nn as N
    from mcav.streams import Branch
    pose_branch = Branch()

    def fwd_bwd(tgt, ref0, ref1, K):          # the sequence of Trainer.process_batch (unsupervised-pseuso-lidar_amd/trainer.py)
        opt.zero_grad()
        N.refresh_packed_weights(tgt.device)                               # both streams read the packed filters
        poses = pose_branch.fork(pose, tgt, [ref0, ref1])                  # independent of the depth net until the loss: second stream
        disps = list(depth.forward_pair(tgt, ref0)) if pair else [depth(tgt), depth(ref0)]
        poses = pose_branch.join(poses)
        loss = crit.forward(tgt, [ref0, ref1], disps, poses, K, None)
        sum(loss).backward()
        return tuple(loss)

    runner = fwd_bwd
    adam_in_graph = False
    if graph:
        try:
            from mcav.graph import GraphedStep
            adam_in_graph = mdist.world() == 1            # one rank: the fused Adam is part of the graph (device-side step / lr scalars)
            runner = GraphedStep(fwd_bwd, opt, [tgt, refs[0], refs[1], K], capture_adam=adam_in_graph, buffers=list(depth.buffers()))
        except Exception as e:       # a failed capture must not cost the measurement: same launches, issued eagerly
            print("bench.py: hipGraph capture failed (%s: %s); running eagerly" % (type(e).__name__, e), file=sys.stderr)
            torch.cuda.synchronize()
            runner, adam_in_graph = fwd_bwd, False
    if graph:
        make_step.graphed = runner is not fwd_bwd

    def step(collective=True):
        # collective=False: rank 0's instrumented steps after the timed region (no other rank takes part in them)
        hook, N.GRADS_READY = N.GRADS_READY, (N.GRADS_READY if collective else None)
        loss = runner(tgt, refs[0], refs[1], K)
        N.GRADS_READY = hook
        if adam_in_graph:
            return loss
        opt.grad_scale = mdist.allreduce_gradients(opt.arena()) if collective else 1.0 / mdist.world()
        opt.step()
        return loss
    return step


def cpu_baseline(H, W, depth_layers=18, ssim=False, batch=4, warm=1, steps=5):
    """The CPU oracle's full step (stock PyTorch ops) on the host cores; bounded sample (SURVEY.md 8d asks for 3 + 10 steps at B = 4 and 12:
    at ~4 s per B = 4 step and ~11.5 s per B = 12 step that is minutes of CPU work inside a bench run that must finish in minutes, so the
    default is 1 warm-up + 5 timed steps at B = 4 (~25 s), median reported; images/s per core count is batch-size independent to ~10 %)."""
    from oracle import nets as onets
    from oracle.step import make_optimizer, synthetic_batch, train_step
    torch.manual_seed(0)
    depth, pose = onets.DispResNet(depth_layers), onets.PoseNet()
    pose.init_weights()
    depth.train()
    pose.train()
    opt = make_optimizer(depth, pose, 1e-4)
    s = synthetic_batch(batch, H, W, seed=1234)
    for _ in range(warm):
        train_step(depth, pose, opt, s, ssim_weight=0.85 if ssim else 0.0)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        train_step(depth, pose, opt, s, ssim_weight=0.85 if ssim else 0.0)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    dt = ts[len(ts) // 2]
    return {"value": round(batch / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "median of %d full steps (after %d warm-up) of the same %dx%d ResNet-%d+PoseNet fp32 workload at batch=%d, oracle/ in stock "
                      "PyTorch CPU ops, %.2f s/step (min %.2f, max %.2f), os.cpu_count()=%d" % (steps, warm, H, W, depth_layers, batch, dt, ts[0], ts[-1], os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 20 warm-up + 100 timed steps, median and p10 / p90 reported
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--depth-layers", type=int, default=18, help="ResNet depth of the encoder (18 = the metric's config; 50 = BASELINE.json configs[3])")
    ap.add_argument("--ssim", action="store_true", help="photometric term = 0.85 SSIM + 0.15 L1 (Losses(ssim=True); BASELINE.json configs[3] "
                                                        "stresses this kernel) instead of the reference's live L1")
    ap.add_argument("--dtype", choices=("fp32", "fp32-mfma", "fp32-split", "bf16"), default="fp32",
                    help="fp32 (default, the headline): fp32 results, kernel chosen per launch -- the trunk's 3x3 stride-1 convolutions and their data "
                         "gradients as fp32 contractions on split operands (three bf16 planes per operand, six bf16-MFMA plane products, fp32 "
                         "accumulation), everything else on the fp32 MFMA; fp32-split: the same, named explicitly; fp32-mfma: every launch on "
                         "v_mfma_f32_32x32x2_f32 (rounds 1-3's default); bf16: the depth net's conv tiles on bf16-rounded operands (BASELINE.json "
                         "configs[2] / [4]; fp32 accumulation, fp32 master weights and activations in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--separate-passes", action="store_true", help="run the two depth passes as separate launch sets (default: stacked)")
    ap.add_argument("--graph", dest="graph", action="store_true",
                    help="replay forward+backward as one captured hipGraph.  Default is eager issue: the host keeps ahead of the GPU, and the "
                         "two-stream backward (wgrad chain beside the dgrad chain) measured 18.5 ms/step eager against 20.2 ms replayed -- "
                         "the hipGraph executor of this ROCm serialises the captured branches onto one queue")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="(default) issue every launch eagerly")
    ap.set_defaults(graph=False)
    ap.add_argument("--serial", action="store_true", help="issue the whole step on ONE stream (for rocprofv3 kernel statistics whose per-kernel "
                                                          "durations are free of cross-stream overlap; slower than the default three streams)")
    ap.add_argument("--layer-report", default=None, help="write a per-launch table of the instrumented step to this file")
    args = ap.parse_args()
    if os.environ.get("MCAV_BENCH_WATCHDOG"):          # debugging aid: dump every thread's stack and exit if the run exceeds N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MCAV_BENCH_WATCHDOG"]), exit=True)

    from mcav import dist as mdist
    from mcav import nn as N
    rank, world = mdist.init_from_env(os.environ.get("MCAV_DIST_BACKEND", "nccl"))      # "gloo": rehearse the N > 1 flow on one GPU
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    B, H, W = args.batch, args.height, args.width

    depth, pose, opt, crit = build(device, depth_layers=args.depth_layers, ssim=args.ssim, dtype=args.dtype)
    mdist.broadcast_parameters(opt.arena())
    if os.environ.get("MCAV_DP_OVERLAP", "1") != "0":
        mdist.enable_overlap(opt.arena())          # N > 1: bucketed all-reduce behind the rest of backward (no-op on one rank)
    s = synthetic_samples(B, H, W, rank)
    samples = {"tgt": s["tgt"].to(device), "ref_imgs": [r.to(device) for r in s["ref_imgs"]], "intrinsics": s["intrinsics"].to(device)}
    if args.serial:
        from mcav import streams as _streams
        _streams.SERIAL = True
    step = make_step(depth, pose, opt, crit, samples, pair=not args.separate_passes, graph=args.graph)
    eager_step = step if not args.graph else make_step(depth, pose, opt, crit, samples, pair=not args.separate_passes, graph=False)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # The cyclic collector's full pass over the module / descriptor heap takes ~70 ms on the host (MCAV_BENCH_TRACE shows it landing in
    # the first timed step, where the host has no lead over the GPU to hide it): collect now and move the long-lived objects out of the
    # collector's reach, as the trainer does after its first step (trainer.py train()).
    import gc
    gc.collect()
    gc.freeze()
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # per-step durations without a host sync in the loop
    trace = os.environ.get("MCAV_BENCH_TRACE")         # debugging aid: host-side issue time of every step and the collector's pauses
    host_ms, gc_log = [], []
    if trace:
        gc.callbacks.append(lambda phase, info: gc_log.append((phase, info.get("generation"), time.perf_counter())))
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        h0 = time.perf_counter()
        loss = step()
        marks[i + 1].record()           # on the stream every stream of the step has been joined into
        host_ms.append(1000.0 * (time.perf_counter() - h0))
    fence()
    elapsed = time.perf_counter() - t0
    if trace and rank == 0:
        sys.stderr.write("host issue ms per step: %s\n" % " ".join("%.1f" % h for h in host_ms))
        starts = [(g, t) for ph, g, t in gc_log if ph == "start"]
        stops = [t for ph, g, t in gc_log if ph == "stop"]
        sys.stderr.write("gc: %s\n" % " ".join("gen%d@%.1fms(%.1fms)" % (g, 1000 * (t - t0), 1000 * (e - t)) for (g, t), e in zip(starts, stops)))
    # self-validation of the N > 1 line: every rank must hold the same parameters after the timed loop (checked BEFORE rank 0's
    # instrumented steps, which no other rank takes part in); raises -- and the run prints no line -- when they differ
    agree = mdist.check_ranks_agree(opt.arena(), loss)
    raw_steps = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    per_step = sorted(raw_steps)
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1000.0 * elapsed / args.steps
    value = world * B * args.steps / elapsed

    out = {"metric": "images/sec (fwd+bwd) KITTI 192x640 triplets, full training step", "value": round(value, 3), "unit": "images/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "fp32-split": "f32", "fp32-mfma": "f32", "bf16": "bf16"}[args.dtype],
           "data": "synthetic",
           "config": {"workload": "%s: per-GPU batch=%d, %dx%d KITTI-shaped triplets, ResNet-%d depth encoder + 6-DoF PoseNet, "
                                  "%s, one step = 2x depth fwd + pose fwd + warp/%s/smooth loss + backward + Adam%s" %
                                  (workload_label(B, H, W, args.depth_layers, args.ssim, args.dtype),
                                   B, H, W, args.depth_layers,
                                   {"fp32": MODE_FP32, "fp32-split": MODE_FP32, "fp32-mfma": "fp32, every contraction on the fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                                    "bf16": "bf16 MFMA conv tiles (fp32 accumulate / storage)"}[args.dtype],
                                   "SSIM+L1" if args.ssim else "L1", " + 1 RCCL all-reduce of the gradient arena" if world > 1 else ""),
                      "global_batch": B * world, "parallelism": "dp%d" % world},
           "loss": [round(float(l.detach()), 6) for l in loss], "hipgraph": bool(getattr(make_step, "graphed", False)),
           "ms_per_step_median": round(pct(0.5), 4), "ms_per_step_p10": round(pct(0.1), 4), "ms_per_step_p90": round(pct(0.9), 4),
           "ms_per_step_max": round(per_step[-1], 4), "slowest_step": int(max(range(args.steps), key=lambda i: raw_steps[i]))}
    if world > 1:
        gs = mdist._SYNC.get(id(opt.arena()))
        out["config"]["dp"] = {"rccl_ranks": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                               "gradient_arena_bytes": int(opt.arena().numel) * 4,
                               "allreduce_buckets_bytes": [int(b) for b in (gs.last_buckets if gs is not None else [int(opt.arena().numel) * 4])],
                               "overlap_with_backward": bool(gs is not None and not getattr(make_step, "graphed", False)),
                               "MCAV_DP_OVERLAP": os.environ.get("MCAV_DP_OVERLAP", "1"),
                               "MCAV_DP_BUCKETS": os.environ.get("MCAV_DP_BUCKETS", "decoder,layer4"),
                               "MCAV_DP_BUCKET_MB": os.environ.get("MCAV_DP_BUCKET_MB", "0"),
                               "parameters_equal_across_ranks": agree["parameters_equal_across_ranks"],
                               "parameter_checksum": agree["parameter_checksum"], "loss_per_rank": agree.get("loss_per_rank")}

    if rank == 0 and not args.no_roofline:
        # instrumented step(s): every conv kernel dispatched with its own start/stop HIP events (csrc/kernel_timer.h), on one stream
        from mcav import streams
        serial_before, streams.SERIAL = streams.SERIAL, True      # one stream: no other kernel runs beside the one being timed
        N.kernel_timer_begin()
        N.PROFILE = []
        N.PROFILE_LOSS = []
        N.PROFILE_TAGS = [] if args.layer_report else None
        for _ in range(3):
            eager_step(collective=False)      # instrumented launches must be issued eagerly (events are not graph nodes)
        torch.cuda.synchronize()
        durs = N.kernel_timer_end()
        prof = N.PROFILE
        recs = [(r.kind, r.flops, sum(durs[r.i0:r.i1])) for r in prof]
        # the slab reduction of a weight gradient = every dispatch of its record after the first (the GEMM); round 2's figure left it out
        # ... or, for the batched form the step really runs (one presum + one reduce launch per gradient bucket), a record of its own without FLOPs
        slab_ms = sum(sum(durs[(r.i0 + 1 if r.flops else r.i0):r.i1]) for r in prof if r.kind == "wgrad") / 3.0
        executed_flops = sum(r.executed for r in prof) / 3.0
        N.PROFILE = None
        streams.SERIAL = serial_before
        # every launch against ITS OWN ceiling, min(MFMA peak of the pipe it runs on, arithmetic intensity x HBM peak) with the
        # launch's algorithmic bytes (operands read once, result written once).  A launch of the split form executes six bf16-MFMA plane
        # products per fp32 product, so its MFMA ceiling in fp32-equivalent FLOPs is the bf16 peak / 6.
        def bound_of(r):
            mfma = PEAK_BF16_MFMA_TFLOPS / r.bf16_planes if r.bf16_planes else PEAK_F32_MFMA_TFLOPS
            hbm = (r.flops / r.abytes) * PEAK_HBM_GBS / 1e3 if r.abytes > 0 else float("inf")
            return (mfma, "mfma") if mfma <= hbm else (hbm, "hbm")
        n1 = len(prof) // 3
        last = [(r, sum(durs[r.i0:r.i1])) for r in prof[2 * n1:]]
        if args.layer_report:
            tags, N.PROFILE_TAGS = N.PROFILE_TAGS, None
            with open(args.layer_report, "w") as f:
                f.write("# kind tag | algorithmic GF | ms | TF/s | algorithmic MB | FLOP/B | bound TF/s = min(MFMA peak of the pipe, FLOP/B x 8 TB/s) | fraction of that bound\n")
                for (r, ms1), tag in zip(last, tags[2 * n1:]):
                    if not r.flops:
                        f.write("%-6s %-96s %8.2f GF %8.3f ms\n" % (r.kind, tag, 0.0, ms1))
                        continue
                    bd, which = bound_of(r)
                    tf = r.flops / (ms1 * 1e-3) / 1e12
                    f.write("%-6s %-96s %8.2f GF %8.3f ms %7.2f TF/s %8.1f MB %7.1f F/B  bound %6.1f (%s)  frac %.3f\n" %
                            (r.kind, tag, r.flops / 1e9, ms1, tf, r.abytes / 1e6, r.flops / max(r.abytes, 1.0), bd, which, tf / bd))
        per_class = {}
        for r, ms1 in last:
            if not r.flops:
                continue
            bd, which = bound_of(r)
            pipe = "bf16-mfma x%d planes" % r.bf16_planes if r.bf16_planes else "f32-mfma"
            c = per_class.setdefault("%s | %s | %s-bound" % (r.kind, pipe, which), {"launches": 0, "gflop": 0.0, "ms": 0.0, "bound_ms": 0.0})
            c["launches"] += 1
            c["gflop"] += r.flops / 1e9
            c["ms"] += ms1
            c["bound_ms"] += r.flops / (bd * 1e12) * 1e3            # the time the launch would take AT its own ceiling
        for c in per_class.values():
            c["tflops"] = round(c["gflop"] / c["ms"], 2)
            c["frac_of_bound"] = round(c["bound_ms"] / c["ms"], 4)   # time-weighted: sum of ceiling times / sum of measured times
            c["gflop"], c["ms"], c["bound_ms"] = round(c["gflop"], 2), round(c["ms"], 4), round(c["bound_ms"], 4)
        lrecs = [durs[i0:i1] for (_, i0, i1) in N.PROFILE_LOSS if i1 - i0 == 1]      # the forward call's ONE launch (the backward's re-run is a device-side no-op)
        N.PROFILE_LOSS = None
        ms = sum(t for (_, _, t) in recs) / 3.0
        flops = sum(f for (_, f, _) in recs) / 3.0
        by_kind = {}
        for kind, f, t in recs:
            a = by_kind.setdefault(kind, [0.0, 0.0, 0])
            a[0] += f / 3.0
            a[1] += t / 3.0
            a[2] += 1
        ach = flops / (ms * 1e-3) / 1e12
        conv_traffic, warp_traffic = measured_traffic() if (B, H, W) == (12, 192, 640) else (None, None)
        if args.ssim:
            warp_traffic = None
        peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
        mode = "fp32" if args.dtype == "fp32-split" else args.dtype
        if mode != PROFILE_MODE(TRAFFIC_JSON):
            conv_traffic = None                                    # (the committed PMC passes are of the mode they were taken in)
        # the two MFMA pipes apart: fp32-equivalent FLOPs of the launches that run the fp32 MFMA over its peak, and the bf16-MFMA FLOPs really
        # executed (planes x algorithmic) by the launches on the bf16 pipe over ITS peak
        f32_l = [(r, t) for r, t in last if r.flops and not r.bf16_planes]
        b16_l = [(r, t) for r, t in last if r.flops and r.bf16_planes]
        pipes = {}
        if f32_l:
            fl, t = sum(r.flops for r, _ in f32_l), sum(t for _, t in f32_l)
            pipes["f32_mfma"] = {"launches": len(f32_l), "algorithmic_gflop": round(fl / 1e9, 2), "ms": round(t, 3), "tflops": round(fl / t / 1e9, 2),
                                 "peak": PEAK_F32_MFMA_TFLOPS, "frac": round(fl / t / 1e9 / PEAK_F32_MFMA_TFLOPS, 4)}
        if b16_l:
            fl, t = sum(r.flops for r, _ in b16_l), sum(t for _, t in b16_l)
            ex = sum(r.flops * r.bf16_planes for r, _ in b16_l)
            pipes["bf16_mfma"] = {"launches": len(b16_l), "algorithmic_gflop": round(fl / 1e9, 2), "executed_bf16_gflop": round(ex / 1e9, 2), "ms": round(t, 3),
                                  "algorithmic_tflops": round(fl / t / 1e9, 2), "executed_bf16_tflops": round(ex / t / 1e9, 2), "peak": PEAK_BF16_MFMA_TFLOPS,
                                  "frac": round(ex / t / 1e9 / PEAK_BF16_MFMA_TFLOPS, 4),
                                  "algorithmic_over_f32_mfma_peak": round(fl / t / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
                                  "note": "algorithmic_over_f32_mfma_peak may exceed 1: these launches do not run on the fp32 MFMA instruction; "
                                          "frac is what they execute over the pipe they do run on"}
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": conv_traffic,
                           "frac_note": ("frac = the reference's algorithmic fp32 FLOPs of the whole conv stage over the fp32 MFMA instruction peak (157.3); part of the "
                                         "stage runs on the bf16 pipe (`pipes`), so this is a work-rate figure that an fp32-MFMA-only kernel set cannot exceed 1 on "
                                         "but this one could -- read `pipes` and `per_class` for each launch class against its own ceiling") if b16_l and args.dtype != "bf16" else None,
                           "pipes": pipes, "per_class": per_class,
                           "executed_frac": round(executed_flops / (ms * 1e-3) / 1e12 / peak, 4),
                           "executed_gflop_per_step": round(executed_flops / 1e9, 2),
                           "executed_note": "frac counts the reference's algorithmic FLOPs (2 M N K of every convolution); executed_frac counts the products the "
                                            "kernels form: 4 / 9 of the upsampled half in the merged-tap launches, 168 / 147 in the stem kernels (plane products "
                                            "of the split form are in pipes.bf16_mfma)",
                           "traffic_note": "HBM bytes of the conv stage per step, (2*FETCH_SIZE + WRITE_SIZE)*1024 from the rocprofv3 --pmc passes "
                                           "in %s (null when no pass of this mode is committed)" % os.path.relpath(TRAFFIC_JSON, REPO),
                           "kernel": "conv stage = implicit-GEMM forward / adjoint, weight-gradient (GEMM + its slab reduction: presum + reduce), halo and "
                                     "stencil kernels, all launches of one step; durations from per-dispatch HIP start/stop events",
                           "frac_excluding_slab_reduction": round(flops / ((ms - slab_ms) * 1e-3) / 1e12 / peak, 4),
                           "slab_reduction_ms_per_step": round(slab_ms, 3),
                           "launches_per_step": len(recs) // 3, "algorithmic_gflop_per_step": round(flops / 1e9, 2),
                           "kernel_ms_per_step": round(ms, 3),
                           "by_kind": {k: {"gflop": round(v[0] / 1e9, 2), "ms": round(v[1], 3), "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2),
                                           "launches": v[2] // 3} for k, v in by_kind.items()}}
        rp_ms = rocprof_conv_ms_per_step() if (B, H, W, args.depth_layers, mode, args.ssim) == (12, 192, 640, 18, PROFILE_MODE(ROCPROF_CSV), False) else None
        if rp_ms:
            out["roofline"]["frac_rocprof"] = round(flops / (rp_ms * 1e-3) / 1e12 / peak, 4)
            out["roofline"]["rocprof_kernel_ms_per_step"] = round(rp_ms, 3)
            out["roofline"]["rocprof_source"] = os.path.relpath(ROCPROF_CSV, REPO) + " (committed profile of an earlier run of this command with --serial; not this run)"
        if lrecs:
            lms = sum(sum(d) for d in lrecs) / len(lrecs)                     # the fused kernel (per-sample constants and the finalize are inside it)
            lmain = lms
            lbytes = 52.0 * B * H * W
            out["roofline_warp"] = {"bound": "hbm", "achieved": round(lbytes / (lms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(lbytes / (lms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": warp_traffic,
                                    "kernel": "%s (one launch: per-sample constants, warp + loss forward/backward, finalize), 52 B/pixel x %d pixels" % ("warp_loss_ssim_kernel" if args.ssim else "warp_loss_l1_kernel", B * H * W),
                                    "ms": round(lms, 4), "main_kernel_ms": round(lmain, 4),
                                    "note": "achieved = 52 B/pixel over the loss stage's one launch (per-dispatch HIP events); traffic = PMC bytes of the kernel"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # SURVEY.md 8d asks for 3 warm-up + 10 timed steps: taken when the run itself is the full-length one (--steps >= 100, the default)
        full = args.steps >= 100
        out["cpu_baseline"] = cpu_baseline(H, W, args.depth_layers, args.ssim, warm=3 if full else 1, steps=10 if full else 5)
        if full and B != 4:
            # SURVEY.md 8d names B = 4 AND the metric's own batch: a second, shorter sample at this run's batch (1 warm-up + 3 timed steps, ~45 s at 12)
            out["cpu_baseline"]["at_bench_batch"] = cpu_baseline(H, W, args.depth_layers, args.ssim, batch=B, warm=1, steps=3)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
