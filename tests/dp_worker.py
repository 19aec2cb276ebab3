"""One data-parallel rank of tests/test_dist_gpu.py (spawned as a subprocess; ranks share GPU 0 and talk over gloo).

Runs the REAL DispResNet + PoseNet through Trainer.train_step on this rank's half of a seeded batch and writes the all-reduced gradient
arena and the updated parameters to a file for the parent to compare with the single-process result."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "unsupervised-pseuso-lidar_amd")
for p in (REPO, PKG, os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def build_config(H, W, batch, graph):
    import yaml
    cfg = yaml.full_load(open(os.path.join(PKG, "configs", "basic_config.yaml")))
    cfg["datasets"]["augmentation"].update(image_width=W, image_height=H)
    cfg["datasets"]["synthetic_length"] = 8
    cfg["action"].update(batch_size=batch, verbose=False, save_checkpoints=False, from_scratch=True, hipgraph=bool(graph), num_workers=0)
    return cfg


def seed_models(trainer):
    """Identical, seeded weights on every rank and in the single-process reference (by parameter name)."""
    import torch
    from seeding import reinit_by_name
    reinit_by_name(trainer.depth_model, 141)
    reinit_by_name(trainer.pose_model, 121)
    with torch.no_grad():
        trainer.pose_model.pose_pred.weight.mul_(0.1)
        trainer.pose_model.pose_pred.bias.mul_(0.1)
    trainer.model_optimizer.arena().bump()


def half(samples, r, world):
    n = samples["tgt"].shape[0] // world
    sl = slice(r * n, (r + 1) * n)
    return {"tgt": samples["tgt"][sl].contiguous(), "ref_imgs": [x[sl].contiguous() for x in samples["ref_imgs"]],
            "intrinsics": samples["intrinsics"][sl].contiguous(), "groundtruth": samples["groundtruth"][sl].contiguous()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--overlap", type=int, default=1)
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--dtype", default="fp32")
    args = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.port), RANK=str(args.rank), WORLD_SIZE=str(args.world), LOCAL_RANK="0",
                      MCAV_DP_OVERLAP=str(args.overlap))
    if args.world == 1:
        os.environ["MCAV_DP_FORCE"] = "1"             # a 1-rank group still takes the collective path (the RCCL dry run of test_dist_gpu.py)
    import torch
    from mcav import dist as mdist
    mdist.init_from_env(args.backend)                 # gloo: several ranks on one GPU (RCCL needs one GPU per rank: "nccl" only with --world 1)
    from oracle.step import synthetic_batch
    from trainer import Trainer
    H, W, B = 64, 128, 4
    t = Trainer(build_config(H, W, B // args.world, args.graph))
    if args.dtype != "fp32":
        from mcav import nn as N
        N.set_compute_dtype(t.depth_model, args.dtype)
    seed_models(t)
    t.set_train()
    for k in range(args.steps):
        s = half(synthetic_batch(B, H, W, seed=70 + k), args.rank, args.world)
        t.train_step(s)
    torch.cuda.synchronize()
    opt = t.model_optimizer
    g = getattr(t, "_graphs", None)
    marks = max([len(gs.marks) for gs in g.graphs.values()], default=0) if g is not None else 0
    torch.save({"gflat": opt.arena().gflat.cpu(), "flat": opt.arena().flat.cpu(), "scale": opt.grad_scale, "backend": torch.distributed.get_backend(),
                "buckets": list(getattr(mdist._SYNC.get(id(opt.arena())), "last_buckets", [])), "graph_marks": marks}, args.out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
