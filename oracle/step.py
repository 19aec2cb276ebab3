"""Oracle (test infrastructure): one full training step in stock PyTorch on CPU.

Restates reference trainer.py:290-313 (process_batch: depth net on tgt and on ref_imgs[0] as two
separate passes, pose net, Losses.forward) and trainer.py:261-266 (zero_grad, backward of
loss_mam + loss_smooth, Adam step with lr = optimizer.depth.lr over depth+pose parameters,
trainer.py:71-76).  Also the synthetic-batch generator shared by tests and bench.py.
"""
import torch

from .losses import losses_forward


def kitti_like_intrinsics(B, H, W, dtype=torch.float64):
    K = torch.tensor([[0.58 * W, 0.0, 0.5 * W], [0.0, 1.92 * H, 0.5 * H], [0.0, 0.0, 1.0]], dtype=dtype)
    return K.repeat(B, 1, 1)


def synthetic_batch(B, H, W, seed=1234, smooth=True):
    """Seeded synthetic triplet batch: randn images (3x3 box low-passed), KITTI-like K (fp64, as the loader gives)."""
    g = torch.Generator().manual_seed(seed)
    imgs = []
    for _ in range(3):
        x = torch.randn(B, 3, H, W, generator=g)
        if smooth:
            x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1)
        imgs.append(x.contiguous())
    return {"tgt": imgs[0], "ref_imgs": [imgs[1], imgs[2]], "intrinsics": kitti_like_intrinsics(B, H, W),
            "groundtruth": torch.zeros(B, 1, H, W)}


def process_batch(depth_model, pose_model, samples, ssim_weight=0.0):
    tgt = samples["tgt"]
    refs = samples["ref_imgs"]
    disps = [depth_model(tgt), depth_model(refs[0])]
    poses = pose_model(tgt, refs)
    loss = losses_forward(tgt, refs, disps, poses, samples["intrinsics"], ssim_weight)
    return [disps, poses], loss


def train_step(depth_model, pose_model, optimizer, samples, ssim_weight=0.0):
    optimizer.zero_grad()
    outputs, loss = process_batch(depth_model, pose_model, samples, ssim_weight)
    sum(loss).backward()
    optimizer.step()
    return outputs, loss


def make_optimizer(depth_model, pose_model, lr=1e-4):
    params = list(depth_model.parameters()) + list(pose_model.parameters())
    return torch.optim.Adam(params, lr)
