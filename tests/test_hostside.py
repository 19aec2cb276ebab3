"""CPU: host-side logic of the product package that needs no GPU (module structure, arenas, config plumbing)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PKG


def test_state_dict_keys_equal_reference():
    """Parameter / buffer names and shapes are the reference's (dumped from its classes by make_golden.py)."""
    from models.depth.resnet_dispnet import DepthDecoder, DispResNet
    from models.pose.pose_net import PoseNet
    want = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    for name, m in (("DispResNet", DispResNet()), ("PoseNet", PoseNet()), ("DepthDecoder", DepthDecoder(np.array([64, 64, 128, 256, 512])))):
        got = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert got == want[name], name


def test_state_dict_interchangeable_with_oracle():
    from models.depth.resnet_dispnet import DispResNet
    from oracle import nets as on
    a, b = DispResNet(), on.DispResNet()
    b.load_state_dict(a.state_dict())
    a.load_state_dict(b.state_dict())


def test_arena_views_and_grads():
    from mcav.arena import Arena, arena_of
    from models.pose.pose_net import PoseNet
    m = PoseNet()
    m.init_weights()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    params = list(m.parameters())
    a = Arena(params)
    assert a.intact() and a.numel >= sum(p.numel() for p in params)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    # parameters and gradients are views of the flat buffers
    a.flat.add_(1.0)
    assert torch.equal(m.state_dict()["conv1.0.weight"], before["conv1.0.weight"] + 1.0)
    params[0].grad.fill_(2.0)
    assert float(a.gflat[:params[0].numel()].min()) == 2.0
    a.zero_grad()
    assert float(params[0].grad.abs().max()) == 0.0
    assert arena_of(params) is a
    # load_state_dict copies in place: views survive
    m.load_state_dict(before)
    assert a.intact()
    # a re-allocation (e.g. module.to()) is detected
    params[3].data = params[3].data.clone()
    assert not a.intact()


def test_config_schema_is_the_references():
    import yaml
    cfg = yaml.full_load(open(os.path.join(PKG, "configs", "basic_config.yaml")))
    ref_keys = {"model": {"name", "depth", "pose"}, "datasets": {"path", "split", "augmentation", "sequence_length", "dataset"},
                "action": {"mode", "MLOps", "log_freq", "from_scratch", "split", "random_seed", "batch_size", "num_epochs", "num_workers",
                           "optimizer", "scheduler"}}
    for sec, keys in ref_keys.items():
        assert keys <= set(cfg[sec]), sec
    assert {"name", "file"} <= set(cfg["model"]["depth"]) and {"name", "file"} <= set(cfg["model"]["pose"])


def test_trainer_needs_gpu_and_fails_loudly():
    import yaml
    from trainer import Trainer
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = yaml.full_load(open(os.path.join(PKG, "configs", "basic_config.yaml")))
    with pytest.raises(RuntimeError):
        Trainer(cfg)


def test_synthetic_dataset_sample_contract():
    import yaml
    from dataloaders import UnSupKittiDataset
    cfg = yaml.full_load(open(os.path.join(PKG, "configs", "basic_config.yaml")))
    ds = UnSupKittiDataset(cfg)
    s = ds[3]
    assert s["tgt"].shape == (3, 192, 640) and len(s["ref_imgs"]) == 2 and s["intrinsics"].dtype == torch.float64
    assert torch.equal(ds[3]["tgt"], s["tgt"])


def test_fused_adam_loads_a_torch_adam_checkpoint():
    """The reference checkpoint's 'optimizer_state_dict' (torch.optim.Adam, trainer.py:136) loads into the fused optimiser:
    moments land in the flat buffers the kernel reads, the step count is restored, and the dict written back has Adam's format."""
    from mcav.optim import FusedAdam
    torch.manual_seed(3)
    shapes = [(4, 3, 3, 3), (4,), (2, 4, 1, 1)]
    ref_params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    ref = torch.optim.Adam(ref_params, lr=1e-3)
    for _ in range(3):
        for p in ref_params:
            p.grad = torch.randn_like(p)
        ref.step()
    sd = ref.state_dict()
    mine_params = [torch.nn.Parameter(p.detach().clone()) for p in ref_params]
    opt = FusedAdam(mine_params, lr=5e-4)
    opt.load_state_dict(sd)
    a = opt.arena()
    assert opt._step == 3 and opt.param_groups[0]["lr"] == 1e-3
    for p, o, q in zip(a.params, a.offsets, ref_params):
        n = p.numel()
        assert torch.equal(opt._m[o:o + n], ref.state[q]["exp_avg"].reshape(-1))
        assert torch.equal(opt._v[o:o + n], ref.state[q]["exp_avg_sq"].reshape(-1))
        assert opt.state[p]["exp_avg"].data_ptr() == opt._m[o:o + n].data_ptr()       # still views of the flat buffers
    back = opt.state_dict()
    fresh = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ref_params], lr=1.0)
    fresh.load_state_dict(back)                                                            # and Adam accepts what we write
    for q, f in zip(ref_params, fresh.param_groups[0]["params"]):
        assert torch.equal(fresh.state[f]["exp_avg"], ref.state[q]["exp_avg"])
        assert float(fresh.state[f]["step"]) == 3.0


def test_compute_dtype_names_select_the_contraction_form():
    """mcav.nn.set_compute_dtype: fp32 (the default since round 4: fp32 results, the trunk's 3x3 stride-1 convolutions as fp32 contractions on split
    operands, mcav_igemm_desc.mma = 2), fp32-mfma (every launch on the fp32 MFMA), fp32-split (the default's form, named), bf16 (BASELINE.json
    configs[2] / [4]) mark every convolution of a module; anything else is refused.  Host logic only."""
    import pytest
    import torch
    from mcav import depthnet as E
    from mcav import nn as N
    from mcav.lib import MCAVError
    from models.depth.resnet_dispnet import DispResNet
    net = DispResNet(18)
    convs = [m for m in net.modules() if getattr(m, "weight", None) is not None and m.weight.dim() == 4]
    assert N.DEFAULT_MMA == N.MMA_SPLIT
    assert convs and all(not hasattr(m, "_mcav_mma") for m in convs)                       # untouched modules take the default ...
    assert E.spec_of(net.encoder.encoder.layer1[0].conv1, 1, 1, N.PAD_ZERO).mma == N.DEFAULT_MMA      # ... when the engine builds their ConvSpec
    assert N.ConvSpec(convs[0].weight, None, 1, 1, N.PAD_ZERO).mma == N.MMA_FP32           # a ConvSpec built directly (kernel tests) is pinned
    for name, want in (("bf16", N.MMA_BF16), (torch.bfloat16, N.MMA_BF16), ("fp32-split", N.MMA_SPLIT), ("f32s", N.MMA_SPLIT),
                       ("fp32-mfma", N.MMA_FP32), ("fp32", N.DEFAULT_MMA), (torch.float32, N.DEFAULT_MMA), (None, N.DEFAULT_MMA)):
        N.set_compute_dtype(net, name)
        assert all(m._mcav_mma == want for m in convs), name
    with pytest.raises(MCAVError):
        N.set_compute_dtype(net, "fp16")
    assert DispResNet(18, dtype="fp32-split").encoder.encoder.conv1._mcav_mma == N.MMA_SPLIT
    assert (N.MMA_FP32, N.MMA_BF16, N.MMA_SPLIT, N.MMA_SPLIT_ALL) == (0, 1, 2, 3)        # the values of mcav_igemm_desc.mma (include/mcav_conv.h)
