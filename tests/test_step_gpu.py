"""GPU: one full training step (trainer.py:261-266 + 290-313) of the HIP path against the CPU oracle's step."""
import copy

import numpy as np
import pytest
import torch

from conftest import rel_err
from seeding import reinit_by_name

pytestmark = pytest.mark.gpu
DEV = "cuda"


def damp_residual_branches(net, gamma3):
    """The last BatchNorm of every Bottleneck gets its gamma scaled by gamma3 (the standard "zero-init residual" idea, not all the way to
    zero so that every convolution of the branch keeps a gradient): the residual branches add a fifth of a unit of variance each instead of
    a whole one, which is the regime trained ResNet-50s live in.  At the seeding's gamma ~ 1 the 16 stacked train-mode BatchNorm branches
    amplify a 1e-6 perturbation into 3e-2 of layer3 / layer4's gradients in float64 (tests/arbiter.py envelope); damped, the CPU fp32 oracle
    sits within 8e-4 of float64 on every tensor (median 3e-6), so an absolute bound on the HIP path means something."""
    with torch.no_grad():
        for n, m in net.named_modules():
            if n.endswith("bn3"):
                m.weight.mul_(gamma3)
    return net


def build_pair(seed_d=141, seed_p=121, layers=18, gamma3=None):
    from models.depth.resnet_dispnet import DispResNet
    from models.pose.pose_net import PoseNet
    from oracle import nets as on
    hip_d, hip_p = reinit_by_name(DispResNet(layers), seed_d), reinit_by_name(PoseNet(), seed_p)
    if gamma3 is not None:
        damp_residual_branches(hip_d, gamma3)
    ref_d, ref_p = on.DispResNet(layers), on.PoseNet()
    ref_d.load_state_dict(hip_d.state_dict())        # same names and shapes: the state_dict is interchangeable
    ref_p.load_state_dict(hip_p.state_dict())
    with torch.no_grad():                            # small pose outputs, as a trained/initialised PoseNet gives
        for m in (hip_p, ref_p):
            m.pose_pred.weight.mul_(0.1)
            m.pose_pred.bias.mul_(0.1)
    return hip_d.to(DEV).train(), hip_p.to(DEV).train(), ref_d.train(), ref_p.train()


# the last two cases are BASELINE.json configs[3]'s combination (ResNet-50 encoder + the fused warp + SSIM loss) as ONE step
@pytest.mark.parametrize("B,H,W,pair,ssim,layers", [(2, 64, 128, False, False, 18), (3, 96, 160, False, False, 18), (2, 64, 128, True, False, 18),
                                                     (3, 96, 160, True, False, 18), (2, 64, 128, True, True, 18),
                                                     (4, 64, 128, True, True, 50), (2, 128, 192, True, True, 50),
                                                     # ResNet-50 where the problem is well conditioned (damped residual branches, batch 8): the
                                                     # absolute 2e-3 bound of the ResNet-18 cases applies
                                                     (8, 96, 160, True, False, -50)])
def test_train_step_vs_oracle(B, H, W, pair, ssim, layers):
    damped = layers < 0
    layers = abs(layers)
    from losses import Losses
    from mcav.optim import FusedAdam
    from oracle.step import make_optimizer, synthetic_batch, train_step
    hip_d, hip_p, ref_d, ref_p = build_pair(layers=layers, gamma3=0.2 if damped else None)
    s = synthetic_batch(B, H, W, seed=5)
    from arbiter import Verdicts, double_copy, to_double
    d64, p64 = double_copy(ref_d), double_copy(ref_p)          # the float64 arbiter: same weights, same code, before the update
    ropt = make_optimizer(ref_d, ref_p, 1e-4)
    (rdisps, rposes), rloss = train_step(ref_d, ref_p, ropt, s, ssim_weight=0.85 if ssim else 0.0)

    opt = FusedAdam(list(hip_d.parameters()) + list(hip_p.parameters()), 1e-4)
    tgt, refs, K = s["tgt"].to(DEV), [r.to(DEV) for r in s["ref_imgs"]], s["intrinsics"].to(DEV)
    opt.zero_grad()
    disps = list(hip_d.forward_pair(tgt, refs[0])) if pair else [hip_d(tgt), hip_d(refs[0])]
    poses = hip_p(tgt, refs)
    loss = Losses(ssim=ssim).forward(tgt, refs, disps, poses, K, None)
    sum(loss).backward()
    # forward parity: depth maps within 1e-3 relative (north_star), AbsRel reported
    for got, want in zip(disps, rdisps):
        dg, dw = 1 / (10 * got[0].detach().cpu() + 0.01), 1 / (10 * want[0].detach() + 0.01)
        assert float(((dg - dw).abs() / dw).max()) < 1e-3
        assert float(((dg - dw).abs() / dw).mean()) < 1e-4
    assert rel_err(poses, rposes) < 1e-3
    assert abs(float(loss[0]) - float(rloss[0])) < 1e-3 * abs(float(rloss[0]))
    assert abs(float(loss[1]) - float(rloss[1])) < 1e-3 * abs(float(rloss[1]))
    # backward parity on every parameter that receives a gradient: the fp64 arbiter (tests/arbiter.py).  The step in float64 (same weights),
    # once as it is and twice on 1e-6-perturbed weights and images (the envelope); per parameter the HIP gradient must be as close to the
    # float64 one as max(CPU fp32 oracle, envelope) allows, and within an absolute bound read off the MI355X runs.
    from arbiter import perturb_, perturb_tensor
    from oracle.step import process_batch
    s64 = to_double(s)

    def step64(dnet, pnet, smp):
        (dd, pp), ll = process_batch(dnet, pnet, smp, ssim_weight=0.85 if ssim else 0.0)
        sum(ll).backward()
        return dd, pp, [q.grad for q in list(dnet.parameters()) + list(pnet.parameters())]
    envs = []
    for e in range(2):
        de, pe = perturb_(double_copy(d64), 1e-6, 300 + e), perturb_(double_copy(p64), 1e-6, 400 + e)
        se = dict(s64, tgt=perturb_tensor(s64["tgt"], 1e-6, 500 + e), ref_imgs=[perturb_tensor(r, 1e-6, 600 + 10 * e + i) for i, r in enumerate(s64["ref_imgs"])])
        envs.append(step64(de, pe, se)[2])
    # Selection flips: dL/dposes moves in discrete steps when ONE pixel's bilinear cell / L1 sign / SSIM clamp is decided the other way, and a
    # pixel within rounding of such a kink is decided by the rounding of whichever fp32 evaluation looks at it (tests/flip_finder.py;
    # test_loss_gpu.py::test_hip_pose_gradient_gaps_are_named_tie_pixels names them on this very case).  Round 2 widened the envelope by the
    # CPU fp32 oracle's step on 1e-5-perturbed inputs; that is gone.  Instead the HIP loss kernel's per-pixel dump is diffed against float64
    # on IDENTICAL inputs, every pixel decided differently must be a named tie (float64 margin at rounding level), and the float64 reference
    # of the tensors downstream of the poses -- PoseNet's parameters, nothing else -- takes the same side at exactly those pixels.
    import flip_finder as ff
    taps, _, _ = ff.hip_taps(tgt, refs, disps[0][0].detach().contiguous(), disps[1][0].detach().contiguous(), poses.detach().contiguous(), K, ssim=ssim)
    hip_inputs = (s["tgt"], s["ref_imgs"], disps[0][0].detach().cpu(), disps[1][0].detach().cpu(), poses.detach().cpu())
    delta, flips, untied = ff.flip_correction(hip_inputs, taps, s["intrinsics"], 0.85 if ssim else 0.0)
    print("pixels the HIP loss kernel decides differently from float64: %s" % [(f["b"], f["warp"], f["y"], f["x"], f["kind"], "%.1e" % f["margin"]) for f in flips])
    assert not untied, "decided differently from float64 without a tie: %s" % untied
    disps64, poses64, g64 = step64(d64, p64, s64)
    n_depth = len(list(d64.parameters()))
    if flips:      # float64 PoseNet gradients with the named pixels on the HIP kernel's side: g + (d poses / d theta)^T delta
        pp = p64(s64["tgt"], s64["ref_imgs"])
        pparams = [q for q in p64.parameters()]
        extra = torch.autograd.grad(pp, pparams, grad_outputs=delta, allow_unused=True)
        g64 = list(g64[:n_depth]) + [g if e is None else g + e for g, e in zip(g64[n_depth:], extra)]
    v = Verdicts(floor=2.5e-4)
    v.add("disp(tgt)", disps[0][0], rdisps[0][0], disps64[0][0])
    v.add("poses", poses, rposes, poses64)
    names = [n for n, _ in list(hip_d.named_parameters()) + list(hip_p.named_parameters())]
    hip_params = list(hip_d.parameters()) + list(hip_p.parameters())
    ref_params = list(ref_d.parameters()) + list(ref_p.parameters())
    for i, (n, p, q) in enumerate(zip(names, hip_params, ref_params)):
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        v.add(n, p.grad, q.grad, g64[i], [env[i] for env in envs])
    v.check("test_train_step_vs_oracle[%d-%d-%d-%s-%s-R%d%s]" % (B, H, W, pair, ssim, layers, "-damped" if damped else ""),
            hip_abs=2e-3 if (layers == 18 or damped) else None)
    # (ResNet-50 at the seeding's gamma ~ 1 and batch 2-4 is ill-conditioned -- the 1e-6 perturbation alone moves layer3/4 gradients by 3e-2
    #  in float64 -- so those two cases get the relative rule only; the damped batch-8 case carries the absolute bound)
    # optimiser parity after the update
    opt.step()
    torch.cuda.synchronize()
    for (n, p), (_, q) in zip(list(hip_d.named_parameters()) + list(hip_p.named_parameters()),
                              list(ref_d.named_parameters()) + list(ref_p.named_parameters())):
        assert float((p.detach().cpu() - q.detach()).abs().max()) <= 2.05e-4, n        # |Adam step| <= lr; sign flips of tiny grads allowed
    moved = sum(float((p.detach().cpu() - q.detach()).abs().mean()) for (_, p), (_, q) in zip(hip_p.named_parameters(), ref_p.named_parameters()))
    assert moved < 1e-4
    # BatchNorm running statistics were updated twice (two depth passes), in order
    assert int(hip_d.encoder.encoder.bn1.num_batches_tracked) == 2
    assert rel_err(hip_d.encoder.encoder.bn1.running_var, ref_d.encoder.encoder.bn1.running_var) < 1e-4


def test_mixed_resolution_steps_share_no_state():
    """BASELINE.json configs[4] feeds batches of different resolutions through one process: offset tables, workspaces and packed
    filters cached for one shape must not leak into the next.  forward+backward at shape A, then B, then A again: bit-identical."""
    from losses import Losses
    from oracle.step import synthetic_batch
    hip_d, hip_p, _, _ = build_pair()
    params = list(hip_d.parameters()) + list(hip_p.parameters())

    def run(B, H, W, seed):
        s = synthetic_batch(B, H, W, seed=seed)
        tgt, refs, K = s["tgt"].to(DEV), [r.to(DEV) for r in s["ref_imgs"]], s["intrinsics"].to(DEV)
        for p in params:
            p.grad = None
        disps = list(hip_d.forward_pair(tgt, refs[0]))
        loss = Losses().forward(tgt, refs, disps, hip_p(tgt, refs), K, None)
        sum(loss).backward()
        torch.cuda.synchronize()
        return [float(l) for l in loss], [p.grad.clone() for p in params if p.grad is not None]

    la, ga = run(2, 64, 128, 5)
    lb, _ = run(3, 96, 160, 6)
    lc, _ = run(1, 128, 416, 7)            # 256x832 / 2
    la2, ga2 = run(2, 64, 128, 5)
    assert la == la2 and lb != la and lc != la
    assert all(torch.equal(x, y) for x, y in zip(ga, ga2))


def test_mixed_resolution_full_size_alternation():
    """BASELINE.json configs[4] at its real shapes: batch-12 steps alternating 192x640 and 256x832 through one process (one resolution per
    step).  The step at 192x640 gives bit-identical losses and gradients before and after a 256x832 step from the same weights: nothing
    cached for one resolution (offset tables, workspaces, packed filters, stream state) leaks into the other."""
    from losses import Losses
    from mcav.optim import FusedAdam
    from mcav.streams import Branch
    from oracle.step import synthetic_batch
    hip_d, hip_p, _, _ = build_pair()
    opt = FusedAdam(list(hip_d.parameters()) + list(hip_p.parameters()), 1e-4)
    state = {k: v.clone() for k, v in hip_d.state_dict().items()}
    branch = Branch()

    def run(H, W, seed):
        s = synthetic_batch(12, H, W, seed=seed)
        tgt, refs, K = s["tgt"].to(DEV), [r.to(DEV) for r in s["ref_imgs"]], s["intrinsics"].to(DEV)
        hip_d.load_state_dict(state)
        opt.zero_grad()
        poses = branch.fork(hip_p, tgt, refs)
        disps = list(hip_d.forward_pair(tgt, refs[0]))
        poses = branch.join(poses)
        loss = Losses().forward(tgt, refs, disps, poses, K, None)
        sum(loss).backward()
        torch.cuda.synchronize()
        return [float(l.detach()) for l in loss], opt.arena().gflat.clone()

    la, ga = run(192, 640, 21)
    lb, gb = run(256, 832, 22)
    la2, ga2 = run(192, 640, 21)
    lb2, gb2 = run(256, 832, 22)
    assert la == la2 and torch.equal(ga, ga2)
    assert lb == lb2 and torch.equal(gb, gb2)
    assert la != lb and torch.isfinite(gb).all() and float(gb.abs().max()) > 0


def test_config3_full_size_r50_ssim_properties():
    """BASELINE.json configs[3] at full size (batch 12, 320x1024, ResNet-50 encoder, fused warp + SSIM loss), properties that need no oracle:
    the whole step is bit-reproducible; the stacked depth passes equal two separate passes (loss to 1e-5, gradients to 1e-3 L2);
    eval-mode depth maps of the two forms agree within the north_star's 1e-3."""
    from losses import Losses
    from mcav.optim import FusedAdam
    from mcav.streams import Branch
    from oracle.step import synthetic_batch
    hip_d, hip_p, _, _ = build_pair(layers=50)
    s = synthetic_batch(12, 320, 1024, seed=19)
    tgt, refs, K = s["tgt"].to(DEV), [r.to(DEV) for r in s["ref_imgs"]], s["intrinsics"].to(DEV)
    opt = FusedAdam(list(hip_d.parameters()) + list(hip_p.parameters()), 1e-4)
    state = {k: v.clone() for k, v in hip_d.state_dict().items()}
    branch = Branch()
    crit = Losses(ssim=True)

    def step(pair=True):
        hip_d.load_state_dict(state)
        opt.zero_grad()
        poses = branch.fork(hip_p, tgt, refs)
        disps = list(hip_d.forward_pair(tgt, refs[0])) if pair else [hip_d(tgt), hip_d(refs[0])]
        poses = branch.join(poses)
        loss = crit.forward(tgt, refs, disps, poses, K, None)
        sum(loss).backward()
        torch.cuda.synchronize()
        return [float(l.detach()) for l in loss], opt.arena().gflat.clone()

    l1, g1 = step()
    l2, g2 = step()
    assert l1 == l2 and torch.equal(g1, g2)
    assert float(g1.abs().max()) > 0 and torch.isfinite(g1).all()
    l3, g3 = step(pair=False)
    for a, b in zip(l1, l3):
        assert abs(a - b) <= 1e-5 * abs(b)
    assert float((g1 - g3).norm() / g3.norm()) < 1e-3
    del g1, g2, g3
    hip_d.eval()
    with torch.no_grad():
        pa, _ = hip_d.forward_pair(tgt, refs[0])
        sa = hip_d(tgt)
    ds, dn = 1 / (10 * pa[0] + 0.01), 1 / (10 * sa[0] + 0.01)
    assert float(((ds - dn).abs() / dn).max()) < 1e-3 and float(((ds - dn).abs() / dn).mean()) < 3e-5


def test_trainer_resume_equals_uninterrupted(tmp_path, monkeypatch):
    """Reference trainer.py:143-152: a resumed run restores the networks AND the optimiser (Adam moments, step count).  Two steps, checkpoint,
    third step == construct a new Trainer from the checkpoint (from_scratch: False), third step: parameters and moments bit for bit."""
    import os
    import yaml
    from conftest import PKG
    from oracle.step import synthetic_batch
    from trainer import Trainer
    monkeypatch.chdir(tmp_path)
    cfg = yaml.full_load(open(os.path.join(PKG, "configs", "basic_config.yaml")))
    cfg["datasets"]["augmentation"].update(image_width=128, image_height=64)
    cfg["datasets"]["synthetic_length"] = 10
    cfg["action"].update(batch_size=2, verbose=False, save_checkpoints=True, from_scratch=True)
    batches = [synthetic_batch(2, 64, 128, seed=50 + i) for i in range(3)]
    a = Trainer(cfg)
    a.set_train()
    a.train_step(batches[0])
    a.train_step(batches[1])
    a.save_chkpnt()
    a.train_step(batches[2])
    torch.cuda.synchronize()
    cfg["action"]["from_scratch"] = False
    b = Trainer(cfg)
    b.set_train()
    assert b.model_optimizer._step == 2
    b.train_step(batches[2])
    torch.cuda.synchronize()
    oa, ob = a.model_optimizer, b.model_optimizer
    assert torch.equal(oa.arena().flat, ob.arena().flat)
    assert torch.equal(oa._m, ob._m) and torch.equal(oa._v, ob._v) and oa._step == ob._step == 3
    assert [float(x) for x in a.loss] == [float(x) for x in b.loss]
    for (n, x), (_, y) in zip(a.depth_model.state_dict().items(), b.depth_model.state_dict().items()):
        assert torch.equal(x, y), n          # BatchNorm running statistics and counters included


def test_second_step_uses_updated_weights():
    """Packed weight copies must follow the fused Adam update (arena epoch), and a state_dict round trip must hold."""
    from losses import Losses
    from mcav.optim import FusedAdam
    from oracle.step import synthetic_batch
    hip_d, hip_p, _, _ = build_pair()
    s = synthetic_batch(2, 64, 128, seed=6)
    tgt, refs, K = s["tgt"].to(DEV), [r.to(DEV) for r in s["ref_imgs"]], s["intrinsics"].to(DEV)
    opt = FusedAdam(list(hip_d.parameters()) + list(hip_p.parameters()), 1e-3)
    outs = []
    for _ in range(2):
        opt.zero_grad()
        disps = [hip_d(tgt), hip_d(refs[0])]
        loss = Losses().forward(tgt, refs, disps, hip_p(tgt, refs), K, None)
        sum(loss).backward()
        opt.step()
        outs.append(disps[0][0].detach().clone())
    assert float((outs[0] - outs[1]).abs().max()) > 1e-5
    sd = {k: v.clone() for k, v in hip_d.state_dict().items()}
    from models.depth.resnet_dispnet import DispResNet
    fresh = DispResNet().to(DEV).train()
    fresh.load_state_dict(sd)
    hip_d.eval()
    fresh.eval()
    with torch.no_grad():
        a, b = hip_d(tgt)[0], fresh(tgt)[0]
    assert float((a - b).abs().max()) == 0.0


def test_trainer_runs_synthetic_epoch():
    import yaml
    import os
    from conftest import PKG
    from trainer import Trainer
    cfg = yaml.full_load(open(os.path.join(PKG, "configs", "basic_config.yaml")))
    cfg["datasets"]["augmentation"].update(image_width=128, image_height=64)
    cfg["datasets"]["synthetic_length"] = 10
    cfg["action"].update(batch_size=2, verbose=False, save_checkpoints=False)
    t = Trainer(cfg)
    t.train()
    assert t.step == 4 and torch.isfinite(sum(t.loss)).item()


def test_full_size_step_is_bit_reproducible_and_pair_equals_separate():
    """BASELINE configs[1] size (batch 12, 192x640), properties that need no oracle:
    (1) the whole step -- three HIP streams, slab reductions, no float atomics -- gives bit-identical gradients when repeated from the same
        state (a missing stream dependency would show up here as run-to-run differences);
    (2) the stacked (tgt, ref0) depth passes equal two separate passes: the same loss and gradients to fp32 rounding in train mode (per-pass
        BatchNorm statistics), the same depth maps within the 1e-3 bound in eval mode."""
    from losses import Losses
    from mcav.optim import FusedAdam
    from mcav.streams import Branch
    from oracle.step import synthetic_batch
    hip_d, hip_p, _, _ = build_pair()
    s = synthetic_batch(12, 192, 640, seed=9)
    tgt, refs, K = s["tgt"].to(DEV), [r.to(DEV) for r in s["ref_imgs"]], s["intrinsics"].to(DEV)
    opt = FusedAdam(list(hip_d.parameters()) + list(hip_p.parameters()), 1e-4)
    state = {k: v.clone() for k, v in hip_d.state_dict().items()}
    branch = Branch()

    def step(pair=True):
        hip_d.load_state_dict(state)                       # same BatchNorm running statistics every time
        opt.zero_grad()
        poses = branch.fork(hip_p, tgt, refs)
        disps = list(hip_d.forward_pair(tgt, refs[0])) if pair else [hip_d(tgt), hip_d(refs[0])]
        poses = branch.join(poses)
        loss = Losses().forward(tgt, refs, disps, poses, K, None)
        sum(loss).backward()
        torch.cuda.synchronize()
        return [float(l.detach()) for l in loss], opt.arena().gflat.clone(), [d[0].detach().clone() for d in disps]

    l1, g1, d1 = step()
    l2, g2, d2 = step()
    assert l1 == l2 and torch.equal(g1, g2) and all(torch.equal(a, b) for a, b in zip(d1, d2))
    assert float(g1.abs().max()) > 0 and torch.isfinite(g1).all()
    l3, g3, d3 = step(pair=False)
    for a, b in zip(l1, l3):
        assert abs(a - b) <= 1e-5 * abs(b)
    assert float((g1 - g3).norm() / g3.norm()) < 1e-3
    hip_d.eval()
    with torch.no_grad():
        pa, pb = hip_d.forward_pair(tgt, refs[0])
        sa, sb = hip_d(tgt), hip_d(refs[0])
    for stacked, single in ((pa[0], sa[0]), (pb[0], sb[0])):      # (not bit-equal: twice the rows select other tile shapes / summation orders)
        ds, dn = 1 / (10 * stacked + 0.01), 1 / (10 * single + 0.01)
        # two fp32 evaluations of the same eval-mode network: max within the north_star's 1e-3, mean at rounding level (measured 1.0e-5: the
        # stacked batch takes other tile shapes, the stem kernel walks its tiles in another order)
        assert float(((ds - dn).abs() / dn).max()) < 1e-3 and float(((ds - dn).abs() / dn).mean()) < 3e-5
    hip_d.train()


def test_full_size_forward_and_losses_vs_oracle_on_the_default_mode():
    """BASELINE.json configs[1]'s own shape (batch 12, 192x640, ResNet-18 + PoseNet) on the DEFAULT compute mode (fp32 results; the trunk's 3x3
    stride-1 convolutions on split operands) against the CPU oracle's forward on identical inputs and weights: depth
    maps of BOTH passes within north_star's 1e-3 max-relative with AbsRel < 1e-4, poses to 1e-4, both loss scalars to 1e-4 relative.  (The
    oracle's forward of this size takes ~15 s of host time; its backward is what the smaller step cases and the float64 arbiter cover.)"""
    from losses import Losses
    from oracle.step import process_batch, synthetic_batch
    hip_d, hip_p, ref_d, ref_p = build_pair()
    s = synthetic_batch(12, 192, 640, seed=23)
    with torch.no_grad():
        (rdisps, rposes), rloss = process_batch(ref_d, ref_p, s)
    tgt, refs, K = s["tgt"].to(DEV), [r.to(DEV) for r in s["ref_imgs"]], s["intrinsics"].to(DEV)
    with torch.no_grad():
        disps = list(hip_d.forward_pair(tgt, refs[0]))
        poses = hip_p(tgt, refs)
        loss = Losses().forward(tgt, refs, disps, poses, K, None)
    for got, want in zip(disps, rdisps):
        dg, dw = 1 / (10 * got[0].cpu().double() + 0.01), 1 / (10 * want[0].double() + 0.01)
        rel = (dg - dw).abs() / dw
        print("12x192x640 depth vs oracle: max-rel %.3e AbsRel %.3e" % (float(rel.max()), float(rel.mean())))
        assert float(rel.max()) < 1e-3 and float(rel.mean()) < 1e-4
    assert rel_err(poses, rposes) < 1e-3
    for a, b in zip(loss, rloss):
        assert abs(float(a) - float(b)) < 1e-4 * abs(float(b)), (float(a), float(b))


def test_trainer_validate_runs():
    """The validation loop of the reference (trainer.py:315-337, never called there and broken in compute_errors) runs on the GPU metrics kernel."""
    import yaml
    import os
    from conftest import PKG
    from trainer import Trainer
    cfg = yaml.full_load(open(os.path.join(PKG, "configs", "basic_config.yaml")))
    cfg["datasets"]["augmentation"].update(image_width=128, image_height=64)
    cfg["datasets"]["synthetic_length"] = 20       # validation split: 20 %
    cfg["action"].update(batch_size=2, verbose=False, save_checkpoints=False)
    t = Trainer(cfg)
    acc = t.validate()
    assert acc is not None and set(("abs_rel", "rms", "d1", "silog", "count")) <= set(acc)
