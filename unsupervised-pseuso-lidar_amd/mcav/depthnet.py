"""Forward / backward engines of the depth network (ResNet encoder + monodepth2-style decoder) on NHWC tensors.

Restates what the reference gets from torchvision's resnet18/34 (BasicBlock; models/depth/resnet_dispnet.py:20-46)
and its DepthDecoder (resnet_dispnet.py:48-96, layers.py:22-58) as explicit sequences of HIP launches:
  conv (+ per-tile BatchNorm statistics in the epilogue) -> bn_finalize -> bn_apply(+ReLU, +residual)
  decoder: reflection-padded 3x3 conv + bias + ELU, with the nearest x2 upsample and the skip concat fused into the
  A-operand gather of the second conv of every level; dispconv + sigmoid.
Backward is hand-scheduled: every activation derivative, the 2x2 sum-pool (adjoint of the upsample), the concat split
and the branch accumulations are fused into dgrad epilogues; parameter gradients accumulate straight into .grad.
"""
import torch

from . import nn as N


def spec_of(holder, stride, pad, pad_mode, smallc=False):
    s = getattr(holder, "_mcav_spec", None)
    if s is None or s.weight is not holder.weight:
        s = N.ConvSpec(holder.weight, holder.bias, stride, pad, pad_mode, smallc)
        s.mma = getattr(holder, "_mcav_mma", N.DEFAULT_MMA)   # set by mcav.nn.set_compute_dtype
        holder._mcav_spec = s
    return s


def hw(t):
    return (t.shape[1], t.shape[2])


# ------------------------------------------------------------------------------------------------ BatchNorm'd conv
def conv_bn(conv, bn, x, stride, pad, train, relu, residual=None, smallc=False, groups=1):
    """-> (raw conv output, BN state, activated output).  groups: independent passes stacked along the batch (per-pass statistics)."""
    spec = spec_of(conv, stride, pad, N.PAD_ZERO, smallc)
    if train:
        raw, slab = N.conv_fwd(spec, x, stats=True, groups=groups)
        st = N.bn_train_coeffs(bn, slab, raw.shape[0] * raw.shape[1] * raw.shape[2] // groups, groups)
    else:
        raw = N.conv_fwd(spec, x)
        st = N.bn_eval_coeffs(bn)
    return raw, st, N.bn_apply(raw, st, relu, residual)


# ------------------------------------------------------------------------------------------------ BasicBlock
def block_forward(blk, x, stride, train, groups=1):
    sv = {"x": x, "stride": stride}
    sv["r1"], sv["st1"], sv["h1"] = conv_bn(blk.conv1, blk.bn1, x, stride, 1, train, True, groups=groups)
    if blk.downsample is not None:
        sv["rd"], sv["std"], idt = conv_bn(blk.downsample[0], blk.downsample[1], x, stride, 0, train, False, groups=groups)
    else:
        idt = x
    sv["r2"], sv["st2"], sv["out"] = conv_bn(blk.conv2, blk.bn2, sv["h1"], 1, 1, train, True, residual=idt, groups=groups)
    return sv["out"], sv


def block_backward(blk, sv, dout, addend=None, fused_in=None, consumer=None):
    """fused_in: (slab, tiles per group) when `dout` already carries this block's bn2 ReLU mask and partial sums (produced by the previous
    call's last data gradient).  consumer: saved tensors of the block that will receive this call's result, when that gradient may carry
    ITS last BatchNorm's mask and partial sums (same stage, stride-1 first conv here) -> returns (dx, (slab, tiles per group))."""
    x, stride = sv["x"], sv["stride"]
    c1 = spec_of(blk.conv1, stride, 1, N.PAD_ZERO)
    c2 = spec_of(blk.conv2, 1, 1, N.PAD_ZERO)
    dr2, dz = N.bn_backward(blk.bn2, sv["st2"], dout, sv["out"], sv["r2"], True, want_dres=True, fused=fused_in)
    N.conv_wgrad(c2, sv["h1"], dr2)
    if N.can_fuse_bn_stats(c2) and sv["st1"].mean is not None:
        # dh1 arrives masked by ReLU'(h1) with bn1's backward partial sums in the epilogue's slab: no reduce pass over dh1 / h1 / r1
        dh1, slab, mtg = N.conv_dgrad(c2, dr2, hw(sv["h1"]), dact_aux=sv["h1"], dact=N.ACT_RELU, bn_stats=(sv["r1"], sv["st1"]))
        dr1 = N.bn_backward(blk.bn1, sv["st1"], dh1, None, sv["r1"], True, fused=(slab, mtg))
    else:
        dh1 = N.conv_dgrad(c2, dr2, hw(sv["h1"]))
        dr1 = N.bn_backward(blk.bn1, sv["st1"], dh1, sv["h1"], sv["r1"], True)
    N.conv_wgrad(c1, x, dr1)
    out_stats = None
    kw = {}
    if consumer is not None and N.can_fuse_bn_stats(c1) and consumer["st2"].mean is not None:
        # the consumer's output ReLU masks the SUM of this gradient and the residual branch's (the addend)
        kw = dict(dact_aux=consumer["out"], dact=N.ACT_RELU | N.DACT_AFTER_ADDEND, bn_stats=(consumer["r2"], consumer["st2"]))
    if blk.downsample is None:
        if addend is not None:
            dz = N.add(dz, addend)
        res = N.conv_dgrad(c1, dr1, hw(x), addend=dz, **kw)
    else:
        cd = spec_of(blk.downsample[0], stride, 0, N.PAD_ZERO)
        drd = N.bn_backward(blk.downsample[1], sv["std"], dz, None, sv["rd"], False)
        N.conv_wgrad(cd, x, drd)
        dxd = N.conv_dgrad(cd, drd, hw(x), addend=addend)
        res = N.conv_dgrad(c1, dr1, hw(x), addend=dxd, **kw)
    if kw:
        return res[0], (res[1], res[2])
    return res, None


# ------------------------------------------------------------------------------------------------ Bottleneck (ResNet-50/101/152)
def bottleneck_forward(blk, x, stride, train, groups=1):
    """torchvision Bottleneck (v1.5: the stride sits on the 3x3): 1x1 -> 3x3(stride) -> 1x1(x4), BN after each, residual, ReLU."""
    sv = {"x": x, "stride": stride}
    sv["r1"], sv["st1"], sv["h1"] = conv_bn(blk.conv1, blk.bn1, x, 1, 0, train, True, groups=groups)
    sv["r2"], sv["st2"], sv["h2"] = conv_bn(blk.conv2, blk.bn2, sv["h1"], stride, 1, train, True, groups=groups)
    if blk.downsample is not None:
        sv["rd"], sv["std"], idt = conv_bn(blk.downsample[0], blk.downsample[1], x, stride, 0, train, False, groups=groups)
    else:
        idt = x
    sv["r3"], sv["st3"], sv["out"] = conv_bn(blk.conv3, blk.bn3, sv["h2"], 1, 0, train, True, residual=idt, groups=groups)
    return sv["out"], sv


def bottleneck_backward(blk, sv, dout, addend=None, fused_in=None, consumer=None):
    """As block_backward; the BatchNorms whose incoming gradient a stride-1 data gradient produces (bn2 after the 1x1 conv3, bn1 after a
    stride-1 conv2, the consumer block's bn3 after this block's 1x1 conv1) take their backward partial sums from that launch's epilogue."""
    x, stride = sv["x"], sv["stride"]
    c1 = spec_of(blk.conv1, 1, 0, N.PAD_ZERO)
    c2 = spec_of(blk.conv2, stride, 1, N.PAD_ZERO)
    c3 = spec_of(blk.conv3, 1, 0, N.PAD_ZERO)
    dr3, dz = N.bn_backward(blk.bn3, sv["st3"], dout, sv["out"], sv["r3"], True, want_dres=True, fused=fused_in)
    N.conv_wgrad(c3, sv["h2"], dr3)
    if N.can_fuse_bn_stats(c3) and sv["st2"].mean is not None:
        dh2, slab, mtg = N.conv_dgrad(c3, dr3, hw(sv["h2"]), dact_aux=sv["h2"], dact=N.ACT_RELU, bn_stats=(sv["r2"], sv["st2"]))
        dr2 = N.bn_backward(blk.bn2, sv["st2"], dh2, None, sv["r2"], True, fused=(slab, mtg))
    else:
        dh2 = N.conv_dgrad(c3, dr3, hw(sv["h2"]))
        dr2 = N.bn_backward(blk.bn2, sv["st2"], dh2, sv["h2"], sv["r2"], True)
    N.conv_wgrad(c2, sv["h1"], dr2)
    if N.can_fuse_bn_stats(c2) and sv["st1"].mean is not None:
        dh1, slab, mtg = N.conv_dgrad(c2, dr2, hw(sv["h1"]), dact_aux=sv["h1"], dact=N.ACT_RELU, bn_stats=(sv["r1"], sv["st1"]))
        dr1 = N.bn_backward(blk.bn1, sv["st1"], dh1, None, sv["r1"], True, fused=(slab, mtg))
    else:
        dh1 = N.conv_dgrad(c2, dr2, hw(sv["h1"]))
        dr1 = N.bn_backward(blk.bn1, sv["st1"], dh1, sv["h1"], sv["r1"], True)
    N.conv_wgrad(c1, x, dr1)
    kw = {}
    if consumer is not None and N.can_fuse_bn_stats(c1) and consumer["st3"].mean is not None:
        kw = dict(dact_aux=consumer["out"], dact=N.ACT_RELU | N.DACT_AFTER_ADDEND, bn_stats=(consumer["r3"], consumer["st3"]))
    if blk.downsample is None:
        if addend is not None:
            dz = N.add(dz, addend)
        res = N.conv_dgrad(c1, dr1, hw(x), addend=dz, **kw)
    else:
        cd = spec_of(blk.downsample[0], stride, 0, N.PAD_ZERO)
        drd = N.bn_backward(blk.downsample[1], sv["std"], dz, None, sv["rd"], False)
        N.conv_wgrad(cd, x, drd)
        dxd = N.conv_dgrad(cd, drd, hw(x), addend=addend)
        res = N.conv_dgrad(c1, dr1, hw(x), addend=dxd, **kw)
    if kw:
        return res[0], (res[1], res[2])
    return res, None


# ------------------------------------------------------------------------------------------------ encoder
STAGES = ("layer1", "layer2", "layer3", "layer4")


def encoder_forward(net, x4, train, groups=1):
    """net: ResNetParams holder; x4: NHWC4 image (groups passes stacked along the batch).  -> ([f0..f4], saved)"""
    sv = {"x4": x4}
    sv["c1"], sv["st"], f0 = conv_bn(net.conv1, net.bn1, x4, 2, 3, train, True, smallc=True, groups=groups)
    p0, sv["idx"] = N.maxpool_fwd(f0)
    feats, blocks = [f0], []
    x = p0
    for si, name in enumerate(STAGES):
        stage_sv = []
        for bi, blk in enumerate(getattr(net, name)):
            fwd = bottleneck_forward if hasattr(blk, "conv3") else block_forward
            x, bsv = fwd(blk, x, 2 if (si > 0 and bi == 0) else 1, train, groups)
            stage_sv.append(bsv)
        blocks.append(stage_sv)
        feats.append(x)
    sv["blocks"], sv["feats"] = blocks, feats
    N.flush_bn_counters()
    return feats, sv


def encoder_backward(net, sv, dfeats, complete=False):
    """dfeats[i]: gradient w.r.t. feature i arriving from outside (the decoder); consumed (may be overwritten).
    complete: this call produces the parameters' whole gradient (stacked passes): stages are announced to N.grads_ready."""
    feats = sv["feats"]
    dcur = dfeats[4]
    fused = None            # (slab, tiles per group) when dcur already carries the next block's output mask and BatchNorm partial sums
    for si in range(3, -1, -1):
        blks = list(getattr(net, STAGES[si]))
        for bi in range(len(blks) - 1, -1, -1):
            addend = dfeats[si] if (bi == 0 and si > 0) else None      # skip-connection gradient of this stage's input
            bwd = bottleneck_backward if hasattr(blks[bi], "conv3") else block_backward
            # the block that receives this call's result (same stage: nothing else adds to that gradient) may take it masked and summed
            consumer = sv["blocks"][si][bi - 1] if bi > 0 else None
            dcur, fused = bwd(blks[bi], sv["blocks"][si][bi], dcur, addend, fused_in=fused, consumer=consumer)
        if complete and N.GRADS_READY is not None:       # layer4 holds half of all parameters and finishes first: its all-reduce hides behind layers 3..1
            from .dist import announced_stages
            if STAGES[si] in announced_stages():
                N.grads_ready(getattr(net, STAGES[si]).parameters())
    # dcur = gradient w.r.t. the max-pool output; f0 also feeds the decoder
    df0 = N.maxpool_bwd(dcur, sv["idx"], tuple(feats[0].shape), dx=dfeats[0], accumulate=True)
    dc1 = N.bn_backward(net.bn1, sv["st"], df0, feats[0], sv["c1"], True)
    N.conv_wgrad(spec_of(net.conv1, 2, 3, N.PAD_ZERO, True), sv["x4"], dc1)


# ------------------------------------------------------------------------------------------------ decoder
def dec_spec(holder):
    return spec_of(holder, 1, 1, N.PAD_REFLECT)


def decoder_forward(dec, feats, scales=(0,)):
    """dec: DepthDecoder module (mcav holders, .conv(kind, i, j)); feats NHWC.  -> ({scale: disp NHWC [B,h,w,1]}, saved)"""
    sv = {"feats": feats, "a": {}, "b": {}, "disp": {}, "scales": tuple(scales)}
    x = feats[4]
    for i in range(4, -1, -1):
        a = N.conv_fwd(dec_spec(dec.conv("upconv", i, 0).conv.conv), x, act=N.ACT_ELU)
        b = N.conv_fwd(dec_spec(dec.conv("upconv", i, 1).conv.conv), a, feats[i - 1] if i > 0 else None, up1=True, act=N.ACT_ELU)
        sv["a"][i], sv["b"][i] = a, b
        if i in scales:
            sd = dec_spec(dec.conv("dispconv", i).conv)
            sv["disp"][i] = N.conv3x3r_c1_fwd(sd, b, N.ACT_SIGMOID) if N.narrow_ok(sd, b) else N.conv_fwd(sd, b, act=N.ACT_SIGMOID)
        x = b
    return sv["disp"], sv


def decoder_backward(dec, sv, ddisp, need_feature_grads=True):
    """ddisp: {scale: gradient NHWC [B,h,w,1]}.  -> [df0..df4] (gradients w.r.t. the encoder features)."""
    feats = sv["feats"]
    dfeats = [None] * 5
    dpre_b = None          # gradient at the pre-activation output of conv (i, 1), accumulated level by level
    for i in range(0, 5):
        a, b = sv["a"][i], sv["b"][i]
        s11 = dec_spec(dec.conv("upconv", i, 1).conv.conv)
        s10 = dec_spec(dec.conv("upconv", i, 0).conv.conv)
        if i in ddisp and ddisp[i] is not None:
            sd = dec_spec(dec.conv("dispconv", i).conv)
            if N.narrow_ok(sd, b):
                # one pass over b_i: sigmoid', the head's weight / bias gradients, and d b_i through ELU'(b_i) (+ what came from level i-1)
                dpre_b = N.conv3x3r_c1_bwd(sd, b, ddisp[i], sv["disp"][i], N.ACT_SIGMOID, N.ACT_ELU, addend=dpre_b)
            else:
                # sigmoid'(disp) * d disp, stored as channel 0 of a zeroed 4-channel map so the 16-byte gathers apply
                dpre_d = N.act_bwd_padded(ddisp[i], sv["disp"][i], N.ACT_SIGMOID, 4)
                N.conv_wgrad(sd, b, dpre_d)
                # d b_i from the disparity head, through ELU'(b_i); joins what came from level i-1 (addend)
                dpre_b = N.conv_dgrad(sd, dpre_d, hw(b), dact_aux=b, dact=N.ACT_ELU, addend=dpre_b)
        if dpre_b is None:
            raise RuntimeError("decoder_backward: no gradient reaches level %d" % i)
        skip = feats[i - 1] if i > 0 else None
        N.conv_wgrad(s11, a, dpre_b, x2=skip, up1=True)
        c1 = a.shape[3]
        # channels [0, c1): adjoint of the nearest upsample (2x2 sum) then ELU'(a_i) -> pre-activation gradient of conv (i, 0)
        dpre_a = N.conv_dgrad(s11, dpre_b, hw(b), n_begin=0, n_count=c1, dact_aux=a, dact=N.ACT_ELU, pool=True)
        if i > 0 and need_feature_grads:
            dfeats[i - 1] = N.conv_dgrad(s11, dpre_b, hw(b), n_begin=c1, n_count=skip.shape[3])
        x_in = sv["b"][i + 1] if i < 4 else feats[4]
        N.conv_wgrad(s10, x_in, dpre_a)
        if i < 4:
            dpre_b = N.conv_dgrad(s10, dpre_a, hw(x_in), dact_aux=x_in, dact=N.ACT_ELU)
        elif need_feature_grads:
            dfeats[4] = N.conv_dgrad(s10, dpre_a, hw(x_in))
    return dfeats
