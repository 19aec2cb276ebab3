"""Forward / backward engine of the pose network (reference models/pose/pose_net.py:31-77) on NHWC tensors.

cat(tgt, ref0, ref1) is packed once into a 16-channel NHWC buffer (9 real channels); seven stride-2 conv + bias + ReLU
launches (activation fused into the conv epilogue), the 1x1 pose head, spatial mean, x 0.06.  Backward: every dgrad
multiplies by the ReLU mask of the layer it feeds in its epilogue; stride-2 dgrads use parity-class tiles.
"""
from . import nn as N
from .depthnet import spec_of, hw

KS = (7, 5, 3, 3, 3, 3, 3)


def pack_inputs(tgt, refs):
    if len(refs) == 2:
        return N.nchw3_to_nhwc(N.L.dev(tgt.contiguous(), "tgt"), N.L.dev(refs[0].contiguous(), "ref"), N.L.dev(refs[1].contiguous(), "ref"), 16)
    buf = None
    for i, img in enumerate([tgt] + list(refs)):
        buf = N.nchw_to_nhwc(img, 16, buf, 3 * i)
    return buf


def conv_specs(net):
    return [spec_of(getattr(net, "conv%d" % (i + 1))[0], 2, (k - 1) // 2, N.PAD_ZERO) for i, k in enumerate(KS)]


def trunk_forward(net, x0):
    acts = [x0]
    for spec in conv_specs(net):
        acts.append(N.conv_fwd(spec, acts[-1], act=N.ACT_RELU))
    head = spec_of(net.pose_pred, 1, 0, N.PAD_ZERO)
    return acts, N.conv_fwd(head, acts[-1])


def trunk_backward(net, acts, dp):
    """dp: gradient at the pose head output [B,h,w,12]."""
    head = spec_of(net.pose_pred, 1, 0, N.PAD_ZERO)
    N.conv_wgrad(head, acts[-1], dp)
    dpre = N.conv_dgrad(head, dp, hw(acts[-1]), dact_aux=acts[-1], dact=N.ACT_RELU)
    specs = conv_specs(net)
    for i in range(6, -1, -1):
        N.conv_wgrad(specs[i], acts[i], dpre)
        if i > 0:
            dpre = N.conv_dgrad(specs[i], dpre, hw(acts[i]), dact_aux=acts[i], dact=N.ACT_RELU)


def forward(net, tgt, refs):
    acts, p = trunk_forward(net, pack_inputs(tgt, refs))
    return N.spatial_mean(p, 0.06), (acts, tuple(p.shape))


def backward(net, saved, dout):
    acts, pshape = saved
    trunk_backward(net, acts, N.spatial_mean_bwd(dout, pshape, 0.06))
