"""models/pose/pose_net.py -- drop-in for the reference PoseNet (models/pose/pose_net.py:31-77) on MI355X.

Same constructor, `init_weights()` (xavier_uniform + zero bias) and `__call__(tgt, [ref0, ref1]) -> [B,2,6]`
(axis-angle, translation); parameters named convN.0.weight/bias and pose_pred.weight/bias as in the reference.
"""
import torch
import torch.nn as nn

from mcav import lib as L
from mcav import posenet as E
from mcav.holders import ConvParams


def conv_gn(in_planes, out_planes, kernel_size=3):
    """Reference helper name kept: conv(stride 2) + ReLU (its GroupNorm is commented out, pose_net.py:27)."""
    return nn.Sequential(ConvParams(in_planes, out_planes, kernel_size))


class _PoseNetFn(torch.autograd.Function):
    """Under a two-graph capture (mcav/graph.py, streams.DUAL) both directions run on the side stream, switched to INSIDE the node: autograd
    then sees the node on the main stream and adds no cross-stream event of its own."""

    @staticmethod
    def forward(ctx, mod, tgt, ref0, ref1, *params):
        from mcav import streams
        imgs = [L.dev(t.contiguous(), "image") for t in (tgt, ref0, ref1)]
        dual = streams.DUAL
        if dual is not None:
            dual.fork(torch.cuda.current_stream())            # side chain: behind the inputs and the re-packed filters
            with torch.cuda.stream(dual.stream):
                out, saved = E.forward(mod, imgs[0], imgs[1:])
        else:
            out, saved = E.forward(mod, imgs[0], imgs[1:])
        ctx.mod, ctx.saved = mod, saved
        return out.view(out.shape[0], mod.nb_ref_imgs, 6)

    @staticmethod
    def backward(ctx, g):
        from mcav import nn as N
        from mcav import streams
        g = L.dev(g.contiguous(), "grad")
        dual = streams.DUAL
        if dual is not None and N.WGRAD_SIDE.dual is dual:
            main = torch.cuda.current_stream()
            if N.WGRAD_SIDE._note(main):                      # (the end-of-backward callback records the side chain's join event)
                N.WGRAD_SIDE.stream, N.WGRAD_SIDE.forked = dual.stream, True
                N.WGRAD_SIDE.keep.append((g,))
                dual.fork(main)                                # side chain: behind the loss kernel's pose gradient
                with torch.cuda.stream(dual.stream):
                    E.backward(ctx.mod, ctx.saved, g.view(g.shape[0], -1))
                ctx.saved = None
                return (None,) * (4 + len(list(ctx.mod.parameters())))
        E.backward(ctx.mod, ctx.saved, g.view(g.shape[0], -1))
        ctx.saved = None
        return (None,) * (4 + len(list(ctx.mod.parameters())))


class PoseNet(nn.Module):
    def __init__(self, nb_ref_imgs=2, rotation_mode='euler', **kwargs):
        super().__init__()
        if nb_ref_imgs != 2:
            raise NotImplementedError("PoseNet: two reference images (the reference's configuration) only")
        self.nb_ref_imgs = nb_ref_imgs
        self.rotation_mode = rotation_mode
        ch = [16, 32, 64, 128, 256, 256, 256]
        cin = 3 * (1 + nb_ref_imgs)
        for i, (c, k) in enumerate(zip(ch, E.KS)):
            setattr(self, "conv%d" % (i + 1), conv_gn(cin, c, k))
            cin = c
        self.pose_pred = ConvParams(cin, 6 * nb_ref_imgs, 1)
        if kwargs.get("dtype") is not None:          # opt-in bf16 MFMA tiles (mcav.nn.set_compute_dtype); default fp32 as the reference
            from mcav import nn as N
            N.set_compute_dtype(self, kwargs["dtype"])

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, ConvParams):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()

    def forward(self, image, context):
        assert len(context) == self.nb_ref_imgs
        return _PoseNetFn.apply(self, image, context[0], context[1], *self.parameters())
