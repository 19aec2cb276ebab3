"""Oracle (test infrastructure): the depth / pose networks in stock torch.nn, CPU fp32.

Restates (module nesting chosen so that state_dict keys equal the reference's):
  * torchvision ResNet-18/34/50/101/152 trunk (third party, NOT in /root/reference; call sites
    models/depth/resnet_dispnet.py:20-30,38-44) -- published architecture, torchvision key names
  * ResnetEncoder / DepthDecoder / DispResNet          models/depth/resnet_dispnet.py:12-107
  * ConvBlock / Conv3x3 / nearest upsample             models/depth/layers.py:22-58
  * PoseNet                                            models/pose/pose_net.py:31-77
  * PoseFc                                             models/pose/pose_fc.py:21-84
  * DispNetS                                           models/depth/disp_net.py:51-141
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- ResNet trunk
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)   # stride on the 3x3 (v1.5)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


RESNET_SPECS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
                101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


class ResNet(nn.Module):
    def __init__(self, num_layers=18, num_classes=1000):
        super().__init__()
        block, counts = RESNET_SPECS[num_layers]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(block, 64, counts[0], 1)
        self.layer2 = self._stage(block, 128, counts[1], 2)
        self.layer3 = self._stage(block, 256, counts[2], 2)
        self.layer4 = self._stage(block, 512, counts[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, block, planes, n, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                               nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*layers)


def resnet_factory(num_layers):
    """Signature-compatible stand-in for torchvision.models.resnetNN(pretrained) (weights: random init)."""
    def make(pretrained=False, **kw):
        return ResNet(num_layers)
    return make


# ----------------------------------------------------------------------------- depth net
class ResnetEncoder(nn.Module):
    def __init__(self, num_layers=18):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNet(num_layers)
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4

    def forward(self, x):
        e = self.encoder
        f0 = e.relu(e.bn1(e.conv1(x)))
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        return [f0, f1, f2, f3, f4]


class Conv3x3(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1)
        self.conv = nn.Conv2d(int(cin), int(cout), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = Conv3x3(cin, cout)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.nonlin(self.conv(x))


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4)):
        super().__init__()
        self.scales = list(scales)
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.index = {}
        mods = []
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.index[("upconv", i, 0)] = len(mods)
            mods.append(ConvBlock(cin, self.num_ch_dec[i]))
            cin = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if i > 0 else 0)
            self.index[("upconv", i, 1)] = len(mods)
            mods.append(ConvBlock(cin, self.num_ch_dec[i]))
        for s in self.scales:
            self.index[("dispconv", s)] = len(mods)
            mods.append(Conv3x3(self.num_ch_dec[s], 1))
        self.decoder = nn.ModuleList(mods)      # positional keys 0..13, as the reference registers them

    def conv(self, *key):
        return self.decoder[self.index[key]]

    def forward(self, feats):
        out = {}
        x = feats[-1]
        for i in range(4, -1, -1):
            x = self.conv("upconv", i, 0)(x)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            if i > 0:
                x = torch.cat([x, feats[i - 1]], 1)
            x = self.conv("upconv", i, 1)(x)
            if i in self.scales:
                out[("disp", i)] = torch.sigmoid(self.conv("dispconv", i)(x))
        return out


class DispResNet(nn.Module):
    def __init__(self, num_layers=18):
        super().__init__()
        self.encoder = ResnetEncoder(num_layers)
        self.decoder = DepthDecoder(self.encoder.num_ch_enc)

    def forward(self, x):
        return [self.decoder(self.encoder(x))[("disp", 0)]]


# ----------------------------------------------------------------------------- pose nets
def _conv_relu(cin, cout, k):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=2, padding=(k - 1) // 2), nn.ReLU(inplace=True))


def _xavier_convs(module):
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.xavier_uniform_(m.weight.data)
            if m.bias is not None:
                m.bias.data.zero_()


class PoseNet(nn.Module):
    CH = [16, 32, 64, 128, 256, 256, 256]
    KS = [7, 5, 3, 3, 3, 3, 3]

    def __init__(self, nb_ref_imgs=2):
        super().__init__()
        self.nb_ref_imgs = nb_ref_imgs
        cin = 3 * (1 + nb_ref_imgs)
        for i, (c, k) in enumerate(zip(self.CH, self.KS)):
            setattr(self, "conv%d" % (i + 1), _conv_relu(cin, c, k))
            cin = c
        self.pose_pred = nn.Conv2d(cin, 6 * nb_ref_imgs, 1)

    def init_weights(self):
        _xavier_convs(self)

    def trunk(self, tgt, refs):
        x = torch.cat([tgt] + list(refs), 1)
        for i in range(7):
            x = getattr(self, "conv%d" % (i + 1))(x)
        return self.pose_pred(x)

    def forward(self, tgt, refs):
        p = self.trunk(tgt, refs)
        p = p.mean(3).mean(2)
        return 0.06 * p.view(p.size(0), self.nb_ref_imgs, 6)


class PoseFc(PoseNet):
    """Same trunk, MLP head 360->128->32->12 (only valid at 384x1280), rotation zeroed."""

    def __init__(self, nb_ref_imgs=2):
        super().__init__(nb_ref_imgs)
        self.fc_loc = nn.Sequential(nn.Linear(12 * 3 * 10, 128), nn.ReLU(True), nn.Linear(128, 32), nn.ReLU(True),
                                    nn.Linear(32, 12))
        self.init_weights()

    def init_weights(self):
        _xavier_convs(self)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)
        self.fc_loc[-1].weight.data.zero_()

    def forward(self, tgt, refs):
        p = self.trunk(tgt, refs).view(-1, 12 * 3 * 10)
        p = self.fc_loc(p).view(-1, self.nb_ref_imgs, 6)
        mask = torch.ones_like(p)
        mask[:, :, :3] = 0
        return p * mask


# ----------------------------------------------------------------------------- DispNetS
def _down(cin, cout, k=3):
    p = (k - 1) // 2
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=2, padding=p), nn.ReLU(inplace=True), nn.BatchNorm2d(cout),
                         nn.Conv2d(cout, cout, k, padding=p), nn.ReLU(inplace=True))


def _up(cin, cout):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1), nn.ReLU(inplace=True))


def _iconv(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.ReLU(inplace=True))


def _pred(cin):
    return nn.Sequential(nn.Conv2d(cin, 1, 3, padding=1), nn.Sigmoid())


def _crop(x, ref):
    return x[:, :, :ref.size(2), :ref.size(3)]


class DispNetS(nn.Module):
    def __init__(self, alpha=10, beta=0.01):
        super().__init__()
        self.alpha, self.beta = alpha, beta
        c = [32, 64, 128, 256, 512, 512, 512]
        u = [512, 512, 256, 128, 64, 32, 16]
        ks = [7, 5, 3, 3, 3, 3, 3]
        cin = 3
        for i in range(7):
            setattr(self, "conv%d" % (i + 1), _down(cin, c[i], ks[i]))
            cin = c[i]
        ins = [c[6]] + u[:6]
        for i in range(7):
            setattr(self, "upconv%d" % (7 - i), _up(ins[i], u[i]))
        self.iconv7 = _iconv(u[0] + c[5], u[0])
        self.iconv6 = _iconv(u[1] + c[4], u[1])
        self.iconv5 = _iconv(u[2] + c[3], u[2])
        self.iconv4 = _iconv(u[3] + c[2], u[3])
        self.iconv3 = _iconv(1 + u[4] + c[1], u[4])
        self.iconv2 = _iconv(1 + u[5] + c[0], u[5])
        self.iconv1 = _iconv(1 + u[6], u[6])
        self.predict_disp4 = _pred(u[3])
        self.predict_disp3 = _pred(u[4])
        self.predict_disp2 = _pred(u[5])
        self.predict_disp1 = _pred(u[6])

    def init_weights(self):
        _xavier_convs(self)

    def forward(self, x):
        o = [None]
        h = x
        for i in range(1, 8):
            h = getattr(self, "conv%d" % i)(h)
            o.append(h)
        i7 = self.iconv7(torch.cat((_crop(self.upconv7(o[7]), o[6]), o[6]), 1))
        i6 = self.iconv6(torch.cat((_crop(self.upconv6(i7), o[5]), o[5]), 1))
        i5 = self.iconv5(torch.cat((_crop(self.upconv5(i6), o[4]), o[4]), 1))
        i4 = self.iconv4(torch.cat((_crop(self.upconv4(i5), o[3]), o[3]), 1))
        d4 = self.alpha * self.predict_disp4(i4) + self.beta
        up = lambda d, ref: _crop(F.interpolate(d, scale_factor=2, mode="bilinear", align_corners=False), ref)
        i3 = self.iconv3(torch.cat((_crop(self.upconv3(i4), o[2]), o[2], up(d4, o[2])), 1))
        d3 = self.alpha * self.predict_disp3(i3) + self.beta
        i2 = self.iconv2(torch.cat((_crop(self.upconv2(i3), o[1]), o[1], up(d3, o[1])), 1))
        d2 = self.alpha * self.predict_disp2(i2) + self.beta
        i1 = self.iconv1(torch.cat((_crop(self.upconv1(i2), x), up(d2, x)), 1))
        d1 = self.alpha * self.predict_disp1(i1) + self.beta
        return d1, d2, d3, d4
