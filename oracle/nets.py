"""fp32 PyTorch-CPU restatement of the SRN networks (TEST INFRASTRUCTURE, see oracle/__init__.py).

Every class cites the reference file:line it follows.  ``state_dict`` key names
and tensor shapes are identical to the reference's so that
checkpoints are interchangeable and per-key golden digests can be compared.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

LRELU_SLOPE = 0.2  # reference: block.py:10-23 (act 'leakyrelu', neg_slope=0.2)


def _conv(cin, cout, k=3, stride=1, pad=1, bias=True):
    return nn.Conv2d(cin, cout, k, stride, pad, bias=bias)


def _conv_act(cin, cout, act=True):
    """conv_block(CNA, norm=None, pad='zero'): Conv3x3 [+ LeakyReLU(0.2)] (block.py:130-156)."""
    mods = [_conv(cin, cout)]
    if act:
        mods.append(nn.LeakyReLU(LRELU_SLOPE, inplace=False))
    return nn.Sequential(*mods)


class RDB5C(nn.Module):
    """ResidualDenseBlock_5C (block.py:254-286): 5 dense convs, out = x5*0.2 + x."""

    def __init__(self, nc=64, gc=32):
        super().__init__()
        self.conv1 = _conv_act(nc, gc)
        self.conv2 = _conv_act(nc + gc, gc)
        self.conv3 = _conv_act(nc + 2 * gc, gc)
        self.conv4 = _conv_act(nc + 3 * gc, gc)
        self.conv5 = _conv_act(nc + 4 * gc, nc, act=False)  # CNA mode -> no last act (block.py:273-278)

    def forward(self, x):
        x1 = self.conv1(x)
        x2 = self.conv2(torch.cat((x, x1), 1))
        x3 = self.conv3(torch.cat((x, x1, x2), 1))
        x4 = self.conv4(torch.cat((x, x1, x2, x3), 1))
        x5 = self.conv5(torch.cat((x, x1, x2, x3, x4), 1))
        return x5 * 0.2 + x


class RRDB(nn.Module):
    """RRDB (block.py:289-309): three RDBs, out*0.2 + x."""

    def __init__(self, nc=64, gc=32):
        super().__init__()
        self.RDB1 = RDB5C(nc, gc)
        self.RDB2 = RDB5C(nc, gc)
        self.RDB3 = RDB5C(nc, gc)

    def forward(self, x):
        return self.RDB3(self.RDB2(self.RDB1(x))) * 0.2 + x


class _Shortcut(nn.Module):
    """ShortcutBlock (block.py:97-105): x + sub(x)."""

    def __init__(self, sub):
        super().__init__()
        self.sub = sub

    def forward(self, x):
        return x + self.sub(x)


class RRDBNet(nn.Module):
    """RRDBNet (architecture.py:174-205).  gc is hard-wired to 32 (architecture.py:183).

    upsample_mode 'upconv' (what define_G selects, networks.py:96-99):
      nearest x2 -> conv3x3 -> lrelu, twice; 'pixelshuffle' (block.py:838-851):
      conv nf->4nf -> PixelShuffle(2) -> lrelu.
    The flattened nn.Sequential index layout reproduces the reference keys
    (model.0, model.1.sub.{i}, model.3, model.6, model.8, model.10).
    """

    def __init__(self, in_nc=3, out_nc=3, nf=64, nb=23, upscale=4, upsample_mode='upconv'):
        super().__init__()
        n_up = 1 if upscale == 3 else int(math.log(upscale, 2))
        trunk = [RRDB(nf, 32) for _ in range(nb)] + [_conv(nf, nf)]  # LR_conv has no act
        mods = [_conv(in_nc, nf), _Shortcut(nn.Sequential(*trunk))]
        for _ in range(n_up):
            if upsample_mode == 'upconv':
                mods += [nn.Upsample(scale_factor=3 if upscale == 3 else 2, mode='nearest'),
                         _conv(nf, nf), nn.LeakyReLU(LRELU_SLOPE)]
            elif upsample_mode == 'pixelshuffle':
                mods += [_conv(nf, nf * 4), nn.PixelShuffle(2), nn.LeakyReLU(LRELU_SLOPE)]
            else:
                raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        mods += [_conv(nf, nf), nn.LeakyReLU(LRELU_SLOPE), _conv(nf, out_nc)]
        self.model = nn.Sequential(*mods)

    def forward(self, x):
        return self.model(x)


class NLayerDiscriminator(nn.Module):
    """PatchGAN discriminator (architecture.py:983-1024), InstanceNorm2d(affine=False)."""

    def __init__(self, input_nc, ndf=64, n_layers=3):
        super().__init__()
        seq = [nn.Conv2d(input_nc, ndf, 4, 2, 1), nn.LeakyReLU(0.2)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [nn.Conv2d(ndf * prev, ndf * mult, 4, 2, 1, bias=False),
                    nn.InstanceNorm2d(ndf * mult), nn.LeakyReLU(0.2)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [nn.Conv2d(ndf * prev, ndf * mult, 4, 1, 1, bias=False),
                nn.InstanceNorm2d(ndf * mult), nn.LeakyReLU(0.2)]
        seq += [nn.Conv2d(ndf * mult, 1, 4, 1, 1)]
        self.model = nn.Sequential(*seq)

    def forward(self, x):
        return self.model(x)


VGG19_CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
               512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def vgg19_features():
    """torchvision VGG19 (cfg 'E', no BN) feature stack, built locally (torchvision is absent)."""
    layers, c = [], 3
    for v in VGG19_CFG_E:
        if v == 'M':
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=False)]
            c = v
    return nn.Sequential(*layers)


def vgg_init_(features, seed):
    """torchvision's own init rule for VGG (kaiming_normal fan_out/relu, bias 0), seeded.

    Pretrained weights are not available offline; the default
    nn.Conv2d init collapses a 16-layer VGG, so fixtures use this rule.
    """
    g = torch.Generator().manual_seed(seed)
    for m in features:
        if isinstance(m, nn.Conv2d):
            fan_out = m.weight.shape[0] * 9
            std = math.sqrt(2.0 / fan_out)
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
                m.bias.zero_()


class VGGFeatureExtractor(nn.Module):
    """VGG19-54 perceptual extractor (architecture.py:1060-1088): features[:35], input norm."""

    def __init__(self, feature_layer=34, seed=None):
        super().__init__()
        full = vgg19_features()
        if seed is not None:
            vgg_init_(full, seed)
        self.register_buffer('mean', torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
        self.features = nn.Sequential(*list(full.children())[:feature_layer + 1])
        for p in self.features.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.features((x - self.mean) / self.std)


def gaussian_kernel2d(k):
    """GaussianFilter weights (architecture.py:1177-1199): mean (k-1)/2, var (k/6)^2, normalised."""
    mean = (k - 1) / 2.0
    var = (k / 6.0) ** 2.0
    ax = torch.arange(k, dtype=torch.float32)
    xx = ax.repeat(k).view(k, k)
    yy = xx.t()
    g = torch.exp(-((xx - mean) ** 2 + (yy - mean) ** 2) / (2 * var))
    return g / g.sum()


class FilterLow(nn.Module):
    """FilterLow (architecture.py:1208-1224): depthwise gaussian or AvgPool2d, zero pad (k-1)/2."""

    def __init__(self, kernel_size=5, gaussian=False, include_pad=True, padding=True):
        super().__init__()
        self.k = kernel_size
        self.pad = int((kernel_size - 1) / 2) if padding else 0
        self.gaussian = gaussian
        self.include_pad = include_pad
        if gaussian:
            self.register_buffer('w', gaussian_kernel2d(kernel_size).view(1, 1, kernel_size, kernel_size).repeat(3, 1, 1, 1))

    def forward(self, x):
        if self.gaussian:
            return F.conv2d(x, self.w, None, 1, self.pad, 1, 3)
        return F.avg_pool2d(x, self.k, 1, self.pad, count_include_pad=self.include_pad)


class FilterHigh(nn.Module):
    """FilterHigh (architecture.py:1227-1243): x - low(x), normalised to 0.5 + 0.5*(.)."""

    def __init__(self, kernel_size=5, gaussian=False, include_pad=True, normalize=True):
        super().__init__()
        self.low = FilterLow(kernel_size, gaussian, include_pad)
        self.normalize = normalize

    def forward(self, x):
        h = x - self.low(x)
        return 0.5 + h * 0.5 if self.normalize else h


class HaarDWT(nn.Module):
    """Level-1 Haar analysis as used through pytorch_wavelets.DWTForward(J=1, wave='haar')
    at DASR_model.py:56,442-452.  PARITY UNPINNED (third-party, un-vendored): the
    convention fixed here, over each 2x2 block [[a, b], [c, d]]:
        LL = (a+b+c+d)/2, LH = (a+b-c-d)/2, HL = (a-b+c-d)/2, HH = (a-b-c+d)/2
    returns (LL [N,C,H/2,W/2], Hc = cat(LH, HL, HH) along channels [N,3C,H/2,W/2]).
    """

    def forward(self, x):
        a = x[:, :, 0::2, 0::2]
        b = x[:, :, 0::2, 1::2]
        c = x[:, :, 1::2, 0::2]
        d = x[:, :, 1::2, 1::2]
        ll = (a + b + c + d) * 0.5
        lh = (a + b - c - d) * 0.5
        hl = (a - b + c - d) * 0.5
        hh = (a - b - c + d) * 0.5
        return ll, torch.cat((lh, hl, hh), 1)


def init_kaiming_(net, scale):
    """init_weights('kaiming', scale) (networks.py:30-44,62-74): every Conv/Linear:
    kaiming_normal_(a=0, fan_in) * scale, bias 0.  Uses the global torch RNG like the reference."""
    for m in net.modules():
        name = m.__class__.__name__
        if name.find('Conv') != -1 or name.find('Linear') != -1:
            nn.init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            m.weight.data *= scale
            if m.bias is not None:
                m.bias.data.zero_()


def tensor_digest(t):
    """Small, order-sensitive digest used by golden fixtures."""
    t = t.detach().double().flatten()
    idx = torch.arange(t.numel(), dtype=torch.float64)
    return [float(t.sum()), float((t * t).sum()), float((t * torch.cos(idx * 0.37)).sum())]


class Discriminator_VGG_128(nn.Module):
    """codes/SRN/models/modules/architecture.py:442-495: VGG-style discriminator for 128 x 128 inputs, BatchNorm2d(affine) + LeakyReLU(0.2)"""

    def __init__(self, in_nc, nf):
        super().__init__()
        self.conv0_0 = nn.Conv2d(in_nc, nf, 3, 1, 1, bias=True)
        self.conv0_1 = nn.Conv2d(nf, nf, 4, 2, 1, bias=False)
        self.bn0_1 = nn.BatchNorm2d(nf, affine=True)
        self.conv1_0 = nn.Conv2d(nf, nf * 2, 3, 1, 1, bias=False)
        self.bn1_0 = nn.BatchNorm2d(nf * 2, affine=True)
        self.conv1_1 = nn.Conv2d(nf * 2, nf * 2, 4, 2, 1, bias=False)
        self.bn1_1 = nn.BatchNorm2d(nf * 2, affine=True)
        self.conv2_0 = nn.Conv2d(nf * 2, nf * 4, 3, 1, 1, bias=False)
        self.bn2_0 = nn.BatchNorm2d(nf * 4, affine=True)
        self.conv2_1 = nn.Conv2d(nf * 4, nf * 4, 4, 2, 1, bias=False)
        self.bn2_1 = nn.BatchNorm2d(nf * 4, affine=True)
        self.conv3_0 = nn.Conv2d(nf * 4, nf * 8, 3, 1, 1, bias=False)
        self.bn3_0 = nn.BatchNorm2d(nf * 8, affine=True)
        self.conv3_1 = nn.Conv2d(nf * 8, nf * 8, 4, 2, 1, bias=False)
        self.bn3_1 = nn.BatchNorm2d(nf * 8, affine=True)
        self.conv4_0 = nn.Conv2d(nf * 8, nf * 8, 3, 1, 1, bias=False)
        self.bn4_0 = nn.BatchNorm2d(nf * 8, affine=True)
        self.conv4_1 = nn.Conv2d(nf * 8, nf * 8, 4, 2, 1, bias=False)
        self.bn4_1 = nn.BatchNorm2d(nf * 8, affine=True)
        self.linear1 = nn.Linear(512 * 4 * 4, 100)
        self.linear2 = nn.Linear(100, 1)
        self.lrelu = nn.LeakyReLU(0.2)

    def forward(self, x):
        f = self.lrelu(self.conv0_0(x))
        f = self.lrelu(self.bn0_1(self.conv0_1(f)))
        for a in ('1', '2', '3', '4'):
            f = self.lrelu(getattr(self, 'bn%s_0' % a)(getattr(self, 'conv%s_0' % a)(f)))
            f = self.lrelu(getattr(self, 'bn%s_1' % a)(getattr(self, 'conv%s_1' % a)(f)))
        f = self.lrelu(self.linear1(f.reshape(f.size(0), -1)))
        return self.linear2(f)
