"""fp32 PyTorch-CPU restatement of the DSN training step (second hot path).
TEST INFRASTRUCTURE (see oracle/__init__.py).

De_resnet / ResidualBlock      : codes/DSN/model.py:25-55, 213-224
Discriminator (FSD, filters)   : codes/DSN/model.py:60-118, 173-210, 227-293
GeneratorLoss / disc. loss     : codes/DSN/loss.py:11-41, 44-107
iteration                      : codes/DSN/train.py:204-285, optimisers :152-157

Update order.  The reference does `d_loss.backward(retain_graph=True); optimizer_d.step(); g_loss.backward()`
(train.py:241-243,263), which only ran under torch 1.1 (p.data updates bypassed autograd's version counters) and then
back-propagated the generator loss through POST-update D weights with PRE-update activations.  torch >= 1.5 refuses
it.  The semantics fixed here: both gradients are taken from the same pre-update graph, then D steps,
then G steps.  The perceptual term: LPIPS needs pretrained AlexNet (unavailable); 'VGG' = MSE between VGG16 features[:31]
(codes/DSN/loss.py:119-130) with seeded random weights, or w_per = 0.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nets

VGG16_CFG_D = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


class ResidualBlock(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv1 = nn.Conv2d(ch, ch, 3, padding=1)
        self.prelu = nn.PReLU()
        self.conv2 = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return x + self.conv2(self.prelu(self.conv1(x)))


class DeResnet(nn.Module):
    """De_resnet(n_res_blocks=8, scale=4): 8 residual blocks at HR, two stride-2 convs, conv -> sigmoid"""

    def __init__(self, n_res_blocks=8, scale=4):
        super().__init__()
        assert scale == 4
        self.block_input = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.PReLU())
        self.res_blocks = nn.ModuleList([ResidualBlock(64) for _ in range(n_res_blocks)])
        self.down_sample = nn.Sequential(nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.PReLU(),
                                         nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.PReLU())
        self.block_output = nn.Conv2d(64, 3, 3, padding=1)

    def forward(self, x):
        b = self.block_input(x)
        for r in self.res_blocks:
            b = r(b)
        return torch.sigmoid(self.block_output(self.down_sample(b)))


class GeneratorDSGAN(nn.Module):
    """`Generator` (--generator DSGAN, codes/DSN/model.py:7-22): De_resnet without the stride-2 convs, applied to the bicubic LR image"""

    def __init__(self, n_res_blocks=8):
        super().__init__()
        self.block_input = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.PReLU())
        self.res_blocks = nn.ModuleList([ResidualBlock(64) for _ in range(n_res_blocks)])
        self.block_output = nn.Conv2d(64, 3, 3, padding=1)

    def forward(self, x):
        b = self.block_input(x)
        for r in self.res_blocks:
            b = r(b)
        return torch.sigmoid(self.block_output(b))


class _GaussConv(nn.Module):
    def __init__(self, k, pad):
        super().__init__()
        self.gaussian_filter = nn.Conv2d(3, 3, k, padding=pad, groups=3, bias=False)
        self.gaussian_filter.weight.data = nets.gaussian_kernel2d(k).view(1, 1, k, k).repeat(3, 1, 1, 1)
        self.gaussian_filter.weight.requires_grad = False

    def forward(self, x):
        return self.gaussian_filter(x)


class FilterLow(nn.Module):
    def __init__(self, kernel_size=5, padding=True, include_pad=True, gaussian=False):
        super().__init__()
        pad = int((kernel_size - 1) / 2) if padding else 0
        self.filter = _GaussConv(kernel_size, pad) if gaussian else nn.AvgPool2d(kernel_size, 1, pad, count_include_pad=include_pad)

    def forward(self, x):
        return self.filter(x)


class FilterHigh(nn.Module):
    def __init__(self, kernel_size=5, include_pad=True, gaussian=False):
        super().__init__()
        self.filter_low = FilterLow(kernel_size, True, include_pad, gaussian)

    def forward(self, x):
        return 0.5 + (x - self.filter_low(x)) * 0.5


class DiscriminatorBasic(nn.Module):
    """FSD net (model.py:173-210): 5x5 convs (bias) with Instance/Batch norm, 1x1 head"""

    def __init__(self, nc=3, norm='Instance'):
        super().__init__()
        N = nn.InstanceNorm2d if norm == 'Instance' else nn.BatchNorm2d
        self.net = nn.Sequential(nn.Conv2d(nc, 64, 5, padding=2), nn.LeakyReLU(0.2),
                                 nn.Conv2d(64, 128, 5, padding=2), N(128), nn.LeakyReLU(0.2),
                                 nn.Conv2d(128, 256, 5, padding=2), N(256), nn.LeakyReLU(0.2),
                                 nn.Conv2d(256, 1, 1))

    def forward(self, x):
        return self.net(x)


class NLayerDiscriminatorDSN(nn.Module):
    """codes/DSN/model.py:121-170 with n_layers=2, kw=4, padw=1; stride 1 (nld_s1) or 2 (nld_s2).  `use_bias` (model.py:139-142) is True under
    InstanceNorm and False under BatchNorm: the two normalised convs carry a bias only with norm_layer 'Instance'"""

    def __init__(self, input_nc, ndf=64, stride=2, norm='Instance'):
        super().__init__()
        N = nn.InstanceNorm2d if norm == 'Instance' else nn.BatchNorm2d
        ub = norm == 'Instance'
        self.model = nn.Sequential(nn.Conv2d(input_nc, ndf, 4, stride, 1), nn.LeakyReLU(0.2),
                                   nn.Conv2d(ndf, 2 * ndf, 4, stride, 1, bias=ub), N(2 * ndf), nn.LeakyReLU(0.2),
                                   nn.Conv2d(2 * ndf, 4 * ndf, 4, 1, 1, bias=ub), N(4 * ndf), nn.LeakyReLU(0.2),
                                   nn.Conv2d(4 * ndf, 1, 4, 1, 1))

    def forward(self, x):
        return self.model(x)


class Discriminator(nn.Module):
    """Discriminator(D_arch='FSD' | 'nld_s1' | 'nld_s2') with the frequency-separation front end (model.py:60-118); output = sigmoid"""

    def __init__(self, kernel_size=5, norm_layer='Instance', filter_type='gau', D_arch='FSD', cs='cat', wgan=False):
        super().__init__()
        self.wgan = bool(wgan)   # --wgan (model.py:61,65,104-105): no sigmoid on the output map
        self.filter_type = filter_type.lower()
        self.cs = cs.lower()
        if self.cs not in ('cat', 'sum'):
            raise NotImplementedError('Wavelet format [{:s}] not recognized'.format(cs))   # model.py:117-118
        nc = 3
        if self.filter_type in ('gau', 'avg_pool'):
            self.filter = FilterHigh(kernel_size, include_pad=False, gaussian=self.filter_type == 'gau')
        elif self.filter_type == 'wavelet':
            self.dwt = nets.HaarDWT()
            nc = 9 if self.cs == 'cat' else 3   # model.py:78
        else:
            raise NotImplementedError('Frequency Separation type [{:s}] not recognized'.format(filter_type))
        if D_arch.lower() == 'fsd':
            self.net = DiscriminatorBasic(nc, norm_layer)
        elif D_arch.lower() in ('nld_s1', 'nld_s2'):
            self.net = NLayerDiscriminatorDSN(nc, 64, 1 if D_arch.lower() == 'nld_s1' else 2, norm_layer)
        else:
            raise NotImplementedError('Discriminator architecture [{:s}] not recognized'.format(D_arch))

    def front(self, x):
        if self.filter_type == 'wavelet':
            hc = self.dwt(x)[1] * 0.5 + 0.5   # cat(LH, HL, HH), normalised (model.py:108-118)
            if self.cs == 'sum':               # (LH + HL + HH) / 3 (model.py:113-114)
                c = hc.shape[1] // 3
                return (hc[:, :c] + hc[:, c:2 * c] + hc[:, 2 * c:]) / 3.
            return hc
        return self.filter(x)

    def forward(self, x, y=None):
        """y given (train.py --ragan, :221-223): relativistic -- the per-pixel batch mean of D(y)'s logits is subtracted (model.py:98-106)"""
        z = self.net(self.front(x))
        if y is not None:
            z = z - self.net(self.front(y)).mean(0, keepdim=True)
        return z if self.wgan else torch.sigmoid(z)


def vgg16_features31(seed):
    layers, c = [], 3
    for v in VGG16_CFG_D:
        if v == 'M':
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=False)]
            c = v
    net = nn.Sequential(*layers[:31])
    nets.vgg_init_(net, seed)
    for p in net.parameters():
        p.requires_grad = False
    return net


def rot_flip_pair(x, y):
    """PerceptualLoss.forward with rotations = flips = True (codes/DSN/loss.py:155-168): the same random symmetry of the square on both images,
    drawn from python's global `random`: k_rot, then the row-flip coin, then the column-flip coin"""
    import random
    k_rot = random.choice([-1, 0, 1])
    x, y = torch.rot90(x, k_rot, [2, 3]), torch.rot90(y, k_rot, [2, 3])
    if random.choice([True, False]):
        x, y = torch.flip(x, (2,)), torch.flip(y, (2,))
    if random.choice([True, False]):
        x, y = torch.flip(x, (3,)), torch.flip(y, (3,))
    return x, y


class DSNTrainer:
    """one training iteration of codes/DSN/train.py:204-285 (non-wgan; `ragan`: the relativistic discriminator calls of :221-223)"""

    def __init__(self, netG=None, netD=None, lr=1e-4, beta1=0.5, w_col=1.0, w_tex=0.005, w_per=0.01, per_type='VGG',
                 kernel_size=5, filter_type='gau', norm_layer='Instance', vgg_seed=78, num_epochs=400, num_decay_epochs=150, netF=None, ragan=False,
                 disc_freq=1, gen_freq=1, lpips_rot_flip=False, wgan=False):
        self.ragan = ragan
        self.wgan = bool(wgan)   # --wgan (train.py:45,231-241; loss.py:11-41): Wasserstein terms on the un-squashed map + gradient penalty
        self.lpips_rot_flip = bool(lpips_rot_flip)   # --lpips_rot_flip (train.py:52, loss.py:66,149-168)
        # --disc_freq / --gen_freq (train.py:55-56): `iteration += 1` at the top of the loop body (:206), the discriminator steps when
        # iteration % disc_freq == 0 (:229), the generator when iteration % gen_freq == 0 (:251)
        self.disc_freq, self.gen_freq, self.iteration_count = int(disc_freq), int(gen_freq), 0
        self.G = netG if netG is not None else DeResnet()
        self.D = netD if netD is not None else Discriminator(kernel_size, norm_layer, filter_type)
        self.w_col, self.w_tex, self.w_per = w_col, w_tex, w_per
        self.filter_type = filter_type.lower()
        if self.filter_type in ('gau', 'avg_pool'):
            self.color_filter = FilterLow(kernel_size, padding=False, gaussian=self.filter_type == 'gau')
        else:
            dwt = nets.HaarDWT()
            self.color_filter = lambda x: dwt(x)[0] * 0.5
        self.per = None
        self.lpips = None
        if w_per > 0 and per_type == 'LPIPS':   # loss.py:68-69: PerceptualLoss() -> LPIPS(alex)(x, y, normalize=True).mean()
            from . import lpips as _lp
            self.lpips = netF if netF is not None else _lp.PerceptualLossLPIPS(_lp.LPIPSAlex(seed=vgg_seed))
        elif w_per > 0:
            assert per_type == 'VGG'
            self.per = vgg16_features31(vgg_seed)
        self.opt_g = torch.optim.Adam(self.G.parameters(), lr=lr, betas=(beta1, 0.999))
        self.opt_d = torch.optim.Adam([p for p in self.D.parameters() if p.requires_grad], lr=lr, betas=(beta1, 0.999))
        start_decay = num_epochs - num_decay_epochs
        rule = lambda e: 1.0 if e < start_decay else 1.0 - max(0.0, float(e - start_decay) / num_decay_epochs)
        self.sched_g = torch.optim.lr_scheduler.LambdaLR(self.opt_g, rule)
        self.sched_d = torch.optim.lr_scheduler.LambdaLR(self.opt_d, rule)
        self.log = OrderedDict()

    def iteration(self, hr, bicubic_lr, real_lr):
        fake = self.G(bicubic_lr if isinstance(self.G, GeneratorDSGAN) else hr)   # codes/DSN/train.py:213-217
        real_tex, fake_tex = (self.D(real_lr, fake), self.D(fake, real_lr)) if self.ragan else (self.D(real_lr), self.D(fake))
        self.iteration_count += 1
        upd_d, upd_g = self.iteration_count % self.disc_freq == 0, self.iteration_count % self.gen_freq == 0
        if self.wgan:
            # train.py:231-236: ONE random mixing weight for the whole batch (torch's global RNG, drawn only when the discriminator steps), the gradient
            # of the MEAN output w.r.t. the mixed images, and 10 (||gradient||_2 - 1)^2 with the norm over the whole batch tensor
            grad_pen = fake.new_zeros(())
            if upd_d:
                rand = torch.rand(1).item()
                sample = rand * real_lr + (1 - rand) * fake
                gp_tex = self.D(sample)
                gradient = torch.autograd.grad(gp_tex.mean(), sample, create_graph=True)[0]
                grad_pen = 10 * (gradient.norm() - 1) ** 2
            d_loss = -real_tex.mean() + fake_tex.mean() + grad_pen   # loss.py:33-36
            tex = torch.mean(-fake_tex)                              # loss.py:18-19
            self.grad_pen = float(grad_pen)
        else:
            d_loss = -torch.log(real_tex + 1e-8).mean() - torch.log(1 - fake_tex + 1e-8).mean()
            tex = torch.mean(-torch.log(fake_tex + 1e-8))
        col = F.l1_loss(self.color_filter(fake), self.color_filter(bicubic_lr))
        g_loss = self.w_col * col + self.w_tex * tex
        per = torch.zeros(())
        if self.lpips is not None:
            x, y = rot_flip_pair(fake, bicubic_lr) if self.lpips_rot_flip else (fake, bicubic_lr)
            per = self.lpips(x, y)
            g_loss = g_loss + self.w_per * per
        if self.per is not None:
            per = F.mse_loss(self.per(fake), self.per(bicubic_lr))
            g_loss = g_loss + self.w_per * per
        d_params = [p for p in self.D.parameters() if p.requires_grad]
        g_params = list(self.G.parameters())
        gd = torch.autograd.grad(d_loss, d_params, retain_graph=True) if upd_d else None
        gg = torch.autograd.grad(g_loss, g_params) if upd_g else None
        if upd_d:
            for p, g in zip(d_params, gd):
                p.grad = g
            self.opt_d.step()
        if upd_g:
            for p, g in zip(g_params, gg):
                p.grad = g
            self.opt_g.step()
        self.fake = fake.detach()
        self.log.update({'loss/d_tex_loss': d_loss.item(), 'loss/g_tex_loss': tex.item(), 'loss/color_loss': col.item(),
                         'loss/perceptual_loss': per.item(), 'loss/g_overall_loss': g_loss.item(),
                         'disc_score/real': real_tex.mean().item(), 'disc_score/fake': fake_tex.mean().item()})
        if self.wgan:
            self.log['disc_score/gradient_penalty'] = self.grad_pen   # train.py:247-248

    def end_epoch(self):
        self.sched_d.step()
        self.sched_g.step()
