"""Generate tests/golden/dsn_*.npz by running the REFERENCE DSN modules (codes/DSN/model.py, loss.py), imported from
/root/reference in a process of their own (DSN and SRN both define top-level `model`/`utils`/`loss` modules).
TEST INFRASTRUCTURE (see oracle/__init__.py).   python -m oracle.gen_golden_dsn
The reference training loop cannot run on torch >= 1.5 (stale-graph update order); the fixture pins the
modules and the losses, with both gradients taken from the pre-update graph (the semantics oracle/dsn.py fixes)."""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import fixtures, nets, dsn

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
REF = '/root/reference/codes'

DSN_CASES = {
    'dsn_gau5_inst_b2_128': dict(filter='gau', k=5, norm='Instance', n=2, crop=128),
    'dsn_wavelet_inst_b2_128': dict(filter='wavelet', k=5, norm='Instance', n=2, crop=128),
    'dsn_avg5_inst_b1_160': dict(filter='avg_pool', k=5, norm='Instance', n=1, crop=160),
    # --per_type LPIPS (the reference default): its GeneratorLoss builds PerceptualLoss() = LPIPS(alex) with the real linear heads on the
    # stand-in AlexNet(seed 78); 64 x 64 LR images
    'dsn_gau5_inst_b1_256_lpips': dict(filter='gau', k=5, norm='Instance', n=1, crop=256, per='LPIPS'),
    # --discriminator nld_s1 / nld_s2 (model.py:84-89,121-170): 4x4 convs, stride 1 / 2 in the first two
    'dsn_wavelet_nld_s2_b2_128': dict(filter='wavelet', k=5, norm='Instance', n=2, crop=128, arch='nld_s2'),
    'dsn_gau5_nld_s1_b1_128': dict(filter='gau', k=5, norm='Instance', n=1, crop=128, arch='nld_s1'),
    # --generator DSGAN (model.py:7-22, train.py:213-215): the generator runs on the bicubic LR image
    'dsn_dsgan_gau5_inst_b2_128': dict(filter='gau', k=5, norm='Instance', n=2, crop=128, gen='DSGAN'),
    # --ragan (train.py:221-223, model.py:98-106): relativistic discriminator outputs, n = 3 so that the batch means matter
    'dsn_gau5_inst_b3_128_ragan': dict(filter='gau', k=5, norm='Instance', n=3, crop=128, ragan=True),
    # --norm_layer Batch (model.py:176-189): BatchNorm2d in training mode, statistics per discriminator call (the layout of the reference's test.tar)
    'dsn_gau5_batch_b2_128': dict(filter='gau', k=5, norm='Batch', n=2, crop=128),
    'dsn_avg5_batch_b3_128_ragan': dict(filter='avg_pool', k=5, norm='Batch', n=3, crop=128, ragan=True),
    # --cat_or_sum sum (round 3; model.py:113-114): the discriminator sees (LH + HL + HH) / 3, three channels instead of nine
    'dsn_wavelet_sum_inst_b2_128': dict(filter='wavelet', k=5, norm='Instance', n=2, crop=128, cs='sum'),
    # --lpips_rot_flip (round 4; train.py:52, loss.py:66,149-168): random rot90 / flips of both images in front of LPIPS, python `random` seeded
    # per call (seed 3 draws k_rot = -1, flip rows, no column flip; seed + 1 for the second iteration)
    # --wgan (round 4; train.py:45,231-241, model.py:104-105, loss.py:11-41): no sigmoid, Wasserstein terms, gradient penalty 10 (||d mean D(sample) / d sample|| - 1)^2
    # on ONE random mix of the real and fake batch (torch.manual_seed(c['tseed']) in front of the draw), second-order pass through the discriminator
    'dsn_gau5_inst_b2_128_wgan': dict(filter='gau', k=5, norm='Instance', n=2, crop=128, wgan=True, tseed=11),
    'dsn_wavelet_inst_b2_128_wgan': dict(filter='wavelet', k=5, norm='Instance', n=2, crop=128, wgan=True, tseed=12),
    # round 6: --norm_layer Batch with the nld discriminators (model.py:136-142: `use_bias` False, BatchNorm2d behind the 2nd / 3rd conv) and with --wgan
    # (the gradient penalty's second-order pass goes through BatchNorm in training mode; D(sample) is a third training-mode call per iteration)
    'dsn_gau5_nld_s1_batch_b2_128': dict(filter='gau', k=5, norm='Batch', n=2, crop=128, arch='nld_s1'),
    'dsn_wavelet_nld_s2_batch_b3_128': dict(filter='wavelet', k=5, norm='Batch', n=3, crop=128, arch='nld_s2'),
    'dsn_gau5_batch_b2_128_wgan': dict(filter='gau', k=5, norm='Batch', n=2, crop=128, wgan=True, tseed=13),
    'dsn_wavelet_nld_s2_batch_b2_128_wgan': dict(filter='wavelet', k=5, norm='Batch', n=2, crop=128, arch='nld_s2', wgan=True, tseed=14),
    'dsn_gau5_inst_b2_256_lpips_rotflip': dict(filter='gau', k=5, norm='Instance', n=2, crop=256, per='LPIPS', rot_flip=True, rseed=3),
}


def dsn_state(template_sd, seed, scale):
    sd = fixtures.seeded_state_dict({k: v for k, v in template_sd.items() if 'gaussian_filter' not in k and 'num_batches' not in k}, seed, scale)
    out = {}
    for k, v in template_sd.items():
        out[k] = sd[k] if k in sd else v.clone()
    return out


def dsn_batch(c, seed=4321):
    g = torch.Generator().manual_seed(seed)
    n, s = c['n'], c['crop']
    return (torch.rand(n, 3, s, s, generator=g), torch.rand(n, 3, s // 4, s // 4, generator=g), torch.rand(n, 3, s // 4, s // 4, generator=g))


def collect(G, D, color_filter, per_net, c, w=(1.0, 0.005, 0.01)):
    hr, bic, real = dsn_batch(c)
    if c.get('rot_flip'):
        import random
        random.seed(c['rseed'])
    fake = G(bic if c.get('gen') == 'DSGAN' else hr)
    rt, ft = (D(real, fake), D(fake, real)) if c.get('ragan') else (D(real), D(fake))
    if c.get('wgan'):   # the statements of train.py:231-241 / loss.py:18-19,33-36 with the reference's modules
        torch.manual_seed(c['tseed'])
        rand = torch.rand(1).item()
        sample = rand * real + (1 - rand) * fake
        gp_tex = D(sample)
        gradient = torch.autograd.grad(gp_tex.mean(), sample, create_graph=True)[0]
        grad_pen = 10 * (gradient.norm() - 1) ** 2
        d_loss = -rt.mean() + ft.mean() + grad_pen
        tex = torch.mean(-ft)
    else:
        d_loss = -torch.log(rt + 1e-8).mean() - torch.log(1 - ft + 1e-8).mean()
        tex = torch.mean(-torch.log(ft + 1e-8))
    col = torch.nn.functional.l1_loss(color_filter(fake), color_filter(bic))
    per = per_net(fake, bic) if c.get('per') == 'LPIPS' else torch.nn.functional.mse_loss(per_net(fake), per_net(bic))
    g_loss = w[0] * col + w[1] * tex + w[2] * per
    dp = [p for p in D.parameters() if p.requires_grad]
    gd = torch.autograd.grad(d_loss, dp, retain_graph=True)
    gg = torch.autograd.grad(g_loss, list(G.parameters()))
    return {'fake_sub': fixtures.subsample(fake).numpy(), 'real_tex_sub': fixtures.subsample(rt).detach().numpy(),
            'fake_tex_sub': fixtures.subsample(ft).detach().numpy(),
            'losses': np.array([d_loss.item(), tex.item(), col.item(), per.item(), g_loss.item()] + ([float(grad_pen)] if c.get('wgan') else [])),
            'gradG_norm': np.array([float(x.double().norm()) for x in gg]), 'gradD_norm': np.array([float(x.double().norm()) for x in gd]),
            'G_keys': np.array(list(G.state_dict().keys())), 'D_keys': np.array(list(D.state_dict().keys()))}


def main():
    from .ref_import import _mod
    class DWTForward(nn.Module):
        def __init__(self, J=1, mode='reflect', wave='haar'):
            super().__init__()
            self.h = nets.HaarDWT()
        def forward(self, x):
            ll, hc = self.h(x)
            c = hc.shape[1] // 3
            return ll, [torch.stack((hc[:, :c], hc[:, c:2 * c], hc[:, 2 * c:]), 2)]
        def cuda(self):
            return self
    _mod('pytorch_wavelets', DWTForward=DWTForward)
    tv = _mod('torchvision')
    class _V(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = dsn.vgg16_features31(78)
    class _Alex(nn.Module):
        def __init__(self):
            super().__init__()
            from . import lpips
            self.features = lpips.alexnet_init_(lpips.alexnet_features(), 78)
    tv.models = _mod('torchvision.models', alexnet=lambda pretrained=False: _Alex())
    _mod('torchvision.models.vgg', vgg16=lambda pretrained=False: types.SimpleNamespace(features=list(dsn.vgg16_features31(78)) + [nn.Identity()] * 0),
         vgg19=lambda pretrained=False: None)
    sk = _mod('skimage'); _mod('skimage.measure', compare_ssim=None); _mod('skimage.color'); _mod('skimage.transform')
    _mod('IPython', embed=lambda *a, **k: None)
    _mod('cv2')
    sys.path[:0] = [os.path.join(REF, 'DSN'), REF]
    import model as rmodel
    import loss as rloss
    os.makedirs(OUT, exist_ok=True)
    only = [a for a in sys.argv[1:] if a in DSN_CASES]
    for name, c in DSN_CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(0)
        G = rmodel.Generator(n_res_blocks=8) if c.get('gen') == 'DSGAN' else rmodel.De_resnet(n_res_blocks=8, scale=4)
        D = rmodel.Discriminator(kernel_size=c['k'], D_arch=c.get('arch', 'FSD'), norm_layer=c['norm'], filter_type=c['filter'], cs=c.get('cs', 'cat'),
                                 wgan=bool(c.get('wgan')))
        G.load_state_dict(dsn_state(G.state_dict(), 21, 0.5))
        D.load_state_dict(dsn_state(D.state_dict(), 22, 1.0))
        _cuda = nn.Module.cuda
        nn.Module.cuda = lambda self, *a, **k: self   # loss.py:63-64 moves the colour filter to the GPU unconditionally
        try:
            gl = rloss.GeneratorLoss(kernel_size=c['k'], per_type=c.get('per', 'VGG'), filter=c['filter'], w_col=1, w_tex=0.005, w_per=0.01,
                                     lpips_rot_flip=bool(c.get('rot_flip')))
        finally:
            nn.Module.cuda = _cuda
        if c['filter'] == 'wavelet':
            cf = lambda x: nets.HaarDWT()(x)[0] * 0.5   # reference filter_wavelet_LL calls .cuda(); same arithmetic
        else:
            cf = gl.color_filter
        fx = collect(G, D, cf, gl.perceptual_loss if c.get('per') == 'LPIPS' else gl.perceptual_loss.loss_network, c)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **fx)
        print(name, fx['losses'])
    if only:
        return
    # real-weights known-answer test: the only weights file the reference ships, codes/DSN/test.tar = state_dict of
    # Discriminator(D_arch='FSD', norm_layer='Batch', filter_type='gau', kernel_size=5) (model.py:60-118,173-189).  The fixture carries the
    # weights (a data file of the reference) and the eval-mode output of the reference module on a seeded input.
    sd = torch.load(os.path.join(REF, 'DSN', 'test.tar'), map_location='cpu', weights_only=False)
    D = rmodel.Discriminator(kernel_size=5, D_arch='FSD', norm_layer='Batch', filter_type='gau', cs='cat')
    D.load_state_dict(sd)
    D.eval()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(97))
    with torch.no_grad():
        y = D(x)
    fx = {'w/' + k: v.numpy() for k, v in sd.items()}
    fx['out'] = y.numpy()
    np.savez_compressed(os.path.join(OUT, 'dsn_fsd_batch_test_tar.npz'), **fx)
    print('dsn_fsd_batch_test_tar', tuple(y.shape), float(y.mean()))


if __name__ == '__main__':
    main()
