"""Seed recipes shared by oracle/gen_golden.py (reference side) and tests (oracle / HIP side).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Weights and inputs are functions of
integer seeds only, independent of module construction order, so the reference, the
oracle and the HIP path can all be fed bit-identical tensors without shipping them.
"""
import math
from collections import OrderedDict

import torch

CASES = OrderedDict([
    # name: (trainer, nf, nb, n (dataloader batch), lr_size, fs, d_in_nc)
    ('cfg1_sr_nf32_nb4_b2_64', dict(kind='sr', nf=32, nb=4, n=2, lr=64)),
    ('sr_nf64_nb2_b2_32', dict(kind='sr', nf=64, nb=2, n=2, lr=32)),
    ('sr_nf64_nb1_b1_24x40', dict(kind='sr', nf=64, nb=1, n=1, lr=(24, 40))),
    ('dasr_wavelet_nf32_nb2_n2_32', dict(kind='dasr', nf=32, nb=2, n=2, lr=32, fs='wavelet', d_in_nc=9)),
    ('dasr_gau9_nf64_nb1_n1_32', dict(kind='dasr', nf=64, nb=1, n=1, lr=32, fs='gau', d_in_nc=3)),
    # round 2: the production schedule (batch >= 8 -> two sub-batch streams) and the full ESRGAN depth
    ('sr_nf64_nb2_b8_32', dict(kind='sr', nf=64, nb=2, n=8, lr=32)),
    ('sr_nf64_nb23_b2_32', dict(kind='sr', nf=64, nb=23, n=2, lr=32)),
    # PixelShuffle upsampler (architecture.py:186-191, block.py:838-851; the reference's define_G never selects it: the generator of
    # the fixture is the reference's RRDBNet(upsample_mode='pixelshuffle') put behind its SRModel)
    ('sr_ps_nf64_nb1_b2_32', dict(kind='sr', nf=64, nb=1, n=2, lr=32, upsample_mode='pixelshuffle')),
    # feature_criterion LPIPS (what the shipped train_DASR*.json use): the reference's PerceptualLossLPIPS with its real linear heads on a
    # seeded stand-in AlexNet (oracle/ref_import.py), behind its DASR_Model
    ('dasr_lpips_wavelet_nf32_nb2_n2_32', dict(kind='dasr', nf=32, nb=2, n=2, lr=32, fs='wavelet', d_in_nc=9, fea='LPIPS')),
    # source-domain discriminator (gan_H_source > 0, which_model_pairD 'discriminator_patch' as in the shipped aim2019 JSON): DASR_model.py:250-259,287-303
    ('dasr_srcD_wavelet_nf32_nb2_n2_32', dict(kind='dasr', nf=32, nb=2, n=2, lr=32, fs='wavelet', d_in_nc=9, gan_src=0.02)),
    # relativistic average GAN (`ragan: true`, DASR_model.py:240-244,252-256,273-275,291-293) on both discriminators, n = 3 so that the batch means matter
    ('dasr_ragan_wavelet_nf32_nb1_n3_32', dict(kind='dasr', nf=32, nb=1, n=3, lr=32, fs='wavelet', d_in_nc=9, gan_src=0.02, ragan=True)),
    # source-domain discriminator = Discriminator_VGG_128 (which_model_pairD 'discriminator_vgg_128', architecture.py:442-495): BatchNorm in training
    # mode, two Linear layers; needs 128 x 128 inputs: gaussian frequency split at HR = 4 x 32
    ('dasr_srcVGG128_gau5_nf32_nb1_n3_32', dict(kind='dasr', nf=32, nb=1, n=3, lr=32, fs='gau', d_in_nc=3, gan_src=0.02, pairD='discriminator_vgg_128')),
    # gan_type 'lsgan' / 'wgan-gp' (GANLoss, loss.py:8-40) on both discriminators
    ('dasr_lsgan_wavelet_nf32_nb1_n2_32', dict(kind='dasr', nf=32, nb=1, n=2, lr=32, fs='wavelet', d_in_nc=9, gan_src=0.02, gan_type='lsgan')),
    ('dasr_ragan_lsgan_wavelet_nf32_nb1_n3_32', dict(kind='dasr', nf=32, nb=1, n=3, lr=32, fs='wavelet', d_in_nc=9, gan_src=0.02, ragan=True, gan_type='lsgan')),
    ('dasr_wgan_gau9_nf32_nb1_n2_32', dict(kind='dasr', nf=32, nb=1, n=2, lr=32, fs='gau', d_in_nc=3, gan_src=0.02, gan_type='wgan-gp')),
    # round 3: pixel_criterion / feature_criterion 'l2' = nn.MSELoss (SR_model.py:33-36, DASR_model.py:79-80,95-96); multiweights off so that the
    # plain pixel term really goes through cri_pix (with multiweights the reference hard-codes |.|, DASR_model.py:213-215)
    ('sr_l2_nf32_nb1_b2_32', dict(kind='sr', nf=32, nb=1, n=2, lr=32, pix='l2')),
    ('dasr_l2_wavelet_nf32_nb1_n2_32', dict(kind='dasr', nf=32, nb=1, n=2, lr=32, fs='wavelet', d_in_nc=9, pix='l2', fea='l2', multiweights=False)),
    # round 3: the GAN step at the full ESRGAN depth (nb = 23; n = 1 -> 2 crops through G)
    ('dasr_wavelet_nf64_nb23_n1_32', dict(kind='dasr', nf=64, nb=23, n=1, lr=32, fs='wavelet', d_in_nc=9)),
])


def seeded_state_dict(template_sd, seed, scale):
    """kaiming-normal(fan_in)*scale weights / zero biases, one generator per key index."""
    out = OrderedDict()
    for i, (k, v) in enumerate(template_sd.items()):
        if v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            g = torch.Generator().manual_seed(seed * 100003 + i)
            out[k] = torch.randn(v.shape, generator=g) * (math.sqrt(2.0 / fan_in) * scale)
        elif v.dim() == 2:   # nn.Linear (Discriminator_VGG_128)
            g = torch.Generator().manual_seed(seed * 100003 + i)
            out[k] = torch.randn(v.shape, generator=g) * (math.sqrt(2.0 / v.shape[1]) * scale)
        elif v.dim() == 1 and k.endswith('weight') and k.split('.')[-2].startswith('bn'):   # BatchNorm gamma: around 1, not exactly
            g = torch.Generator().manual_seed(seed * 100003 + i)
            out[k] = 1.0 + (torch.rand(v.shape, generator=g) - 0.5) * 0.2
        elif k.endswith('bias'):
            g = torch.Generator().manual_seed(seed * 100003 + i)
            out[k] = (torch.rand(v.shape, generator=g) - 0.5) * 0.02  # non-zero so bias paths are exercised
        else:
            out[k] = v.clone()
    return out


def make_opt(case):
    c = CASES[case] if isinstance(case, str) else case
    opt = {
        'is_train': True, 'gpu_ids': None, 'scale': 4, 'chop': False, 'val_lpips': False,
        'model': 'sr' if c['kind'] == 'sr' else 'DASR', 'multiweights': bool(c.get('multiweights', True)),
        'allow_random_perceptual': True,   # fixtures run seeded stand-ins of the pretrained VGG19 / AlexNet (they cannot be downloaded offline)
        'path': {'pretrain_model_G': None, 'pretrain_model_D_target': None, 'pretrain_model_D_source': None,
                 'models': '/tmp/dasr_golden', 'training_state': '/tmp/dasr_golden'},
        'network_G': {'which_model_G': 'RRDB_net', 'norm_type': None, 'mode': 'CNA', 'nf': c['nf'], 'nb': c['nb'],
                      'in_nc': 3, 'out_nc': 3, 'gc': 32, 'scale': 4, 'upsample_mode': c.get('upsample_mode')},
        'train': {'lr_G': 1e-4, 'weight_decay_G': 0, 'beta1_G': 0.9, 'lr_scheme': 'MultiStepLR',
                  'lr_steps': [2, 4], 'lr_gamma': 0.5, 'pixel_criterion': c.get('pix', 'l1'), 'pixel_weight': 1.0,
                  'manual_seed': 0},
    }
    if c['kind'] == 'dasr':
        opt['network_D'] = {'which_model_D': 'discriminator_patch', 'which_model_pairD': c.get('pairD', 'discriminator_patch'), 'norm_type': 'Batch',
                            'act_type': 'leakyrelu', 'mode': 'CNA', 'nf': 64, 'in_nc': c['d_in_nc'], 'n_layers': 2}
        opt['train'].update({'lr_D': 1e-4, 'weight_decay_D': 0, 'beta1_D': 0.9, 'fs': c['fs'], 'fs_kernel_size': 9,
                             'norm': True, 'sup_LL': True, 'pixel_LL_weight': 1, 'feature_criterion': c.get('fea', 'l1'),
                             'feature_weight': 1, 'gan_type': c.get('gan_type', 'vanilla'), 'ragan': bool(c.get('ragan', False)), 'gan_H_target': 0.01,
                             'gan_H_source': c.get('gan_src', 0), 'G_update_inter': 1, 'D_update_inter': 1})
    return opt


def make_batch(case, seed=1234):
    """Synthetic batch dict: torch.rand from Generator(seed)."""
    c = CASES[case] if isinstance(case, str) else case
    g = torch.Generator().manual_seed(seed)
    n = c['n']
    h, w = (c['lr'], c['lr']) if isinstance(c['lr'], int) else c['lr']
    if c['kind'] == 'sr':
        return {'LR': torch.rand(n, 3, h, w, generator=g), 'HR': torch.rand(n, 3, 4 * h, 4 * w, generator=g)}
    return {'LR_fake': torch.rand(n, 3, h, w, generator=g), 'LR_real': torch.rand(n, 3, h, w, generator=g),
            'HR': torch.rand(n, 3, 4 * h, 4 * w, generator=g), 'HR_unpair': torch.rand(n, 3, 4 * h, 4 * w, generator=g),
            'fake_w': torch.rand(n, 1, h, w, generator=g)}


def subsample(t, k=64):
    """Deterministic k-element strided sub-sample of a tensor (flattened)."""
    f = t.detach().flatten()
    step = max(1, f.numel() // k)
    return f[::step][:k].clone()
