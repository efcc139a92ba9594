"""fp32 PyTorch-CPU restatement of the two SRN trainers (TEST INFRASTRUCTURE, see oracle/__init__.py).

SRTrainer   follows codes/SRN/models/SR_model.py:18-85   (generator-only L1/L2 step)
DASRTrainer follows codes/SRN/models/DASR_model.py:24-330 (full GAN step)
Reference quirks reproduced on purpose: double pixel weight with
``multiweights``, scheduler stepped before the optimizer, gc ignored, D not frozen in
the G step (its discarded wgrads have no numeric effect and are skipped here).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nets


def _opt(d, k, default=None):
    v = d.get(k) if isinstance(d, dict) else None
    return default if v is None else v


class SRTrainer:
    def __init__(self, opt, netG=None):
        t = opt['train']
        g = opt['network_G']
        self.netG = netG if netG is not None else nets.RRDBNet(g['in_nc'], g['out_nc'], g['nf'], g['nb'], opt['scale'])
        if netG is None:
            nets.init_kaiming_(self.netG, 0.1)  # networks.py:142-143
        self.l_pix_w = t['pixel_weight']
        self.cri = F.l1_loss if t['pixel_criterion'] == 'l1' else F.mse_loss
        self.opt_G = torch.optim.Adam(self.netG.parameters(), lr=t['lr_G'],
                                      weight_decay=_opt(t, 'weight_decay_G', 0))  # SR_model.py:50-51
        self.sched = torch.optim.lr_scheduler.MultiStepLR(self.opt_G, t['lr_steps'], t['lr_gamma'])
        self.log = OrderedDict()

    def update_learning_rate(self):
        self.sched.step()  # base_model.py:35-37

    def feed_data(self, data):
        self.var_L, self.real_H = data['LR'], data['HR']

    def optimize_parameters(self, step):
        self.opt_G.zero_grad()
        self.fake_H = self.netG(self.var_L)
        l_pix = self.l_pix_w * self.cri(self.fake_H, self.real_H)
        l_pix.backward()
        self.opt_G.step()
        self.log['l_pix'] = l_pix.item()


class DASRTrainer:
    def __init__(self, opt, netG=None, netD=None, netF=None, vgg_seed=77, netD_source=None):
        t = opt['train']
        g, d = opt['network_G'], opt['network_D']
        self.multiweights = opt.get('multiweights')
        self.l_gan_w = t['gan_H_target']
        self.l_gan_src_w = _opt(t, 'gan_H_source', 0)
        self.ragan = bool(t.get('ragan'))
        self.gan_type = str(t.get('gan_type') or 'vanilla').lower()
        if self.gan_type not in ('vanilla', 'lsgan', 'wgan-gp'):
            raise NotImplementedError('GAN type [{:s}] is not found'.format(self.gan_type))
        self.netG = netG if netG is not None else nets.RRDBNet(g['in_nc'], g['out_nc'], g['nf'], g['nb'], opt['scale'])
        if netG is None:
            nets.init_kaiming_(self.netG, 0.1)
        self.netD = netD
        if self.l_gan_w > 0 and netD is None:
            self.netD = nets.NLayerDiscriminator(d['in_nc'], n_layers=d['n_layers'])  # networks.py:184-185
            nets.init_kaiming_(self.netD, 1)
        self.norm = t.get('norm')
        self.fs_type = t['fs']
        if t['fs'] == 'wavelet':
            self.dwt = nets.HaarDWT()
        elif t['fs'] in ('gau', 'avgpool'):
            k = t['fs_kernel_size']
            self.f_low = nets.FilterLow(k, gaussian=t['fs'] == 'gau')
            self.f_high = nets.FilterHigh(k, gaussian=t['fs'] == 'gau')
        else:
            raise NotImplementedError('FS type [{:s}] not recognized.'.format(t['fs']))
        self.l_pix_w = t['pixel_weight']
        self.l_pix_LL_w = _opt(t, 'pixel_LL_weight', 0)
        self.sup_LL = t.get('sup_LL')
        self.cri_pix = (F.l1_loss if t['pixel_criterion'] == 'l1' else F.mse_loss) if self.l_pix_w > 0 else None
        self.l_fea_w = _opt(t, 'feature_weight', 0)
        self.cri_fea = None
        self.lpips = None
        if self.l_fea_w > 0 and t['feature_criterion'] == 'LPIPS':      # DASR_model.py:97-98,231-233: PerceptualLoss() on the images
            from . import lpips as _lp
            self.lpips = netF if netF is not None else _lp.PerceptualLossLPIPS(_lp.LPIPSAlex(seed=vgg_seed))
        elif self.l_fea_w > 0:
            self.cri_fea = F.l1_loss if t['feature_criterion'] == 'l1' else F.mse_loss
            self.netF = netF if netF is not None else nets.VGGFeatureExtractor(34, seed=vgg_seed)
            self.netF.eval()
        self.G_int = _opt(t, 'G_update_inter', 1)
        self.D_int = _opt(t, 'D_update_inter', 1)
        self.opt_G = torch.optim.Adam(self.netG.parameters(), lr=t['lr_G'], weight_decay=_opt(t, 'weight_decay_G', 0),
                                      betas=(t['beta1_G'], 0.999))
        self.optimizers = [self.opt_G]
        if self.l_gan_w > 0:
            self.opt_D = torch.optim.Adam(self.netD.parameters(), lr=t['lr_D'], weight_decay=_opt(t, 'weight_decay_D', 0),
                                          betas=(t['beta1_D'], 0.999))
            self.optimizers.append(self.opt_D)
        self.netD_src = None
        if self.l_gan_src_w > 0:   # DASR_model.py:45-47,139-143; define_pairD 'discriminator_patch' passes nf as ndf (networks.py:217-218)
            if d.get('which_model_pairD') not in ('discriminator_patch', 'discriminator_vgg_128'):
                raise NotImplementedError('Discriminator model [{:s}] not recognized'.format(str(d.get('which_model_pairD'))))
            self.netD_src = netD_source
            if self.netD_src is None:
                if d['which_model_pairD'] == 'discriminator_vgg_128':    # networks.py:201-202
                    self.netD_src = nets.Discriminator_VGG_128(d['in_nc'], d['nf'])
                else:
                    self.netD_src = nets.NLayerDiscriminator(d['in_nc'], d['nf'], d['n_layers'])
                nets.init_kaiming_(self.netD_src, 1)
            self.netD_src.train()
            self.opt_D_src = torch.optim.Adam(self.netD_src.parameters(), lr=t['lr_D'], weight_decay=_opt(t, 'weight_decay_D', 0),
                                              betas=(t['beta1_D'], 0.999))
            self.optimizers.append(self.opt_D_src)
        self.scheds = [torch.optim.lr_scheduler.MultiStepLR(o, t['lr_steps'], t['lr_gamma']) for o in self.optimizers]
        self.log = OrderedDict()

    # --- frequency separation (DASR_model.py:442-458) ---
    def fs(self, x):
        if self.fs_type == 'wavelet':
            ll, hc = self.dwt(x)
            if self.norm:
                ll, hc = ll * 0.5, hc * 0.5 + 0.5
            return ll, hc
        low, high = self.f_low(x), self.f_high(x)
        if self.norm:
            high = high * 0.5 + 0.5  # double normalisation, App. C-11
        return low, high

    def update_learning_rate(self):
        for s in self.scheds:
            s.step()

    def feed_data(self, data):
        """DASR_model.py:161-179"""
        self.var_L = torch.cat([data['LR_fake'], data['LR_real']], 0)
        self.var_H = torch.cat([data['HR'], data['HR_unpair']], 0)
        hr = data['HR']
        self.weights = F.interpolate(data['fake_w'], size=(hr.shape[2], hr.shape[3]), mode='bilinear', align_corners=False)
        self.n = self.var_L.shape[0] // 2

    def _bce(self, logits, target_val):
        """GANLoss(gan_type, 1.0, 0.0)(logits, target_is_real) (loss.py:8-40); the name is historical: 'vanilla' is BCEWithLogits"""
        if self.gan_type == 'lsgan':
            return F.mse_loss(logits, torch.full_like(logits, target_val))
        if self.gan_type == 'wgan-gp':   # loss.py:21-23; the gradient penalty is built (DASR_model.py:114-118) but never applied
            return -logits.mean() if target_val > 0.5 else logits.mean()
        return F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, target_val))

    def optimize_parameters(self, step):
        n = self.n
        self.fake_H = self.netG(self.var_L)
        fake_LL, fake_Hc = self.fs(self.fake_H)
        real_LL, real_Hc = self.fs(self.var_H)
        fake_src, real_src = self.fake_H[:n], self.var_H[:n]
        if step % self.G_int == 0:
            tot = 0
            if self.cri_pix is not None:
                if self.multiweights:
                    l_pix = self.l_pix_w * torch.mean(self.weights * torch.abs(fake_src - real_src))
                else:
                    l_pix = self.cri_pix(fake_src, real_src)
                tot = tot + self.l_pix_w * l_pix
                self.log['loss/l_g_pix'] = l_pix.item()
                if self.sup_LL:
                    l_ll = self.cri_pix(fake_LL[:n], real_LL[:n])
                    tot = tot + self.l_pix_LL_w * l_ll
                    self.log['loss/l_g_LL_pix'] = l_ll.item()
            if self.lpips is not None:
                l_fea = self.lpips(fake_src, real_src)
                tot = tot + self.l_fea_w * l_fea
                self.log['loss/l_g_fea'] = l_fea.item()
            if self.cri_fea is not None:
                real_fea = self.netF(real_src).detach()
                fake_fea = self.netF(fake_src)
                l_fea = self.cri_fea(fake_fea, real_fea)
                tot = tot + self.l_fea_w * l_fea
                self.log['loss/l_g_fea'] = l_fea.item()
            if self.l_gan_w > 0:
                pred = self.netD(fake_Hc[n:])
                if self.ragan:   # DASR_model.py:240-244: relativistic average form; note the weight enters twice (here and in the total)
                    pr = self.netD(real_Hc[n:]).detach()
                    l_gan = self.l_gan_w * (self._bce(pred - pr.mean(0, keepdim=True), 1.0) + self._bce(pr - pred.mean(0, keepdim=True), 0.0)) / 2
                else:
                    l_gan = self._bce(pred, 1.0)
                tot = tot + self.l_gan_w * l_gan
                self.log['loss/l_g_gan_target_Hf'] = l_gan.item()
            if self.l_gan_src_w > 0:   # DASR_model.py:250-259,316: the WEIGHTED value is what is added and logged
                ps = self.netD_src(fake_Hc[:n])
                if self.ragan:   # DASR_model.py:252-256
                    pr = self.netD_src(real_Hc[:n]).detach()
                    l_src = self.l_gan_src_w * (self._bce(ps - pr.mean(0, keepdim=True), 1.0) + self._bce(pr - ps.mean(0, keepdim=True), 0.0)) / 2
                else:
                    l_src = self.l_gan_src_w * self._bce(ps, 1.0)
                tot = tot + l_src
                self.log['loss/l_g_gan_source_H'] = l_src.item()
            self.opt_G.zero_grad()
            tot.backward()
            self.opt_G.step()
        if step % self.D_int == 0 and self.l_gan_w > 0:
            pr = self.netD(real_Hc[n:].detach())
            pf = self.netD(fake_Hc[n:].detach())
            if self.ragan:   # DASR_model.py:273-275
                l_d = (self._bce(pr - pf.mean(0, keepdim=True), 1.0) + self._bce(pf - pr.mean(0, keepdim=True), 0.0)) / 2
            else:
                l_d = (self._bce(pr, 1.0) + self._bce(pf, 0.0)) / 2
            self.opt_D.zero_grad()
            l_d.backward()
            self.opt_D.step()
            self.log['loss/l_d_target_total'] = l_d.item()
            self.log['disc_Score/D_real_target_H'] = pr.detach().mean().item()
            self.log['disc_Score/D_fake_target_H'] = pf.detach().mean().item()
        if step % self.D_int == 0 and self.l_gan_src_w > 0:   # DASR_model.py:287-303,327-330
            pr = self.netD_src(real_Hc[:n].detach())
            pf = self.netD_src(fake_Hc[:n].detach())
            if self.ragan:   # DASR_model.py:291-293
                l_d = (self._bce(pf - pr.mean(0, keepdim=True), 0.0) + self._bce(pr - pf.mean(0, keepdim=True), 1.0)) / 2
            else:
                l_d = (self._bce(pf, 0.0) + self._bce(pr, 1.0)) / 2
            self.opt_D_src.zero_grad()
            l_d.backward()
            self.opt_D_src.step()
            self.log['loss/l_d_total'] = l_d.item()
            self.log['disc_Score/D_real_source_H'] = pr.detach().mean().item()
            self.log['disc_Score/D_fake_source_H'] = pf.detach().mean().item()
