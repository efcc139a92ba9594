"""DASR_Model: the SRN GAN training step (reference: codes/SRN/models/DASR_model.py:24-460) on the MI355X kernels.

Step (DASR_model.py:192-330), n = dataloader batch per rank, the generator sees 2n crops (first n = fake-LR "source"
samples with HR pairs and a domain-distance map, last n = real-LR "target" samples):
    fake_H = G([LR_fake; LR_real]);  (LL, Hf) = fs(fake_H), fs([HR; HR_unpair])
    L_G = w_pix * [w_pix * mean(W |fake_s - HR_s|)] + w_LL * L1(LL_s) + w_fea * L1(F(fake_s), F(HR_s)) + w_gan * BCE(D(Hf_t(fake)), 1)
    L_D = 0.5 * [BCE(D(Hf_t(real)), 1) + BCE(D(Hf_t(fake)), 0)]
Reference quirks kept on purpose: double pixel weight with `multiweights`, gc ignored, scheduler
stepped before the optimizer, FilterHigh normalised twice under `norm`.  D(Hf_t(fake)) is evaluated once and shared by
the G and the D loss (same weights, same input in the reference too); the reference's discarded D weight-gradients of
the G step are not computed.
"""
import ctypes as C
import logging
import math
import os
from collections import OrderedDict

import torch

from . import _lib
from .engine import BTensor, Op, OpList, NULL_T, Tensor, _stream
from .gan_nets import (NLayerDiscriminatorHIP, DiscriminatorVGG128HIP, VGGFeatureHIP, VGG_MEAN, VGG_STD, nlayer_d_spec, vgg128_spec,
                       vgg128_init_state_dict)
from .init import kaiming_state_dict
from .lpips import load_lpips, lpips_metric
from .models import BaseModel, AdamHIP, MultiStepLR, _define_G

logger = logging.getLogger('base')

A_PIX, A_LL, A_FEA, A_GAN, A_DREAL, A_DFAKE, A_SREAL, A_SFAKE = range(8)
A_GAN_SRC, A_DREAL_SRC, A_DFAKE_SRC, A_SREAL_SRC, A_SFAKE_SRC = range(8, 13)   # source-domain discriminator (gan_H_source > 0)
N_ACC = 16


def _op(kind):
    o = Op()
    o.op = kind
    return o


def _nview(bt, n0):
    """view of images [n0:] of a blocked tensor"""
    v = bt.view()
    return Tensor(v.p + n0 * v.n_stride * bt.esz, v.n_stride, v.cb_stride)


def gaussian_kernel2d(k):
    """GaussianFilter weights (architecture.py:1177-1199)"""
    mean, var = (k - 1) / 2.0, (k / 6.0) ** 2.0
    ax = torch.arange(k, dtype=torch.float32)
    xx = ax.repeat(k).view(k, k)
    g = torch.exp(-((xx - mean) ** 2 + (xx.t() - mean) ** 2) / (2 * var))
    return g / g.sum()


def vgg_random_state_dict(spec, seed):
    """torchvision's VGG init rule (kaiming_normal fan_out / relu, zero bias) from one seeded generator; used when no
    pretrained VGG19 file is supplied (`path.pretrain_model_F`): the weights cannot be downloaded offline."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shape in spec:
        if k.endswith('weight'):
            sd[k] = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[0] * 9))
        else:
            sd[k] = torch.zeros(shape)
    return sd


class DASR_Model(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        t = opt['train']
        self.scale = opt['scale']
        self.multiweights = opt['multiweights']
        if opt['adaptive_weights']:
            raise NotImplementedError('adaptive_weights (DASR_Adaptive_Model) is outside the hot path')
        gan_modes = {'vanilla': 0, 'lsgan': 1, 'wgan-gp': 2}   # GANLoss (loss.py:8-27); the reference never applies its gradient penalty
        if str(t['gan_type']).lower() not in gan_modes:
            raise NotImplementedError('GAN type [{:s}] is not found'.format(str(t['gan_type'])))
        self.gan_mode = gan_modes[str(t['gan_type']).lower()]
        self.ragan = bool(t['ragan'])   # relativistic average GAN: per-pixel batch means of the logits (all-reduced across data-parallel ranks)
        self.l_gan_H_target_w = t['gan_H_target'] or 0
        self.l_gan_H_source_w = (t['gan_H_source'] or 0) if self.is_train else 0
        # the BatchNorm source discriminator (which_model_pairD: discriminator_vgg_128) makes the generator's gradient ill-conditioned in the
        # generator's own weights: with bf16 dense blocks the G gradients sit 1.2e-2 from an exact evaluation whatever the discriminator path does
        # (profiles/r04_bn_probe.txt); f16 storage of the dense blocks (same MFMA rate, 8x finer operands) brings them inside the 1e-2 tolerance.
        # Selected automatically for that configuration unless DASR_RDB_PREC says otherwise.
        rdb_prec = None
        if (self.is_train and self.l_gan_H_source_w > 0 and (opt['network_D'] or {}).get('which_model_pairD') == 'discriminator_vgg_128'
                and not os.environ.get('DASR_RDB_PREC')):
            rdb_prec = 2
            logger.info('which_model_pairD discriminator_vgg_128: dense blocks of the generator in f16 storage (DASR_RDB_PREC=1 restores bf16)')
        self.netG = _define_G(opt, self.device, rdb_prec)
        self.netD_target = None
        if self.is_train and self.l_gan_H_target_w > 0:
            d = opt['network_D']
            if d['which_model_D'] != 'discriminator_patch':
                raise NotImplementedError('Discriminator model [{:s}] not recognized'.format(str(d['which_model_D'])))
            self.netD_target = NLayerDiscriminatorHIP(d['in_nc'], 64, d['n_layers'], device=self.device)  # networks.py:184-185
            spec, _ = nlayer_d_spec(d['in_nc'], 64, d['n_layers'])
            self.netD_target.load_state_dict(kaiming_state_dict(spec, 1))  # init_weights(kaiming, scale=1), networks.py:191
        self.netD_source = None
        if self.is_train and self.l_gan_H_source_w > 0:   # DASR_model.py:45-47 -> define_pairD (networks.py:196-227)
            d = opt['network_D']
            if d['which_model_pairD'] == 'discriminator_patch':
                self.netD_source = NLayerDiscriminatorHIP(d['in_nc'], d['nf'], d['n_layers'], device=self.device)  # networks.py:217-218: nf IS passed here
                spec, _ = nlayer_d_spec(d['in_nc'], d['nf'], d['n_layers'])
                self.netD_source.load_state_dict(kaiming_state_dict(spec, 1))
            elif d['which_model_pairD'] == 'discriminator_vgg_128':   # networks.py:201-202; BatchNorm in training mode, 128 x 128 inputs
                self.netD_source = DiscriminatorVGG128HIP(d['in_nc'], d['nf'], device=self.device)
                sd = vgg128_init_state_dict(vgg128_spec(d['in_nc'], d['nf'])[0], int(t['manual_seed'] or 0) + 2)
                self.netD_source.load_state_dict(sd, strict=False)
            else:
                raise NotImplementedError('Discriminator model [{:s}] not recognized'.format(str(d['which_model_pairD'])))
        self.load()
        self.norm = bool(t['norm'])
        self.fs = t['fs']
        if self.fs == 'wavelet':
            pass
        elif self.fs in ('gau', 'avgpool'):
            k = t['fs_kernel_size']
            w = gaussian_kernel2d(k) if self.fs == 'gau' else torch.full((k, k), 1.0 / (k * k))  # count_include_pad=True
            self.fs_k, self.fs_w = k, w.contiguous().to(self.device)
        else:
            raise NotImplementedError('FS type [{:s}] not recognized.'.format(str(self.fs)))
        if self.is_train:
            self.l_pix_w = t['pixel_weight'] or 0
            if self.l_pix_w > 0 and t['pixel_criterion'] not in ('l1', 'l2'):   # nn.L1Loss / nn.MSELoss (DASR_model.py:76-82)
                raise NotImplementedError('Loss type [{:s}] not recognized.'.format(str(t['pixel_criterion'])))
            self.pix_l2 = t['pixel_criterion'] == 'l2'   # used by the plain pixel term and the LL term; the multiweights pixel term is |.| always (:213-215)
            self.l_pix_LL_w = t['pixel_LL_weight'] or 0
            self.sup_LL = bool(t['sup_LL']) and self.l_pix_w > 0
            self.l_fea_w = t['feature_weight'] or 0
            self.netG.loss_weight = float(max(self.l_pix_w, self.l_fea_w, self.l_gan_H_target_w, self.l_gan_H_source_w) or 1.0)   # f16 dense blocks: sizes TrunkStore.gscale
            self.netF = None
            self.l_fea_type = t['feature_criterion']
            if self.l_fea_w > 0:
                if self.l_fea_type in ('l1', 'l2'):                          # VGG19-54 feature L1 / MSE (DASR_model.py:93-96,105-106)
                    self.netF = VGGFeatureHIP(34, device=self.device, mse_target=(self.l_fea_type == 'l2'))
                    pf = opt['path']['pretrain_model_F']
                    if pf:
                        self.netF.load_state_dict(torch.load(pf, map_location='cpu'), strict=False)
                    elif opt['allow_random_perceptual']:
                        logger.warning('allow_random_perceptual: VGG19-54 uses SEEDED RANDOM weights (torchvision init rule), no path.pretrain_model_F')
                        self.netF.load_state_dict(vgg_random_state_dict(self.netF.spec, int(t['vgg_seed'] or 77)))
                    else:   # the reference always runs torchvision's pretrained VGG19 (architecture.py:1070); it cannot be downloaded offline
                        raise FileNotFoundError('feature_criterion l1 needs pretrained VGG19 weights: set path.pretrain_model_F (torchvision vgg19 '
                                                'state_dict) or allow_random_perceptual: true to train against a seeded random network')
                elif self.l_fea_type == 'LPIPS':                             # PerceptualLoss() = LPIPS(alex) (DASR_model.py:97-98)
                    self.netF = load_lpips(opt, self.device, int(t['vgg_seed'] or 77))
                else:
                    raise NotImplementedError('Loss type [{:s}] not recognized.'.format(str(self.l_fea_type)))
            self.G_update_inter = t['G_update_inter'] or 1
            self.D_update_inter = t['D_update_inter'] or 1
            wd_G = t['weight_decay_G'] or 0
            self.optimizer_G = AdamHIP(self.netG.params, t['lr_G'], (t['beta1_G'], 0.999), wd_G, gate=self.netG.chain_err)
            self.optimizers.append(self.optimizer_G)
            if self.netD_target is not None:
                wd_D = t['weight_decay_D'] or 0
                self.optimizer_D_target = AdamHIP(self.netD_target.params, t['lr_D'], (t['beta1_D'], 0.999), wd_D, gate=self.netG.chain_err)   # (its input is G's output)
                self.optimizers.append(self.optimizer_D_target)
            if self.netD_source is not None:   # DASR_model.py:139-143
                self.optimizer_D_source = AdamHIP(self.netD_source.params, t['lr_D'], (t['beta1_D'], 0.999), t['weight_decay_D'] or 0, gate=self.netG.chain_err)
                self.optimizers.append(self.optimizer_D_source)
            if t['lr_scheme'] != 'MultiStepLR':
                raise NotImplementedError('MultiStepLR learning rate scheme is enough.')
            for o in self.optimizers:
                self.schedulers.append(MultiStepLR(o.lr, t['lr_steps'], t['lr_gamma']))
            self.log_dict = OrderedDict()
            self.acc = torch.zeros(N_ACC, dtype=torch.float32, device=self.device)
        if opt['val_lpips']:   # validation metric (DASR_model.py:158-159): shares the training criterion's network when that is LPIPS
            own = getattr(self, 'netF', None) if getattr(self, 'l_fea_type', None) == 'LPIPS' else None
            self.cri_fea_lpips = own or load_lpips(opt, self.device)
        self._plans = {}

    def networks(self):
        return [self.netG] + [d for d in (self.netD_target, self.netD_source) if d is not None]

    def _lr_of(self, optimizer):
        return self.schedulers[self.optimizers.index(optimizer)].get_lr()

    # ---- data (DASR_model.py:161-187) --------------------------------------------------------------------------
    def feed_data(self, data, istrain=True):
        dev = self.device
        if istrain and 'HR' in data:
            self.var_L = torch.cat([data['LR_fake'], data['LR_real']], 0).to(dev, non_blocking=True)
            self.var_H = torch.cat([data['HR'], data['HR_unpair']], 0).to(dev, non_blocking=True).contiguous()
            self.fake_w = data['fake_w'].to(dev, non_blocking=True).contiguous()
        else:
            self.var_L = data['LR'].to(dev)
            self.needHR = 'HR' in data
            if self.needHR:
                self.var_H = data['HR'].to(dev)

    # ---- plan ----------------------------------------------------------------------------------------------------
    def _plan(self, n, h, w):
        k = (n, h, w)
        if k not in self._plans:
            self._plans[k] = _StepPlan(self, n, h, w)
        return self._plans[k]

    def optimize_parameters(self, step):
        N2, _, h, w = self.var_L.shape
        n = N2 // 2
        P = self._plan(n, h, w)
        do_g = step % self.G_update_inter == 0
        do_d = step % self.D_update_inter == 0 and self.netD_target is not None
        do_ds = step % self.D_update_inter == 0 and self.netD_source is not None
        P.hr_nchw.copy_(self.var_H)
        P.fake_w.copy_(self.fake_w)
        P.g.set_input(self.var_L)
        P.fwd.run()                       # G forward, frequency separation, D forward, VGG forward, all losses + loss gradients
        scale = 1.0
        dp_on = self.dp is not None and self.dp.active
        if dp_on:
            scale = self.dp.grad_scale
            P.g.set_grad_scale(scale)
            P.set_d_grad_scale(scale)
        if do_g:
            if P.ds_run_g.ops:
                P.ds_run_g.run()
            if self.ragan:
                self._run_ragan(P.rg, P, dp_on)   # relativistic GAN terms of the generator loss (value + dL/dlogits of the fake halves)
            P.g_loss_bwd.run()            # D / VGG / fs data-gradients into dL/dSR
            gG = self.netG.params.grad
            if P.g.store.calibrate_due():   # f16 dense blocks (rrdbnet.TrunkStore): measure dL/d(trunk output) behind the HR tail, patch the scale, restart
                P.g.bwd.run(0, P.g.tail_end)
                P.g.store.set_gscale_from(float(P.g.g_t0.t.abs().max().item()))
            if not dp_on:
                P.g.bwd.run()
            else:
                P.g.run_backward_dp(self.dp, gG)   # bucket-wise exchange overlapped with the backward, never across a chained launch
            self.optimizer_G.step(self.schedulers[0].get_lr())
            self.netG.repack()
        if do_d:
            if self.ragan:
                self._run_ragan(P.rd, P, dp_on)
            P.d_step.run()                # BCE(real,1), BCE(fake,0), D backward with weight gradients
            if dp_on:
                self.dp.allreduce_mean(self.netD_target.params.grad)
            self.optimizer_D_target.step(self._lr_of(self.optimizer_D_target))
            self.netD_target.repack()
        if do_ds:                         # source domain (DASR_model.py:287-303)
            if P.ds_run_d.ops:
                P.ds_run_d.run()
            if self.ragan:
                self._run_ragan(P.rs, P, dp_on)
            P.ds_step.run()
            if dp_on:
                self.dp.allreduce_mean(self.netD_source.params.grad)
            self.optimizer_D_source.step(self._lr_of(self.optimizer_D_source))
            self.netD_source.repack()
        self._fake_H, self._fake_plan = None, P.g   # NCHW copy of the SR batch on demand (fake_H property)
        self._acc_snapshot = (self.acc, do_g, do_d, do_ds)
        self._pix_div = getattr(P, 'pix_log_div', 1.0)

    @property
    def fake_H(self):
        if getattr(self, '_fake_H', None) is None and getattr(self, '_fake_plan', None) is not None:
            return self._fake_plan.read_output()
        return getattr(self, '_fake_H', None)

    @fake_H.setter
    def fake_H(self, v):
        self._fake_H, self._fake_plan = v, None

    def _run_ragan(self, lists, P, dp_on):
        """stage 0 -> (all-reduce of the per-pixel logit sums) -> stage 1 -> (all-reduce of the per-pixel sigmoid sums) -> stage 2: the batch
        means of the relativistic loss are means over the GLOBAL batch, as in the reference's single-process nn.DataParallel"""
        for stage in range(3):
            if not lists[stage].ops:
                return
            lists[stage].run()
            if dp_on and stage < 2:
                for sums, part in P.r_bufs:
                    self.dp.all_reduce_here(sums if stage == 0 else part)

    def sync_error_words(self):
        """as SRModel.sync_error_words: every rank, logging interval, in front of check_finite"""
        if getattr(self, 'dp', None) is not None and self.dp.active:
            self.dp.sync_error_words([o.nonfinite for o in self.optimizers] + [self.netG.chain_err])

    def check_finite(self):
        for o, what in zip(self.optimizers, ('generator', 'discriminator', 'source discriminator')):
            o.check_finite(what)
        for plan in self.netG.plans.values():   # chained trunk launches (rrdbnet._Plan.check_chain): a broken neighbour wait invalidates the step
            plan.check_chain()

    def get_current_log(self):
        """one device->host sync, only when the caller logs (the reference syncs 5-9 times every step, App. C-8)"""
        if getattr(self, '_acc_snapshot', None) is not None:
            acc, do_g, do_d, do_ds = self._acc_snapshot
            a = acc.tolist()
            self.check_finite()
            if do_g:
                if self.l_pix_w > 0:
                    self.log_dict['loss/l_g_pix'] = a[A_PIX] / self._pix_div
                    if self.sup_LL:
                        self.log_dict['loss/l_g_LL_pix'] = a[A_LL]
                if self.netF is not None:
                    self.log_dict['loss/l_g_fea'] = a[A_FEA]
                if self.netD_target is not None:
                    self.log_dict['loss/l_g_gan_target_Hf'] = a[A_GAN]
                if self.netD_source is not None:
                    self.log_dict['loss/l_g_gan_source_H'] = a[A_GAN_SRC]   # the reference logs the WEIGHTED value here (DASR_model.py:258,316)
            if do_d:
                self.log_dict['loss/l_d_target_total'] = a[A_DREAL] + a[A_DFAKE]
                self.log_dict['disc_Score/D_real_target_H'] = a[A_SREAL]
                self.log_dict['disc_Score/D_fake_target_H'] = a[A_SFAKE]
            if do_ds:
                self.log_dict['loss/l_d_total'] = a[A_DREAL_SRC] + a[A_DFAKE_SRC]
                self.log_dict['disc_Score/D_real_source_H'] = a[A_SREAL_SRC]
                self.log_dict['disc_Score/D_fake_source_H'] = a[A_SFAKE_SRC]
            self._acc_snapshot = None
        return self.log_dict

    def test(self, tsamples=False):
        """inference on var_L (DASR_model.py:333-345; `chop`: quadrant inference, utils/util.py:87-147)"""
        if self.opt['chop']:
            from .util import forward_chop
            self.fake_H = forward_chop(self.var_L, self.opt['scale'], lambda x: self.netG.forward(x).clone(), min_size=320000)
        else:
            self.fake_H = self.netG.forward(self.var_L).clone()
        if not tsamples and self.opt['val_lpips']:
            self.LPIPS = lpips_metric(self.cri_fea_lpips, self.fake_H, self.var_H)   # DASR_model.py:340-344

    def filter_high(self, x):
        """self.filter_high of the reference (DASR_model.py:58,61-66: FilterHigh(kernel_size=fs_kernel_size, gaussian = fs is not 'avgpool'),
        architecture.py:1228-1243): 0.5 + 0.5 * (x - lowpass(x)), k x k depthwise filter with zero padding, on the device (dasr_lowpass).
        x: NCHW fp32 CUDA tensor -> NCHW fp32 CUDA tensor.  Used by the training-sample visuals (train.py:123-172)."""
        from .engine import BTensor
        N, C_, H, W = x.shape
        k = int(self.opt['train']['fs_kernel_size'] or 5)
        key = ('fh', k, self.fs == 'avgpool')
        if getattr(self, '_fh_w', (None,))[0] != key:
            w = torch.full((k, k), 1.0 / (k * k)) if self.fs == 'avgpool' else gaussian_kernel2d(k)
            self._fh_w = (key, w.contiguous().to(self.device))
        xb, hb = BTensor(N, 16, H, W, True, self.device), BTensor(N, 16, H, W, True, self.device)
        out = torch.empty((N, C_, H, W), dtype=torch.float32, device=self.device)
        L = _lib.lib()
        st = _stream()
        _lib.check(L.dasr_nchw_to_blocked(x.contiguous().data_ptr(), N, C_, H, W, xb.view(), NULL_T, st), 'nchw_to_blocked')
        _lib.check(L.dasr_lowpass(xb.view(), NULL_T, self._fh_w[1].data_ptr(), k, N, C_, H, W, 0, 0.5, 0.5, NULL_T, hb.view(), 0, st), 'lowpass')
        _lib.check(L.dasr_blocked_to_nchw(hb.view(), N, C_, H, W, out.data_ptr(), st), 'blocked_to_nchw')
        return out

    def get_current_visuals(self, need_HR=True, tsamples=False):
        out = OrderedDict()
        out['LR'] = self.var_L.detach()[0].float().cpu()
        if tsamples:   # DASR_model.py:353-357: high-frequency views of the SR batch and of the ground truth
            out['hf'] = self.filter_high(self.fake_H).float().cpu()
            out['gt_hf'] = self.filter_high(self.var_H).float().cpu()
            out['HR'] = self.var_H.detach()[0].float().cpu()
            out['HR_hf'] = out['gt_hf'].clone()
        out['SR'] = self.fake_H.detach().float().cpu() if tsamples else self.fake_H.detach()[0].float().cpu()
        if not tsamples and self.opt['val_lpips']:
            out['LPIPS'] = self.LPIPS.detach().float().cpu()
        if need_HR and getattr(self, 'var_H', None) is not None:
            out['HR'] = self.var_H.detach()[0].float().cpu()
        return out

    def load(self):
        pg = self.opt['path']['pretrain_model_G']
        if pg is not None:
            logger.info('Loading pretrained model for G [{:s}] ...'.format(pg))
            self.load_network(pg, self.netG)
        pd = self.opt['path']['pretrain_model_D_target']
        if self.opt['is_train'] and pd is not None and self.netD_target is not None:
            logger.info('Loading pretrained model for D_target [{:s}] ...'.format(pd))
            self.load_network(pd, self.netD_target)
        ps = self.opt['path']['pretrain_model_D_source']
        if self.opt['is_train'] and ps is not None and self.netD_source is not None:
            logger.info('Loading pretrained model for D_source [{:s}] ...'.format(ps))
            self.load_network(ps, self.netD_source)

    def save(self, iter_step):
        self.save_network(self.netG, 'G', iter_step)
        if self.netD_target is not None:
            self.save_network(self.netD_target, 'D_target', iter_step)
        if self.netD_source is not None:
            self.save_network(self.netD_source, 'D_source', iter_step)


def _ragan_ops(lists, a, b, n, H, W, n_glob, ta, tb, coef, gcoef, sums, part, p_loss, p_sa, p_sb, score_coef, ga, gb, form=0, eps=0.0, stages=(0, 1, 2)):
    """the three dasr_ragan stages of one relativistic loss, appended to lists[0..2] (include/dasr_hip.h); form 1 / eps: the DSN's form"""
    import struct
    for stage in stages:
        o = _op(_lib.OP_RAGAN)
        o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5] = a, b, n, H, W, stage, n_glob, form
        o.l[2] = struct.unpack('<I', struct.pack('<f', eps))[0]
        o.f[0], o.f[1], o.f[2], o.f[3] = ta, tb, coef, gcoef
        o.p[0], o.p[1], o.p[2], o.p[3] = sums.data_ptr(), part.data_ptr(), p_loss, p_sa
        o.l[0] = p_sb or 0
        o.l[1] = struct.unpack('<I', struct.pack('<f', score_coef))[0]
        o.t[2], o.t[3] = ga, gb
        lists[stage].add(o)
    lists[0].keep += [sums, part]


class _StepPlan:
    """All buffers and op lists of one DASR step for (n, h, w)."""

    def __init__(self, m, n, h, w):
        self.m = m
        dev = m.device
        N2, H, W = 2 * n, 4 * h, 4 * w
        self.g = m.netG.plan(N2, h, w)
        g = self.g
        self.hr_nchw = torch.zeros((N2, 3, H, W), dtype=torch.float32, device=dev)
        self.hr_b = BTensor(N2, 16, H, W, True, dev)
        self.fake_w = torch.zeros((n, 1, h, w), dtype=torch.float32, device=dev)
        self.wmap = torch.zeros((n, 1, H, W), dtype=torch.float32, device=dev)
        acc = m.acc.data_ptr()
        wavelet = m.fs == 'wavelet'
        Hd, Wd = (H // 2, W // 2) if wavelet else (H, W)
        fwd, gl = OpList(), OpList()

        def add(ol, o):
            ol.add(o)
            return o

        # ---- forward -------------------------------------------------------------------------------------------
        fwd.extend(g.fwd)
        o = add(fwd, _op(_lib.OP_FILL))
        o.p[0], o.l[0], o.f[0] = acc, N_ACC, 0.0
        o = add(fwd, _op(_lib.OP_NCHW2B))
        o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[0], o.t[1] = self.hr_nchw.data_ptr(), N2, 3, H, W, self.hr_b.view(), NULL_T
        o = add(fwd, _op(_lib.OP_BILINEAR))  # ddm -> HR size (DASR_model.py:173-174)
        o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.p[1] = self.fake_w.data_ptr(), n, h, w, 4, self.wmap.data_ptr()
        o = add(fwd, _op(_lib.OP_FILL))      # dL/dSR accumulates contributions of every loss term
        o.p[0], o.l[0], o.f[0] = g.g_sr.t.data_ptr(), g.g_sr.t.numel(), 0.0
        # pixel loss on the source half
        if m.l_pix_w > 0:
            o = add(fwd, _op(_lib.OP_L1LOSS))
            cnt = float(n * 3 * H * W)
            o.t[0], o.p[0] = g.sr.view(), self.hr_nchw.data_ptr()
            o.p[1] = self.wmap.data_ptr() if m.multiweights else None
            o.i[0], o.i[1], o.i[2], o.i[3], o.i[4] = n, 3, H, W, 1 | (2 if (m.pix_l2 and not m.multiweights) else 0)
            # multiweights: l_g_pix = w * mean(W|d|) and total += w * l_g_pix  (DASR_model.py:213-218)
            self.pix_log_div = 1.0
            if m.multiweights:
                o.f[0] = float(m.l_pix_w) * float(m.l_pix_w) / cnt
                self.pix_log_div = float(m.l_pix_w)
            else:
                o.f[0] = float(m.l_pix_w) / cnt
                self.pix_log_div = float(m.l_pix_w)
            o.p[2], o.t[1] = acc + 4 * A_PIX, g.g_sr.view()
        # frequency separation
        D = m.netD_target
        self.d = D.plan(N2, Hd, Wd) if D is not None else None
        d = self.d
        self.fake_low = BTensor(n, 16, Hd, Wd, True, dev)
        self.real_low = BTensor(n, 16, Hd, Wd, True, dev)
        self.g_low = BTensor(n, 16, Hd, Wd, True, dev)
        dx_fake = d.x.view() if d is not None else NULL_T
        dx_real = _nview(d.x, n) if d is not None else NULL_T
        Ds = m.netD_source                     # source-domain discriminator on the high frequencies of [fake_s ; real_s] (DASR_model.py:250-259)
        if isinstance(Ds, DiscriminatorVGG128HIP) and (Hd, Wd) != (128, 128):
            raise ValueError('Discriminator_VGG_128 needs 128 x 128 inputs (Linear(512 * 4 * 4, 100)), got %d x %d' % (Hd, Wd))
        self.ds = Ds.plan(N2, Hd, Wd) if Ds is not None else None
        ds = self.ds
        # BatchNorm running statistics follow the reference's forwards: D_s(fake) in the G step, D_s(real) then D_s(fake) in the D step
        bn = ds is not None and hasattr(Ds, 'buffers')
        self.ds_run_g = OpList()
        if bn:
            self.ds_run_g.extend(ds.running_ops(0))
            if m.ragan:   # the relativistic G loss also evaluates D_s(real) in training mode (DASR_model.py:253): fake, then real
                self.ds_run_g.extend(ds.running_ops(1))
        self.ds_run_d = OpList()
        if bn:
            self.ds_run_d.extend(ds.running_ops(1))
            self.ds_run_d.extend(ds.running_ops(0))
        sx_fake = ds.x.view() if ds is not None else NULL_T
        sx_real = _nview(ds.x, n) if ds is not None else NULL_T
        if wavelet:
            for src, n0, ll, hc in ((g.sr, 0, self.fake_low.view(), sx_fake), (g.sr, n, NULL_T, dx_fake),
                                    (self.hr_b, 0, self.real_low.view(), sx_real), (self.hr_b, n, NULL_T, dx_real)):
                if ll.p is None and hc.p is None:
                    continue
                o = add(fwd, _op(_lib.OP_DWT_FWD))
                o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.t[1], o.t[2] = _nview(src, n0), n, 3, Hd, Wd, int(m.norm), ll, hc
        else:
            a_h, b_h = (0.25, 0.75) if m.norm else (0.5, 0.5)  # FilterHigh normalises, filter_func normalises again (App. C-11)
            self.ab = (a_h, b_h)
            for src, n0, lo, hi in ((g.sr, 0, self.fake_low.view(), sx_fake), (g.sr, n, NULL_T, dx_fake),
                                    (self.hr_b, 0, self.real_low.view(), sx_real), (self.hr_b, n, NULL_T, dx_real)):
                if lo.p is None and hi.p is None:
                    continue
                o = add(fwd, _op(_lib.OP_LOWPASS))
                o.t[0], o.t[1], o.p[0], o.i[4] = _nview(src, n0), NULL_T, m.fs_w.data_ptr(), m.fs_k
                o.i[0], o.i[1], o.i[2], o.i[3], o.i[5], o.i[6] = n, 3, H, W, 0, 0
                o.f[0], o.f[1], o.t[2], o.t[3] = a_h, b_h, lo, hi
        # LL loss (source half)
        if m.sup_LL:
            o = add(fwd, _op(_lib.OP_L1DIFF))
            cnt = float(n * 3 * Hd * Wd)
            o.t[0], o.t[1], o.i[4] = self.fake_low.view(), self.real_low.view(), 1 | (2 if m.pix_l2 else 0)
            o.i[0], o.i[1], o.i[2], o.i[3] = n, 3, Hd, Wd
            o.f[0], o.f[1], o.p[0], o.t[2] = 1.0 / cnt, float(m.l_pix_LL_w) / cnt, acc + 4 * A_LL, self.g_low.view()
        # VGG feature loss (source half): batch [fake_s; real_s]
        self.v = None
        lpips = m.netF is not None and m.l_fea_type == 'LPIPS'
        if lpips:                                # LPIPS(fake_s, real_s).mean() (DASR_model.py:231-233): loss and head gradients in the forward list
            self.v = m.netF.plan(N2, n, H, W)
            fwd.add(self.v.input_op(g.sr.view(), 0, n))
            fwd.add(self.v.input_op(self.hr_b.view(), n, n))
            fwd.extend(self.v.fwd)
            for o in self.v.head_ops(acc + 4 * A_FEA, float(m.l_fea_w)):
                fwd.add(o)
        elif m.netF is not None:
            self.v = m.netF.plan(N2, n, H, W)
            v = self.v
            sc = [1.0 / s for s in VGG_STD] + [0.0]
            sh = [-mu / s for mu, s in zip(VGG_MEAN, VGG_STD)] + [0.0]
            for src, dst in ((g.sr.view(), v.x.view()), (self.hr_b.view(), _nview(v.x, n))):
                o = add(fwd, _op(_lib.OP_AFFINE4))
                o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1], o.i[4], o.i[5] = src, n, 3, H, W, dst, v.x_flag, 0
                for j in range(4):
                    o.f[j] = sc[j]
                C.memmove(C.addressof(o.l), (C.c_float * 4)(*sh), 16)
            fwd.extend(v.fwd)
            o = add(fwd, _op(_lib.OP_L1DIFF))
            f = v.feat
            cnt = float(n * f.C * f.H * f.W)
            o.t[0], o.t[1], o.i[4] = f.view(), _nview(f, n), 1 | (2 if m.l_fea_type == 'l2' else 0)
            o.i[0], o.i[1], o.i[2], o.i[3] = n, f.C, f.H, f.W
            o.f[0], o.f[1], o.p[0], o.t[2] = 1.0 / cnt, float(m.l_fea_w) / cnt, acc + 4 * A_FEA, v.g_feat.view()
        # discriminator forward on [fake_t; real_t] and the generator's GAN loss
        if d is not None:
            fwd.extend(d.fwd)
            lg = d.logits
            cnt = float(n * 1 * lg.H * lg.W)
            if not m.ragan:
                o = add(fwd, _op(_lib.OP_BCE))
                o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4] = lg.view(), n, 1, lg.H, lg.W, m.gan_mode
                o.f[0], o.f[1], o.f[2], o.p[0], o.p[1], o.f[3], o.t[1] = 1.0, 1.0 / cnt, float(m.l_gan_H_target_w) / cnt, acc + 4 * A_GAN, None, 0.0, d.g_logits.view()
        if ds is not None:   # l_g_gan_source_Hf = w_src * BCE(D_s(fake_s), 1): value logged WITH the weight (DASR_model.py:258,316)
            fwd.extend(ds.fwd)
            lg = ds.logits
            cnt = float(n * 1 * lg.H * lg.W)
            if not m.ragan:
                o = add(fwd, _op(_lib.OP_BCE))
                o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4] = lg.view(), n, 1, lg.H, lg.W, m.gan_mode
                o.f[0], o.f[1], o.f[2], o.p[0], o.p[1], o.f[3], o.t[1] = 1.0, float(m.l_gan_H_source_w) / cnt, float(m.l_gan_H_source_w) / cnt, \
                    acc + 4 * A_GAN_SRC, None, 0.0, ds.g_logits.view()
        # relativistic average form (`ragan`, DASR_model.py:240-244,252-256): three stages per loss, the per-pixel batch sums are all-reduced between
        # them under data parallelism (DASR_Model._run_ragan).  Weights: target w * (..)/2 enters the total times w AGAIN (w^2 on the gradient, w on
        # the logged value); source w * (..)/2 is added as is.
        self.rg = [OpList(), OpList(), OpList()]
        self.rd = [OpList(), OpList(), OpList()]
        self.rs = [OpList(), OpList(), OpList()]
        self.r_bufs = []
        if m.ragan:
            world = m.dp.world if (getattr(m, 'dp', None) is not None and m.dp.active) else 1
            for D_, slot_g, w_log, w_grad, lists_d, slots_d in (
                    (d, A_GAN, m.l_gan_H_target_w, m.l_gan_H_target_w * m.l_gan_H_target_w, self.rd, (A_DREAL, A_SREAL, A_SFAKE)),
                    (ds, A_GAN_SRC, m.l_gan_H_source_w, m.l_gan_H_source_w, self.rs, (A_DREAL_SRC, A_SREAL_SRC, A_SFAKE_SRC))):
                if D_ is None:
                    continue
                lg = D_.logits
                hw = lg.H * lg.W
                cnt = float(n * hw)
                sums, part = torch.zeros(2 * hw, dtype=torch.float32, device=dev), torch.zeros(2 * hw, dtype=torch.float32, device=dev)
                self.r_bufs.append((sums, part))
                fake, real = lg.view(), _nview(lg, n)
                g_fake, g_real = D_.g_logits.view(), _nview(D_.g_logits, n)
                # generator: a = fake (target 1, gradient), b = real (target 0, detached)
                _ragan_ops(self.rg, fake, real, n, lg.H, lg.W, n * world, 1.0, 0.0, 0.5 * float(w_log) / cnt, 0.5 * float(w_grad) / cnt, sums, part,
                           acc + 4 * slot_g, None, None, 0.0, g_fake, NULL_T, form=(0, 2, 3)[m.gan_mode])
                # discriminator: a = real (target 1), b = fake (target 0), both carry gradient; the whole loss goes to the "real" slot
                _ragan_ops(lists_d, real, fake, n, lg.H, lg.W, n * world, 1.0, 0.0, 0.5 / cnt, 0.5 / cnt, sums, part,
                           acc + 4 * slots_d[0], acc + 4 * slots_d[1], acc + 4 * slots_d[2], 1.0 / cnt, g_real, g_fake, form=(0, 2, 3)[m.gan_mode])
        self.fwd = fwd

        # ---- generator-loss backward: everything that lands in dL/dSR ----------------------------------------------------
        if lpips:
            gl.extend(self.v.bwd)
            gl.add(self.v.adjoint_op(g.g_sr.view()))   # accumulated into dL/dSR (source half = first n images)
        elif self.v is not None:
            v = self.v
            gl.extend(v.bwd)
            o = add(gl, _op(_lib.OP_AFFINE4))  # adjoint of the input normalisation, accumulated into dL/dSR (source half)
            o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1], o.i[4], o.i[5] = v.gx.view(), n, 3, H, W, g.g_sr.view(), 1, 1
            for j in range(4):
                o.f[j] = ([1.0 / s for s in VGG_STD] + [0.0])[j]
            C.memmove(C.addressof(o.l), (C.c_float * 4)(0.0, 0.0, 0.0, 0.0), 16)
        if d is not None:
            gl.extend(d.bwd_data_ops(n))
        if ds is not None:
            gl.extend(ds.bwd_data_ops(n))
        g_hi_t = d.gx.view() if d is not None else NULL_T
        g_hi_s = ds.gx.view() if ds is not None else NULL_T
        g_lo_s = self.g_low.view() if m.sup_LL else NULL_T
        if wavelet:
            for n0, gll, ghc in ((0, g_lo_s, g_hi_s), (n, NULL_T, g_hi_t)):
                if gll.p is None and ghc.p is None:
                    continue
                o = add(gl, _op(_lib.OP_DWT_BWD))
                o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.t[2], o.i[5] = gll, ghc, n, 3, Hd, Wd, int(m.norm), _nview(g.g_sr, n0), 1
        else:
            for n0, glo, ghi in ((0, g_lo_s, g_hi_s), (n, NULL_T, g_hi_t)):
                if glo.p is None and ghi.p is None:
                    continue
                o = add(gl, _op(_lib.OP_LOWPASS))
                o.t[0], o.t[1], o.p[0], o.i[4] = glo, ghi, m.fs_w.data_ptr(), m.fs_k
                o.i[0], o.i[1], o.i[2], o.i[3], o.i[5], o.i[6] = n, 3, H, W, 1, 1
                o.f[0], o.f[1], o.t[2], o.t[3] = self.ab[0], 0.0, _nview(g.g_sr, n0), NULL_T
        self.g_loss_bwd = gl

        # ---- discriminator step: BCE(real, 1), BCE(fake, 0), backward with weight gradients -----------------------------------
        dstep = OpList()
        if d is not None:
            lg = d.logits
            cnt = float(n * lg.H * lg.W)
            for n0, target, a_loss, a_score in (() if m.ragan else ((n, 1.0, A_DREAL, A_SREAL), (0, 0.0, A_DFAKE, A_SFAKE))):
                o = add(dstep, _op(_lib.OP_BCE))
                o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4] = _nview(lg, n0), n, 1, lg.H, lg.W, m.gan_mode
                o.f[0], o.f[1], o.f[2] = target, 0.5 / cnt, 0.5 / cnt
                o.p[0], o.p[1], o.f[3], o.t[1] = acc + 4 * a_loss, acc + 4 * a_score, 1.0 / cnt, _nview(d.g_logits, n0)
            dstep.extend(d.bwd_full)
        self.d_step = dstep
        # ---- source-domain discriminator step (DASR_model.py:287-303): the same on [fake_s ; real_s] ------------------------------------
        sstep = OpList()
        if self.ds is not None:
            lg = self.ds.logits
            cnt = float(n * lg.H * lg.W)
            for n0, target, a_loss, a_score in (() if m.ragan else ((n, 1.0, A_DREAL_SRC, A_SREAL_SRC), (0, 0.0, A_DFAKE_SRC, A_SFAKE_SRC))):
                o = add(sstep, _op(_lib.OP_BCE))
                o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4] = _nview(lg, n0), n, 1, lg.H, lg.W, m.gan_mode
                o.f[0], o.f[1], o.f[2] = target, 0.5 / cnt, 0.5 / cnt
                o.p[0], o.p[1], o.f[3], o.t[1] = acc + 4 * a_loss, acc + 4 * a_score, 1.0 / cnt, _nview(self.ds.g_logits, n0)
            sstep.extend(self.ds.bwd_full)
        self.ds_step = sstep

    def set_d_grad_scale(self, scale):
        for ol in (self.d_step, self.ds_step):
            for o in ol.ops:
                if o.op == _lib.OP_WGRAD_REDUCE and o.f[0] != scale:
                    o.f[0] = scale
                    ol._arr = None
                if o.op == _lib.OP_BNORM_BWD and o.f[1] != scale:   # dgamma / dbeta of the BatchNorm layers
                    o.f[1] = scale
                    ol._arr = None
