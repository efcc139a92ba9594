"""GPU parity of the GAN-step kernels, the patch discriminator, the VGG19-54 extractor and the full DASR_Model step
against the oracle (fp32 CPU) and the reference fixtures.  Tolerances: activations 1e-3, gradients 1e-2 (north_star)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ACT_TOL, GRAD_TOL = 1e-3, 1e-2


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from dasr_amd import engine
    engine.ensure_runtime_ready()
    return torch.device('cuda')


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def to_blocked(x, dev, f32=True):
    from dasr_amd.engine import BTensor
    N, Cc, H, W = x.shape
    b = BTensor(N, Cc, H, W, f32, dev)
    xp = torch.zeros(N, b.planes * 16, H, W)
    xp[:, :Cc] = x
    b.t.copy_(xp.view(N, b.planes, 16, H, W).permute(0, 1, 3, 4, 2).to(b.t.dtype))
    return b


def test_instance_norm_lrelu_fwd_bwd():
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, _stream
    L = _lib.lib()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 48, 13, 17, generator=g) * 2 + 0.5
    go = torch.randn(2, 48, 13, 17, generator=g)
    xb, gb = to_blocked(x, dev), to_blocked(go, dev)
    y, gx = BTensor(2, 48, 13, 17, True, dev), BTensor(2, 48, 13, 17, True, dev)
    st = torch.zeros(2 * 48 * 2, device=dev)
    _lib.check(L.dasr_inorm_lrelu_fwd(xb.view(), 2, 48, 13, 17, 1e-5, 0.2, y.view(), st.data_ptr(), _stream()))
    _lib.check(L.dasr_inorm_lrelu_bwd(y.view(), gb.view(), 2, 48, 13, 17, 0.2, st.data_ptr(), gx.view(), _stream()))
    xr = x.double().requires_grad_(True)
    yr = F.leaky_relu(F.instance_norm(xr, eps=1e-5), 0.2)
    yr.backward(go.double())
    assert rel(y.nchw().cpu(), yr.detach().float()) < 1e-5
    assert rel(gx.nchw().cpu(), xr.grad.float()) < 1e-4


def test_bce_dwt_lowpass_pool_misc():
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, _stream, NULL_T
    from oracle import nets
    L = _lib.lib()
    g = torch.Generator().manual_seed(2)
    # BCE with logits
    x = torch.randn(3, 1, 9, 11, generator=g) * 3
    xb = to_blocked(x, dev)
    gr = BTensor(3, 16, 9, 11, True, dev)
    acc = torch.zeros(4, device=dev)
    cnt = x.numel()
    _lib.check(L.dasr_bce_logits(xb.view(), 3, 1, 9, 11, 1.0, 1.0 / cnt, 0.01 / cnt, acc.data_ptr(), acc.data_ptr() + 4, 1.0 / cnt, gr.view(), _stream()))
    xr = x.clone().requires_grad_(True)
    l = F.binary_cross_entropy_with_logits(xr, torch.ones_like(xr))
    (0.01 * l).backward()
    assert abs(float(acc[0]) - float(l)) < 1e-5 and abs(float(acc[1]) - float(x.mean())) < 1e-5
    assert rel(gr.nchw(1).cpu(), xr.grad) < 1e-5
    # GANLoss 'lsgan' (MSE) and 'wgan-gp' (-mean / +mean) against both labels (loss.py:8-40)
    for mode, target, fn in ((1, 1.0, lambda v: F.mse_loss(v, torch.ones_like(v))), (1, 0.0, lambda v: F.mse_loss(v, torch.zeros_like(v))),
                             (2, 1.0, lambda v: -v.mean()), (2, 0.0, lambda v: v.mean())):
        acc.zero_()
        _lib.check(L.dasr_gan_loss(xb.view(), 3, 1, 9, 11, mode, target, 1.0 / cnt, 0.01 / cnt, acc.data_ptr(), acc.data_ptr() + 4, 1.0 / cnt, gr.view(), _stream()))
        xr = x.clone().requires_grad_(True)
        l = fn(xr)
        (0.01 * l).backward()
        assert abs(float(acc[0]) - float(l)) < 1e-5 * max(1.0, abs(float(l))) and abs(float(acc[1]) - float(x.mean())) < 1e-5, (mode, target)
        assert rel(gr.nchw(1).cpu(), xr.grad) < 1e-5, (mode, target)
    assert L.dasr_gan_loss(xb.view(), 3, 1, 9, 11, 3, 1.0, 1.0, 1.0, acc.data_ptr(), None, 0.0, gr.view(), _stream()) != 0   # unknown gan_type
    # Haar DWT forward / adjoint
    img = torch.rand(2, 3, 16, 24, generator=g)
    ib = to_blocked(img, dev)
    ll, hc, gx = BTensor(2, 16, 8, 12, True, dev), BTensor(2, 16, 8, 12, True, dev), BTensor(2, 16, 16, 24, True, dev)
    _lib.check(L.dasr_dwt_fwd(ib.view(), 2, 3, 8, 12, 1, ll.view(), hc.view(), _stream()))
    ir = img.clone().requires_grad_(True)
    rl, rh = nets.HaarDWT()(ir)
    rl, rh = rl * 0.5, rh * 0.5 + 0.5
    assert rel(ll.nchw(3).cpu(), rl.detach()) < 1e-6 and rel(hc.nchw(9).cpu(), rh.detach()) < 1e-6
    gl, gh = torch.randn(2, 3, 8, 12, generator=g), torch.randn(2, 9, 8, 12, generator=g)
    (rl * gl).sum().backward(retain_graph=True)
    (rh * gh).sum().backward()
    glb, ghb = to_blocked(gl, dev), to_blocked(gh, dev)  # keep the buffers alive: a dasr_tensor is a raw pointer
    _lib.check(L.dasr_dwt_bwd(glb.view(), ghb.view(), 2, 3, 8, 12, 1, gx.view(), 0, _stream()))
    assert rel(gx.nchw(3).cpu(), ir.grad) < 1e-6
    # gaussian low/high split and its adjoint
    for k in (5, 9):
        w = nets.gaussian_kernel2d(k)
        wd = w.contiguous().to(dev)
        img = torch.rand(2, 3, 20, 28, generator=g)
        ib = to_blocked(img, dev)
        lo, hi, gxx = BTensor(2, 16, 20, 28, True, dev), BTensor(2, 16, 20, 28, True, dev), BTensor(2, 16, 20, 28, True, dev)
        _lib.check(L.dasr_lowpass(ib.view(), NULL_T, wd.data_ptr(), k, 2, 3, 20, 28, 0, 0.25, 0.75, lo.view(), hi.view(), 0, _stream()))
        ir = img.clone().requires_grad_(True)
        rlo = nets.FilterLow(k, gaussian=True)(ir)
        rhi = nets.FilterHigh(k, gaussian=True)(ir) * 0.5 + 0.5
        assert rel(lo.nchw(3).cpu(), rlo.detach()) < 1e-5 and rel(hi.nchw(3).cpu(), rhi.detach()) < 1e-5
        g1, g2 = torch.randn(2, 3, 20, 28, generator=g), torch.randn(2, 3, 20, 28, generator=g)
        ((rlo * g1).sum() + (rhi * g2).sum()).backward()
        g1b, g2b = to_blocked(g1, dev), to_blocked(g2, dev)
        _lib.check(L.dasr_lowpass(g1b.view(), g2b.view(), wd.data_ptr(), k, 2, 3, 20, 28, 1, 0.25, 0.0, gxx.view(), NULL_T, 0, _stream()))
        assert rel(gxx.nchw(3).cpu(), ir.grad) < 1e-5
    # max-pool forward / backward (with the ReLU' of the producer)
    # (13 x 17: odd input sizes -- nn.MaxPool2d floors, the last row / column takes no part and gets no gradient; round 3: mis-addressed before)
    for Hi, Wi in ((12, 16), (13, 17), (5, 5)):
        Ho, Wo = Hi // 2, Wi // 2
        pb, gab = BTensor(2, 32, Ho, Wo, True, dev), BTensor(2, 32, Hi, Wi, True, dev)
        pre = torch.randn(2, 32, Hi, Wi, generator=torch.Generator().manual_seed(2)).requires_grad_(True)
        ar = F.relu(pre)
        pr = F.max_pool2d(ar, 2)
        gp = torch.randn(2, 32, Ho, Wo, generator=g)
        ab2 = to_blocked(ar.detach(), dev)
        pr.backward(gp)
        _lib.check(L.dasr_maxpool2(ab2.view(), 1, 2, 32, Ho, Wo, pb.view(), Wi, _stream()))
        assert torch.equal(pb.nchw().cpu(), pr.detach()), (Hi, Wi)
        gpb = to_blocked(gp, dev)
        _lib.check(L.dasr_maxpool2_bwd(ab2.view(), gpb.view(), 1, 2, 32, Ho, Wo, gab.view(), 1, Wi, _stream()))
        assert rel(gab.nchw().cpu(), pre.grad) < 1e-6, (Hi, Wi)
    # bilinear x4 of the ddm
    wm = torch.rand(2, 1, 7, 9, generator=g)
    dst = torch.zeros(2, 1, 28, 36, device=dev)
    wmd = wm.to(dev)
    _lib.check(L.dasr_bilinear_up(wmd.data_ptr(), 2, 7, 9, 4, dst.data_ptr(), _stream()))
    assert torch.allclose(dst.cpu(), F.interpolate(wm, size=(28, 36), mode='bilinear', align_corners=False), atol=1e-6)
    # L1 between feature maps + VGG affine
    fa, fb = torch.randn(2, 40, 5, 6, generator=g), torch.randn(2, 40, 5, 6, generator=g)
    ga = BTensor(2, 40, 5, 6, True, dev)
    acc.zero_()
    fab, fbb = to_blocked(fa, dev), to_blocked(fb, dev)
    _lib.check(L.dasr_l1_diff(fab.view(), fbb.view(), 1, 2, 40, 5, 6, 1.0 / fa.numel(), 2.0 / fa.numel(), acc.data_ptr(), ga.view(), _stream()))
    assert abs(float(acc[0]) - float((fa - fb).abs().mean())) < 1e-6
    assert rel(ga.nchw().cpu(), 2.0 * torch.sign(fa - fb) / fa.numel()) < 1e-6


@pytest.mark.parametrize('nc,hw', [(9, 64), (3, 72)])
def test_discriminator_forward_backward(nc, hw):
    dev = _gpu()
    from dasr_amd.gan_nets import NLayerDiscriminatorHIP
    from dasr_amd.engine import OpList
    from dasr_amd import _lib
    from oracle import nets, fixtures
    N = 4
    ref = nets.NLayerDiscriminator(nc, n_layers=2)
    sd = fixtures.seeded_state_dict(ref.state_dict(), 10 + nc, 1.0)
    ref.load_state_dict(sd)
    D = NLayerDiscriminatorHIP(nc, 64, 2, device=dev)
    D.load_state_dict(sd)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(N, nc, hw, hw, generator=g)
    p = D.plan(N, hw, hw)
    p.x.t.copy_(to_blocked(x, dev).t)
    p.fwd.run()
    xr = x.clone().requires_grad_(True)
    y = ref(xr)
    assert tuple(y.shape[2:]) == (p.logits.H, p.logits.W)
    assert rel(p.logits.nchw(1).cpu(), y.detach()) < ACT_TOL
    gl = torch.randn(y.shape, generator=g)
    p.g_logits.t.copy_(to_blocked(gl, dev).t)
    (y * gl).sum().backward()
    p.bwd_full.run()
    gd = D.params.grad_dict()
    for (k, gv), pr in zip(gd.items(), ref.parameters()):
        assert rel(gv, pr.grad) < GRAD_TOL, (k, rel(gv, pr.grad))
    # data gradient of the first 2 images (generator step)
    p.g_logits.t.copy_(to_blocked(gl, dev).t)
    p.bwd_data_ops(2).run()
    assert rel(p.gx.nchw(nc).cpu()[:2], xr.grad[:2]) < GRAD_TOL


# prec 3: the default (split-bf16, ~fp32).  prec 2: the opt-in f16-storage path (one MFMA pass): operand rounding of 2^-12 over 16 un-damped
# layers lands AT the activation tolerance, and the gradient is far outside its tolerance: max-pool routes the gradient to the arg-max of each
# 2x2 window, and values rounded to 11 bits tie or swap order in ~1e-3 of the windows (normwise error ~ sqrt of that fraction; even the
# ~fp32 path shows 6.5e-3 from this mechanism).  The path is checked against its own documented bounds so that it stays correct; it is
# NOT what the trainers use by default.
@pytest.mark.parametrize('prec,act_tol,grad_tol', [(5, ACT_TOL, GRAD_TOL), (55, ACT_TOL, GRAD_TOL), (3, ACT_TOL, GRAD_TOL), (4, ACT_TOL, GRAD_TOL), (2, 2.5e-3, 0.2)],
                         ids=['split_f16_tensors', 'split_f16_tensors_3pass_bwd', 'split_bf16', 'split_f16', 'f16_storage_optin'])
def test_vgg_forward_and_input_gradient(prec, act_tol, grad_tol, margins, size=64):
    dev = _gpu()
    from dasr_amd.gan_nets import VGGFeatureHIP
    from oracle import nets
    ref = nets.VGGFeatureExtractor(34, seed=77)
    bwd_prec = 5 if prec == 55 else None   # 55: prec 5 with the three-pass data gradient on split gradients (the default is one f16 pass)
    prec = 5 if prec == 55 else prec
    V = VGGFeatureHIP(34, device=dev, prec=prec, bwd_prec=bwd_prec)
    V.load_state_dict({k: v for k, v in ref.state_dict().items() if k.startswith('features')})
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, size, size, generator=g)
    xn = ((x - ref.mean) / ref.std)
    p = V.plan(2, 1, size, size)
    if V.split:   # prec 5 (the default since round 3): split tensor, hi plane + remainder plane
        xb = to_blocked(xn, dev).t
        p.x.t[:, :1] = xb.half()
        p.x.t[:, 1:] = (xb - xb.half().float()).half()
    else:
        p.x.t.copy_(to_blocked(xn, dev).t.to(p.x.t.dtype))
    p.fwd.run()
    xr = x.clone().requires_grad_(True)
    f = ref(xr)
    # image 0 carries the gradient (three-pass operands); image 1 is a no-gradient target: ONE f16 pass by default in the split modes (round 3,
    # VGGFeatureHIP.nograd_prec): its features are held to the f16 path's documented bound, and only enter an L1 / MSE as the target
    e_f = rel(p.feat.nchw().cpu()[:1], f.detach()[:1])
    e_t = rel(p.feat.nchw().cpu()[1:], f.detach()[1:])
    margins('VGG19-54 prec %d: no-gradient (target) image features rel err %.2e (bound 2.5e-3; one f16 pass when prec is 3 / 4)' % (prec, e_t))
    assert e_t < 2.5e-3
    # dL/dfeat at the magnitude a mean feature loss produces (~1 / element count): the f16 path pre-scales gradients by a power of two
    # chosen for that magnitude (a unit-scale random gradient would overflow f16 in the middle of the stack)
    gf = torch.randn(f.shape, generator=g) / f[:1].numel()
    p.g_feat.t.copy_(to_blocked(gf, dev).t)
    (f[:1] * gf[:1]).sum().backward()
    p.bwd.run()
    got = p.gx.nchw(3).cpu()[:1] / ref.std  # adjoint of the input normalisation
    e_g = rel(got, xr.grad[:1])
    margins('VGG19-54 prec %d (bwd %d): feature rel err %.2e (tol %.1e), input-gradient rel err %.2e (tol %.1e)' % (prec, V.bwd_prec, e_f, act_tol, e_g, grad_tol))
    assert e_f < act_tol and e_g < grad_tol


@pytest.mark.parametrize('prec', [5, 4], ids=['split_f16_tensors', 'split_f16'])
def test_vgg_on_odd_sizes(prec, margins):
    """40 x 40 input: the pools see 40 -> 20 -> 10 -> 5 -> 2, i.e. one ODD input (nn.MaxPool2d floors: row / column 4 of the 5 x 5 map is dropped).
    The DSN's perceptual loss runs at this size for 160 x 160 crops (fixture dsn_avg5_inst_b1_160)."""
    test_vgg_forward_and_input_gradient(prec, ACT_TOL, GRAD_TOL, margins, size=40)


@pytest.mark.parametrize('case', ['dasr_wavelet_nf32_nb2_n2_32', 'dasr_gau9_nf64_nb1_n1_32', 'dasr_lpips_wavelet_nf32_nb2_n2_32',
                                  'dasr_srcD_wavelet_nf32_nb2_n2_32', 'dasr_ragan_wavelet_nf32_nb1_n3_32', 'dasr_srcVGG128_gau5_nf32_nb1_n3_32',
                                  'dasr_srcVGG128_gau5_nf32_nb1_n3_32+bf16', 'dasr_wavelet_nf32_nb2_n2_32+f16',
                                  'dasr_lsgan_wavelet_nf32_nb1_n2_32', 'dasr_wgan_gau9_nf32_nb1_n2_32', 'dasr_ragan_lsgan_wavelet_nf32_nb1_n3_32',
                                  'dasr_l2_wavelet_nf32_nb1_n2_32', 'dasr_wavelet_nf64_nb23_n1_32'])
def test_dasr_step_matches_oracle_and_reference_fixture(case, golden_dir, margins, monkeypatch):
    """round 3 additions: pixel_criterion / feature_criterion 'l2' (MSE; multiweights off so the plain pixel term goes through cri_pix), and the GAN
    step at the full ESRGAN depth nb = 23"""
    dev = _gpu()
    case_id = case
    if case.endswith('+bf16'):     # the BatchNorm case with the generator's dense blocks forced back to bf16 (default for it: f16 storage, dasr_model.py)
        case = case[:-5]
        monkeypatch.setenv('DASR_RDB_PREC', '1')
    elif case.endswith('+f16'):    # an ordinary GAN fixture with f16 dense blocks
        case = case[:-4]
        monkeypatch.setenv('DASR_RDB_PREC', '2')
    GRAD_TOL = globals()['GRAD_TOL']
    G_TOL = VGG128_STEP_G_TOL if case_id.endswith('+bf16') else GRAD_TOL   # see the note at VGG128_STEP_G_TOL: the generator's bf16 operands on an ill-conditioned case
    torch.set_num_threads(8)
    from oracle import fixtures, nets, trainers
    from dasr_amd import options
    from dasr_amd.models import create_model
    c = fixtures.CASES[case]
    # relativistic loss / Wasserstein loss: shifting every logit by a constant changes nothing (wgan: -mean(real) + mean(fake)), so the TRUE
    # gradient of the discriminators' last bias is 0 (rounding noise on both sides)
    zero_last_bias = bool(c.get('ragan')) or c.get('gan_type') == 'wgan-gp'
    opt = fixtures.make_opt(case)
    netG = nets.RRDBNet(3, 3, c['nf'], c['nb'], 4)
    sdG = fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1)
    netG.load_state_dict(sdG)
    netD = nets.NLayerDiscriminator(c['d_in_nc'], n_layers=2)
    sdD = fixtures.seeded_state_dict(netD.state_dict(), 2, 1.0)
    netD.load_state_dict(sdD)
    crit, sdF = None, None
    if c.get('fea') == 'LPIPS':     # feature_criterion LPIPS: the reference's linear heads + the seeded stand-in AlexNet of the fixture
        from oracle import lpips
        crit, sdF = lpips.golden_criterion(77, golden_dir)
    netD2 = sdD2 = None
    if c.get('gan_src', 0) > 0:     # source-domain patch discriminator (define_pairD passes nf = 64 as ndf)
        netD2 = nets.Discriminator_VGG_128(c['d_in_nc'], 64) if c.get('pairD') == 'discriminator_vgg_128' else nets.NLayerDiscriminator(c['d_in_nc'], 64, n_layers=2)
        sdD2 = fixtures.seeded_state_dict(netD2.state_dict(), 3, 1.0)
        netD2.load_state_dict(sdD2)
    t = trainers.DASRTrainer(opt, netG=netG, netD=netD, netF=crit, vgg_seed=77, netD_source=netD2)
    batch = fixtures.make_batch(case)
    t64 = None
    if 'VGG128' in case:
        # ill-conditioned case (BatchNorm discriminator on high-frequency maps): the fp32 oracle is itself 3e-3 / 1.2e-2 away from exact arithmetic,
        # so the HIP gradients are ALSO compared with the same oracle run in fp64, against the north_star's 1e-2
        import copy
        g64, d64, s64 = copy.deepcopy(netG).double(), copy.deepcopy(netD).double(), copy.deepcopy(netD2).double()
        t64 = trainers.DASRTrainer(fixtures.make_opt(case), netG=g64, netD=d64, netF=None, vgg_seed=77, netD_source=s64)
        for v in vars(t64).values():
            if isinstance(v, torch.nn.Module):
                v.double()
        t64.update_learning_rate()
        t64.feed_data({k: v.double() for k, v in batch.items()})
        t64.optimize_parameters(1)
    opt2 = fixtures.make_opt(case)
    opt2['gpu_ids'] = [0]
    opt2['train']['vgg_seed'] = 77
    m = create_model(options.dict_to_nonedict(opt2))
    m.netG.load_state_dict(sdG)
    m.netD_target.load_state_dict(sdD)
    if sdD2 is not None:
        m.netD_source.load_state_dict(sdD2)
    m.netF.load_state_dict(sdF if sdF is not None else {k: v for k, v in t.netF.state_dict().items() if k.startswith('features')})
    gold = np.load(os.path.join(golden_dir, case + '.npz'))
    keys = list(gold['log_keys'])
    for step in (1, 2):
        t.update_learning_rate(); m.update_learning_rate()
        t.feed_data(batch); m.feed_data(batch, True)
        t.optimize_parameters(step); m.optimize_parameters(step)
        log = m.get_current_log()
        for k in keys:
            ref_v, gold_v = t.log[k], float(gold['logs'][step - 1][keys.index(k)])
            tol = 2e-3 if step == 1 else 2e-2  # step 2 sees weights moved by a sign-normalised Adam update
            # disc_Score = mean of logits of magnitude ~0.1-1 that nearly cancels: absolute tolerance on that scale
            # (the 'wgan-gp' losses ARE such means: -mean(real) + mean(fake))
            scorelike = k.startswith('disc_Score') or (c.get('gan_type') == 'wgan-gp' and ('l_d_' in k or '_gan_' in k))
            atol = (2e-4 if step == 1 else 2e-3) if scorelike else 1e-5
            assert abs(log[k] - ref_v) <= tol * max(1e-3, abs(ref_v)) + atol, (step, k, log[k], ref_v)
            assert abs(log[k] - gold_v) <= tol * max(1e-3, abs(gold_v)) + atol, (step, k, log[k], gold_v)
        if step == 1:
            assert rel(m.fake_H.cpu(), t.fake_H.detach()) < ACT_TOL
            gd = m.netG.params.grad_dict()
            worst = 0.0
            for (k, gv), pr in zip(gd.items(), netG.parameters()):
                r = rel(gv, pr.grad)
                worst = max(worst, r)
                assert r < G_TOL, ('G', k, r)
            dd = m.netD_target.params.grad_dict()
            for (k, gv), pr in zip(dd.items(), netD.parameters()):
                if zero_last_bias and k.endswith('model.8.bias'):
                    assert float(gv.abs().max()) < 1e-6 and float(pr.grad.abs().max()) < 1e-6
                    continue
                r = rel(gv, pr.grad)
                assert r < GRAD_TOL, ('D', k, r)
            if netD2 is not None:
                d2 = m.netD_source.params.grad_dict()
                for (k, gv), pr in zip(d2.items(), netD2.parameters()):
                    if zero_last_bias and k.endswith('model.8.bias'):
                        continue
                    assert rel(gv.reshape(pr.grad.shape), pr.grad) < GRAD_TOL, ('D_source', k, rel(gv.reshape(pr.grad.shape), pr.grad))
                np.testing.assert_allclose(np.array([float(v.double().norm()) for v in d2.values()]), gold['gradD2_norm'], rtol=GRAD_TOL, atol=1e-6)
            np.testing.assert_allclose(np.array([float(v.double().norm()) for v in gd.values()]), gold['gradG_norm'], rtol=G_TOL)
            np.testing.assert_allclose(np.array([float(v.double().norm()) for v in dd.values()]), gold['gradD_norm'], rtol=GRAD_TOL, atol=1e-6)
            w2 = max([rel(gv.reshape(pr.grad.shape), pr.grad) for (k, gv), pr in zip(m.netD_source.params.grad_dict().items(), netD2.parameters())
                      if not (zero_last_bias and k.endswith('model.8.bias'))]) if netD2 is not None else 0.0
            margins('%s: worst gradient rel err G %.2e (bound %.0e), D_source %.2e (bound %.0e)' % (case_id, worst, G_TOL, w2, GRAD_TOL))
            if t64 is not None:
                per_g = sorted(((rel(gv, pr.grad), k) for (k, gv), pr in zip(gd.items(), t64.netG.parameters())), reverse=True)
                e_g = per_g[0][0]
                e_d = max(rel(gv, pr.grad) for (k, gv), pr in zip(dd.items(), t64.netD.parameters()))
                e_s = max(rel(gv.reshape(pr.grad.shape), pr.grad) for (k, gv), pr in zip(m.netD_source.params.grad_dict().items(), t64.netD_src.parameters()))
                o_g = max(rel(p32.grad, p64.grad) for p32, p64 in zip(netG.parameters(), t64.netG.parameters()))
                o_s = max(rel(p32.grad, p64.grad) for p32, p64 in zip(netD2.parameters(), t64.netD_src.parameters()))
                margins('%s vs the fp64 oracle: HIP worst gradient rel err G %.2e, D_target %.2e, D_source %.2e; the fp32 oracle itself: G %.2e, '
                        'D_source %.2e; worst G tensors: %s' % (case_id, e_g, e_d, e_s, o_g, o_s, ' '.join('%s %.1e' % (k, e) for e, k in per_g[:4])))
                assert e_g < (VGG128_STEP_G_TOL if case_id.endswith('+bf16') else 1.1e-2) and e_d < 1e-2, (e_g, e_d)
                assert e_s < 1.1 * o_s, (e_s, o_s)   # no further from exact arithmetic than the fp32 PyTorch-CPU reference is (fp32 accumulation behind nine BatchNorms)


@pytest.mark.parametrize('form', [2, 3], ids=['lsgan', 'wgan'])
def test_ragan_lsgan_wgan_forms(form):
    """dasr_ragan forms 2 / 3: GANLoss('lsgan') / ('wgan-gp') terms on the relativistic logits (DASR_model.py:273-275 with loss.py:17-23)"""
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor
    g = torch.Generator().manual_seed(13)
    n, h, w = 3, 9, 12
    A = torch.randn(n, 1, h, w, generator=g).requires_grad_(True)
    B = torch.randn(n, 1, h, w, generator=g).requires_grad_(True)
    if form == 2:
        term = lambda x, t: F.mse_loss(x, torch.full_like(x, t))
    else:
        term = lambda x, t: -x.mean() if t > 0.5 else x.mean()
    L = (term(A - B.mean(0, keepdim=True), 1.0) + term(B - A.mean(0, keepdim=True), 0.0)) / 2
    L.backward()
    a, b = to_blocked(A.detach(), dev), to_blocked(B.detach(), dev)
    ga, gb = BTensor(n, 16, h, w, True, dev), BTensor(n, 16, h, w, True, dev)
    hw, cnt = h * w, float(n * h * w)
    sums, part = torch.zeros(2 * hw, device=dev), torch.zeros(2 * hw, device=dev)
    acc = torch.zeros(4, device=dev)
    for k in range(3):
        _lib.check(_lib.lib().dasr_ragan(a.view(), b.view(), n, h, w, k, n, form, 1.0, 0.0, 0.5 / cnt, 0.5 / cnt, 0.0, sums.data_ptr(), part.data_ptr(),
                                         acc.data_ptr(), acc.data_ptr() + 4, acc.data_ptr() + 8, 1.0 / cnt, ga.view(), gb.view(), None))
    assert abs(float(acc[0]) - float(L)) < 1e-5 * max(1.0, abs(float(L)))
    assert abs(float(acc[1]) - float(A.detach().mean())) < 1e-5 and abs(float(acc[2]) - float(B.detach().mean())) < 1e-5
    assert rel(ga.nchw(1).cpu(), A.grad) < 1e-5 and rel(gb.nchw(1).cpu(), B.grad) < 1e-5


@pytest.mark.parametrize('n,n_glob,h,w', [(3, 3, 14, 14), (2, 6, 9, 11)])
def test_ragan_three_stage_loss_and_gradients(n, n_glob, h, w):
    """dasr_ragan against torch autograd of the reference formula (DASR_model.py:273-275); n_glob > n: this rank's samples next to the
    other ranks' (their per-pixel sums are added by hand where the all-reduce would add them)"""
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor
    g = torch.Generator().manual_seed(11)
    A = torch.randn(n_glob, 1, h, w, generator=g).requires_grad_(True)
    B = torch.randn(n_glob, 1, h, w, generator=g).requires_grad_(True)
    bce = lambda x, t: F.binary_cross_entropy_with_logits(x, torch.full_like(x, t))
    L = (bce(A - B.mean(0, keepdim=True), 1.0) + bce(B - A.mean(0, keepdim=True), 0.0)) / 2
    L.backward()
    a, b = to_blocked(A.detach()[:n], dev), to_blocked(B.detach()[:n], dev)
    ga, gb = BTensor(n, 16, h, w, True, dev), BTensor(n, 16, h, w, True, dev)
    hw = h * w
    sums, part = torch.zeros(2 * hw, device=dev), torch.zeros(2 * hw, device=dev)
    acc = torch.zeros(4, device=dev)
    cnt = float(n_glob * hw)     # the reference's mean runs over the global batch
    Lib = _lib.lib()

    def stage(k):
        _lib.check(Lib.dasr_ragan(a.view(), b.view(), n, h, w, k, n_glob, 0, 1.0, 0.0, 0.5 / cnt, 0.5 / cnt, 0.0, sums.data_ptr(), part.data_ptr(),
                                  acc.data_ptr(), acc.data_ptr() + 4, acc.data_ptr() + 8, 1.0 / float(n * hw), ga.view(), gb.view(), None))
        torch.cuda.synchronize()

    def others(fn):   # what the other ranks' stage would contribute to the all-reduced buffer
        if n_glob == n:
            return 0.0
        return torch.cat([fn(A.detach()[n:]).sum(0).reshape(-1), fn(B.detach()[n:]).sum(0).reshape(-1)]).to(dev)

    stage(0)
    sums += others(lambda t: t) if n_glob > n else 0.0
    mA, mB = A.detach().mean(0, keepdim=True), B.detach().mean(0, keepdim=True)
    assert rel(sums.cpu(), torch.cat([A.detach().sum(0).reshape(-1), B.detach().sum(0).reshape(-1)])) < 1e-6
    stage(1)
    if n_glob > n:
        part += torch.cat([(torch.sigmoid(A.detach()[n:] - mB) - 1.0).sum(0).reshape(-1), torch.sigmoid(B.detach()[n:] - mA).sum(0).reshape(-1)]).to(dev)
    stage(2)
    la = (F.binary_cross_entropy_with_logits(A.detach()[:n] - mB, torch.ones(n, 1, h, w), reduction='sum') +
          F.binary_cross_entropy_with_logits(B.detach()[:n] - mA, torch.zeros(n, 1, h, w), reduction='sum')) * 0.5 / cnt
    assert abs(float(acc[0]) - float(la)) < 1e-5 * abs(float(la))
    assert abs(float(acc[1]) - float(A.detach()[:n].mean())) < 1e-5 and abs(float(acc[2]) - float(B.detach()[:n].mean())) < 1e-5
    assert rel(ga.nchw(1).cpu(), A.grad[:n]) < 1e-5 and rel(gb.nchw(1).cpu(), B.grad[:n]) < 1e-5


def test_ragan_dsn_form_loss_and_gradients():
    """dasr_ragan form 1 (the DSN's --ragan: -log(sigmoid(x - mean_n(y)) + eps) terms, codes/DSN/train.py:221-223, loss.py:11-41) against torch
    autograd: the discriminator loss (both terms, both gradients) and the generator's texture loss (fake term only, target 'real')"""
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, NULL_T
    g = torch.Generator().manual_seed(12)
    n, h, w, eps = 3, 10, 13, 1e-8
    R = (torch.randn(n, 1, h, w, generator=g) * 2).requires_grad_(True)   # real logits
    Fk = (torch.randn(n, 1, h, w, generator=g) * 2).requires_grad_(True)  # fake logits
    rt, ft = torch.sigmoid(R - Fk.mean(0, keepdim=True)), torch.sigmoid(Fk - R.mean(0, keepdim=True))
    d_loss = -torch.log(rt + eps).mean() - torch.log(1 - ft + eps).mean()
    gR, gF = torch.autograd.grad(d_loss, [R, Fk], retain_graph=True)
    tex = torch.mean(-torch.log(ft + eps))
    gF_tex, = torch.autograd.grad(0.005 * tex, [Fk])
    a, b = to_blocked(R.detach(), dev), to_blocked(Fk.detach(), dev)
    ga, gb = BTensor(n, 16, h, w, True, dev), BTensor(n, 16, h, w, True, dev)
    hw, cnt = h * w, float(n * h * w)
    sums, part = torch.zeros(2 * hw, device=dev), torch.zeros(2 * hw, device=dev)
    acc = torch.zeros(4, device=dev)
    Lib = _lib.lib()
    for k in range(3):
        _lib.check(Lib.dasr_ragan(a.view(), b.view(), n, h, w, k, n, 1, 1.0, 0.0, 1.0 / cnt, 1.0 / cnt, eps, sums.data_ptr(), part.data_ptr(),
                                  acc.data_ptr(), acc.data_ptr() + 4, acc.data_ptr() + 8, 1.0 / cnt, ga.view(), gb.view(), None))
    assert abs(float(acc[0]) - float(d_loss)) < 1e-5 * abs(float(d_loss))
    assert abs(float(acc[1]) - float(rt.mean())) < 1e-5 and abs(float(acc[2]) - float(ft.mean())) < 1e-5
    assert rel(ga.nchw(1).cpu(), gR) < 1e-5 and rel(gb.nchw(1).cpu(), gF) < 1e-5
    acc.zero_()
    for k in (1, 2):   # generator: the sums of stage 0 are still valid; real term absent (t < 0), fake term against the 'real' label
        _lib.check(Lib.dasr_ragan(a.view(), b.view(), n, h, w, k, n, 1, -1.0, 1.0, 1.0 / cnt, 0.005 / cnt, eps, sums.data_ptr(), part.data_ptr(),
                                  acc.data_ptr(), None, None, 0.0, NULL_T, gb.view(), None))
    assert abs(float(acc[0]) - float(tex)) < 1e-5 * abs(float(tex))
    assert rel(gb.nchw(1).cpu(), gF_tex) < 1e-5


def test_batchnorm_train_lrelu_forward_backward_two_groups():
    """dasr_bnorm_lrelu_fwd / _bwd / _running against nn.BatchNorm2d(train) + LeakyReLU applied to the two halves of the batch separately"""
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor
    g = torch.Generator().manual_seed(5)
    n, C_, H, W = 3, 40, 9, 7
    x = torch.randn(2 * n, C_, H, W, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(C_)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(C_, generator=g))
        bn.bias.copy_(0.2 * torch.randn(C_, generator=g))
    bn.train()
    xr = x.clone().requires_grad_(True)
    ga = torch.randn(2 * n, C_, H, W, generator=g)
    y = torch.cat([F.leaky_relu(bn(xr[n:]), 0.2), F.leaky_relu(bn(xr[:n]), 0.2)])      # the reference's order: real half first, then fake
    y = torch.cat([y[n:], y[:n]])
    (y * ga).sum().backward()
    Lib = _lib.lib()
    xb, gab = to_blocked(x, dev), to_blocked(ga, dev)
    yb, gxb = BTensor(2 * n, C_, H, W, True, dev), BTensor(2 * n, C_, H, W, True, dev)
    gam, bet = bn.weight.detach().to(dev), bn.bias.detach().to(dev)
    stats = torch.zeros(2 * 48 * 3, device=dev)
    dg, db = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    _lib.check(Lib.dasr_bnorm_lrelu_fwd(xb.view(), 2 * n, C_, H, W, n, 1e-5, 0.2, gam.data_ptr(), bet.data_ptr(), yb.view(), stats.data_ptr(), None))
    _lib.check(Lib.dasr_bnorm_lrelu_bwd(xb.view(), gab.view(), 2 * n, C_, H, W, n, 0.2, gam.data_ptr(), bet.data_ptr(), stats.data_ptr(), gxb.view(),
                                        dg.data_ptr(), db.data_ptr(), 1.0, None))
    rm, rv, nbt = torch.zeros(C_, device=dev), torch.ones(C_, device=dev), torch.zeros(1, device=dev)
    for grp in (1, 0):   # real, then fake: the order of the two reference forwards above
        _lib.check(Lib.dasr_bnorm_running(stats.data_ptr(), grp, C_, n * H * W, 0.1, rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), None))
    torch.cuda.synchronize()
    assert rel(yb.nchw().cpu(), y.detach()) < 1e-5
    assert rel(gxb.nchw().cpu(), xr.grad) < 1e-4
    assert rel(dg.cpu(), bn.weight.grad) < 1e-4 and rel(db.cpu(), bn.bias.grad) < 1e-4
    assert rel(rm.cpu(), bn.running_mean) < 1e-5 and rel(rv.cpu(), bn.running_var) < 1e-5 and int(nbt.item()) == int(bn.num_batches_tracked)


# Discriminator_VGG_128: nine training-mode BatchNorm layers in a row with tiny statistics groups (n x 4 x 4 values per channel at the top).
# Forward is accurate (logits 5e-6), the data gradient too (dL/dinput 4e-6: split-f16 operands).  The weight / BatchNorm-parameter gradients sit
# 3.5 - 4.6e-3 from the fp32 torch reference: that is the noise floor of fp32 ACCUMULATION on this network (each BatchNorm backward subtracts
# the group means of gz and gz * xhat from gz), not operand rounding -- 22-bit weight-gradient operands (round 4, dW = g.x + g.x_lo + g_lo.x)
# leave it unchanged.  Bound: the north_star's 1e-2 (round 3: 2e-2).
VGG128_GRAD_TOL = 1e-2
# The DASR step with this network as D_source (dasr_srcVGG128_gau5_nf32_nb1_n3_32) is ill-conditioned: the discriminator sees high-frequency maps
# (0.75 + small detail), conv outputs with |mean| >> std in front of every BatchNorm.  Measured (oracle/bn_probe.py, profiles/r04_bn_probe.txt:
# the reference step in fp64 = truth, ONE component at a time degraded to the arithmetic of the HIP path):
#   * fp32 PyTorch-CPU (the reference itself): G 3.2e-3, D_source 1.2e-2 from the fp64 result;
#   * D_source in the HIP path's arithmetic (22-bit conv operands, fp32 BatchNorm): 3.5e-4 on the G gradients, 1.9 - 4.6e-3 on D_source's;
#   * the generator's dense blocks with bf16 operands -- the north_star's dtype -- and EVERYTHING ELSE EXACT: G 1.24e-2 (9.8e-3 from the bf16
#     rounding of the weights alone: a coherent 2^-9 perturbation of the network that the BatchNorm discriminator's gradient amplifies).
# So: D_target and D_source must meet the north_star's 1e-2 against the fp32 reference, and D_source must be no further from the fp64 result than
# the fp32 reference is (+10 %).  For the G gradients the product path answers the probe: for `which_model_pairD: discriminator_vgg_128` DASR_Model
# runs the generator's dense blocks in f16 STORAGE (rrdbnet.py rdb_prec 2: 11-bit operands, gradient slabs scaled by a calibrated power of two) --
# the case then meets 1e-2 like every other (observed 8.5e-3 against fp32, 8.4e-3 against fp64; other cases 3e-4 .. 5e-3).  The `+bf16` variant of
# the case forces bf16 dense blocks back (DASR_RDB_PREC=1) and keeps the 2e-2 bound that bf16 operands give on it (observed 1.46e-2);
# tests/test_bn_conditioning.py pins the 1.24e-2 of bf16 dense blocks alone on the CPU.
VGG128_STEP_G_TOL = 2e-2


def test_discriminator_vgg128_forward_backward_and_state_dict(margins):
    dev = _gpu()
    from dasr_amd.gan_nets import DiscriminatorVGG128HIP
    from oracle import nets, fixtures
    ref = nets.Discriminator_VGG_128(3, 64)
    sd = fixtures.seeded_state_dict(ref.state_dict(), 3, 1.0)
    ref.load_state_dict(sd)
    ref.train()
    D = DiscriminatorVGG128HIP(3, 64, device=dev)
    D.load_state_dict(sd)
    n = 2
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2 * n, 3, 128, 128, generator=g)
    p = D.plan(2 * n, 128, 128)
    p.x.t.copy_(to_blocked(x, dev).t)
    p.fwd.run()
    out = torch.cat([ref(x[:n]), ref(x[n:])])              # separate calls per half: separate batch statistics
    got = p.logits.nchw(1).cpu().reshape(2 * n, 1)
    assert rel(got, out.detach()) < ACT_TOL, rel(got, out.detach())
    gl = torch.randn(2 * n, 1, generator=g)
    p.g_logits.t.copy_(to_blocked(gl.reshape(2 * n, 1, 1, 1), dev).t)
    (out * gl).sum().backward()
    p.bwd_full.run()
    p.running_ops(0).run()
    p.running_ops(1).run()
    torch.cuda.synchronize()
    gd = D.params.grad_dict()
    worst, errs = 0.0, []
    for (k, gv), pr in zip(gd.items(), ref.parameters()):
        r = rel(gv.reshape(pr.grad.shape), pr.grad)
        worst = max(worst, r)
        errs.append('%s %.1e' % (k, r))
    # generator-step path: data gradient of the fake half alone (its own batch statistics), down to the input image
    xf = x[:n].clone().requires_grad_(True)
    (ref(xf) * gl[:n]).sum().backward()
    p.g_logits.t.copy_(to_blocked(gl.reshape(2 * n, 1, 1, 1), dev).t)
    p.bwd_data_ops(n).run()
    p.running_ops(0).run()        # the reference side just ran a third forward (fake half again), as the generator step does
    torch.cuda.synchronize()
    e_in = rel(p.gx.nchw(3).cpu()[:n], xf.grad)
    margins('Discriminator_VGG_128 (BatchNorm, train mode): logits rel err %.2e (tol 1e-3); worst weight-gradient rel err %.2e, dL/dinput %.2e (bound %.0e): %s'
            % (rel(got, out.detach()), worst, e_in, VGG128_GRAD_TOL, ' | '.join(errs[:6])))
    assert worst < VGG128_GRAD_TOL and e_in < VGG128_GRAD_TOL, (worst, e_in)
    sd2 = D.state_dict()
    assert list(sd2.keys()) == list(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert tuple(sd2[k].shape) == tuple(v.shape) and sd2[k].dtype == v.dtype, k
        if 'running' in k or 'num_batches' in k:
            assert rel(sd2[k].float(), v.float()) < 1e-4, k
    print('Discriminator_VGG_128: worst grad rel err %.2e' % worst)


def test_gan_step_with_chained_trunk_launches_is_bit_identical(monkeypatch):
    """the GAN trainer on a shape whose generator batch fills the chip (8 source + 8 target crops of 128 x 128: 16 x 32 tiles): the trunk of its generator plan
    runs as persistent chained launches (rrdbnet._Plan, dasr_conv_chain) -- generator / discriminator weights after two steps and every logged term must be
    bit-identical to the per-layer launches"""
    _gpu()
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip('the chained launches need a whole 256-CU MI355X (RRDBNetHIP.chain_ok)')
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    case = dict(kind='dasr', nf=64, nb=1, n=8, lr=128, fs='wavelet', d_in_nc=9)
    batch = fixtures.make_batch(case, seed=21)
    outs = []
    for chain in ('0', '1'):
        monkeypatch.setenv('DASR_CHAIN', chain)
        torch.manual_seed(0)
        o = fixtures.make_opt(case)
        o['gpu_ids'] = [0]
        o['train']['vgg_seed'] = 77
        m = create_model(options.dict_to_nonedict(o))
        for step in (1, 2):
            m.update_learning_rate()
            m.feed_data(batch, True)
            m.optimize_parameters(step)
        log = dict(m.get_current_log())   # (host sync; includes the chains' error word)
        plans = list(m.netG.plans.values())
        assert any(getattr(p, 'chain', None) is not None for p in plans) == (chain == '1')
        outs.append((m.netG.params.flat.clone(), m.netD_target.params.flat.clone(), m.fake_H.clone(), log))
    (g0, d0, s0, l0), (g1, d1, s1, l1) = outs
    assert torch.equal(s0, s1) and torch.equal(g0, g1) and torch.equal(d0, d1)
    for k in l0:   # (the logged terms are fixed-order grid sums since round 5 -- grid_sum_commit, csrc/common.h -- so they take part in the bit comparison)
        assert l0[k] == l1[k], (k, l0[k], l1[k])
