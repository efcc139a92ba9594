"""GPU parity of the DSN path: the extra kernels (5x5 / 1x1 / 3x3-stride-2 convolutions and
weight gradients, PReLU, sigmoid, -log losses, un-padded low-pass), De_resnet forward/backward and the full DSN iteration
against oracle/dsn.py (fp32 CPU) and the fixtures made from the reference's codes/DSN/model.py + loss.py.
Tolerances: activations 1e-3, gradients 1e-2 (north_star)."""
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


def _check_grads(got, want, tag):
    """per-tensor relative error < GRAD_TOL.  Two kinds of entries are compared on an absolute scale instead:
    - the 11 scalar nn.PReLU slopes: each is a cancelling sum over ~4e5 terms, and a (P/Leaky)ReLU kink that falls on the
      other side of zero in a handful of pixels (|pre-activation| < fp32 noise) moves it by ~1e-3 of the typical slope
      gradient; they are compared jointly as one vector;
    - biases in front of an InstanceNorm: their true gradient is exactly zero, both sides hold rounding noise."""
    worst, sg, sw = 0.0, [], []
    for (k, gv), wv in zip(got.items(), want):
        if gv.numel() == 1:
            sg.append(gv.flatten())
            sw.append(wv.flatten())
            continue
        if float(wv.double().norm()) < 1e-6:
            assert float(gv.double().norm()) < 1e-6, (tag, k)
            continue
        r = rel(gv, wv)
        worst = max(worst, r)
        assert r < GRAD_TOL, (tag, k, r)
    rs = 0.0
    if sg:
        rs = rel(torch.cat(sg), torch.cat(sw))
        assert rs < GRAD_TOL, (tag, 'prelu slopes', rs)
    return worst, rs


@pytest.mark.parametrize('kh,stride,cin,cout,hw', [(5, 1, 3, 64, (19, 33)), (5, 1, 64, 128, (20, 24)), (1, 1, 256, 1, (13, 17)),
                                                   (3, 2, 64, 64, (24, 40))])
def test_conv_variants_fwd_wgrad(kh, stride, cin, cout, hw):
    """split-bf16 convolution + weight gradient for the DSN kernel sizes, PReLU slope read from device memory"""
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, ParamStore, PackRegistry, OpList, WgradGroup, Workspace, conv_op, ceil_div
    N, (H, W) = 2, hw
    pad = (kh - 1) // 2
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kh) // stride + 1
    g = torch.Generator().manual_seed(kh * 10 + stride)
    w = torch.randn(cout, cin, kh, kh, generator=g) / (cin * kh * kh) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    a = torch.tensor([0.3])
    x = torch.randn(N, cin, H, W, generator=g)
    P = ParamStore([('w', tuple(w.shape)), ('b', (cout,)), ('a', (1,))], dev)
    P.load_state_dict({'w': w, 'b': b, 'a': a})
    pack = PackRegistry(P)
    cin_pad = ceil_div(cin, 16) * 16
    ref = pack.add(cout, cin_pad, kh * kh, 1, 3, [(P.off('w'), cout, cin, 0, cin, 0, 0)])
    pack.finalize()
    pack.run()
    xb = to_blocked(x, dev)
    y = BTensor(N, max(cout, 16), Ho, Wo, True, dev)
    ops = OpList()
    ops.add(conv_op(pack, ref, xb.view(), True, cin_pad, H, W, Ho, Wo, N, bias=P.ptr('b'), kh=kh, stride=stride, pad=pad, act=1,
                    slope_ptr=P.ptr('a'), out_f32=y.view()))
    ops.run()
    yr = F.prelu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad), a.double())
    assert rel(y.nchw(cout).cpu(), yr.float()) < 2e-5
    # weight / bias gradient
    go = torch.randn(N, cout, Ho, Wo, generator=g)
    gb = to_blocked(go, dev)
    ws = Workspace(dev)
    grp = WgradGroup(kh, stride)
    grp.add_conv(gb.view, True, gb.planes, xb.view, True, xb.planes, cout, cin, H, W, Ho, Wo, N, P.off('w'), P.off('b'), pad=pad)
    grp.finalize(ws, dev)
    wl = OpList()
    for o in grp.ops(P.grad.data_ptr()):
        wl.add(o)
    ws.finalize()
    wl.run()
    xr, wr = x.double(), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    (F.conv2d(xr, wr, br, stride=stride, padding=pad) * go.double()).sum().backward()
    gd = P.grad_dict()
    # weight gradients use single-bf16 operands with fp32 accumulation (~2e-3 of the gradient norm; gradient tolerance 1e-2)
    assert rel(gd['w'], wr.grad.float()) < 5e-3, rel(gd['w'], wr.grad.float())
    assert rel(gd['b'], br.grad.float()) < 1e-4


def test_logloss_sigmoid_prelu_lowpass_valid():
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, _stream, NULL_T
    from oracle import dsn
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    # -log(sigmoid) losses
    x = torch.randn(3, 1, 9, 11, generator=g) * 2
    xb = to_blocked(x, dev)
    cnt = float(x.numel())
    for mode in (0, 1):
        gr = BTensor(3, 16, 9, 11, True, dev)
        acc = torch.zeros(4, device=dev)
        _lib.check(L.dasr_logloss(xb.view(), 3, 9, 11, mode, 1e-8, 1.0 / cnt, 0.5 / cnt, acc.data_ptr(), acc.data_ptr() + 4, 1.0 / cnt, gr.view(), 0,
                                  _stream()))
        xr = x.clone().requires_grad_(True)
        p = torch.sigmoid(xr)
        l = (-torch.log(p + 1e-8)).mean() if mode == 0 else (-torch.log(1 - p + 1e-8)).mean()
        (0.5 * l).backward()
        assert abs(float(acc[0]) - float(l)) < 1e-5 and abs(float(acc[1]) - float(p.mean())) < 1e-5
        assert rel(gr.nchw(1).cpu(), xr.grad) < 1e-5
    # sigmoid backward on 3 channels
    y = torch.rand(2, 3, 7, 9, generator=g)
    go = torch.randn(2, 3, 7, 9, generator=g)
    yb, gb, gz = to_blocked(y, dev), to_blocked(go, dev), BTensor(2, 16, 7, 9, True, dev)
    _lib.check(L.dasr_sigmoid_bwd(yb.view(), gb.view(), 2, 3, 7, 9, gz.view(), _stream()))
    assert rel(gz.nchw(3).cpu(), go * y * (1 - y)) < 1e-6
    # PReLU slope gradient from (y, masked gradient)
    a = torch.tensor([0.25], requires_grad=True)
    pre = torch.randn(2, 64, 10, 12, generator=g)
    gpost = torch.randn(2, 64, 10, 12, generator=g)
    yv = F.prelu(pre, a)
    yv.backward(gpost)
    gpre = gpost * torch.where(pre > 0, torch.ones(()), a.detach())
    yb, gb = to_blocked(yv.detach(), dev), to_blocked(gpre, dev)
    scratch = torch.zeros(1024, device=dev)
    out = torch.zeros(2, device=dev)
    sl = a.detach().to(dev)
    _lib.check(L.dasr_prelu_grad(yb.view(), gb.view(), 2, 64, 10, 12, sl.data_ptr(), scratch.data_ptr(), out.data_ptr(), 0.5, _stream()))
    assert abs(float(out[0]) - 0.5 * float(a.grad)) < 1e-4 * max(1.0, abs(float(a.grad)))
    # un-padded low-pass (colour loss filter) and its adjoint
    for gaussian in (True, False):
        f = dsn.FilterLow(5, padding=False, gaussian=gaussian)
        w = (dsn.nets.gaussian_kernel2d(5) if gaussian else torch.full((5, 5), 1 / 25.0)).contiguous().to(dev)
        img = torch.rand(2, 3, 14, 18, generator=g)
        ir = img.clone().requires_grad_(True)
        lo = f(ir)
        gl = torch.randn(lo.shape, generator=g)
        (lo * gl).sum().backward()
        ib, ob = to_blocked(img, dev), BTensor(2, 16, 10, 14, True, dev)
        _lib.check(L.dasr_lowpass_valid(ib.view(), w.data_ptr(), 5, 2, 3, 14, 18, 0, ob.view(), 0, _stream()))
        assert rel(ob.nchw(3).cpu(), lo.detach()) < 1e-5
        glb, gx = to_blocked(gl, dev), BTensor(2, 16, 14, 18, True, dev)
        _lib.check(L.dasr_lowpass_valid(glb.view(), w.data_ptr(), 5, 2, 3, 14, 18, 1, gx.view(), 0, _stream()))
        assert rel(gx.nchw(3).cpu(), ir.grad) < 1e-5
    # high-pass front end with AvgPool2d(count_include_pad=False) and its adjoint
    fh = dsn.FilterHigh(5, include_pad=False, gaussian=False)
    w = torch.full((5, 5), 1 / 25.0).contiguous().to(dev)
    img = torch.rand(2, 3, 12, 15, generator=g)
    ir = img.clone().requires_grad_(True)
    hi = fh(ir)
    gh = torch.randn(hi.shape, generator=g)
    (hi * gh).sum().backward()
    ib, ob = to_blocked(img, dev), BTensor(2, 16, 12, 15, True, dev)
    _lib.check(L.dasr_lowpass(ib.view(), NULL_T, w.data_ptr(), 5, 2, 3, 12, 15, 0 | 2, 0.5, 0.5, NULL_T, ob.view(), 0, _stream()))
    assert rel(ob.nchw(3).cpu(), hi.detach()) < 1e-5
    ghb, gx = to_blocked(gh, dev), BTensor(2, 16, 12, 15, True, dev)
    _lib.check(L.dasr_lowpass(NULL_T, ghb.view(), w.data_ptr(), 5, 2, 3, 12, 15, 1 | 2, 0.5, 0.0, gx.view(), NULL_T, 0, _stream()))
    assert rel(gx.nchw(3).cpu(), ir.grad) < 1e-5


def test_deresnet_forward_backward():
    dev = _gpu()
    from dasr_amd.dsn_model import DeResnetHIP
    from oracle import dsn
    from oracle.gen_golden_dsn import dsn_state
    ref = dsn.DeResnet()
    sd = dsn_state(ref.state_dict(), 21, 0.5)
    ref.load_state_dict(sd)
    G = DeResnetHIP(8, device=dev)
    assert list(G.params.spec) == list(ref.state_dict().keys())
    G.load_state_dict(sd)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 48, 64, generator=g)
    p = G.plan(2, 48, 64)
    p.x_nchw.copy_(x)
    p.fwd.run()
    y = ref(x)
    assert rel(p.fake_nchw.cpu(), y.detach()) < ACT_TOL
    gy = torch.randn(y.shape, generator=g)
    p.g_fake.t.copy_(to_blocked(gy, dev).t)
    (y * gy).sum().backward()
    p.bwd.run()
    worst, slopes = _check_grads(G.params.grad_dict(), [p.grad for p in ref.parameters()], 'G')
    print('De_resnet worst grad rel err %.2e, PReLU slopes %.2e' % (worst, slopes))


@pytest.mark.parametrize('shape', [(2, 48, 64), (1, 40, 44), (3, 64, 32)])
def test_prelu_slope_gradient_from_the_conv_epilogue(shape, monkeypatch, margins):
    """round 6 (dasr_conv_params::prelu_part + dasr_prelu_final): the PReLU-slope gradients of the residual blocks taken inside the data-gradient conv's epilogue against
    the two-pass form (dasr_prelu_grad_f16 on the stored h and dL/dz) on the same plan inputs -- every other gradient must be bit-identical (nothing else changes), the
    slopes agree to the rounding of dL/dz to f16 that only the two-pass form sees (a slope gradient is a small difference of large sums: 5e-4 per term shows as up to
    3e-3 of the sum; tolerance = north_star's 1e-2 for gradients); odd sizes: partial tiles contribute nothing"""
    dev = _gpu()
    from dasr_amd.dsn_model import DeResnetHIP
    from oracle import dsn
    from oracle.gen_golden_dsn import dsn_state
    sd = dsn_state(dsn.DeResnet().state_dict(), 21, 0.5)
    n, h, w = shape
    g = torch.Generator().manual_seed(13)
    x, gy = torch.rand(n, 3, h, w, generator=g), torch.randn(n, 3, h // 4, w // 4, generator=g)
    grads = {}
    for fused in ('1', '0'):
        monkeypatch.setenv('DASR_DSN_PRELU_FUSED', fused)
        G = DeResnetHIP(8, device=dev)
        G.load_state_dict(sd)
        p = G.plan(n, h, w)
        assert (p._prelu_final is not None) == (fused == '1')
        p.x_nchw.copy_(x)
        p.fwd.run()
        p.g_fake.t.copy_(to_blocked(gy, dev).t)
        p.bwd.run()
        torch.cuda.synchronize()
        grads[fused] = {k: v.clone() for k, v in G.params.grad_dict().items()}
    worst = 0.0
    for k, v in grads['1'].items():
        if k.startswith('res_blocks.') and k.endswith('prelu.weight'):
            e = float((v - grads['0'][k]).abs() / grads['0'][k].abs().clamp_min(1e-12))
            worst = max(worst, e)
            assert e < 1e-2, (k, float(v), float(grads['0'][k]))
        else:
            assert torch.equal(v, grads['0'][k]), k
    margins('DSN generator %d x %d x %d: PReLU-slope gradients from the conv epilogue vs the two-pass form: worst rel diff %.2e (tol 1e-2), everything else bit-identical' % (n, h, w, worst))


@pytest.mark.parametrize('case', ['dsn_gau5_inst_b2_128', 'dsn_wavelet_inst_b2_128', 'dsn_avg5_inst_b1_160', 'dsn_gau5_inst_b1_256_lpips',
                                  'dsn_wavelet_nld_s2_b2_128', 'dsn_gau5_nld_s1_b1_128', 'dsn_dsgan_gau5_inst_b2_128', 'dsn_gau5_inst_b3_128_ragan',
                                  'dsn_gau5_batch_b2_128', 'dsn_avg5_batch_b3_128_ragan', 'dsn_wavelet_sum_inst_b2_128',
                                  'dsn_gau5_inst_b2_128+fwd32', 'dsn_dsgan_gau5_inst_b2_128+fwd32', 'dsn_gau5_inst_b2_256_lpips_rotflip',
                                  'dsn_gau5_inst_b2_128_wgan', 'dsn_wavelet_inst_b2_128_wgan',
                                  'dsn_gau5_nld_s1_batch_b2_128', 'dsn_wavelet_nld_s2_batch_b3_128', 'dsn_gau5_batch_b2_128_wgan',
                                  'dsn_wavelet_nld_s2_batch_b2_128_wgan'])
def test_dsn_iteration_matches_oracle_and_reference_fixture(case, golden_dir, monkeypatch, margins):
    dev = _gpu()
    case_id = case
    if case.endswith('+fwd32'):   # the forward on fp32 tensors with separate f16 shadows (default: split f16 tensors, dsn_model.DeResnetHIP.fwd16)
        case = case[:-6]
        monkeypatch.setenv('DASR_DSN_FWD16', '0')
    torch.set_num_threads(8)
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn
    from oracle.gen_golden_dsn import DSN_CASES, dsn_state, dsn_batch
    c = DSN_CASES[case]
    gold = np.load(os.path.join(golden_dir, case + '.npz'))
    G = dsn.GeneratorDSGAN() if c.get('gen') == 'DSGAN' else dsn.DeResnet()
    D = dsn.Discriminator(c['k'], c['norm'], c['filter'], D_arch=c.get('arch', 'FSD'), cs=c.get('cs', 'cat'), wgan=bool(c.get('wgan')))
    sdG, sdD = dsn_state(G.state_dict(), 21, 0.5), dsn_state(D.state_dict(), 22, 1.0)
    G.load_state_dict(sdG)
    D.load_state_dict(sdD)
    crit, sdF = None, None
    if c.get('per') == 'LPIPS':     # --per_type LPIPS (reference default): real linear heads + the fixture's stand-in AlexNet
        from oracle import lpips
        crit, sdF = lpips.golden_criterion(78, golden_dir)
    t = dsn.DSNTrainer(G, D, kernel_size=c['k'], filter_type=c['filter'], norm_layer=c['norm'], vgg_seed=78, w_per=0.01, per_type=c.get('per', 'VGG'), netF=crit, ragan=bool(c.get('ragan')),
                       lpips_rot_flip=bool(c.get('rot_flip')), wgan=bool(c.get('wgan')))
    m = DSNModel(dict(ragan=bool(c.get('ragan')), filter=c['filter'], kernel_size=c['k'], norm_layer=c['norm'], w_per=0.01, vgg_seed=78, per_type=c.get('per', 'VGG'), discriminator=c.get('arch', 'FSD'), generator=c.get('gen', 'DeResnet'), allow_random_perceptual=True, cat_or_sum=c.get('cs', 'cat'),
                      lpips_rot_flip=bool(c.get('rot_flip')), wgan=bool(c.get('wgan'))), device=dev)
    d_keys = list(m.netD.state_dict()) if c['norm'] == 'Batch' else list(m.netD.params.spec)   # BatchNorm: buffers are part of the reference layout
    assert list(m.netG.params.spec) == list(gold['G_keys']) and d_keys == list(gold['D_keys'])
    m.netG.load_state_dict(sdG)
    m.load_discriminator_state(sdD)
    m.netF.load_state_dict(sdF if sdF is not None else {'features.' + k: v for k, v in t.per.state_dict().items()})
    hr, bic, real = dsn_batch(c)
    from oracle import fixtures
    import random
    for step in (1, 2):
        # --lpips_rot_flip: both sides draw the symmetry from python's `random` like the reference (loss.py:155-168); step 1 = the fixture's seed
        # (k_rot -1, rows flipped), step 2 = seed + 1 (k_rot -1, columns flipped)
        random.seed(c.get('rseed', 0) + step - 1)
        torch.manual_seed(c.get('tseed', 0) + step - 1)    # --wgan: the mixing weight of the gradient penalty comes from torch's global RNG (train.py:232)
        t.iteration(hr, bic, real)
        random.seed(c.get('rseed', 0) + step - 1)
        torch.manual_seed(c.get('tseed', 0) + step - 1)
        m.iteration(hr.to(dev), bic.to(dev), real.to(dev))
        log = m.get_current_log()
        tol = 2e-3 if step == 1 else 2e-2
        for k, ref_v in t.log.items():
            assert abs(log[k] - ref_v) <= tol * max(1e-3, abs(ref_v)) + 1e-5, (step, k, log[k], ref_v)
        if step == 1:
            gl = gold['losses'][:5]  # d_loss, tex, col, per, g_loss from the reference modules (--wgan fixtures: + the gradient penalty)
            if c.get('wgan'):
                assert abs(log['disc_score/gradient_penalty'] - float(gold['losses'][5])) < 2e-3 * float(gold['losses'][5])
            got = [log['loss/d_tex_loss'], log['loss/g_tex_loss'], log['loss/color_loss'], log['loss/perceptual_loss'], log['loss/g_overall_loss']]
            np.testing.assert_allclose(got, gl, rtol=2e-3, atol=1e-5)
            assert rel(m.fake.cpu(), t.fake) < ACT_TOL
            np.testing.assert_allclose(fixtures.subsample(m.fake.cpu()).numpy(), gold['fake_sub'], rtol=0, atol=2e-4)
            gd, dd = m.netG.params.grad_dict(), m.netD.params.grad_dict()
            wg_, ws_ = _check_grads(gd, [p.grad for p in G.parameters()], 'G')
            e_fake = rel(m.fake.cpu(), t.fake)
            dpar = dict((k, v) for k, v in dd.items() if 'gaussian_filter' not in k)
            dwant = [p.grad for p in D.parameters() if p.requires_grad]
            if c.get('ragan'):   # relativistic logits: a constant shift of every logit changes nothing -> the true gradient of the last bias is 0
                k_last = list(dpar)[-1]
                assert k_last.endswith('bias') and float(dpar[k_last].abs().max()) < 1e-5 and float(dwant[-1].abs().max()) < 1e-5   # rounding noise of sums of O(1) terms
                wd_, _ = _check_grads(dict(list(dpar.items())[:-1]), dwant[:-1], 'D')
            else:
                wd_, _ = _check_grads(dpar, dwant, 'D')
            margins('DSN iteration %s: fake rel err %.2e (tol %.0e); worst gradient rel err G %.2e, G PReLU slopes (jointly) %.2e, D %.2e (tol %.0e)'
                    % (case_id, e_fake, ACT_TOL, wg_, ws_, wd_, GRAD_TOL))
            big = np.array([v.numel() > 1 for v in gd.values()])
            np.testing.assert_allclose(np.array([float(v.double().norm()) for v in gd.values()])[big], gold['gradG_norm'][big], rtol=GRAD_TOL)
            np.testing.assert_allclose(np.array([float(v.double().norm()) for v in dpar.values()]), gold['gradD_norm'], rtol=GRAD_TOL, atol=1e-5 if c.get('ragan') else 1e-6)
    if c['norm'] == 'Batch':   # running statistics after two iterations = four training-mode calls (real, fake, real, fake; --ragan: real, fake, fake, real
        # per iteration = eight) + the reference key layout
        sd_ref, sd_hip = D.state_dict(), m.netD.state_dict()
        assert list(sd_hip.keys()) == list(sd_ref.keys())
        for k in sd_ref:
            if k.endswith('num_batches_tracked'):
                assert int(sd_hip[k]) == int(sd_ref[k]) == (8 if c.get('ragan') else 4) + (2 if c.get('wgan') else 0)   # --wgan: D(sample), a third call per discriminator step
            elif 'running' in k:
                assert rel(sd_hip[k], sd_ref[k]) < 2e-3, (k, rel(sd_hip[k], sd_ref[k]))
        # eval-mode inference with the trained statistics (translate / ddm_of fold them into the convs)
        D.eval()
        with torch.no_grad():
            want = D(real)
        got, _ = m.ddm_of(real.to(dev))
        assert rel(got.cpu(), want) < 2e-3


def test_dsn_iteration_without_colour_loss(margins):
    """--w_col 0: only the texture (0.005) and perceptual (0.01) terms drive the generator, dL/dfake is 2-3 orders smaller than with the
    colour term; the power-of-two pre-scale of the generator's 16-bit backward folds the largest loss weight in (dsn_model._GPlan.gscale), so the
    pre-scaled gradients stay in f16's normal range.  Against the oracle (pinned to the reference's modules by the fixtures above), north_star tolerances."""
    dev = _gpu()
    torch.set_num_threads(8)
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn
    from oracle.gen_golden_dsn import DSN_CASES, dsn_state, dsn_batch
    c = DSN_CASES['dsn_gau5_inst_b2_128']
    G, D = dsn.DeResnet(), dsn.Discriminator(c['k'], c['norm'], c['filter'])
    sdG, sdD = dsn_state(G.state_dict(), 21, 0.5), dsn_state(D.state_dict(), 22, 1.0)
    G.load_state_dict(sdG)
    D.load_state_dict(sdD)
    t = dsn.DSNTrainer(G, D, kernel_size=c['k'], filter_type=c['filter'], norm_layer=c['norm'], vgg_seed=78, w_col=0.0, w_per=0.01, per_type='VGG')
    m = DSNModel(dict(filter=c['filter'], kernel_size=c['k'], norm_layer=c['norm'], w_col=0.0, w_per=0.01, vgg_seed=78, per_type='VGG', allow_random_perceptual=True), device=dev)
    m.netG.load_state_dict(sdG)
    m.load_discriminator_state(sdD)
    m.netF.load_state_dict({'features.' + k: v for k, v in t.per.state_dict().items()})
    hr, bic, real = dsn_batch(c)
    t.iteration(hr, bic, real)
    m.iteration(hr.to(dev), bic.to(dev), real.to(dev))
    log = m.get_current_log()
    for k, ref_v in t.log.items():
        assert abs(log[k] - ref_v) <= 2e-3 * max(1e-3, abs(ref_v)) + 1e-5, (k, log[k], ref_v)
    wg_, ws_ = _check_grads(m.netG.params.grad_dict(), [p.grad for p in G.parameters()], 'G')
    assert m._plan(2, 128, 128).g.gscale == 65536.0   # 1 / max(w_tex, w_per) = 100 folded in: 2^7 above the 512 of w_col = 1
    margins('DSN iteration with w_col 0 (texture + perceptual terms only): worst gradient rel err G %.2e, PReLU slopes %.2e (tol %.0e)' % (wg_, ws_, GRAD_TOL))


def test_dsn_checkpoint_roundtrip(tmp_path):
    dev = _gpu()
    from dasr_amd.dsn_model import DSNModel
    torch.manual_seed(0)
    m = DSNModel(dict(w_per=0.0), device=dev)
    g = torch.Generator().manual_seed(1)
    hr, bic, real = torch.rand(1, 3, 64, 64, generator=g), torch.rand(1, 3, 16, 16, generator=g), torch.rand(1, 3, 16, 16, generator=g)
    m.iteration(hr.to(dev), bic.to(dev), real.to(dev))
    m.end_epoch()
    path = str(tmp_path / 'ck.tar')
    m.save(path)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    for k in ('epoch', 'iteration', 'model_g_state_dict', 'models_d_state_dict', 'optimizer_g_state_dict', 'optimizer_d_state_dict',
              'scheduler_g_state_dict', 'scheduler_d_state_dict'):
        assert k in ck, k
    torch.manual_seed(5)
    m2 = DSNModel(dict(w_per=0.0), device=dev)
    m2.load(path)
    m.iteration(hr.to(dev), bic.to(dev), real.to(dev))
    m2.iteration(hr.to(dev), bic.to(dev), real.to(dev))
    for k, v in m.netG.state_dict().items():
        assert torch.equal(v, m2.netG.state_dict()[k]), k
    for k, v in m.netD.state_dict().items():
        assert torch.equal(v, m2.netD.state_dict()[k]), k


@pytest.mark.parametrize('filt,arch,gen', [('gau', 'FSD', 'DeResnet'), ('wavelet', 'FSD', 'DeResnet'), ('avg_pool', 'FSD', 'DeResnet'),
                                           ('gau', 'nld_s1', 'DeResnet'), ('wavelet', 'nld_s2', 'DeResnet'), ('avg_pool', 'nld_s2', 'DSGAN'),
                                           ('gau', 'FSD', 'DSGAN'), ('gau', 'nld_s1+Batch', 'DeResnet'), ('wavelet', 'nld_s2+Batch', 'DeResnet'),
                                           ('gau', 'FSD+wgan', 'DeResnet'), ('avg_pool', 'nld_s2+Batch+wgan', 'DeResnet')])
def test_dsn_translate_and_domain_distance_map(filt, arch, gen):
    """dataset-generation inference: fake LR, discriminator map and ddm against the oracle nets + the restated
    receptive-field spreading.  +Batch: BatchNorm discriminators in eval() mode (running statistics folded into the convs; round 6 also the nld ones,
    whose BatchNorm-ed convs have no bias); +wgan: the map is the raw logit map (model.py:104-105)"""
    dev = _gpu()
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn, dsn_dataset
    from oracle.gen_golden_dsn import dsn_state
    arch, *flags = arch.split('+')
    norm, wgan = ('Batch' if 'Batch' in flags else 'Instance'), 'wgan' in flags
    G = dsn.GeneratorDSGAN() if gen == 'DSGAN' else dsn.DeResnet()
    D = dsn.Discriminator(5, norm, filt, D_arch=arch, wgan=wgan)
    sdG, sdD = dsn_state(G.state_dict(), 21, 0.5), dsn_state(D.state_dict(), 22, 1.0)
    if norm == 'Batch':   # running statistics of a trained network: not the (0, 1) of a fresh BatchNorm2d
        for k in sdD:
            if k.endswith('running_mean'):
                sdD[k] = 0.05 * torch.randn(sdD[k].shape, generator=torch.Generator().manual_seed(5))
            elif k.endswith('running_var'):
                sdD[k] = 0.5 + torch.rand(sdD[k].shape, generator=torch.Generator().manual_seed(6))
    G.load_state_dict(sdG)
    D.load_state_dict(sdD)
    D.eval()
    m = DSNModel(dict(filter=filt, w_per=0.0, discriminator=arch, generator=gen, norm_layer=norm, wgan=wgan), device=dev)
    m.netG.load_state_dict(sdG)
    m.load_discriminator_state(sdD)
    g = torch.Generator().manual_seed(31)
    if gen == 'DSGAN':
        img = torch.rand(1, 3, 46, 38, generator=g)     # the DSGAN Generator keeps the size
    elif arch == 'nld_s2':
        img = torch.rand(1, 3, 232, 200, generator=g)   # two stride-2 convs behind the wavelet front end: 29 x 25 -> 5 x 4 outputs
    else:
        img = torch.rand(1, 3, 104, 88, generator=g)
    fake, d_out, ddm = m.translate(img.to(dev))
    rf, rd, rddm = dsn_dataset.translate(G, D, img, filt, arch)
    assert rel(fake.cpu(), rf) < ACT_TOL
    assert tuple(d_out.shape) == tuple(rd.shape)
    assert rel(d_out.cpu(), torch.from_numpy(rd)) < ACT_TOL
    assert rel(ddm.cpu().double(), torch.from_numpy(rddm)) < ACT_TOL
    # discriminator map of a given LR image (source-domain ddm)
    lr = torch.rand(1, 3, 27, 34, generator=g)
    d2, ddm2 = m.ddm_of(lr.to(dev))
    lr_c = lr[..., :26, :34] if filt == 'wavelet' else lr
    with torch.no_grad():
        rd2 = D(lr_c).numpy()
    assert rel(d2.cpu(), torch.from_numpy(rd2)) < ACT_TOL
    assert rel(ddm2.cpu().double(), torch.from_numpy(dsn_dataset.domain_distance_map(rd2, lr_c.shape, filt, arch))) < ACT_TOL


def test_dsn_dataset_cli_end_to_end(tmp_path):
    _gpu()
    from dasr_amd import dsn_train, dsn_create_dataset
    save = str(tmp_path / 'dsn')
    dsn_train.main(['--debug', '--batch_size', '2', '--crop_size', '128', '--filter', 'wavelet', '--save_path', save, '--save_model_interval', '1',
                    '--no_per_loss', '--dataset', 'synthetic'])
    ck = os.path.join(save, 'checkpoints', 'last_iteration.tar')
    assert os.path.exists(ck)
    out, n = dsn_create_dataset.main(['--checkpoint', ck, '--filter', 'wavelet', '--name', 'gen', '--out_root', str(tmp_path / 'res'),
                                      '--including_source_ddm', '--n_synthetic', '2'])
    assert n == 2 and os.path.exists(os.path.join(out, 'gen.tar'))
    pngs = sorted(os.listdir(os.path.join(out, 'imgs_from_target')))
    assert pngs == ['target_000.png', 'target_001.png']
    from PIL import Image
    assert Image.open(os.path.join(out, 'imgs_from_target', pngs[0])).size == (48, 40)   # 192x160 HR / 4
    ddm = np.load(os.path.join(out, 'ddm_target', 'target_000.npy'))
    assert ddm.dtype == np.float64 and ddm.shape == (1, 1, 20, 24) and 0 < ddm.min() and ddm.max() < 1
    dds = np.load(os.path.join(out, 'ddm_source', 'source_001.npy'))
    assert dds.shape == (1, 1, 20, 24)
    # --dataset <name> resolved through the reference's paths.yml layout (create_dataset_modified.py:49-81), here realsr_tdrealsr -> PATHS['realsr']['tdrealsr']
    from PIL import Image as _I
    src, tgt = tmp_path / 'src', tmp_path / 'tgt'
    src.mkdir(), tgt.mkdir()
    g = np.random.default_rng(3)
    for i in range(2):
        _I.fromarray(g.integers(0, 255, (64, 80, 3), dtype=np.uint8)).save(str(tgt / ('t%d.png' % i)))
        _I.fromarray(g.integers(0, 255, (16, 20, 3), dtype=np.uint8)).save(str(src / ('s%d.png' % i)))
    py = tmp_path / 'paths.yml'
    py.write_text('realsr:\n  tdrealsr:\n    source: %s\n    target: %s\n' % (src, tgt))
    out2, n2 = dsn_create_dataset.main(['--checkpoint', ck, '--filter', 'wavelet', '--name', 'gen2', '--out_root', str(tmp_path / 'res'), '--including_source_ddm',
                                        '--dataset', 'realsr_tdrealsr', '--paths', str(py)])
    assert n2 == 2 and sorted(os.listdir(os.path.join(out2, 'imgs_from_target'))) == ['t0.png', 't1.png'] and sorted(os.listdir(os.path.join(out2, 'ddm_source'))) == ['s0.npy', 's1.npy']
    with pytest.raises(NotImplementedError):
        dsn_create_dataset.main(['--checkpoint', ck, '--filter', 'wavelet', '--name', 'gen3', '--out_root', str(tmp_path / 'res'), '--dataset', 'nope'])


def test_fsd_batch_discriminator_matches_reference_test_tar(golden_dir, margins):
    """Real-weights known-answer test: the FSD-Batch discriminator (codes/DSN/model.py:60-118,173-189) with the weights of the reference's
    codes/DSN/test.tar, BatchNorm in eval mode folded into the convs, on the HIP inference path (DSNModel.ddm_of, what
    create_dataset_modified.py:147-164 runs) against the output of the reference module itself (tests/golden/dsn_fsd_batch_test_tar.npz)."""
    _gpu()
    import numpy as np
    from dasr_amd.dsn_model import DSNModel
    fx = np.load(os.path.join(golden_dir, 'dsn_fsd_batch_test_tar.npz'))
    sd = {k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith('w/')}
    m = DSNModel(dict(filter='gau', kernel_size=5, norm_layer='Batch', w_per=0.0))
    m.load_discriminator_state(sd)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(97))
    dout, ddm = m.ddm_of(x.cuda())
    torch.cuda.synchronize()
    want = torch.from_numpy(fx['out'])
    e = rel(dout.cpu(), want)
    margins('FSD-Batch discriminator with the reference test.tar weights: D_out rel err %.2e (tol 1e-3), max abs %.2e' % (
        e, float((dout.cpu() - want).abs().max())))
    assert dout.shape == want.shape and e < 1e-3
    # the checkpoint can be trained on (BatchNorm in training mode) and written back in the reference layout
    sd0 = m.netD.state_dict()
    assert list(sd0.keys()) == [k for k in sd.keys()]
    m.iteration(torch.rand(2, 3, 128, 128).cuda(), torch.rand(2, 3, 32, 32).cuda(), torch.rand(2, 3, 32, 32).cuda())
    sd1 = m.netD.state_dict()
    assert int(sd1['net.net.3.num_batches_tracked']) == int(sd0['net.net.3.num_batches_tracked']) + 2
    assert not torch.equal(sd1['net.net.2.weight'], sd0['net.net.2.weight']) and not torch.equal(sd1['net.net.3.running_mean'], sd0['net.net.3.running_mean'])
    assert all(torch.isfinite(v.float()).all() for v in sd1.values())


def test_dsn_update_frequencies_follow_the_reference_loop():
    """--disc_freq 2 --gen_freq 3 (codes/DSN/train.py:55-56, 206, 229, 251): the iteration counter advances first; D steps on iterations 2 and 4, G on
    iteration 3; forward and losses run every iteration.  Checked against the oracle loop: which network moved when, and the weights after four
    iterations (Adam: a weight moves by <= lr per step)."""
    dev = _gpu()
    torch.set_num_threads(8)
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn
    from oracle.gen_golden_dsn import dsn_state, dsn_batch
    G, D = dsn.DeResnet(), dsn.Discriminator(5, 'Instance', 'gau')
    sdG, sdD = dsn_state(G.state_dict(), 21, 0.5), dsn_state(D.state_dict(), 22, 1.0)
    G.load_state_dict(sdG)
    D.load_state_dict(sdD)
    t = dsn.DSNTrainer(G, D, kernel_size=5, filter_type='gau', vgg_seed=78, w_per=0.0, disc_freq=2, gen_freq=3)
    m = DSNModel(dict(filter='gau', kernel_size=5, w_per=0.0, disc_freq=2, gen_freq=3), device=dev)
    m.netG.load_state_dict(sdG)
    m.load_discriminator_state(sdD)
    hr, bic, real = dsn_batch(dict(n=2, crop=64))
    snap = lambda net: {k: v.clone() for k, v in net.state_dict().items() if 'gaussian_filter' not in k}
    same = lambda a, b: all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a)
    g_prev, d_prev = snap(m.netG), snap(m.netD)
    moved = []
    for it in (1, 2, 3, 4):
        t.iteration(hr, bic, real)
        m.iteration(hr.to(dev), bic.to(dev), real.to(dev))
        log = m.get_current_log()
        for k, ref_v in t.log.items():
            if k in ('loss/g_tex_loss', 'loss/g_overall_loss') and it % 3:
                continue   # the generator's texture term is evaluated with its backward pass: only on generator iterations (the reference logs it there, train.py:263-268)
            assert abs(log[k] - ref_v) <= 2e-2 * max(1e-3, abs(ref_v)) + 1e-5, (it, k, log[k], ref_v)
        g_now, d_now = snap(m.netG), snap(m.netD)
        moved.append((not same(g_now, g_prev), not same(d_now, d_prev)))
        g_prev, d_prev = g_now, d_now
    assert moved == [(False, False), (False, True), (True, False), (False, True)], moved
    assert m.iteration_count == 4
    for net, ref in ((m.netG, G), (m.netD, D)):
        rsd = ref.state_dict()
        for k, v in net.state_dict().items():
            if 'gaussian_filter' in k:
                continue
            d = (v.cpu() - rsd[k]).abs().max().item()
            assert d <= 2.2e-4, (k, d)   # at most two Adam steps of lr 1e-4 apart (sign flips of ~0 gradients)
