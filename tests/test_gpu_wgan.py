"""DSN --wgan (codes/DSN/train.py:231-241): the second-order pieces (InstanceNorm tangent / second-order adjoint, gradient-penalty finalisation) against
torch's forward-mode / double-backward autograd, and the whole penalty -- value and weight gradients -- of the FSD discriminator against
torch.autograd.grad(..., create_graph=True) on the oracle net; the iteration fixtures (from the reference's modules) run in tests/test_gpu_dsn.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda')


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def to_blocked(x, dev):
    from dasr_amd.engine import BTensor
    N, C_, H, W = x.shape
    b = BTensor(N, C_, H, W, True, dev)
    t = torch.zeros((N, b.planes * 16, H, W))
    t[:, :C_] = x
    b.t.copy_(t.reshape(N, b.planes, 16, H, W).permute(0, 1, 3, 4, 2).contiguous().to(dev))
    return b


def test_instance_norm_tangent_and_second_order_adjoint():
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, _stream
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    N, C_, H, W = 2, 24, 9, 13
    z = (torch.randn(N, C_, H, W, generator=g) * 1.5 + 0.3).double().requires_grad_(True)
    zd = torch.randn(N, C_, H, W, generator=g).double()
    ga = torch.randn(N, C_, H, W, generator=g).double()
    fn = lambda t: F.leaky_relu(F.instance_norm(t, eps=1e-5), 0.2)
    _, adot_fm = torch.autograd.functional.jvp(fn, (z.detach(),), (zd,))            # forward-mode value (its graph is NOT differentiable in z: checked by hand)
    mean = lambda t: t.mean((2, 3), keepdim=True)
    mu = mean(z)
    r = 1.0 / torch.sqrt(mean((z - mu) ** 2) + 1e-5)
    y = (z - mu) * r
    adot = torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2)) * r * (zd - mean(zd) - y * mean(y * zd))   # lrelu'(a) J(z) zd, differentiable in z
    assert rel(adot.detach(), adot_fm) < 1e-12
    want_second, = torch.autograd.grad((adot * ga).sum(), z)    # d <ga, lrelu'(a) J(z) zd> / dz (agrees with central differences to 1e-9)
    zb, zdb, gab = to_blocked(z.detach().float(), dev), to_blocked(zd.float(), dev), to_blocked(ga.float(), dev)
    ab, out = BTensor(N, C_, H, W, True, dev), BTensor(N, C_, H, W, True, dev)
    stats = torch.zeros(N * 32 * 2, device=dev)
    _lib.check(L.dasr_inorm_lrelu_fwd(zb.view(), N, C_, H, W, 1e-5, 0.2, ab.view(), stats.data_ptr(), _stream()))
    _lib.check(L.dasr_inorm_lrelu_jvp(ab.view(), zdb.view(), N, C_, H, W, 0.2, stats.data_ptr(), out.view(), _stream()))
    torch.cuda.synchronize()
    assert rel(out.nchw(C_).cpu(), adot.detach()) < 1e-5
    _lib.check(L.dasr_inorm_second(ab.view(), zdb.view(), gab.view(), N, C_, H, W, 0.2, stats.data_ptr(), out.view(), 0, _stream()))
    torch.cuda.synchronize()
    assert rel(out.nchw(C_).cpu(), want_second) < 2e-5
    out.t.fill_(1.0)
    _lib.check(L.dasr_inorm_second(ab.view(), zdb.view(), gab.view(), N, C_, H, W, 0.2, stats.data_ptr(), out.view(), 1, _stream()))
    torch.cuda.synchronize()
    assert rel(out.nchw(C_).cpu(), want_second + 1.0) < 2e-5
    # gradient-penalty finalisation and the scaled fill
    gi = torch.randn(3, 3, 10, 12, generator=g) * 0.05
    gb = to_blocked(gi, dev)
    part, out3, acc = torch.zeros(256, device=dev), torch.zeros(4, device=dev), torch.zeros(1, device=dev)
    _lib.check(L.dasr_grad_penalty(gb.view(), 3, 3, 10, 12, 10.0, part.data_ptr(), out3.data_ptr(), acc.data_ptr(), 0, 1, _stream()))
    nrm = float(gi.double().norm())
    o = out3.cpu().tolist()
    assert abs(o[0] - nrm) < 1e-6 * nrm and abs(o[1] - 10 * (nrm - 1) ** 2) < 1e-5 and abs(o[2] - 20 * (nrm - 1) / nrm) < 1e-4 * abs(o[2]) and abs(float(acc) - o[1]) < 1e-6
    # the data-parallel form: stage 1 -> (all-reduce of out3[3]) -> stage 2; with the same tensor on "both of two ranks" the global norm is ||g|| / sqrt(2)
    out3b, acc2 = torch.zeros(4, device=dev), torch.zeros(1, device=dev)
    _lib.check(L.dasr_grad_penalty(gb.view(), 3, 3, 10, 12, 10.0, part.data_ptr(), out3b.data_ptr(), acc2.data_ptr(), 1, 2, _stream()))
    torch.cuda.synchronize()
    assert abs(float(out3b[3]) - nrm * nrm) < 1e-5 * nrm * nrm and float(acc2) == 0.0
    out3b[3] *= 2.0
    _lib.check(L.dasr_grad_penalty(gb.view(), 3, 3, 10, 12, 10.0, part.data_ptr(), out3b.data_ptr(), acc2.data_ptr(), 2, 2, _stream()))
    torch.cuda.synchronize()
    n2 = nrm / 2 ** 0.5
    o2 = out3b.cpu().tolist()
    assert abs(o2[0] - n2) < 1e-6 * n2 and abs(o2[1] - 10 * (n2 - 1) ** 2) < 1e-5 and abs(o2[2] - 20 * (n2 - 1) / n2 / 2) < 1e-4 * abs(o2[2]) and abs(float(acc2) - o2[1]) < 1e-6
    fb = BTensor(3, 16, 4, 5, True, dev)
    fb.t.fill_(7.0)
    _lib.check(L.dasr_fill_scaled(fb.view(), 3, 1, 4, 5, out3.data_ptr() + 8, 0.25, _stream()))
    torch.cuda.synchronize()
    v = fb.t.cpu()
    assert torch.allclose(v[..., 0], torch.full_like(v[..., 0], 0.25 * o[2])) and float(v[..., 1:].abs().max()) == 0.0

def test_batch_norm_tangent_and_second_order_adjoint():
    """round 6 (--wgan with --norm_layer Batch): dasr_bnorm_lrelu_jvp / dasr_bnorm_second against torch's double backward through
    leaky_relu(batch_norm(z, training)) -- the adjoint of z AND of gamma -- in fp64"""
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, _stream
    L = _lib.lib()
    g = torch.Generator().manual_seed(7)
    N, C_, H, W = 3, 24, 7, 11
    z = (torch.randn(N, C_, H, W, generator=g) * 1.5 + 0.3).double().requires_grad_(True)
    zd = torch.randn(N, C_, H, W, generator=g).double()
    ga = torch.randn(N, C_, H, W, generator=g).double()
    gamma = (torch.rand(C_, generator=g) + 0.5).double().requires_grad_(True)
    beta = (torch.randn(C_, generator=g) * 0.3).double()
    mean = lambda t: t.mean((0, 2, 3), keepdim=True)
    mu = mean(z)
    r = 1.0 / torch.sqrt(mean((z - mu) ** 2) + 1e-5)
    xh = (z - mu) * r
    y = xh * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    adot = torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2)) * gamma.view(1, -1, 1, 1) * r * (zd - mean(zd) - xh * mean(xh * zd))
    fn = lambda t: F.leaky_relu(F.batch_norm(t, None, None, gamma.detach(), beta, True, 0.1, 1e-5), 0.2)
    _, adot_fm = torch.autograd.functional.jvp(fn, (z.detach(),), (zd,))
    assert rel(adot.detach(), adot_fm) < 1e-12
    want_z, want_gamma = torch.autograd.grad((adot * ga).sum(), (z, gamma))
    zb, zdb, gab = to_blocked(z.detach().float(), dev), to_blocked(zd.float(), dev), to_blocked(ga.float(), dev)
    ab, out = BTensor(N, C_, H, W, True, dev), BTensor(N, C_, H, W, True, dev)
    stats = torch.zeros(32 * 3, device=dev)
    gm, bt = gamma.detach().float().to(dev), beta.float().to(dev)
    dgm = torch.full((C_,), 3.0, device=dev)
    _lib.check(L.dasr_bnorm_lrelu_fwd(zb.view(), N, C_, H, W, N, 1e-5, 0.2, gm.data_ptr(), bt.data_ptr(), ab.view(), stats.data_ptr(), _stream()))
    _lib.check(L.dasr_bnorm_lrelu_jvp(zb.view(), zdb.view(), N, C_, H, W, N, 0.2, gm.data_ptr(), bt.data_ptr(), stats.data_ptr(), out.view(), _stream()))
    torch.cuda.synchronize()
    assert rel(out.nchw(C_).cpu(), adot.detach()) < 1e-5
    _lib.check(L.dasr_bnorm_second(zb.view(), zdb.view(), gab.view(), N, C_, H, W, N, 0.2, gm.data_ptr(), bt.data_ptr(), stats.data_ptr(), out.view(), 0,
                                   dgm.data_ptr(), 0.5, _stream()))
    torch.cuda.synchronize()
    assert rel(out.nchw(C_).cpu(), want_z) < 2e-5
    assert rel(dgm.cpu(), 0.5 * want_gamma) < 2e-5
    out.t.fill_(1.0)
    _lib.check(L.dasr_bnorm_second(zb.view(), zdb.view(), gab.view(), N, C_, H, W, N, 0.2, gm.data_ptr(), bt.data_ptr(), stats.data_ptr(), out.view(), 1,
                                   dgm.data_ptr(), 0.5, _stream()))
    torch.cuda.synchronize()
    assert rel(out.nchw(C_).cpu(), want_z + 1.0) < 2e-5 and rel(dgm.cpu(), want_gamma) < 2e-5


@pytest.mark.parametrize('filt,arch', [('gau', 'FSD'), ('wavelet', 'FSD'), ('avg_pool', 'nld_s2'), ('gau', 'nld_s1'),
                                       ('gau', 'FSD+Batch'), ('wavelet', 'nld_s2+Batch'), ('avg_pool', 'nld_s1+Batch')])
def test_gradient_penalty_value_and_weight_gradients(filt, arch, margins):
    """the penalty and d penalty / d theta of one mixing weight against torch.autograd.grad(create_graph=True) + backward on the oracle discriminator
    (+Batch, round 6: BatchNorm2d in training mode -- group statistics over the mixed batch, gamma / beta gradients of the penalty)"""
    dev = _gpu()
    torch.set_num_threads(8)
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn
    from oracle.gen_golden_dsn import dsn_state, dsn_batch
    arch, *flags = arch.split('+')
    norm = 'Batch' if 'Batch' in flags else 'Instance'
    D = dsn.Discriminator(5, norm, filt, D_arch=arch, wgan=True)
    sdD = dsn_state(D.state_dict(), 22, 1.0)
    D.load_state_dict(sdD)
    m = DSNModel(dict(filter=filt, kernel_size=5, discriminator=arch, w_per=0.0, wgan=True, norm_layer=norm), device=dev)
    m.load_discriminator_state(sdD)
    N, crop = 2, 128
    _, fake, real = dsn_batch(dict(n=N, crop=crop))
    P = m._plan(N, crop, crop)
    # the plan's discriminator input = front end of [fake; real]: fill it through the plan's own front-end ops by feeding `fake` as G's output
    P.g.fake.t.copy_(to_blocked(fake, dev).t)
    P.real_nchw.copy_(real)
    P.bic_nchw.copy_(real)
    # run only the ops behind the generator: everything from the first op after G's forward list
    P.fwd.run(len(P.g.fwd.ops))
    r = 0.37
    P.gp.set_mix(r)
    P.gp.ops.run()
    torch.cuda.synchronize()
    sample = (r * real + (1 - r) * fake).requires_grad_(True)
    out = D(sample)
    grad = torch.autograd.grad(out.mean(), sample, create_graph=True)[0]
    pen = 10 * (grad.norm() - 1) ** 2
    dparams = [p for p in D.parameters() if p.requires_grad]
    want = torch.autograd.grad(pen, dparams, allow_unused=True)
    got_pen = float(P.gp.out3[1])
    got = m.netD.params.spec
    errs = []
    for (k, p), wv in zip([(k, p) for k, p in D.named_parameters() if p.requires_grad], want):
        o_, shape, n_ = got[k]
        gv = P.gp.grad[o_:o_ + n_].view(shape).cpu()
        if wv is None or float(wv.norm()) < 1e-9:
            assert float(gv.abs().max()) < 1e-6, k
            continue
        if k.endswith(('.2.bias', '.5.bias')):   # bias in front of an InstanceNorm / BatchNorm: it cancels in the norm, the true gradient is 0 -- rounding noise on both sides
            assert float(gv.abs().max()) < 1e-4 * max(float(x.abs().max()) for x in want if x is not None), k
            continue
        errs.append((rel(gv, wv), k))
    errs.sort(reverse=True)
    margins('DSN --wgan gradient penalty (%s front end, %s' % (filt, arch + ('' if norm == 'Instance' else ' with BatchNorm')) + '): value %.6f vs %.6f, ||g|| %.5f; worst weight-gradient rel err %.2e at %s (tol 1e-2)'
            % (got_pen, float(pen), float(P.gp.out3[0]), errs[0][0], errs[0][1]))
    assert abs(got_pen - float(pen)) < 2e-4 * abs(float(pen))
    assert errs[0][0] < 1e-2, errs[:4]


@pytest.mark.parametrize('filt', ['gau', 'wavelet'])
def test_wgan_with_ragan_iteration_matches_the_oracle(filt, margins):
    """--wgan together with --ragan (codes/DSN/train.py:221-241, model.py:98-106): relativistic logits without the sigmoid.  One iteration (both updates)
    against the fp32 oracle trainer: every logged term, the fake batch, generator and discriminator gradients at the north_star tolerances."""
    dev = _gpu()
    torch.set_num_threads(8)
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn
    from oracle.gen_golden_dsn import dsn_state, dsn_batch
    G, D = dsn.DeResnet(), dsn.Discriminator(5, 'Instance', filt, wgan=True)
    sdG, sdD = dsn_state(G.state_dict(), 21, 0.5), dsn_state(D.state_dict(), 22, 1.0)
    G.load_state_dict(sdG)
    D.load_state_dict(sdD)
    t = dsn.DSNTrainer(G, D, kernel_size=5, filter_type=filt, norm_layer='Instance', w_per=0.0, ragan=True, wgan=True)
    m = DSNModel(dict(filter=filt, kernel_size=5, norm_layer='Instance', w_per=0.0, ragan=True, wgan=True), device=dev)
    m.netG.load_state_dict(sdG)
    m.load_discriminator_state(sdD)
    hr, bic, real = dsn_batch(dict(n=3, crop=128))
    torch.manual_seed(5)
    t.iteration(hr, bic, real)          # (draws the mixing weight from torch's global RNG ...)
    torch.manual_seed(5)
    m.iteration(hr.to(dev), bic.to(dev), real.to(dev))   # (... and so does the HIP trainer: same seed, same weight)
    log = m.get_current_log()
    for k, ref_v in t.log.items():
        assert abs(log[k] - ref_v) <= 2e-3 * max(1e-3, abs(ref_v)) + 1e-5, (k, log[k], ref_v)
    e_fake = rel(m.fake.cpu(), t.fake)
    gd, dd = m.netG.params.grad_dict(), m.netD.params.grad_dict()
    eg = sorted(((rel(gd[k], p.grad), k) for k, p in G.named_parameters() if p.numel() > 1), reverse=True)
    ed = sorted(((rel(dd[k], p.grad), k) for k, p in D.named_parameters() if p.requires_grad and p.grad is not None and float(p.grad.norm()) > 1e-6), reverse=True)
    margins('DSN --wgan --ragan iteration (%s front end) vs the fp32 oracle: fake %.2e (tol 1e-3); worst gradient rel err G %.2e at %s, D %.2e at %s (tol 1e-2); d_tex_loss %.5f '
            'gradient_penalty %.5f' % (filt, e_fake, eg[0][0], eg[0][1], ed[0][0], ed[0][1], log['loss/d_tex_loss'], log['disc_score/gradient_penalty']))
    assert e_fake < 1e-3 and eg[0][0] < 1e-2 and ed[0][0] < 1e-2, (e_fake, eg[:3], ed[:3])


def test_dsn_train_cli_with_wgan_and_tensorboard_scalars(tmp_path):
    """`python -m dasr_amd.dsn_train --wgan ...` end to end: two epochs of three iterations on the synthetic loader, the log carries the gradient penalty, the
    event file the reference's tags (codes/DSN/train.py:245-270), the checkpoint loads back"""
    _gpu()
    import os
    from dasr_amd import dsn_train, tb_writer
    save = str(tmp_path / 'dsn_wgan')
    m = dsn_train.main(['--debug', '--wgan', '--batch_size', '2', '--crop_size', '128', '--filter', 'gau', '--save_path', save, '--save_model_interval', '1',
                        '--no_per_loss', '--dataset', 'synthetic', '--disc_freq', '1', '--gen_freq', '2'])
    log = m.get_current_log()
    assert m.wgan and log['disc_score/gradient_penalty'] > 0 and all(v == v for v in log.values())
    assert abs(log['loss/d_tex_loss'] - (-log['disc_score/real'] + log['disc_score/fake'] + log['disc_score/gradient_penalty'])) < 1e-4 * abs(log['loss/d_tex_loss'])
    ev = tb_writer.read_events([os.path.join(save, 'logs', f) for f in os.listdir(os.path.join(save, 'logs'))][0])
    assert {'loss/d_tex_loss', 'loss/g_tex_loss', 'disc_score/real', 'disc_score/fake', 'disc_score/gradient_penalty'} <= set(t for _, t, _ in ev)
    assert os.path.exists(os.path.join(save, 'checkpoints', 'last_iteration.tar'))
