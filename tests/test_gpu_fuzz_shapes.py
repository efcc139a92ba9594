"""Whole-step parity on randomly drawn shapes: the fixtures and the full-size tests use a handful of sizes (24x40, 32, 64, 128; DSN crops 128 / 160 / 256); here
batch size, image height / width (not multiples of the kernels' tiles), depth and filter are drawn per seed and one training step is compared with the CPU
oracle -- activations 1e-3, gradients 1e-2 (north_star), losses 2e-3.  The seeds are fixed: every run tests the same shapes."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
ACT_TOL, GRAD_TOL = 1e-3, 1e-2


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda')


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize('seed', range(10))
def test_sr_step_on_random_shapes(seed, margins):
    dev = _gpu()
    torch.set_num_threads(8)
    from oracle import fixtures, nets, trainers
    from dasr_amd import options
    from dasr_amd.models import create_model
    rng = random.Random(4000 + seed)
    case = dict(kind='sr', nf=rng.choice([32, 64]), nb=rng.choice([1, 2]), n=rng.choice([1, 2, 3, 5]) if seed < 8 else seed,   # seeds 8, 9: batch 8 / 9 = the two-stream schedule (9: uneven sub-batches)
                lr=(rng.randint(6, 44), rng.randint(6, 52)),
                pix=rng.choice(['l1', 'l1', 'l2']), upsample_mode=rng.choice(['upconv', 'upconv', 'pixelshuffle']))
    netG = nets.RRDBNet(3, 3, case['nf'], case['nb'], 4, upsample_mode=case['upsample_mode'])
    sd0 = fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1)
    netG.load_state_dict(sd0)
    t = trainers.SRTrainer(fixtures.make_opt(case), netG=netG)
    o = fixtures.make_opt(case)
    o['gpu_ids'] = [0]
    m = create_model(options.dict_to_nonedict(o))
    m.netG.load_state_dict(sd0)
    batch = fixtures.make_batch(case, seed=77 + seed)
    for tr in (t, m):
        tr.update_learning_rate()
        tr.feed_data(batch)
        tr.optimize_parameters(1)
    e_sr = rel(m.fake_H.cpu(), t.fake_H.detach())
    got, want = m.get_current_log()['l_pix'], t.log['l_pix']
    gd = m.netG.params.grad_dict()
    worst = max(rel(gv, p.grad) for gv, p in zip(gd.values(), netG.parameters()))
    margins('SR step on a drawn shape %s: SR %.2e (tol %.0e), worst gradient %.2e (tol %.0e), loss %.1e' % (case, e_sr, ACT_TOL, worst, GRAD_TOL, abs(got - want) / want))
    assert e_sr < ACT_TOL and worst < GRAD_TOL and abs(got - want) <= 1e-4 * want, (case, e_sr, worst, got, want)
    assert len(m._out_plans) == (2 if case['n'] >= 8 else 1)


@pytest.mark.parametrize('seed', range(6))
def test_dsn_iteration_on_random_shapes(seed, margins):
    dev = _gpu()
    torch.set_num_threads(8)
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn
    from oracle.gen_golden_dsn import dsn_state
    rng = random.Random(5000 + seed)
    filt = rng.choice(['gau', 'avg_pool', 'wavelet'])
    arch = rng.choice(['FSD', 'FSD', 'nld_s1', 'nld_s2'])
    n = rng.choice([1, 2, 3])
    lh, lw = rng.randint(12, 46), rng.randint(12, 46)
    if filt == 'wavelet':
        lh, lw = lh // 2 * 2, lw // 2 * 2          # Haar front end: even LR sides (the reference pads odd ones; the HIP front end refuses them)
    if arch != 'FSD':
        lh, lw = max(lh, 24), max(lw, 24)          # the 4x4 discriminators need a few pixels after their strides
    k = rng.choice([3, 5, 7])
    G, D = dsn.DeResnet(), dsn.Discriminator(k, 'Instance', filt, D_arch=arch)
    sdG, sdD = dsn_state(G.state_dict(), 21, 0.5), dsn_state(D.state_dict(), 22, 1.0)
    G.load_state_dict(sdG)
    D.load_state_dict(sdD)
    t = dsn.DSNTrainer(G, D, kernel_size=k, filter_type=filt, norm_layer='Instance', w_per=0.0)
    m = DSNModel(dict(filter=filt, kernel_size=k, norm_layer='Instance', w_per=0.0, discriminator=arch), device=dev)
    m.netG.load_state_dict(sdG)
    m.load_discriminator_state(sdD)
    g = torch.Generator().manual_seed(900 + seed)
    hr, bic, real = torch.rand(n, 3, 4 * lh, 4 * lw, generator=g), torch.rand(n, 3, lh, lw, generator=g), torch.rand(n, 3, lh, lw, generator=g)
    t.iteration(hr, bic, real)
    m.iteration(hr.to(dev), bic.to(dev), real.to(dev))
    log = m.get_current_log()
    for key, ref_v in t.log.items():
        assert abs(log[key] - ref_v) <= 2e-3 * max(1e-3, abs(ref_v)) + 1e-5, (seed, key, log[key], ref_v)
    e_fake = rel(m.fake.cpu(), t.fake)
    from test_gpu_dsn import _check_grads
    wg, ws = _check_grads(m.netG.params.grad_dict(), [p.grad for p in G.parameters()], 'G')
    dpar = dict((kk, v) for kk, v in m.netD.params.grad_dict().items() if 'gaussian_filter' not in kk)
    wd, _ = _check_grads(dpar, [p.grad for p in D.parameters() if p.requires_grad], 'D')
    margins('DSN iteration on a drawn shape (%s k%d %s, n %d, LR %dx%d): fake %.2e, gradients G %.2e slopes %.2e D %.2e' % (filt, k, arch, n, lh, lw, e_fake, wg, ws, wd))
    assert e_fake < ACT_TOL


@pytest.mark.parametrize('seed', range(4))
def test_dasr_gan_step_on_random_shapes(seed, margins):
    dev = _gpu()
    torch.set_num_threads(8)
    from oracle import fixtures, nets, trainers
    from dasr_amd import options
    from dasr_amd.models import create_model
    rng = random.Random(6000 + seed)
    fs = rng.choice(['wavelet', 'gau'])
    case = dict(kind='dasr', nf=32, nb=1, n=rng.choice([1, 2, 3]), lr=(rng.randint(16, 36), rng.randint(16, 36)), fs=fs, d_in_nc=9 if fs == 'wavelet' else 3)
    netG = nets.RRDBNet(3, 3, case['nf'], case['nb'], 4)
    sdG = fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1)
    netG.load_state_dict(sdG)
    netD = nets.NLayerDiscriminator(case['d_in_nc'], n_layers=2)
    sdD = fixtures.seeded_state_dict(netD.state_dict(), 2, 1.0)
    netD.load_state_dict(sdD)
    t = trainers.DASRTrainer(fixtures.make_opt(case), netG=netG, netD=netD, netF=None, vgg_seed=77)
    o = fixtures.make_opt(case)
    o['gpu_ids'] = [0]
    o['train']['vgg_seed'] = 77
    m = create_model(options.dict_to_nonedict(o))
    m.netG.load_state_dict(sdG)
    m.netD_target.load_state_dict(sdD)
    m.netF.load_state_dict({k: v for k, v in t.netF.state_dict().items() if k.startswith('features')})
    batch = fixtures.make_batch(case, seed=55 + seed)
    t.update_learning_rate(); m.update_learning_rate()
    t.feed_data(batch); m.feed_data(batch, True)
    t.optimize_parameters(1); m.optimize_parameters(1)
    log = m.get_current_log()
    for key, ref_v in t.log.items():
        atol = 2e-4 if key.startswith('disc_Score') else 1e-5
        assert abs(log[key] - ref_v) <= 2e-3 * max(1e-3, abs(ref_v)) + atol, (seed, key, log[key], ref_v)
    e_sr = rel(m.fake_H.cpu(), t.fake_H.detach())
    wg = max(rel(gv, p.grad) for gv, p in zip(m.netG.params.grad_dict().values(), netG.parameters()))
    dd = [(kk, gv, p.grad) for (kk, gv), p in zip(m.netD_target.params.grad_dict().items(), netD.parameters())]
    wd = max(rel(gv, pg) for kk, gv, pg in dd if float(pg.double().norm()) > 1e-6)   # (biases in front of an InstanceNorm: true gradient 0)
    margins('GAN step on a drawn shape %s: SR %.2e, worst gradient G %.2e D %.2e' % (case, e_sr, wg, wd))
    assert e_sr < ACT_TOL and wg < GRAD_TOL and wd < GRAD_TOL, (case, e_sr, wg, wd)
