"""Multi-step GPU parity: the fixtures pin two optimiser steps; these runs follow the CPU oracle for tens of steps with changing batches, across
learning-rate milestones (SRN: MultiStepLR, base_model.py:35-37) and epoch boundaries (DSN: LambdaLR decay, codes/DSN/train.py:154-157,287-288), so
that state carried between steps (Adam moments and step counts, schedulers, repacked weights, cached plans) is compared too, not only one step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda')


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def test_sr_training_follows_the_oracle_for_40_steps(margins):
    dev = _gpu()
    torch.set_num_threads(8)
    from oracle import fixtures, nets, trainers
    from dasr_amd import options
    from dasr_amd.models import create_model
    case = dict(kind='sr', nf=32, nb=2, n=2, lr=32)
    steps = 40

    def opt_():
        o = fixtures.make_opt(case)
        o['train'].update({'lr_G': 2e-4, 'lr_steps': [10, 25], 'lr_gamma': 0.5})
        return o

    netG = nets.RRDBNet(3, 3, case['nf'], case['nb'], 4)
    sd0 = fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1)
    netG.load_state_dict(sd0)
    t = trainers.SRTrainer(opt_(), netG=netG)
    o = opt_()
    o['gpu_ids'] = [0]
    m = create_model(options.dict_to_nonedict(o))
    m.netG.load_state_dict(sd0)
    batches = [fixtures.make_batch(case, seed=100 + i) for i in range(5)]
    worst, lrs = 0.0, set()
    for step in range(1, steps + 1):
        b = batches[step % len(batches)]
        for tr in (t, m):
            tr.update_learning_rate()
            tr.feed_data(b)
            tr.optimize_parameters(step)
        got, want = m.get_current_log()['l_pix'], t.log['l_pix']
        worst = max(worst, abs(got - want) / want)
        assert abs(got - want) <= 2e-3 * want, (step, got, want)
        assert abs(m.get_current_learning_rate() - t.opt_G.param_groups[0]['lr']) < 1e-12
        lrs.add(round(m.get_current_learning_rate(), 9))
    assert lrs == {2e-4, 1e-4, 5e-5}                       # both milestones were crossed
    sd = m.netG.state_dict()
    ew = max(rel(sd[k], v) for k, v in netG.state_dict().items() if v.numel() > 64)
    moved = max(rel(v, sd0[k]) for k, v in netG.state_dict().items() if v.numel() > 64)
    margins('SR trajectory, %d steps over 5 batches, lr 2e-4 -> 1e-4 -> 5e-5: worst l_pix rel err %.2e (tol 2e-3); weights: worst tensor rel err %.2e '
            'after moving by up to %.2e' % (steps, worst, ew, moved))
    assert ew < 0.05 * max(moved, 1e-3) + 2e-3, (ew, moved)   # the difference stays a small fraction of the distance travelled


def test_dsn_training_follows_the_oracle_across_epochs(margins):
    dev = _gpu()
    torch.set_num_threads(8)
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn
    from oracle.gen_golden_dsn import dsn_state
    G, D = dsn.DeResnet(), dsn.Discriminator(5, 'Instance', 'gau')
    sdG, sdD = dsn_state(G.state_dict(), 21, 0.5), dsn_state(D.state_dict(), 22, 1.0)
    G.load_state_dict(sdG)
    D.load_state_dict(sdD)
    kw = dict(num_epochs=4, num_decay_epochs=2)
    t = dsn.DSNTrainer(G, D, kernel_size=5, filter_type='gau', norm_layer='Instance', vgg_seed=78, w_per=0.01, per_type='VGG', **kw)
    m = DSNModel(dict(filter='gau', kernel_size=5, norm_layer='Instance', w_per=0.01, vgg_seed=78, per_type='VGG', allow_random_perceptual=True, **kw), device=dev)
    m.netG.load_state_dict(sdG)
    m.load_discriminator_state(sdD)
    m.netF.load_state_dict({'features.' + k: v for k, v in t.per.state_dict().items()})
    g = torch.Generator().manual_seed(77)
    worst = {}
    for epoch in range(1, 5):
        for it in range(5):
            hr, bic, real = torch.rand(2, 3, 128, 128, generator=g), torch.rand(2, 3, 32, 32, generator=g), torch.rand(2, 3, 32, 32, generator=g)
            t.iteration(hr, bic, real)
            m.iteration(hr.to(dev), bic.to(dev), real.to(dev))
            log = m.get_current_log()
            for k, ref_v in t.log.items():
                e = abs(log[k] - ref_v) / max(abs(ref_v), 1e-3)
                worst[k] = max(worst.get(k, 0.0), e)
                assert e < 1e-2, (epoch, it, k, log[k], ref_v)
        t.end_epoch()
        m.end_epoch()
        assert abs(m.lr() - t.opt_g.param_groups[0]['lr']) < 1e-12, (epoch, m.lr(), t.opt_g.param_groups[0]['lr'])
    assert m.lr() < 1e-4                                    # the linear decay of the last num_decay_epochs epochs was reached
    eg = max(rel(v, G.state_dict()[k]) for k, v in m.netG.state_dict().items() if v.numel() > 64)
    margins('DSN trajectory, 4 epochs x 5 iterations with LambdaLR decay: worst rel err of the logged terms %s (tol 1e-2); generator weights %.2e'
            % (' '.join('%s %.1e' % (k.split('/')[-1], v) for k, v in worst.items()), eg))
    assert eg < 5e-3, eg


def test_dasr_gan_training_follows_the_oracle_for_12_steps(margins):
    """the north-star GAN step (DASR_model.py:192-330) with changing batches across the MultiStepLR milestones of G and D, and with D_update_inter = 2
    (the discriminator steps every other iteration, :284): every logged term against the fp32 oracle at every step"""
    dev = _gpu()
    torch.set_num_threads(8)
    from oracle import fixtures, nets, trainers
    from dasr_amd import options
    from dasr_amd.models import create_model
    case = dict(kind='dasr', nf=32, nb=1, n=2, lr=32, fs='wavelet', d_in_nc=9)
    steps = 12

    def opt_():
        o = fixtures.make_opt(case)
        o['train'].update({'lr_G': 2e-4, 'lr_D': 2e-4, 'lr_steps': [4, 8], 'lr_gamma': 0.5, 'D_update_inter': 2, 'vgg_seed': 77})
        return o

    netG = nets.RRDBNet(3, 3, case['nf'], case['nb'], 4)
    sdG = fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1)
    netG.load_state_dict(sdG)
    netD = nets.NLayerDiscriminator(case['d_in_nc'], n_layers=2)
    sdD = fixtures.seeded_state_dict(netD.state_dict(), 2, 1.0)
    netD.load_state_dict(sdD)
    t = trainers.DASRTrainer(opt_(), netG=netG, netD=netD, netF=None, vgg_seed=77)
    o = opt_()
    o['gpu_ids'] = [0]
    m = create_model(options.dict_to_nonedict(o))
    m.netG.load_state_dict(sdG)
    m.netD_target.load_state_dict(sdD)
    m.netF.load_state_dict({k: v for k, v in t.netF.state_dict().items() if k.startswith('features')})
    batches = [fixtures.make_batch(case, seed=300 + i) for i in range(4)]
    worst = {}
    for step in range(1, steps + 1):
        b = batches[step % len(batches)]
        t.update_learning_rate(); m.update_learning_rate()
        t.feed_data(b); m.feed_data(b, True)
        t.optimize_parameters(step); m.optimize_parameters(step)
        log = m.get_current_log()
        for k, ref_v in t.log.items():
            scorelike = k.startswith('disc_Score')   # means of logits that nearly cancel: absolute scale
            e = abs(log[k] - ref_v) / max(abs(ref_v), 1e-3)
            if scorelike:
                assert abs(log[k] - ref_v) < 5e-3, (step, k, log[k], ref_v)
                continue
            worst[k] = max(worst.get(k, 0.0), e)
            assert e < 1e-2, (step, k, log[k], ref_v)
    eg = max(rel(v, netG.state_dict()[k]) for k, v in m.netG.state_dict().items() if v.numel() > 64)
    ed = max(rel(v, netD.state_dict()[k]) for k, v in m.netD_target.state_dict().items() if v.numel() > 64)
    margins('DASR GAN trajectory, %d steps over 4 batches, lr 2e-4 -> 1e-4 -> 5e-5, D every 2nd step: worst rel err of the logged terms %s (tol 1e-2); '
            'weights G %.2e D %.2e' % (steps, ' '.join('%s %.1e' % kv for kv in worst.items()), eg, ed))
    assert eg < 1e-2 and ed < 1e-2, (eg, ed)
