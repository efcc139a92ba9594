"""Whole training steps at the BASELINE.json sizes, where the CPU oracle would take minutes: configs[1] (RRDBNet nf64 nb23, batch 16 of
128x128 LR), configs[2] (full GAN step, 32 G-crops of 128x128 LR -> 512x512) and the configs[4] per-GPU shape (DSN, batch 8 of 256x256
crops).  Size-independent property: every loss is a mean over independent samples, so the gradient of a batch equals the mean
of the gradients of its halves -- the full-size step, with its production schedule, is compared with the two half-batch steps run on
fresh models with the same weights.  Also: finite losses and bit-exact determinism of the full-size step.

Round 4: the GPU box has 256 host cores and 3 TB of memory, so the oracle CAN run the exact configurations (in chunks where its autograd state would
not fit comfortably): test_cfg1_exact_* (46 s), test_cfg2_exact_* (3 min), test_cfg4_exact_* (6 s) compare the production steps of configs[1], configs[2]
and configs[4]'s per-GPU shape with the fp32 CPU oracle directly -- SR / fake batch, every logged loss, every gradient tensor."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda')


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _shard(batch, lo, hi):
    return {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in batch.items()}


def _sr_model(nf, nb):
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    opt = fixtures.make_opt(dict(kind='sr', nf=nf, nb=nb, n=1, lr=8))
    opt['gpu_ids'] = [0]
    m = create_model(options.dict_to_nonedict(opt))
    m.netG.load_state_dict(fixtures.seeded_state_dict(m.netG.state_dict(), 1, 0.1))
    return m


@pytest.mark.parametrize('chain', ['1', '0'], ids=['chained_trunk', 'two_streams'])
def test_cfg1_step_batch16_equals_mean_of_halves_and_is_deterministic(chain, margins, monkeypatch):
    """configs[1] exactly.  chained_trunk (production schedule since round 4): batch 16 = ONE plan whose trunk runs as two persistent chained launches
    (dasr_conv_chain); two_streams (DASR_CHAIN=0, the schedule of rounds 1-3): two sub-batch streams of 8.  The halves of 8 run as one batch-8 plan each:
    8 x 32 tiles do not fill the layer-by-layer chain's grid, so since round 6 their trunk runs as the input-stationary chained launches (dasr_rdb_chain)
    -- the comparison below is therefore also layer form against input-stationary form."""
    dev = _gpu()
    if chain == '1' and torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip('the chained launches need a whole 256-CU MI355X (RRDBNetHIP.chain_ok)')
    monkeypatch.setenv('DASR_CHAIN', chain)
    g = torch.Generator().manual_seed(1234)
    data = {'LR': torch.rand(16, 3, 128, 128, generator=g), 'HR': torch.rand(16, 3, 512, 512, generator=g)}
    grads, losses = [], []
    for lo, hi in ((0, 16), (0, 16), (0, 8), (8, 16)):
        monkeypatch.setenv('DASR_STREAMS', '2' if hi - lo == 16 else '1')
        m = _sr_model(64, 23)
        m.update_learning_rate()
        m.feed_data(_shard(data, lo, hi))
        m.optimize_parameters(1)
        torch.cuda.synchronize()
        assert len(m._out_plans) == (2 if (hi - lo == 16 and chain == '0') else 1)
        ch = m._out_plans[0].chain
        assert (ch is not None) == (chain == '1') and (ch is None or ch.form == ('layer' if hi - lo == 16 else 'is'))
        m.check_finite()
        grads.append(m.netG.params.grad.clone())
        losses.append(m.get_current_log()['l_pix'])
        del m
        torch.cuda.empty_cache()
    assert torch.equal(grads[0], grads[1])                              # fixed-order reductions: bit-exact run to run
    assert losses[0] == losses[1]                                      # ... and so is the logged loss (fixed-order grid sum, csrc/common.h)
    assert all(torch.isfinite(x).all() for x in grads) and 0.1 < losses[0] < 1.0   # x0.1 weights: |HR - small output| ~ 0.5
    e = rel(grads[0], 0.5 * (grads[2] + grads[3]))
    margins('configs[1] full size: grad(batch 16, %s) vs mean of the two batch-8 halves: rel err %.2e (tol 1e-5); losses %.6f vs %.6f' % (
        'chained trunk' if chain == '1' else 'two streams', e, losses[0], 0.5 * (losses[2] + losses[3])))
    assert e < 1e-5 and abs(losses[0] - 0.5 * (losses[2] + losses[3])) < 5e-6   # the logged loss is an fp32 atomic sum over 12.6 M terms


def test_cfg2_gan_step_32_crops_equals_mean_of_halves(margins):
    """configs[2]: n = 16 source + 16 target crops (32 through G), wavelet frequency separation, VGG19-54 (seeded), patch discriminator"""
    dev = _gpu()
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    case = dict(kind='dasr', nf=64, nb=23, n=16, lr=128, fs='wavelet', d_in_nc=9)
    n = 16
    batch = fixtures.make_batch(case)
    out = []
    for lo, hi in ((0, n), (0, n // 2), (n // 2, n)):
        opt = fixtures.make_opt(case)
        opt['gpu_ids'] = [0]
        opt['train']['vgg_seed'] = 77
        m = create_model(options.dict_to_nonedict(opt))
        m.netG.load_state_dict(fixtures.seeded_state_dict(m.netG.state_dict(), 1, 0.1))
        m.netD_target.load_state_dict(fixtures.seeded_state_dict(m.netD_target.state_dict(), 2, 1.0))
        m.update_learning_rate()
        m.feed_data(_shard(batch, lo, hi), True)
        m.optimize_parameters(1)
        torch.cuda.synchronize()
        out.append((m.netG.params.grad.clone(), m.netD_target.params.grad.clone(), dict(m.get_current_log())))
        del m
        torch.cuda.empty_cache()
    (gG, gD, log), (gGa, gDa, la), (gGb, gDb, lb) = out
    eG, eD = rel(gG, 0.5 * (gGa + gGb)), rel(gD, 0.5 * (gDa + gDb))
    worst_log = max(abs(log[k] - 0.5 * (la[k] + lb[k])) / max(1e-3, abs(log[k])) for k in log)
    margins('configs[2] full size (32 G-crops @128^2): grad G / D vs mean of halves rel err %.2e / %.2e (tol 1e-4); worst log entry %.2e (tol 1e-4); '
            'l_g_pix %.4f l_g_fea %.4f l_d %.4f' % (eG, eD, worst_log, log['loss/l_g_pix'], log['loss/l_g_fea'], log['loss/l_d_target_total']))
    assert all(v == v and abs(v) < 1e4 for v in log.values())
    assert eG < 1e-4 and eD < 1e-4 and worst_log < 1e-4


def test_cfg2_gan_step_and_its_log_are_deterministic():
    """configs[2] twice from the same weights and batch: gradients, weights AND every entry of get_current_log() bit-identical (the
    logged terms were fp32 atomicAdd sums whose order varied from run to run; now fixed-order grid sums)"""
    dev = _gpu()
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    case = dict(kind='dasr', nf=64, nb=23, n=16, lr=128, fs='wavelet', d_in_nc=9)
    batch = fixtures.make_batch(case)
    out = []
    for _ in range(2):
        opt = fixtures.make_opt(case)
        opt['gpu_ids'] = [0]
        opt['train']['vgg_seed'] = 77
        m = create_model(options.dict_to_nonedict(opt))
        m.netG.load_state_dict(fixtures.seeded_state_dict(m.netG.state_dict(), 1, 0.1))
        m.netD_target.load_state_dict(fixtures.seeded_state_dict(m.netD_target.state_dict(), 2, 1.0))
        for step in (1, 2):
            m.update_learning_rate()
            m.feed_data(batch, True)
            m.optimize_parameters(step)
        log = dict(m.get_current_log())
        out.append((m.netG.params.grad.clone(), m.netD_target.params.grad.clone(), m.netG.params.flat.clone(), log))
        del m
        torch.cuda.empty_cache()
    (gG, gD, wG, l0), (gG1, gD1, wG1, l1) = out
    assert torch.equal(gG, gG1) and torch.equal(gD, gD1) and torch.equal(wG, wG1)
    assert list(l0.keys()) == list(l1.keys()) and len(l0) >= 5
    for k in l0:
        assert l0[k] == l1[k], (k, l0[k], l1[k])


@pytest.mark.parametrize('bwd16', [1, 0], ids=['f16_backward_default', 'fp32_tensor_backward'])
def test_dsn_iteration_batch8_of_256_equals_mean_of_halves(bwd16, margins, monkeypatch):
    """configs[4] per-GPU shape: batch 8 of 256x256 HR crops, wavelet filter, VGG16 term on.  InstanceNorm / all losses are per sample.
    The full batch and its halves differ by ~3e-5 already in dL/dfake (different reduction orders in the losses).  The fp32-tensor backward
    (DASR_DSN_BWD16=0, ~fp32 operands) carries that through: 1e-4.  The default backward rounds the gradient stream to f16 (11 bits) per layer:
    a 3e-5 perturbation flips roundings, so the two runs differ like two draws of the rounding noise -- ~1e-4 on the conv weight gradients
    (averaged over 5e5 pixels), up to ~1e-3 on the scalar PReLU slope gradients (sums that cancel); both far inside the 1e-2 parity budget
    (profiles/r03_parity_margins.log: 16-bit vs fp32-tensor backward 3-5e-4 / 4.7e-3)."""
    dev = _gpu()
    monkeypatch.setenv('DASR_DSN_BWD16', str(bwd16))
    from dasr_amd.dsn_model import DSNModel
    from oracle.gen_golden_dsn import dsn_state, dsn_batch
    hr, bic, real = dsn_batch(dict(n=8, crop=256))
    out = []
    for lo, hi in ((0, 8), (0, 4), (4, 8)):
        torch.manual_seed(0)
        m = DSNModel(dict(filter='wavelet', w_per=0.01, vgg_seed=78, allow_random_perceptual=True), device=dev)
        assert m.netG.bwd16 == bool(bwd16)
        m.netG.load_state_dict(dsn_state(m.netG.state_dict(), 21, 0.5))
        m.netD.load_state_dict(dsn_state(m.netD.state_dict(), 22, 1.0))
        m.iteration(hr[lo:hi].to(dev), bic[lo:hi].to(dev), real[lo:hi].to(dev))
        torch.cuda.synchronize()
        out.append((m.netG.params.grad_dict(), m.netD.params.grad_dict(), dict(m.get_current_log())))
        del m
        torch.cuda.empty_cache()
    (gG, gD, log), (gGa, gDa, la), (gGb, gDb, lb) = out
    worst, worst_slope = 0.0, 0.0
    for full, a, b in ((gG, gGa, gGb), (gD, gDa, gDb)):
        for k in full:
            want = 0.5 * (a[k] + b[k])
            if float(want.double().norm()) < 1e-7 * max(1.0, float(full[k].numel()) ** 0.5):
                continue   # biases in front of an InstanceNorm: true gradient 0, rounding noise on both sides
            if full is gG and full[k].numel() == 1:
                worst_slope = max(worst_slope, rel(full[k], want))
            else:
                worst = max(worst, rel(full[k], want))
    wl = max(abs(log[k] - 0.5 * (la[k] + lb[k])) / max(1e-3, abs(log[k])) for k in log)
    tol, tol_slope = (3e-4, 3e-3) if bwd16 else (1e-4, 1e-4)
    margins('DSN full size (batch 8 of 256^2, %s backward): worst per-tensor grad rel err vs mean of halves %.2e (tol %.0e), PReLU slopes %.2e (tol %.0e), '
            'worst log entry %.2e (tol 1e-4)' % ('f16' if bwd16 else 'fp32-tensor', worst, tol, worst_slope, tol_slope, wl))
    assert worst < tol and worst_slope < tol_slope and wl < 1e-4


def test_cfg1_exact_step_matches_the_oracle(margins):
    """configs[1] EXACTLY -- RRDBNet nf64 nb23, batch 16 of 128 x 128 LR, the production two-stream schedule with the deferred grouped weight
    gradients -- against the fp32 CPU oracle (until round 3 the oracle comparison stopped at batch 2 @ 128^2 and batch 8 @ 32^2; the
    full size was property-checked only, HIP against HIP).  The oracle evaluates the batch in four chunks of four crops (the loss is a mean over
    independent samples: the gradient of the batch is the sum of the chunk gradients of l * chunk / batch), ~5 GB of autograd state at a time.
    Compared: SR output, loss, all 702 gradient tensors, north_star tolerances."""
    dev = _gpu()
    torch.set_num_threads(max(8, min(32, (torch.get_num_threads() or 8))))
    from oracle import fixtures, nets
    g = torch.Generator().manual_seed(1234)
    data = {'LR': torch.rand(16, 3, 128, 128, generator=g), 'HR': torch.rand(16, 3, 512, 512, generator=g)}
    m = _sr_model(64, 23)
    sd0 = m.netG.state_dict()
    m.update_learning_rate()
    m.feed_data(data)
    m.optimize_parameters(1)
    torch.cuda.synchronize()
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert len(m._out_plans) == 1 and m._out_plans[0].chain is not None and m._out_plans[0].chain_b is not None   # the production schedule: chained trunk launches
    m.check_finite()
    got_g = m.netG.params.grad_dict()
    got_loss = m.get_current_log()['l_pix']
    got_sr = m.fake_H.cpu()
    ref = nets.RRDBNet(3, 3, 64, 23, 4)
    ref.load_state_dict(sd0)
    loss, srs = 0.0, []
    for c0 in range(0, 16, 4):
        sr = ref(data['LR'][c0:c0 + 4])
        l = (sr - data['HR'][c0:c0 + 4]).abs().mean() * (4.0 / 16.0)
        l.backward()
        loss += float(l)
        srs.append(sr.detach())
    e_sr = rel(got_sr, torch.cat(srs, 0))
    errs = sorted(((rel(got_g[k], p.grad), k) for k, p in ref.named_parameters()), reverse=True)
    margins('configs[1] exactly (nf64 nb23, batch 16 x 128^2, chained trunk launches) vs the fp32 oracle: SR rel err %.2e (tol 1e-3), loss %.6f vs %.6f, worst gradient rel err %.2e '
            'at %s (tol 1e-2, %d tensors), median %.2e' % (e_sr, got_loss, loss, errs[0][0], errs[0][1], len(errs), errs[len(errs) // 2][0]))
    assert e_sr < 1e-3 and abs(got_loss - loss) < 2e-5 * abs(loss) + 1e-6
    assert errs[0][0] < 1e-2, errs[:3]


def test_cfg2_exact_gan_step_matches_the_oracle(margins):
    """configs[2] EXACTLY -- the north-star GAN step: RRDBNet nf64 nb23, n = 16 source + 16 target crops of 128 x 128 LR (32 through G), wavelet frequency
    separation, VGG19-54 features (seeded), NLayer patch discriminator -- against the fp32 CPU oracle of DASR_Model.optimize_parameters (DASR_model.py:192-330;
    oracle/trainers.py, pinned to the reference by the step fixtures).  ~60 GB and about two minutes of host time on the GPU box (3 TB / 256 cores).
    Compared after one step: every logged loss, the SR batch, all G and D_target gradient tensors, north_star tolerances."""
    dev = _gpu()
    torch.set_num_threads(64)
    from oracle import fixtures, nets, trainers
    from dasr_amd import options
    from dasr_amd.models import create_model
    case = dict(kind='dasr', nf=64, nb=23, n=16, lr=128, fs='wavelet', d_in_nc=9)
    netG = nets.RRDBNet(3, 3, 64, 23, 4)
    sdG = fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1)
    netG.load_state_dict(sdG)
    netD = nets.NLayerDiscriminator(9, n_layers=2)
    sdD = fixtures.seeded_state_dict(netD.state_dict(), 2, 1.0)
    netD.load_state_dict(sdD)
    t = trainers.DASRTrainer(fixtures.make_opt(case), netG=netG, netD=netD, netF=None, vgg_seed=77)
    batch = fixtures.make_batch(case)
    opt = fixtures.make_opt(case)
    opt['gpu_ids'] = [0]
    opt['train']['vgg_seed'] = 77
    m = create_model(options.dict_to_nonedict(opt))
    m.netG.load_state_dict(sdG)
    m.netD_target.load_state_dict(sdD)
    m.netF.load_state_dict({k: v for k, v in t.netF.state_dict().items() if k.startswith('features')})
    m.update_learning_rate()
    m.feed_data(batch, True)
    m.optimize_parameters(1)
    torch.cuda.synchronize()
    log = dict(m.get_current_log())
    gd, dd = m.netG.params.grad_dict(), m.netD_target.params.grad_dict()
    sr = m.fake_H.cpu()
    t.update_learning_rate()
    t.feed_data(batch)
    t.optimize_parameters(1)
    worst_log = 0.0
    for k, ref_v in t.log.items():
        scorelike = k.startswith('disc_Score')
        tol = 2e-3 * max(1e-3, abs(ref_v)) + (2e-4 if scorelike else 1e-5)
        worst_log = max(worst_log, abs(log[k] - ref_v) / tol)
        assert abs(log[k] - ref_v) <= tol, (k, log[k], ref_v)
    e_sr = rel(sr, t.fake_H.detach())
    eg = sorted(((rel(gd[k], p.grad), k) for k, p in netG.named_parameters()), reverse=True)
    ed = sorted(((rel(dd[k], p.grad), k) for k, p in netD.named_parameters()), reverse=True)
    margins('configs[2] exactly (nf64 nb23, 16 + 16 crops @128^2, VGG19-54, patch D) vs the fp32 oracle: SR rel err %.2e (tol 1e-3); worst gradient rel err G %.2e at %s '
            '(702 tensors, median %.2e), D_target %.2e at %s (tol 1e-2); logged losses within %.2f of their tolerance: %s'
            % (e_sr, eg[0][0], eg[0][1], eg[len(eg) // 2][0], ed[0][0], ed[0][1], worst_log, ' '.join('%s %.5g' % (k.split('/')[-1], v) for k, v in log.items())))
    assert e_sr < 1e-3 and eg[0][0] < 1e-2 and ed[0][0] < 1e-2, (e_sr, eg[:3], ed[:3])


@pytest.mark.parametrize('per_type', ['VGG', 'LPIPS', 'LPIPS-one_f16_pass'])
def test_cfg4_exact_dsn_iteration_matches_the_oracle(per_type, margins, golden_dir, monkeypatch):
    """configs[4]'s per-GPU shape EXACTLY -- De_resnet (8 blocks) + FSD discriminator (wavelet front end), batch 8 of 256 x 256 HR crops, colour / texture /
    perceptual (VGG16 MSE or the reference default LPIPS) losses -- one iteration against the fp32 CPU oracle (codes/DSN/train.py:204-285 as oracle/dsn.py fixes
    it): losses, fake LR batch, all generator and discriminator gradients, north_star tolerances."""
    dev = _gpu()
    torch.set_num_threads(32)
    label = per_type
    if per_type.endswith('one_f16_pass'):   # DASR_DSN_FWD16=2 (opt-in, dsn_model.DeResnetHIP): the residual blocks' forward in one f16 pass; inside the budget at this shape
        monkeypatch.setenv('DASR_DSN_FWD16', '2')
        per_type = 'LPIPS'
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn
    from oracle.gen_golden_dsn import dsn_state, dsn_batch
    G, D = dsn.DeResnet(), dsn.Discriminator(5, 'Instance', 'wavelet')
    sdG, sdD = dsn_state(G.state_dict(), 21, 0.5), dsn_state(D.state_dict(), 22, 1.0)
    G.load_state_dict(sdG)
    D.load_state_dict(sdD)
    crit, sdF = None, None
    if per_type == 'LPIPS':
        from oracle import lpips
        crit, sdF = lpips.golden_criterion(78, golden_dir)
    t = dsn.DSNTrainer(G, D, kernel_size=5, filter_type='wavelet', norm_layer='Instance', vgg_seed=78, w_per=0.01, per_type=per_type, netF=crit)
    m = DSNModel(dict(filter='wavelet', kernel_size=5, norm_layer='Instance', w_per=0.01, vgg_seed=78, per_type=per_type, allow_random_perceptual=True), device=dev)
    assert m.netG.fwd1p == (os.environ.get('DASR_DSN_FWD16') == '2')
    m.netG.load_state_dict(sdG)
    m.load_discriminator_state(sdD)
    m.netF.load_state_dict(sdF if sdF is not None else {'features.' + k: v for k, v in t.per.state_dict().items()})
    hr, bic, real = dsn_batch(dict(n=8, crop=256))
    t.iteration(hr, bic, real)
    m.iteration(hr.to(dev), bic.to(dev), real.to(dev))
    log = m.get_current_log()
    for k, ref_v in t.log.items():
        assert abs(log[k] - ref_v) <= 2e-3 * max(1e-3, abs(ref_v)) + 1e-5, (k, log[k], ref_v)
    e_fake = rel(m.fake.cpu(), t.fake)
    gd, dd = m.netG.params.grad_dict(), m.netD.params.grad_dict()
    eg = sorted(((rel(gd[k], p.grad), k) for k, p in G.named_parameters() if p.numel() > 1), reverse=True)
    slopes = rel(torch.cat([gd[k].flatten() for k, p in G.named_parameters() if p.numel() == 1]), torch.cat([p.grad.flatten() for p in G.parameters() if p.numel() == 1]))
    ed = sorted(((rel(dd[k], p.grad), k) for k, p in D.named_parameters() if p.requires_grad and float(p.grad.norm()) > 1e-6), reverse=True)
    margins('configs[4] exactly (DSN batch 8 x 256^2, wavelet FSD, %s term) vs the fp32 oracle: fake rel err %.2e (tol 1e-3); worst gradient rel err G %.2e at %s, PReLU slopes '
            '(jointly) %.2e, D %.2e at %s (tol 1e-2)' % (label, e_fake, eg[0][0], eg[0][1], slopes, ed[0][0], ed[0][1]))
    assert e_fake < 1e-3 and eg[0][0] < 1e-2 and ed[0][0] < 1e-2 and slopes < 1e-2, (e_fake, eg[:3], slopes, ed[:3])
