"""Whole training steps at the BASELINE.json sizes, where the CPU oracle would take minutes: configs[1] (RRDBNet nf64 nb23, batch 16 of
128x128 LR), configs[2] (full GAN step, 32 G-crops of 128x128 LR -> 512x512) and the configs[4] per-GPU shape (DSN, batch 8 of 256x256
crops).  Size-independent property: every loss is a mean over independent samples, so the gradient of a batch equals the mean
of the gradients of its halves -- the full-size step, with its production schedule, is compared with the two half-batch steps run on
fresh models with the same weights.  Also: finite losses and bit-exact determinism of the full-size step."""
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


def test_cfg1_step_batch16_equals_mean_of_halves_and_is_deterministic(margins, monkeypatch):
    """configs[1] exactly: batch 16 -> two sub-batch streams of 8 (production schedule); the halves of 8 run as ONE batch-8 plan each"""
    dev = _gpu()
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
        assert len(m._out_plans) == (2 if hi - lo == 16 else 1)
        grads.append(m.netG.params.grad.clone())
        losses.append(m.get_current_log()['l_pix'])
        del m
        torch.cuda.empty_cache()
    assert torch.equal(grads[0], grads[1])                              # fixed-order reductions: bit-exact run to run
    assert all(torch.isfinite(x).all() for x in grads) and 0.1 < losses[0] < 1.0   # x0.1 weights: |HR - small output| ~ 0.5
    e = rel(grads[0], 0.5 * (grads[2] + grads[3]))
    margins('configs[1] full size: grad(batch 16, two streams) vs mean of the two batch-8 halves: rel err %.2e (tol 1e-5); losses %.6f vs %.6f' % (
        e, losses[0], 0.5 * (losses[2] + losses[3])))
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
