"""GPU parity of the DSN driver's validation pass (codes/DSN/train.py:293-355) and its image-folder datasets end to end: the forward-only generator
pass, the six validation terms against the oracle's modules (oracle/dsn.py), and `python -m dasr_amd.dsn_train --dataset aim2019` on PNG folders."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from dasr_amd import engine
    engine.ensure_runtime_ready()
    return torch.device('cuda')


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize('filt,per,size', [('gau', 'VGG', 64), ('avg_pool', 'VGG', 72), ('wavelet', 'LPIPS', 64), ('wavelet', 'VGG', 54)])
def test_validation_terms_match_the_oracle(filt, per, size, golden_dir, margins):
    """fake = clamp(G(hr), 0, 1) and mse / psnr / rgb / mean / perceptual / colour errors (train.py:311-321) vs the oracle's generator,
    GeneratorLoss filters and perceptual nets.  size 54: an LR side that is no multiple of 4 and pools 54 -> 27 -> 13 -> 6 -> 3 -> 1 in the VGG16 term."""
    dev = _gpu()
    torch.set_num_threads(8)
    from dasr_amd.dsn_model import DSNModel
    from oracle import dsn, nets
    from oracle.gen_golden_dsn import dsn_state
    G = dsn.DeResnet()
    sdG = dsn_state(G.state_dict(), 21, 0.5)
    G.load_state_dict(sdG)
    crit, sdF = None, None
    if per == 'LPIPS':
        from oracle import lpips
        crit, sdF = lpips.golden_criterion(78, golden_dir)
    t = dsn.DSNTrainer(G, None, kernel_size=5, filter_type=filt, vgg_seed=78, w_per=0.01, per_type=per, netF=crit)
    m = DSNModel(dict(filter=filt, kernel_size=5, w_per=0.01, vgg_seed=78, per_type=per, allow_random_perceptual=True), device=dev)
    m.netG.load_state_dict(sdG)
    m.netF.load_state_dict(sdF if sdF is not None else {'features.' + k: v for k, v in t.per.state_dict().items()})
    g = torch.Generator().manual_seed(31)
    for n in (1, 2):   # the reference validates with batch size 1; a larger batch must give the batch-mean forms of the same terms
        hr = torch.rand(n, 3, 4 * size, 4 * size, generator=g)
        target = torch.rand(n, 3, size, size, generator=g)
        with torch.no_grad():
            fake_ref = G(hr).clamp(0, 1)
            mse = ((fake_ref - target) ** 2).mean()
            want = {'mse': mse, 'psnr': -10 * torch.log10(mse),
                    'rgb_error': F.l1_loss(fake_ref.mean(3).mean(2), target.mean(3).mean(2)),
                    'mean_error': F.l1_loss(fake_ref.view(n, -1).mean(1), target.view(n, -1).mean(1)),
                    'perceptual_error': t.lpips(fake_ref, target) if per == 'LPIPS' else F.mse_loss(t.per(fake_ref), t.per(target)),
                    'color_error': F.l1_loss(t.color_filter(fake_ref), t.color_filter(target))}
        fake = m.generate(hr.to(dev)).clamp(0, 1)
        e_fake = rel(fake.cpu(), fake_ref)
        assert e_fake < 1e-3, e_fake
        got = m.validation_metrics(fake, target.to(dev))
        assert list(got.keys()) == list(want.keys())
        errs = {}
        for k in want:
            errs[k] = abs(float(got[k]) - float(want[k])) / max(abs(float(want[k])), 1e-6)
            assert errs[k] < 2e-3, (k, float(got[k]), float(want[k]))
        margins('DSN validation %s/%s %dpx n=%d: fake %.2e; ' % (filt, per, size, n, e_fake) + ' '.join('%s %.1e' % kv for kv in errs.items()) + ' (tol 2e-3)')
    # the image strips' filters (train.py:141-142: FilterLow / FilterHigh with include_pad=False, same size)
    lo, hi = dsn.FilterLow(5, include_pad=False, gaussian=filt == 'gau'), dsn.FilterHigh(5, include_pad=False, gaussian=filt == 'gau')
    if filt != 'wavelet':
        with torch.no_grad():
            assert rel(m.filter_low(fake).cpu(), lo(fake.cpu())) < 1e-6 and rel(m.filter_high(fake).cpu(), hi(fake.cpu())) < 1e-6
    if per == 'VGG':   # vgg16.features[:31] has five pools: a 16-pixel LR crop (crop_size 64) leaves nothing -- refused when the plan is built
        with pytest.raises(ValueError, match='too small'):
            m.netF.plan(2, 2, 16, 16)


def test_generate_keeps_two_shapes_and_leaves_training_plans_alone():
    dev = _gpu()
    from dasr_amd.dsn_model import DSNModel
    m = DSNModel(dict(filter='gau', w_per=0.0), device=dev)
    g = torch.Generator().manual_seed(3)
    hr, bic, real = torch.rand(2, 3, 64, 64, generator=g), torch.rand(2, 3, 16, 16, generator=g), torch.rand(2, 3, 16, 16, generator=g)
    m.iteration(hr.to(dev), bic.to(dev), real.to(dev))
    train_plan = m.netG.plans[(2, 64, 64)]
    outs = {}
    for s in (64, 48, 80, 96, 64):
        x = torch.rand(2, 3, s, s, generator=torch.Generator().manual_seed(s)).to(dev)
        outs.setdefault(s, []).append(m.generate(x).clone())
    assert torch.equal(outs[64][0], outs[64][1])                      # same weights, same input -> same image after the plan was rebuilt
    assert m.netG.plans[(2, 64, 64)] is train_plan                     # the shape the training plan uses is never evicted
    assert (2, 48, 48) not in m.netG.plans and len(m._gen_lru) == 2    # older validation shapes are
    assert bool(torch.isnan(m.validation_metrics(outs[64][0].clamp(0, 1), bic.to(dev))['perceptual_error']))   # w_per = 0: no perceptual net was built


def test_dsn_train_cli_on_image_folders_with_validation(tmp_path):
    """`--dataset aim2019` from a paths.yml (train.py:82-91): PNG folders -> dsn_data loaders (2 worker processes) -> iterations; the validation pass
    writes the six `val/*` scalars and, on val_img_interval epochs, the image strips; checkpoint + TensorBoard file as the reference lays them out"""
    _gpu()
    import yaml
    from PIL import Image
    from dasr_amd import dsn_train, tb_writer
    rng = np.random.RandomState(5)
    dirs = {}
    for sub, n, (h, w) in (('src', 6, (140, 150)), ('tgt', 4, (140, 132)), ('vhr', 3, (130, 150)), ('vlr', 3, (36, 40))):
        d = tmp_path / sub
        d.mkdir()
        for i in range(n):
            Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(str(d / ('%02d.png' % i)))
        dirs[sub] = str(d)
    paths = tmp_path / 'paths.yml'
    paths.write_text(yaml.safe_dump({'aim2019': {'tdsr': {'source': dirs['src'], 'target': dirs['tgt'], 'valid_hr': dirs['vhr'], 'valid_lr': dirs['vlr']}}}))
    save = str(tmp_path / 'exp')
    m = dsn_train.main(['--dataset', 'aim2019', '--artifacts', 'tdsr', '--paths', str(paths), '--batch_size', '4', '--crop_size', '128', '--crop_size_val', '128',
                        '--num_epochs', '2', '--num_decay_epochs', '1', '--val_interval', '1', '--val_img_interval', '2', '--save_model_interval', '2',
                        '--flips', '--rotations', '--num_workers', '2', '--filter', 'gau', '--per_type', 'VGG', '--allow_random_perceptual',
                        '--save_path', save])
    assert m.iteration_count == 4 and m.epoch == 2                     # 6 source images / batch 4 -> 2 iterations per epoch (the second one short)
    assert os.path.exists(os.path.join(save, 'checkpoints', 'iteration_4.tar'))
    ev = tb_writer.read_events([os.path.join(save, 'logs', f) for f in os.listdir(os.path.join(save, 'logs'))][0])
    tags = {}
    for step, tag, val in ev:
        tags.setdefault(tag, []).append((step, val))
    for k in ('val/mse', 'val/psnr', 'val/rgb_error', 'val/mean_error', 'val/perceptual_error', 'val/color_error'):
        assert [s for s, _ in tags[k]] == [2, 4] and all(np.isfinite(v) for _, v in tags[k]), (k, tags.get(k))
    assert abs(tags['val/psnr'][0][1] + 10 * np.log10(tags['val/mse'][0][1])) < 0.5     # (mean of psnr vs psnr of the mean mse: same ballpark)
    img = tags['val/target_fake_crop_low_high_0']
    assert len(img) == 1 and img[0][0] == 4 and img[0][1][0] == 'image'
    h, w = img[0][1][1], img[0][1][2]
    assert (h, w) == (3 * 405 + 5, 5 * 405 + 5)                        # 3 validation images x 5 views, 400 x 400 tiles with 5 pixels of padding
    assert 'param/learning_rate' in tags and 'loss/color_loss' in tags
