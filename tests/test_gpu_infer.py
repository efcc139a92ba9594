"""GPU: validation / inference path: test() with and without quadrant inference against the oracle net, the
validation pass of the training driver and the evaluation CLI on synthetic LR/HR pairs."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from dasr_amd import engine
    engine.ensure_runtime_ready()
    return torch.device('cuda')


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _json_opt(tmp_path, name, is_train, extra=None):
    opt = {
        'name': name, 'use_tb_logger': False, 'model': 'sr', 'scale': 4, 'gpu_ids': [0], 'chop': False, 'val_lpips': False,
        'datasets': {},
        'path': {'root': str(tmp_path), 'pretrain_model_G': None},
        'network_G': {'which_model_G': 'RRDB_net', 'norm_type': None, 'mode': 'CNA', 'nf': 32, 'nb': 1, 'in_nc': 3, 'out_nc': 3, 'gc': 32},
    }
    if is_train:
        opt['datasets'] = {'train': {'name': 'syn', 'mode': 'synthetic', 'batch_size': 4, 'HR_size': 64, 'n_batches': 8},
                           'val': {'name': 'synval', 'mode': 'synthetic', 'n_images': 2, 'LR_size': 24}}
        opt['train'] = {'lr_G': 2e-4, 'weight_decay_G': 0, 'beta1_G': 0.9, 'lr_scheme': 'MultiStepLR', 'lr_steps': [100], 'lr_gamma': 0.5,
                        'pixel_criterion': 'l1', 'pixel_weight': 1.0, 'manual_seed': 0, 'niter': 4, 'val_freq': 2}
        opt['logger'] = {'print_freq': 2, 'save_checkpoint_freq': 4}
    else:
        opt['datasets'] = {'test_1': {'name': 'synset', 'mode': 'synthetic', 'n_images': 3, 'LR_size': 64}}  # > 2 * shave (20) for the chop run
    opt.update(extra or {})
    p = tmp_path / (name + '.json')
    p.write_text(json.dumps(opt))
    return str(p)


@pytest.mark.parametrize('hw', [(52, 44), (53, 47)], ids=['52x44', '53x47_odd'])
@pytest.mark.parametrize('chop', [False, True])
def test_inference_matches_oracle(chop, hw):
    dev = _gpu()
    from oracle import fixtures, nets, util_ref
    from dasr_amd import options
    from dasr_amd.models import create_model
    opt = fixtures.make_opt('sr_nf64_nb1_b1_24x40')
    opt['gpu_ids'] = [0]
    opt['chop'] = chop
    m = create_model(options.dict_to_nonedict(opt))
    net = nets.RRDBNet(3, 3, 64, 1, 4)
    sd = fixtures.seeded_state_dict(net.state_dict(), 3, 0.1)
    net.load_state_dict(sd)
    m.netG.load_state_dict(sd)
    g = torch.Generator().manual_seed(8)
    H, W = hw
    x = torch.rand(1, 3, H, W, generator=g)
    m.feed_data({'LR': x, 'HR': torch.rand(1, 3, 4 * H, 4 * W, generator=g)}, False)
    m.test()
    with torch.no_grad():
        want = util_ref.forward_chop(x, 4, net, shave=20, min_size=320000) if chop else net(x)
    got = m.fake_H.cpu()
    assert tuple(got.shape) == tuple(want.shape)
    assert rel(got, want) < 1e-3, rel(got, want)
    vis = m.get_current_visuals()
    assert set(vis) >= {'LR', 'SR', 'HR'} and tuple(vis['SR'].shape) == (3, 4 * H, 4 * W)


def test_training_driver_validates_and_eval_cli_reports_metrics(tmp_path):
    _gpu()
    from dasr_amd import train, test as dtest
    train.main(['-opt', _json_opt(tmp_path, 'f1_train', True)])
    root = tmp_path / 'experiments' / 'f1_train'
    logs = [f for f in os.listdir(root) if f.startswith('val_') and f.endswith('.log')]
    assert logs and 'psnr:' in (root / logs[0]).read_text()
    imgs = list((root / 'val_images').rglob('*.png'))
    assert len(imgs) == 4  # 2 images x 2 validation passes
    g_path = root / 'models' / 'latest_G.pth'
    assert g_path.exists()
    summary = dtest.main(['-opt', _json_opt(tmp_path, 'f1_test', False, {'path': {'root': str(tmp_path), 'pretrain_model_G': str(g_path)}})])
    s = summary['synset']
    assert all(np.isfinite(s[k]) for k in ('psnr', 'ssim', 'psnr_y', 'ssim_y')) and 5 < s['psnr'] < 60 and 0 < s['ssim'] <= 1
    out = list((tmp_path / 'results' / 'f1_test' / 'synset' / 'imgs').glob('*.png'))
    assert len(out) == 3
    # same images through the quadrant path: PSNR within a small margin of the plain forward
    s2 = dtest.main(['-opt', _json_opt(tmp_path, 'f1_test_chop', False, {'chop': True, 'path': {'root': str(tmp_path), 'pretrain_model_G': str(g_path)}})])
    assert abs(s2['synset']['psnr'] - s['psnr']) < 1.0


def test_dasr_training_driver_with_lpips_criterion_source_discriminator_and_val_lpips(tmp_path):
    """the option surface of the shipped train_DASR*.json (feature_criterion LPIPS, val_lpips, patch discriminators for both domains) through
    `python -m dasr_amd.train`: logs, validation with the LPIPS column, the three checkpoint files and a resumable training state"""
    _gpu()
    from dasr_amd import train
    opt = json.loads(open(_json_opt(tmp_path, 'f4_dasr', True)).read())
    opt.update(model='DASR', val_lpips=True, multiweights=True, allow_random_perceptual=True, use_tb_logger=True)   # no pretrained AlexNet offline: explicit opt-in
    opt['datasets']['train'].update(batch_size=2, HR_size=128, n_batches=4)
    opt['datasets']['val'].update(LR_size=32)
    opt['path'].update(pretrain_model_D_target=None, pretrain_model_D_source=None)
    opt['network_D'] = {'which_model_D': 'discriminator_patch', 'which_model_pairD': 'discriminator_patch', 'norm_type': 'Batch', 'act_type': 'leakyrelu',
                        'mode': 'CNA', 'nf': 64, 'in_nc': 9, 'n_layers': 2}
    opt['train'].update({'lr_D': 1e-4, 'weight_decay_D': 0, 'beta1_D': 0.9, 'fs': 'wavelet', 'fs_kernel_size': 9, 'norm': True, 'sup_LL': True,
                         'pixel_LL_weight': 1, 'feature_criterion': 'LPIPS', 'feature_weight': 1, 'gan_type': 'vanilla', 'ragan': False,
                         'gan_H_target': 0.005, 'gan_H_source': 0.005, 'G_update_inter': 1, 'D_update_inter': 1, 'niter': 4, 'val_freq': 2,
                         'save_tsamples': 4})
    opt['logger']['print_freq'] = 2
    p = tmp_path / 'f4_dasr.json'
    p.write_text(json.dumps(opt))
    train.main(['-opt', str(p)])
    root = tmp_path / 'experiments' / 'f4_dasr'
    val = [f for f in os.listdir(root) if f.startswith('val_') and f.endswith('.log')]
    txt = (root / val[0]).read_text()
    assert 'psnr:' in txt and 'LPIPS(random):' in txt   # a seeded network is labelled as such, never as LPIPS
    tr = [f for f in os.listdir(root) if f.startswith('train_') and f.endswith('.log')]
    log = (root / tr[0]).read_text()
    for key in ('loss/l_g_pix', 'loss/l_g_fea', 'loss/l_g_gan_target_Hf', 'loss/l_g_gan_source_H', 'loss/l_d_target_total', 'loss/l_d_total'):
        assert key in log, key
    for f in ('latest_G.pth', 'latest_D_target.pth', 'latest_D_source.pth'):
        assert (root / 'models' / f).exists(), f
    st = torch.load(root / 'training_state' / '4.state', weights_only=False)
    assert len(st['optimizers']) == 3 and len(st['schedulers']) == 3 and st['iter'] == 4
    # round 4: the save_tsamples branch (train.py:123-172) and the tensorboard scalars / images (train.py:112-121,168,231-233)
    assert 'Saved training Samples' in log
    pngs = sorted((root / 'tsamples').glob('4_*.png'))
    assert len(pngs) == 5
    from dasr_amd import tb_writer
    tb_files = list((tmp_path / 'SRN_tb_logger' / 'f4_dasr').glob('events.out.tfevents.*'))
    assert len(tb_files) == 1
    ev = tb_writer.read_events(str(tb_files[0]))
    tags = set(t for _, t, _ in ev)
    assert {'loss/l_g_pix', 'loss/l_d_target_total', 'psnr', 'LPIPS'} <= tags and {'train/train_samples_%d' % i for i in range(5)} <= tags
    img = [v for _, t, v in ev if t == 'train/train_samples_0'][0]
    assert img[0] == 'image' and (img[1], img[2]) == (2 * 512, 3 * 512)   # [fake SR | HR | real SR] over their high-frequency views


def test_filter_high_visual_matches_reference_filter(golden_dir):
    """DASR_Model.filter_high (the `hf` / `HR_hf` visuals of get_current_visuals(tsamples=True), DASR_model.py:353-357) against the oracle's
    FilterHigh(kernel_size, gaussian=True) (architecture.py:1228-1243)"""
    dev = _gpu()
    from oracle import nets
    from dasr_amd import options
    from dasr_amd.dasr_model import DASR_Model
    m = DASR_Model.__new__(DASR_Model)
    m.opt = options.dict_to_nonedict({'train': {'fs_kernel_size': 9}})
    m.device, m.fs = dev, 'wavelet'
    x = torch.rand(2, 3, 40, 56, generator=torch.Generator().manual_seed(3))
    want = nets.FilterHigh(kernel_size=9, gaussian=True)(x)
    got = m.filter_high(x.to(dev)).cpu()
    assert rel(got, want) < 1e-5, rel(got, want)


@pytest.mark.gpu
def test_graft_entry_in_fresh_process():
    """the driver's round-end check: build() then smoke() in a fresh interpreter (the library is loaded BEFORE anything else imports
    torch there -- dasr_amd._lib must still end up on torch's HIP runtime)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.build(); g.smoke()'], cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'smoke: l_pix' in r.stdout
