"""CPU: oracle/dsn.py (own restatement of the DSN nets and losses) reproduces the fixtures produced by the reference's
codes/DSN/model.py + loss.py (oracle/gen_golden_dsn.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import dsn, fixtures, nets
from oracle.gen_golden_dsn import DSN_CASES, dsn_state, collect


@pytest.mark.parametrize('case', list(DSN_CASES))
def test_dsn_oracle_matches_reference_fixture(case, golden_dir):
    torch.set_num_threads(8)
    c = DSN_CASES[case]
    ref = np.load(os.path.join(golden_dir, case + '.npz'))
    G = dsn.GeneratorDSGAN() if c.get('gen') == 'DSGAN' else dsn.DeResnet()
    D = dsn.Discriminator(c['k'], c['norm'], c['filter'], D_arch=c.get('arch', 'FSD'), cs=c.get('cs', 'cat'), wgan=bool(c.get('wgan')))
    assert list(G.state_dict().keys()) == list(ref['G_keys'])
    assert list(D.state_dict().keys()) == list(ref['D_keys'])
    G.load_state_dict(dsn_state(G.state_dict(), 21, 0.5))
    D.load_state_dict(dsn_state(D.state_dict(), 22, 1.0))
    crit = None
    if c.get('per') == 'LPIPS':
        from oracle import lpips
        crit = lpips.golden_criterion(78, golden_dir)[0]
    t = dsn.DSNTrainer(G, D, kernel_size=c['k'], filter_type=c['filter'], norm_layer=c['norm'], vgg_seed=78, per_type=c.get('per', 'VGG'), netF=crit)
    per_net = (t.lpips if crit is not None else t.per) if not c.get('rot_flip') else (lambda x, y: t.lpips(*dsn.rot_flip_pair(x, y)))
    got = collect(G, D, t.color_filter, per_net, c)
    for k in ('fake_sub', 'real_tex_sub', 'fake_tex_sub', 'losses'):
        np.testing.assert_allclose(got[k], ref[k], rtol=2e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(got['gradG_norm'], ref['gradG_norm'], rtol=1e-3, atol=1e-10)
    np.testing.assert_allclose(got['gradD_norm'], ref['gradD_norm'], rtol=1e-3, atol=1e-10)


def test_dsn_trainer_runs_two_iterations():
    torch.manual_seed(0)
    t = dsn.DSNTrainer(w_per=0.0)
    g = torch.Generator().manual_seed(1)
    hr, bic, real = torch.rand(1, 3, 64, 64, generator=g), torch.rand(1, 3, 16, 16, generator=g), torch.rand(1, 3, 16, 16, generator=g)
    t.iteration(hr, bic, real)
    l0 = dict(t.log)
    t.iteration(hr, bic, real)
    t.end_epoch()
    assert all(np.isfinite(v) for v in t.log.values()) and t.log != l0


def test_oracle_fsd_batch_reproduces_reference_test_tar(golden_dir):
    """real-weights KAT: the oracle Discriminator (FSD, BatchNorm, gaussian k=5) in eval mode with the weights of the reference's
    codes/DSN/test.tar reproduces the reference module's output"""
    import numpy as np
    fx = np.load(os.path.join(golden_dir, 'dsn_fsd_batch_test_tar.npz'))
    sd = {k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith('w/')}
    D = dsn.Discriminator(kernel_size=5, norm_layer='Batch', filter_type='gau')
    missing = D.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if 'gaussian' not in k] and not [k for k in missing.unexpected_keys if 'gaussian' not in k]
    D.eval()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(97))
    with torch.no_grad():
        y = D(x)
    np.testing.assert_allclose(y.numpy(), fx['out'], rtol=1e-5, atol=1e-6)
