"""CPU: the oracle (own fp32 restatement) reproduces the fixtures produced by the REFERENCE
code (oracle/gen_golden.py).  This is what pins the oracle (prompt section 3 / SURVEY 8(c))."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures, gen_golden, nets

CASES = list(fixtures.CASES)


@pytest.mark.parametrize('case', CASES)
def test_oracle_matches_reference_fixture(case, golden_dir):
    torch.set_num_threads(8)
    ref = np.load(os.path.join(golden_dir, case + '.npz'))
    got = gen_golden.run_oracle(case)
    assert list(ref['state_keys']) == list(got['state_keys'])
    assert list(ref['log_keys']) == list(got['log_keys'])
    # same code path class (torch CPU fp32) -> expect round-off level agreement
    np.testing.assert_allclose(got['w0_digest'], ref['w0_digest'], rtol=0, atol=0)
    np.testing.assert_allclose(got['logs'], ref['logs'], rtol=1e-6, atol=1e-7)
    for k in ref.files:
        if k.startswith('tap_'):
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-5, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(got['gradG_norm'], ref['gradG_norm'], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(got['gradG_sub'], ref['gradG_sub'], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(got['wN_digest'], ref['wN_digest'], rtol=1e-5, atol=1e-6)
    if 'dN_digest' in ref.files:
        np.testing.assert_allclose(got['gradD_norm'], ref['gradD_norm'], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(got['dN_digest'], ref['dN_digest'], rtol=1e-5, atol=1e-6)
    if 'd2N_digest' in ref.files:   # source-domain discriminator
        np.testing.assert_allclose(got['gradD2_norm'], ref['gradD2_norm'], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(got['d2N_digest'], ref['d2N_digest'], rtol=1e-5, atol=1e-6)


def test_oracle_modules_match_reference(golden_dir):
    ref = np.load(os.path.join(golden_dir, 'misc_modules.npz'))
    g = torch.Generator().manual_seed(4321)
    for nc in (3, 9):
        d = nets.NLayerDiscriminator(nc, n_layers=2)
        d.load_state_dict(fixtures.seeded_state_dict(d.state_dict(), 10 + nc, 1.0))
        y = d(torch.rand(2, nc, 64, 64, generator=g))
        assert tuple(ref['nld%d_shape' % nc]) == tuple(y.shape) == (2, 1, 14, 14)
        np.testing.assert_allclose(fixtures.subsample(y).detach().numpy(), ref['nld%d_sub' % nc], rtol=1e-5, atol=1e-6)
    for k in (5, 9):
        x = torch.rand(1, 3, 40, 40, generator=g)
        np.testing.assert_allclose(fixtures.subsample(nets.FilterLow(k, gaussian=True)(x)).numpy(), ref['flow_gau%d' % k], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(fixtures.subsample(nets.FilterHigh(k, gaussian=True)(x)).numpy(), ref['fhigh_gau%d' % k], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(fixtures.subsample(nets.FilterHigh(k)(x)).numpy(), ref['fhigh_avg%d' % k], rtol=1e-6, atol=1e-7)
    # init rule: same RNG consumption order as the reference's define_G
    torch.manual_seed(5)
    netG = nets.RRDBNet(3, 3, 32, 1, 4)
    nets.init_kaiming_(netG, 0.1)
    np.testing.assert_allclose(np.array([nets.tensor_digest(v) for v in netG.state_dict().values()]), ref['init_digest'], rtol=0, atol=0)
    f = nets.VGGFeatureExtractor(34, seed=77)
    y = f(torch.rand(1, 3, 64, 64, generator=g))
    assert tuple(ref['vgg_shape']) == tuple(y.shape)
    np.testing.assert_allclose(fixtures.subsample(y).numpy(), ref['vgg_sub'], rtol=1e-5, atol=1e-6)


def test_haar_invariants():
    """Haar convention is unpinned by the reference; check the invariants that do not depend on it."""
    x = torch.rand(2, 3, 16, 24)
    ll, hc = nets.HaarDWT()(x)
    assert ll.shape == (2, 3, 8, 12) and hc.shape == (2, 9, 8, 12)
    # orthonormal: energy preserved
    assert abs(float((x ** 2).sum()) - float((ll ** 2).sum() + (hc ** 2).sum())) < 1e-3
    assert float((ll * 0.5).min()) >= 0 and float((ll * 0.5).max()) <= 1.0 + 1e-6
    # perfect reconstruction of the top-left sample: a = (LL+LH+HL+HH)/2
    a = (ll + hc[:, 0:3] + hc[:, 3:6] + hc[:, 6:9]) * 0.5
    assert torch.allclose(a, x[:, :, 0::2, 0::2], atol=1e-6)


def test_precision_probe_winograd_emulation_is_a_convolution():
    """oracle/precision_probe.py::winograd_conv (the F(2x2,3x3) emulation behind the Winograd precision figure in DESIGN 4.8): with exact
    transforms it must equal the direct 3x3 / pad 1 convolution of the bf16-rounded operands; with re-rounded transformed operands it must stay
    within a few bf16 ulps of it"""
    import torch
    import torch.nn.functional as F
    from oracle.precision_probe import winograd_conv
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(2, 8, 8, 12, generator=g), torch.randn(5, 8, 3, 3, generator=g)
    ref = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), None, 1, 1)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(winograd_conv(x, w, 'f32'), ref) < 1e-6
    assert 1e-4 < rel(winograd_conv(x, w, 'bf16'), ref) < 1e-2
