"""CPU: the LPIPS restatement (oracle/lpips.py) against the fixture made by the REFERENCE's own PerceptualLossLPIPS
(oracle/gen_golden_lpips.py: reference code + its real linear heads, seeded stand-in AlexNet backbone)."""
import os

import numpy as np
import torch


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, 'lpips_alex.npz'))


def test_lpips_oracle_matches_reference_fixture(golden_dir):
    from oracle import fixtures, lpips
    from oracle.gen_golden_lpips import CASES, SEED, lpips_batch
    gold = _load(golden_dir)
    lin = [torch.from_numpy(gold['lin%d' % i]) for i in range(5)]
    assert [len(w) for w in lin] == list(lpips.CHNS) and all(float(w.min()) >= 0 for w in lin)   # LPIPS heads are non-negative by construction
    crit = lpips.PerceptualLossLPIPS(lpips.LPIPSAlex(lpips.alexnet_init_(lpips.alexnet_features(), SEED), lin))
    for name, c in CASES.items():
        x, y = lpips_batch(c)
        x.requires_grad_(True)
        l = crit(x, y)
        gx, = torch.autograd.grad(l, x)
        np.testing.assert_allclose(float(l), gold[name + '_loss'][0], rtol=1e-5)
        per = crit.net(2 * y - 1, 2 * x.detach() - 1).reshape(-1).numpy()
        np.testing.assert_allclose(per, gold[name + '_per_image'], rtol=1e-5)
        np.testing.assert_allclose(fixtures.subsample(gx).numpy(), gold[name + '_gx_sub'], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(float(gx.double().norm()), gold[name + '_gx_norm'][0], rtol=1e-5)


def test_lpips_is_zero_on_identical_images_and_symmetric():
    from oracle import lpips
    net = lpips.LPIPSAlex(seed=3)
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(1, 3, 48, 48, generator=g) * 2 - 1, torch.rand(1, 3, 48, 48, generator=g) * 2 - 1
    assert float(net(a, a)) == 0.0
    assert abs(float(net(a, b)) - float(net(b, a))) < 1e-7 and float(net(a, b)) > 0


def test_conv1_space_to_depth_rewrite_is_the_same_convolution():
    """host-side weight transform of the product (dasr_amd/lpips.py::conv1_to_s2d) against F.conv2d(k=11, s=4, p=2)"""
    import torch.nn.functional as F
    from dasr_amd.lpips import conv1_to_s2d
    g = torch.Generator().manual_seed(1)
    w, x = torch.randn(8, 3, 11, 11, generator=g), torch.randn(2, 3, 64, 72, generator=g)
    ref = F.conv2d(x, w, stride=4, padding=2)
    xp = F.pad(x, (2, 2, 2, 2))
    n, c, h, wd = xp.shape
    s2d = xp.view(n, c, h // 4, 4, wd // 4, 4).permute(0, 1, 3, 5, 2, 4).reshape(n, 48, h // 4, wd // 4)   # channel c*16 + by*4 + bx
    got = F.conv2d(s2d, conv1_to_s2d(w))
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 1e-4 * float(ref.abs().max())
