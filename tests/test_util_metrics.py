"""CPU: validation helpers -- dasr_amd/util.py and the oracle restatement against the fixture produced by the
reference's own codes/SRN/utils/util.py / data/util.py (oracle/gen_golden_util.py)."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'util_metrics.npz'))


def test_tensor2img_psnr_ssim_ycbcr_match_reference(gold):
    from dasr_amd import util
    from oracle import util_ref
    sr, hr = torch.from_numpy(gold['sr']), torch.from_numpy(gold['hr'])
    a, b = util.tensor2img(sr), util.tensor2img(hr)
    assert np.array_equal(a, gold['sr_img']) and np.array_equal(b, gold['hr_img'])
    assert np.array_equal(util_ref.tensor2img(sr), gold['sr_img'])
    af, bf, c = a / 255., b / 255., 4
    assert abs(util.calculate_psnr(af[c:-c, c:-c] * 255, bf[c:-c, c:-c] * 255) - float(gold['psnr'])) < 1e-9
    assert abs(util_ref.psnr(af[c:-c, c:-c] * 255, bf[c:-c, c:-c] * 255) - float(gold['psnr'])) < 1e-9
    assert abs(util.calculate_ssim(af[c:-c, c:-c] * 255, bf[c:-c, c:-c] * 255) - float(gold['ssim'])) < 1e-9
    assert abs(util_ref.ssim(af[c:-c, c:-c] * 255, bf[c:-c, c:-c] * 255) - float(gold['ssim'])) < 1e-9
    ay, by = util.bgr2ycbcr(af, only_y=True), util.bgr2ycbcr(bf, only_y=True)
    np.testing.assert_allclose(ay, gold['sr_y'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(util_ref.bgr2y(af), gold['sr_y'], rtol=0, atol=1e-6)
    assert abs(util.calculate_psnr(ay[c:-c, c:-c] * 255, by[c:-c, c:-c] * 255) - float(gold['psnr_y'])) < 1e-4
    assert abs(util.calculate_ssim(ay[c:-c, c:-c] * 255, by[c:-c, c:-c] * 255) - float(gold['ssim_y'])) < 1e-6
    assert np.array_equal(util.bgr2ycbcr(a, only_y=False), gold['sr_ycbcr_u8'])
    # the input must not be modified (the reference scales float inputs in place)
    assert af.max() <= 1.0


def test_ssim_properties():
    from dasr_amd import util
    g = np.random.RandomState(0)
    x = (g.rand(32, 36) * 255).round()
    assert abs(util.calculate_ssim(x, x) - 1.0) < 1e-12
    y = np.clip(x + g.randn(32, 36) * 20, 0, 255)
    s = util.calculate_ssim(x, y)
    assert 0 < s < 1 and abs(s - util.calculate_ssim(y, x)) < 1e-12
    assert util.calculate_psnr(x, x) == float('inf')
    with pytest.raises(ValueError):
        util.calculate_ssim(x, x[:-1])


def test_forward_chop_matches_reference(gold):
    """quadrant inference + stitching through the oracle RRDBNet (fp32 CPU): identical to the reference function's output"""
    from dasr_amd import util
    from oracle import nets, fixtures, util_ref
    net = nets.RRDBNet(3, 3, 32, 1, 4)
    net.load_state_dict(fixtures.seeded_state_dict(net.state_dict(), 5, 0.1))
    x = torch.from_numpy(gold['chop_x'])
    with torch.no_grad():
        for ms, key in ((100000, 'chop_y'), (100, 'chop_y_rec')):
            y = util.forward_chop(x, 4, net, shave=3, min_size=ms)
            np.testing.assert_allclose(y.numpy(), gold[key], rtol=0, atol=2e-6)
            y2 = util_ref.forward_chop(x, 4, net, shave=3, min_size=ms)
            np.testing.assert_allclose(y2.numpy(), gold[key], rtol=0, atol=2e-6)
        # away from the seams the chopped result equals the plain forward (receptive field permitting), everywhere it is close
        full = net(x)
    assert float((y - full).abs().max()) < 0.2


def test_make_grid_layout():
    from dasr_amd import util
    t = torch.arange(4 * 3 * 2 * 2, dtype=torch.float32).view(4, 3, 2, 2) / 100
    g = util.make_grid(t, nrow=2)
    assert tuple(g.shape) == (3, 2 * 4 + 2, 2 * 4 + 2)
    assert torch.equal(g[:, 2:4, 2:4], t[0]) and torch.equal(g[:, 2:4, 6:8], t[1]) and torch.equal(g[:, 6:8, 2:4], t[2])


def test_matlab_imresize_matches_the_reference(golden_dir):
    """dasr_amd.data.imresize_matlab (LR images of LRHR datasets without an LR folder) against the reference's data/util.py::imresize_np on five seeded images
    (tests/golden/imresize.npz, oracle/gen_golden_imresize.py): x1/4 and x1/2, sizes where the antialiased kernel reaches over both edges"""
    import numpy as np
    import torch
    from dasr_amd.data import imresize_matlab
    g = np.load(os.path.join(golden_dir, 'imresize.npz'))
    for i in range(5):
        x = torch.from_numpy(g['in%d' % i]).permute(2, 0, 1)
        y = imresize_matlab(x, 1.0 / int(g['scale%d' % i])).permute(1, 2, 0).numpy()
        assert y.shape == g['out%d' % i].shape and float(np.abs(y - g['out%d' % i]).max()) < 5e-7, i
