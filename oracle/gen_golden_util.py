"""tests/golden/util_metrics.npz from the reference's own codes/SRN/utils/util.py and data/util.py (python -m oracle.gen_golden_util).
TEST INFRASTRUCTURE.  cv2 is absent: ssim's cv2.getGaussianKernel / cv2.filter2D are served by scipy stand-ins (noted in the fixture)."""
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    from scipy.ndimage import correlate
    from .ref_import import _mod
    from . import nets, fixtures
    cv2 = _mod('cv2')

    def getGaussianKernel(n, sigma):
        x = np.arange(n, dtype=np.float64) - (n - 1) / 2
        k = np.exp(-x * x / (2 * sigma * sigma))
        return (k / k.sum()).reshape(n, 1)

    def filter2D(img, ddepth, kernel):  # default border of cv2.filter2D = BORDER_REFLECT_101 = scipy 'mirror'; correlation
        if img.ndim == 2:
            return correlate(img, kernel, mode='mirror')
        return np.stack([correlate(img[:, :, c], kernel, mode='mirror') for c in range(img.shape[2])], 2)
    cv2.getGaussianKernel, cv2.filter2D = getGaussianKernel, filter2D
    tv = _mod('torchvision')
    _mod('lmdb')
    _mod('torchvision.utils', make_grid=lambda *a, **k: None)
    sys.path[:0] = ['/root/reference/codes/SRN', '/root/reference/codes']
    import utils.util as rutil
    import data.util as dutil
    import torch.nn.parallel as P
    P.data_parallel = lambda model, x, ids=None: model(x)     # single device: the quadrant goes straight through the model
    g = torch.Generator().manual_seed(99)
    sr = torch.rand(3, 40, 44, generator=g) * 1.2 - 0.1
    hr = (sr + 0.05 * torch.randn(3, 40, 44, generator=g)).clamp(0, 1)
    a, b = rutil.tensor2img(sr), rutil.tensor2img(hr)
    af, bf = a / 255., b / 255.
    c = 4
    out = {'sr': sr.numpy(), 'hr': hr.numpy(), 'sr_img': a, 'hr_img': b,
           'psnr': rutil.calculate_psnr(af[c:-c, c:-c] * 255, bf[c:-c, c:-c] * 255),
           'ssim': rutil.calculate_ssim(af[c:-c, c:-c] * 255, bf[c:-c, c:-c] * 255)}
    ay, by = dutil.bgr2ycbcr(af.copy(), only_y=True), dutil.bgr2ycbcr(bf.copy(), only_y=True)
    out.update(sr_y=ay, psnr_y=rutil.calculate_psnr(ay[c:-c, c:-c] * 255, by[c:-c, c:-c] * 255),
               ssim_y=rutil.calculate_ssim(ay[c:-c, c:-c] * 255, by[c:-c, c:-c] * 255))
    ycc = dutil.bgr2ycbcr(a.copy(), only_y=False)
    out['sr_ycbcr_u8'] = ycc
    # forward_chop through a small RRDBNet (the oracle net; the function under test is the chopping/stitching)
    net = nets.RRDBNet(3, 3, 32, 1, 4)
    net.load_state_dict(fixtures.seeded_state_dict(net.state_dict(), 5, 0.1))
    x = torch.rand(1, 3, 24, 20, generator=g)
    with torch.no_grad():
        y = rutil.forward_chop(x, 4, net, shave=3, min_size=100000)
        y2 = rutil.forward_chop(x, 4, net, shave=3, min_size=100)   # forces one level of recursion
    out.update(chop_x=x.numpy(), chop_y=y.numpy(), chop_y_rec=y2.numpy())
    np.savez_compressed(os.path.join(OUT, 'util_metrics.npz'), **out)
    print({k: (v if np.isscalar(v) else getattr(v, 'shape', None)) for k, v in out.items()})


if __name__ == '__main__':
    main()
