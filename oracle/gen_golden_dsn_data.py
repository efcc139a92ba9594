"""Generate tests/golden/dsn_imresize.npz from the REFERENCE `imresize` (codes/DSN/utils.py:101-160), imported from /root/reference with
stand-ins for the torchvision names its module imports at the top (never called here).  TEST INFRASTRUCTURE (see oracle/__init__.py).
    python -m oracle.gen_golden_dsn_data
"""
import os
import sys

import numpy as np
import torch

from .ref_import import _mod

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
CASES = [((3, 64, 64), 0.25, True), ((3, 40, 52), 0.25, True), ((3, 37, 53), 0.25, True), ((3, 32, 48), 0.5, True), ((3, 96, 128), 0.25, True),
         ((3, 16, 12), 2.0, True), ((3, 21, 10), 4.0, True)]   # (antialiasing=False trips over an empty mirror patch in the reference: not a case its datasets use)


def main():
    _mod('torchvision')
    _mod('torchvision.transforms', Compose=None, ToTensor=None, ToPILImage=None, CenterCrop=None, Resize=None)
    sys.path.insert(0, '/root/reference/codes/DSN')
    import utils as rutils
    fx = {}
    g = torch.Generator().manual_seed(515)
    for i, (shape, scale, aa) in enumerate(CASES):
        x = torch.rand(*shape, generator=g)
        if i == 1:
            x = (x * 255).round() / 255   # 8-bit image values, as the datasets see them
        if i == 2:
            x = x * 1.2 - 0.1             # leaves [0, 1]: the final clamp matters
        y = rutils.imresize(x, scale, aa)
        fx['x%d' % i], fx['y%d' % i], fx['cfg%d' % i] = x.numpy(), y.numpy(), np.array([scale, float(aa)])
        print(shape, scale, aa, '->', tuple(y.shape))
    fx['valid_crop'] = np.array([rutils.calculate_valid_crop_size(c, u) for c, u in ((255, 4), (256, 4), (101, 2), (7, 4))])
    np.savez_compressed(os.path.join(OUT, 'dsn_imresize.npz'), **fx)


if __name__ == '__main__':
    main()
