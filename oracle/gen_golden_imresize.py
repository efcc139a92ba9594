"""tests/golden/imresize.npz: outputs of the reference's MATLAB-style bicubic down-sampling `data/util.py::imresize_np` (codes/SRN/data/util.py:367-433, what
LRHR_dataset.py:85 makes LR images with when no LR folder is given) on seeded random images (python -m oracle.gen_golden_imresize).  TEST INFRASTRUCTURE."""
import os
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    from .ref_import import _mod
    _mod('cv2')
    _mod('lmdb')
    sys.path[:0] = ['/root/reference/codes/SRN', '/root/reference/codes']
    import data.util as dutil
    g = np.random.RandomState(11)
    out = {}
    for i, (h, w, sc) in enumerate(((24, 36, 4), (32, 20, 4), (40, 28, 2), (16, 16, 4), (52, 44, 4))):
        img = g.rand(h, w, 3).astype(np.float32)
        out['in%d' % i] = img
        out['scale%d' % i] = np.int64(sc)
        out['out%d' % i] = dutil.imresize_np(img, 1.0 / sc, True).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'imresize.npz'), **out)
    print('wrote imresize.npz', {k: v.shape for k, v in out.items() if k.startswith('out')})


if __name__ == '__main__':
    main()
