"""tests/golden/dsn_ddm.npz from the reference's codes/DSN/receptive_cal.py + the handler of create_dataset_modified.py
(python -m oracle.gen_golden_dsn_dataset).  TEST INFRASTRUCTURE.  receptive_cal.py is numpy-only and imported as is; the handler is
the 10-line function at create_dataset_modified.py:14-24, which cannot be imported without running the script, so its two shape
rules are applied here around the imported getWeights / receptive_cal."""
import os
import sys

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    sys.path.insert(0, '/root/reference/codes/DSN')
    import receptive_cal as rc
    out = {}
    rs = np.random.RandomState(3)
    # conv tables of create_dataset_modified.py:112-121; the FSD cases come first (and draw first from the generator), as in round 1
    tables = {'FSD': [[5, 1, 2]] * 4, 'nld_s1': [[4, 1, 1]] * 4, 'nld_s2': [[4, 2, 1], [4, 2, 1], [4, 1, 1], [4, 1, 1]]}
    for name, (h, w), fs, arch in (('gau_23x31', (23, 31), 'gau', 'FSD'), ('wav_40x36', (40, 36), 'wavelet', 'FSD'), ('avg_12x9', (12, 9), 'avg_pool', 'FSD'),
                                   ('nld_s1_gau_26x22', (26, 22), 'gau', 'nld_s1'), ('nld_s2_avg_52x44', (52, 44), 'avg_pool', 'nld_s2'),
                                   ('nld_s2_wav_54x68', (54, 68), 'wavelet', 'nld_s2'), ('nld_s1_wav_31x24', (31, 24), 'wavelet', 'nld_s1')):
        convnet = tables[arch]
        hh, ww = (h // 2, w // 2) if fs == 'wavelet' else (h, w)
        ddm = torch.zeros((1, 1, hh, ww))
        lh, lw = rc.receptive_cal(ddm.shape[2], convnet), rc.receptive_cal(ddm.shape[3], convnet)
        d_out = rs.rand(1, 1, lh[0], lw[0])  # the discriminator map (FSD: the size of the image); batch 1: the reference broadcasts patch[:, :, i, j] against a window
        out[name + '_dout'] = d_out
        out[name + '_ddm'] = np.asarray(rc.getWeights(d_out, ddm, lh, lw))
        out[name + '_layers'] = np.array([lh, lw], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'dsn_ddm.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
