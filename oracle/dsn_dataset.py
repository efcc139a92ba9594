"""CPU restatement of the DSN dataset generation step.  TEST INFRASTRUCTURE (see oracle/__init__.py).

receptive field walk   : codes/DSN/receptive_cal.py:10-25, 45-52   (outFromIn / receptive_cal)
domain-distance map    : codes/DSN/receptive_cal.py:34-60          (weights_matrix / getWeights)
handler, conv tables   : codes/DSN/create_dataset_modified.py:14-24, 119-126
"""
import math

import numpy as np
import torch

CONVNETS = {'FSD': [[5, 1, 2]] * 4, 'nld_s1': [[4, 1, 1]] * 4, 'nld_s2': [[4, 2, 1], [4, 2, 1], [4, 1, 1], [4, 1, 1]]}


def receptive(imsize, convnet):
    """(n features, jump, receptive field, centre of the first feature) after the conv table"""
    n, j, r, start = imsize, 1, 1, 0.5
    for k, s, p in convnet:
        n_out = math.floor((n - k + 2 * p) / s) + 1
        pad_l = math.floor(((n_out - 1) * s - n + k) / 2)
        j, r, start, n = j * s, r + (k - 1) * j, start + ((k - 1) / 2 - pad_l) * j, n_out
    return n, j, r, start


def spread(patch, shape, lay_h, lay_w):
    """every patch value added over its receptive-field window (the reference takes jump / rf / start of the WIDTH walk for both axes)"""
    out = np.zeros(shape)
    n_h, (n_w, jump, rf, start) = lay_h[0], lay_w
    for i in range(n_h):
        for j in range(n_w):
            h0, h1 = int(max(0, start + i * jump - rf // 2)), int(start + i * jump + rf - rf // 2)
            w0, w1 = int(max(0, start + j * jump - rf // 2)), int(start + j * jump + rf - rf // 2)
            out[:, :, h0:h1, w0:w1] += patch[:, :, i, j][:, :, None, None]
    return out


def domain_distance_map(d_out, img_shape, fs_type='gau', arch='FSD'):
    n, _, h, w = img_shape
    if fs_type.lower() == 'wavelet':
        h, w = h // 2, w // 2
    elif fs_type.lower() not in ('gau', 'avg_pool'):
        raise NotImplementedError('Frequency Separation [{:s}] not recognized'.format(fs_type))
    shape = (n, 1, h, w)
    lh, lw = receptive(h, CONVNETS[arch]), receptive(w, CONVNETS[arch])
    return spread(d_out, shape, lh, lw) / spread(np.ones_like(d_out), shape, lh, lw)


def translate(G, D, img, fs_type, arch='FSD'):
    with torch.no_grad():
        fake = G(img)
        d_out = D(fake).numpy()
    return fake, d_out, domain_distance_map(d_out, fake.shape, fs_type, arch)
