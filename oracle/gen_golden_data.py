"""tests/golden/data_pipeline.npz: batches of the reference dataset class codes/SRN/data/LRHR_wavelet_unpairEq_fake_w_dataset.py
(`__getitem__`, train phase) on synthetic in-memory images (python -m oracle.gen_golden_data).  TEST INFRASTRUCTURE.
File IO is patched (util.read_img serves arrays from a dict; np.load serves the ddm arrays); cv2 is absent: cv2.resize(INTER_LINEAR) is
a numpy stand-in with cv2's convention (half-pixel centres, edge replicate), so the ddm resize is pinned only up to that stand-in."""
import os
import random
import sys

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def resize_linear(img, dsize, interpolation=None):
    W2, H2 = dsize
    a = img if img.ndim == 3 else img[:, :, None]
    H, W = a.shape[:2]
    fy = (np.arange(H2) + 0.5) * (H / H2) - 0.5
    fx = (np.arange(W2) + 0.5) * (W / W2) - 0.5
    y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    wy, wx = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
    y1, x1 = np.clip(y0 + 1, 0, H - 1), np.clip(x0 + 1, 0, W - 1)
    y0, x0 = np.clip(y0, 0, H - 1), np.clip(x0, 0, W - 1)
    out = (1 - wy) * ((1 - wx) * a[y0][:, x0] + wx * a[y0][:, x1]) + wy * ((1 - wx) * a[y1][:, x0] + wx * a[y1][:, x1])
    return out[:, :, 0] if out.shape[2] == 1 else out   # cv2 drops a trailing singleton channel


def make_images(seed=5, n=5):
    g = np.random.RandomState(seed)
    imgs = {'fake_LR': [], 'real_LR': [], 'HR': [], 'fake_w': []}
    for i in range(n):
        h, w = 12 + 3 * i, 14 + 2 * i
        imgs['fake_LR'].append(g.rand(3, h, w).astype(np.float32))
        imgs['HR'].append(g.rand(3, 4 * h, 4 * w).astype(np.float32))
        imgs['fake_w'].append(g.rand(1, h // 2, w // 2).astype(np.float64))       # wavelet-filter ddm: half the LR size
    for i in range(4):
        imgs['real_LR'].append(g.rand(3, 11 + 5 * i, 15 + i).astype(np.float32))
    return imgs


def main():
    from .ref_import import _mod
    cv2 = _mod('cv2')
    cv2.resize, cv2.INTER_LINEAR = resize_linear, 1
    _mod('lmdb')
    class _D:
        def __init__(self, *a, **k):
            pass
    _mod('pytorch_wavelets', DWTForward=_D, DWTInverse=_D)
    sys.path[:0] = ['/root/reference/codes/SRN', '/root/reference/codes']
    import data.util as dutil
    import data.LRHR_wavelet_unpairEq_fake_w_dataset as D
    imgs = make_images()
    table = {}
    ds = D.LRHR_wavelet_Equnpair_Dataset.__new__(D.LRHR_wavelet_Equnpair_Dataset)
    ds.opt = {'scale': 4, 'HR_size': 32, 'phase': 'train', 'color': None, 'use_flip': True, 'use_rot': True}
    ds.LR_env = ds.HR_env = None
    for k, attr in (('fake_LR', 'paths_fake_LR'), ('real_LR', 'paths_real_LR'), ('HR', 'paths_HR'), ('fake_w', 'paths_fake_weights')):
        paths = []
        for i, a in enumerate(imgs[k]):
            p = '%s/%03d.%s' % (k, i, 'npy' if k == 'fake_w' else 'png')
            table[p] = a
            paths.append(p)
        setattr(ds, attr, paths)
    dutil.read_img = lambda env, path: np.ascontiguousarray(np.transpose(table[path], (1, 2, 0))[:, :, ::-1])  # HWC BGR float32 [0,1]
    D.util.read_img = dutil.read_img
    np_load = np.load
    np.load = lambda p, *a, **k: table[p][None] if p in table else np_load(p, *a, **k)   # [1,1,h,w] as create_dataset saves it
    out = {'n_images': np.array([len(imgs['fake_LR']), len(imgs['real_LR'])])}
    try:
        for case, (s1, s2) in enumerate(((11, 12), (21, 22))):
            random.seed(s1)
            np.random.seed(s2)
            items = [ds[i] for i in (0, 3, 4)]
            for key in ('LR_fake', 'LR_real', 'HR', 'HR_unpair', 'fake_w'):
                out['c%d_%s' % (case, key)] = torch.stack([it[key] for it in items]).numpy()
            out['c%d_seeds' % case] = np.array([s1, s2])
    finally:
        np.load = np_load
    # paired LRHR dataset (codes/SRN/data/LRHR_dataset.py), LR files given
    import data.LRHR_dataset as DL
    DL.util.read_img = dutil.read_img
    dl = DL.LRHRDataset.__new__(DL.LRHRDataset)
    dl.opt = {'scale': 4, 'HR_size': 32, 'phase': 'train', 'color': None, 'use_flip': True, 'use_rot': True}
    dl.LR_env = dl.HR_env = None
    dl.paths_LR, dl.paths_HR, dl.random_scale_list = ds.paths_fake_LR, ds.paths_HR, [1]
    random.seed(31)
    items = [dl[i] for i in (2, 0, 4)]
    out['p_LR'] = torch.stack([it['LR'] for it in items]).numpy()
    out['p_HR'] = torch.stack([it['HR'] for it in items]).numpy()
    np.savez_compressed(os.path.join(OUT, 'data_pipeline.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
