"""Device-side input pipeline for the DASR unpaired dataset.

Reference: codes/SRN/data/LRHR_wavelet_unpairEq_fake_w_dataset.py:50-166 (`__getitem__`), data/util.py:78-128 (read_img, augment),
DataLoader collation.  The reference decodes four images per sample on CPU workers, resizes the domain-distance map with
cv2.INTER_LINEAR, crops, flips / transposes with numpy and ships the batch over PCIe.  Here every image (and every ddm array) is
uploaded ONCE as a CHW fp32 RGB tensor; a batch is n descriptors (image, crop origin, flags) and five `dasr_gather_crops`
launches that write the five batch tensors of the reference's dict directly in HBM.  Random draws are made on the host in the
reference's order (np.random for the unpaired indices, `random` for the crop origins and the three augmentation coins), so with
equal seeds and num_workers=0 the batches coincide.  Image decoding (PIL) stays on the host and happens once per file.
"""
import ctypes as C
import os
import random

import numpy as np
import torch

from . import _lib
from .engine import _stream, ensure_runtime_ready

IMG_EXTENSIONS = ('.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP')


def image_paths(root):
    """sorted image (and .npy) files under root (data/util.py:24-38)"""
    out = []
    for dirpath, _, fnames in sorted(os.walk(root)):
        for f in sorted(fnames):
            if f.endswith(IMG_EXTENSIONS) or f.endswith('.npy'):
                out.append(os.path.join(dirpath, f))
    return sorted(out)


def load_image(path):
    """file -> CHW fp32 RGB in [0,1] (read_img gives HWC BGR; the dataset flips to RGB at the end: same values)"""
    if path.endswith('.npy'):
        a = np.load(path)
        return torch.from_numpy(np.ascontiguousarray(a[0] if a.ndim == 4 else a)).float()
    from PIL import Image
    a = np.asarray(Image.open(path).convert('RGB'), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1).contiguous()


class DeviceUnpairedDataset:
    """Iterable of batch dicts {'LR_fake','LR_real','HR','HR_unpair','fake_w'} (CUDA tensors) built on the device.

    images: dict with lists of CHW fp32 tensors 'fake_LR', 'real_LR', 'HR' and 'fake_w' ([1,h',w'] domain-distance maps, any size)
    -- or None to read `dataroot_*` folders of `ds_opt` (PNG / .npy files)."""

    def __init__(self, ds_opt, scale=4, images=None, device=None, shuffle=None, drop_last=True):
        ensure_runtime_ready()
        self.opt, self.scale = ds_opt, scale
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.n = int(ds_opt['batch_size'])
        self.hr_size = int(ds_opt['HR_size'])
        self.use_flip, self.use_rot = bool(ds_opt.get('use_flip')), bool(ds_opt.get('use_rot'))
        self.shuffle = bool(ds_opt.get('use_shuffle')) if shuffle is None else shuffle
        if images is None:
            images = {k: [load_image(p) for p in image_paths(ds_opt[r])] for k, r in
                      (('fake_LR', 'dataroot_fake_LR'), ('real_LR', 'dataroot_real_LR'), ('HR', 'dataroot_HR'), ('fake_w', 'dataroot_fake_weights'))}
        self.img = {k: [t.to(self.device, torch.float32).contiguous() for t in v] for k, v in images.items()}
        assert self.img['HR'], 'Error: HR path is empty.'
        assert len(self.img['fake_LR']) == len(self.img['HR']) == len(self.img['fake_w'])
        self.drop_last = drop_last

    def __len__(self):
        m = len(self.img['fake_LR'])
        return m // self.n if self.drop_last else (m + self.n - 1) // self.n

    def sample(self, index):
        """descriptor of one sample; consumes the RNGs exactly like the reference's __getitem__ (train phase)"""
        s, HRs = self.scale, self.hr_size
        LRs = HRs // s
        index_real = np.random.randint(0, len(self.img['real_LR']))
        index_unpair = np.random.randint(0, len(self.img['HR']))
        lf, lr_, hr, hu, fw = (self.img['fake_LR'][index], self.img['real_LR'][index_real], self.img['HR'][index], self.img['HR'][index_unpair],
                               self.img['fake_w'][index])
        H, W = lf.shape[1:]
        Hr, Wr = lr_.shape[1:]
        y_f, x_f = random.randint(0, max(0, H - LRs)), random.randint(0, max(0, W - LRs))
        y_r, x_r = random.randint(0, max(0, Hr - LRs)), random.randint(0, max(0, Wr - LRs))
        Hu, Wu = hu.shape[1:]
        y_u, x_u = random.randint(0, max(0, Hu - HRs)), random.randint(0, max(0, Wu - HRs))
        hflip = self.use_flip and random.random() < 0.5
        vflip = self.use_rot and random.random() < 0.5
        rot90 = self.use_rot and random.random() < 0.5
        flags = int(hflip) | (int(vflip) << 1) | (int(rot90) << 2)
        return {'LR_fake': (lf, None, y_f, x_f), 'LR_real': (lr_, None, y_r, x_r), 'HR': (hr, None, y_f * s, x_f * s), 'HR_unpair': (hu, None, y_u, x_u),
                'fake_w': (fw, (H, W), y_f, x_f), 'flags': flags}

    def batch(self, indices):
        samples = [self.sample(i) for i in indices]
        out = {}
        L = _lib.lib()
        for key, size in (('LR_fake', self.hr_size // self.scale), ('LR_real', self.hr_size // self.scale), ('HR', self.hr_size), ('HR_unpair', self.hr_size),
                          ('fake_w', self.hr_size // self.scale)):
            Cc = 1 if key == 'fake_w' else 3
            descs = (_lib.CropDesc * len(samples))()
            for d, smp in zip(descs, samples):
                t, virt, y0, x0 = smp[key]
                d.src, d.C, d.H, d.W = t.data_ptr(), t.shape[0], t.shape[1], t.shape[2]
                d.vH, d.vW = virt if virt is not None else (t.shape[1], t.shape[2])
                d.y0, d.x0, d.flags = y0, x0, smp['flags']
            dd = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)
            dst = torch.empty((len(samples), Cc, size, size), dtype=torch.float32, device=self.device)
            _lib.check(L.dasr_gather_crops(dd.data_ptr(), len(samples), Cc, size, dst.data_ptr(), _stream()), 'gather_crops')
            out[key] = dst
            out.setdefault('_keep', []).append(dd)
        return out

    def __iter__(self):
        m = len(self.img['fake_LR'])
        order = list(range(m))
        if self.shuffle:
            order = torch.randperm(m).tolist()   # DataLoader(shuffle=True): RandomSampler draws from torch's global generator
        for b in range(len(self)):
            idx = order[b * self.n:(b + 1) * self.n]
            if idx:
                yield self.batch(idx)


def _bicubic_resample_matrix(n_in, scale):
    """(n_out x n_in) matrix of MATLAB's imresize along one axis: bicubic kernel (a = -0.5), widened by 1 / scale and scaled by `scale` when shrinking (antialiasing), rows
    normalised to sum 1, samples beyond the ends mirrored about the edge (the edge sample itself repeated).  Output sample k (1-based) sits at u = k / scale + (1 - 1 / scale) / 2
    in input coordinates.  What codes/SRN/data/util.py:243-297 + the per-row products of imresize_np (:367-433) compute, as one dense matrix in float64."""
    import math
    n_out = int(math.ceil(n_in * scale))
    width = 4.0 / scale if scale < 1 else 4.0
    taps = int(math.ceil(width)) + 2
    k = torch.arange(1, n_out + 1, dtype=torch.float64)
    u = k / scale + 0.5 * (1.0 - 1.0 / scale)
    left = torch.floor(u - width / 2.0)
    idx = left[:, None] + torch.arange(taps, dtype=torch.float64)[None, :]        # 1-based input positions
    d = (u[:, None] - idx) * (scale if scale < 1 else 1.0)
    a = d.abs()
    w = torch.where(a <= 1, 1.5 * a ** 3 - 2.5 * a ** 2 + 1, torch.where(a <= 2, -0.5 * a ** 3 + 2.5 * a ** 2 - 4 * a + 2, torch.zeros_like(a)))
    if scale < 1:
        w = w * scale
    w = w / w.sum(1, keepdim=True)
    j = idx.long() - 1                                                              # 0-based; mirror: -1 -> 0, -2 -> 1, n -> n - 1, n + 1 -> n - 2
    j = torch.where(j < 0, -j - 1, j)
    j = torch.where(j >= n_in, 2 * n_in - 1 - j, j).clamp_(0, n_in - 1)
    M = torch.zeros(n_out, n_in, dtype=torch.float64)
    M.scatter_add_(1, j, w)
    return M


def imresize_matlab(img, scale):
    """MATLAB-style bicubic resize with antialiasing of a CHW float tensor by `scale` (both axes; rows first, as the reference): the LR images of `mode: "LRHR"` datasets
    without an LR folder (codes/SRN/data/LRHR_dataset.py:85, util.imresize_np).  Pinned by tests/golden/imresize.npz (generated by the reference's function)."""
    C_, H, W = img.shape
    Mh, Mw = _bicubic_resample_matrix(H, scale), _bicubic_resample_matrix(W, scale)
    x = img.to(torch.float64)
    return torch.einsum('oh,chw,pw->cop', Mh, x, Mw).to(img.dtype)


class DevicePairedDataset:
    """`mode: "LRHR"` with LR files given (codes/SRN/data/LRHR_dataset.py:44-126, train phase): {'LR','HR'} batches assembled on the
    device.  Per sample the reference draws random.randint twice (crop origin in the LR image) and then util.augment's coins.
    Without `dataroot_LR` the LR images are made from the HR images by MATLAB-style bicubic down-sampling (imresize_matlab), once, at construction: the reference's train
    phase does the same per sample with random_scale_list = [1] (LRHR_dataset.py:63-88) whenever the HR size is a multiple of `scale` -- the case taken here; other sizes
    go through cv2.resize(INTER_LINEAR) there first, and an HR image smaller than HR_size is resized: both stay on the reference's side (NotImplementedError)."""

    def __init__(self, ds_opt, scale=4, images=None, device=None, shuffle=None, drop_last=True):
        ensure_runtime_ready()
        self.opt, self.scale = ds_opt, scale
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.n, self.hr_size = int(ds_opt['batch_size']), int(ds_opt['HR_size'])
        self.use_flip, self.use_rot = bool(ds_opt.get('use_flip')), bool(ds_opt.get('use_rot'))
        self.shuffle = bool(ds_opt.get('use_shuffle')) if shuffle is None else shuffle
        if images is None:
            hr = [load_image(p) for p in image_paths(ds_opt['dataroot_HR'])]
            images = {'HR': hr, 'LR': [load_image(p) for p in image_paths(ds_opt['dataroot_LR'])] if ds_opt.get('dataroot_LR') else None}
        if images.get('LR') is None:   # down-sampling on the fly (LRHR_dataset.py:63-88)
            for t in images['HR']:
                if t.shape[1] % scale or t.shape[2] % scale:
                    raise NotImplementedError('LRHR without dataroot_LR: HR image of %d x %d is not a multiple of scale %d (the reference resizes it with cv2 first); '
                                              'crop the HR images or provide LR files' % (t.shape[1], t.shape[2], scale))
            images = dict(images, LR=[imresize_matlab(t.float().cpu(), 1.0 / scale) for t in images['HR']])
        self.img = {k: [t.to(self.device, torch.float32).contiguous() for t in v] for k, v in images.items()}
        assert self.img['HR'], 'Error: HR path is empty.'
        assert len(self.img['LR']) == len(self.img['HR']), 'HR and LR datasets have different number of images - {}, {}.'.format(
            len(self.img['LR']), len(self.img['HR']))
        self.drop_last = drop_last

    def __len__(self):
        m = len(self.img['HR'])
        return m // self.n if self.drop_last else (m + self.n - 1) // self.n

    def batch(self, indices):
        s, HRs = self.scale, self.hr_size
        LRs = HRs // s
        L = _lib.lib()
        descs = {'LR': (_lib.CropDesc * len(indices))(), 'HR': (_lib.CropDesc * len(indices))()}
        for k, idx in enumerate(indices):
            lr, hr = self.img['LR'][idx], self.img['HR'][idx]
            if hr.shape[1] < HRs or hr.shape[2] < HRs:
                raise NotImplementedError('HR image smaller than HR_size (the reference resizes it and re-derives LR by imresize on the host)')
            y0, x0 = random.randint(0, max(0, lr.shape[1] - LRs)), random.randint(0, max(0, lr.shape[2] - LRs))
            hflip = self.use_flip and random.random() < 0.5
            vflip = self.use_rot and random.random() < 0.5
            rot90 = self.use_rot and random.random() < 0.5
            flags = int(hflip) | (int(vflip) << 1) | (int(rot90) << 2)
            for key, t, yy, xx in (('LR', lr, y0, x0), ('HR', hr, y0 * s, x0 * s)):
                d = descs[key][k]
                d.src, d.C, d.H, d.W, d.vH, d.vW = t.data_ptr(), t.shape[0], t.shape[1], t.shape[2], t.shape[1], t.shape[2]
                d.y0, d.x0, d.flags = yy, xx, flags
        out = {'_keep': []}
        for key, size in (('LR', LRs), ('HR', HRs)):
            dd = torch.frombuffer(bytearray(bytes(descs[key])), dtype=torch.uint8).to(self.device)
            dst = torch.empty((len(indices), 3, size, size), dtype=torch.float32, device=self.device)
            _lib.check(L.dasr_gather_crops(dd.data_ptr(), len(indices), 3, size, dst.data_ptr(), _stream()), 'gather_crops')
            out[key] = dst
            out['_keep'].append(dd)
        return out

    def __iter__(self):
        m = len(self.img['HR'])
        order = torch.randperm(m).tolist() if self.shuffle else list(range(m))
        for b in range(len(self)):
            idx = order[b * self.n:(b + 1) * self.n]
            if idx:
                yield self.batch(idx)
