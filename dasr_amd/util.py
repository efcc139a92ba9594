"""Validation / inference helpers of the SRN trainer boundary: image conversion, PSNR / SSIM, quadrant inference.

Reference: codes/SRN/utils/util.py:87-147 (forward_chop), :180-204 (tensor2img), :236-291 (calculate_psnr / ssim / calculate_ssim),
codes/SRN/data/util.py:169-190 (bgr2ycbcr).  The metrics are host-side numpy on uint8-range images (they are not on the hot path);
the network forward underneath test() / forward_chop runs on the HIP kernels.  cv2 is not required: the 11x11 Gaussian window
(sigma 1.5) and the 'valid' filtering that the reference obtains from cv2.getGaussianKernel / cv2.filter2D(...)[5:-5, 5:-5] are
written out with numpy (the cropped region never sees cv2's border handling, so the result is the same).
"""
import math
import os

import numpy as np
import torch


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def make_grid(t, nrow, padding=2):
    """4-D batch -> one image, row-major tiles with `padding` zero pixels (what torchvision.utils.make_grid(normalize=False) builds)"""
    n, c, h, w = t.shape
    if n == 1:
        return t[0]
    xmaps = min(nrow, n)
    ymaps = int(math.ceil(float(n) / xmaps))
    H, W = h + padding, w + padding
    grid = t.new_zeros((c, H * ymaps + padding, W * xmaps + padding))
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * H + padding:y * H + padding + h, x * W + padding:x * W + padding + w] = t[k]
            k += 1
    return grid


def tensor2img(tensor, out_type=np.uint8, min_max=(0, 1)):
    """RGB tensor 4D (B,3|1,H,W) / 3D (C,H,W) / 2D (H,W), any range -> HWC BGR (or HW) numpy, [0,255] uint8 by default"""
    tensor = tensor.squeeze().float().cpu().clamp_(*min_max)
    tensor = (tensor - min_max[0]) / (min_max[1] - min_max[0])
    n_dim = tensor.dim()
    if n_dim == 4:
        img = make_grid(tensor, nrow=int(math.sqrt(len(tensor)))).numpy()
        img = np.transpose(img[[2, 1, 0], :, :], (1, 2, 0))
    elif n_dim == 3:
        img = np.transpose(tensor.numpy()[[2, 1, 0], :, :], (1, 2, 0))
    elif n_dim == 2:
        img = tensor.numpy()
    else:
        raise TypeError('Only support 4D, 3D and 2D tensor. But received with dimension: {:d}'.format(n_dim))
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return img.astype(out_type)


def save_img(img, img_path, mode='RGB'):
    """BGR (or gray) uint8 array -> PNG (the reference writes through cv2.imwrite)"""
    from PIL import Image
    arr = img[:, :, ::-1] if img.ndim == 3 else img
    Image.fromarray(np.ascontiguousarray(arr)).save(img_path)


def bgr2ycbcr(img, only_y=True):
    """MATLAB-style BT.601 conversion of a BGR image; uint8 [0,255] or float [0,1] in, same type out"""
    in_type = img.dtype
    x = img.astype(np.float32)
    if in_type != np.uint8:
        x = x * 255.0
    if only_y:
        out = np.dot(x, [24.966, 128.553, 65.481]) / 255.0 + 16.0
    else:
        out = np.matmul(x, [[24.966, 112.0, -18.214], [128.553, -74.203, -93.786], [65.481, -37.797, 112.0]]) / 255.0 + [16, 128, 128]
    if in_type == np.uint8:
        out = out.round()
    else:
        out = out / 255.0
    return out.astype(in_type)


def calculate_psnr(img1, img2):
    """images in [0, 255]"""
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * math.log10(255.0 / math.sqrt(mse))


def _gauss_window(size=11, sigma=1.5):
    ax = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    k = np.exp(-(ax ** 2) / (2.0 * sigma ** 2))
    k /= k.sum()
    return np.outer(k, k)


def _filter_valid(img, win):
    """correlation of every channel with `win`, interior ('valid') region only"""
    k = win.shape[0]
    if img.ndim == 2:
        v = np.lib.stride_tricks.sliding_window_view(img, (k, k))
        return np.einsum('ijkl,kl->ij', v, win)
    return np.stack([_filter_valid(img[:, :, c], win) for c in range(img.shape[2])], axis=2)


def ssim(img1, img2):
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    win = _gauss_window(11, 1.5)
    mu1, mu2 = _filter_valid(a, win), _filter_valid(b, win)
    mu1_sq, mu2_sq, mu12 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _filter_valid(a ** 2, win) - mu1_sq
    s2 = _filter_valid(b ** 2, win) - mu2_sq
    s12 = _filter_valid(a * b, win) - mu12
    return (((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()


def calculate_ssim(img1, img2):
    """images in [0, 255]; HW, HW1 or HW3 (for 3 channels the reference averages three evaluations of the whole 3-channel image)"""
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    if img1.ndim == 2:
        return ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return np.array([ssim(img1, img2) for _ in range(3)]).mean()
        if img1.shape[2] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
    raise ValueError('Wrong input image dimensions.')


def forward_chop(img, scale, model, shave=20, min_size=160000):
    """Inference on four overlapping quadrants (recursively while a quadrant has >= min_size pixels), outputs stitched without the
    `shave` overlap.  `model` maps [n,3,h,w] -> [n,3,h*scale,w*scale]; the quadrants of one level are run as one batch of 4."""
    h, w = img.shape[-2:]
    top, bottom = slice(0, h // 2 + shave), slice(h - h // 2 - shave, h)
    left, right = slice(0, w // 2 + shave), slice(w - w // 2 - shave, w)
    quads = torch.cat([img[..., top, left], img[..., top, right], img[..., bottom, left], img[..., bottom, right]])
    if h * w < 4 * min_size:
        ys = list(model(quads).chunk(4, dim=0))
    else:
        b = img.shape[0]
        ys = [forward_chop(quads[i * b:(i + 1) * b], scale, model, shave=shave, min_size=min_size) for i in range(4)]
    H, W = int(round(h * scale)), int(round(w * scale))
    H += H % 2
    W += W % 2
    t, bt, bt_r = slice(0, H // 2), slice(H - H // 2, H), slice(H // 2 - H, None)
    l, r, r_r = slice(0, W // 2), slice(W - W // 2, W), slice(W // 2 - W, None)
    y = ys[0].new_zeros(ys[0].shape[:-2] + (H, W))
    y[..., t, l] = ys[0][..., t, l]
    y[..., t, r] = ys[1][..., t, r_r]
    y[..., bt, l] = ys[2][..., bt_r, l]
    y[..., bt, r] = ys[3][..., bt_r, r_r]
    return y
