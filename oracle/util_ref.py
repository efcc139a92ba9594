"""CPU restatement of the reference's validation helpers.  TEST INFRASTRUCTURE (see oracle/__init__.py).

forward_chop   : codes/SRN/utils/util.py:87-147  (quadrant inference; the reference routes each quadrant through
                 torch.nn.parallel.data_parallel on one device -- here a plain call)
tensor2img     : utils/util.py:180-204            calculate_psnr : utils/util.py:236-243
ssim           : utils/util.py:246-267            (cv2.getGaussianKernel(11, 1.5) outer product, cv2.filter2D(...)[5:-5, 5:-5])
bgr2ycbcr      : codes/SRN/data/util.py:169-190

Pinned by tests/golden/util_metrics.npz (oracle/gen_golden_util.py imports the reference functions; cv2 is absent from this image,
so ssim's two cv2 calls are served by a scipy stand-in in the generator -- a 'reflect' correlate cropped to the interior, which is what
filter2D's default border gives on the cropped region; **ssim parity is therefore pinned only up to that stand-in**).
"""
import math

import numpy as np
import torch


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float('inf') if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))


def gaussian_kernel_1d(n=11, sigma=1.5):
    x = np.arange(n, dtype=np.float64) - (n - 1) / 2
    k = np.exp(-x * x / (2 * sigma * sigma))
    return k / k.sum()


def ssim(a, b):
    from scipy.ndimage import correlate
    k = gaussian_kernel_1d()
    win = np.outer(k, k)
    a, b = a.astype(np.float64), b.astype(np.float64)

    def f(x):
        if x.ndim == 2:
            return correlate(x, win, mode='mirror')[5:-5, 5:-5]
        return np.stack([correlate(x[:, :, c], win, mode='mirror')[5:-5, 5:-5] for c in range(x.shape[2])], 2)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    m1, m2 = f(a), f(b)
    s1, s2, s12 = f(a * a) - m1 * m1, f(b * b) - m2 * m2, f(a * b) - m1 * m2
    return (((2 * m1 * m2 + C1) * (2 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s1 + s2 + C2))).mean()


def bgr2y(img):
    """float [0,1] BGR -> Y in [0,1]"""
    return (np.dot(img.astype(np.float32) * 255.0, [24.966, 128.553, 65.481]) / 255.0 + 16.0) / 255.0


def tensor2img(t):
    t = t.squeeze().float().cpu().clamp(0, 1)
    assert t.dim() == 3
    return (np.transpose(t.numpy()[[2, 1, 0]], (1, 2, 0)) * 255.0).round().astype(np.uint8)


def forward_chop(img, scale, model, shave=20, min_size=160000):
    h, w = img.shape[-2:]
    hs, ws = h // 2 + shave, w // 2 + shave
    parts = [img[..., :hs, :ws], img[..., :hs, w - ws:], img[..., h - hs:, :ws], img[..., h - hs:, w - ws:]]
    if h * w < 4 * min_size:
        outs = [model(p) for p in parts]
    else:
        outs = [forward_chop(p, scale, model, shave, min_size) for p in parts]
    H, W = round(h * scale), round(w * scale)
    H, W = H + H % 2, W + W % 2
    h2, w2 = H // 2, W // 2
    y = outs[0].new_zeros(outs[0].shape[:-2] + (H, W))
    y[..., :h2, :w2] = outs[0][..., :h2, :w2]
    y[..., :h2, W - w2:] = outs[1][..., :h2, outs[1].shape[-1] - w2:]
    y[..., H - h2:, :w2] = outs[2][..., outs[2].shape[-2] - h2:, :w2]
    y[..., H - h2:, W - w2:] = outs[3][..., outs[3].shape[-2] - h2:, outs[3].shape[-1] - w2:]
    return y
