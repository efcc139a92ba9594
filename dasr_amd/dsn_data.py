"""Image-folder datasets of the DSN training driver (the caller side of codes/DSN/train.py:81-121): the host half of the hand-off to
`DSNModel.iteration(hr, bicubic_lr, real_lr)`.

    TrainDeresnetDataset   codes/DSN/data_loader.py:12-59    (clean HR crop, its bicubic x1/4 image, a crop/4 crop of a source-domain image)
    ValDeresnetDataset     codes/DSN/data_loader.py:157-190  (HR centre crop, its bicubic x1/4 image, a random and the centre crop of the paired LR)
    imresize               codes/DSN/utils.py:37-160         (MATLAB-style bicubic with antialiasing, clamped to [0, 1])
    display_transform      codes/DSN/utils.py:25-31          (validation image strips: Resize(400) + CenterCrop(400))

The reference builds these on PIL + torchvision transforms; torchvision is not a dependency here.  Images are decoded with PIL and handled as
CHW float tensors in [0, 1] (`to_tensor` semantics); flips / crops / quarter-turn rotations are tensor ops driven by python's `random` (the
reference's transforms draw from torch's and python's global generators: the crops are random either way, no stream is reproduced).  `imresize`
is evaluated as two dense resampling matrices (rows = output pixels, symmetric boundary folded into the columns) instead of the reference's
per-row loops: same weights, same result up to fp32 summation order (tests/test_dsn_data.py pins it against vectors of the reference function).
"""
import math
import os
import random

import numpy as np
import torch

IMG_EXTENSIONS = ('.png', '.jpg', '.jpeg', '.PNG', '.JPG', '.JPEG')
DERESNET_DATASETS = ('aim2019', 'ntire2020', 'realsr', 'camerasr')   # codes/DSN/train.py:84-115: the branches whose loaders return (hr, bicubic, real) triples


def is_image_file(filename):
    return filename.endswith(IMG_EXTENSIONS)


def calculate_valid_crop_size(crop_size, upscale_factor):
    return crop_size - (crop_size % upscale_factor)


def _cubic(x):
    a = x.abs()
    a2, a3 = a * a, a * a * a
    return (1.5 * a3 - 2.5 * a2 + 1) * (a <= 1).to(x.dtype) + (-0.5 * a3 + 2.5 * a2 - 4 * a + 2) * ((a > 1) & (a <= 2)).to(x.dtype)


def resize_matrix(in_length, scale, antialiasing=True, dtype=torch.float32):
    """[out_length, in_length] matrix of the 1-D bicubic resampling of utils.py:46-98: output pixel k takes the kernel (stretched by 1 / scale when
    shrinking with antialiasing) centred on u = k / scale + 0.5 (1 - 1 / scale), weights normalised per output pixel, taps outside the image
    mirrored back in (symmetric extension including the edge pixel)."""
    out_length = math.ceil(in_length * scale)
    kernel_width = 4.0
    shrink = scale < 1 and antialiasing
    if shrink:
        kernel_width = kernel_width / scale
    x = torch.arange(1, out_length + 1, dtype=dtype)
    u = x / scale + 0.5 * (1 - 1 / scale)
    left = torch.floor(u - kernel_width / 2)
    P = math.ceil(kernel_width) + 2
    idx = left[:, None] + torch.arange(P, dtype=dtype)[None, :]            # 1-based input coordinates of the taps
    dist = u[:, None] - idx
    w = scale * _cubic(dist * scale) if shrink else _cubic(dist)
    w = w / w.sum(1, keepdim=True)
    j = idx.long() - 1                                                     # 0-based, may leave [0, in_length)
    j = torch.where(j < 0, -j - 1, j)                                      # symmetric: -1 -> 0, -2 -> 1
    j = torch.where(j >= in_length, 2 * in_length - 1 - j, j)              # in_length -> in_length - 1
    j = j.clamp(0, in_length - 1)                                          # (zero-weight taps further out)
    R = torch.zeros(out_length, in_length, dtype=dtype)
    R.scatter_add_(1, j, w)
    return R


def imresize(img, scale, antialiasing=True):
    """img CHW in [0, 1] -> CHW [C, ceil(H scale), ceil(W scale)], rows first then columns, clamped to [0, 1] (utils.py:101-160)"""
    img = img.float()
    Rh = resize_matrix(img.shape[1], scale, antialiasing)
    Rw = resize_matrix(img.shape[2], scale, antialiasing)
    out = torch.matmul(torch.matmul(Rh, img), Rw.t())
    return out.clamp(0, 1)


def open_image(path):
    """file -> CHW fp32 RGB in [0, 1] (Image.open + to_tensor; grey / palette / alpha files are converted to RGB)"""
    from PIL import Image
    with Image.open(path) as im:
        a = np.array(im.convert('RGB'), dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).float().div_(255.0)


def _list_images(dirs):
    if isinstance(dirs, str):
        dirs = [dirs]
    files = []
    for d in dirs:
        files += [os.path.join(d, x) for x in os.listdir(d) if is_image_file(x)]
    return files


def random_crop(img, size):
    """T.RandomCrop(size) on a CHW tensor"""
    _, h, w = img.shape
    if h < size or w < size:
        raise ValueError('image %dx%d is smaller than the crop size %d' % (h, w, size))
    y, x = random.randint(0, h - size), random.randint(0, w - size)
    return img[:, y:y + size, x:x + size]


def center_crop(img, size):
    """T.CenterCrop(size) on a CHW tensor (torchvision rounds the offsets)"""
    _, h, w = img.shape
    if h < size or w < size:
        raise ValueError('image %dx%d is smaller than the crop size %d' % (h, w, size))
    y, x = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return img[:, y:y + size, x:x + size]


def load_augmented_crop(path, crop_size, flips, rotations):
    """RandomVerticalFlip, RandomHorizontalFlip (p = 0.5 when `flips`), RandomCrop(crop_size), then a quarter-turn by a random multiple of 90 degrees
    when `rotations` (TF.rotate of the square crop, counter-clockwise): data_loader.py:27-31,43-47 -- without converting the whole image to floats:
    the crop window is cut from the decoded 8-bit image first and the flips are applied to the crop (a uniformly random window of the flipped
    image = the flip of a uniformly random window)"""
    from PIL import Image
    with Image.open(path) as im:
        w, h = im.size
        if h < crop_size or w < crop_size:
            raise ValueError('%s: image %dx%d is smaller than the crop size %d' % (path, h, w, crop_size))
        y, x = random.randint(0, h - crop_size), random.randint(0, w - crop_size)
        a = np.array(im.convert('RGB').crop((x, y, x + crop_size, y + crop_size)), dtype=np.uint8)
    img = torch.from_numpy(a).permute(2, 0, 1).float().div_(255.0)
    if flips and random.random() < 0.5:
        img = img.flip(1)
    if flips and random.random() < 0.5:
        img = img.flip(2)
    if rotations:
        img = torch.rot90(img, random.choice([0, 1, 2, 3]), (1, 2))
    return img.contiguous()


class TrainDeresnetDataset:
    """item -> (clean HR crop [3,c,c], its bicubic x1/upscale image, crop [3,c/up,c/up] of a source-domain image): data_loader.py:12-59 with
    `cropped=True`, as train.py:85-111 builds it (`noisy_dir` = the `source` folder(s), `cleandir` = the `target` folder(s) of paths.yml).
    The clean image is drawn at random per item, the epoch length is the number of source images."""

    def __init__(self, noisy_dir, cleandir, crop_size, upscale_factor=4, cropped=False, flips=False, rotations=False, **kwargs):
        self.noisy_dir_files = _list_images(noisy_dir)
        self.cleandir_files = _list_images(cleandir)
        if not self.noisy_dir_files or not self.cleandir_files:
            raise FileNotFoundError('no image files under %s / %s' % (noisy_dir, cleandir))
        self.crop_size, self.upscale_factor = int(crop_size), int(upscale_factor)
        self.cropped, self.flips, self.rotations = bool(cropped), bool(flips), bool(rotations)

    def __len__(self):
        return len(self.noisy_dir_files)

    def __getitem__(self, index):
        index_clean = np.random.randint(0, len(self.cleandir_files))
        noisy = load_augmented_crop(self.noisy_dir_files[index], self.crop_size, self.flips, self.rotations)
        clean = load_augmented_crop(self.cleandir_files[index_clean], self.crop_size, self.flips, self.rotations)
        resized = imresize(clean, 1.0 / self.upscale_factor, True)
        if self.cropped:
            return clean, resized, random_crop(noisy, self.crop_size // self.upscale_factor).contiguous()
        return resized


class ValDeresnetDataset:
    """item -> (HR centre crop, its bicubic x1/upscale image, random LR crop, centre LR crop) of the sorted file pair `index`:
    data_loader.py:157-190.  cs = min(w, h) rounded down to a multiple of the factor, capped by crop_size_val."""

    def __init__(self, hr_dir, upscale_factor, lr_dir=None, crop_size_val=None, **kwargs):
        self.hr_files = sorted(_list_images(hr_dir))
        self.lr_files = None if lr_dir is None else sorted(_list_images(lr_dir))
        self.upscale_factor, self.crop_size = int(upscale_factor), crop_size_val
        if self.lr_files is None:
            raise NotImplementedError('Val_Deresnet_Dataset without lr_dir returns an undefined name in the reference (data_loader.py:180-181)')

    def __len__(self):
        return len(self.hr_files)

    def __getitem__(self, index):
        hr = open_image(self.hr_files[index])
        cs = calculate_valid_crop_size(min(hr.shape[1], hr.shape[2]), self.upscale_factor)
        if self.crop_size is not None:
            cs = min(cs, int(self.crop_size))
        hr = center_crop(hr, cs).contiguous()
        resized = imresize(hr, 1.0 / self.upscale_factor, True)
        lr = open_image(self.lr_files[index])
        return hr, resized, random_crop(lr, cs // self.upscale_factor).contiguous(), center_crop(lr, cs // self.upscale_factor).contiguous()


class ShardSampler:
    """sampler of torch.utils.data.DataLoader: a new permutation per epoch (shuffle=True of train.py:87), of which data-parallel rank r walks the
    strided share r, r + world, ... (every rank draws the SAME permutation from `seed + epoch`; equal item counts on every rank)"""

    def __init__(self, n, shuffle=True, seed=0, rank=0, world=1):
        self.n, self.shuffle, self.seed, self.rank, self.world, self.epoch = n, shuffle, seed, rank, world, 0

    def __len__(self):
        return self.n // self.world if self.world > 1 else self.n

    def __iter__(self):
        order = list(range(self.n))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(order)
        self.epoch += 1
        if self.world > 1:   # every rank must run the same number of iterations (collectives): the n % world left-over items sit out this epoch
            order = order[:self.n // self.world * self.world]
        return iter(order[self.rank::self.world])


def make_loader(dataset, batch_size, shuffle, num_workers=0, seed=0, rank=0, world=1):
    """DataLoader(dataset, num_workers, batch_size, shuffle) of train.py:87,90 (drop_last=False: the last batch of an epoch may be short); under data
    parallelism the per-rank batch is batch_size // world"""
    from torch.utils.data import DataLoader
    return DataLoader(dataset, batch_size=max(1, int(batch_size) // world), sampler=ShardSampler(len(dataset), shuffle, seed, rank, world),
                      num_workers=int(num_workers))


def load_paths(path):
    import yaml
    with open(path, 'r') as f:
        return yaml.safe_load(f)


def make_datasets(o, paths):
    """train.py:84-115: the four `Train_Deresnet_Dataset` branches.  camerasr takes its target folder from aim2019 (train.py:110)"""
    ds = o.dataset
    if ds not in DERESNET_DATASETS:
        raise NotImplementedError("dataset [%s]: the reference's training loop unpacks (hr, bicubic, real) triples, which only its aim2019 / ntire2020 / "
                                  "realsr / camerasr branches provide (train.py:84-115, 204); built in here: those four and 'synthetic'" % ds)
    src = paths[ds][o.artifacts]
    target = paths['aim2019'][o.artifacts]['target'] if ds == 'camerasr' else src['target']
    kw = dict(crop_size=o.crop_size, upscale_factor=o.upscale_factor, flips=o.flips, rotations=o.rotations)
    train_set = TrainDeresnetDataset(src['source'], target, cropped=True, **kw)
    val_set = ValDeresnetDataset(src['valid_hr'], o.upscale_factor, lr_dir=src['valid_lr'], crop_size_val=o.crop_size_val)
    return train_set, val_set


def display_transform(img, size=400):
    """utils.py:25-31: ToPILImage -> Resize(400) (shorter side, bilinear) -> CenterCrop(400) -> ToTensor, on a CHW tensor in [0, 1]"""
    from PIL import Image
    a = img.detach().float().cpu().mul(255).byte().permute(1, 2, 0).numpy()   # ToPILImage: mul(255).byte()
    im = Image.fromarray(a if a.shape[2] == 3 else a[:, :, 0])
    w, h = im.size
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    im = im.resize((nw, nh), Image.BILINEAR)
    x0, y0 = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    im = im.crop((x0, y0, x0 + size, y0 + size))
    a = np.array(im.convert('RGB'), dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).float().div_(255.0)
