"""LPIPS(alex, v0.1) perceptual loss as a recorded op list over the MI355X kernels: forward on [fake ; real], loss, and the data
gradient w.r.t. the fake images.

Replaces PerceptualLossLPIPS (codes/SRN/models/modules/loss.py:66-72; `feature_criterion: "LPIPS"` of the shipped train_DASR*.json,
DASR_model.py:97-98,231-233) = PNetLin.forward (codes/PerceptualSimilarity/models/networks_basic.py:64-92, version 0.1, net-lin, eval mode)
on the torchvision AlexNet slices relu1..relu5 (codes/PerceptualSimilarity/models/pretrained_networks.py:57-95), and the DSN's
`--per_type LPIPS` (codes/DSN/loss.py:82-92).

MI355X mapping: the 11x11 / stride 4 / pad 2 first conv on 3 channels is rewritten as a 3x3 / stride 1 conv on the 4x4 space-to-depth
grid (48 channels; the ScalingLayer and the [0,1] -> [-1,1] map are folded into the space-to-depth kernel), so all five convs and their
data gradients run on the split-bf16 MFMA conv kernel (prec 3, ~fp32: the loss gradient goes straight into the generator gradient, where
the 1e-2 tolerance is decided).  The per-layer heads (unit-normalise over channels, squared difference, lin weights, spatial mean) compute
the loss AND the gradient w.r.t. the fake features in one launch; pools are gather kernels (no atomics).  ~11 GFLOP per image pair
forward + backward, against ~610 for the VGG19-54 feature loss: with LPIPS the configs[2] step is ~25 % shorter than with VGG features.

Weights: `load_state_dict` takes torchvision's alexnet keys (`features.{0,3,6,8,10}.{weight,bias}`) and the reference's linear heads
(`lin{0..4}.model.1.weight`, codes/PerceptualSimilarity/models/weights/v0.1/alex.pth).  Neither can be downloaded here: without files the
caller seeds them (`lpips_random_state_dict`) and says so in the log."""
import ctypes as C
import logging
import math
from collections import OrderedDict

import torch

from . import _lib
from .engine import BTensor, ParamStore, PackRegistry, OpList, conv_op, ceil_div, NULL_T
from ._lib import Op, Tensor

SHIFT = (-.030, -.088, -.188)     # ScalingLayer, networks_basic.py:94-101
SCALE = (.458, .448, .450)
CHNS = (64, 192, 384, 256, 256)   # networks_basic.py:41-43
EPS = 1e-10                       # normalize_tensor, models/util.py:42-44
PREC = 4                          # split-f16 conv operands (f16 hi + lo pairs, 22 bits, three MFMA passes); 3 = split-bf16 (16 bits)
#          key           cout cin  k  pad  followed by a MaxPool2d(3, 2)
CONVS = (('features.0', 64, 3, 11, 2, True), ('features.3', 192, 64, 5, 2, True), ('features.6', 384, 192, 3, 1, False),
         ('features.8', 256, 384, 3, 1, False), ('features.10', 256, 256, 3, 1, False))


def _op(kind):
    o = Op()
    o.op = kind
    return o


def lpips_keys():
    spec = []
    for key, cout, cin, k, pad, pool in CONVS:
        spec += [(key + '.weight', (cout, cin, k, k)), (key + '.bias', (cout,))]
    spec += [('lin%d.model.1.weight' % i, (1, c, 1, 1)) for i, c in enumerate(CHNS)]
    return spec


def lpips_random_state_dict(seed):
    """seeded stand-in when no weight files are supplied (kaiming-normal fan_in convs, small biases, uniform [0,1) non-negative heads)"""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shape in lpips_keys():
        if k.startswith('lin'):
            sd[k] = torch.rand(shape, generator=g)
        elif k.endswith('weight'):
            sd[k] = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
        else:
            sd[k] = (torch.rand(shape, generator=g) - 0.5) * 0.1
    return sd


def conv1_to_s2d(w):
    """[64][3][11][11] -> [64][48][3][3]: tap (ky, kx) = (4a + by, 4b + bx) becomes tap (a, b) of input channel c*16 + by*4 + bx
    (zero where 4a + by > 10): the same sum of products, on the zero-padded image cut into 4x4 blocks"""
    cout = w.shape[0]
    wp = torch.zeros((cout, 3, 12, 12), dtype=w.dtype)
    wp[:, :, :11, :11] = w
    return wp.view(cout, 3, 3, 4, 3, 4).permute(0, 1, 3, 5, 2, 4).reshape(cout, 48, 3, 3).contiguous()


logger = logging.getLogger('base')


def load_lpips(opt, device, seed=77):
    """LPIPS(alex) with weights from files: path.lpips_alexnet = torchvision's alexnet state_dict (features.*), path.lpips_lin = the
    reference's codes/PerceptualSimilarity/models/weights/v0.1/alex.pth (lin*.model.1.weight).  The reference downloads / reads both itself
    (pretrained_networks.py:60, dist_model.py:74-80), i.e. it ALWAYS runs the pretrained network.  Offline the files cannot be fetched, so a
    missing file is an error unless the caller opts in with `allow_random_perceptual: true` (option file / --allow_random_perceptual): the
    network is then seeded, `net.seeded` is set and every logged value is labelled LPIPS(random) -- it is not the LPIPS metric."""
    pf, pl = opt['path']['lpips_alexnet'], opt['path']['lpips_lin']
    allow = bool(opt.get('allow_random_perceptual') or opt['path'].get('allow_random_perceptual'))
    missing = [n for n, v in (('path.lpips_alexnet', pf), ('path.lpips_lin', pl)) if not v]
    if missing and not allow:
        raise FileNotFoundError('LPIPS(alex) needs pretrained weights (%s missing): supply the files (INTEGRATION.md, "Perceptual-network weight '
                                'files") or set allow_random_perceptual to train / evaluate against a seeded random network' % ', '.join(missing))
    net = LPIPSAlexHIP(device=device)
    sd = lpips_random_state_dict(seed)
    if pf:
        src = torch.load(pf, map_location='cpu')
        sd.update({k: v for k, v in src.items() if k.startswith('features.') and k in sd})
    else:
        logger.warning('allow_random_perceptual: the LPIPS AlexNet backbone uses SEEDED RANDOM weights (no path.lpips_alexnet)')
    if pl:
        src = torch.load(pl, map_location='cpu')
        sd.update({k: v for k, v in src.items() if k.startswith('lin') and k in sd})
    else:
        logger.warning('allow_random_perceptual: the LPIPS linear heads use SEEDED RANDOM non-negative weights (no path.lpips_lin)')
    net.load_state_dict(sd)
    net.seeded = bool(missing)
    return net


def lpips_label(net):
    """column label of the validation / test logs: a seeded network does not measure LPIPS"""
    return 'LPIPS(random)' if getattr(net, 'seeded', False) else 'LPIPS'


def lpips_metric(net, fake, real):
    """validation LPIPS of the reference: on the 8-bit images (tensor2img -> im2tensor, DASR_model.py:340-344 / SR_model.py:95-99), first image of the batch"""
    q = lambda t: (t[:1].detach().float().clamp(0, 1) * 255.0).round() / 255.0
    return net.distance(q(fake), q(real))


class LPIPSAlexHIP:
    """Frozen LPIPS network.  plan(N, n, H, W): N = 2n images [fake (n) ; real (n)] of H x W (multiples of 4)."""

    def __init__(self, device='cuda'):
        self.device = torch.device(device)
        spec = [('c1.weight', (64, 48, 3, 3))]
        for key, cout, cin, k, pad, pool in CONVS:
            if key != 'features.0':
                spec.append((key + '.weight', (cout, cin, k, k)))
            spec.append((key + '.bias', (cout,)))
        spec += [('lin%d' % i, (c,)) for i, c in enumerate(CHNS)]
        self.params = ParamStore(spec, self.device)
        self.pack = PackRegistry(self.params)
        P = self.params
        self.pk = {}
        for key, cout, cin, k, pad, pool in CONVS:
            wkey, kk, cin_ = (('c1.weight', 3, 48) if key == 'features.0' else (key + '.weight', k, cin))
            w = P.off(wkey)
            self.pk[key] = self.pack.add(cout, cin_, kk * kk, 1, PREC, [(w, cout, cin_, 0, cin_, 0, 0)])
            self.pk[(key, 'b')] = self.pack.add(cin_, cout, kk * kk, 1, PREC, [(w, cout, cin_, 0, cout, 0, 1)])
        self.pack.finalize()
        self._sd = None
        self.plans = {}

    def load_state_dict(self, sd, strict=True):
        want = [k for k, _ in lpips_keys()]
        missing = [k for k in want if k not in sd]
        if strict and missing:
            raise RuntimeError('Error(s) in loading state_dict for LPIPS(alex): missing %s' % missing)
        own = {}
        for k, shape in lpips_keys():
            if k not in sd:
                continue
            v = sd[k].detach().float().cpu()
            if tuple(v.shape) != tuple(shape):
                raise RuntimeError('size mismatch for %s: %s vs %s' % (k, tuple(v.shape), tuple(shape)))
            if k == 'features.0.weight':
                own['c1.weight'] = conv1_to_s2d(v)
            elif k.startswith('lin'):
                own['lin' + k[3]] = v.reshape(-1)
            else:
                own[k] = v
        self.params.load_state_dict(own, strict=False)
        self._sd = OrderedDict((k, sd[k].detach().float().cpu().clone()) for k in want if k in sd)
        self.pack.run()

    def state_dict(self):
        return OrderedDict(self._sd or {})

    def plan(self, N, n, H, W):
        k = (N, n, H, W)
        if k not in self.plans:
            self.plans[k] = _LPIPSPlan(self, N, n, H, W)
        return self.plans[k]

    def distance(self, a, b):
        """LPIPS distance of two image batches [n][3][H][W] in [0, 1] (device tensors), mean over the batch: the validation metric
        (DASR_model.py:340-344 / SR_model.py:95-99: `cri_fea_lpips(im2tensor(fake), im2tensor(real))`, whose im2tensor maps uint8 to
        [-1, 1]: the caller passes the 8-bit-quantised images / 255).  Plans of the last two image sizes are kept."""
        n, c, H, W = a.shape
        if c != 3 or tuple(b.shape) != tuple(a.shape) or H % 4 or W % 4:
            raise ValueError('LPIPS distance: two [n,3,H,W] batches with H, W multiples of 4, got %s / %s' % (tuple(a.shape), tuple(b.shape)))
        key = ('val', n, H, W)
        lru = self.__dict__.setdefault('_val_lru', OrderedDict())
        if key in lru:
            lru.move_to_end(key)
        else:
            p = _LPIPSPlan(self, 2 * n, n, H, W)
            img = BTensor(2 * n, 16, H, W, True, self.device)
            nchw = torch.zeros((2 * n, 3, H, W), dtype=torch.float32, device=self.device)
            acc = torch.zeros(1, dtype=torch.float32, device=self.device)
            ops = OpList()
            o = _op(_lib.OP_FILL)
            o.p[0], o.l[0], o.f[0] = acc.data_ptr(), 1, 0.0
            ops.add(o)
            o = _op(_lib.OP_NCHW2B)
            o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[0], o.t[1] = nchw.data_ptr(), 2 * n, 3, H, W, img.view(), NULL_T
            ops.add(o)
            ops.add(p.input_op(img.view(), 0, 2 * n))
            ops.extend(p.fwd)
            for o in p.head_ops(acc.data_ptr(), 0.0):
                ops.add(o)
            ops.keep += [p, img, nchw, acc]
            lru[key] = (ops, nchw, acc)
            while len(lru) > 2:
                lru.popitem(last=False)
        ops, nchw, acc = lru[key]
        nchw[:n].copy_(a)
        nchw[n:].copy_(b)
        ops.run()
        return acc[0].clone()


class _LPIPSPlan:
    """fwd: input_ops (image -> x), convs / pools, heads (loss into loss_ptr, head gradients);  bwd: data gradient down to gx_s2d, and
    `adjoint_op(dst)` that accumulates it into a blocked image gradient."""

    def __init__(self, net, N, n, H, W):
        assert H % 4 == 0 and W % 4 == 0 and N >= 2 * n, (N, n, H, W)
        self.net, self.N, self.n, self.H, self.W = net, N, n, H, W
        dev, P, pack = net.device, net.params, net.pack
        Hs, Ws = (H + 4) // 4, (W + 4) // 4
        self.x = BTensor(N, 48, Hs, Ws, True, dev)
        self.relu, self.pool, self.dims = [], [], []
        fwd = OpList()
        src, h, w, cin = self.x, Hs, Ws, 48
        for key, cout, _, k, pad, pool in CONVS:
            first = key == 'features.0'
            kk, pd = (3, 0) if first else (k, pad)
            ho, wo = (h - 2, w - 2) if first else (h, w)
            out = BTensor(N, cout, ho, wo, True, dev)
            fwd.add(conv_op(pack, net.pk[key], src.view(), True, cin, h, w, ho, wo, N, bias=P.ptr(key + '.bias'), kh=kk, pad=pd, act=1, slope=0.0,
                            out_f32=out.view(), flops=2.0 * N * ho * wo * k * k * (3 if first else cin) * cout))
            self.relu.append(out)
            self.dims.append((ho, wo))
            src, h, w, cin = out, ho, wo, cout
            if pool:
                hp, wp = (h - 3) // 2 + 1, (w - 3) // 2 + 1
                pl = BTensor(N, cout, hp, wp, True, dev)
                o = _op(_lib.OP_MAXPOOL3)
                o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1] = src.view(), N, cout, h, w, pl.view()
                fwd.add(o)
                self.pool.append(pl)
                src, h, w = pl, hp, wp
            else:
                self.pool.append(None)
        self.fwd = fwd.tag(6)
        # head gradients (w.r.t. the pre-activations of relu_k, fake images only); the data-gradient chain adds into / reads them
        self.ghead = [BTensor(n, r.C, r.H, r.W, True, dev) for r in self.relu]
        self.gx = BTensor(n, 48, Hs, Ws, True, dev)
        # split-f16: the head gradients (~1 / (n * pixels)) are pre-scaled by a power of two into f16's normal range (exact, undone on the accumulator)
        gsc = float(2.0 ** max(0, int(math.floor(math.log2(max(1, n * H * W // 16)))) - 3)) if PREC == 4 else 0.0
        bwd = OpList()
        g = self.ghead[4]                                   # total gradient at relu5 = its head gradient
        for li in range(4, 0, -1):
            key, cout, cin_, k, pad, _ = CONVS[li]
            below = self.relu[li - 1]
            if CONVS[li - 1][5]:                            # relu_{li} -> pool -> conv: dgrad to the pooled grid, pool backward ADDS into the head gradient
                pl = self.pool[li - 1]
                gp = BTensor(n, pl.C, pl.H, pl.W, True, dev)
                bwd.add(conv_op(pack, net.pk[(key, 'b')], g.view(), True, cout, pl.H, pl.W, pl.H, pl.W, n, kh=k, pad=pad, out_f32=gp.view(), in_scale=gsc))
                o = _op(_lib.OP_MAXPOOL3_BWD)
                o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.t[2], o.i[4], o.i[5] = below.view(), gp.view(), n, below.C, below.H, below.W, \
                    self.ghead[li - 1].view(), 1, 1
                bwd.add(o)
                bwd.keep.append(gp)
                g = self.ghead[li - 1]
            else:                                           # relu -> conv: ReLU' mask in the epilogue, head gradient added behind it
                gt = BTensor(n, below.C, below.H, below.W, True, dev)
                bwd.add(conv_op(pack, net.pk[(key, 'b')], g.view(), True, cout, below.H, below.W, below.H, below.W, n, kh=k, pad=pad,
                                mask=below.view(), mask_f32=1, slope=0.0, res1=self.ghead[li - 1].view(), beta1=1.0, out_f32=gt.view(), in_scale=gsc))
                bwd.keep.append(gt)
                g = gt
        r1 = self.relu[0]
        bwd.add(conv_op(pack, net.pk[('features.0', 'b')], g.view(), True, 64, r1.H, r1.W, Hs, Ws, n, kh=3, pad=2, out_f32=self.gx.view(),
                        flops=2.0 * n * r1.H * r1.W * 121 * 3 * 64, in_scale=gsc))
        self.bwd = bwd.tag(8)
        self._sc = [2.0 / s for s in SCALE] + [0.0]
        self._sh = [-(1.0 + sh) / s for sh, s in zip(SHIFT, SCALE)] + [0.0]

    def _s2d(self, img_view, n_imgs, y_view, mode):
        o = _op(_lib.OP_LPIPS_S2D)
        o.t[0], o.i[0], o.i[1], o.i[2], o.t[1], o.i[3] = img_view, n_imgs, self.H, self.W, y_view, mode
        for j in range(4):
            o.f[j] = self._sc[j]
        C.memmove(C.addressof(o.l), (C.c_float * 4)(*self._sh), 16)
        return o

    def input_op(self, img_view, n0, n_imgs):
        """blocked f32 image (3 channels in plane 0, values in [0,1]) -> scaled space-to-depth input of images [n0, n0 + n_imgs)"""
        v = self.x.view()
        return self._s2d(img_view, n_imgs, Tensor(v.p + n0 * v.n_stride * 4, v.n_stride, v.cb_stride), 0)

    def adjoint_op(self, gimg_view):
        """dL/dx of the n fake images ACCUMULATED into channels 0..2 of a blocked f32 image gradient"""
        return self._s2d(gimg_view, self.n, self.gx.view(), 1)

    def head_ops(self, loss_ptr, weight):
        """loss_ptr (device float*) += mean over the n pairs of the LPIPS distance; head gradients of weight * that mean"""
        ops = []
        P = self.net.params
        for i, r in enumerate(self.relu):
            o = _op(_lib.OP_LPIPS_HEAD)
            cnt = float(self.n * r.H * r.W)
            o.t[0], o.l[0], o.i[0], o.i[1], o.i[2], o.i[3] = r.view(), self.n, self.n, r.C, r.H, r.W
            o.p[0], o.f[0], o.f[1], o.f[2], o.p[1], o.t[1], o.i[4] = P.ptr('lin%d' % i), EPS, 1.0 / cnt, float(weight) / cnt, loss_ptr, self.ghead[i].view(), 1
            ops.append(o)
        return ops
