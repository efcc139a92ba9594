"""CPU restatement (fp32 torch) of the LPIPS(alex, v0.1) perceptual loss the reference trains with when
`feature_criterion == "LPIPS"` (the shipped train_DASR*.json):

  PerceptualLossLPIPS          codes/SRN/models/modules/loss.py:66-72      forward(x, y) = net(x, y, normalize=True).mean()
  PerceptualLoss.forward       codes/PerceptualSimilarity/models/util.py:26-40   normalize: 2*t - 1, then DistModel.forward(target, pred)
  PNetLin.forward              codes/PerceptualSimilarity/models/networks_basic.py:64-92   (version '0.1': ScalingLayer; lpips=True, spatial=False)
  ScalingLayer / NetLinLayer   networks_basic.py:94-112                    (Dropout is the identity: the net is in eval(), dist_model.py:93)
  normalize_tensor             codes/PerceptualSimilarity/models/util.py:42-44  (eps 1e-10 added to the NORM)
  alexnet slices               codes/PerceptualSimilarity/models/pretrained_networks.py:57-95 (torchvision alexnet.features[0:12])

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
PARITY: the wrapper arithmetic (scaling, slices, normalisation, lin heads, averaging) is pinned against the reference modules imported from
/root/reference with the real linear-head weights (weights/v0.1/alex.pth) -> tests/golden/lpips_alex.npz (oracle/gen_golden_lpips.py).
The BACKBONE weights are unpinned: pretrained torchvision AlexNet needs a download; fixtures use a seeded random AlexNet.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SHIFT = (-.030, -.088, -.188)
SCALE = (.458, .448, .450)
CHNS = (64, 192, 384, 256, 256)
RELU_IDX = (1, 4, 7, 9, 11)   # alexnet.features indices whose outputs are relu1..relu5


def alexnet_features():
    """torchvision.models.alexnet().features (the architecture is part of torchvision, absent here; restated from its public definition)"""
    return nn.Sequential(
        nn.Conv2d(3, 64, kernel_size=11, stride=4, padding=2), nn.ReLU(inplace=False), nn.MaxPool2d(kernel_size=3, stride=2),
        nn.Conv2d(64, 192, kernel_size=5, padding=2), nn.ReLU(inplace=False), nn.MaxPool2d(kernel_size=3, stride=2),
        nn.Conv2d(192, 384, kernel_size=3, padding=1), nn.ReLU(inplace=False),
        nn.Conv2d(384, 256, kernel_size=3, padding=1), nn.ReLU(inplace=False),
        nn.Conv2d(256, 256, kernel_size=3, padding=1), nn.ReLU(inplace=False), nn.MaxPool2d(kernel_size=3, stride=2))


def alexnet_init_(features, seed):
    """seeded stand-in for the pretrained weights: kaiming-normal (fan_in, relu) weights, small non-zero biases"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in features:
            if isinstance(m, nn.Conv2d):
                fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
                m.bias.copy_((torch.rand(m.bias.shape, generator=g) - 0.5) * 0.1)
    return features


def seeded_lin(seed):
    """non-negative stand-in linear heads (the real ones ship with the reference: weights/v0.1/alex.pth, held as data in the golden fixture)"""
    g = torch.Generator().manual_seed(seed)
    return [torch.rand(c, generator=g) for c in CHNS]


class LPIPSAlex(nn.Module):
    def __init__(self, features=None, lin=None, seed=91):
        super().__init__()
        self.features = features if features is not None else alexnet_init_(alexnet_features(), seed)
        lin = lin if lin is not None else seeded_lin(seed + 1)
        self.lin = nn.ParameterList([nn.Parameter(w.reshape(1, -1, 1, 1).clone().float(), requires_grad=False) for w in lin])
        self.register_buffer('shift', torch.tensor(SHIFT)[None, :, None, None])
        self.register_buffer('scale', torch.tensor(SCALE)[None, :, None, None])
        for p in self.features.parameters():
            p.requires_grad = False

    def slices(self, x):
        outs, h = [], (x - self.shift) / self.scale
        for i, m in enumerate(self.features):
            if i > RELU_IDX[-1]:
                break
            h = m(h)
            if i in RELU_IDX:
                outs.append(h)
        return outs

    def forward(self, in0, in1):
        """inputs in [-1, 1]; returns [N, 1, 1, 1]"""
        val = 0
        for f0, f1, w in zip(self.slices(in0), self.slices(in1), self.lin):
            n0 = f0 / (torch.sqrt(torch.sum(f0 ** 2, dim=1, keepdim=True)) + 1e-10)
            n1 = f1 / (torch.sqrt(torch.sum(f1 ** 2, dim=1, keepdim=True)) + 1e-10)
            val = val + F.conv2d((n0 - n1) ** 2, w).mean([2, 3], keepdim=True)
        return val


class PerceptualLossLPIPS(nn.Module):
    """loss.py:66-72: images in [0, 1]"""

    def __init__(self, net=None):
        super().__init__()
        self.net = net if net is not None else LPIPSAlex()

    def forward(self, x, y):
        return self.net(2 * y - 1, 2 * x - 1).mean()


def golden_lin(golden_dir=None):
    """the reference's linear heads (weights/v0.1/alex.pth), held as data in tests/golden/lpips_alex.npz"""
    import os
    import numpy as np
    golden_dir = golden_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
    g = np.load(os.path.join(golden_dir, 'lpips_alex.npz'))
    return [torch.from_numpy(g['lin%d' % i]) for i in range(5)]


def golden_criterion(seed, golden_dir=None):
    """what the reference builds for `feature_criterion: LPIPS` under oracle.ref_import's stand-in AlexNet(seed): (criterion, state_dict in
    the key layout the product loads: torchvision alexnet `features.*` + `lin*.model.1.weight`)"""
    feats, lin = alexnet_init_(alexnet_features(), seed), golden_lin(golden_dir)
    sd = {'features.' + k: v.clone() for k, v in feats.state_dict().items()}
    sd.update({'lin%d.model.1.weight' % i: w.reshape(1, -1, 1, 1).clone() for i, w in enumerate(lin)})
    return PerceptualLossLPIPS(LPIPSAlex(feats, lin)), sd
