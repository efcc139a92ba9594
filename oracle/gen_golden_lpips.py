"""Generate tests/golden/lpips_alex.npz by running the REFERENCE LPIPS loss (codes/SRN/models/modules/loss.py:66-72 ->
codes/PerceptualSimilarity/models/{util,dist_model,networks_basic,pretrained_networks}.py), imported from /root/reference.
TEST INFRASTRUCTURE (see oracle/__init__.py).   python -m oracle.gen_golden_lpips

torchvision is absent: `torchvision.models.alexnet` is replaced by a stand-in with oracle.lpips' restated architecture and SEEDED
weights (pretrained weights need a download: backbone numerics unpinned).  Everything else is the reference's own code,
including its loading of the real linear heads from weights/v0.1/alex.pth; those 1152 non-negative weights are stored in the fixture as
data (the product reads them from the user's reference checkout: INTEGRATION.md)."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

from . import fixtures, lpips, ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
SEED = 91
CASES = {'a': dict(n=2, hw=(64, 64), seed=5), 'b': dict(n=1, hw=(72, 100), seed=6)}


def lpips_batch(c):
    g = torch.Generator().manual_seed(c['seed'])
    h, w = c['hw']
    y = torch.rand(c['n'], 3, h, w, generator=g)
    x = (y + 0.25 * (torch.rand(c['n'], 3, h, w, generator=g) - 0.5)).clamp(0, 1)   # a distorted copy, like an SR output next to its HR target
    return x, y


def main():
    ref_import.install_stubs()

    class _Alex(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = lpips.alexnet_init_(lpips.alexnet_features(), SEED)

    tv = sys.modules['torchvision']
    tv.models.alexnet = lambda pretrained=False: _Alex()
    tv.models.vgg16 = lambda pretrained=False: None
    tv.models.squeezenet1_1 = lambda pretrained=False: None
    tv.models.resnet18 = tv.models.resnet34 = tv.models.resnet50 = tv.models.resnet101 = tv.models.resnet152 = None
    sys.modules['skimage.measure'].compare_ssim = None
    for p in (os.path.join(ref_import.REF_ROOT, 'SRN'), ref_import.REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import models.modules.loss as ref_loss
    crit = ref_loss.PerceptualLossLPIPS()          # use_gpu = torch.cuda.is_available() = False here
    net = crit.loss_network.model.net
    net = getattr(net, 'module', net)
    lin = [l.model[1].weight.detach().reshape(-1).clone() for l in net.lins]
    out = {'lin%d' % i: w.numpy() for i, w in enumerate(lin)}
    mine = lpips.PerceptualLossLPIPS(lpips.LPIPSAlex(lpips.alexnet_init_(lpips.alexnet_features(), SEED), lin))
    for name, c in CASES.items():
        x, y = lpips_batch(c)
        x.requires_grad_(True)
        l = crit(x, y)
        gx, = torch.autograd.grad(l, x)
        per = crit.loss_network.forward(x.detach(), y, normalize=True).reshape(-1)
        x2 = x.detach().clone().requires_grad_(True)
        l2 = mine(x2, y)
        g2, = torch.autograd.grad(l2, x2)
        assert abs(float(l) - float(l2)) <= 1e-6 * abs(float(l)), (float(l), float(l2))
        assert float((gx - g2).norm() / gx.norm()) < 1e-5
        out[name + '_loss'] = np.array([float(l)])
        out[name + '_per_image'] = per.detach().numpy()
        out[name + '_gx_sub'] = fixtures.subsample(gx).numpy()
        out[name + '_gx_norm'] = np.array([float(gx.double().norm())])
        print(name, float(l), per.tolist(), float(gx.norm()))
    np.savez_compressed(os.path.join(OUT, 'lpips_alex.npz'), **out)
    print('wrote lpips_alex.npz')


if __name__ == '__main__':
    main()
